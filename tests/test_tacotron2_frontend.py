"""Host-side input pipeline of Tacotron2 / WaveGlow against fixtures the REFERENCE's own code produced
(tests/golden/tacotron2_frontend.npz, oracle/make_golden.py gen_tacotron2_frontend): text_to_sequence, TextMelCollate, |STFT|.
The Slaney mel filter bank has no reference fixture (librosa is absent where the fixtures are made): checked against its
definition only.  CPU only."""
import os
import types

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "tacotron2_frontend.npz"))


def test_symbol_table_and_text_to_sequence_match_the_reference():
    from deeplearningexamples_amd.tacotron2 import text as T
    from oracle.make_golden import FRONTEND_SENTENCES
    assert T.symbols == [str(s) for s in GOLD["symbols"]] and len(T.symbols) == 148
    for ci, cleaners in enumerate((["english_cleaners"], ["basic_cleaners"])):
        for si, sent in enumerate(FRONTEND_SENTENCES):
            assert T.text_to_sequence(sent, cleaners) == GOLD["seq.%d.%d" % (ci, si)].tolist(), (cleaners, sent)
    seq = T.text_to_sequence("Turn left on {HH AW1 S S T AH0 N} Street.", ["english_cleaners"])
    assert T.sequence_to_text(seq) == "turn left on {HH AW1 S S T AH0 N} street."
    with pytest.raises(ValueError, match="inflect"):
        T.text_to_sequence("in 1984", ["english_cleaners"])
    with pytest.raises(ValueError, match="unidecoder"):
        T.text_to_sequence("café", ["english_cleaners"])
    with pytest.raises(Exception, match="Unknown cleaner"):
        T.text_to_sequence("x", ["nope"])


@pytest.mark.parametrize("nf", [1, 3])
def test_collate_matches_the_reference(nf):
    from deeplearningexamples_amd.tacotron2.data_function import TextMelCollate
    rng = np.random.default_rng(41)
    lens, mels = [7, 12, 3, 12, 9], [19, 31, 8, 25, 31]
    batch = [(torch.from_numpy(rng.integers(1, 148, l).astype(np.int32)), torch.from_numpy(rng.standard_normal((5, m)).astype(np.float32)),
              l + 2) for l, m in zip(lens, mels)]
    out = TextMelCollate(nf)(batch)
    for k, t in zip(("text", "input_lengths", "mel", "gate", "output_lengths", "len_x"), out):
        ref = GOLD["collate%d.%s" % (nf, k)]
        assert t.numpy().dtype == ref.dtype and np.array_equal(t.numpy(), ref), k
    assert out[2].shape[2] % nf == 0


def test_stft_magnitudes_match_the_reference_class():
    from deeplearningexamples_amd.tacotron2.audio import TacotronSTFT
    mag = TacotronSTFT().magnitudes(torch.from_numpy(GOLD["wav"])[None])[0].numpy()
    assert mag.shape == (513, 20)
    np.testing.assert_allclose(mag[::8], GOLD["stft_mag"], rtol=2e-4, atol=2e-4)


def test_mel_filter_bank_follows_its_definition():
    from deeplearningexamples_amd.tacotron2.audio import TacotronSTFT, _hz_to_mel, _mel_to_hz, mel_filter_bank
    assert abs(float(_hz_to_mel(1000.0)) - 15.0) < 1e-12 and abs(float(_mel_to_hz(15.0)) - 1000.0) < 1e-9
    assert abs(float(_hz_to_mel(6400.0)) - 42.0) < 1e-9                     # 27 log-spaced steps per factor 6.4 above 1 kHz
    fb = mel_filter_bank(22050, 1024, 80, 0.0, 8000.0)
    assert fb.shape == (80, 513) and fb.min() >= 0
    freqs = np.linspace(0, 11025, 513)
    assert fb[:, freqs > 8000.0].max() == 0 and (fb.sum(1) > 0).all()
    peaks = freqs[fb.argmax(1)]
    assert (np.diff(peaks) > 0).all()                                     # centres increase; at most two triangles overlap per bin
    assert ((fb > 0).sum(0) <= 2).all()
    # area normalisation: every triangle integrates to ~1 over frequency (bin width 22050 / 1024 Hz), up to sampling of the triangle
    area = fb.sum(1) * (22050 / 1024)
    assert np.all(np.abs(area - 1.0) < 0.25) and abs(float(np.median(area)) - 1.0) < 0.02
    s = TacotronSTFT()
    y = torch.from_numpy(GOLD["wav"])[None]
    mel = s.mel_spectrogram(y)
    assert mel.shape == (1, 80, 20) and float(mel.min()) >= np.log(1e-5) - 1e-6
    with pytest.raises(ValueError):
        s.mel_spectrogram(y * 3)


def test_loaders_read_filelists_wavs_and_mels(tmp_path):
    from scipy.io.wavfile import write
    from deeplearningexamples_amd.tacotron2.data_function import TextMelCollate, TextMelLoader, batch_to_gpu
    from deeplearningexamples_amd.waveglow.data_function import MelAudioLoader
    rng = np.random.default_rng(3)
    os.makedirs(tmp_path / "wavs"); os.makedirs(tmp_path / "mels")
    lines_wav, lines_mel = [], []
    for i, n in enumerate((6000, 3000, 9000)):
        write(str(tmp_path / "wavs" / ("a%d.wav" % i)), 22050, (rng.standard_normal(n) * 8000).astype(np.int16))
        torch.save(torch.randn(80, 10 + 3 * i), str(tmp_path / "mels" / ("a%d.pt" % i)))
        lines_wav.append("wavs/a%d.wav|Sentence number {N AH1 M B ER0}, take %s." % (i, "abc"[i]))
        lines_mel.append("mels/a%d.pt|Sentence number {N AH1 M B ER0}, take %s." % (i, "abc"[i]))
    (tmp_path / "wav.txt").write_text("\n".join(lines_wav) + "\n")
    (tmp_path / "mel.txt").write_text("\n".join(lines_mel) + "\n")
    args = types.SimpleNamespace(text_cleaners=["english_cleaners"], max_wav_value=32768.0, sampling_rate=22050, load_mel_from_disk=False,
                                 filter_length=1024, hop_length=256, win_length=1024, n_mel_channels=80, mel_fmin=0.0, mel_fmax=8000.0,
                                 segment_length=4000)
    ds = TextMelLoader(str(tmp_path), str(tmp_path / "wav.txt"), args)
    ids, mel, n_chars = ds[0]
    assert len(ds) == 3 and mel.shape == (80, 6000 // 256 + 1) and ids.dtype == torch.int32 and n_chars == len(lines_wav[0].split("|")[1])
    args.load_mel_from_disk = True
    dm = TextMelLoader(str(tmp_path), str(tmp_path / "mel.txt"), args)
    assert dm[2][1].shape == (80, 16)
    batch = TextMelCollate(1)([dm[i] for i in range(3)])
    (text, tl, melp, max_len, ol), (mel_t, gate), n_frames = batch_to_gpu(batch, device="cpu")
    assert text.dtype == torch.int64 and max_len == int(tl[0]) and int(n_frames) == 10 + 13 + 16 and gate[0, int(ol[0]) - 1] == 1
    wg = MelAudioLoader(str(tmp_path), str(tmp_path / "wav.txt"), args)
    m, a, n = wg[1]                                                # 3000 samples: zero padded to the segment
    assert a.shape == (4000,) and n == 4000 and m.shape == (80, 4000 // 256 + 1) and float(a[3000:].abs().max()) == 0
    m, a, n = wg[2]
    assert a.shape == (4000,) and float(a.abs().max()) <= 1
    (tmp_path / "bad.txt").write_text("a|b|c\n")
    with pytest.raises(Exception, match="incorrect line format"):
        TextMelLoader(str(tmp_path), str(tmp_path / "bad.txt"), args)
