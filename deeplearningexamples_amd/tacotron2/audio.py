"""Audio front end of the Tacotron2 / WaveGlow input pipelines (host side): wav -> log-mel spectrogram.

Restates TacotronSTFT.mel_spectrogram (tacotron2_common/layers.py:70-110) = STFT.transform (tacotron2_common/stft.py:83-108:
reflect padding by n_fft / 2, Hann-windowed DFT every hop samples, magnitude) -> mel filter bank -> log(clamp(x, 1e-5))
(audio_processing.py:79-85).  The reference builds the DFT as a conv1d with an explicit Fourier basis; here it is torch.stft
(same arithmetic, pinned against the reference's own STFT class by tests/golden/tacotron2_frontend.npz).  The filter bank is
librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) (Slaney scale: linear below 1 kHz, logarithmic above; area-normalised
triangles).  librosa is absent from this image, so the bank is restated from its published definition and is NOT pinned against
librosa's output ("parity unpinned" for the bank alone; training runs that load mels from disk, --load-mel-from-disk, bypass it).
"""
import numpy as np
import torch


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f / (200.0 / 3)
    log = 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) / (np.log(6.4) / 27.0)
    return np.where(f >= 1000.0, log, lin)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), m * (200.0 / 3))


def mel_filter_bank(sr, n_fft, n_mels, fmin, fmax):
    """[n_mels, n_fft // 2 + 1] float32: triangles between n_mels + 2 points equally spaced on the Slaney mel scale, each scaled
    by 2 / (its band width in Hz)."""
    fft_f = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    pts = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax if fmax is not None else sr / 2.0), n_mels + 2))
    diff = np.diff(pts)
    ramps = pts[:, None] - fft_f[None, :]
    lower = -ramps[:-2] / diff[:-1, None]
    upper = ramps[2:] / diff[1:, None]
    w = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (pts[2:] - pts[:-2]))[:, None]
    return w.astype(np.float32)


class TacotronSTFT:
    def __init__(self, filter_length=1024, hop_length=256, win_length=1024, n_mel_channels=80, sampling_rate=22050, mel_fmin=0.0,
                 mel_fmax=8000.0):
        if win_length > filter_length:
            raise ValueError("win_length must not exceed filter_length")
        self.n_fft, self.hop, self.win = filter_length, hop_length, win_length
        self.n_mel_channels, self.sampling_rate = n_mel_channels, sampling_rate
        self.window = torch.hann_window(win_length, periodic=True, dtype=torch.float32)
        self.mel_basis = torch.from_numpy(mel_filter_bank(sampling_rate, filter_length, n_mel_channels, mel_fmin, mel_fmax))

    def magnitudes(self, y):
        """y fp32 [B, T] -> |STFT| [B, n_fft / 2 + 1, 1 + T // hop]."""
        spec = torch.stft(y, self.n_fft, hop_length=self.hop, win_length=self.win, window=self.window, center=True,
                          pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
        return spec.abs()

    def mel_spectrogram(self, y):
        """y fp32 [B, T] in [-1, 1] -> log-mel [B, n_mel, frames]."""
        if float(y.min()) < -1 or float(y.max()) > 1:
            raise ValueError("audio outside [-1, 1]")
        return torch.log(torch.clamp(torch.matmul(self.mel_basis, self.magnitudes(y)), min=1e-5))


def load_wav_to_torch(path):
    """scipy.io.wavfile.read -> (fp32 samples, sampling rate) (tacotron2_common/utils.py:59-61)."""
    from scipy.io.wavfile import read
    sr, data = read(path)
    return torch.from_numpy(data.astype(np.float32)), sr
