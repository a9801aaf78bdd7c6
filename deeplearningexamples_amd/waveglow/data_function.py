"""Dataset of the WaveGlow trainer (host side): waveglow/data_function.py:33-86 of the reference.

MelAudioLoader: filelist lines `path|text` -> a random segment of `segment_length` samples (zero padded when the file is shorter),
scaled by 1 / max_wav_value, with its log-mel spectrogram (tacotron2/audio.py); batches through torch's default collate;
batch_to_gpu -> ((mel, audio), audio, total samples).
"""
import torch
import torch.utils.data

from ..tacotron2.audio import TacotronSTFT, load_wav_to_torch
from ..tacotron2.data_function import load_filepaths_and_text


class MelAudioLoader(torch.utils.data.Dataset):
    def __init__(self, dataset_path, audiopaths_and_text, args):
        self.audiopaths_and_text = load_filepaths_and_text(dataset_path, audiopaths_and_text)
        self.max_wav_value, self.sampling_rate, self.segment_length = args.max_wav_value, args.sampling_rate, args.segment_length
        self.stft = TacotronSTFT(args.filter_length, args.hop_length, args.win_length, args.n_mel_channels, args.sampling_rate,
                                 args.mel_fmin, args.mel_fmax)

    def get_mel_audio_pair(self, filename):
        audio, sr = load_wav_to_torch(filename)
        if sr != self.stft.sampling_rate:
            raise ValueError("{} {} SR doesn't match target {} SR".format(filename, sr, self.stft.sampling_rate))
        if audio.size(0) >= self.segment_length:
            start = int(torch.randint(0, audio.size(0) - self.segment_length + 1, size=(1,)).item())
            audio = audio[start:start + self.segment_length]
        else:
            audio = torch.nn.functional.pad(audio, (0, self.segment_length - audio.size(0)), "constant")
        audio = audio / self.max_wav_value
        return self.stft.mel_spectrogram(audio.unsqueeze(0)).squeeze(0), audio, len(audio)

    def __getitem__(self, index):
        return self.get_mel_audio_pair(self.audiopaths_and_text[index][0])

    def __len__(self):
        return len(self.audiopaths_and_text)


def batch_to_gpu(batch, device="cuda"):
    x, y, len_y = batch
    x, y = x.contiguous().to(device).float(), y.contiguous().to(device).float()
    return (x, y), y, torch.sum(len_y).to(device)
