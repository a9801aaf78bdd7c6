"""Dataset + collate of the Tacotron2 trainer (host side).

Mirrors tacotron2/data_function.py of the reference:
    :28-98    TextMelLoader    filelist lines `path|text` -> (symbol ids, mel [n_mel, T], len(text)); mels computed from the wav or,
                               with --load-mel-from-disk, torch.load'ed
    :100-138  TextMelCollate   sort by text length (descending), zero-pad ids and mels, gate target 1 from the last frame on,
                               frame count padded to a multiple of n_frames_per_step
    :141-152  batch_to_gpu     -> ((text, text_lengths, mel, max_len, output_lengths), (mel, gate), sum(output_lengths))
"""
import os

import torch
import torch.utils.data

from .audio import TacotronSTFT, load_wav_to_torch
from .text import text_to_sequence


def load_filepaths_and_text(dataset_path, filename, split="|"):
    """tacotron2_common/utils.py:64-76."""
    out = []
    with open(filename, encoding="utf-8") as f:
        for line in f:
            parts = line.strip().split(split)
            if len(parts) > 2:
                raise Exception("incorrect line format for file: {}".format(filename))
            out.append((os.path.join(dataset_path, parts[0]), parts[1]))
    return out


class TextMelLoader(torch.utils.data.Dataset):
    def __init__(self, dataset_path, audiopaths_and_text, args):
        self.audiopaths_and_text = load_filepaths_and_text(dataset_path, audiopaths_and_text)
        self.text_cleaners = args.text_cleaners
        self.max_wav_value = args.max_wav_value
        self.sampling_rate = args.sampling_rate
        self.load_mel_from_disk = args.load_mel_from_disk
        self.stft = TacotronSTFT(args.filter_length, args.hop_length, args.win_length, args.n_mel_channels, args.sampling_rate,
                                 args.mel_fmin, args.mel_fmax)

    def get_mel(self, filename):
        if self.load_mel_from_disk:
            mel = torch.load(filename)
            if mel.size(0) != self.stft.n_mel_channels:
                raise AssertionError("Mel dimension mismatch: given {}, expected {}".format(mel.size(0), self.stft.n_mel_channels))
            return mel
        audio, sr = load_wav_to_torch(filename)
        if sr != self.stft.sampling_rate:
            raise ValueError("{} {} SR doesn't match target {} SR".format(filename, sr, self.stft.sampling_rate))
        return self.stft.mel_spectrogram((audio / self.max_wav_value).unsqueeze(0)).squeeze(0)

    def get_text(self, text):
        return torch.IntTensor(text_to_sequence(text, self.text_cleaners))

    def __getitem__(self, index):
        path, text = self.audiopaths_and_text[index]
        return self.get_text(text), self.get_mel(path), len(text)

    def __len__(self):
        return len(self.audiopaths_and_text)


class TextMelCollate:
    def __init__(self, n_frames_per_step):
        self.n_frames_per_step = n_frames_per_step

    def __call__(self, batch):
        """batch: [(ids, mel [n_mel, T], len_text)] -> (text_padded, input_lengths, mel_padded, gate_padded, output_lengths, len_x)."""
        input_lengths, order = torch.sort(torch.LongTensor([len(x[0]) for x in batch]), dim=0, descending=True)
        n = len(batch)
        text_padded = torch.zeros(n, int(input_lengths[0]), dtype=torch.long)
        for i, j in enumerate(order.tolist()):
            text_padded[i, :batch[j][0].size(0)] = batch[j][0]
        num_mels = batch[0][1].size(0)
        t_max = max(x[1].size(1) for x in batch)
        if t_max % self.n_frames_per_step:
            t_max += self.n_frames_per_step - t_max % self.n_frames_per_step
        mel_padded = torch.zeros(n, num_mels, t_max)
        gate_padded = torch.zeros(n, t_max)
        output_lengths = torch.zeros(n, dtype=torch.long)
        for i, j in enumerate(order.tolist()):
            mel = batch[j][1]
            mel_padded[i, :, :mel.size(1)] = mel
            gate_padded[i, mel.size(1) - 1:] = 1
            output_lengths[i] = mel.size(1)
        len_x = torch.Tensor([x[2] for x in batch])
        return text_padded, input_lengths, mel_padded, gate_padded, output_lengths, len_x


def batch_to_gpu(batch, device="cuda"):
    text_padded, input_lengths, mel_padded, gate_padded, output_lengths, len_x = batch
    to = lambda t: t.contiguous().to(device, non_blocking=True)
    text_padded, input_lengths, output_lengths = to(text_padded).long(), to(input_lengths).long(), to(output_lengths).long()
    mel_padded, gate_padded = to(mel_padded).float(), to(gate_padded).float()
    max_len = int(torch.max(input_lengths).item())
    return ((text_padded, input_lengths, mel_padded, max_len, output_lengths), (mel_padded, gate_padded), torch.sum(output_lengths))
