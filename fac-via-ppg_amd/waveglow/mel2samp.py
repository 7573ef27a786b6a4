"""(mel, audio) training pairs for WaveGlow -- drop-in for src/waveglow/mel2samp.py.

Same constructor and item semantics as the reference's ``Mel2Samp`` (random ``segment_length``
crop, zero padding of short files, audio scaled by 1/32768, mel from TacotronSTFT with the data
config's parameters), but the mel analysis runs on the GPU STFT kernels (facppg_stft_mel):
``get_mel`` for one utterance as in the reference, and ``mel_batch`` for a whole [B, N] batch
already on the device, which is what script.train_waveglow uses so features never touch the host.
"""
import argparse
import json
import os
import random

import torch
import torch.utils.data
from scipy.io import wavfile

from common.layers import TacotronSTFT

MAX_WAV_VALUE = 32768.0


def files_to_list(filename):
    """One path per line (mel2samp.py:42-50)."""
    with open(filename, encoding='utf-8') as f:
        return [line.rstrip() for line in f.readlines()]


def load_wav_to_torch(full_path):
    """16-bit PCM wav -> float tensor in the int16 range (mel2samp.py:52-57)."""
    sampling_rate, data = wavfile.read(full_path)
    return torch.from_numpy(data).float(), sampling_rate


class Mel2Samp(torch.utils.data.Dataset):
    """mel2samp.py:60-113"""

    def __init__(self, training_files, segment_length, filter_length, hop_length, win_length, sampling_rate, mel_fmin,
                 mel_fmax, audio_only=False):
        self.audio_files = files_to_list(training_files)
        random.seed(1234)
        random.shuffle(self.audio_files)
        self.stft = TacotronSTFT(filter_length=filter_length, hop_length=hop_length, win_length=win_length,
                                 sampling_rate=sampling_rate, mel_fmin=mel_fmin, mel_fmax=mel_fmax)
        self.segment_length = segment_length
        self.sampling_rate = sampling_rate
        self.audio_only = audio_only      # items are audio segments; the trainer calls mel_batch on the GPU
        self.wav_cache = {}

    def mel_batch(self, audio_norm):
        """audio_norm [B, N] in [-1, 1] on the GPU -> mel [B, n_mel, N//hop + 1] (one fused pass)."""
        return self.stft.mel_spectrogram(audio_norm)

    def get_mel(self, audio):
        """int16-range audio [N] -> mel [n_mel, N//hop + 1]  (mel2samp.py:79-85)."""
        audio_norm = (audio / MAX_WAV_VALUE).unsqueeze(0).cuda()
        return torch.squeeze(self.mel_batch(audio_norm), 0)

    def __getitem__(self, index):
        filename = self.audio_files[index]
        if filename not in self.wav_cache:
            self.wav_cache[filename] = load_wav_to_torch(filename)
        audio, sampling_rate = self.wav_cache[filename]
        if sampling_rate != self.sampling_rate:
            raise ValueError("{} SR doesn't match target {} SR".format(sampling_rate, self.sampling_rate))
        if audio.size(0) >= self.segment_length:
            start = random.randint(0, audio.size(0) - self.segment_length)
            audio = audio[start:start + self.segment_length]
        else:
            audio = torch.nn.functional.pad(audio, (0, self.segment_length - audio.size(0)), 'constant').data
        if self.audio_only:
            return audio / MAX_WAV_VALUE
        mel = self.get_mel(audio).cpu()
        return (mel, audio / MAX_WAV_VALUE)

    def __len__(self):
        return len(self.audio_files)


if __name__ == "__main__":     # directory of clean audio -> directory of mel .pt files (mel2samp.py:115-147)
    parser = argparse.ArgumentParser()
    parser.add_argument('-f', "--filelist_path", required=True)
    parser.add_argument('-c', '--config', type=str, help='JSON file for configuration')
    parser.add_argument('-o', '--output_dir', type=str, help='Output directory')
    args = parser.parse_args()
    with open(args.config) as f:
        data_config = json.load(f)["data_config"]
    mel2samp = Mel2Samp(**data_config)
    os.makedirs(args.output_dir, exist_ok=True)
    for filepath in files_to_list(args.filelist_path):
        audio, sr = load_wav_to_torch(filepath)
        new_filepath = args.output_dir + '/' + os.path.basename(filepath) + '.pt'
        print(new_filepath)
        torch.save(mel2samp.get_mel(audio).cpu(), new_filepath)
