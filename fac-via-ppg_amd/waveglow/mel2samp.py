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
    """Dataset of fixed-length training segments (mel2samp.py:60-113): item = (mel [n_mel, seg//hop + 1], audio [seg] in
    [-1, 1]) or, with ``audio_only``, just the audio (script.train_waveglow then computes the mels of a whole batch on the
    GPU).  Contract kept from the reference: file order = ``random.seed(1234)`` shuffle of the list, one
    ``random.randint`` draw per long file for the crop start, zero padding of short files, int16 wavs scaled by 1/32768."""

    def __init__(self, training_files, segment_length, filter_length, hop_length, win_length, sampling_rate, mel_fmin,
                 mel_fmax, audio_only=False):
        self.audio_files = files_to_list(training_files)
        random.seed(1234)
        random.shuffle(self.audio_files)
        self.segment_length, self.sampling_rate, self.audio_only = segment_length, sampling_rate, audio_only
        self.stft = TacotronSTFT(filter_length=filter_length, hop_length=hop_length, win_length=win_length,
                                 sampling_rate=sampling_rate, mel_fmin=mel_fmin, mel_fmax=mel_fmax)
        self.wav_cache = {}

    def __len__(self):
        return len(self.audio_files)

    # ---- mel analysis on the GPU STFT kernels
    def mel_batch(self, audio_norm):
        """audio_norm [B, N] in [-1, 1] on the GPU -> mel [B, n_mel, N//hop + 1] (one fused pass)."""
        return self.stft.mel_spectrogram(audio_norm)

    def get_mel(self, audio):
        """int16-range audio [N] -> mel [n_mel, N//hop + 1]  (mel2samp.py:79-85)."""
        return self.mel_batch((audio / MAX_WAV_VALUE)[None].cuda())[0]

    # ---- segments
    def _samples(self, index):
        """The whole file as int16-range floats (cached), checked against the configured rate."""
        path = self.audio_files[index]
        hit = self.wav_cache.get(path)
        if hit is None:
            hit = self.wav_cache[path] = load_wav_to_torch(path)
        samples, rate = hit
        if rate != self.sampling_rate:
            raise ValueError("{} SR doesn't match target {} SR".format(rate, self.sampling_rate))
        return samples

    def _segment(self, samples):
        """Random crop of long files, right zero padding of short ones."""
        spare = samples.size(0) - self.segment_length
        if spare < 0:
            return torch.nn.functional.pad(samples, (0, -spare))
        first = random.randint(0, spare)
        return samples[first:first + self.segment_length]

    def __getitem__(self, index):
        segment = self._segment(self._samples(index))
        audio = segment / MAX_WAV_VALUE
        return audio if self.audio_only else (self.get_mel(segment).cpu(), audio)


def export_mels(filelist_path, config_path, output_dir):
    """Clean audio files -> ``<output_dir>/<name>.pt`` mel tensors (the reference's ``__main__``, mel2samp.py:115-147)."""
    with open(config_path) as f:
        dataset = Mel2Samp(**json.load(f)["data_config"])
    os.makedirs(output_dir, exist_ok=True)
    for wav_path in files_to_list(filelist_path):
        target = os.path.join(output_dir, os.path.basename(wav_path) + '.pt')
        print(target)
        torch.save(dataset.get_mel(load_wav_to_torch(wav_path)[0]).cpu(), target)


if __name__ == "__main__":
    cli = argparse.ArgumentParser()
    cli.add_argument('-f', "--filelist_path", required=True)
    cli.add_argument('-c', '--config', type=str, help='JSON file for configuration')
    cli.add_argument('-o', '--output_dir', type=str, help='Output directory')
    opts = cli.parse_args()
    export_mels(opts.filelist_path, opts.config, opts.output_dir)
