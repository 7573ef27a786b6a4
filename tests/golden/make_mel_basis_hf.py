#!/usr/bin/env python3
"""Second, independent derivation of the Slaney mel filterbank the reference takes from librosa 0.6.2
(`librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)` at src/common/layers.py:82-83; librosa is absent from this image):
Hugging Face transformers' `audio_utils.mel_filter_bank(..., norm="slaney", mel_scale="slaney")`, an implementation that
shares no code with oracle/dsp.py and that its authors test against librosa.  Writes the non-zero entries of the bases
the hot path uses (reference config 16 kHz / 1024 / 80 / 0..8000 Hz; the 22.05 kHz metric config; librosa's documentation
example mel(22050, 2048)) to tests/golden/mel_basis_hf.npz.

  python tests/golden/make_mel_basis_hf.py        (needs `transformers`; version recorded in the file)
"""
import os

import numpy as np
import transformers
from transformers.audio_utils import mel_filter_bank

CASES = {"ref16k": (16000, 1024, 80, 0.0, 8000.0), "metric22k": (22050, 1024, 80, 0.0, 8000.0), "librosa_doc": (22050, 2048, 128, 0.0, 11025.0)}

out = {"transformers_version": np.frombuffer(transformers.__version__.encode(), dtype=np.uint8)}
for tag, (sr, n_fft, n_mels, fmin, fmax) in CASES.items():
    fb = mel_filter_bank(num_frequency_bins=n_fft // 2 + 1, num_mel_filters=n_mels, min_frequency=fmin, max_frequency=fmax,
                         sampling_rate=sr, norm="slaney", mel_scale="slaney").T          # -> [n_mels, 1 + n_fft/2] like librosa
    idx = np.flatnonzero(fb)
    out[tag + "_args"] = np.array([sr, n_fft, n_mels, fmin, fmax], dtype=np.float64)
    out[tag + "_idx"] = idx.astype(np.int32)
    out[tag + "_val"] = fb.reshape(-1)[idx].astype(np.float64)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mel_basis_hf.npz"), **out)
print({k: v.shape for k, v in out.items()})
