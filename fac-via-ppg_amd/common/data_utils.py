"""``get_ppg`` hook of the reference's src/common/data_utils.py:55-59.

The reference computes the PPG of a wav with PyKaldi: features (ppg.compute_feat_for_nnet, built here on HIP kernels) ->
nnet3 acoustic model (data/am/final.raw, a blob the reference does not ship) -> posteriors.  With a model file present
(``deps.nnet``) the same chain runs here on the HIP kernels (ppg.compute_full_ppg_wrapper); without it this build reads a
precomputed PPG ([Tin, n_symbols] float array, 10 ms frame shift, rows = posteriors): either the
given path itself is a ``.npy`` file or a sibling ``<wav>.ppg.npy`` exists next to the wav.
"""
import os

import numpy as np


def ppg_candidates(wav_path):
    """Where the precomputed PPG of ``wav_path`` may live."""
    if wav_path.endswith(".npy"):
        return [wav_path]
    return [wav_path + ".ppg.npy", os.path.splitext(wav_path)[0] + ".ppg.npy"]


def get_ppg(wav_path, deps=None, is_fmllr=False):
    candidates = ppg_candidates(wav_path)
    for c in candidates:
        if os.path.isfile(c):
            ppg = np.load(c)
            if ppg.ndim != 2:
                raise ValueError("PPG file %s must hold a [Tin, n_symbols] array, got shape %s" % (c, ppg.shape))
            return ppg.astype(np.float32)
    if deps is not None and getattr(deps, "nnet", None) is not None and os.path.isfile(wav_path):
        # data_utils.py:55-59: wav -> features -> acoustic model -> full PPG, all on the HIP kernels
        from common import feat
        from ppg import compute_full_ppg_wrapper
        return compute_full_ppg_wrapper(feat.read_wav_kaldi(wav_path), deps.nnet, deps.lda, 10)
    raise NotImplementedError(
        "PPG extraction from audio needs the Kaldi nnet3 acoustic model (data/am/final.raw), which the reference does not ship; "
        "provide a precomputed PPG as %s (the model's input features are available: ppg.compute_feat_for_nnet)" % " or ".join(candidates))
