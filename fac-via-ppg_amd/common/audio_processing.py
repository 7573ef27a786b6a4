"""Host-side helpers of the signal path -- drop-in for src/common/audio_processing.py.

On the synthesis path the work these functions describe happens inside libfacppg_hip.so
(k_overlap_add recomputes the window sum-square envelope per call on the device; the mel GEMM's
epilogue applies the log compression).  The functions remain for API compatibility.
"""
import numpy as np
import torch
from scipy.signal import get_window


def _center_pad(x, size):
    left = (size - len(x)) // 2
    return np.concatenate([np.zeros(left, x.dtype), x, np.zeros(size - len(x) - left, x.dtype)])


def squared_window(window, win_length, n_fft):
    """window**2, zero padded symmetrically to n_fft (audio_processing.py:79-82)."""
    return _center_pad(get_window(window, win_length, fftbins=True) ** 2, n_fft)


def window_sumsquare(window, n_frames, hop_length=200, win_length=800, n_fft=800, dtype=np.float32, norm=None):
    """Sum-square envelope of the analysis window over n_frames hops (audio_processing.py:39-88)."""
    if norm is not None:
        raise NotImplementedError("only norm=None is used by the reference (stft.py:119-123)")
    win_length = n_fft if win_length is None else win_length
    total = n_fft + hop_length * (n_frames - 1)
    env = np.zeros(total, dtype=dtype)
    wsq = squared_window(window, win_length, n_fft)
    for f in range(n_frames):
        lo = f * hop_length
        seg = min(n_fft, total - lo)
        if seg > 0:
            env[lo:lo + seg] += wsq[:seg]
    return env


def griffin_lim(magnitudes, stft_fn, n_iters=30):
    """Phase retrieval by alternating projections (audio_processing.py:91-107); unused on the
    synthesis path, runs on stft_fn's (HIP) transform/inverse."""
    angles = np.angle(np.exp(2j * np.pi * np.random.rand(*magnitudes.size()))).astype(np.float32)
    angles = torch.from_numpy(angles).to(magnitudes.device)
    signal = stft_fn.inverse(magnitudes, angles).squeeze(1)
    for _ in range(n_iters):
        _, angles = stft_fn.transform(signal)
        signal = stft_fn.inverse(magnitudes, angles).squeeze(1)
    return signal


def dynamic_range_compression(x, C=1, clip_val=1e-5):
    """log(clamp(x, clip_val) * C)  (audio_processing.py:110-116)"""
    return torch.log(torch.clamp(x, min=clip_val) * C)


def dynamic_range_decompression(x, C=1):
    """exp(x) / C  (audio_processing.py:119-125)"""
    return torch.exp(x) / C
