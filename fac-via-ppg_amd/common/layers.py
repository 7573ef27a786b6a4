"""LinearNorm / ConvNorm / TacotronSTFT -- drop-in for src/common/layers.py.

LinearNorm and ConvNorm are parameter containers with the reference's xavier initialisation
(layers.py:40-71); their arithmetic runs inside libfacppg_hip's GEMM kernels when the owning
model's ``inference`` is called.  TacotronSTFT.mel_spectrogram runs on the HIP STFT.
"""
import numpy as np
import torch

from common.audio_processing import dynamic_range_compression, dynamic_range_decompression
from common.stft import STFT


class LinearNorm(torch.nn.Module):
    def __init__(self, in_dim, out_dim, bias=True, w_init_gain='linear'):
        super(LinearNorm, self).__init__()
        self.linear_layer = torch.nn.Linear(in_dim, out_dim, bias=bias)
        torch.nn.init.xavier_uniform_(self.linear_layer.weight, gain=torch.nn.init.calculate_gain(w_init_gain))

    def forward(self, x):
        raise NotImplementedError("LinearNorm is a parameter container; the math runs in libfacppg_hip (Tacotron2.inference)")


class ConvNorm(torch.nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=None, dilation=1, bias=True,
                 w_init_gain='linear'):
        super(ConvNorm, self).__init__()
        if padding is None:
            assert kernel_size % 2 == 1
            padding = int(dilation * (kernel_size - 1) / 2)
        self.conv = torch.nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                                    dilation=dilation, bias=bias)
        torch.nn.init.xavier_uniform_(self.conv.weight, gain=torch.nn.init.calculate_gain(w_init_gain))

    def forward(self, signal):
        raise NotImplementedError("ConvNorm is a parameter container; the math runs in libfacppg_hip (Tacotron2.inference)")


def slaney_mel_basis(sampling_rate, n_fft, n_mels, fmin, fmax):
    """Mel filterbank with librosa 0.6.2's defaults (Slaney scale, htk=False, norm=1), the
    function the reference calls at layers.py:82-83.  librosa is not a dependency of this build;
    the published algorithm is implemented here: the scale is linear (200/3 Hz per mel) below
    1 kHz and logarithmic (27 mels per factor 6.4) above, filters are triangles between
    consecutive band edges, each scaled to unit area (2 / bandwidth)."""
    fmax = sampling_rate / 2.0 if fmax is None else fmax
    lin_step, knee_hz, log_step = 200.0 / 3.0, 1000.0, np.log(6.4) / 27.0
    knee_mel = knee_hz / lin_step

    def to_mel(hz):
        hz = np.atleast_1d(np.asarray(hz, dtype=np.float64))
        out = hz / lin_step
        hi = hz >= knee_hz
        out[hi] = knee_mel + np.log(hz[hi] / knee_hz) / log_step
        return out

    def to_hz(mel):
        mel = np.asarray(mel, dtype=np.float64)
        out = mel * lin_step
        hi = mel >= knee_mel
        out[hi] = knee_hz * np.exp(log_step * (mel[hi] - knee_mel))
        return out

    edges = to_hz(np.linspace(to_mel(fmin)[0], to_mel(fmax)[0], n_mels + 2))
    bins = np.linspace(0.0, sampling_rate / 2.0, n_fft // 2 + 1)
    rising = (bins[None, :] - edges[:-2, None]) / (edges[1:-1] - edges[:-2])[:, None]
    falling = (edges[2:, None] - bins[None, :]) / (edges[2:] - edges[1:-1])[:, None]
    tri = np.clip(np.minimum(rising, falling), 0.0, None)
    return tri * (2.0 / (edges[2:] - edges[:-2]))[:, None]


class TacotronSTFT(torch.nn.Module):
    """layers.py:74-112"""

    def __init__(self, filter_length=1024, hop_length=256, win_length=1024, n_mel_channels=80, sampling_rate=22050,
                 mel_fmin=0.0, mel_fmax=8000.0):
        super(TacotronSTFT, self).__init__()
        self.n_mel_channels = n_mel_channels
        self.sampling_rate = sampling_rate
        mel_basis = slaney_mel_basis(sampling_rate, filter_length, n_mel_channels, mel_fmin, mel_fmax).astype(np.float32)
        self.stft_fn = STFT(filter_length, hop_length, win_length, mel_basis=mel_basis)
        self.register_buffer('mel_basis', torch.from_numpy(mel_basis).float())

    def spectral_normalize(self, magnitudes):
        return dynamic_range_compression(magnitudes)

    def spectral_de_normalize(self, magnitudes):
        return dynamic_range_decompression(magnitudes)

    def mel_spectrogram(self, y, lengths=None):
        """y [B, T] in [-1, 1] (GPU) -> log-mel [B, n_mel_channels, T//hop + 1]"""
        assert torch.min(y.data) >= -1
        assert torch.max(y.data) <= 1
        return self.stft_fn.mel(y, lengths)
