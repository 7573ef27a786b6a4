"""Oracle (TEST INFRASTRUCTURE ONLY): STFT / mel / denoiser signal ops on the CPU.

Follows, with file:line citations into /root/reference:
  * src/common/stft.py:46-138          STFT as conv1d with a windowed DFT basis
  * src/common/audio_processing.py:39-88, 110-125   window_sumsquare, log compression
  * src/common/layers.py:74-112        TacotronSTFT.mel_spectrogram
  * src/waveglow/denoiser.py:38-68     Denoiser
  * librosa==0.6.2 (environment.yml:46; third-party, absent): filters.mel,
    util.pad_center, util.tiny -- restated from the published algorithm, PARITY UNPINNED.
"""
import numpy as np
import torch
import torch.nn.functional as F
from scipy.signal import get_window


# ----------------------------------------------------------------------------- librosa 0.6.2 restatement
def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, mels)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm=1) as published for
    0.6.2: Slaney mel scale, triangular filters, area ("slaney") normalisation.
    Call site: src/common/layers.py:82-83."""
    if fmax is None:
        fmax = sr / 2.0
    n_bins = 1 + n_fft // 2
    fft_f = np.linspace(0.0, sr / 2.0, n_bins)
    mel_pts = _mel_to_hz_slaney(np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2))
    fdiff = np.diff(mel_pts)
    ramps = mel_pts[:, None] - fft_f[None, :]
    w = np.zeros((n_mels, n_bins))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_pts[2:n_mels + 2] - mel_pts[:n_mels])
    return w * enorm[:, None]


def pad_center(data, size):
    """librosa.util.pad_center for 1-D data (call site stft.py:69)."""
    n = data.shape[-1]
    lpad = (size - n) // 2
    return np.pad(data, (lpad, size - n - lpad), mode="constant")


def tiny(x):
    """librosa.util.tiny (call site stft.py:126)."""
    return np.finfo(np.asarray(x).dtype).tiny


# ----------------------------------------------------------------------------- audio_processing.py
def window_sumsquare(window, n_frames, hop_length, win_length, n_fft, dtype=np.float32):
    """src/common/audio_processing.py:39-88 (norm=None => librosa normalize is identity)."""
    n = n_fft + hop_length * (n_frames - 1)
    x = np.zeros(n, dtype=dtype)
    win_sq = get_window(window, win_length, fftbins=True) ** 2
    win_sq = pad_center(win_sq, n_fft)
    for i in range(n_frames):
        s = i * hop_length
        x[s:min(n, s + n_fft)] += win_sq[:max(0, min(n_fft, n - s))]
    return x


def dynamic_range_compression(x, C=1, clip_val=1e-5):
    """audio_processing.py:110-116"""
    return torch.log(torch.clamp(x, min=clip_val) * C)


# ----------------------------------------------------------------------------- stft.py
class StftOracle:
    """src/common/stft.py:44-138."""

    def __init__(self, filter_length=800, hop_length=200, win_length=800, window="hann"):
        self.filter_length, self.hop_length, self.win_length, self.window = \
            filter_length, hop_length, win_length, window
        scale = filter_length / hop_length
        fb = np.fft.fft(np.eye(filter_length))                       # stft.py:55
        cutoff = filter_length // 2 + 1
        fb = np.vstack([np.real(fb[:cutoff]), np.imag(fb[:cutoff])])  # stft.py:57-59
        fwd = torch.FloatTensor(fb[:, None, :])
        inv = torch.FloatTensor(np.linalg.pinv(scale * fb).T[:, None, :])  # stft.py:61-63
        win = torch.from_numpy(pad_center(get_window(window, win_length, fftbins=True), filter_length)).float()
        self.forward_basis = (fwd * win).float()                      # stft.py:72-74
        self.inverse_basis = (inv * win).float()

    def transform(self, x):
        """x [B, N] -> magnitude, phase [B, n_fft/2+1, N//hop+1]  (stft.py:79-107)."""
        B, N = x.shape
        half = self.filter_length // 2
        xp = F.pad(x.view(B, 1, 1, N), (half, half, 0, 0), mode="reflect").view(B, 1, -1)
        ft = F.conv1d(xp, self.forward_basis, stride=self.hop_length)
        cutoff = half + 1
        re, im = ft[:, :cutoff], ft[:, cutoff:]
        return torch.sqrt(re ** 2 + im ** 2), torch.atan2(im, re)

    def inverse(self, magnitude, phase):
        """stft.py:109-138 -> [B, 1, hop*(frames-1)]."""
        rec = torch.cat([magnitude * torch.cos(phase), magnitude * torch.sin(phase)], dim=1)
        y = F.conv_transpose1d(rec, self.inverse_basis, stride=self.hop_length)
        wsum = window_sumsquare(self.window, magnitude.size(-1), self.hop_length, self.win_length,
                                self.filter_length, np.float32)
        idx = torch.from_numpy(np.where(wsum > tiny(wsum))[0])
        wsum_t = torch.from_numpy(wsum)
        y[:, :, idx] = y[:, :, idx] / wsum_t[idx]
        y = y * (float(self.filter_length) / self.hop_length)
        half = self.filter_length // 2
        return y[:, :, half:-half]


class TacotronStftOracle:
    """src/common/layers.py:74-112."""

    def __init__(self, filter_length=1024, hop_length=256, win_length=1024, n_mel_channels=80,
                 sampling_rate=22050, mel_fmin=0.0, mel_fmax=8000.0):
        self.stft = StftOracle(filter_length, hop_length, win_length)
        self.mel_basis = torch.from_numpy(
            mel_filterbank(sampling_rate, filter_length, n_mel_channels, mel_fmin, mel_fmax)).float()

    def mel_spectrogram(self, y):
        assert torch.min(y) >= -1 and torch.max(y) <= 1       # layers.py:105-106
        mag, _ = self.stft.transform(y)
        return dynamic_range_compression(torch.matmul(self.mel_basis, mag))


class DenoiserOracle:
    """src/waveglow/denoiser.py:38-68; ``bias_audio`` = WaveGlow.infer(zeros(1,80,88), sigma=0)."""

    def __init__(self, bias_audio, filter_length=1024, hop_length=160, win_length=1024):
        self.stft = StftOracle(filter_length, hop_length, win_length)
        bias_spec, _ = self.stft.transform(bias_audio.float())
        self.bias_spec = bias_spec[:, :, 0][:, :, None]       # denoiser.py:61

    def __call__(self, audio, strength=0.1):
        spec, ang = self.stft.transform(audio.float())
        spec = torch.clamp(spec - self.bias_spec * strength, 0.0)
        return self.stft.inverse(spec, ang)
