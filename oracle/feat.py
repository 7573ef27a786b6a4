"""TEST INFRASTRUCTURE (CPU oracle) -- a step-by-step NumPy restatement of the acoustic model's input features:
Kaldi MFCC -> CMN -> splice -> LDA, and the senone -> monophone reduction of a PPG.

Follows the reference's call sequence (src/ppg/compute_ppg.py:97-134, src/common/feat.py:74-156) and, for what pykaldi
hides, Kaldi 5.3's published algorithm (feat/feature-window.cc: NumFrames / FirstSampleOfFrame / ExtractWindow /
ProcessWindow; feat/mel-computations.cc: MelBanks; feat/feature-mfcc.cc; matrix/matrix-functions.cc: ComputeDctMatrix;
feat/feature-functions.cc: SpliceFrames).

PARITY UNPINNED at the Kaldi boundary: pykaldi==0.1.0 (environment.yml:97) is absent, so no Kaldi output can be captured
here.  What pins this oracle are the reference's own known answers for this part (test/test_feat.py, test/test_ppg.py):
frame count = round(samples / (fs * shift)), 13 MFCC dims, CMN sums to ~0, LDA output dim 40, reduce_dim.mat sums to 5816
and keeps the posterior mass.  Dither (Kaldi default 1.0: random +-1 LSB noise) is not applied."""
import numpy as np

FLT_EPS = np.float32(1.1920929e-7)


def num_frames(n_samples, shift):
    return (n_samples + shift // 2) // shift                       # snip_edges = false


def extract_frames(wav, length, shift):
    """[T, length] raw frames with the reflection of feature-window.cc ExtractWindow (snip_edges = false)."""
    wav = np.asarray(wav, dtype=np.float64)
    N, T = len(wav), num_frames(len(wav), shift)
    out = np.empty((T, length))
    for t in range(T):
        idx = t * shift + shift // 2 - length // 2 + np.arange(length)
        for _ in range(8):
            idx = np.where(idx < 0, -idx - 1, idx)
            idx = np.where(idx >= N, 2 * N - 1 - idx, idx)
        out[t] = wav[idx]
    return out


def resample(wav, fs_in, fs_out, num_zeros=6):
    """Kaldi DownsampleWaveForm -> LinearResample (feat/resample.cc), flush = true: windowed-sinc low-pass at
    0.99 * min(fs) / 2 with `num_zeros` zero crossings, Hanning window; output samples strictly inside the input's span."""
    wav = np.asarray(wav, dtype=np.float64)
    n_in = len(wav)
    num = n_in * int(fs_out)
    last = num // int(fs_in)
    if last * int(fs_in) == num:
        last -= 1
    n_out = last + 1
    cutoff = 0.99 * 0.5 * min(fs_in, fs_out)
    width = num_zeros / (2.0 * cutoff)
    out = np.zeros(n_out)
    ti = np.arange(n_in) / fs_in
    for n in range(n_out):
        t_out = n / fs_out
        i0, i1 = max(0, int(np.ceil((t_out - width) * fs_in))), min(n_in - 1, int(np.floor((t_out + width) * fs_in)))
        t = t_out - ti[i0:i1 + 1]
        keep = np.abs(t) < width
        win = 0.5 * (1.0 + np.cos(2.0 * np.pi * cutoff / num_zeros * t))
        with np.errstate(divide="ignore", invalid="ignore"):
            filt = np.where(t != 0.0, np.sin(2.0 * np.pi * cutoff * t) / (np.pi * t), 2.0 * cutoff)
        out[n] = np.sum(wav[i0:i1 + 1] * filt * win * keep) / fs_in
    return out.astype(np.float32)


def povey_window(n):
    i = np.arange(n)
    return (0.5 - 0.5 * np.cos(2 * np.pi * i / (n - 1))) ** 0.85


def mel_scale(f):
    return 1127.0 * np.log(1.0 + f / 700.0)


def mel_banks(num_bins, n_fft, samp_freq, low_freq=20.0, high_freq=0.0):
    """[num_bins, n_fft/2] triangular weights in the mel domain (MelBanks::MelBanks, no VTLN); the Nyquist bin is not used."""
    nyquist = 0.5 * samp_freq
    high = high_freq if high_freq > 0 else nyquist + high_freq
    nb = n_fft // 2
    lo, hi = mel_scale(low_freq), mel_scale(high)
    delta = (hi - lo) / (num_bins + 1)
    mel = mel_scale(samp_freq / n_fft * np.arange(nb))
    w = np.zeros((num_bins, nb))
    for b in range(num_bins):
        left, center, right = lo + b * delta, lo + (b + 1) * delta, lo + (b + 2) * delta
        up = (mel > left) & (mel <= center)
        dn = (mel > center) & (mel < right)
        w[b, up] = (mel[up] - left) / (center - left)
        w[b, dn] = (right - mel[dn]) / (right - center)
    return w


def dct_matrix(n_rows, n_cols):
    """First n_rows rows of ComputeDctMatrix (type II, orthonormal)."""
    m = np.zeros((n_rows, n_cols))
    m[0] = np.sqrt(1.0 / n_cols)
    n = np.arange(n_cols)
    for k in range(1, n_rows):
        m[k] = np.sqrt(2.0 / n_cols) * np.cos(np.pi / n_cols * (n + 0.5) * k)
    return m


def lifter(n_ceps, q=22.0):
    return 1.0 + 0.5 * q * np.sin(np.pi * np.arange(n_ceps) / q)


def mfcc(wav, samp_freq=16000.0, frame_shift_ms=10.0, frame_length_ms=25.0, num_ceps=13, num_mel_bins=23, preemph=0.97,
         cepstral_lifter=22.0, use_energy=False):
    """[T, num_ceps] float32 (Mfcc::Compute with snip_edges=false, remove_dc_offset, povey window, round_to_power_of_two)."""
    length, shift = int(samp_freq * 0.001 * frame_length_ms), int(samp_freq * 0.001 * frame_shift_ms)
    n_fft = 1 << (length - 1).bit_length()
    frames = extract_frames(wav, length, shift)
    frames = frames - frames.mean(axis=1, keepdims=True)                               # remove_dc_offset
    log_energy = np.log(np.maximum((frames * frames).sum(1), FLT_EPS))                 # raw_energy = true
    pre = frames.copy()
    pre[:, 1:] -= preemph * frames[:, :-1]
    pre[:, 0] -= preemph * frames[:, 0]
    spec = np.fft.rfft(pre * povey_window(length), n=n_fft, axis=1)
    power = (spec.real ** 2 + spec.imag ** 2)[:, :n_fft // 2]
    mel = np.log(np.maximum(power @ mel_banks(num_mel_bins, n_fft, samp_freq).T, FLT_EPS))
    out = (mel @ dct_matrix(num_ceps, num_mel_bins).T) * lifter(num_ceps, cepstral_lifter)
    if use_energy:
        out[:, 0] = log_energy
    return out.astype(np.float32)


def cmn(feats):
    """feat.py:103-118"""
    return feats - feats.mean(axis=0, keepdims=True)


def splice(feats, left, right):
    """SpliceFrames: frame t -> [t-left .. t+right], edge frames replicated."""
    T = feats.shape[0]
    idx = np.clip(np.arange(T)[:, None] + np.arange(-left, right + 1)[None, :], 0, T - 1)
    return feats[idx].reshape(T, -1)


def transform(feats, mat):
    """feat.py:121-156: F T' (linear) or with the implicit 1.0 appended (affine)."""
    D = feats.shape[1]
    if mat.shape[1] == D:
        return feats @ mat.T
    if mat.shape[1] == D + 1:
        return feats @ mat[:, :D].T + mat[:, D]
    raise ValueError("Transform matrix has bad dimension %dx%d versus feat dim %d" % (mat.shape + (D,)))


def feat_for_nnet(wav, lda, samp_freq=16000.0, frame_shift_ms=10.0, left=3, right=3):
    """compute_ppg.py:97-134 (allow_downsample: inputs above 16 kHz are resampled first)"""
    if samp_freq > 16000.0:
        wav, samp_freq = resample(wav, samp_freq, 16000.0), 16000.0
    return transform(splice(cmn(mfcc(wav, samp_freq, frame_shift_ms).astype(np.float64)), left, right), lda.astype(np.float64))


def reduce_ppg(ppgs, dense_transform):
    """compute_ppg.py:73-94"""
    return ppgs.astype(np.float64) @ dense_transform.astype(np.float64).T
