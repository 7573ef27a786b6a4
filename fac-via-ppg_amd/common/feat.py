"""Acoustic-feature helpers of the PPG front-end -- drop-in for the blob-free part of src/common/feat.py (:29-170), on
libfacppg_hip instead of pykaldi.  Kaldi objects become plain tensors: a "Matrix" is a float32 GPU tensor [T, D].

  read_wav_kaldi / read_wav_kaldi_internal   wav file / array -> WaveData (first channel, int16-range floats, on the GPU)
  MfccOptions, compute_mfcc_feats            Kaldi MFCC with the reference's options (facppg_mfcc_*)
  apply_cepstral_mean_norm, splice_frames, apply_feat_transform          (facppg_cmn_splice_transform)
  read_sparse_mat                            Kaldi sparse matrix -> dense float32 GPU tensor

Dither is not applied (Kaldi's default adds random +-1 LSB noise; this path is deterministic); inputs above 16 kHz are
downsampled with Kaldi's LinearResample when allow_downsample is set (facppg_resample), as the reference does."""
import logging

import numpy as np
import torch
from scipy.io import wavfile

from common import kaldi_io
from facppg import lib as _lib


class WaveData(object):
    """What the reference gets from kaldi.feat.wave.WaveData: ``data()`` [channels = 1, N] and ``samp_freq``."""

    def __init__(self, samp_freq, data):
        self.samp_freq = float(samp_freq)
        self._data = data

    def data(self):
        return self._data

    @property
    def duration(self):
        return self._data.shape[1] / self.samp_freq


def read_wav_kaldi_internal(wav, fs):
    """feat.py:29-56: numpy samples [N] or [N, channels] (int16 range) -> WaveData holding the first channel only."""
    wav = np.asarray(wav)
    if wav.ndim == 2:
        wav = wav[:, 0]
    if wav.ndim != 1:
        raise ValueError("wav must be [samples] or [samples, channels]")
    data = torch.from_numpy(np.ascontiguousarray(wav, dtype=np.float32))[None]
    return WaveData(fs, data.cuda() if torch.cuda.is_available() else data)


def read_wav_kaldi(wav_file_path):
    """feat.py:59-71"""
    fs, wav = wavfile.read(wav_file_path, False)
    return read_wav_kaldi_internal(wav, fs)


class FrameExtractionOptions(object):
    """kaldi.feat.window.FrameExtractionOptions defaults (feature-window.h)."""

    def __init__(self):
        self.samp_freq = 16000.0
        self.frame_shift_ms = 10.0
        self.frame_length_ms = 25.0
        self.dither = 1.0
        self.preemph_coeff = 0.97
        self.remove_dc_offset = True
        self.window_type = "povey"
        self.round_to_power_of_two = True
        self.snip_edges = True
        self.allow_downsample = False


class MelBanksOptions(object):
    def __init__(self):
        self.num_bins = 23
        self.low_freq = 20.0
        self.high_freq = 0.0


class MfccOptions(object):
    """kaldi.feat.mfcc.MfccOptions defaults (feature-mfcc.h)."""

    def __init__(self):
        self.frame_opts = FrameExtractionOptions()
        self.mel_opts = MelBanksOptions()
        self.num_ceps = 13
        self.use_energy = True
        self.cepstral_lifter = 22.0


class Mfcc(object):
    """kaldi.feat.mfcc.Mfcc: folds DC removal, pre-emphasis, povey window, zero padding and the DFT into one matrix on the
    host (float64), and hands it with the mel bank and the liftered DCT to facppg_mfcc_create."""

    def __init__(self, opts):
        fo = opts.frame_opts
        if fo.snip_edges:
            raise _lib.FacppgError("Mfcc: only snip_edges=False is built (what the reference uses, compute_ppg.py:106-121)")
        if fo.window_type != "povey" or not fo.remove_dc_offset or not fo.round_to_power_of_two:
            raise _lib.FacppgError("Mfcc: only Kaldi's default povey window / DC removal / power-of-two padding are built")
        if fo.dither != 0.0 and not getattr(Mfcc, "_dither_noted", False):
            logging.info("Mfcc: dither is not applied (deterministic features)")
            Mfcc._dither_noted = True
        self.opts = opts
        fs = fo.samp_freq
        self.length, self.shift = int(fs * 0.001 * fo.frame_length_ms), int(fs * 0.001 * fo.frame_shift_ms)
        n_fft = 1 << (self.length - 1).bit_length()
        self.nbins = n_fft // 2 + 1
        n = self.length
        i = np.arange(n)
        window = (0.5 - 0.5 * np.cos(2 * np.pi * i / (n - 1))) ** 0.85
        dc = np.eye(n) - np.full((n, n), 1.0 / n)                              # x - mean(x)
        pre = np.eye(n) - fo.preemph_coeff * np.eye(n, k=-1)
        pre[0, 0] -= fo.preemph_coeff                                          # w[0] -= c * w[0]
        lin = (window[:, None] * pre) @ dc                                     # [n, n]: frame -> windowed, pre-emphasised frame
        k = np.arange(self.nbins)
        ang = -2.0 * np.pi * np.outer(k, i) / n_fft
        basis = np.concatenate([np.cos(ang) @ lin, np.sin(ang) @ lin], 0)      # [2*nbins, n]
        mo = opts.mel_opts
        nyq = 0.5 * fs
        hi_f = mo.high_freq if mo.high_freq > 0 else nyq + mo.high_freq
        ms = lambda f: 1127.0 * np.log(1.0 + f / 700.0)
        lo, hi = ms(mo.low_freq), ms(hi_f)
        delta = (hi - lo) / (mo.num_bins + 1)
        mel_f = ms(fs / n_fft * np.arange(n_fft // 2))
        mel = np.zeros((mo.num_bins, self.nbins))                              # the Nyquist column stays zero (Kaldi skips it)
        for b in range(mo.num_bins):
            left, center, right = lo + b * delta, lo + (b + 1) * delta, lo + (b + 2) * delta
            up, dn = (mel_f > left) & (mel_f <= center), (mel_f > center) & (mel_f < right)
            mel[b, :n_fft // 2][up] = (mel_f[up] - left) / (center - left)
            mel[b, :n_fft // 2][dn] = (right - mel_f[dn]) / (right - center)
        dct = np.zeros((opts.num_ceps, mo.num_bins))
        dct[0] = np.sqrt(1.0 / mo.num_bins)
        nn = np.arange(mo.num_bins)
        for c in range(1, opts.num_ceps):
            dct[c] = np.sqrt(2.0 / mo.num_bins) * np.cos(np.pi / mo.num_bins * (nn + 0.5) * c)
        if opts.cepstral_lifter != 0.0:
            dct *= (1.0 + 0.5 * opts.cepstral_lifter * np.sin(np.pi * np.arange(opts.num_ceps) / opts.cepstral_lifter))[:, None]
        self._tables = [torch.from_numpy(a.astype(np.float32)).contiguous() for a in (basis, mel, dct)]
        self._handle = None

    def _get(self, dev):
        if self._handle is None or self._handle[1] != dev:
            self._release()
            L = _lib.load()
            basis, mel, dct = [t.to(dev) for t in self._tables]
            out = _lib.ctypes.c_void_p()
            with torch.cuda.device(dev):
                _lib.check(L.facppg_mfcc_create(self.length, self.shift, self.nbins, _lib.ptr(basis), _lib.ptr(mel), mel.shape[0], _lib.ptr(dct),
                                                dct.shape[0], dev.index, _lib.current_stream(dev), _lib.ctypes.byref(out)))
            self._handle = (out, dev)
        return self._handle[0]

    def _release(self):
        if getattr(self, "_handle", None) is not None:
            _lib.load().facppg_mfcc_destroy(self._handle[0])
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def compute_features(self, wave, samp_freq, vtln_warp=1.0):
        """wave: [N] samples (GPU tensor, int16 range) -> [T, num_ceps] (GPU)."""
        _lib.require_cuda(wave, "Mfcc.compute_features: wave")
        if vtln_warp != 1.0:
            raise _lib.FacppgError("VTLN warping is not built (the reference always passes 1.0, feat.py:95)")
        L = _lib.load()
        dev = wave.device
        wave = wave.float().contiguous().reshape(-1)
        target = self.opts.frame_opts.samp_freq
        if float(samp_freq) != float(target):
            # feature-common-inl.h ComputeFeatures: a higher input rate is downsampled when allow_downsample, anything else is an error
            if float(samp_freq) < float(target) or not self.opts.frame_opts.allow_downsample:
                raise _lib.FacppgError("Mfcc: waveform sampled at %g Hz, features need %g Hz (set frame_opts.allow_downsample for higher rates)"
                                       % (samp_freq, target))
            n_out = L.facppg_resample_num_samples(wave.numel(), int(samp_freq), int(target))
            res = torch.empty(n_out, device=dev)
            with torch.cuda.device(dev):
                _lib.check(L.facppg_resample(_lib.ptr(wave), wave.numel(), int(samp_freq), int(target), _lib.ptr(res), _lib.current_stream(dev)))
            wave = res
        h = self._get(dev)
        n = wave.numel()
        T = L.facppg_mfcc_num_frames(h, n)
        out = torch.empty(T, self.opts.num_ceps, device=dev)
        ws = torch.empty(L.facppg_mfcc_workspace_bytes(h, n), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.facppg_mfcc_compute(h, _lib.ptr(wave), n, 1 if self.opts.use_energy else 0, _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                             _lib.current_stream(dev)))
        return out


def compute_mfcc_feats(wav, mfcc_opts):
    """feat.py:74-100: T*D MFCCs of the first channel."""
    return Mfcc(mfcc_opts).compute_features(wav.data()[0], wav.samp_freq, 1.0)


def _cst(feats, do_cmn, left, right, transform):
    _lib.require_cuda(feats, "feature matrix")
    L = _lib.load()
    dev = feats.device
    feats = feats.float().contiguous()
    T, D = feats.shape
    W = (left + right + 1) * D
    tr = None if transform is None else transform.to(dev).float().contiguous()
    if tr is not None and tr.shape[1] not in (W, W + 1):
        logging.error("Transform matrix has bad dimension %dx%d versus feat dim %d" % (tr.shape[0], tr.shape[1], W))   # feat.py:154-155
        raise _lib.FacppgError("Transform matrix has bad dimension %dx%d versus feat dim %d" % (tr.shape[0], tr.shape[1], W))
    out = torch.empty(T, W if tr is None else tr.shape[0], device=dev)
    mean = torch.empty(D, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.facppg_cmn_splice_transform(_lib.ptr(feats), T, D, 1 if do_cmn else 0, left, right, _lib.ptr(tr), 0 if tr is None else tr.shape[0],
                                                 0 if tr is None else tr.shape[1], _lib.ptr(out), _lib.ptr(mean), _lib.current_stream(dev)))
    return out


def apply_cepstral_mean_norm(feats):
    """feat.py:103-118 (mean only)."""
    return _cst(feats, True, 0, 0, None)


def splice_frames(feats, left_context, right_context):
    """kaldi.feat.functions.splice_frames (compute_ppg.py:128): edge frames are replicated."""
    return _cst(feats, False, int(left_context), int(right_context), None)


def apply_feat_transform(feats, transform):
    """feat.py:121-156: linear (D' x D) or affine (D' x (D+1)) transform."""
    return _cst(feats, False, 0, 0, torch.as_tensor(transform))


def cmn_splice_transform(feats, left_context, right_context, transform):
    """The three steps of compute_ppg.py:124-132 in one pass."""
    return _cst(feats, True, int(left_context), int(right_context), torch.as_tensor(transform))


def read_sparse_mat(sparse_mat_dir):
    """feat.py:159-170; returned densified (float32 tensor [rows, cols]), which is how the reference uses it."""
    return torch.from_numpy(kaldi_io.read_sparse_matrix(sparse_mat_dir))
