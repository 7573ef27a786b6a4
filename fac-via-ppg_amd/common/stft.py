"""STFT / inverse STFT -- drop-in for src/common/stft.py, computed by libfacppg_hip.so.

Same constructor, buffers (``forward_basis``/``inverse_basis`` [n_fft+2, 1, n_fft]) and method
signatures as the reference's ``STFT`` module.  The bases are built analytically on the host
(the reference takes ``fft(eye)`` and a numerical ``pinv``, stft.py:55-63; for the real DFT the
pseudo-inverse is the inverse-rFFT weighting w_c/n with w_c = 1 for DC/Nyquist and 2 otherwise,
tests check the two agree), then handed to ``facppg_stft_create`` once per device.  ``transform`` /
``inverse`` / ``forward`` run the framing gather, MFMA GEMMs and overlap-add kernels of
csrc/facppg_dsp.hip; there is no CPU path.
"""
import numpy as np
import torch
from scipy.signal import get_window

from common.audio_processing import _center_pad, squared_window
from facppg import lib as _lib


def dft_bases(filter_length, hop_length, win_length, window):
    """(forward [n+2, n], inverse [n+2, n]) float32, windowed as stft.py:65-74."""
    n = filter_length
    cutoff = n // 2 + 1
    ang = 2.0 * np.pi * np.outer(np.arange(cutoff), np.arange(n)) / n
    fwd = np.vstack([np.cos(ang), -np.sin(ang)])
    wc = np.full(cutoff, 2.0)
    wc[0] = 1.0
    if n % 2 == 0:
        wc[-1] = 1.0
    scale = n / hop_length
    inv = np.vstack([np.cos(ang) * wc[:, None], -np.sin(ang) * wc[:, None]]) / (n * scale)
    fwd32, inv32 = fwd.astype(np.float32), inv.astype(np.float32)
    if window is not None:
        assert filter_length >= win_length
        win = _center_pad(get_window(window, win_length, fftbins=True), n).astype(np.float32)
        fwd32, inv32 = fwd32 * win, inv32 * win
    return fwd32, inv32


class STFT(torch.nn.Module):
    """stft.py:44-143"""

    def __init__(self, filter_length=800, hop_length=200, win_length=800, window='hann', mel_basis=None):
        super(STFT, self).__init__()
        self.filter_length = filter_length
        self.hop_length = hop_length
        self.win_length = win_length
        self.window = window
        self.forward_transform = None
        fwd, inv = dft_bases(filter_length, hop_length, win_length, window)
        self.register_buffer('forward_basis', torch.from_numpy(fwd[:, None, :]).float())
        self.register_buffer('inverse_basis', torch.from_numpy(inv[:, None, :]).float())
        wsq = squared_window(window, win_length, filter_length) if window is not None else \
            np.zeros(filter_length)   # window=None: no normalisation (stft.py:118)
        self.register_buffer('_win_sq', torch.from_numpy(wsq.astype(np.float32)), persistent=False)
        self._mel_basis_np = mel_basis

    # ------------------------------------------------------------ handle / workspace
    def _handle(self, dev):
        h = self.__dict__.get("_facppg_handle")
        if h is not None and h[1] == dev:
            return h[0]
        self._release()
        L = _lib.load()
        fwd = self.forward_basis.squeeze(1).to(dev).contiguous()
        inv_t = self.inverse_basis.squeeze(1).t().to(dev).contiguous()
        wsq = self._win_sq.to(dev).contiguous()
        mel = None if self._mel_basis_np is None else torch.as_tensor(self._mel_basis_np, dtype=torch.float32).to(dev).contiguous()
        out = _lib.ctypes.c_void_p()
        with torch.cuda.device(dev):
            _lib.check(L.facppg_stft_create(self.filter_length, self.hop_length, _lib.ptr(fwd), _lib.ptr(inv_t), _lib.ptr(wsq),
                                            _lib.ptr(mel), 0 if mel is None else mel.shape[0], dev.index,
                                            _lib.current_stream(dev), _lib.ctypes.byref(out)))
        self.__dict__["_facppg_handle"] = (out, dev)
        return out

    def _release(self):
        h = self.__dict__.pop("_facppg_handle", None)
        if h is not None:
            _lib.load().facppg_stft_destroy(h[0])
        self.__dict__.pop("_facppg_ws", None)

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_facppg_handle", None)
        d.pop("_facppg_ws", None)
        return d

    def _workspace(self, h, dev, B, N):
        nbytes = _lib.load().facppg_stft_workspace_bytes(h, B, N)
        ws = self.__dict__.get("_facppg_ws")
        if ws is None or ws.numel() < nbytes or ws.device != dev:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self.__dict__["_facppg_ws"] = ws
        return ws

    @staticmethod
    def _lengths(lengths, B, N, dev):
        if lengths is None:
            return None
        if torch.is_tensor(lengths):
            lt = lengths.to(device=dev, dtype=torch.int32).contiguous()
            if lt.numel() != B or int(lt.max()) > N:
                raise _lib.FacppgError("lengths must be B sample counts <= N")
            return lt
        if len(lengths) != B or max(int(n) for n in lengths) > N:       # host list: checked here, uploaded without a host stall
            raise _lib.FacppgError("lengths must be B sample counts <= N")
        return _lib.upload([int(n) for n in lengths], torch.int32, dev)

    # ------------------------------------------------------------ reference API
    def transform(self, input_data, lengths=None):
        """[B, N] -> magnitude, phase [B, n_fft/2+1, N//hop+1]  (stft.py:79-107)"""
        _lib.require_cuda(input_data, "STFT.transform: input_data")
        x = input_data.float().contiguous()
        B, N = x.shape
        self.num_samples = N
        dev = x.device
        h = self._handle(dev)
        ws = self._workspace(h, dev, B, N)
        F = N // self.hop_length + 1
        cutoff = self.filter_length // 2 + 1
        mag = torch.zeros(B, cutoff, F, device=dev) if lengths is not None else torch.empty(B, cutoff, F, device=dev)
        phase = torch.zeros_like(mag) if lengths is not None else torch.empty_like(mag)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().facppg_stft_transform(h, _lib.ptr(x), _lib.ptr(self._lengths(lengths, B, N, dev)), B, N,
                                                         _lib.ptr(mag), _lib.ptr(phase), _lib.ptr(ws), ws.numel(),
                                                         _lib.current_stream(dev)))
        return mag, phase

    def inverse(self, magnitude, phase):
        """[B, cutoff, F] x2 -> [B, 1, hop*(F-1)]  (stft.py:109-138)"""
        _lib.require_cuda(magnitude, "STFT.inverse: magnitude")
        mag = magnitude.float().contiguous()
        ph = phase.to(mag.device).float().contiguous()
        B, _, F = mag.shape
        dev = mag.device
        h = self._handle(dev)
        ws = self._workspace(h, dev, B, (F - 1) * self.hop_length)
        out = torch.empty(B, 1, self.hop_length * (F - 1), device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().facppg_stft_inverse(h, _lib.ptr(mag), _lib.ptr(ph), B, F, _lib.ptr(out), _lib.ptr(ws),
                                                       ws.numel(), _lib.current_stream(dev)))
        return out

    def forward(self, input_data):
        self.magnitude, self.phase = self.transform(input_data)
        return self.inverse(self.magnitude, self.phase)

    # ------------------------------------------------------------ fused entry points
    def mel(self, y, lengths=None):
        """log-mel of y [B, N] in one pass (used by TacotronSTFT.mel_spectrogram)."""
        _lib.require_cuda(y, "mel_spectrogram: y")
        x = y.float().contiguous()
        B, N = x.shape
        dev = x.device
        h = self._handle(dev)
        ws = self._workspace(h, dev, B, N)
        n_mel = self._mel_basis_np.shape[0]
        F = N // self.hop_length + 1
        out = torch.zeros(B, n_mel, F, device=dev) if lengths is not None else torch.empty(B, n_mel, F, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().facppg_stft_mel(h, _lib.ptr(x), _lib.ptr(self._lengths(lengths, B, N, dev)), B, N,
                                                   _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.current_stream(dev)))
        return out

    def denoise(self, audio, bias_spec, strength, lengths=None):
        """Spectral subtraction (Denoiser.forward) fused: [B, N] -> [B, 1, hop*(N//hop)]."""
        _lib.require_cuda(audio, "Denoiser: audio")
        x = audio.float().contiguous()
        B, N = x.shape
        dev = x.device
        h = self._handle(dev)
        ws = self._workspace(h, dev, B, N)
        bias = bias_spec.to(dev).float().reshape(-1).contiguous()
        n_out = self.hop_length * (N // self.hop_length)
        out = torch.zeros(B, 1, n_out, device=dev) if lengths is not None else torch.empty(B, 1, n_out, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().facppg_denoise(h, _lib.ptr(x), _lib.ptr(self._lengths(lengths, B, N, dev)), _lib.ptr(bias),
                                                  float(strength), B, N, _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                                  _lib.current_stream(dev)))
        return out
