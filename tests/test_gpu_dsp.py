"""GPU parity: STFT / inverse STFT / mel / denoiser on the HIP kernels vs the reference's golden
vectors (tests/golden/stft.npz, denoiser_hop160.npz).  Tolerances: mel <= 1e-4 abs (north_star),
waveform RMS <= 1e-3; frame counts N//hop+1 and output lengths hop*(F-1) exact."""
import numpy as np
import pytest
import torch

from helpers import golden, rms
from facppg import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hop", [160, 256])
def test_stft_transform_inverse_golden(hop):
    from common.stft import STFT
    s = golden("stft.npz")
    y = torch.from_numpy(s["y"]).cuda()
    st = STFT(1024, hop, 1024).cuda()
    mag, ph = st.transform(y)
    assert mag.shape == (2, 513, y.shape[1] // hop + 1)
    assert np.abs(mag.cpu().numpy() - s["mag_%d" % hop]).max() < 1e-4
    # phase is ill-conditioned where the magnitude is ~0; compare the complex spectrum instead
    re, im = (mag * torch.cos(ph)).cpu().numpy(), (mag * torch.sin(ph)).cpu().numpy()
    gre, gim = s["mag_%d" % hop] * np.cos(s["phase_%d" % hop]), s["mag_%d" % hop] * np.sin(s["phase_%d" % hop])
    assert max(np.abs(re - gre).max(), np.abs(im - gim).max()) < 1e-4
    rec = st.inverse(mag, ph)
    assert rec.shape == (2, 1, hop * (mag.shape[2] - 1))
    assert np.abs(rec.cpu().numpy() - s["rec_%d" % hop]).max() < 1e-4
    # KAT (SURVEY section 4 ii): inverse(transform(x)) ~= x away from the edges
    n = rec.shape[2]
    assert np.abs(rec.cpu().numpy()[:, 0, :n] - s["y"][:, :n]).max() < 1e-4
    assert torch.equal(st(y), rec)


def test_mel_spectrogram_golden():
    from common.layers import TacotronSTFT
    s = golden("stft.npz")
    y = torch.from_numpy(s["y"]).cuda()
    ts = TacotronSTFT(1024, 160, 1024, 80, 16000, 0.0, 8000.0).cuda()
    assert np.abs(ts.mel_basis.cpu().numpy() - s["mel_basis_16k"]).max() < 1e-7
    mel = ts.mel_spectrogram(y)
    assert mel.shape == (2, 80, 4000 // 160 + 1)
    err = np.abs(mel.cpu().numpy() - s["mel_16k"]).max()
    print("mel max abs err", err)
    assert err <= 1e-4
    ts2 = TacotronSTFT().cuda()         # library defaults 22.05 kHz / hop 256
    assert np.abs(ts2.mel_spectrogram(y).cpu().numpy() - s["mel_22k"]).max() <= 1e-4
    with pytest.raises(AssertionError):
        ts.mel_spectrogram(y * 10)      # layers.py:105-106 range assert


def _wg():
    from waveglow.glow import WaveGlow
    cfg = dict(synth.WAVEGLOW_CONFIG)
    sd = synth.waveglow_state_dict(cfg)
    keep = WaveGlow(**cfg)              # weight norm still attached, as generate_synthesis.py:58-61
    # express the synthetic plain weights as (g, v): g = ||v|| per output channel, v = w
    wn_sd = {}
    for k, v in sd.items():
        if k.startswith("WN.") and k.endswith(".weight") and ".end." not in k:
            wn_sd[k[:-6] + "weight_v"] = v
            wn_sd[k[:-6] + "weight_g"] = v.flatten(1).norm(dim=1).view(-1, 1, 1)
        else:
            wn_sd[k] = v
    keep.load_state_dict(wn_sd, strict=True)
    return keep.cuda()


def test_denoiser_golden():
    from waveglow.denoiser import Denoiser
    d = golden("denoiser_hop160.npz")
    den = Denoiser(_wg(), mode="zeros")
    assert den.bias_spec.shape == (1, 513, 1)
    assert np.abs(den.bias_spec.cpu().numpy() - d["bias_spec"]).max() < 1e-3 * max(1.0, np.abs(d["bias_spec"]).max())
    x = torch.from_numpy(d["audio_in"]).cuda()
    for strength, key in ((0.005, "out_0005"), (1.0, "out_1")):
        out = den(x, strength=strength)
        assert out.shape == d[key].shape
        e = out.cpu().numpy() - d[key]
        print("denoiser strength", strength, "rms err", rms(e), "max", np.abs(e).max())
        assert rms(e) <= 1e-3 and np.abs(e).max() <= 1e-3
    with pytest.raises(Exception, match="not supported"):
        Denoiser(_wg(), mode="bogus")


def test_denoiser_hop256_golden():
    """Denoiser(hop_length=256) -- what bench.py's end-to-end figures and the 22.05 kHz metric use -- vs the reference's own
    hop-256 denoiser on the hop-256 model: bias spectrum and both strengths."""
    from waveglow.denoiser import Denoiser
    from waveglow.glow import WaveGlow
    d = golden("denoiser_hop256.npz")
    hop = int(d["hop"])
    cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=hop)
    m = WaveGlow.remove_weightnorm(WaveGlow(**cfg))
    m.load_state_dict(synth.waveglow_state_dict(cfg))
    den = Denoiser(m.cuda().eval(), filter_length=1024, hop_length=hop, win_length=1024, mode="zeros")
    assert den.bias_spec.shape == (1, 513, 1)
    assert np.abs(den.bias_spec.cpu().numpy() - d["bias_spec"]).max() < 1e-3 * max(1.0, np.abs(d["bias_spec"]).max())
    x = torch.from_numpy(d["audio_in"]).cuda()
    for strength, key in ((0.005, "out_0005"), (1.0, "out_1")):
        out = den(x, strength=strength)
        assert out.shape == d[key].shape
        e = out.cpu().numpy() - d[key]
        print("denoiser hop 256, strength", strength, "rms err", rms(e), "max", np.abs(e).max())
        assert rms(e) <= 1e-3 and np.abs(e).max() <= 1e-3


def test_denoiser_ragged_batch_equals_single_runs():
    from waveglow.denoiser import Denoiser
    den = Denoiser(_wg(), mode="zeros")
    g = np.random.Generator(np.random.PCG64(3))
    lens = [4000, 1600, 2720]
    x = torch.from_numpy(g.standard_normal((3, 4000), dtype=np.float32) * 0.2).cuda()
    out = den(x, strength=0.1, lengths=lens)
    for b, n in enumerate(lens):
        single = den(x[b:b + 1, :n].contiguous(), strength=0.1)
        assert torch.equal(single[0, 0], out[b, 0, :n])
