"""GPU: edge cases of the hot path -- minimal and ragged sizes, the un-windowed attention, and the
C ABI's error behaviour (negative codes + messages, nothing thrown across the boundary)."""
import ctypes

import numpy as np
import pytest
import torch

from helpers import masks_from_seed, rms
from facppg import lib as flib, synth

pytestmark = pytest.mark.gpu


def _wg(n_flows=12, hop=160):
    from waveglow.glow import WaveGlow
    cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=hop, n_flows=n_flows)
    m = WaveGlow.remove_weightnorm(WaveGlow(**cfg))
    sd = synth.waveglow_state_dict(cfg)
    m.load_state_dict(sd)
    return m.cuda().eval(), cfg, sd


@pytest.mark.parametrize("T", [1, 2, 4])
def test_waveglow_tiny_mels_match_oracle(T):
    """One to four mel frames: L = 20..80 group positions, far inside the 255-position receptive
    field of a flow, so every tap of every layer crosses the zero padding."""
    from oracle import waveglow as owg
    m, cfg, sd = _wg()
    mel = synth.synthetic_mel(1, T, seed=T)
    zs = synth.synthetic_z(1, T * 20, cfg, seed=100 + T)
    hip = m.infer(mel.cuda(), sigma=0.6, z=zs).cpu()
    with torch.no_grad():
        ref = owg.infer(sd, cfg, mel, 0.6, zs)
    assert hip.shape == (1, T * 160) and rms((hip - ref).numpy()) <= 1e-3


def test_waveglow_other_flow_counts_and_bad_configs():
    """n_flows is a run-time property of the handle (4 and 8 flows end on different channel counts
    only when early outputs line up with n_group); unsupported shapes are refused, not mis-computed."""
    L = flib.load()
    bad = flib.WgConfig(80, 160, 12, 8, 4, 2, 8, 256, 5, 1024)        # kernel_size 5
    assert L.facppg_wg_weight_count(bad) == 0 and b"kernel_size=3" in L.facppg_last_error()
    bad = flib.WgConfig(80, 100, 12, 8, 4, 2, 8, 256, 3, 1024)        # hop not a multiple of n_group
    assert L.facppg_wg_weight_count(bad) == 0
    m, cfg, sd = _wg()
    with pytest.raises(flib.FacppgError, match="fp32 only"):
        m.infer(torch.zeros(1, 80, 4, dtype=torch.float16, device="cuda"))
    with pytest.raises(flib.FacppgError, match="lengths must be"):
        m.infer(torch.zeros(2, 80, 4, device="cuda"), lengths=[4, 9])
    with pytest.raises(flib.FacppgError, match="z has"):
        m.infer(torch.zeros(1, 80, 4, device="cuda"), z=torch.zeros(7))
    # raw ABI: undersized workspace -> FACPPG_EWORKSPACE (-4), message names both sizes
    h = m._handle(torch.device("cuda", 0))
    mel = torch.zeros(1, 80, 4, device="cuda")
    out = torch.empty(1, 640, device="cuda")
    ws = torch.empty(1024, dtype=torch.uint8, device="cuda")
    rc = L.facppg_wg_infer(h, flib.ptr(mel), None, None, 0, ctypes.c_float(1.0), 1, 4, flib.ptr(out), flib.ptr(ws), 1024,
                           flib.current_stream(torch.device("cuda", 0)))
    assert rc == -4 and b"workspace has 1024 bytes" in L.facppg_last_error()
    rc = L.facppg_wg_infer(h, None, None, None, 0, ctypes.c_float(1.0), 1, 4, flib.ptr(out), flib.ptr(ws), 1024, None)
    assert rc == -1


@pytest.mark.parametrize("Tin,window", [(1, 20), (2, 20), (7, None), (50, None), (50, 3)])
def test_tacotron_sizes_and_unwindowed_attention(Tin, window):
    """Single-frame PPGs, and attention_window_size=None (every encoder step visible, hparams.py:172):
    the decoder then walks all positions in 64-wide chunks."""
    from common.hparams import create_hparams_stage
    from oracle import tacotron as otac
    from script.train_ppg2mel import load_model
    steps = 12
    hp = create_hparams_stage(max_decoder_steps=steps, attention_window_size=window)
    sd = synth.tacotron_state_dict(hp, gate_bias=-10.0)
    m = load_model(hp)
    m.load_state_dict(sd)
    m.eval()
    ppg = synth.synthetic_ppg(Tin, 5816, seed=Tin)
    em = masks_from_seed(1, (2, 1, Tin, 600))
    dm = masks_from_seed(2, (steps, 2, 1, 300))
    x = torch.from_numpy(ppg).t().unsqueeze(0)
    mel, mel_post, gate, align = m.inference(x.cuda(), dropout_masks=(em, dm))
    r_mel, r_post, r_gate, r_align = otac.inference(sd, hp, x, torch.from_numpy(em.astype(np.float32)),
                                                    torch.from_numpy(dm.astype(np.float32)))
    assert mel.shape == r_mel.shape == (1, 80, steps)
    assert np.abs(mel_post.cpu().numpy() - r_post.numpy()).max() <= 1e-4
    assert np.abs(align.cpu().numpy() - r_align.numpy()).max() <= 1e-4
    assert np.allclose(align.cpu().numpy().sum(2), 1.0, atol=1e-5)


def test_stft_minimum_length_and_errors():
    """Reflect padding needs N > n_fft/2 (torch raises for the reference too); N = 513 is the shortest."""
    from common.stft import STFT
    from oracle import dsp
    st = STFT(1024, 160, 1024).cuda()
    g = np.random.Generator(np.random.PCG64(4))
    y = torch.from_numpy(g.standard_normal((1, 513), dtype=np.float32) * 0.1)
    mag, ph = st.transform(y.cuda())
    omag, _ = dsp.StftOracle(1024, 160, 1024).transform(y)
    assert mag.shape == omag.shape == (1, 513, 4) and np.abs(mag.cpu().numpy() - omag.numpy()).max() < 1e-4
    with pytest.raises(flib.FacppgError, match="reflect padding"):
        st.transform(torch.zeros(1, 512, device="cuda"))
    with pytest.raises(flib.FacppgError, match="GPU tensor"):
        st.transform(torch.zeros(1, 2048))


def test_large_sizes_properties():
    """Maximum-size behaviour (properties only, the oracle would take minutes): a 48-second
    utterance through WaveGlow (L = 96 000 positions, 1 500 tiles), batch independence at that size,
    and a 1 500-frame PPG through the encoder/decoder (LDS-resident attention state scales with Tin)."""
    from common.hparams import create_hparams_stage
    from script.train_ppg2mel import load_model
    m, cfg, _ = _wg(hop=256)
    B, T = 2, 3000
    mel = synth.synthetic_mel(B, T, seed=5).cuda()
    a = m.infer(mel, sigma=0.6, seed=9)
    assert a.shape == (B, T * 256) and torch.isfinite(a).all() and float(a.abs().max()) < 1e3
    assert torch.equal(a, m.infer(mel, sigma=0.6, seed=9))
    zs = synth.synthetic_z(B, T * 32, cfg, seed=3)
    full = m.infer(mel, sigma=0.6, z=zs)
    one = m.infer(mel[1:2].contiguous(), sigma=0.6, z=[z[1:2].contiguous() for z in zs])
    assert torch.equal(one[0], full[1])
    steps, Tin = 64, 1500
    hp = create_hparams_stage(max_decoder_steps=steps)
    t = load_model(hp)
    t.load_state_dict(synth.tacotron_state_dict(hp, gate_bias=-10.0))
    t.eval()
    x = torch.from_numpy(synth.synthetic_ppg(Tin, 5816, seed=1)).t().unsqueeze(0).cuda()
    mel_o, mel_post, gate, align = t.inference(x, seed=4)
    assert mel_post.shape == (1, 80, steps) and align.shape == (1, steps, Tin)
    assert torch.isfinite(mel_post).all()
    al = align[0].cpu().numpy()
    assert np.allclose(al.sum(1), 1.0, atol=1e-4)
    for s_ in (0, 30, 63):                                   # support = the +-20 window of utils.py:64-77
        nz = np.nonzero(al[s_])[0]
        assert nz.min() >= max(0, s_ - 20) and nz.max() <= s_ + 20
