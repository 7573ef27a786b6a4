"""GPU: the stand-alone module surface of waveglow.glow (WN.forward, Invertible1x1Conv.forward, glow.py:82-102,
154-175) on the HIP kernels vs the CPU oracle / plain torch fp32 of the same op, with gradients; and the packed-handle
cache following in-place weight updates (optimizer steps between two inferences)."""
import numpy as np
import pytest
import torch

from facppg import lib as flib, synth
from helpers import rms

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("c", [4, 6, 8])
@pytest.mark.parametrize("L", [1250, 333])          # multiple of 4 (16-byte path) and not
def test_invertible1x1conv_forward_reverse_and_gradients(c, L):
    from waveglow.glow import Invertible1x1Conv
    torch.manual_seed(c * 1000 + L)
    m = Invertible1x1Conv(c)
    m.conv.weight.data += 0.1 * torch.randn(c, c, 1)        # not orthonormal: inverse != transpose
    W = m.conv.weight.detach().squeeze(-1).clone()
    z = torch.randn(3, c, L)
    r = torch.randn(3, c, L)
    # plain torch fp32 reference of the same op (glow.py:98-102) on the CPU
    Wr, zr = W.clone().requires_grad_(True), z.clone().requires_grad_(True)
    ref = torch.nn.functional.conv1d(zr, Wr[..., None])
    ref_logdet = 3 * L * torch.logdet(Wr)
    ((ref * r).sum() + ref_logdet).backward()
    m = m.cuda()
    zc = z.cuda().requires_grad_(True)
    out, logdet = m(zc)
    assert out.is_cuda and out.shape == (3, c, L)
    assert (out.detach().cpu() - ref.detach()).abs().max() <= 1e-5
    assert abs(float(logdet) - float(ref_logdet)) <= 1e-3 * max(1.0, abs(float(ref_logdet)))
    ((out * r.cuda()).sum() + logdet).backward()
    assert (zc.grad.cpu() - zr.grad).abs().max() <= 1e-5
    gw = m.conv.weight.grad.squeeze(-1).cpu()
    assert (gw - Wr.grad).abs().max() <= 2e-4 * max(1.0, float(Wr.grad.abs().max()))
    with torch.no_grad():
        back = m(out.detach(), reverse=True)                 # W^-1 (W z) = z
        assert (back.cpu() - z).abs().max() <= 1e-4
        assert m.W_inverse.shape == (c, c, 1)
        # in-place weight change: the cached inverse must follow (stale W_inverse)
        m.conv.weight.mul_(2.0)
        assert (m(out.detach(), reverse=True).cpu() - 0.5 * z).abs().max() <= 1e-4
    with pytest.raises(flib.FacppgError, match="GPU tensor"):
        m(z)


@pytest.mark.parametrize("n_in,L", [(4, 150), (2, 64), (3, 1)])
def test_wn_forward_matches_oracle_and_backward_matches_torch(n_in, L):
    """WN.forward on (audio [B, n_in, L], spect [B, 640, L]) vs oracle.waveglow.wn_forward (glow.py:154-175) and the
    gradients w.r.t. both inputs and a few weights vs torch autograd through the oracle."""
    from oracle import waveglow as owg
    from waveglow.glow import WN
    cfg = dict(synth.WAVEGLOW_CONFIG)
    sd = synth.waveglow_state_dict(cfg)
    k = {4: 0, 3: 4, 2: 8}[n_in]                              # a flow with that half-width
    wn = WN(n_in, 640, **cfg["WN_config"])
    torch.nn.utils.remove_weight_norm(wn.start)
    for lst in (wn.in_layers, wn.cond_layers, wn.res_skip_layers):
        for conv in lst:
            torch.nn.utils.remove_weight_norm(conv)
    wn.load_state_dict({key[len("WN.%d." % k):]: v for key, v in sd.items() if key.startswith("WN.%d." % k)}, strict=True)
    g = np.random.Generator(np.random.PCG64(L))
    audio = torch.from_numpy(g.standard_normal((2, n_in, L), dtype=np.float32))
    spect = torch.from_numpy(g.standard_normal((2, 640, L), dtype=np.float32))
    r = torch.from_numpy(g.standard_normal((2, 2 * n_in, L), dtype=np.float32))
    sdr = {key: v.clone().requires_grad_(True) for key, v in sd.items() if key.startswith("WN.%d." % k)}
    ar, sr = audio.clone().requires_grad_(True), spect.clone().requires_grad_(True)
    ref = owg.wn_forward(sdr, k, cfg, ar, sr)
    (ref * r).sum().backward()
    wn = wn.cuda()
    ac, sc = audio.cuda().requires_grad_(True), spect.cuda().requires_grad_(True)
    out = wn((ac, sc))
    assert out.shape == ref.shape
    e = (out.detach().cpu() - ref.detach()).abs().max().item()
    (out * r.cuda()).sum().backward()
    ea = (ac.grad.cpu() - ar.grad).abs().max().item() / max(1e-6, ar.grad.abs().max().item())
    es = (sc.grad.cpu() - sr.grad).abs().max().item() / max(1e-6, sr.grad.abs().max().item())
    print("WN n_in=%d L=%d: out %.2e, d audio %.2e, d spect %.2e (relative)" % (n_in, L, e, ea, es))
    assert e <= 1e-4 and ea <= 1e-4 and es <= 1e-4
    for name in ("start.weight", "in_layers.3.weight", "cond_layers.7.bias", "res_skip_layers.7.weight", "end.weight"):
        got = dict(wn.named_parameters())[name].grad.cpu()
        exp = sdr["WN.%d.%s" % (k, name)].grad
        assert (got - exp).abs().max() <= 2e-4 * max(1e-3, float(exp.abs().max())), name
    with pytest.raises(flib.FacppgError, match="n_channels=256"):
        WN(4, 640, n_layers=8, n_channels=512, kernel_size=3).cuda()((ac.detach(), sc.detach()))


def test_handles_follow_inplace_weight_updates():
    """train -> validate -> train -> validate.  The packed-weight handles must be rebuilt after
    an optimizer step / p.data.copy_() -- infer() has to equal a fresh model loaded with the current weights."""
    from common.hparams import create_hparams_stage
    from script.train_ppg2mel import load_model
    from waveglow.glow import WaveGlow, WaveGlowLoss
    cfg = dict(synth.WAVEGLOW_CONFIG, n_flows=4)
    m = WaveGlow(**cfg).cuda()
    with torch.no_grad():
        for wn in m.WN:                                      # non-trivial coupling (end layers start at zero)
            wn.end.weight.normal_(0, 0.02)
    mel = synth.synthetic_mel(1, 8, seed=3).cuda()
    zs = synth.synthetic_z(1, 8 * 20, cfg, seed=4)
    a0 = m.infer(mel, sigma=0.6, z=zs)
    opt = torch.optim.SGD(m.parameters(), lr=1e-2)
    wav = torch.from_numpy(np.random.Generator(np.random.PCG64(1)).standard_normal((1, 1280), dtype=np.float32) * 0.1).cuda()
    m.zero_grad()
    WaveGlowLoss(0.7071)(m((synth.synthetic_mel(1, 9, seed=5).cuda(), wav))).backward()
    opt.step()
    a1 = m.infer(mel, sigma=0.6, z=zs)                       # must see the stepped weights
    fresh = WaveGlow(**cfg)
    fresh.load_state_dict(m.state_dict())
    a_fresh = fresh.cuda().infer(mel, sigma=0.6, z=zs)
    assert torch.equal(a1, a_fresh) and not torch.equal(a1, a0)
    with torch.no_grad():                                    # forward() under no_grad uses the same cached handle
        z1 = m((synth.synthetic_mel(1, 9, seed=5).cuda(), wav))[0]
        zf = fresh((synth.synthetic_mel(1, 9, seed=5).cuda(), wav))[0]
    assert torch.equal(z1, zf)
    # audio with a ragged tail: the reference's unfold drops it (glow.py:224); same under no_grad and with grad
    with torch.no_grad():
        zt = m((synth.synthetic_mel(1, 9, seed=5).cuda(), torch.cat([wav, wav[:, :5]], 1)))[0]
    assert torch.equal(zt, z1)
    # Tacotron2: p.data.copy_() on a decoder weight
    hp = create_hparams_stage(max_decoder_steps=6)
    t = load_model(hp)
    t.load_state_dict(synth.tacotron_state_dict(hp, gate_bias=-10.0))
    t.eval()
    x = torch.from_numpy(synth.synthetic_ppg(9, 5816, seed=1)).t().unsqueeze(0).cuda()
    m0 = t.inference(x, seed=5)[1].clone()
    with torch.no_grad():
        t.decoder.linear_projection.linear_layer.bias.add_(0.25)         # in-place op: version counter moves
    m1 = t.inference(x, seed=5)[1]
    assert not torch.equal(m0, m1)
    t2 = load_model(hp)
    t2.load_state_dict(t.state_dict())
    assert torch.equal(t2.eval().inference(x, seed=5)[1], m1)
    # a write through .data is invisible to the version counter: covered by the train()/eval() switch and the explicit call
    saved = t.decoder.linear_projection.linear_layer.bias.detach().clone()
    t.decoder.linear_projection.linear_layer.bias.data.add_(0.25)
    t.train()
    t.eval()
    m2 = t.inference(x, seed=5)[1].clone()
    assert not torch.equal(m2, m1)
    t.decoder.linear_projection.linear_layer.bias.data.copy_(saved)
    t.invalidate_packed_weights()
    assert torch.equal(t.inference(x, seed=5)[1], m1)


def test_training_path_rejects_unsupported_configs():
    """the training kernels hard-code the WN shape; other configs must raise, not compute garbage."""
    from waveglow.glow import WaveGlow
    cfg = dict(synth.WAVEGLOW_CONFIG, n_flows=4, WN_config={"n_layers": 8, "n_channels": 512, "kernel_size": 3})
    m = WaveGlow(**cfg).cuda()
    mel, wav = synth.synthetic_mel(1, 9, seed=5).cuda(), torch.zeros(1, 1280, device="cuda")
    with pytest.raises(flib.FacppgError):
        m((mel, wav))
    cfg = dict(synth.WAVEGLOW_CONFIG, n_flows=4, WN_config={"n_layers": 8, "n_channels": 256, "kernel_size": 5})
    with pytest.raises(flib.FacppgError):
        WaveGlow(**cfg).cuda()((mel, wav))


def test_two_handles_on_two_streams_concurrently():
    """Two Tacotron handles decoding at the same time on two streams (each a cooperative launch with its own exchange
    words) plus a WaveGlow inference on a third: results equal the serial runs."""
    from common.hparams import create_hparams_stage
    from script.train_ppg2mel import load_model
    from waveglow.glow import WaveGlow
    hp = create_hparams_stage(max_decoder_steps=40)
    sd = synth.tacotron_state_dict(hp, gate_bias=-10.0)
    models = []
    for _ in range(2):
        t = load_model(hp)
        t.load_state_dict(sd)
        models.append(t.eval())
    xs = [torch.from_numpy(synth.synthetic_ppg(30 + 7 * i, 5816, seed=10 + i)).t().unsqueeze(0).cuda() for i in range(2)]
    cfg = dict(synth.WAVEGLOW_CONFIG, n_flows=4)
    wg = WaveGlow.remove_weightnorm(WaveGlow(**cfg)).cuda().eval()
    mel = synth.synthetic_mel(1, 30, seed=2).cuda()
    serial = [m.inference(x, seed=3 + i)[1].clone() for i, (m, x) in enumerate(zip(models, xs))]
    a_serial = wg.infer(mel, sigma=0.6, seed=1).clone()
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    outs = [None, None]
    for rep in range(3):
        for i, (m, x) in enumerate(zip(models, xs)):
            with torch.cuda.stream(streams[i]):
                outs[i] = m.inference(x, seed=3 + i)[1]
        with torch.cuda.stream(streams[2]):
            a = wg.infer(mel, sigma=0.6, seed=1)
        torch.cuda.synchronize()
        assert torch.equal(outs[0], serial[0]) and torch.equal(outs[1], serial[1]) and torch.equal(a, a_serial)


def test_logdet_kernel_matches_torch():
    """facppg_logdet (LU with partial pivoting in one launch) vs torch.logdet and its gradient W^-T, for the mixing-matrix
    sizes the flows use, a matrix that needs pivoting, a negative determinant (NaN) and a singular matrix (-inf)."""
    import torch
    from waveglow.glow import _LogDetFunction
    g = torch.Generator().manual_seed(7)
    for c in (2, 4, 6, 8):
        W = torch.linalg.qr(torch.randn(c, c, generator=g))[0] * 1.3 + 0.05 * torch.randn(c, c, generator=g)
        if torch.det(W) < 0:
            W[:, 0] = -W[:, 0]
        Wg = W.cuda().requires_grad_(True)
        out = _LogDetFunction.apply(Wg)
        out.backward()
        Wr = W.double().requires_grad_(True)
        ref = torch.logdet(Wr)
        ref.backward()
        assert abs(float(out) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
        assert torch.allclose(Wg.grad.cpu().double(), Wr.grad, rtol=1e-4, atol=1e-5)
    P = torch.tensor([[0.0, 2.0, 0.0], [0.0, 0.0, 3.0], [4.0, 0.0, 0.0]])       # zero diagonal: needs row exchanges, det = +24
    assert abs(float(_LogDetFunction.apply(P.cuda())) - float(torch.log(torch.tensor(24.0)))) < 1e-5
    N = torch.diag(torch.tensor([1.0, -2.0, 3.0]))
    assert torch.isnan(_LogDetFunction.apply(N.cuda()))
    S = torch.tensor([[1.0, 2.0], [2.0, 4.0]])
    assert float(_LogDetFunction.apply(S.cuda())) == float("-inf")


def test_glow_loss_node_and_segment_sums_match_the_reference_formula():
    """WaveGlowLoss (glow.py:43-59) on CUDA tensors is one autograd node over facppg_segment_sums; value and every gradient
    (z, each log_s -- a channel slice of a larger tensor, as the flows hand them over -- and each log_det_W) equal the
    reference's chain of torch reductions."""
    from waveglow.glow import WaveGlowLoss, _GlowLossFunction
    g = torch.Generator().manual_seed(5)
    B, L = 3, 1250
    z = torch.randn(B, 8, L, generator=g).cuda().requires_grad_()
    outs = [torch.randn(B, 2 * h, L, generator=g).cuda().requires_grad_() for h in (4, 4, 3, 3, 2)]
    dets = [torch.randn((), generator=g).cuda().requires_grad_() for _ in outs]
    ref_in = [t.detach().clone().requires_grad_() for t in [z] + outs + dets]

    def reference(z_, outs_, dets_, sigma):
        ls = [o[:, o.shape[1] // 2:, :] for o in outs_]
        loss = torch.sum(z_ * z_) / (2 * sigma * sigma) - sum(torch.sum(t) for t in ls) - sum(dets_)
        return loss / z_.numel()

    crit = WaveGlowLoss(sigma=0.8)
    seen = []
    orig = _GlowLossFunction.apply
    _GlowLossFunction.apply = staticmethod(lambda *a: (seen.append(1), orig(*a))[1])
    try:
        loss = crit((z, [o[:, o.shape[1] // 2:, :] for o in outs], [d * 7.0 for d in dets]))
    finally:
        _GlowLossFunction.apply = orig
    assert seen, "the fused node did not run"
    want = reference(ref_in[0], ref_in[1:6], [d * 7.0 for d in ref_in[6:]], 0.8)
    assert abs(float(loss) - float(want)) <= 1e-6 * max(1.0, abs(float(want)))
    (loss * 3.0).backward()
    (want * 3.0).backward()
    for a, b in zip([z] + outs + dets, ref_in):
        assert torch.allclose(a.grad, b.grad, rtol=1e-6, atol=1e-9), (a.shape, (a.grad - b.grad).abs().max())
    # CPU tensors take the reference's own chain of torch ops
    zc = torch.randn(2, 8, 10)
    assert torch.isfinite(crit((zc, [torch.randn(2, 4, 10)], [torch.tensor(0.3)])))
