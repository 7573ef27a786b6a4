"""GPU parity: WaveGlow.infer on the HIP kernels (through the C ABI) vs the golden vectors
captured from the reference and vs the CPU oracle.  Tolerance (BASELINE.json north_star):
waveform RMS error <= 1e-3; integer output length T*hop exact."""
import numpy as np
import pytest
import torch

from helpers import golden, rms
from facppg import synth

pytestmark = pytest.mark.gpu

RMS_TOL = 1e-3


def _check_expected_launch_shape(m):
    """test_alternate_kernel_paths_match_golden sets a launch-shape switch and FACPPG_TEST_EXPECT_SHAPE =
    "<frames per tile>x<waves>": the switch must really have selected that kernel."""
    import os
    want = os.environ.get("FACPPG_TEST_EXPECT_SHAPE")
    if want:
        tile, waves, _ = m.last_launch_shape()
        assert "%dx%d" % (tile, waves) == want, "expected tile x waves %s, the launch used %dx%d" % (want, tile, waves)


def make_model(hop, n_flows=12):
    from waveglow.glow import WaveGlow
    cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=hop, n_flows=n_flows)
    m = WaveGlow(**cfg)
    m = WaveGlow.remove_weightnorm(m)
    m.load_state_dict(synth.waveglow_state_dict(cfg), strict=True)
    return m.cuda().eval(), cfg


@pytest.fixture(scope="module")
def model160():
    return make_model(160)


def _golden_check(m, cfg, tag, hop):
    d = golden("waveglow_%s.npz" % tag)
    B, T = int(d["B"]), int(d["T"])
    mel = synth.synthetic_mel(B, T, seed=int(d["mel_seed"])).cuda()
    zs = synth.synthetic_z(B, T * hop // 8, cfg, seed=int(d["z_seed"]))
    audio = m.infer(mel, sigma=float(d["sigma"]), z=zs).cpu().numpy()
    assert audio.shape == (B, T * hop)
    _check_expected_launch_shape(m)
    err = audio - d["audio"]
    print("max abs err", np.abs(err).max(), "rms err", rms(err), "rms ref", rms(d["audio"]))
    assert rms(err) <= RMS_TOL
    assert np.abs(err).max() <= 5e-3


@pytest.fixture(scope="module")
def model256():
    return make_model(256)


@pytest.mark.parametrize("tag,hop", [("hop160", 160), ("hop256", 256)])
def test_infer_matches_reference_golden(tag, hop, model160, model256):
    m, cfg = model160 if hop == 160 else model256
    _golden_check(m, cfg, tag, hop)


def _ragged_check(m, cfg):
    """Padded batch with per-utterance lengths == B independent batch-1 runs (bit-exact), and each
    matches the oracle on the unpadded mel.  Lengths chosen so L is not a multiple of the 64-wide
    tile and the receptive field (255 positions) crosses utterance ends."""
    from oracle import waveglow as owg
    sd = synth.waveglow_state_dict(cfg)
    lengths = [37, 5, 23, 1]
    T = max(lengths)
    B = len(lengths)
    mel = synth.synthetic_mel(B, T, seed=7)
    L = T * 160 // 8
    zs = synth.synthetic_z(B, L, cfg, seed=8)
    out = m.infer(mel.cuda(), sigma=0.6, z=zs, lengths=lengths).cpu()
    _check_expected_launch_shape(m)
    for b, Tb in enumerate(lengths):
        Lb = Tb * 20
        zb = [z[b:b + 1, :, :Lb].contiguous() for z in zs]
        single = m.infer(mel[b:b + 1, :, :Tb].contiguous().cuda(), sigma=0.6, z=zb).cpu()
        assert torch.equal(single[0], out[b, :Tb * 160]), "utterance %d differs from its batch-1 run" % b
        assert torch.count_nonzero(out[b, Tb * 160:]) == 0
        if (b, Tb) not in _RAGGED_ORACLE:                 # (the oracle's answer does not depend on the kernel that ran)
            with torch.no_grad():
                _RAGGED_ORACLE[(b, Tb)] = owg.infer(sd, cfg, mel[b:b + 1, :, :Tb], 0.6, zb)
        assert rms((single - _RAGGED_ORACLE[(b, Tb)]).numpy()) <= RMS_TOL


_RAGGED_ORACLE = {}


def test_ragged_batch_equals_independent_runs_and_oracle(model160):
    _ragged_check(*model160)


def test_two_concurrent_half_batches_change_no_bit(model256):
    """WaveGlow.infer(groups=2) (two interleaved half-batches, the second on a side stream) == groups=1, bit for bit, with
    injected z and with per-utterance seeds; the 16-utterance ragged batch of BASELINE config 3 is one the launch-shape
    heuristic splits, a uniform batch and a batch without host-side lengths are not."""
    m, cfg = model256
    lens = (100 + np.random.Generator(np.random.PCG64(7)).integers(0, 301, size=16)).tolist()
    assert m._tail_loss(lens, 256) >= 0.03 and m._tail_loss([1000] * 8, 256) < 0.03 and m._tail_loss([37, 5, 23, 1], 256) == 0.0
    B, T = len(lens), max(lens)
    mel = synth.synthetic_mel(B, T, seed=21).cuda()
    zs = synth.synthetic_z(B, T * 32, cfg, seed=22)
    one = m.infer(mel, sigma=0.6, z=zs, lengths=lens, groups=1)
    two = m.infer(mel, sigma=0.6, z=zs, lengths=lens, groups=2)
    auto = m.infer(mel, sigma=0.6, z=zs, lengths=lens)
    assert torch.equal(one, two) and torch.equal(one, auto)
    seeds = list(range(500, 500 + B))
    one = m.infer(mel, sigma=0.6, lengths=lens, utterance_seeds=seeds, groups=1)
    two = m.infer(mel, sigma=0.6, lengths=lens, utterance_seeds=seeds, groups=2)
    assert torch.equal(one, two)
    for b in (0, 7, 15):                                  # padding stays silent, the valid part is not
        assert torch.count_nonzero(two[b, lens[b] * 256:]) == 0 and torch.count_nonzero(two[b, :lens[b] * 256]) > 0
    a = m.infer(mel, sigma=0.6, lengths=lens, seed=9)
    assert torch.equal(a, m.infer(mel, sigma=0.6, lengths=lens, seed=9)) and torch.isfinite(a).all()
    with pytest.raises(Exception):
        m.infer(mel, sigma=0.6, lengths=torch.tensor(lens), groups=2)
    # back-to-back calls reuse the two workspaces and the side stream: no cross-call interference
    for _ in range(3):
        assert torch.equal(m.infer(mel, sigma=0.6, lengths=lens, utterance_seeds=seeds), one)


def test_device_noise_is_standard_normal_and_seeded(model160):
    m, cfg = model160
    mel = synth.synthetic_mel(2, 16, seed=3).cuda()
    a = m.infer(mel, sigma=0.6, seed=123)
    b = m.infer(mel, sigma=0.6, seed=123)
    c = m.infer(mel, sigma=0.6, seed=124)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert torch.isfinite(a).all()
    # sigma = 0 -> deterministic, equals injected zeros (this is the Denoiser bias path)
    z0 = [torch.zeros_like(z) for z in synth.synthetic_z(2, 16 * 20, cfg)]
    assert torch.equal(m.infer(mel, sigma=0.0, seed=5), m.infer(mel, sigma=0.0, z=z0))


def test_full_size_properties_hop256_benched_shape(monkeypatch, model256):
    """The shape bench.py times (BASELINE configs[1] at the metric's rate): B = 8, mel 80 x 1000, hop 256 -> 256 000
    group positions per launch, 32 phases, 4 000 tiles.  Properties only (the oracle would need minutes): determinism,
    finiteness, exact output length, noise linearity (sigma = 0 removes every z term: audio = f(mel) alone), batch
    independence (utterance 5 alone gives the bits it has inside the batch), and per-utterance lengths cutting exactly."""
    m, cfg = model256
    B, T = 8, 1000
    mel = synth.synthetic_mel(B, T, seed=1234).cuda()
    a = m.infer(mel, sigma=0.6, seed=5)
    assert a.shape == (B, T * 256) and torch.isfinite(a).all() and float(a.abs().max()) < 1e3
    assert torch.equal(a, m.infer(mel, sigma=0.6, seed=5)) and not torch.equal(a, m.infer(mel, sigma=0.6, seed=6))
    zs = synth.synthetic_z(B, T * 32, cfg, seed=3)
    full = m.infer(mel, sigma=0.6, z=zs)
    assert m.last_launch_shape()[:2] == (64, 4)             # the benched instantiation: k_wn_layer<., 2, false, true, true>
    one = m.infer(mel[5:6].contiguous(), sigma=0.6, z=[z[5:6].contiguous() for z in zs])
    assert torch.equal(one[0], full[5])
    # the same utterance through every OTHER tile width (each anchored to the oracle at this hop by
    # test_hop256_every_tile_width_matches_oracle): all widths accumulate in the same K order, so bit for bit
    for tile in (16, 32, 128):
        monkeypatch.setenv("FACPPG_WN_TILE", str(tile))
        other = m.infer(mel[5:6].contiguous(), sigma=0.6, z=[z[5:6].contiguous() for z in zs])
        assert m.last_launch_shape()[0] == tile
        assert torch.equal(other[0], full[5]), "tile width %d differs from the benched 64-frame kernel" % tile
    monkeypatch.delenv("FACPPG_WN_TILE")
    a0 = m.infer(mel, sigma=0.0, z=zs)
    assert torch.equal(a0, m.infer(mel, sigma=0.0, seed=123))                    # sigma = 0: the noise cannot matter
    lens = [1000, 1, 999, 64, 65, 512, 777, 1000]
    rag = m.infer(mel, sigma=0.6, z=zs, lengths=lens)
    for b, n in enumerate(lens):
        assert torch.equal(rag[b, :n * 256], m.infer(mel[b:b + 1, :, :n].contiguous(), sigma=0.6,
                                                     z=[z[b:b + 1, :, :n * 32].contiguous() for z in zs])[0])
        assert torch.count_nonzero(rag[b, n * 256:]) == 0


def test_full_size_properties(model160):
    """BASELINE config 2 shape (B=8, 80x1000): size-independent properties -- determinism, batch
    independence (item b of the batch == its own batch-1 run, bit-exact), finite output, and a
    200-frame prefix-free spot check against the oracle on one item."""
    from oracle import waveglow as owg
    m, cfg = model160
    B, T = 8, 1000
    mel = synth.synthetic_mel(B, T, seed=1234).cuda()
    zs = [z.cuda() for z in synth.synthetic_z(B, T * 20, cfg, seed=4321)]
    a = m.infer(mel, sigma=0.6, z=zs)
    assert a.shape == (B, T * 160) and torch.isfinite(a).all()
    assert torch.equal(a, m.infer(mel, sigma=0.6, z=zs))
    b = 5
    single = m.infer(mel[b:b + 1].contiguous(), sigma=0.6, z=[z[b:b + 1].contiguous() for z in zs])
    assert torch.equal(single[0], a[b])
    Ts = 200
    sd = synth.waveglow_state_dict(cfg)
    zsub = [z[b:b + 1, :, :Ts * 20].contiguous() for z in zs]
    hip = m.infer(mel[b:b + 1, :, :Ts].contiguous(), sigma=0.6, z=zsub).cpu()
    with torch.no_grad():
        ref = owg.infer(sd, cfg, mel[b:b + 1, :, :Ts].cpu(), 0.6, [z.cpu() for z in zsub])
    e = (hip - ref).numpy()
    print("T=200 rms err", rms(e), "max", np.abs(e).max())
    assert rms(e) <= RMS_TOL


@pytest.mark.parametrize("tag,hop", [("hop160", 160), ("hop256", 256)])
def test_forward_matches_reference_golden_and_inverts(tag, hop, model160, model256):
    """Training direction audio -> z (WaveGlow.forward, glow.py:208-250) vs the reference's golden
    z / sum(log_s) / logdet, the loss value, and the flow-invertibility KAT: infer with the
    forward's z re-injected returns the audio (KAT i)."""
    from waveglow.glow import WaveGlowLoss
    d = golden("waveglow_%s.npz" % tag)
    B, T = int(d["B"]), int(d["T"])
    m, cfg = model160 if hop == 160 else model256
    mel = synth.synthetic_mel(B, T, seed=int(d["mel_seed"])).cuda()
    wav = torch.from_numpy(d["fwd_audio_in"]).cuda()
    with torch.no_grad():
        z, log_s, log_det = m((mel, wav))
    assert z.shape == (B, 8, T * hop // 8) and len(log_s) == 12 and [x.shape[1] for x in log_s] == [4] * 4 + [3] * 4 + [2] * 4
    ez = np.abs(z.cpu().numpy() - d["fwd_z"]).max()
    sums = np.array([float(x.double().sum()) for x in log_s])
    print("forward z max err %.2e, log_s sum err %.2e" % (ez, np.abs(sums - d["fwd_log_s_sum"]).max()))
    assert ez <= 1e-4
    assert np.allclose(sums, d["fwd_log_s_sum"], atol=2e-3)
    assert np.allclose([float(x) for x in log_det], d["fwd_log_det"], atol=1e-3)
    loss = WaveGlowLoss(sigma=0.7071)((z, log_s, log_det))
    from oracle import waveglow as owg
    ref_loss = owg.loss(torch.from_numpy(d["fwd_z"]), [torch.tensor(v) for v in d["fwd_log_s_sum"]],
                        [torch.tensor(v) for v in d["fwd_log_det"]], sigma=0.7071)
    assert abs(float(loss) - float(ref_loss)) <= 1e-4 * max(1.0, abs(float(ref_loss)))
    back = m.infer(mel, sigma=1.0, z=[z[:, 4:8].contiguous(), z[:, 2:4].contiguous(), z[:, 0:2].contiguous()])
    e = (back - wav).cpu().numpy()
    print("round trip rms err", rms(e))
    assert rms(e) <= 1e-4


def test_training_step_loss_and_gradients_match_reference():
    """One WaveGlow training step (train_waveglow.py:121-133): forward, WaveGlowLoss, backward on the
    weight-normed model; loss and every parameter gradient vs the reference's autograd results."""
    import json
    from test_gpu_e2e import weightnorm_state_dict
    from waveglow.glow import WaveGlow, WaveGlowLoss
    d = golden("waveglow_train.npz")
    B, T, hop = int(d["B"]), int(d["T"]), int(d["hop"])
    cfg = dict(synth.WAVEGLOW_CONFIG)
    m = WaveGlow(**cfg)
    m.load_state_dict(weightnorm_state_dict(synth.waveglow_state_dict(cfg)), strict=True)
    m = m.cuda().train()
    mel = synth.synthetic_mel(B, T + 1, seed=int(d["mel_seed"])).cuda()
    wav = torch.from_numpy(d["wav"]).cuda()
    m.zero_grad()
    loss = WaveGlowLoss(0.7071)(m((mel, wav)))
    loss.backward()
    print("loss", float(loss), "ref", float(d["loss"]))
    assert abs(float(loss) - float(d["loss"])) <= 1e-5 * max(1.0, abs(float(d["loss"])))
    names = json.loads(bytes(d["names"]).decode())
    grads = {k: p.grad for k, p in m.named_parameters()}
    assert sorted(grads) == names
    norms = np.array([float(grads[k].double().norm()) for k in names])
    rel = np.abs(norms - d["norms"]) / np.maximum(d["norms"], 1e-6)
    print("grad-norm rel err: max %.2e (%s)" % (rel.max(), names[int(rel.argmax())]))
    assert rel.max() <= 1e-3
    for key in d.files:
        if key.startswith("g:"):
            g = grads[key[2:]].detach().cpu().reshape(-1)
            sub = g[::max(1, -(-g.numel() // 4096))].numpy()
            ref = d[key]
            assert sub.shape == ref.shape, key
            assert np.abs(sub - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()) + 1e-6, key


def test_training_step_fp32_at_config5_shape_matches_reference():
    """The fp32 training path at BASELINE config 5's own shape (batch 3 x segment 10000, waveglow/config.json:8,14) vs the imported
    reference's loss and gradients (tests/golden/waveglow_train_cfg5_B3.npz)."""
    import json
    from helpers import cfg5_batch
    from test_gpu_e2e import weightnorm_state_dict
    from waveglow.glow import WaveGlow, WaveGlowLoss
    d = golden("waveglow_train_cfg5_B3.npz")
    mel, wav = cfg5_batch(3)
    cfg = dict(synth.WAVEGLOW_CONFIG)
    m = WaveGlow(**cfg)
    m.load_state_dict(weightnorm_state_dict(synth.waveglow_state_dict(cfg)), strict=True)
    m = m.cuda().train()
    m.zero_grad()
    loss = WaveGlowLoss(0.7071)(m((mel.cuda(), wav.cuda())))
    loss.backward()
    print("loss", float(loss), "ref", float(d["loss"]))
    assert abs(float(loss) - float(d["loss"])) <= 1e-5 * max(1.0, abs(float(d["loss"])))
    names = json.loads(bytes(d["names"]).decode())
    grads = {k: p.grad for k, p in m.named_parameters()}
    assert sorted(grads) == names
    norms = np.array([float(grads[k].double().norm()) for k in names])
    rel = np.abs(norms - d["norms"]) / np.maximum(d["norms"], 1e-6)
    print("grad-norm rel err: max %.2e (%s)" % (rel.max(), names[int(rel.argmax())]))
    assert rel.max() <= 1e-3
    for key in d.files:
        if key.startswith("g:"):
            g = grads[key[2:]].detach().cpu().reshape(-1)
            sub = g[::max(1, -(-g.numel() // 4096))].numpy()
            assert np.abs(sub - d[key]).max() <= 2e-4 * max(1.0, np.abs(d[key]).max()) + 1e-6, key


def test_legacy_glow_old_layout_and_convert_model():
    """waveglow.glow_old.WaveGlow (stride 256, alternating halves, glow_old.py:121-255) vs the
    reference's golden; convert_model.update_model merges separate res/skip layers."""
    from waveglow import convert_model, glow_old
    d = golden("waveglow_old_hop256.npz")
    cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=256)
    m = glow_old.WaveGlow(**{k: v for k, v in cfg.items() if k != "hop_length"})
    m = glow_old.WaveGlow.remove_weightnorm(m)
    m.load_state_dict(synth.waveglow_state_dict(cfg), strict=True)
    m = m.cuda().eval()
    B, T = int(d["B"]), int(d["T"])
    mel = synth.synthetic_mel(B, T, seed=int(d["mel_seed"])).cuda()
    zs = synth.synthetic_z(B, T * 32, cfg, seed=int(d["z_seed"]))
    audio = m.infer(mel, sigma=float(d["sigma"]), z=zs)
    e = audio.cpu().numpy() - d["audio"]
    print("glow_old rms err", rms(e))
    assert rms(e) <= RMS_TOL and np.abs(e).max() <= 5e-3
    assert m((mel, audio)) is None                       # forward is disabled in the legacy file
    # pre-merge checkpoints: split res_skip into res + skip layers, convert back, same audio
    from waveglow.glow import WaveGlow
    new = WaveGlow(**cfg)
    old = convert_model._clone(new)
    wn = torch.nn.utils.weight_norm
    for w_old in old.WN:
        w_old.res_layers, w_old.skip_layers = torch.nn.ModuleList(), torch.nn.ModuleList()
        for i, rs in enumerate(w_old.res_skip_layers):
            rs = torch.nn.utils.remove_weight_norm(convert_model._clone(rs))
            nc = w_old.n_channels
            sk = torch.nn.Conv1d(nc, nc, 1)
            if i < w_old.n_layers - 1:
                r = torch.nn.Conv1d(nc, nc, 1)
                r.weight.data, r.bias.data = rs.weight.data[:nc].clone(), rs.bias.data[:nc].clone()
                sk.weight.data, sk.bias.data = rs.weight.data[nc:].clone(), rs.bias.data[nc:].clone()
                w_old.res_layers.append(wn(r, name="weight"))
            else:
                sk.weight.data, sk.bias.data = rs.weight.data.clone(), rs.bias.data.clone()
            w_old.skip_layers.append(wn(sk, name="weight"))
        del w_old.res_skip_layers
    conv = convert_model.update_model(old)
    assert not hasattr(conv.WN[0], "res_layers") and convert_model.update_model(new) is new
    a1 = WaveGlow.remove_weightnorm(conv).cuda().eval().infer(mel, sigma=0.6, z=zs)
    a2 = WaveGlow.remove_weightnorm(new).cuda().eval().infer(mel, sigma=0.6, z=zs)
    assert rms((a1 - a2).cpu().numpy()) <= 1e-5


@pytest.mark.parametrize("hop,B,T,lengths", [(256, 5, 420, None), (160, 4, 850, [850, 3, 417, 702])])
def test_flow_end_four_frames_per_thread_same_bits(hop, B, T, lengths, monkeypatch, model160, model256):
    """k_flow_end4 (16-byte row accesses, four frames per thread; picked for launches of >= 65 536 positions) must give
    the bits of the one-position-per-thread kernel: same fmaf chains per position.  Ragged lengths exercise its tail
    stores (frames past an utterance's end must stay untouched: they are the next layer's zero padding)."""
    m, cfg = model160 if hop == 160 else model256
    assert B * T * (hop // 8) >= 65536
    mel = synth.synthetic_mel(B, T, seed=3).cuda()
    a4 = m.infer(mel, sigma=0.6, seed=11, lengths=lengths)
    monkeypatch.setenv("FACPPG_FLOW_END_NO4", "1")
    a1 = m.infer(mel, sigma=0.6, seed=11, lengths=lengths)
    assert torch.equal(a4, a1)
    if lengths:
        one = m.infer(mel[1:2, :, :3].contiguous(), sigma=0.6, z=None, seed=11)          # smoke: a 3-frame utterance alone runs
        assert one.shape == (1, 3 * hop)


@pytest.mark.parametrize("env,shape", [("FACPPG_WG_UNFOLDED=1", "32x4"), ("FACPPG_WN_TILE=32 FACPPG_WN_8W=0", "32x4"), ("FACPPG_WN_TILE=32", "32x8"),
                                       ("FACPPG_WN_NO_XCD_MAP=1", None), ("FACPPG_WN_NO_FLAT=1 FACPPG_WN_TILE=64", "64x4"),
                                       ("FACPPG_WN_TILE=16", "16x8"), ("FACPPG_WN_TILE=64", "64x4"), ("FACPPG_WN_TILE=64 FACPPG_WN_8W=2", "64x8"),
                                       ("FACPPG_WN_TILE=128", "128x8"), ("FACPPG_WG_EDGE_FOLD=0", None),
                                       ("FACPPG_WG_EDGE_FOLD=0 FACPPG_WN_TILE=64", "64x4"), ("FACPPG_WG_EDGE_FOLD=0 FACPPG_WN_TILE=32", "32x8")])
def test_alternate_kernel_paths_match_golden(env, shape, model160, model256, monkeypatch):
    """The A/B switches select other kernels for the same call (the unfolded K=1408 layer the training direction uses;
    every tile width -- 16 / 32 / 64 / 128 frames -- forced against the cost model, on 4 or 8 waves; no XCD-aware phase
    mapping; per-utterance tiles; layers without the folded flow edges).  Each runs the reference-golden comparison at both
    hops and the ragged batch-vs-oracle test (the library reads the switches per call), and asserts
    (facppg_wg_last_launch_shape) that the switch really selected the kernel it names: at the golden sizes the cost model
    alone would pick 16-frame tiles for every one of them."""
    for kv in env.split():
        k, v = kv.split("=")
        monkeypatch.setenv(k, v)
    if shape:
        monkeypatch.setenv("FACPPG_TEST_EXPECT_SHAPE", shape)
    _golden_check(*model160, "hop160", 160)
    _golden_check(*model256, "hop256", 256)
    _ragged_check(*model160)


_HOP256_ORACLE = {}


@pytest.mark.parametrize("tile,waves8", [(64, None), (128, None), (64, "2"), (32, None)])
@pytest.mark.parametrize("lengths", [None, [152, 37, 149]])
def test_hop256_every_tile_width_matches_oracle(tile, waves8, lengths, monkeypatch, model256):
    """The kernel bench.py times (hop 256: 32 phases, kc = 320 folded conditioning rows, 64-frame tiles on 4 waves) and
    its siblings against the CPU ORACLE at hop 256 -- uniform (flat tile cut) and ragged (group table; lengths that
    straddle tile edges) -- with the tile width forced and asserted.  B = 3 x 152 frames: 4 864 positions per utterance,
    so the 255-position receptive field of a flow crosses tile and utterance boundaries many times."""
    from oracle import waveglow as owg
    hop = 256
    m, cfg = model256
    sd = synth.waveglow_state_dict(cfg)
    B, T = 3, 152            # T % 4 == 0: the uniform batch takes the flat tile cut, as the benched shape does
    lens = lengths or [T] * B
    mel = synth.synthetic_mel(B, T, seed=31)
    zs = synth.synthetic_z(B, T * hop // 8, cfg, seed=32)
    monkeypatch.setenv("FACPPG_WN_TILE", str(tile))
    if waves8:
        monkeypatch.setenv("FACPPG_WN_8W", waves8)
    out = m.infer(mel.cuda(), sigma=0.6, z=zs, lengths=lengths).cpu()
    got = m.last_launch_shape()
    assert got[0] == tile and got[1] == (8 if (tile in (32, 128) or waves8) else 4), got
    for b, Tb in enumerate(lens):
        Lb = Tb * hop // 8
        zb = [z[b:b + 1, :, :Lb].contiguous() for z in zs]
        if (b, Tb) not in _HOP256_ORACLE:                # the oracle's answer does not depend on the tile width
            with torch.no_grad():
                _HOP256_ORACLE[(b, Tb)] = owg.infer(sd, cfg, mel[b:b + 1, :, :Tb], 0.6, zb)
        ref = _HOP256_ORACLE[(b, Tb)]
        err = (out[b:b + 1, :Tb * hop] - ref).numpy()
        print("tile %d utt %d (%d frames): rms err %.2e max %.2e (rms ref %.3f)" % (tile, b, Tb, rms(err), np.abs(err).max(), rms(ref.numpy())))
        assert rms(err) <= RMS_TOL and np.abs(err).max() <= 5e-3
        assert torch.count_nonzero(out[b, Tb * hop:]) == 0


@pytest.mark.parametrize("hop,B,T,lengths", [(256, 2, 300, None), (256, 3, 90, [90, 41, 7]), (160, 1, 64, None)])
def test_folded_flow_edges_agree_with_unfolded_layers(hop, B, T, lengths, monkeypatch, model160, model256):
    """The default inference path folds the flow edges into the WN layers (end conv through each layer's skip rows, the
    first layer's taps through the start conv: csrc/facppg_wg.hip k_fold_end_rows / k_fold_first).  It is the same linear
    algebra re-associated, so it must agree with the layer-by-layer form (FACPPG_WG_EDGE_FOLD=0: 512-row res_skip GEMM,
    256-channel skip sum, end conv in k_flow_end) to fp32 round-off -- across tile widths (64 / 32 / 16 frames) and at the
    ragged ends, where the folded start bias must vanish exactly where the reference zero-pads h."""
    m, cfg = model160 if hop == 160 else model256
    mel = synth.synthetic_mel(B, T, seed=5).cuda()
    folded = m.infer(mel, sigma=0.6, seed=3, lengths=lengths)
    monkeypatch.setenv("FACPPG_WG_EDGE_FOLD", "0")
    plain = m.infer(mel, sigma=0.6, seed=3, lengths=lengths)
    err = (folded - plain).cpu().numpy()
    print("rms", rms(err), "max", np.abs(err).max(), "rms audio", rms(plain.cpu().numpy()))
    assert rms(err) <= 2e-5 and np.abs(err).max() <= 5e-4
    assert not torch.equal(folded, plain)   # (the switch did select another path)


@pytest.mark.parametrize("hop,n_flows,n_early_every,n_layers,T", [(256, 3, 2, 1, 40), (160, 2, 4, 3, 70), (256, 5, 4, 2, 9)])
def test_other_stack_depths_match_oracle(hop, n_flows, n_early_every, n_layers, T):
    """The folded flow edges distinguish first / middle / last layers of a stack: stacks of 1 (first = last), 2 (no middle
    layer) and 3 layers, with early outputs at other flows, against the oracle."""
    from oracle import waveglow as owg
    from waveglow.glow import WaveGlow
    cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=hop, n_flows=n_flows, n_early_every=n_early_every,
               WN_config=dict(synth.WAVEGLOW_CONFIG["WN_config"], n_layers=n_layers))
    sd = synth.waveglow_state_dict(cfg)
    m = WaveGlow.remove_weightnorm(WaveGlow(**cfg))
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    B = 2
    mel = synth.synthetic_mel(B, T, seed=21)
    zs = synth.synthetic_z(B, T * hop // 8, cfg, seed=22)
    out = m.infer(mel.cuda(), sigma=0.6, z=zs).cpu()
    with torch.no_grad():
        ref = owg.infer(sd, cfg, mel, 0.6, zs)
    err = (out - ref).numpy()
    print("rms err", rms(err), "rms ref", rms(ref.numpy()))
    assert rms(err) <= RMS_TOL and np.abs(err).max() <= 5e-3


@pytest.mark.parametrize("T", [12, 37, 112, 113, 144, 150, 161, 176, 192, 200, 208, 256])
def test_persistent_launch_equals_layer_launches_bit_for_bit(T, model256, monkeypatch):
    """ONE short utterance can run as one persistent launch (csrc/facppg_wgp.hip: work split over output channels, LDS-DMA
    operand streams, in-launch producer -> consumer hand-offs; picked by a measured rule for 128 < T <= 192 frames,
    FACPPG_WG_PERSIST=1 forces it); everything else is one launch per WaveNet layer.  Both form every sum in the same
    order, so the audio must agree BIT FOR BIT -- with injected noise and with the device noise -- at every column-block
    count the persistent kernel is built for (T = 16 NB boundaries included) and at the metric's T = 200."""
    m, cfg = model256
    hop = 256
    mel = synth.synthetic_mel(1, T, seed=900 + T).cuda()
    zs = synth.synthetic_z(1, T * hop // 8, cfg, seed=901 + T)
    monkeypatch.setenv("FACPPG_WG_PERSIST", "1")
    a = m.infer(mel, sigma=0.6, z=zs)
    tile, waves, wgs = m.last_launch_shape()
    assert waves == 8 and wgs == 256 and tile >= T and tile % 16 == 0, "the persistent launch did not run: %s" % ((tile, waves, wgs),)
    a_seed = m.infer(mel, sigma=0.6, seed=77)
    monkeypatch.setenv("FACPPG_WG_PERSIST", "0")
    b = m.infer(mel, sigma=0.6, z=zs)
    assert m.last_launch_shape()[0] in (16, 32, 64)          # a per-layer tile width
    b_seed = m.infer(mel, sigma=0.6, seed=77)
    assert torch.isfinite(a).all() and a.shape == (1, T * hop)
    d = (a - b).abs().max().item()
    assert torch.equal(a, b), "persistent vs per-layer launches differ: max abs %.3e" % d
    assert torch.equal(a_seed, b_seed)
    monkeypatch.delenv("FACPPG_WG_PERSIST")
    c = m.infer(mel, sigma=0.6, z=zs)                        # the measured rule picks one of the two: same bits either way
    assert torch.equal(a, c)
    assert (m.last_launch_shape()[1:] == (8, 256) and m.last_launch_shape()[0] % 16 == 0 and m.last_launch_shape()[0] > 64) == (128 < T <= 192)




def test_persistent_launch_matches_oracle_hop256(model256, monkeypatch):
    """The persistent launch against the CPU oracle directly (T = 152 frames: 10 column blocks, 4 864 positions), twice in a
    row on one workspace (the second call finds the first one's flags and arrival counter: they are reset per call)."""
    from oracle import waveglow as owg
    m, cfg = model256
    T, hop = 152, 256
    mel = synth.synthetic_mel(1, T, seed=31)
    zs = synth.synthetic_z(1, T * hop // 8, cfg, seed=32)
    monkeypatch.setenv("FACPPG_WG_PERSIST", "1")
    got = m.infer(mel.cuda(), sigma=0.6, z=zs).cpu()
    again = m.infer(mel.cuda(), sigma=0.6, z=zs).cpu()
    assert m.last_launch_shape() == (160, 8, 256)
    assert torch.equal(got, again)
    with torch.no_grad():
        ref = owg.infer(synth.waveglow_state_dict(cfg), cfg, mel, 0.6, zs)
    e = (got - ref).numpy()
    print("persistent launch vs oracle: rms err %.2e (rms ref %.3f)" % (rms(e), rms(ref.numpy())))
    assert rms(e) <= RMS_TOL
