"""CPU: pin the oracle against the golden vectors captured from the reference itself
(tests/golden/make_golden.py).  These are the vectors SURVEY.md section 8c calls for; the
reference's own tests hold none for the hot path."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from helpers import golden, tacotron_case
from facppg import synth
from oracle import dsp, tacotron as otac, waveglow as owg


@pytest.mark.parametrize("tag,hop", [("hop160", 160), ("hop256", 256)])
def test_waveglow_infer_and_forward(tag, hop):
    d = golden("waveglow_%s.npz" % tag)
    B, T = int(d["B"]), int(d["T"])
    cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=hop)
    sd = synth.waveglow_state_dict(cfg)
    mel = synth.synthetic_mel(B, T, seed=int(d["mel_seed"]))
    zs = synth.synthetic_z(B, T * hop // 8, cfg, seed=int(d["z_seed"]))
    with torch.no_grad():
        audio = owg.infer(sd, cfg, mel, float(d["sigma"]), zs)
        z, log_s, log_det = owg.forward(sd, cfg, mel, torch.from_numpy(d["fwd_audio_in"]))
    assert audio.shape == (B, T * hop)                      # integer length: T*hop exactly
    assert np.abs(audio.numpy() - d["audio"]).max() < 5e-5  # fp32 reassociation only
    assert np.abs(z.numpy() - d["fwd_z"]).max() < 1e-5
    assert np.allclose([float(x.double().sum()) for x in log_s], d["fwd_log_s_sum"], atol=1e-3)
    # round trip (SURVEY section 4 KAT i): forward's z re-injected into infer gives the audio back
    zf = z
    zs_rt = [zf[:, 4:8], zf[:, 2:4], zf[:, 0:2]]
    with torch.no_grad():
        back = owg.infer(sd, cfg, mel, 1.0, zs_rt)
    assert np.abs(back.numpy() - d["fwd_audio_in"]).max() < 2e-5


def test_waveglow_legacy_alternating_layout():
    d = golden("waveglow_old_hop256.npz")
    cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=256)
    sd = synth.waveglow_state_dict(cfg)
    B, T = int(d["B"]), int(d["T"])
    mel = synth.synthetic_mel(B, T, seed=int(d["mel_seed"]))
    zs = synth.synthetic_z(B, T * 32, cfg, seed=int(d["z_seed"]))
    with torch.no_grad():
        a = owg.infer(sd, cfg, mel, float(d["sigma"]), zs, alternate=True)
        plain = owg.infer(sd, cfg, mel, float(d["sigma"]), zs)
    assert np.abs(a.numpy() - d["audio"]).max() < 5e-5
    assert np.abs(plain.numpy() - d["audio"]).max() > 1e-2       # the layouts really differ


def test_stft_mel_denoiser():
    s = golden("stft.npz")
    y = torch.from_numpy(s["y"])
    for hop in (160, 256):
        st = dsp.StftOracle(1024, hop, 1024)
        mag, ph = st.transform(y)
        assert mag.shape[2] == y.shape[1] // hop + 1        # frames = N//hop + 1
        assert np.abs(mag.numpy() - s["mag_%d" % hop]).max() < 1e-5
        rec = st.inverse(mag, ph)
        assert rec.shape[2] == hop * (mag.shape[2] - 1)
        assert np.abs(rec.numpy() - s["rec_%d" % hop]).max() < 1e-5
    ts = dsp.TacotronStftOracle(1024, 160, 1024, 80, 16000, 0.0, 8000.0)
    assert np.abs(ts.mel_basis.numpy() - s["mel_basis_16k"]).max() == 0
    assert np.abs(ts.mel_spectrogram(y).numpy() - s["mel_16k"]).max() < 1e-5
    assert np.abs(dsp.TacotronStftOracle().mel_spectrogram(y).numpy() - s["mel_22k"]).max() < 1e-5
    # librosa documentation example: librosa.filters.mel(22050, 2048)[0, 1] == 0.016 (3 decimals)
    assert abs(dsp.mel_filterbank(22050, 2048)[0, 1] - 0.016) < 5e-4
    d = golden("denoiser_hop160.npz")
    cfg = dict(synth.WAVEGLOW_CONFIG)
    sd = synth.waveglow_state_dict(cfg)
    L = 88 * 160 // 8
    with torch.no_grad():
        bias = owg.infer(sd, cfg, torch.zeros(1, 80, 88), 0.0,
                         [torch.zeros(1, 4, L), torch.zeros(1, 2, L), torch.zeros(1, 2, L)])
    den = dsp.DenoiserOracle(bias)
    assert np.abs(den.bias_spec.numpy() - d["bias_spec"]).max() < 1e-4
    x = torch.from_numpy(d["audio_in"])
    assert np.abs(den(x, 0.005).numpy() - d["out_0005"]).max() < 1e-5
    assert np.abs(den(x, 1.0).numpy() - d["out_1"]).max() < 1e-5


def test_denoiser_hop256():
    """The denoiser at the metric's rate (hop 256 / 22.05 kHz): DenoiserOracle(hop_length=256) vs the reference's
    Denoiser(hop_length=256) on the hop-256 model (tests/golden/make_golden.py gen_denoiser_hop256)."""
    d = golden("denoiser_hop256.npz")
    hop = int(d["hop"])
    cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=hop)
    sd = synth.waveglow_state_dict(cfg)
    L = 88 * hop // 8
    with torch.no_grad():
        bias = owg.infer(sd, cfg, torch.zeros(1, 80, 88), 0.0,
                         [torch.zeros(1, 4, L), torch.zeros(1, 2, L), torch.zeros(1, 2, L)])
    den = dsp.DenoiserOracle(bias, hop_length=hop)
    assert np.abs(den.bias_spec.numpy() - d["bias_spec"]).max() < 1e-4
    x = torch.from_numpy(d["audio_in"])
    assert x.shape == (int(d["B"]), int(d["T"]) * hop)
    assert np.abs(den(x, 0.005).numpy() - d["out_0005"]).max() < 1e-5
    assert np.abs(den(x, 1.0).numpy() - d["out_1"]).max() < 1e-5


def test_attention_window_mask_bit_exact():
    d = golden("attn_masks.npz")
    cases = json.loads(bytes(d["cases"]).decode())
    assert len(cases) > 100
    for c in cases:
        m = otac.window_mask(c["lengths"], c["W"], c["t"]).numpy().astype(np.uint8)
        assert np.array_equal(m, d[c["key"]]), c
        # the product's host-side function (lists / CPU tensors; GPU lengths go through the HIP kernel, -m gpu)
        from common.utils import get_mask_from_lengths_window_and_time_step as product_mask
        pm = product_mask(torch.tensor(c["lengths"]), c["W"], c["t"])
        assert pm.dtype == torch.bool and np.array_equal(pm.numpy().astype(np.uint8), d[c["key"]]), c


@pytest.mark.parametrize("tag", ["nostop", "stop", "mono40"])
def test_tacotron_inference(tag):
    d, hp, sd, ppg, em, dm = tacotron_case(tag)
    x = torch.from_numpy(ppg).t().unsqueeze(0)
    mel, mel_post, gate, align = otac.inference(
        sd, hp, x, torch.from_numpy(em.astype(np.float32)), torch.from_numpy(dm.astype(np.float32)))
    assert mel.shape == d["mel"].shape                      # Tout incl. the stopping frame
    assert np.abs(mel.numpy() - d["mel"]).max() < 1e-5
    assert np.abs(mel_post.numpy() - d["mel_post"]).max() < 1e-5
    assert np.abs(gate.numpy() - d["gate"]).max() < 1e-5
    assert np.abs(align.numpy() - d["align"]).max() < 1e-6


def metric_case():
    """Inputs of tests/golden/e2e_metric.npz (make_golden.py gen_e2e_metric): the metric's own case at the shape bench.py times."""
    from common.hparams import create_hparams_stage
    from helpers import masks_from_seed
    d = golden("e2e_metric.npz")
    Tin, hop, ns = int(d["Tin"]), int(d["hop"]), int(d["n_symbols"])
    hp = create_hparams_stage(max_decoder_steps=Tin, n_symbols=ns)
    tsd = synth.tacotron_state_dict(hp, seed=16807, gate_bias=float(d["gate_bias"]))
    ppg = synth.synthetic_ppg(Tin, ns, seed=int(d["ppg_seed"]), alpha=float(d["ppg_alpha"]))
    em = masks_from_seed(int(d["enc_mask_seed"]), (2, 1, Tin, hp.symbols_embedding_dim))
    dm = masks_from_seed(int(d["dec_mask_seed"]), (Tin, 2, 1, hp.prenet_dim))
    cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=hop)
    zs = synth.synthetic_z(1, Tin * hop // 8, cfg, seed=int(d["z_seed"]))
    return d, hp, tsd, ppg, em, dm, cfg, zs


def test_end_to_end_at_the_metric_shape():
    """The oracle of the WHOLE path at the headline's shape -- PPG [200 x 5816] -> Tacotron2 (200 steps) -> WaveGlow at hop 256
    -> Denoiser(hop_length=256) -- against the imported reference's output for exactly that run (e2e_metric.npz;
    generate_synthesis.py:74-98)."""
    d, hp, tsd, ppg, em, dm, cfg, zs = metric_case()
    hop, Tout = int(d["hop"]), int(d["Tout"])
    x = torch.from_numpy(ppg).t().unsqueeze(0)
    with torch.no_grad():
        _, mel_post, _, _ = otac.inference(tsd, hp, x, torch.from_numpy(em.astype(np.float32)), torch.from_numpy(dm.astype(np.float32)))
        assert mel_post.shape == d["mel_post"].shape == (1, 80, Tout)
        assert np.abs(mel_post.numpy() - d["mel_post"]).max() < 1e-5
        wsd = synth.waveglow_state_dict(cfg)
        audio = owg.infer(wsd, cfg, mel_post, float(d["sigma"]), zs)
        assert audio.shape == d["audio"].shape == (1, Tout * hop)                   # N = Tout * hop exactly
        assert np.sqrt(np.mean((audio.numpy().astype(np.float64) - d["audio"]) ** 2)) < 1e-5
        nb = 88 * hop // 8
        bias = owg.infer(wsd, cfg, torch.zeros(1, 80, 88), 0.0, [torch.zeros(1, 4, nb), torch.zeros(1, 2, nb), torch.zeros(1, 2, nb)])
        out = dsp.DenoiserOracle(bias, hop_length=hop)(audio, float(d["strength"]))[:, 0]
    assert np.sqrt(np.mean((out.numpy().astype(np.float64) - d["audio_denoised"]) ** 2)) < 1e-5


@pytest.mark.parametrize("B", [3, 12])
def test_training_step_at_config5_shape(B):
    """BASELINE config 5 at its own shape (waveglow/config.json:8,14: batch 3 x segment 10000, hop 160; and batch 12, the
    second shape bench.py times): the oracle's forward + WaveGlowLoss (glow.py:208-250, 43-59) with torch autograd through
    the weight-norm parametrisation w = g v / |v| against the imported reference's loss and its 938 gradient norms
    (tests/golden/waveglow_train_cfg5_B*.npz).  The batch regenerates from its seeds (checked by hash)."""
    from helpers import cfg5_batch, sha16
    d = golden("waveglow_train_cfg5_B%d.npz" % B)
    assert int(d["B"]) == B and int(d["segment_length"]) == 10000 and int(d["hop"]) == 160
    mel, wav = cfg5_batch(B)
    assert sha16(wav.numpy()) == bytes(d["wav_sha"]).decode()
    cfg = dict(synth.WAVEGLOW_CONFIG)
    sd = synth.waveglow_state_dict(cfg)
    leaves, eff = {}, {}
    for k, v in sd.items():
        if k.startswith("WN.") and k.endswith(".weight") and ".end." not in k:
            wv = v.clone().requires_grad_(True)
            wg = v.flatten(1).norm(dim=1).view(-1, 1, 1).clone().requires_grad_(True)
            leaves[k[:-6] + "weight_v"], leaves[k[:-6] + "weight_g"] = wv, wg
            eff[k] = wg * wv / wv.flatten(1).norm(dim=1).view(-1, 1, 1)      # torch.nn.utils.weight_norm, dim 0
        else:
            leaves[k] = eff[k] = v.clone().requires_grad_(True)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    z, log_s, log_det = owg.forward(eff, cfg, mel, wav)
    loss = owg.loss(z, log_s, log_det, 0.7071)
    loss.backward()
    assert abs(float(loss) - float(d["loss"])) <= 1e-5 * max(1.0, abs(float(d["loss"])))
    names = json.loads(bytes(d["names"]).decode())
    assert sorted(leaves) == names
    norms = np.array([float(leaves[k].grad.double().norm()) for k in names])
    rel = np.abs(norms - d["norms"]) / np.maximum(d["norms"], 1e-6)
    print("config 5, B = %d: loss %.6f (ref %.6f); grad-norm rel err max %.2e (%s)" % (B, float(loss), float(d["loss"]), rel.max(),
                                                                                      names[int(rel.argmax())]))
    assert rel.max() <= 1e-3
    for key in d.files:
        if key.startswith("g:"):
            g = leaves[key[2:]].grad.reshape(-1)
            sub = g[::max(1, -(-g.numel() // 4096))].numpy()
            assert np.abs(sub - d[key]).max() <= 1e-4 * max(1.0, np.abs(d[key]).max()) + 1e-6, key


def test_hparams_surface():
    from common import hparams
    with open(os.path.join(GOLDEN, "hparams.json")) as f:
        ref = json.load(f)
    assert vars(hparams.create_hparams()) == ref["create_hparams"]
    assert vars(hparams.create_hparams_stage()) == ref["create_hparams_stage"]
    assert hparams.create_hparams_stage(n_symbols=40).n_symbols == 40
    with pytest.raises(ValueError, match="The hyper-parameter bogus is not supported."):
        hparams.create_hparams_stage(bogus=1)
    with pytest.raises(ValueError):
        hparams.create_hparams(is_large_set=True)           # stage-only key


def test_folded_flow_edges_are_the_same_linear_algebra():
    """The identities the inference kernels rely on (csrc/facppg_wg.hip, k_fold_end_rows / k_fold_first), checked on the
    oracle in fp64: (i) end(sum_i (Ws_i a_i + bs_i)) = sum_i (W_end Ws_i) a_i + (W_end sum_i bs_i + b_end)  (glow.py:167-175);
    (ii) the first layer's dilated conv over start(x) = W_start x + b_start equals a 3-tap conv over [x; 1_inside] with
    weights [W_in[tap] W_start | W_in[tap] b_start] -- including the two edge positions, where the reference zero-pads h,
    so b_start must NOT arrive through the tap that falls outside (glow.py:156-160)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(3)
    nc, nh, nl, L = 16, 3, 4, 23
    r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    # (i)
    W_end, b_end = r(2 * nh, nc), r(2 * nh)
    acts = [r(1, nc, L) for _ in range(nl)]
    Ws, bs = [r(nc, nc) for _ in range(nl)], [r(nc) for _ in range(nl)]
    skip_sum = sum(F.conv1d(a, W[:, :, None], b) for a, W, b in zip(acts, Ws, bs))
    ref = F.conv1d(skip_sum, W_end[:, :, None], b_end)
    folded = sum(F.conv1d(a, (W_end @ W)[:, :, None]) for a, W in zip(acts, Ws)) + (W_end @ sum(bs) + b_end)[None, :, None]
    assert torch.allclose(ref, folded, rtol=1e-12, atol=1e-12)
    # (ii)
    W_start, b_start, W_in, b_in = r(nc, nh), r(nc), r(2 * nc, nc, 3), r(2 * nc)
    x = r(1, nh, L)
    h = F.conv1d(x, W_start[:, :, None], b_start)
    ref = F.conv1d(h, W_in, b_in, dilation=1, padding=1)
    xa = torch.cat([x, torch.ones(1, 1, L, dtype=torch.float64)], 1)                  # indicator channel: 1 inside the utterance
    W_fold = torch.cat([torch.einsum("oct,cj->ojt", W_in, W_start), torch.einsum("oct,c->ot", W_in, b_start)[:, None, :]], 1)
    folded = F.conv1d(xa, W_fold, b_in, dilation=1, padding=1)                        # zero padding of xa = zero padding of h
    assert torch.allclose(ref, folded, rtol=1e-12, atol=1e-12)
    naive = F.conv1d(x, W_fold[:, :nh], b_in + torch.einsum("oct,c->o", W_in, b_start), dilation=1, padding=1)
    assert not torch.allclose(ref[..., 0], naive[..., 0]) and torch.allclose(ref[..., 1:-1], naive[..., 1:-1])


def test_mel_filterbank_matches_an_independent_slaney_implementation():
    """a11 / f2: the reference takes its mel basis from librosa 0.6.2 (layers.py:82-83), which this image lacks; oracle/dsp.py
    restates it.  Second, independent derivation: the Slaney filterbank of Hugging Face transformers (audio_utils.mel_filter_bank,
    norm = mel_scale = "slaney"; shares no code with the oracle; fixture written by tests/golden/make_mel_basis_hf.py).  The
    two agree to 1e-12 at the reference config, at the 22.05 kHz metric config and on librosa's documentation example, whose
    printed values (0., 0.016, 0.032 in row 0) both reproduce -- the same non-zero pattern included.  The HIP mel analysis
    is held to this basis by tests/test_gpu_dsp.py."""
    from oracle import dsp
    d = golden("mel_basis_hf.npz")
    for tag in ("ref16k", "metric22k", "librosa_doc"):
        sr, n_fft, n_mels, fmin, fmax = d[tag + "_args"]
        mine = dsp.mel_filterbank(int(sr), int(n_fft), int(n_mels), float(fmin), float(fmax))
        other = np.zeros(mine.size)
        other[d[tag + "_idx"]] = d[tag + "_val"]
        other = other.reshape(mine.shape)
        assert np.array_equal(mine != 0, other != 0), tag
        assert np.abs(mine - other).max() <= 1e-12, tag
    doc = dsp.mel_filterbank(22050, 2048)
    assert np.allclose(doc[0, :3], [0.0, 0.016, 0.032], atol=5e-4)        # librosa.filters.mel docstring
    try:                                                                  # where transformers is importable, ask it directly too
        from transformers.audio_utils import mel_filter_bank
    except Exception:
        return
    live = mel_filter_bank(513, 80, 0.0, 8000.0, 16000, norm="slaney", mel_scale="slaney").T
    assert np.abs(live - dsp.mel_filterbank(16000, 1024, 80, 0.0, 8000.0)).max() <= 1e-12
