"""Deterministic synthetic weights and inputs (no checkpoints ship with the reference).

The reference publishes no pretrained weights (README.md:71,80-82), so every
fixture, test and bench run uses weights drawn here from NumPy PCG64 streams
keyed by (seed, tensor name).  The same call gives bit-identical tensors in the
dev container and on the GPU box, so the 350 MB of WaveGlow weights never need
to be committed.

Key layouts follow the reference's state dicts:
  * WaveGlow after ``remove_weightnorm`` (src/waveglow/glow.py:111-152,179-206)
  * Tacotron2 (src/common/model.py ctor lines)
"""
import zlib

import numpy as np
import torch

WAVEGLOW_CONFIG = {
    # src/waveglow/config.json:29-41
    "n_mel_channels": 80,
    "hop_length": 160,
    "n_flows": 12,
    "n_group": 8,
    "n_early_every": 4,
    "n_early_size": 2,
    "WN_config": {"n_layers": 8, "n_channels": 256, "kernel_size": 3},
}


def _rng(seed, name):
    return np.random.Generator(np.random.PCG64([int(seed), zlib.crc32(name.encode())]))


def _normal(seed, name, shape, std):
    g = _rng(seed, name)
    return torch.from_numpy((g.standard_normal(shape, dtype=np.float32) * np.float32(std)))


def flow_channels(cfg):
    """Per-flow (n_remaining_channels, n_half) as built by WaveGlow.__init__
    (src/waveglow/glow.py:195-206)."""
    n_half = cfg["n_group"] // 2
    n_rem = cfg["n_group"]
    out = []
    for k in range(cfg["n_flows"]):
        if k % cfg["n_early_every"] == 0 and k > 0:
            n_half -= cfg["n_early_size"] // 2
            n_rem -= cfg["n_early_size"]
        out.append((n_rem, n_half))
    return out


def waveglow_state_dict(cfg=None, seed=16807):
    """Synthetic WaveGlow weights in the post-remove_weightnorm key layout."""
    cfg = cfg or WAVEGLOW_CONFIG
    nm, ng = cfg["n_mel_channels"], cfg["n_group"]
    nc = cfg["WN_config"]["n_channels"]
    nl = cfg["WN_config"]["n_layers"]
    ks = cfg["WN_config"]["kernel_size"]
    hop = cfg["hop_length"]
    sd = {}
    # ~1024/hop taps x nm channels contribute to each upsampled sample; the gain keeps the
    # conditioning O(1) for log-mel inputs of magnitude ~5 so the gates are not saturated.
    up_std = 0.2 / np.sqrt(nm * 1024.0 / hop)
    sd["upsample.weight"] = _normal(seed, "upsample.weight", (nm, nm, 1024), up_std)
    sd["upsample.bias"] = _normal(seed, "upsample.bias", (nm,), 0.01)
    for k, (n_rem, n_half) in enumerate(flow_channels(cfg)):
        p = "WN.%d." % k
        sd[p + "start.weight"] = _normal(seed, p + "start.weight", (nc, n_half, 1), 1.0 / np.sqrt(n_half))
        sd[p + "start.bias"] = _normal(seed, p + "start.bias", (nc,), 0.01)
        for i in range(nl):
            q = p + "in_layers.%d." % i
            sd[q + "weight"] = _normal(seed, q + "weight", (2 * nc, nc, ks), 1.0 / np.sqrt(nc * ks))
            sd[q + "bias"] = _normal(seed, q + "bias", (2 * nc,), 0.01)
            q = p + "cond_layers.%d." % i
            sd[q + "weight"] = _normal(seed, q + "weight", (2 * nc, nm * ng, 1), 1.0 / np.sqrt(nm * ng))
            sd[q + "bias"] = _normal(seed, q + "bias", (2 * nc,), 0.01)
            q = p + "res_skip_layers.%d." % i
            rs = 2 * nc if i < nl - 1 else nc
            sd[q + "weight"] = _normal(seed, q + "weight", (rs, nc, 1), 1.0 / np.sqrt(nc))
            sd[q + "bias"] = _normal(seed, q + "bias", (rs,), 0.01)
        # default init is zero (glow.py:129-130); non-trivial coupling for parity
        sd[p + "end.weight"] = _normal(seed, p + "end.weight", (2 * n_half, nc, 1), 0.01)
        sd[p + "end.bias"] = _normal(seed, p + "end.bias", (2 * n_half,), 0.02)
        # orthonormal with det > 0 (glow.py:74-80)
        g = _rng(seed, "convinv.%d" % k)
        q_, _ = np.linalg.qr(g.standard_normal((n_rem, n_rem)))
        if np.linalg.det(q_) < 0:
            q_[:, 0] = -q_[:, 0]
        sd["convinv.%d.conv.weight" % k] = torch.from_numpy(q_.astype(np.float32)).reshape(n_rem, n_rem, 1).contiguous()
    return sd


def _bn(sd, seed, prefix, n):
    sd[prefix + "weight"] = 1.0 + _normal(seed, prefix + "weight", (n,), 0.05)
    sd[prefix + "bias"] = _normal(seed, prefix + "bias", (n,), 0.05)
    sd[prefix + "running_mean"] = _normal(seed, prefix + "running_mean", (n,), 0.05)
    g = _rng(seed, prefix + "running_var")
    sd[prefix + "running_var"] = torch.from_numpy((1.0 + 0.1 * g.random(n, dtype=np.float32)).astype(np.float32))
    sd[prefix + "num_batches_tracked"] = torch.tensor(0, dtype=torch.long)


def tacotron_state_dict(hp, seed=16807, gate_bias=-10.0):
    """Synthetic Tacotron2 weights (keys: SURVEY.md Appendix B).  ``gate_bias=-10`` keeps
    the stop gate closed so Tout == max_decoder_steps deterministically."""
    sd = {}
    E = hp.encoder_embedding_dim
    S = hp.symbols_embedding_dim

    def lin(name, out_d, in_d, bias=False, gain=1.0):
        sd[name + ".linear_layer.weight"] = _normal(seed, name + ".w", (out_d, in_d), gain / np.sqrt(in_d))
        if bias:
            sd[name + ".linear_layer.bias"] = _normal(seed, name + ".b", (out_d,), 0.01)

    # PPG rows are sparse posteriors (sum 1), so the first prenet layer sees tiny inputs:
    # use a large gain to keep the embedding O(1).
    lin("encoder.prenet.layers.0", S, hp.n_symbols, gain=3.0 * float(np.sqrt(hp.n_symbols)))
    lin("encoder.prenet.layers.1", S, S, gain=2.0)
    for j in range(hp.encoder_n_convolutions):
        p = "encoder.convolutions.%d." % j
        K = hp.encoder_kernel_size
        sd[p + "0.conv.weight"] = _normal(seed, p + "w", (E, E, K), 1.5 / np.sqrt(E * K))
        sd[p + "0.conv.bias"] = _normal(seed, p + "b", (E,), 0.01)
        _bn(sd, seed, p + "1.", E)
    H = E // 2
    for sfx in ("", "_reverse"):
        sd["encoder.lstm.weight_ih_l0" + sfx] = _normal(seed, "enc.lstm.wih" + sfx, (4 * H, E), 1.5 / np.sqrt(E))
        sd["encoder.lstm.weight_hh_l0" + sfx] = _normal(seed, "enc.lstm.whh" + sfx, (4 * H, H), 0.8 / np.sqrt(H))
        sd["encoder.lstm.bias_ih_l0" + sfx] = _normal(seed, "enc.lstm.bih" + sfx, (4 * H,), 0.01)
        sd["encoder.lstm.bias_hh_l0" + sfx] = _normal(seed, "enc.lstm.bhh" + sfx, (4 * H,), 0.01)
    A, D, P = hp.attention_rnn_dim, hp.decoder_rnn_dim, hp.prenet_dim
    nf = hp.n_acoustic_feat_dims
    lin("decoder.prenet.layers.0", P, nf, gain=2.0)
    lin("decoder.prenet.layers.1", P, P, gain=2.0)
    for nm_, idim, hdim in (("attention_rnn", P + E, A), ("decoder_rnn", A + E, D)):
        sd["decoder.%s.weight_ih" % nm_] = _normal(seed, nm_ + ".wih", (4 * hdim, idim), 1.2 / np.sqrt(idim))
        sd["decoder.%s.weight_hh" % nm_] = _normal(seed, nm_ + ".whh", (4 * hdim, hdim), 0.7 / np.sqrt(hdim))
        sd["decoder.%s.bias_ih" % nm_] = _normal(seed, nm_ + ".bih", (4 * hdim,), 0.01)
        sd["decoder.%s.bias_hh" % nm_] = _normal(seed, nm_ + ".bhh", (4 * hdim,), 0.01)
    ad = hp.attention_dim
    lin("decoder.attention_layer.query_layer", ad, A, gain=2.0)
    lin("decoder.attention_layer.memory_layer", ad, E, gain=3.0)
    lin("decoder.attention_layer.v", 1, ad, gain=3.0)
    nfil, ksz = hp.attention_location_n_filters, hp.attention_location_kernel_size
    sd["decoder.attention_layer.location_layer.location_conv.conv.weight"] = _normal(
        seed, "loc.conv", (nfil, 2, ksz), 1.0 / np.sqrt(2 * ksz))
    lin("decoder.attention_layer.location_layer.location_dense", ad, nfil, gain=3.0)
    lin("decoder.linear_projection", nf, D + E, bias=True, gain=2.0)
    lin("decoder.gate_layer", 1, D + E, bias=True)
    sd["decoder.gate_layer.linear_layer.bias"] = torch.tensor([gate_bias], dtype=torch.float32)
    pe, pk, pn = hp.postnet_embedding_dim, hp.postnet_kernel_size, hp.postnet_n_convolutions
    dims = [nf] + [pe] * (pn - 1) + [nf]
    for j in range(pn):
        p = "postnet.convolutions.%d." % j
        sd[p + "0.conv.weight"] = _normal(seed, p + "w", (dims[j + 1], dims[j], pk), 1.0 / np.sqrt(dims[j] * pk))
        sd[p + "0.conv.bias"] = _normal(seed, p + "b", (dims[j + 1],), 0.01)
        _bn(sd, seed, p + "1.", dims[j + 1])
    return sd


def synthetic_mel(B, T, n_mel=80, seed=1234):
    """Log-mel-like input: clip(N(-5, 2^2), log(1e-5), 2)."""
    g = np.random.Generator(np.random.PCG64(seed))
    x = g.standard_normal((B, n_mel, T), dtype=np.float32) * 2.0 - 5.0
    return torch.from_numpy(np.clip(x, np.float32(np.log(1e-5)), 2.0).astype(np.float32))


def synthetic_ppg(Tin, D=5816, seed=0, alpha=0.002):
    """Sparse posteriors: rows ~ Dirichlet(alpha), sum to 1 (test/test_ppg.py:48-54)."""
    g = np.random.Generator(np.random.PCG64(seed))
    x = g.gamma(alpha, 1.0, size=(Tin, D)).astype(np.float64) + 1e-30
    x /= x.sum(axis=1, keepdims=True)
    return x.astype(np.float32)


def synthetic_z(B, L, cfg=None, seed=4321):
    """The three N(0,1) draws of WaveGlow.infer in call order (glow.py:261-270,285-290):
    [B, n_remaining, L], then [B, n_early_size, L] at each early-output flow."""
    cfg = cfg or WAVEGLOW_CONFIG
    g = np.random.Generator(np.random.PCG64(seed))
    n_rem = flow_channels(cfg)[-1][0]
    n_early = sum(1 for k in range(cfg["n_flows"]) if k % cfg["n_early_every"] == 0 and k > 0)
    shapes = [(B, n_rem, L)] + [(B, cfg["n_early_size"], L)] * n_early
    return [torch.from_numpy(g.standard_normal(s, dtype=np.float32)) for s in shapes]
