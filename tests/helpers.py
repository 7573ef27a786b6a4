"""Shared helpers for the parity tests: golden loading and seeded inputs that match
tests/golden/make_golden.py exactly."""
import os

import numpy as np
import torch

from conftest import GOLDEN


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def masks_from_seed(seed, shape):
    g = np.random.Generator(np.random.PCG64(seed))
    return (g.random(shape) < 0.5).astype(np.uint8)


def rms(a):
    a = np.asarray(a, dtype=np.float64)
    return float(np.sqrt(np.mean(a * a)))


def tacotron_case(tag):
    """Rebuild the inputs of a tests/golden/tacotron_<tag>.npz fixture."""
    from common.hparams import create_hparams_stage
    from facppg import synth
    d = golden("tacotron_%s.npz" % tag)
    ns, Tin, ms = int(d["n_symbols"]), int(d["Tin"]), int(d["max_steps"])
    hp = create_hparams_stage(max_decoder_steps=ms, n_symbols=ns)
    sd = synth.tacotron_state_dict(hp, seed=16807, gate_bias=float(d["gate_bias"]))
    ppg = synth.synthetic_ppg(Tin, ns, seed=int(d["ppg_seed"]), alpha=0.002 if ns > 100 else 0.1)
    enc_masks = masks_from_seed(int(d["enc_mask_seed"]), (2, 1, Tin, hp.symbols_embedding_dim))
    dec_masks = masks_from_seed(int(d["dec_mask_seed"]), (ms, 2, 1, hp.prenet_dim))
    return d, hp, sd, ppg, enc_masks, dec_masks


def cfg5_batch(B):
    """The seeded training batch of tests/golden/waveglow_train_cfg5_B<B>.npz (make_golden.py::cfg5_batch): audio
    [B, 10000] ~ N(0, 0.1^2) clipped to +-1 from PCG64(500 + B), mel [B, 80, 63] from synthetic_mel(seed 700 + B)."""
    from facppg import synth
    g = np.random.Generator(np.random.PCG64(500 + B))
    wav = torch.from_numpy(np.clip(g.standard_normal((B, 10000), dtype=np.float32) * 0.1, -1, 1))
    return synth.synthetic_mel(B, 10000 // 160 + 1, seed=700 + B), wav


def sha16(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
