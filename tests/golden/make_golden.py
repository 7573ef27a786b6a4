#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the dev container (needs /root/reference); the GPU box never sees the
reference -- only the small .npz/.json files this script writes.  Harness shims follow
SURVEY.md Appendix A: the reference hard-wires CUDA types and imports kaldi/textgrid/
librosa, none of which exist here, so we
  * register a synthetic ``common`` package pointing at src/common (skips __init__.py),
  * stub ``librosa`` with the oracle's restatement (PARITY UNPINNED at that boundary),
  * alias torch.cuda.{Float,Long,Half}Tensor to CPU types, ByteTensor to a bool ctor,
  * make ``.cuda()`` the identity.
Stochastic ops are made reproducible by INJECTION: ``torch.Tensor.normal_`` is replaced
while WaveGlow.infer runs (z comes from facppg.synth.synthetic_z) and the reference's
``F.dropout`` (prenet, model.py:134) multiplies by masks drawn from a seeded NumPy stream.

Usage:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/src"
sys.path.insert(0, os.path.join(ROOT, "fac-via-ppg_amd"))
sys.path.insert(0, ROOT)

from facppg import synth  # noqa: E402
from oracle import dsp as odsp  # noqa: E402


# ------------------------------------------------------------------ shims
def install_shims():
    sys.path.insert(0, REF)
    for name in [m for m in sys.modules if m == "common" or m.startswith("common.") or
                 m == "waveglow" or m.startswith("waveglow.")]:
        del sys.modules[name]
    pkg = types.ModuleType("common")
    pkg.__path__ = [os.path.join(REF, "common")]
    sys.modules["common"] = pkg
    wg = types.ModuleType("waveglow")
    wg.__path__ = [os.path.join(REF, "waveglow")]
    sys.modules["waveglow"] = wg
    lib = types.ModuleType("librosa")
    lf = types.ModuleType("librosa.filters")
    lu = types.ModuleType("librosa.util")
    lf.mel = lambda sr, n_fft, n_mels=128, fmin=0.0, fmax=None, htk=False, norm=1: \
        odsp.mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    lu.pad_center = lambda data, size, axis=-1: odsp.pad_center(data, size)
    lu.tiny = odsp.tiny
    lu.normalize = lambda S, norm=None: S
    lib.filters, lib.util = lf, lu
    sys.modules.update({"librosa": lib, "librosa.filters": lf, "librosa.util": lu})
    torch.cuda.FloatTensor = torch.FloatTensor
    torch.cuda.LongTensor = torch.LongTensor
    torch.cuda.HalfTensor = torch.HalfTensor
    torch.cuda.ByteTensor = lambda *s: torch.zeros(*s, dtype=torch.bool)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self


class InjectNormal:
    """Replace Tensor.normal_ so the n-th call copies the n-th injected tensor."""

    def __init__(self, zs):
        self.zs, self.i = list(zs), 0

    def __enter__(self):
        self.orig = torch.Tensor.normal_
        inj = self

        def fake(t, *a, **k):
            z = inj.zs[inj.i]
            inj.i += 1
            assert tuple(t.shape) == tuple(z.shape), (t.shape, z.shape)
            return t.copy_(z)
        torch.Tensor.normal_ = fake
        return self

    def __exit__(self, *a):
        torch.Tensor.normal_ = self.orig


def masks_from_seed(seed, shape):
    g = np.random.Generator(np.random.PCG64(seed))
    return (g.random(shape) < 0.5).astype(np.uint8)


class InjectDropout:
    """Replace model.F.dropout for training=True calls by x * mask * 2 with queued masks."""

    def __init__(self, model_mod, masks):
        self.mod, self.masks, self.i = model_mod, masks, 0

    def __enter__(self):
        self.orig = self.mod.F.dropout
        inj = self

        def fake(x, p=0.5, training=True, inplace=False):
            if not training:
                return x
            assert p == 0.5
            m = torch.from_numpy(inj.masks[inj.i].astype(np.float32))
            inj.i += 1
            assert tuple(m.shape) == tuple(x.shape), (m.shape, x.shape)
            return x * m * 2.0
        class _FProxy:
            dropout = staticmethod(fake)

            def __getattr__(self, k):
                return getattr(torch.nn.functional, k)
        self.mod.F = _FProxy()
        return self

    def __exit__(self, *a):
        self.mod.F = torch.nn.functional


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest()[:16]


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print("wrote", name, os.path.getsize(path) // 1024, "KiB")


# ------------------------------------------------------------------ generators
def gen_hparams():
    from common import hparams as rh
    out = {"create_hparams": vars(rh.create_hparams()), "create_hparams_stage": vars(rh.create_hparams_stage())}
    with open(os.path.join(HERE, "hparams.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote hparams.json")


def ref_waveglow(cfg, seed=16807):
    from waveglow import glow
    m = glow.WaveGlow(**cfg)
    m = glow.WaveGlow.remove_weightnorm(m)
    missing = m.load_state_dict(synth.waveglow_state_dict(cfg, seed), strict=True)
    m.eval()
    return m


def gen_waveglow():
    for tag, hop, B, T in (("hop160", 160, 2, 20), ("hop256", 256, 1, 12)):
        cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=hop)
        m = ref_waveglow(cfg)
        mel = synth.synthetic_mel(B, T, seed=1234)
        L = T * hop // 8
        zs = synth.synthetic_z(B, L, cfg, seed=4321)
        with torch.no_grad(), InjectNormal(zs) as inj:
            audio = m.infer(mel, sigma=0.6)
            assert inj.i == 3
        assert audio.shape == (B, T * hop)
        # forward (training direction) KAT on the same weights: audio -> z, log_s, logdet
        g = np.random.Generator(np.random.PCG64(99))
        wav = torch.from_numpy(np.clip(g.standard_normal((B, T * hop), dtype=np.float32) * 0.1, -1, 1))
        with torch.no_grad():
            z, log_s, log_det = m((mel, wav))
        save("waveglow_%s.npz" % tag, hop=hop, B=B, T=T, sigma=0.6, mel_seed=1234, z_seed=4321,
             mel_sha=sha(mel.numpy()), z_sha=sha(np.concatenate([z_.numpy().ravel() for z_ in zs])),
             audio=audio, fwd_audio_in=wav, fwd_z=z,
             fwd_log_s_sum=np.array([float(x.double().sum()) for x in log_s]),
             fwd_log_det=np.array([float(x) for x in log_det]))
        if tag == "hop160":
            # denoiser bias path: infer(zeros(1,80,88), sigma=0)  (denoiser.py:44-61)
            from waveglow.denoiser import Denoiser
            den = Denoiser(m, filter_length=1024, hop_length=160, win_length=1024, mode="zeros")
            with torch.no_grad():
                den_out = den(audio, strength=0.005)
                den_out_strong = den(audio, strength=1.0)
            save("denoiser_hop160.npz", bias_spec=den.bias_spec, audio_in=audio, out_0005=den_out,
                 out_1=den_out_strong)


def gen_denoiser_hop256():
    """Denoiser at the metric's rate (hop 256 / 22.05 kHz, what bench.py's end-to-end figures build): the bias spectrum
    of the hop-256 model (denoiser.py:44-61) and the spectral subtraction of a hop-256 utterance (denoiser.py:63-68)."""
    from waveglow.denoiser import Denoiser
    hop, B, T = 256, 2, 14
    cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=hop)
    m = ref_waveglow(cfg)
    mel = synth.synthetic_mel(B, T, seed=4242)
    zs = synth.synthetic_z(B, T * hop // 8, cfg, seed=2424)
    with torch.no_grad(), InjectNormal(zs):
        audio = m.infer(mel, sigma=0.6)
    den = Denoiser(m, filter_length=1024, hop_length=hop, win_length=1024, mode="zeros")
    with torch.no_grad():
        out_0005 = den(audio, strength=0.005)
        out_1 = den(audio, strength=1.0)
    save("denoiser_hop256.npz", hop=hop, B=B, T=T, mel_seed=4242, z_seed=2424, bias_spec=den.bias_spec, audio_in=audio,
         out_0005=out_0005, out_1=out_1)


def gen_stft():
    from common.stft import STFT
    from common.layers import TacotronSTFT
    g = np.random.Generator(np.random.PCG64(5))
    y = torch.from_numpy(np.clip(g.standard_normal((2, 4000), dtype=np.float32) * 0.3, -1, 1))
    out = {"y": y}
    for hop in (160, 256):
        st = STFT(1024, hop, 1024)
        mag, ph = st.transform(y)
        rec = st.inverse(mag, ph)
        out["mag_%d" % hop], out["phase_%d" % hop], out["rec_%d" % hop] = mag, ph, rec
    ts = TacotronSTFT(1024, 160, 1024, 80, 16000, 0.0, 8000.0)
    out["mel_16k"] = ts.mel_spectrogram(y)
    out["mel_basis_16k"] = ts.mel_basis
    ts2 = TacotronSTFT()  # library defaults: 22.05 kHz / hop 256 (layers.py:75-77)
    out["mel_22k"] = ts2.mel_spectrogram(y)
    save("stft.npz", **out)


def gen_masks():
    from common.utils import get_mask_from_lengths_window_and_time_step as ref_mask
    cases, outs = [], {}
    for lengths in ([30], [1], [5, 30, 17], [64, 3]):
        for W in (20, 3):
            for t in (0, 1, 2, 3, 10, 19, 20, 21, 29, 30, 45, 49, 50, 51, 63, 64, 90, 200):
                m = ref_mask(torch.LongTensor(lengths), W, t)
                key = "m_%d" % len(cases)
                cases.append({"lengths": lengths, "W": W, "t": t, "key": key})
                outs[key] = m.numpy().astype(np.uint8)
    outs["cases"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    save("attn_masks.npz", **outs)


def gen_tacotron():
    from common import hparams as rh
    from common import model as rmodel
    for tag, Tin, max_steps, gate_bias, n_sym in (("nostop", 30, 60, -10.0, 5816),
                                                  ("stop", 24, 60, -0.08, 5816),
                                                  ("mono40", 16, 16, -10.0, 40)):
        hp = rh.create_hparams_stage(max_decoder_steps=max_steps, n_symbols=n_sym)
        sd = synth.tacotron_state_dict(hp, seed=16807, gate_bias=gate_bias)
        m = rmodel.Tacotron2(hp)
        m.load_state_dict(sd, strict=True)
        m.eval()
        ppg = synth.synthetic_ppg(Tin, n_sym, seed=0, alpha=0.002 if n_sym > 100 else 0.1)
        E, P = hp.symbols_embedding_dim, hp.prenet_dim
        enc_masks = masks_from_seed(777, (2, 1, Tin, E))
        dec_masks = masks_from_seed(778, (max_steps, 2, 1, P))
        queue = [enc_masks[0], enc_masks[1]] + [dec_masks[t, j] for t in range(max_steps) for j in range(2)]
        x = torch.from_numpy(ppg).float().transpose(0, 1).unsqueeze(0)
        with torch.no_grad(), InjectDropout(rmodel, queue):
            mel, mel_post, gate, align = m.inference(x)
            memory = None
        # encoder output alone (same masks) for stage-wise parity
        with torch.no_grad(), InjectDropout(rmodel, queue[:2]):
            memory = m.encoder.inference(x)
        print(tag, "Tout =", mel.shape[2])
        save("tacotron_%s.npz" % tag, Tin=Tin, max_steps=max_steps, gate_bias=gate_bias, n_symbols=n_sym,
             ppg_seed=0, enc_mask_seed=777, dec_mask_seed=778, ppg_sha=sha(ppg),
             memory=memory, mel=mel, mel_post=mel_post, gate=gate, align=align)


def gen_waveglow_old():
    """Legacy layout (src/waveglow/glow_old.py): stride 256, odd flows condition on the second half."""
    from waveglow import glow_old as rold
    cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=256)
    m = rold.WaveGlow(**{k: v for k, v in cfg.items() if k != "hop_length"})
    m = rold.WaveGlow.remove_weightnorm(m)
    m.load_state_dict(synth.waveglow_state_dict(cfg), strict=True)
    m.eval()
    B, T = 1, 10
    mel = synth.synthetic_mel(B, T, seed=77)
    zs = synth.synthetic_z(B, T * 32, cfg, seed=78)
    with torch.no_grad(), InjectNormal(zs):
        audio = m.infer(mel, sigma=0.6)
    save("waveglow_old_hop256.npz", B=B, T=T, sigma=0.6, mel_seed=77, z_seed=78, audio=audio)


def gen_waveglow_train():
    """Reference forward + WaveGlowLoss + autograd backward on the weight-normed model (what
    train_waveglow.py:121-133 does per step): loss value and per-parameter gradients."""
    from waveglow import glow
    cfg = dict(synth.WAVEGLOW_CONFIG)
    B, T, hop = 2, 12, 160
    m = glow.WaveGlow(**cfg)
    sd = synth.waveglow_state_dict(cfg)
    wn_sd = {}
    for k, v in sd.items():
        if k.startswith("WN.") and k.endswith(".weight") and ".end." not in k:
            wn_sd[k[:-6] + "weight_v"] = v
            wn_sd[k[:-6] + "weight_g"] = v.flatten(1).norm(dim=1).view(-1, 1, 1)
        else:
            wn_sd[k] = v
    m.load_state_dict(wn_sd, strict=True)
    m.train()
    g = np.random.Generator(np.random.PCG64(123))
    N = T * hop
    wav = torch.from_numpy(np.clip(g.standard_normal((B, N), dtype=np.float32) * 0.1, -1, 1))
    mel = synth.synthetic_mel(B, T + 1, seed=55)          # frames = N//hop + 1 as Mel2Samp yields
    m.zero_grad()
    out = m((mel, wav))
    loss = glow.WaveGlowLoss(0.7071)(out)
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    keep = ["upsample.weight", "upsample.bias", "WN.0.start.weight_v", "WN.0.start.weight_g", "WN.0.start.bias",
            "WN.0.in_layers.0.weight_v", "WN.0.in_layers.0.weight_g", "WN.0.in_layers.0.bias",
            "WN.0.in_layers.7.weight_v", "WN.5.cond_layers.3.weight_v", "WN.5.cond_layers.3.bias",
            "WN.11.res_skip_layers.7.weight_v", "WN.11.res_skip_layers.6.weight_v", "WN.11.res_skip_layers.6.bias",
            "WN.3.end.weight", "WN.3.end.bias", "WN.11.end.weight", "convinv.0.conv.weight", "convinv.4.conv.weight",
            "convinv.11.conv.weight"]
    arrs = {"loss": loss.detach(), "B": B, "T": T, "hop": hop, "mel_seed": 55, "wav": wav,
            "names": np.frombuffer(json.dumps(sorted(grads)).encode(), dtype=np.uint8),
            "norms": np.array([float(grads[k].double().norm()) for k in sorted(grads)])}
    for k in keep:                       # strided subsample (<= 4096 values) of each kept gradient
        flat = grads[k].reshape(-1)
        arrs["g:" + k] = flat[::max(1, -(-flat.numel() // 4096))].contiguous()
    save("waveglow_train.npz", **arrs)


TRAIN_KEEP = ["upsample.weight", "upsample.bias", "WN.0.start.weight_v", "WN.0.start.weight_g", "WN.0.start.bias",
              "WN.0.in_layers.0.weight_v", "WN.0.in_layers.0.weight_g", "WN.0.in_layers.0.bias",
              "WN.0.in_layers.7.weight_v", "WN.5.cond_layers.3.weight_v", "WN.5.cond_layers.3.bias",
              "WN.11.res_skip_layers.7.weight_v", "WN.11.res_skip_layers.6.weight_v", "WN.11.res_skip_layers.6.bias",
              "WN.3.end.weight", "WN.3.end.bias", "WN.11.end.weight", "convinv.0.conv.weight", "convinv.4.conv.weight",
              "convinv.11.conv.weight"]


def cfg5_batch(B):
    """The seeded training batch of gen_waveglow_train_cfg5 (regenerable: the audio is NOT committed).  Audio
    [B, segment_length 10000] ~ N(0, 0.1^2) clipped to +-1 from PCG64(500 + B); mel [B, 80, 10000 // 160 + 1 = 63]
    (the frame count Mel2Samp yields, mel2samp.py:107-118) from facppg.synth.synthetic_mel(seed 700 + B)."""
    g = np.random.Generator(np.random.PCG64(500 + B))
    wav = torch.from_numpy(np.clip(g.standard_normal((B, 10000), dtype=np.float32) * 0.1, -1, 1))
    mel = synth.synthetic_mel(B, 10000 // 160 + 1, seed=700 + B)
    return mel, wav


def gen_waveglow_train_cfg5():
    """BASELINE config 5 at ITS OWN shape (waveglow/config.json:8,14: batch_size 3, segment_length 10000, hop 160): the
    imported reference's forward + WaveGlowLoss + autograd backward (train_waveglow.py:118-147, glow.py:43-59,208-250) on
    the weight-normed model, at batch 3 (the reference's per-GPU batch) and at batch 12 (the second entry bench.py times,
    where the bf16 launch plan flips to fused forward layers / 256 x 256 weight-gradient tiles / 64 positions per tile).
    Commits loss, all 938 gradient norms and strided sub-samples of 20 gradients; the inputs regenerate from seeds."""
    import time
    from waveglow import glow
    cfg = dict(synth.WAVEGLOW_CONFIG)
    sd = synth.waveglow_state_dict(cfg)
    wn_sd = {}
    for k, v in sd.items():
        if k.startswith("WN.") and k.endswith(".weight") and ".end." not in k:
            wn_sd[k[:-6] + "weight_v"] = v
            wn_sd[k[:-6] + "weight_g"] = v.flatten(1).norm(dim=1).view(-1, 1, 1)
        else:
            wn_sd[k] = v
    for B in (3, 12):
        m = glow.WaveGlow(**cfg)
        m.load_state_dict(wn_sd, strict=True)
        m.train()
        mel, wav = cfg5_batch(B)
        m.zero_grad()
        t0 = time.time()
        out = m((mel, wav))
        loss = glow.WaveGlowLoss(0.7071)(out)
        loss.backward()
        print("reference step at B = %d x 10000: %.1f s, loss %.6f" % (B, time.time() - t0, float(loss)))
        grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        arrs = {"loss": loss.detach(), "B": B, "segment_length": 10000, "hop": 160, "wav_seed": 500 + B, "mel_seed": 700 + B,
                "wav_sha": np.frombuffer(sha(wav.numpy()).encode(), dtype=np.uint8),
                "names": np.frombuffer(json.dumps(sorted(grads)).encode(), dtype=np.uint8),
                "norms": np.array([float(grads[k].double().norm()) for k in sorted(grads)])}
        for k in TRAIN_KEEP:
            flat = grads[k].reshape(-1)
            arrs["g:" + k] = flat[::max(1, -(-flat.numel() // 4096))].contiguous()
        save("waveglow_train_cfg5_B%d.npz" % B, **arrs)


def gen_e2e():
    """The body of the reference CLI (generate_synthesis.py:55-62,75-95) on synthetic checkpoints: PPG ->
    get_inference -> waveglow_audio(sigma 0.6, is_cuda_output=True) -> Denoiser('zeros')(strength 0.005), with the
    weight-normed model for the denoiser and the weight-norm-removed one for synthesis, exactly as the script
    builds them.  Dropout masks and z are injected; the decoder runs under the CLI's own hparams
    (create_hparams_stage(): max_decoder_steps 1000) and stops on its gate."""
    from common import hparams as rh
    from common import model as rmodel
    from common.utils import get_inference, waveglow_audio
    from waveglow import glow
    from waveglow.denoiser import Denoiser
    Tin, gate_bias, n_sym = 40, -0.1, 5816
    hp = rh.create_hparams_stage()
    taco = rmodel.Tacotron2(hp)
    taco.load_state_dict(synth.tacotron_state_dict(hp, seed=16807, gate_bias=gate_bias), strict=True)
    taco.eval()
    cfg = dict(synth.WAVEGLOW_CONFIG)
    sd = synth.waveglow_state_dict(cfg)
    wn_sd = {}
    for k, v in sd.items():
        if k.startswith("WN.") and k.endswith(".weight") and ".end." not in k:
            wn_sd[k[:-6] + "weight_v"] = v
            wn_sd[k[:-6] + "weight_g"] = v.flatten(1).norm(dim=1).view(-1, 1, 1)
        else:
            wn_sd[k] = v
    wg_for_denoiser = glow.WaveGlow(**cfg)
    wg_for_denoiser.load_state_dict(wn_sd, strict=True)
    denoiser = Denoiser(wg_for_denoiser, mode="zeros")          # generate_synthesis.py:58-61 (bias: sigma 0, any z)
    wg = ref_waveglow(cfg)                                      # load_waveglow_model: weight norm removed, eval
    ppg = synth.synthetic_ppg(Tin, n_sym, seed=3, alpha=0.002)
    steps = hp.max_decoder_steps
    enc_masks = masks_from_seed(881, (2, 1, Tin, hp.symbols_embedding_dim))
    dec_masks = masks_from_seed(882, (steps, 2, 1, hp.prenet_dim))
    queue = [enc_masks[0], enc_masks[1]] + [dec_masks[t, j] for t in range(steps) for j in range(2)]
    with torch.no_grad(), InjectDropout(rmodel, queue):
        ac_mel = get_inference(ppg, taco, False)
    Tout = ac_mel.shape[2]
    assert Tout < steps, "the gate never fired"
    with torch.no_grad(), InjectDropout(rmodel, queue):       # stop decisions must not sit on the fp32 edge
        gate = taco.inference(torch.from_numpy(ppg).float().t().unsqueeze(0))[2]
    margin = float(gate.abs().min())
    print("e2e: smallest |gate logit| over the %d steps: %.4f" % (Tout, margin))
    assert margin > 1e-3
    zs = synth.synthetic_z(1, Tout * 20, cfg, seed=883)
    with InjectNormal(zs) as inj:
        ac_wav = waveglow_audio(ac_mel, wg, 0.6, True)
        assert inj.i == 3
    with torch.no_grad():
        out = denoiser(ac_wav, strength=0.005)[:, 0].cpu().numpy().T       # what wavfile.write receives: [N, 1]
    print("e2e: Tin", Tin, "Tout", Tout, "samples", out.shape)
    save("e2e_cli.npz", Tin=Tin, gate_bias=gate_bias, n_symbols=n_sym, ppg_seed=3, ppg_alpha=0.002, enc_mask_seed=881,
         dec_mask_seed=882, z_seed=883, sigma=0.6, strength=0.005, Tout=Tout, ppg_sha=sha(ppg), mel_post=ac_mel,
         audio=ac_wav, ac_wav=out)


def gen_e2e_metric():
    """The metric's own case at the shape bench.py times (BASELINE.json `metric`, SURVEY.md 8d config 1 at 22.05 kHz / hop 256):
    PPG [200 x 5816] seed 0 -> get_inference (gate bias -10: the decoder runs its 200 steps) -> waveglow_audio(sigma 0.6) at
    hop 256 -> Denoiser(hop_length=256, 'zeros')(strength 0.005), on the imported reference (generate_synthesis.py:74-98,
    glow.py:252-293) with the models bench.py's EndToEnd builds; dropout masks and z injected.  ~3 s of CPU."""
    from common import hparams as rh
    from common import model as rmodel
    from common.utils import get_inference, waveglow_audio
    from waveglow.denoiser import Denoiser
    Tin, hop, n_sym = 200, 256, 5816
    hp = rh.create_hparams_stage(max_decoder_steps=Tin, n_symbols=n_sym)
    taco = rmodel.Tacotron2(hp)
    taco.load_state_dict(synth.tacotron_state_dict(hp, seed=16807, gate_bias=-10.0), strict=True)
    taco.eval()
    cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=hop)
    wg = ref_waveglow(cfg)
    denoiser = Denoiser(wg, filter_length=1024, hop_length=hop, win_length=1024, mode="zeros")
    ppg = synth.synthetic_ppg(Tin, n_sym, seed=0, alpha=0.002)
    enc_masks = masks_from_seed(991, (2, 1, Tin, hp.symbols_embedding_dim))
    dec_masks = masks_from_seed(992, (Tin, 2, 1, hp.prenet_dim))
    queue = [enc_masks[0], enc_masks[1]] + [dec_masks[t, j] for t in range(Tin) for j in range(2)]
    with torch.no_grad(), InjectDropout(rmodel, queue):
        ac_mel = get_inference(ppg, taco, False)
    Tout = ac_mel.shape[2]
    assert Tout == Tin, Tout
    zs = synth.synthetic_z(1, Tout * hop // 8, cfg, seed=993)
    with InjectNormal(zs) as inj:
        ac_wav = waveglow_audio(ac_mel, wg, 0.6, True)
        assert inj.i == 3
    with torch.no_grad():
        out = denoiser(ac_wav, strength=0.005)[:, 0]
    assert ac_wav.shape == out.shape == (1, Tout * hop)
    print("e2e_metric: Tout", Tout, "samples", out.shape[1], "rms", float(out.double().pow(2).mean().sqrt()))
    save("e2e_metric.npz", Tin=Tin, hop=hop, gate_bias=-10.0, n_symbols=n_sym, ppg_seed=0, ppg_alpha=0.002, enc_mask_seed=991,
         dec_mask_seed=992, z_seed=993, sigma=0.6, strength=0.005, Tout=Tout, ppg_sha=sha(ppg), mel_post=ac_mel, audio=ac_wav,
         audio_denoised=out)


def gen_kaldi_data():
    """The front-end's DATA files as the reference ships them (data/feats: LDA matrix, pdf -> monophone reduction, splice
    options; its own tests read the same files through test/data symlinks, test_feat.py:74-87, test_ppg.py:25-27)."""
    import shutil
    dst = os.path.join(HERE, "kaldi_feats")
    os.makedirs(dst, exist_ok=True)
    for name in ("final.mat", "reduce_dim.mat", "splice_opts"):
        shutil.copyfile(os.path.join(os.path.dirname(REF), "data", "feats", name), os.path.join(dst, name))
        os.chmod(os.path.join(dst, name), 0o644)
    print("copied data/feats ->", dst)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    install_shims()
    gen_hparams()
    gen_masks()
    gen_stft()
    gen_waveglow()
    gen_denoiser_hop256()
    gen_waveglow_old()
    gen_waveglow_train()
    gen_waveglow_train_cfg5()
    gen_tacotron()
    gen_e2e()
    gen_e2e_metric()
    gen_kaldi_data()


if __name__ == "__main__":
    if len(sys.argv) > 1:           # python make_golden.py gen_denoiser_hop256 ...: only the named generators
        torch.manual_seed(0)
        torch.set_num_threads(8)
        install_shims()
        for name in sys.argv[1:]:
            globals()[name]()
    else:
        main()
