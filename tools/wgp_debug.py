#!/usr/bin/env python3
"""Debugging aid for the persistent WaveGlow launch (csrc/facppg_wgp.hip): run it against the per-layer launches on one
utterance and report where the audio differs (per phase / frame block).  FACPPG_POLL_LIMIT bounds a lost hand-off."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd")]
import torch  # noqa: E402
from facppg import synth  # noqa: E402
from waveglow.glow import WaveGlow  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
hop = 256
cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=hop)
m = WaveGlow.remove_weightnorm(WaveGlow(**cfg))
m.load_state_dict(synth.waveglow_state_dict(cfg))
m = m.cuda().eval()
mel = synth.synthetic_mel(1, T, seed=5).cuda()
zs = synth.synthetic_z(1, T * hop // 8, cfg, seed=6)
os.environ["FACPPG_WG_PERSIST"] = "0"
ref = m.infer(mel, sigma=0.6, z=zs)
torch.cuda.synchronize()
print("per-layer launches done", m.last_launch_shape(), flush=True)
os.environ.pop("FACPPG_WG_PERSIST")
got = m.infer(mel, sigma=0.6, z=zs)
torch.cuda.synchronize()
print("persistent launch done", m.last_launch_shape(), flush=True)
d = (got - ref).abs()
print("max abs diff %.3e, finite %s, equal %s" % (d.max().item(), bool(torch.isfinite(got).all()), bool(torch.equal(got, ref))))
if d.max().item() > 0:
    P = hop // 8
    dd = d.view(T, P, 8).amax(2)          # [frame q][phase]
    bad_ph = (dd.amax(0) > 0).nonzero().flatten().tolist()
    bad_q = (dd.amax(1) > 0).nonzero().flatten().tolist()
    print("phases with differences:", bad_ph)
    print("frames with differences: %d of %d, first %s" % (len(bad_q), T, bad_q[:20]))
import time
for name, env in (("persistent", None), ("per-layer", "0")):
    if env is None:
        os.environ.pop("FACPPG_WG_PERSIST", None)
    else:
        os.environ["FACPPG_WG_PERSIST"] = env
    for _ in range(3):
        m.infer(mel, sigma=0.6, seed=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        m.infer(mel, sigma=0.6, seed=2 + i)
    torch.cuda.synchronize()
    print("%s: %.3f ms per infer (T = %d)" % (name, (time.perf_counter() - t0) / 20 * 1e3, T))
