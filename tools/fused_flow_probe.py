#!/usr/bin/env python3
"""k_wn_flow8 (all layers of a flow in one launch, FACPPG_WN_FUSED = 1 | 2 workers per tile) against one launch per layer
(FACPPG_WN_FUSED = 0) on one utterance: same bits?  how long?  FACPPG_POLL_LIMIT bounds a lost hand-off."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd")]
import torch  # noqa: E402
from facppg import synth  # noqa: E402
from waveglow.glow import WaveGlow  # noqa: E402

hop = 256
cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=hop)
m = WaveGlow.remove_weightnorm(WaveGlow(**cfg))
m.load_state_dict(synth.waveglow_state_dict(cfg))
m = m.cuda().eval()
os.environ["FACPPG_WG_PERSIST"] = "0"
for T in [int(a) for a in sys.argv[1:]] or [200]:
    mel = synth.synthetic_mel(1, T, seed=5).cuda()
    zs = synth.synthetic_z(1, T * hop // 8, cfg, seed=6)
    os.environ["FACPPG_WN_FUSED"] = "0"
    ref = m.infer(mel, sigma=0.6, z=zs)
    torch.cuda.synchronize()
    print("T = %d: per-layer launches %s" % (T, m.last_launch_shape()), flush=True)
    for mode in ("1", "2"):
        os.environ["FACPPG_WN_FUSED"] = mode
        got = m.infer(mel, sigma=0.6, z=zs)
        torch.cuda.synchronize()
        d = (got - ref).abs()
        print("  fused, %s worker(s): max abs diff %.3e, equal %s" % (mode, d.max().item(), bool(torch.equal(got, ref))), flush=True)
    for mode in ("0", "1", "2"):
        os.environ["FACPPG_WN_FUSED"] = mode
        for _ in range(3):
            m.infer(mel, sigma=0.6, seed=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20):
            m.infer(mel, sigma=0.6, seed=2 + i)
        torch.cuda.synchronize()
        print("  FACPPG_WN_FUSED=%s: %.3f ms per infer" % (mode, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
