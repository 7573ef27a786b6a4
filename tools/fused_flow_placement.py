#!/usr/bin/env python3
"""Where does the dispatcher put k_wn_flow8's workgroups?  (FACPPG_WN_FUSED_DEBUG=16 records XCC_ID / HW_ID per workgroup.)"""
import collections, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd")]
import torch
from facppg import synth, lib as flib
from waveglow.glow import WaveGlow
T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=256)
m = WaveGlow.remove_weightnorm(WaveGlow(**cfg)); m.load_state_dict(synth.waveglow_state_dict(cfg)); m = m.cuda().eval()
mel = synth.synthetic_mel(1, T).cuda()
os.environ["FACPPG_WG_PERSIST"] = "0"
os.environ["FACPPG_WN_FUSED_DEBUG"] = "16"
L = flib.load()
for workers in (1, 2):
    os.environ["FACPPG_WN_FUSED"] = str(workers)
    m.infer(mel, sigma=0.6, seed=0); torch.cuda.synchronize()
    tiles = m.last_launch_shape()[2]
    out = (ctypes.c_uint * 1024)()
    L.facppg_debug_flow_ids(out)
    ids = list(out)[:tiles * workers]
    per_cu = collections.Counter(ids)
    print("workers %d: %d workgroups on %d distinct CUs; workgroups per CU histogram %s" % (
        workers, len(ids), len(per_cu), dict(collections.Counter(per_cu.values()))))
    print("  XCC of workgroup i == i %% 8 for %d of %d" % (sum((v >> 16) == i % 8 for i, v in enumerate(ids)), len(ids)))
    if workers == 2:
        same = sum(ids[i] == ids[i + tiles] for i in range(tiles))
        both_even = sum(1 for cu, n in collections.Counter(ids[:tiles]).items() if n > 1)
        print("  tile's two workers on the same CU: %d of %d; CUs holding two layer-0 workers: %d; CUs holding two layer-1 workers: %d" % (
            same, tiles, both_even, sum(1 for cu, n in collections.Counter(ids[tiles:]).items() if n > 1)))
    print("  first 40 ids:", ["%x" % v for v in ids[:40]])
