#!/usr/bin/env python3
"""Soak: N end-to-end batch-1 steps (decoder heaters, resident rows, split decoder hand-offs) + N/10 three-utterance batches;
every result must equal the first one bit for bit (same seed)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd")]
import torch, bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
for lens in ([200], [120, 90, 57]):
    e = bench.EndToEnd(dev, lens)
    ref = None
    t0 = time.time()
    n = N if len(lens) == 1 else max(10, N // 10)
    for i in range(n):
        wavs, tout = e.step(7)
        torch.cuda.synchronize()
        if ref is None:
            ref = [w.clone() for w in wavs]
        else:
            assert all(torch.equal(a, b) for a, b in zip(wavs, ref)), "step %d differs" % i
    print("lens %s: %d identical steps, %.2f ms each" % (lens, n, (time.time() - t0) / n * 1e3), flush=True)
