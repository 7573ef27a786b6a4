#!/usr/bin/env python3
"""Host-side profile of the batch-1 headline step (where the Python time goes while the GPU waits): cProfile over 30 steps."""
import cProfile
import contextlib
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd")]
import torch
import bench

dev = torch.device("cuda", 0)
e = bench.EndToEnd(dev, [200])
for i in range(5):
    e.step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(30):
    e.step(100 + i)
    torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
