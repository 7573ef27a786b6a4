#!/usr/bin/env python3
"""Median ms of the graphed bf16 training step (fwd + loss + bwd + Adam) at the given batch sizes: python tools/time_train_step.py 3 12"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd")]
import torch
import bench
from waveglow.graphed import GraphedTrainStep
from waveglow.optim import Adam

dev = torch.device("cuda", 0)
m, crit = bench.make_train_model(dev)
out = []
for B in [int(a) for a in sys.argv[1:]] or [3, 12]:
    mel, audio = bench.train_batch(dev, B)
    opt = Adam(m.parameters(), lr=1e-5)
    st = GraphedTrainStep(m, crit, opt, warmup=2)
    ts = []
    for i in range(int(os.environ.get("STEPS", "14"))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st(mel, audio)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts = sorted(ts[4:] or ts)
    out.append("B=%d %.2f ms" % (B, ts[len(ts) // 2] * 1e3))
print("  ".join(out))
