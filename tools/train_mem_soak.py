#!/usr/bin/env python3
"""Eager bf16 training steps in a row: allocated / reserved device memory must not grow from step to step (round 6 found a reference
cycle through the step's z buffer that kept every step's saved activations alive until the cyclic collector ran).
  python tools/train_mem_soak.py [batch] [steps]"""
import gc
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd")]
import torch
import bench
from waveglow.optim import Adam

B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
gc.disable()                      # nothing may depend on the cyclic collector
dev = torch.device("cuda", 0)
m, crit = bench.make_train_model(dev)
mel, audio = bench.train_batch(dev, B)
opt = Adam(m.parameters(), lr=1e-5)
peaks = []
for i in range(N):
    m.zero_grad()
    loss = crit(m((mel, audio)))
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    peaks.append((torch.cuda.memory_allocated(dev) / 1e9, torch.cuda.memory_reserved(dev) / 1e9))
    if i in (0, 1, 2, N // 2, N - 1):
        print("step %3d: loss %.5f, allocated %.2f GB, reserved %.2f GB" % (i, float(loss), *peaks[-1]), flush=True)
assert peaks[-1][0] <= peaks[2][0] * 1.02 + 0.05, ("allocated memory grows", peaks[2], peaks[-1])
assert peaks[-1][1] <= peaks[2][1] * 1.05 + 0.1, ("reserved memory grows", peaks[2], peaks[-1])
print("ok: no growth over %d steps at batch %d (cyclic GC disabled)" % (N, B))
