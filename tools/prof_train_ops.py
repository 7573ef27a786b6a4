#!/usr/bin/env python3
"""torch.profiler view of one bf16 training step (which host ops launch the small kernels): python tools/prof_train_ops.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile
from facppg import synth
from test_gpu_e2e import weightnorm_state_dict
from waveglow.glow import WaveGlow, WaveGlowLoss

B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = dict(synth.WAVEGLOW_CONFIG)
m = WaveGlow(**cfg)
m.load_state_dict(weightnorm_state_dict(synth.waveglow_state_dict(cfg)), strict=True)
m = m.cuda().train()
m.train_precision = "bf16"
from waveglow.optim import Adam
opt = Adam(m.parameters(), lr=1e-5)
crit = WaveGlowLoss(0.7071)
g = np.random.Generator(np.random.PCG64(1))
audio = torch.from_numpy(np.clip(g.standard_normal((B, 10000), dtype=np.float32) * 0.1, -1, 1)).cuda()
mel = synth.synthetic_mel(B, 63, seed=1).cuda()


def step():
    m.zero_grad()
    loss = crit(m((mel, audio)))
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=70, max_name_column_width=90))
print(prof.key_averages().table(sort_by="count", row_limit=30, max_name_column_width=60))
# who issues the fills / copies: aggregate by the innermost Python frames of this repo
from collections import Counter
for op in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::zeros_like"):
    c = Counter()
    for ev in prof.events():
        if ev.name == op:
            frames = [f for f in (ev.stack or []) if "fac-via-ppg_amd" in f or "tools/" in f or "optim" in f or "autograd" in f]
            c[" <- ".join(frames[:3]) or "(no python frame: autograd engine / C++)"] += 1
    print("==", op, sum(c.values()))
    for k, v in c.most_common(8):
        print("   %5d  %s" % (v, k[:260]))
