#!/usr/bin/env python3
"""Host time from the start of a batch-1 synthesize() call to the encoder's launch, the decoder's launch and the length read
(the GPU has nothing to do before the first of them): python tools/host_gap_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd")]
import torch
import bench
from facppg import lib as flib, pipeline

dev = torch.device("cuda", 0)
e = bench.EndToEnd(dev, [200])
for i in range(5):
    e.step(i)
torch.cuda.synchronize()
L = flib.load()
marks = {}


def wrap(name):
    fn = getattr(L, name)

    def w(*a):
        marks.setdefault(name, time.perf_counter())
        return fn(*a)
    w.restype, w.argtypes = fn.restype, fn.argtypes
    setattr(L, name, w)


for n in ("facppg_taco_encode", "facppg_taco_decode", "facppg_taco_collect_frames", "facppg_wg_infer_seeded"):
    wrap(n)
pad = pipeline.pad_ppgs


def pad_w(*a, **k):
    marks.setdefault("pad_ppgs_in", time.perf_counter())
    out = pad(*a, **k)
    marks.setdefault("pad_ppgs_out", time.perf_counter())
    return out


pipeline.pad_ppgs = pad_w
rows = []
for i in range(20):
    marks.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e.step(100 + i)
    t_ret = time.perf_counter()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    rows.append({k: (v - t0) * 1e3 for k, v in marks.items()} | {"returned": (t_ret - t0) * 1e3, "done": (t1 - t0) * 1e3})
keys = ["pad_ppgs_in", "pad_ppgs_out", "facppg_taco_encode", "facppg_taco_decode", "facppg_taco_collect_frames", "facppg_wg_infer_seeded", "returned", "done"]
for k in keys:
    v = sorted(r[k] for r in rows if k in r)
    print("%-28s median %.3f ms (min %.3f)" % (k, v[len(v) // 2], v[0]))
