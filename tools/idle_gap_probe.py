#!/usr/bin/env python3
"""Does what runs BEFORE WaveGlow.infer change how long it takes?  One 200-frame utterance from a drained GPU: back to back, after
an idle gap, after a Tacotron2.inference (the end-to-end step's order)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd")]
import torch
import bench
from facppg import pipeline
dev = torch.device("cuda", 0)
e = bench.EndToEnd(dev, [200])
for i in range(3):
    e.step(i)
wg, taco = e.waveglow, e.tacotron
x, lens = pipeline.pad_ppgs(e.ppgs, device=dev)
mel = taco.inference(x, seed=1)[1].contiguous()
def infer_ms():
    torch.cuda.synchronize(); t0 = time.perf_counter(); wg.infer(mel, sigma=0.6, seed=1); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
def stat(v): v = sorted(v); return "median %.3f  min %.3f  max %.3f ms" % (v[len(v) // 2], v[0], v[-1])
print("back to back               ", stat([infer_ms() for _ in range(15)]))
for gap in (0.002, 0.006, 0.02, 0.1):
    r = []
    for _ in range(10):
        time.sleep(gap); r.append(infer_ms())
    print("after %5.0f ms idle         " % (gap * 1e3), stat(r))
r = []
for _ in range(10):
    taco.inference(x, seed=1); r.append(infer_ms())
print("after Tacotron2.inference  ", stat(r))
r = []
for _ in range(10):
    a = torch.randn(4096, 4096, device=dev); b = a @ a; torch.cuda.synchronize(); r.append(infer_ms())
print("after a 4096^3 fp32 matmul ", stat(r))

# what kind of activity in front of the vocoder removes the slow start?  (after 20 ms of idle each time)
arena = torch.empty(400 * 1024 * 1024 // 4, device=dev)           # stands in for "touch a few hundred MB" (not the weights themselves)
big = torch.randn(8192, 8192, device=dev)
def after(label, fn, n=10):
    r = []
    for _ in range(n):
        torch.cuda.synchronize(); time.sleep(0.02); fn(); torch.cuda.synchronize(); r.append(infer_ms())
    print("%-46s" % label, stat(r))
after("20 ms idle, then nothing", lambda: None)
after("20 ms idle, then read 400 MB (sum)", lambda: arena.sum())
after("20 ms idle, then 1 fp32 matmul 8192^3 (~8 ms)", lambda: big @ big)
after("20 ms idle, then 1 fp32 matmul 2048^3 (~0.2 ms)", lambda: big[:2048, :2048] @ big[:2048, :2048])
after("20 ms idle, then WaveGlow.infer itself", lambda: wg.infer(mel, sigma=0.6, seed=1))
