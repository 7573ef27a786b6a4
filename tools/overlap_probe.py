#!/usr/bin/env python3
"""Where does the time of back-to-back end-to-end batches go?  hipEvents around the acoustic model (side stream) and the vocoder
(main stream) of every job of facppg.pipeline.synthesize_stream, with and without the overlap:
python tools/overlap_probe.py [batch [decoder CUs while overlapped [steps]]]"""
import contextlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd")]
import torch
import bench
from facppg import pipeline

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
CUS = int(sys.argv[2]) if len(sys.argv) > 2 else 32
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda", 0)
lens = bench.config3_lengths(B, 7)
e = bench.EndToEnd(dev, lens)
marks = []
ac0, vo0 = pipeline._acoustic, pipeline._vocode


def timed(fn, tag):
    def f(*a, **k):
        s = torch.cuda.current_stream()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record(s)
        h0 = time.perf_counter()
        r = fn(*a, **k)
        a1.record(s)
        marks.append((tag, a0, a1, h0, time.perf_counter()))
        return r
    return f


pipeline._acoustic, pipeline._vocode = timed(ac0, "acoustic"), timed(vo0, "vocoder")
for overlap in (False, True):
    with contextlib.redirect_stdout(sys.stderr):
        gen = e.steady_state(overlap=overlap, acoustic_workgroups=CUS)
        for _ in range(2):
            next(gen)
        torch.cuda.synchronize()
        del marks[:]
        t0 = time.perf_counter()
        base = torch.cuda.Event(enable_timing=True)
        base.record(torch.cuda.current_stream())
        for _ in range(STEPS):
            next(gen)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / STEPS * 1e3
        gen.close()
    print("overlap=%s: %.2f ms per batch" % (overlap, wall))
    for tag, a0, a1, h0, h1 in marks[:12]:
        print("  %-8s device %7.2f .. %7.2f ms (%6.2f)   host call %7.2f .. %7.2f ms" % (
            tag, base.elapsed_time(a0), base.elapsed_time(a1), a0.elapsed_time(a1), (h0 - t0) * 1e3, (h1 - t0) * 1e3))
