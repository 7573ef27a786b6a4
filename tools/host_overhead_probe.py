#!/usr/bin/env python3
"""Where the host spends its time in ONE end-to-end batch-1 step (the metric's case): perf_counter around the Python-level pieces,
with the GPU drained in between so that every figure is pure host + launch latency (the latency the step pays when the GPU has
nothing else queued)."""
import contextlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd")]
import torch  # noqa: E402
import bench  # noqa: E402
from facppg import pipeline  # noqa: E402

dev = torch.device("cuda", 0)
e = bench.EndToEnd(dev, [200])
with contextlib.redirect_stdout(sys.stderr):
    for i in range(3):
        e.step(i)
torch.cuda.synchronize()


def t(label, fn, n=10):
    """every call starts on a drained GPU: host = until the call returns, total = until the GPU is done as well"""
    host = tot = 0.0
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        host += t1 - t0
        tot += time.perf_counter() - t0
    print("%-52s host %.3f ms   total %.3f ms" % (label, host / n * 1e3, tot / n * 1e3), flush=True)
    return r


wg, taco, den = e.waveglow, e.tacotron, e.denoiser
t("waveglow._handle (packed-weight validity check)", lambda: wg._handle(dev))
t("tacotron._handle", lambda: taco._handle(dev))
x, lens = t("pad_ppgs (PPG upload + transpose)", lambda: pipeline.pad_ppgs(e.ppgs, device=dev))
with contextlib.redirect_stdout(sys.stderr):
    mel = t("tacotron.inference (encoder + decoder + postnet)", lambda: taco.inference(x, seed=1)[1], n=5).contiguous()
audio = t("waveglow.infer", lambda: wg.infer(mel, sigma=0.6, seed=1), n=5)
t("waveglow.infer after prepare()", lambda: (wg.prepare(dev), wg.infer(mel, sigma=0.6, seed=1))[1], n=5)
t("denoiser", lambda: den(audio, strength=0.005), n=5)
with contextlib.redirect_stdout(sys.stderr):
    t("whole step (pipeline.synthesize)", lambda: e.step(7), n=10)
