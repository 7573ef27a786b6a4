"""Round 5: the streamed batch-1 step (ConditioningStream) under different settings, INTERLEAVED in one process (boxes differ by
a few per cent; so do minutes on one box): CONFIGS = ';'-separated env settings 'K=V,K=V', each run REPS times for N steps in
round-robin order; prints the median ms/step per setting and the stage events of one further step."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fac-via-ppg_amd")):
    sys.path.insert(0, p)

import bench  # noqa: E402
from facppg import pipeline  # noqa: E402


def setenv(cfg):
    keys = []
    for kv in cfg.split(","):
        if "=" in kv:
            k, v = kv.split("=", 1)
            os.environ[k] = v
            keys.append(k)
    return keys


def main():
    dev = torch.device("cuda", 0)
    T = int(os.environ.get("T", "200"))
    e = bench.EndToEnd(dev, [T])
    configs = os.environ.get("CONFIGS", "FACPPG_STREAM=0;FACPPG_STREAM=1").split(";")
    reps, n = int(os.environ.get("REPS", "3")), int(os.environ.get("N", "10"))
    res = {c: [] for c in configs}
    for r in range(reps):
        for c in configs:
            keys = setenv(c)
            for i in range(3):
                e.step(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                e.step(10 + i)
            torch.cuda.synchronize()
            res[c].append((time.perf_counter() - t0) / n * 1e3)
            for k in keys:
                del os.environ[k]
    for c in configs:
        keys = setenv(c)
        timer = pipeline.StageTimer()
        e.step(99, timer=timer)
        st = timer.stages_ms()
        for k in keys:
            del os.environ[k]
        print("%-60s median %.3f ms/step (%s); stages %s" % (c, sorted(res[c])[len(res[c]) // 2], " ".join("%.2f" % v for v in res[c]),
                                                              {k: round(v, 2) for k, v in st.items()}))


if __name__ == "__main__":
    main()
