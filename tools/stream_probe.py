"""Round 5: the streamed batch-1 step (ConditioningStream) vs the unstreamed one: host-clock ms per step and stage events."""
import contextlib
import io
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fac-via-ppg_amd")):
    sys.path.insert(0, p)

import bench  # noqa: E402
from facppg import pipeline  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    T = int(os.environ.get("T", "200"))
    e = bench.EndToEnd(dev, [T])
    for mode in os.environ.get("MODES", "0,1").split(","):
        os.environ["FACPPG_STREAM"] = mode
        for i in range(4):
            e.step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for i in range(n):
            e.step(10 + i)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        timer = pipeline.StageTimer()
        e.step(99, timer=timer)
        st = timer.stages_ms()
        cs = e.waveglow.__dict__.get("_facppg_cond_stream")
        print("FACPPG_STREAM=%s chunk=%s: %.3f ms/step; stages %s; blocks %s" % (
            mode, os.environ.get("FACPPG_STREAM_CHUNK", "64"), ms, {k: round(v, 3) for k, v in st.items()},
            getattr(cs, "cuts", None) if mode == "1" else None))


if __name__ == "__main__":
    main()
