#!/usr/bin/env python3
"""Why does bench.py's batch-1 headline run at 0.23 of the fp32 MFMA peak (0.20 ms per layer launch instead of 0.08) when it is a
subprocess of the whole GPU test suite (tests/test_gpu_rccl_single.py), and at ~0.55 on its own?  The child is always
`bench.py --steps 3 --warmup 2 --no-train --no-cpu-baseline [--force-dist]`; what varies is what the PARENT process holds on the GPU
while it waits.  A sampler thread logs the shader clock (rocm-smi) while the child runs.

  python tools/insuite_probe.py            all variants, one line each
"""
import json
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd"), os.path.join(ROOT, "tests")]


def sclk():
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
        m = re.search(r"sclk clock level:\s*\d+:?\s*\((\d+)Mhz\)", out)
        return int(m.group(1)) if m else None
    except Exception:   # noqa: BLE001
        return None


def child(extra, port, child_env=None):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.update(child_env or {})
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    clocks, stop = [], threading.Event()

    def sample():
        while not stop.is_set():
            c = sclk()
            if c:
                clocks.append(c)
            time.sleep(0.2)
    th = threading.Thread(target=sample)
    th.start()
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--no-train", "--no-cpu-baseline"] + extra,
                       capture_output=True, text=True, env=env, timeout=900)
    stop.set()
    th.join()
    try:
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        return {"frac": round(d["roofline"]["frac"], 3), "launch_ms": round(d["roofline"]["avg_launch_ms"], 4), "ms_per_step": round(d["ms_per_step"], 2),
                "stage_ms": {k: round(v, 2) for k, v in d.get("stage_ms", {}).items()}, "sclk_min_max": (min(clocks), max(clocks)) if clocks else None,
                "wall_s": round(time.time() - t0, 1)}
    except Exception as e:   # noqa: BLE001
        return {"error": repr(e), "stderr": r.stderr[-500:]}


def main():
    only = sys.argv[1:] or None
    res = {}

    def run(name, extra, port, child_env=None):
        if only and name not in only:
            return
        res[name] = child(extra, port, child_env)
        print(name, json.dumps(res[name]), flush=True)
    run("alone", [], 29601)
    import torch
    torch.zeros(1, device="cuda")
    torch.cuda.synchronize()
    if os.environ.get("PROBE_PARENT_PLAIN") == "1":
        os.environ["FACPPG_COOP_PLAIN"] = "1"          # the parent's own cooperative kernels as ordinary launches
    # what the suite's earlier tests leave behind: the models, a streamed utterance (priority streams, cooperative launches), a training step
    import contextlib
    import io
    import bench
    dev = torch.device("cuda", 0)
    e = bench.EndToEnd(dev, [200])
    with contextlib.redirect_stdout(io.StringIO()):
        for i in range(3):
            e.step(i)
    torch.cuda.synchronize()
    run("parent_streamed_utterance", [], 29606)
    run("parent_streamed_utterance_child_coop_plain", [], 29607, {"FACPPG_COOP_PLAIN": "1"})
    del e
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    run("parent_models_released", [], 29608)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
