"""CPU: bench.py's own N-rank launcher (--gpus N without torch.distributed.run): it must spawn N ranks, rendezvous on
127.0.0.1, count N connected ranks and print one JSON line from rank 0 with n_gpus == N.  --launch-check runs that
path without the GPU work (gloo)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=300)


@pytest.mark.parametrize("n", [2, 4])
def test_bench_spawns_n_ranks(n):
    r = _run(["--gpus", str(n), "--launch-check", "--dist-backend", "gloo"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]          # (gloo itself prints a connection notice)
    assert len(lines) == 1                                   # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["world_size"] == n and d["ranks_connected"] == n and d["max_over_ranks"] == float(n)


def test_bench_refuses_world_size_mismatch():
    """Launched as ONE process with WORLD_SIZE=1 but --gpus 2 (what silently produced an n_gpus:1 line before):
    must fail loudly, not report a 1-rank number as a 2-GPU one."""
    r = _run(["--gpus", "2", "--launch-check", "--dist-backend", "gloo"], env={"WORLD_SIZE": "1", "RANK": "0", "MASTER_PORT": "29999"})
    assert r.returncode != 0 and "WORLD_SIZE 1 != --gpus 2" in r.stderr


def test_bench_corpus_workload_data_path_world2():
    """--workload corpus (BASELINE config 4) at world 2 on gloo with a CPU stand-in synthesiser: the corpus is sharded by
    facppg.shard.partition, every rank runs script.synthesize_corpus.synthesize_shard, and rank 0 gets every utterance
    back through the all_gather + padded gather with N_i = Tin_i * hop."""
    r = _run(["--gpus", "2", "--launch-check", "--dist-backend", "gloo", "--workload", "corpus", "--utterances", "37"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["workload"] == "corpus" and d["utterances"] == 37 and d["samples"] == d["expected_samples"] and d["lengths_ok"]


def test_bench_train_workload_exchange_world2():
    """--workload train (BASELINE config 5) at world 2 on gloo: flat parameter broadcast + the bucketed GradientExchange
    leave the mean of the ranks' gradients on every rank."""
    r = _run(["--gpus", "2", "--launch-check", "--dist-backend", "gloo", "--workload", "train"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["workload"] == "train" and d["exchange_ok"] is True and d["exchange_bytes"] > 0


def test_metric_is_baselines_own():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.METRIC == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert len(b.kernel_source_id()) == 16
