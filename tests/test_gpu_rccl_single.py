"""GPU: the RCCL (torch.distributed backend "nccl") code paths of the N > 1 workloads on ONE GPU, with a process group of
world size 1.  No multi-GPU node is available to the builder, and RCCL refuses two ranks on one device, so this cannot
test the exchange's arithmetic across ranks (the gloo world-2 tests do that); it does run every collective the multi-GPU
paths issue -- flat broadcast, bucketed asynchronous all_reduce on the communication stream (fp32 and bf16), all_gather of
lengths, padded gather, barrier, MAX / MIN reductions -- through RCCL itself, with the stream ordering, the HIP-graph
replay in between and the tensor dtypes they use on 8 GPUs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, port):
    e = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist"] + args, capture_output=True, text=True, env=e, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("grad_dtype", ["fp32", "bf16"])
def test_training_workload_through_rccl_world1(grad_dtype):
    d = _bench(["--workload", "train", "--steps", "4", "--grad-dtype", grad_dtype], 29521 if grad_dtype == "fp32" else 29522)
    assert d["collective_backend"].startswith("RCCL") and d["ranks_connected"] == 1
    ex = d["gradient_exchange"]
    assert d["config"]["graph_captured"] and ex["buckets"] == 3 and ex["ms"] > 0
    assert ex["bytes"] == 87879272 * (4 if grad_dtype == "fp32" else 2)           # 351.5 MB fp32 / 175.8 MB bf16
    assert d["ms_per_step"] < 40.0


def test_corpus_and_infer_extras_through_rccl_world1():
    d = _bench(["--workload", "corpus", "--utterances", "48", "--steps", "1", "--warmup", "1"], 29523)
    assert d["collective_backend"].startswith("RCCL") and d["config"]["utterances"] == 48 and d["value"] > 0
    d = _bench(["--steps", "3", "--warmup", "2", "--utterances", "32"], 29524)
    assert d["train_dp"] is not None and d["corpus_dp"] is not None and d["roofline"]["frac"] > 0.3   # the default run is the batch-1 utterance: ~0.5 of the fp32 MFMA peak
