"""What does work on the CUs the batch-1 decoder leaves empty cost the decoder?  (round 5, before building the conditioning
producer that is to run there.)  Decoder stage of one 200-frame utterance alone, then under (a) an HBM-streaming copy,
(b) a rocBLAS fp32 GEMM, (c) both, enqueued on a second stream just before Tacotron2.inference."""
import contextlib
import io
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fac-via-ppg_amd")):
    sys.path.insert(0, p)

from common.hparams import create_hparams_stage  # noqa: E402
from facppg import pipeline, synth  # noqa: E402
from script.train_ppg2mel import load_model  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    T = 200
    hp = create_hparams_stage(max_decoder_steps=T)
    with contextlib.redirect_stdout(io.StringIO()):
        taco = load_model(hp)
    taco.load_state_dict(synth.tacotron_state_dict(hp, gate_bias=-10.0))
    taco.eval()
    ppg = synth.synthetic_ppg(T, 5816, seed=0, alpha=0.002)
    x, _ = pipeline.pad_ppgs([ppg], device=dev)
    side = torch.cuda.Stream(dev)
    big_a = torch.empty(256 << 20, device=dev)      # 1 GiB of floats
    big_b = torch.empty_like(big_a)
    ma = torch.randn(4096, 4096, device=dev)
    mb = torch.randn(4096, 4096, device=dev)

    def load(kind, n):
        with torch.cuda.stream(side):
            for _ in range(n):
                if kind in ("copy", "both"):
                    big_b.copy_(big_a)
                if kind in ("gemm", "both"):
                    torch.mm(ma, mb)

    def run(kind, n):
        res = []
        for it in range(6):
            torch.cuda.synchronize()
            if kind:
                load(kind, n)
            timer = pipeline.StageTimer()
            with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
                taco.inference(x, seed=it, timer=timer)
            st = timer.stages_ms()
            torch.cuda.synchronize()
            res.append(st)
        res = res[2:]
        return {k: sum(r[k] for r in res) / len(res) for k in res[0]}

    for heat in (0, -1):
        pass
        print("heaters", heat, {k: round(v, 3) for k, v in run(None, 0).items()})
    pass
    # size the loads to ~6 ms: time one unit of each alone
    for kind in ("copy", "gemm"):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(side):
            load(kind, 2)
            e0.record()
            load(kind, 4)
            e1.record()
        torch.cuda.synchronize()
        print(kind, "unit ms", e0.elapsed_time(e1) / 4)
    for kind, n in (("copy", 12), ("gemm", 8), ("both", 5)):
        print("under", kind, {k: round(v, 3) for k, v in run(kind, n).items()})


if __name__ == "__main__":
    main()
