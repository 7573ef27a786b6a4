#!/usr/bin/env python3
"""Whole-step HIP graph capture of the bf16 WaveGlow training step (forward + loss + backward + fused Adam):
python tools/graph_train.py [B ...].  Prints eager vs replayed ms/step and checks the two give the same loss trajectory."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
from facppg import synth
from test_gpu_e2e import weightnorm_state_dict
from waveglow.glow import WaveGlow, WaveGlowLoss, reserve_pinned


def build(prec):
    cfg = dict(synth.WAVEGLOW_CONFIG)
    m = WaveGlow(**cfg)
    m.load_state_dict(weightnorm_state_dict(synth.waveglow_state_dict(cfg)), strict=True)
    m = m.cuda().train()
    m.train_precision = prec
    return m


def main():
    prec = "bf16"
    for B in ([int(a) for a in sys.argv[1:] if a.isdigit()] or (3, 12)):
        g = np.random.Generator(np.random.PCG64(1))
        audio = torch.from_numpy(np.clip(g.standard_normal((B, 10000), dtype=np.float32) * 0.1, -1, 1)).cuda()
        mel = synth.synthetic_mel(B, 63, seed=1).cuda()
        crit = WaveGlowLoss(0.7071)
        # eager
        m = build(prec)
        opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=True)
        losses_e, ts = [], []
        for i in range(8):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m.zero_grad()
            loss = crit(m((mel, audio)))
            loss.backward()
            opt.step()
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            losses_e.append(float(loss))
        t_eager = min(ts[2:])
        # graph
        m = build(prec)
        opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=True, capturable=True)
        s_mel, s_audio = mel.clone(), audio.clone()
        losses_g = []
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(3):
                opt.zero_grad(set_to_none=True)
                loss = crit(m((s_mel, s_audio)))
                loss.backward()
                opt.step()
                losses_g.append(float(loss))
        torch.cuda.current_stream().wait_stream(side)
        reserve_pinned()
        graph = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(graph):
            s_loss = crit(m((s_mel, s_audio)))
            s_loss.backward()
            opt.step()
        ts = []
        for i in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            graph.replay()
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            losses_g.append(float(s_loss))
        t_graph = min(ts[1:])
        print("bf16 B=%d: eager %.2f ms/step, graph replay %.2f ms/step" % (B, t_eager * 1e3, t_graph * 1e3))
        print("  eager losses", ["%.5f" % x for x in losses_e])
        print("  graph losses", ["%.5f" % x for x in losses_g], "(3 eager warm-up steps, then the captured step is NOT applied, then 5 replays)")


if __name__ == "__main__":
    main()
