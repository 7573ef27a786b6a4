#!/usr/bin/env python3
"""Experiment: WaveGlow.infer of one batch as TWO half-batches on two HIP streams, so that the tail round of one half's layer
launch is filled by the other half's tiles: python tools/two_stream_wg.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd")]
import numpy as np
import torch
from facppg import synth
from waveglow.glow import WaveGlow

hop = 256
cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=hop)
sd = synth.waveglow_state_dict(cfg)


def model():
    m = WaveGlow.remove_weightnorm(WaveGlow(**cfg))
    m.load_state_dict(sd)
    return m.cuda().eval()


m0 = model()
ms = [model() for _ in range(4)]
ss = [torch.cuda.Stream() for _ in range(4)]


def timeit(fn, n=5):
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[n // 2] * 1e3


def run(name, mel, lengths):
    B = mel.shape[0]
    order = sorted(range(B), key=lambda i: -(lengths[i] if lengths else 0))

    def single():
        return m0.infer(mel, sigma=0.6, seed=1, lengths=lengths)

    def groups(G):
        def f():
            cur = torch.cuda.current_stream()
            outs = []
            for gi in range(G):
                idx = order[gi::G]
                st, mm = ss[gi], ms[gi]
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    la = [lengths[i] for i in idx] if lengths else None
                    Tm = max(la) if la else mel.shape[2]
                    outs.append(mm.infer(mel[idx][:, :, :Tm].contiguous(), sigma=0.6, seed=1, lengths=la))
            for gi in range(G):
                cur.wait_stream(ss[gi])
            return outs
        return f

    t1 = timeit(single)
    res = ["%s: one call %.2f ms" % (name, t1)]
    for G in (2, 3, 4):
        if B >= G:
            t = timeit(groups(G))
            res.append("%d streams %.2f ms (%+.1f %%)" % (G, t, 100 * (t1 - t) / t1))
    print(" | ".join(res))


mel8 = synth.synthetic_mel(8, 1000, seed=1234).cuda()
run("B=8 x 1000 uniform", mel8, None)
g = np.random.Generator(np.random.PCG64(7))
lens = (100 + g.integers(0, 301, size=16)).tolist()
mel16 = synth.synthetic_mel(16, max(lens), seed=5).cuda()
run("B=16 ragged 100..400", mel16, lens)
mel64 = synth.synthetic_mel(64, 400, seed=6).cuda()
lens64 = (100 + g.integers(0, 301, size=64)).tolist()
run("B=64 ragged", mel64, lens64)
mel4 = synth.synthetic_mel(4, 250, seed=8).cuda()
run("B=4 x 250 uniform", mel4, None)
mel2 = synth.synthetic_mel(2, 300, seed=9).cuda()
run("B=2 x 300 uniform", mel2, None)
