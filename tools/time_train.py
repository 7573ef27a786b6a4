#!/usr/bin/env python3
"""WaveGlow training-step timing on one GPU (BASELINE config 5 shape: segment 10000, batch 3 per GPU)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd")]
import numpy as np
import torch
from facppg import synth
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_e2e import weightnorm_state_dict  # noqa: E402
from waveglow.glow import WaveGlow, WaveGlowLoss
from waveglow.mel2samp import Mel2Samp  # noqa: F401


def main():
    cfg = dict(synth.WAVEGLOW_CONFIG)
    m = WaveGlow(**cfg)
    m.load_state_dict(weightnorm_state_dict(synth.waveglow_state_dict(cfg)), strict=True)
    m = m.cuda().train()
    from waveglow.optim import Adam
    opt = Adam(m.parameters(), lr=1e-5)
    crit = WaveGlowLoss(0.7071)
    from common.layers import TacotronSTFT
    stft = TacotronSTFT(1024, 160, 1024, 80, 16000, 0.0, 8000.0).cuda()
    precisions = [a for a in sys.argv[1:] if a in ("fp32", "bf16")] or ["fp32", "bf16"]
    for prec in precisions:
      m.train_precision = prec
      for B in ([int(a) for a in sys.argv[1:] if a.isdigit()] or (3, 12)):
        g = np.random.Generator(np.random.PCG64(1))
        audio = torch.from_numpy(np.clip(g.standard_normal((B, 10000), dtype=np.float32) * 0.1, -1, 1)).cuda()
        with torch.no_grad():
            mel = stft.mel_spectrogram(audio)
        ts, parts = [], None
        for i in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.zero_grad()
            loss = crit(m((mel, audio)))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            loss.backward()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            opt.step()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            if ts[-1] == min(ts[1:] or ts):
                parts = (t1 - t0, t2 - t1, ts[-1] - (t2 - t0))
        t = min(ts[1:])
        flops = 3 * 20.26e6 * B * 10000      # fwd + bwd ~ 3x the 20.26 MFLOP/sample of the flows
        peak = 2500.0 if prec == "bf16" else 157.3
        print("%s B=%d seg=10000: %.1f ms/step (fwd %.1f, bwd %.1f, Adam %.1f), %.0f samples/s, ~%.1f TFLOP/s (3x fwd FLOPs) = %.1f %% of the "
              "%s MFMA peak, loss %.4f" % (prec, B, t * 1e3, parts[0] * 1e3, parts[1] * 1e3, parts[2] * 1e3, B * 10000 / t,
                                          flops / t / 1e12, 100 * flops / t / 1e12 / peak, prec, float(loss)), flush=True)


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    main()
