#!/usr/bin/env python3
"""Stage timings of the end-to-end PPG->wav path on one GPU (BASELINE configs 1 and 3)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd")]
import numpy as np
import torch
from common.hparams import create_hparams_stage
from facppg import pipeline, synth
from script.train_ppg2mel import load_model
from waveglow.denoiser import Denoiser
from waveglow.glow import WaveGlow


def sync_time(fn, n=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def main():
    hop = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    sr = 16000 if hop == 160 else 22050
    cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=hop)
    wg = WaveGlow.remove_weightnorm(WaveGlow(**cfg))
    wg.load_state_dict(synth.waveglow_state_dict(cfg))
    wg = wg.cuda().eval()
    den = Denoiser(wg, hop_length=hop, mode="zeros")
    for B, Tin in ((1, 200), (1, 1000), (16, 250)):
        hp = create_hparams_stage(max_decoder_steps=Tin)
        taco = load_model(hp)
        taco.load_state_dict(synth.tacotron_state_dict(hp, gate_bias=-10.0))
        taco.eval()
        g = np.random.Generator(np.random.PCG64(7))
        lens = [Tin] if B == 1 else (100 + g.integers(0, Tin - 99, size=B)).tolist()
        ppgs = [synth.synthetic_ppg(n, 5816, seed=i) for i, n in enumerate(lens)]
        x, _ = pipeline.pad_ppgs(ppgs)
        x = x.cuda()
        ln = lens if B > 1 else None
        t_taco = sync_time(lambda: taco.inference(x, lengths=ln, seed=1))
        mel = taco.inference(x, lengths=ln, seed=1)[1].contiguous()
        tout = [int(v) for v in taco.last_output_lengths]
        tl = tout if B > 1 else None
        t_wg = sync_time(lambda: wg.infer(mel, sigma=0.6, lengths=tl, seed=2))
        audio = wg.infer(mel, sigma=0.6, lengths=tl, seed=2)
        t_den = sync_time(lambda: den(audio, strength=0.005, lengths=[t * hop for t in tout] if B > 1 else None))
        t_all = sync_time(lambda: pipeline.synthesize(ppgs, taco, wg, den, seed=3, return_device=True))
        samples = sum(tout) * hop
        print("B=%d Tin<=%d hop=%d: frames=%d samples=%d | tacotron %.2f ms (%.1f us/step) | waveglow %.2f ms | denoiser %.2f ms | "
              "end-to-end %.2f ms -> %.0f samples/s = %.1fx real time @%d Hz" % (
                  B, Tin, hop, sum(tout), samples, t_taco * 1e3, t_taco * 1e6 / max(tout), t_wg * 1e3, t_den * 1e3,
                  t_all * 1e3, samples / t_all, samples / t_all / sr, sr), flush=True)


if __name__ == "__main__":
    main()
