#!/usr/bin/env python3
"""A few end-to-end PPG -> wav syntheses for profiling: python tools/e2e_once.py <batch> <frames> [hop]"""
import contextlib
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd")]
import numpy as np
import torch
from common.hparams import create_hparams_stage
from common.layers import TacotronSTFT
from facppg import pipeline, synth
from script.train_ppg2mel import load_model
from waveglow.denoiser import Denoiser
from waveglow.glow import WaveGlow

B, T = int(sys.argv[1]), int(sys.argv[2])
hop = int(sys.argv[3]) if len(sys.argv) > 3 else 256
cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=hop)
wg = WaveGlow.remove_weightnorm(WaveGlow(**cfg))
wg.load_state_dict(synth.waveglow_state_dict(cfg))
wg = wg.cuda().eval()
den = Denoiser(wg, hop_length=hop, mode="zeros")
g = np.random.Generator(np.random.PCG64(7))
lens = [T] if B == 1 else (100 + g.integers(0, max(1, T - 99), size=B)).tolist()
hp = create_hparams_stage(max_decoder_steps=max(lens))
taco = load_model(hp)
taco.load_state_dict(synth.tacotron_state_dict(hp, gate_bias=-10.0))
taco.eval()
ppgs = [synth.synthetic_ppg(n, 5816, seed=i) for i, n in enumerate(lens)]
stft = TacotronSTFT(1024, hop, 1024, 80, 22050 if hop == 256 else 16000, 0.0, 8000.0).cuda()
with contextlib.redirect_stdout(io.StringIO()):
    for i in range(4):
        wavs, tout = pipeline.synthesize(ppgs, taco, wg, den, sigma=0.6, strength=0.005, seed=i, return_device=True,
                                         step_limits=lens if B > 1 else None)
        mel = stft.mel_spectrogram(torch.clamp(wavs[0][None], -1, 1))      # the analysis direction (a11) on the synthesised audio
torch.cuda.synchronize()
print("done", B, lens[:4], tout[:4], tuple(mel.shape), file=sys.stderr)
