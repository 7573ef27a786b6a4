"""Kernel-level look at a single short utterance through WaveGlow (B=1, T=200, hop 256)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd")]
import torch
from facppg import synth
from waveglow.glow import WaveGlow
cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=256)
m = WaveGlow.remove_weightnorm(WaveGlow(**cfg)); m.load_state_dict(synth.waveglow_state_dict(cfg)); m = m.cuda().eval()
mel = synth.synthetic_mel(1, int(sys.argv[1]) if len(sys.argv) > 1 else 200).cuda()
for i in range(4):
    torch.cuda.synchronize(); t = time.perf_counter(); m.infer(mel, sigma=0.6, seed=i); torch.cuda.synchronize()
    print("infer %.2f ms" % ((time.perf_counter() - t) * 1e3))
