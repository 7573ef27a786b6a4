"""WaveGlow.infer latency at a few (B, T) points (hop 256)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd")]
import torch
from facppg import synth
from waveglow.glow import WaveGlow
cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=256)
m = WaveGlow.remove_weightnorm(WaveGlow(**cfg)); m.load_state_dict(synth.waveglow_state_dict(cfg)); m = m.cuda().eval()
pts = [(1, 50), (1, 100), (1, 200), (1, 400), (1, 800), (2, 400), (4, 400), (4, 1000), (8, 1000), (16, 1000)]
if len(sys.argv) > 1:
    pts = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for B, T in pts:
    mel = synth.synthetic_mel(B, T).cuda()
    m.infer(mel, sigma=0.6, seed=0); torch.cuda.synchronize()
    n = 5 if B * T <= 2000 else 2
    t = time.perf_counter()
    for i in range(n):
        m.infer(mel, sigma=0.6, seed=i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / n * 1e3
    print("B=%d T=%d: %.2f ms  %.2f M samples/s  %.0fx RT" % (B, T, ms, B * T * 256 / ms / 1e3, B * T * 256 / 22050 / (ms / 1e3)))
