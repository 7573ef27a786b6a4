"""Phase profile of k_wn_layer8 at batch 1 (library built with -DFACPPG_WN8_PROF): python tools/prof_wn8.py [T]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd")]
import torch
from facppg import synth, lib as flib
from waveglow.glow import WaveGlow
T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=256)
m = WaveGlow.remove_weightnorm(WaveGlow(**cfg)); m.load_state_dict(synth.waveglow_state_dict(cfg)); m = m.cuda().eval()
mel = synth.synthetic_mel(1, T).cuda()
L = flib.load()
out = (ctypes.c_ulonglong * 12)()
m.infer(mel, sigma=0.6, seed=0); torch.cuda.synchronize()
L.facppg_debug_wn8_prof(out, 1)
for i in range(3):
    m.infer(mel, sigma=0.6, seed=i)
torch.cuda.synchronize()
L.facppg_debug_wn8_prof(out, 1)
n = max(1, out[6])
names = ["prologue", "K loop", "gate", "second GEMM", "end rows", "epilogue"]
tot = sum(out[i] for i in range(6))
print("k_wn_layer8 workgroup 0, %d launches, clock64 ticks per launch (100 MHz ticks x 10 ns if the counter is the constant clock):" % n)
for i, nm in enumerate(names):
    print("  %-12s %9.1f ticks  %5.1f %%" % (nm, out[i] / n, 100.0 * out[i] / tot))
print("  total        %9.1f ticks" % (tot / n))
print("  (k_wn_flow8: of the K loop, waiting for the previous layer %9.1f ticks; hand-off after the epilogue %9.1f ticks)" % (out[7] / n, out[8] / n))
