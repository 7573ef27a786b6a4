"""The fp32 WaveGlow training step (batch 3, segment 10000), a few steps: for rocprofv3 --kernel-trace --stats."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd")]
import torch
import bench
dev = torch.device("cuda", 0)
m, crit = bench.make_train_model(dev)
m.train_precision = "fp32"
from waveglow.optim import Adam
opt = Adam(m.parameters(), lr=1e-5)
mel, audio = bench.train_batch(dev, 3)
for i in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.zero_grad(); loss = crit(m((mel, audio))); loss.backward(); opt.step()
    torch.cuda.synchronize(); print("step %d: %.2f ms, loss %.4f" % (i, (time.perf_counter() - t0) * 1e3, float(loss)), flush=True)
