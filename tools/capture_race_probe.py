"""How often does c10d's watchdog trip over the capture of the data-parallel step?  (RCCL, world 1.)  Repeats
[eager hooked step -> capture with the collectives inside] N times in one process; a trip aborts the process, so the
count reached is printed as it goes.  FACPPG_CAPTURE_SETTLE_MS sets the settle time before each capture."""
import os, sys, time
import numpy as np, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fac-via-ppg_amd"))
from facppg import synth
from waveglow.glow import WaveGlow, WaveGlowLoss
from waveglow.graphed import GraphedTrainStep
from waveglow.distributed import GradientExchange
from waveglow.optim import Adam
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29631")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
cfg = dict(synth.WAVEGLOW_CONFIG, n_flows=int(os.environ.get("FLOWS", "12")))
g = np.random.Generator(np.random.PCG64(5))
wav = torch.from_numpy((0.1 * g.standard_normal((2, 4000))).astype(np.float32)).cuda()
mel = synth.synthetic_mel(2, 4000 // 160 + 1, seed=10).cuda()
crit = WaveGlowLoss(0.7071)
m = WaveGlow(**cfg).cuda().train(); m.train_precision = "bf16"
ex = GradientExchange(m, n_buckets=3, grad_dtype=torch.bfloat16)
opt = Adam(m.parameters(), lr=1e-5)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for it in range(N):
    step = GraphedTrainStep(m, crit, opt, warmup=1, exchange=ex)
    for _ in range(3):
        loss = step(mel, wav)
    assert step.graph_holds_step, "capture fell back"
    print("capture %d ok, loss %.4f" % (it, float(loss)), flush=True)
    del step
    time.sleep(0.013 * (it % 9))         # walk the phase against the watchdog's 100 ms period
print("ALL %d CAPTURES OK" % N, flush=True)
dist.destroy_process_group()
