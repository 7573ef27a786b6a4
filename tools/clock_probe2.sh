#!/bin/bash
# second look at "why does the vocoder start slow behind low activity": the SMU's own averaged clocks / activity (amd-smi / rocm-smi),
# sampled while the acoustic model loops and while the vocoder loops
which amd-smi rocm-smi 2>&1
python - <<'PY' &
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd")]
import torch, bench
from facppg import pipeline
dev = torch.device("cuda", 0)
e = bench.EndToEnd(dev, [200])
for i in range(3): e.step(i)
x, _ = pipeline.pad_ppgs(e.ppgs, device=dev)
mel = e.tacotron.inference(x, seed=1)[1].contiguous()
open("/tmp/phase", "w").write("tacotron")
t0 = time.time()
while time.time() - t0 < 6: e.tacotron.inference(x, seed=1)
open("/tmp/phase", "w").write("waveglow")
t0 = time.time()
while time.time() - t0 < 6: e.waveglow.infer(mel, sigma=0.6, seed=1); torch.cuda.synchronize()
open("/tmp/phase", "w").write("done")
PY
sleep 1
while [ "$(cat /tmp/phase 2>/dev/null)" != "done" ]; do
  ph=$(cat /tmp/phase 2>/dev/null)
  if [ -n "$ph" ]; then
    echo "== $ph"
    (amd-smi metric --clock --usage 2>/dev/null || rocm-smi --showclocks --showuse 2>/dev/null) | grep -i "gfx\|sclk\|busy\|use\|clk" | grep -v "N/A" | sort | uniq -c | sort -rn | head -12
  fi
  sleep 1.5
done
wait
