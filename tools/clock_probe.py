#!/usr/bin/env python3
"""Is it the clock?  Samples the GPU's shader clock (sysfs pp_dpm_sclk / rocm-smi) while the chip is idle, while it runs the
batch-1 acoustic model in a loop, and while it runs the vocoder in a loop -- the check behind the 'decoder heaters'
(tools/idle_gap_probe.py shows WaveGlow.infer slower behind idle time or behind Tacotron2.inference; this asks why)."""
import glob, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd")]
import torch
import bench
from facppg import pipeline

_SCLK_FILE = [None]


def read_sclk():
    """current shader clock of THIS process's GPU (sysfs of its PCI function), MHz"""
    import re
    if _SCLK_FILE[0] is None:
        pr = torch.cuda.get_device_properties(0)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        _SCLK_FILE[0] = "/sys/bus/pci/devices/%s/pp_dpm_sclk" % bdf
        print("clock file:", _SCLK_FILE[0], os.path.exists(_SCLK_FILE[0]))
    out = []
    base = os.path.dirname(_SCLK_FILE[0])
    for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk"):
        try:
            for line in open(os.path.join(base, name)):
                if "*" in line:
                    m = re.search(r"(\d+)\s*Mhz", line, re.I)
                    out.append(25 * round(int(m.group(1)) / 25) if m else -1)
                    break
            else:
                out.append(None)
        except OSError:
            out.append(None)
    return tuple(out)   # (sclk, mclk, fclk, socclk) MHz

dev = torch.device("cuda", 0)
e = bench.EndToEnd(dev, [200])
for i in range(3):
    e.step(i)
x, _ = pipeline.pad_ppgs(e.ppgs, device=dev)
mel = e.tacotron.inference(x, seed=1)[1].contiguous()

def sample(label, fn, seconds=1.5):
    stop, seen = [False], []
    def loop():
        while not stop[0]:
            seen.append(read_sclk()); time.sleep(0.02)
    th = threading.Thread(target=loop); th.start()
    t0 = time.time()
    while time.time() - t0 < seconds:
        fn()
    torch.cuda.synchronize(); stop[0] = True; th.join()
    from collections import Counter
    print("%-34s %s" % (label, Counter(seen).most_common(4)), flush=True)

print("source sample:", read_sclk())
sample("idle", lambda: time.sleep(0.05))
sample("Tacotron2.inference loop (batch 1)", lambda: e.tacotron.inference(x, seed=1))
sample("WaveGlow.infer loop (batch 1)", lambda: (e.waveglow.infer(mel, sigma=0.6, seed=1), torch.cuda.synchronize()))
sample("end-to-end step loop", lambda: e.step(5))
