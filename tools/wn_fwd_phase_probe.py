#!/usr/bin/env python3
"""Per-phase times of the fused WN layer kernels of the bf16 training step (k_wn_fwd, k_wn_bwd) from in-kernel wall_clock64
stamps of wave 0 of every tile (FACPPG_WN_FWD_STAMPS / FACPPG_WN_BWD_STAMPS; the stamped builds run one launch at a time):
python tools/wn_fwd_phase_probe.py [B ...]   (segment 10 000 -> L = 1250 positions per item; medians over the tiles, us)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd")]
import numpy as np
import torch
from facppg import synth
from waveglow.glow import WN

path = "/tmp/wn_stamps.txt"
cfg = dict(synth.WAVEGLOW_CONFIG)
sd = synth.waveglow_state_dict(cfg)
os.environ["FACPPG_TRAIN_FUSED_FWD"] = "1"
os.environ["FACPPG_TRAIN_FUSED_BWD"] = "1"
for B in [int(a) for a in sys.argv[1:]] or [3, 12]:
    wn = WN(4, 640, **cfg["WN_config"])
    torch.nn.utils.remove_weight_norm(wn.start)
    for lst in (wn.in_layers, wn.cond_layers, wn.res_skip_layers):
        for conv in lst:
            torch.nn.utils.remove_weight_norm(conv)
    wn.load_state_dict({k[len("WN.0."):]: v for k, v in sd.items() if k.startswith("WN.0.")}, strict=True)
    wn.train_precision = "bf16"
    wn = wn.cuda()
    a = torch.randn(B, 4, 1250, device="cuda").requires_grad_(True)
    s = torch.randn(B, 640, 1250, device="cuda").requires_grad_(True)
    for _ in range(2):
        wn((a, s)).sum().backward()
    torch.cuda.synchronize()
    if os.path.exists(path):
        os.remove(path)
    os.environ["FACPPG_WN_FWD_STAMPS"] = path
    os.environ["FACPPG_WN_BWD_STAMPS"] = path
    wn((a, s)).sum().backward()
    torch.cuda.synchronize()
    del os.environ["FACPPG_WN_FWD_STAMPS"], os.environ["FACPPG_WN_BWD_STAMPS"]
    launches, cur = [], None
    for line in open(path):
        if line.startswith("launch"):
            cur = []
            launches.append((line.strip(), cur))
        else:
            cur.append([int(x) for x in line.split()])
    for hdr, rows in launches:
        r = np.array(rows, dtype=np.float64) * 0.01     # us
        r = r[r[:, 1] > 0]
        if " fwd " in hdr:
            med = np.median(np.diff(r[:, :16], axis=1), axis=0)
            print("B=%d %s: chunks %s | gate -> LDS %.2f | gemm2 + copy-out %.2f | fp32 tile + row stores %.2f | drain %.2f | total med %.2f max %.2f" % (
                B, hdr, " ".join("%.2f" % x for x in med[:11]), med[11], med[12], med[13], med[14], np.median(r[:, 15]), r[:, 15].max()))
        else:
            med = np.median(np.diff(r[:, :18], axis=1), axis=0)
            print("B=%d %s: prologue %.2f | chunks %s | dh epilogue %.2f | gemm2 %.2f | gate' %.2f | row stores %.2f | total med %.2f max %.2f" % (
                B, hdr, med[0], " ".join("%.2f" % x for x in med[1:13]), med[13], med[14], med[15], med[16], np.median(r[:, 17]), r[:, 17].max()))
