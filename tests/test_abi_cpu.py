"""CPU: the C-ABI library loads and exports every symbol include/facppg.h declares; argument
validation that needs no device works; the product refuses to run without a GPU."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT
from facppg import lib as flib


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "facppg.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(facppg_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = flib.load()
    names = _header_symbols()
    assert len(names) >= 9
    for n in names:
        assert hasattr(L, n), "libfacppg_hip.so does not export %s" % n
    assert sorted(flib.exported_symbols()) == [n for n in names if n in flib.exported_symbols()]
    assert L.facppg_version() == 103


def test_single_hip_runtime():
    flib.load()
    maps = open("/proc/self/maps").read()
    libs = set(l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l)
    assert len(libs) == 1, libs          # must share PyTorch's HIP runtime


def test_config_validation_without_device():
    L = flib.load()
    cfg = flib.WgConfig(80, 160, 12, 8, 4, 2, 8, 256, 3, 1024)
    n = L.facppg_wg_weight_count(cfg)
    from facppg import synth
    sd = synth.waveglow_state_dict()
    # the blob carries every state-dict tensor plus W_inverse next to each convinv W
    assert n == sum(v.numel() for v in sd.values()) + sum(v.numel() for k, v in sd.items() if k.startswith("convinv"))
    bad = flib.WgConfig(80, 160, 12, 8, 4, 2, 8, 128, 3, 1024)
    assert L.facppg_wg_weight_count(bad) == 0
    assert b"n_channels=256" in L.facppg_last_error()
    assert L.facppg_wg_workspace_bytes(None, 1, 1) == 0


def test_training_launch_plan(monkeypatch):
    """Which launches the bf16 training step picks (host-side decision, no device needed): the reference's batch 3 keeps the
    two-launch forward layer, fuses the backward chain at 32 positions per tile and forms the weight gradients on 128 x 128 tiles;
    config 5's batch 12 (segment 10 000 -> 1 250 positions per item) runs every fused launch at 64 positions and the 256 x 256
    weight-gradient tiles; the environment switches force either way (train_waveglow.py:121-134)."""
    L = flib.load()
    for k in ("FACPPG_TRAIN_FUSED_FWD", "FACPPG_TRAIN_FUSED_BWD", "FACPPG_TRAIN_TILE", "FACPPG_WGRAD_TILE"):
        monkeypatch.delenv(k, raising=False)

    def plan(B, Lg=1250, nl=8):
        p = L.facppg_wn_bf16_launch_plan(nl, B, Lg)
        return {"fwd": bool(p & 1), "bwd": bool(p & 2), "wgrad256": bool(p & 4), "tile": p >> 8}
    assert plan(3) == {"fwd": False, "bwd": True, "wgrad256": False, "tile": 32}
    assert plan(6) == {"fwd": True, "bwd": True, "wgrad256": True, "tile": 32}
    assert plan(12) == {"fwd": True, "bwd": True, "wgrad256": True, "tile": 64}
    assert plan(1, Lg=16) == {"fwd": False, "bwd": False, "wgrad256": False, "tile": 32}
    assert plan(12, nl=1)["bwd"] is False                      # a one-layer stack has no (conv_i, gate_{i-1}) pair
    assert L.facppg_wn_bf16_launch_plan(9, 3, 1250) < 0 and L.facppg_wn_bf16_launch_plan(8, 0, 1250) < 0
    monkeypatch.setenv("FACPPG_TRAIN_FUSED_FWD", "0")
    monkeypatch.setenv("FACPPG_TRAIN_FUSED_BWD", "0")
    monkeypatch.setenv("FACPPG_WGRAD_TILE", "128")
    assert plan(12) == {"fwd": False, "bwd": False, "wgrad256": False, "tile": 64}
    monkeypatch.setenv("FACPPG_TRAIN_FUSED_FWD", "1")
    monkeypatch.setenv("FACPPG_TRAIN_TILE", "32")
    monkeypatch.setenv("FACPPG_WGRAD_TILE", "256")
    assert plan(3) == {"fwd": True, "bwd": False, "wgrad256": True, "tile": 32}


def test_no_cpu_fallback():
    from waveglow.glow import WaveGlow
    from facppg import synth
    cfg = dict(synth.WAVEGLOW_CONFIG, n_flows=1)
    m = WaveGlow(**cfg)
    with pytest.raises(flib.FacppgError, match="no CPU path"):
        m.infer(torch.zeros(1, 80, 4))


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under fac-via-ppg_amd/ may import it (only tests/,
    __graft_entry__.smoke() and bench.py's cpu_baseline worker do)."""
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fac-via-ppg_amd")
    pat = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b)", re.M)
    offenders = []
    for d, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py") and pat.search(open(os.path.join(d, f)).read()):
                offenders.append(os.path.join(d, f))
    assert not offenders, offenders


def test_committed_pmc_summary_belongs_to_this_kernel_source():
    """bench.py quotes roofline.traffic from the newest profiles/rNN_pmc.json only if that summary was taken from THIS
    build of csrc/facppg_wg.hip (tools/make_pmc_json.py records the source identity); a kernel edit without a fresh
    tools/profile_bench.sh run must not pass silently."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod_pmc", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    traffic, src = b.pmc_traffic()
    assert traffic is not None and traffic > 0, src
