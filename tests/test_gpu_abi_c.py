"""The C ABI from a plain C program (no Python, no torch): tests/abi/abi_smoke.c is compiled against
include/facppg.h as C99, linked with libfacppg_hip.so and the HIP runtime, and run on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "abi", "abi_smoke.c")
LIBDIR = os.path.join(ROOT, "fac-via-ppg_amd", "csrc")


def _compile(out):
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", SRC, "-o", out,
           "-L" + LIBDIR, "-lfacppg_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"]
    return subprocess.run(cmd, capture_output=True, text=True)


def test_header_and_client_compile_as_c99(tmp_path):
    """CPU: the header is valid C99 and a C client links against the library (no GPU needed to build)."""
    r = _compile(str(tmp_path / "abi_smoke"))
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.gpu
def test_c_client_runs_waveglow_infer(tmp_path):
    exe = str(tmp_path / "abi_smoke")
    r = _compile(exe)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "abi_smoke ok" in r.stdout
