"""Source-level drop-in: a translation unit spelled against the REFERENCE's header
names and factory idiom (cuda/main.cu:51-70, 84-100, 117-164) compiles with hipcc against include/
and links with libgab_hip.so.  Compiling needs no GPU; running a benchmark through it does."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "drop_in", "reference_shaped_driver.cpp")
EXE = os.path.join(ROOT, "tests", "drop_in", "_build", "reference_shaped_driver")
LIBDIR = os.path.join(ROOT, "gpuaudiobench_amd")
REFERENCE_HEADERS = ["globals.cuh", "bench_base.cuh", "bench_utils.cuh", "benchmark_constants.cuh", "thread_config.cuh"] + [
    "bench_%s.cuh" % n for n in ("noop", "gain", "gainstats", "datatransfer", "fft", "iir", "conv1d", "conv1d_accel",
                                 "modal", "dwg", "fdtd3d", "rndmem")]


def build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    newest = max(os.path.getmtime(p) for p in [SRC, os.path.join(LIBDIR, "libgab_hip.so")])
    if os.path.exists(EXE) and os.path.getmtime(EXE) >= newest:
        return
    cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"), "-x", "hip",
           SRC, "-o", EXE, "-L" + LIBDIR, "-l:libgab_hip.so", "-Wl,-rpath,$ORIGIN/../../../gpuaudiobench_amd"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_every_reference_header_name_exists_and_forwards():
    for h in REFERENCE_HEADERS:
        text = open(os.path.join(ROOT, "include", h)).read()
        assert '#include "gab/' in text, h


def test_reference_shaped_translation_unit_compiles_and_lists():
    build()
    r = subprocess.run([EXE, "--list"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.split()
    for name in ("NoOp", "gain", "Conv1D_accel", "DWG1DAccel", "FDTD3D", "RndMemRead"):
        assert name in lines
    assert "FS=48000" in lines and "BUFSIZE=512" in lines and "NTRACKS=128" in lines and "NRUNS=100" in lines


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["gain", "Conv1D_accel", "IIRFilter"])
def test_reference_shaped_driver_runs_and_validates(name):
    build()
    r = subprocess.run([EXE, name], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert '"benchmark"' in r.stdout or '"name"' in r.stdout or "{" in r.stdout
