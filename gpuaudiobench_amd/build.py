"""Builds libgab_hip.so (HIP kernels + C ABI + C++ harness) for gfx950, in-tree.

    python gpuaudiobench_amd/build.py [--force]   (run as a script: the package import needs the .so)

hipcc cross-compiles without a GPU.  The shared object stays next to this file
so it travels with the source tree to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.environ.get("GAB_CSRC") or os.path.join(HERE, "csrc")      # (GAB_CSRC: an A/B build from another copy of the sources)
# GAB_BUILD_TAG=<tag>: a diagnostic build beside the product one (libgab_hip_<tag>.so, own object
# directory); load it with GAB_LIB_PATH.  The product library is always libgab_hip.so.
_TAG = os.environ.get("GAB_BUILD_TAG", "")
OBJ = os.path.join(HERE, "_build" + ("_" + _TAG if _TAG else ""))
LIB = os.path.join(HERE, "libgab_hip%s.so" % ("_" + _TAG if _TAG else ""))
DRIVER = os.path.join(HERE, "gpubench" + ("_" + _TAG if _TAG else ""))

ARCH = "gfx950"
# -ffp-contract=off: a*b+c is never fused behind our back; kernels that want an
# FMA say fmaf().  That is what keeps gain/iir/conv1d/dwg/fdtd bit-reproducible.
FLAGS = ["-O3", "-std=c++17", "--offload-arch=" + ARCH, "-ffp-contract=off", "-fPIC",
         "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]

if os.environ.get("GAB_EXTRA_FLAGS"):  # experiments, e.g. "-mllvm -amdgpu-sched-strategy=max-ilp"
    FLAGS += os.environ["GAB_EXTRA_FLAGS"].split()

if os.environ.get("GAB_ABLATE"):      # diagnostic build: stage-ablation variants of the conv kernel
    FLAGS.append("-DGAB_ABLATE")

DRIVER_MAIN = "gpubench_main.cpp"


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def sources():
    return sorted(f for f in os.listdir(CSRC)
                  if f.endswith((".hip", ".cpp")) and f != DRIVER_MAIN)


def _deps_mtime():
    m = 0.0
    for d in (CSRC, os.path.join(ROOT, "include"), os.path.join(ROOT, "include", "gab")):
        if os.path.isdir(d):
            for f in os.listdir(d):
                if f.endswith((".h", ".hpp")):
                    m = max(m, os.path.getmtime(os.path.join(d, f)))
    return m


def _compile(src, force, hdr_m):
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    spath = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj)
            and os.path.getmtime(obj) >= max(os.path.getmtime(spath), hdr_m)):
        return obj
    cmd = [_hipcc()] + FLAGS + ["-x", "hip", "-c", spath, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force=False, driver=True, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hdr_m = _deps_mtime()
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, hdr_m), srcs))
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        cmd = [_hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs
        subprocess.check_call(cmd)
        if verbose:
            print("linked", LIB)
    main_src = os.path.join(CSRC, DRIVER_MAIN)
    if driver and os.path.exists(main_src):
        if (force or not os.path.exists(DRIVER)
                or os.path.getmtime(DRIVER) < max(os.path.getmtime(LIB), os.path.getmtime(main_src), hdr_m)):
            cmd = [_hipcc()] + FLAGS + ["-x", "hip", main_src, "-o", DRIVER,
                                        "-L" + HERE, "-l:" + os.path.basename(LIB), "-Wl,-rpath,$ORIGIN"]
            subprocess.check_call(cmd)
            if verbose:
                print("linked", DRIVER)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
