"""Compile-only resource check of every product kernel (no GPU): nothing may spill to scratch, and the headline
kernels keep the occupancy their design assumes.  (Round 5's classic-cut batch kernel sat at 256 registers with
24 bytes of scratch per lane and two modal-bank instantiations spilled: a later change must not bring that back.)
The same numbers by hand: tools/kernel_resources.sh <file.hip>."""
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gpuaudiobench_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def _resources(src):
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-x", "hip", "-c", os.path.join(CSRC, src),
           "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    kernels, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: [^ ]* *Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        for key, pat in (("vgprs", r" VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return kernels


@pytest.fixture(scope="module")
def resources():
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    files = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    with ThreadPoolExecutor(max_workers=4) as ex:
        per_file = list(ex.map(_resources, files))
    out = {}
    for f, k in zip(files, per_file):
        assert k, "no kernels reported for " + f
        out.update({name: dict(v, file=f) for name, v in k.items()})
    return out


def test_no_product_kernel_spills_to_scratch(resources):
    spilling = {k: v for k, v in resources.items() if v.get("scratch", 0) != 0}
    assert not spilling, spilling
    assert len(resources) >= 60          # every .hip file was seen (64 kernels in round 6; round 5's eight-wave launches are diagnostic-only now)


def test_headline_kernels_keep_their_occupancy(resources):
    def one(fragment):
        hits = [v for k, v in resources.items() if fragment in k]
        assert len(hits) == 1, (fragment, [k for k in resources if fragment in k])
        return hits[0]
    for name in ("conv_split_batch12_kernel", "conv_split_engine12_kernel"):   # bench.py's `value` and the doorbell engine: twelve waves, three per SIMD
        r = one(name)
        assert r["occupancy"] >= 3 and r["vgprs"] <= 168 and r["lds"] <= 160 * 1024, (name, r)
    assert one("17conv_batch_kernel")["vgprs"] <= 208                        # the looped body is the single-buffer launch's
    assert one("keep_warm_kernel")["vgprs"] <= 16                            # eight idle waves must fit beside anything but the engine
