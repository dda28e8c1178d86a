"""Diagnostic builds only (GAB_BUILD_TAG=ablate GAB_ABLATE=1 python gpuaudiobench_amd/build.py, then
GAB_LIB_PATH=.../libgab_hip_ablate.so GAB_FDTD_RES_ABLATE=2 python tools/fdtd_timeout_check.py):
workgroup 0 of the LDS-resident FDTD kernel never publishes its boundary pressures, so its neighbours'
bounded polls must give up, the launch must END, gab_fdtd_status must report GAB_ERR_RUNTIME for THAT call and the
calls after that must run on the step kernels and match the oracle again; the failing call's output is NaN and
gab_fdtd_status reports it at that call."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gpuaudiobench_amd as gab  # noqa: E402
import oracle as orc  # noqa: E402

assert os.environ.get("GAB_FDTD_RES_ABLATE") == "2", "run with GAB_FDTD_RES_ABLATE=2 on the ablate library"
n, T, B = 32, 4, 8
G = gab.fdtd_default_params(n)
plan = gab.FdtdPlan(G)
assert plan.resident()[0]
x = orc.Rand(1).bipolar(T * B)
out = torch.zeros(T * B, device="cuda")
t0 = time.time()
plan.process(torch.from_numpy(x).cuda(), out, T, B, 0, B)
try:
    plan.status()                                  # synchronises: the error belongs to THIS call
    raise SystemExit("FAIL: gab_fdtd_status did not report the launch that gave up")
except gab.GabError as e:
    assert e.code == -2, e.code
    print("launch with a silent workgroup ended after %.3f s; gab_fdtd_status reported: %s" % (time.time() - t0, str(e)[:140]))
got = out.cpu().numpy()
assert np.isnan(got).all(), "the output of a launch that gave up must be NaN in every sample"
print("that call's output is NaN in all %d samples" % got.size)
try:
    plan.status()
except gab.GabError:
    raise SystemExit("FAIL: the error was reported twice")
assert not plan.resident()[0]
plan.reset()
P = orc.fdtd_params(n)
grids = orc.fdtd_grids(P)
ref = np.zeros(T * B, np.float32)
orc.fdtd(P, grids, x, ref, T, B, 0, B, fused=True)
plan.process(torch.from_numpy(x).cuda(), out, T, B, 0, B)
assert np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32))
print("after the report the plan runs the step kernels and matches the oracle bit for bit: ok")
