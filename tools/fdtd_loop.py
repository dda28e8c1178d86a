"""Times the FDTD3D plan: grid n^3, `samples` audio samples (3 leapfrog steps each)."""
import os
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import gpuaudiobench_amd as gab
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
samples = int(sys.argv[2]) if len(sys.argv) > 2 else 334
T = int(sys.argv[3]) if len(sys.argv) > 3 else 128
P = gab.fdtd_default_params(n)
plan = gab.FdtdPlan(P)
x = torch.from_numpy(gab.harness.noise(T * samples, seed=1)).cuda()
out = torch.zeros(T * samples, device="cuda")
plan.process(x, out, T, samples, 0, min(samples, 8))
torch.cuda.synchronize()
plan.reset()
import time
steps = samples * 3
cells = n ** 3 + 3 * (n + 1) * n * n
alg = 2 * 4 * cells
for rep in range(4):          # rep 0 pays the hipGraph capture for this signature
    plan.reset()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    plan.process(x, out, T, samples, 0, samples)
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    ms = e0.elapsed_time(e1)
    print("rep %d grid %d^3 T=%d: %d steps in %.3f ms device (%.3f ms wall) -> %.2f us/step, %.0f GB/s algorithmic (%.1f%% of 8 TB/s)"
          % (rep, n, T, steps, ms, wall, ms * 1e3 / steps, alg / (ms * 1e-3 / steps) / 1e9,
             alg / (ms * 1e-3 / steps) / 8e12 * 100))
if os.environ.get("GAB_FDTD_RES_ABLATE", "0") != "0" and hasattr(gab.lib, "gab_debug_fdtd_rounds"):
    import ctypes
    r = (ctypes.c_ulonglong * 4)()
    gab.lib.gab_debug_fdtd_rounds(r)
    if r[1]:
        print("resident kernel polls (GAB_FDTD_RES_ABLATE=4; one lane per wave of one workgroup): %.2f extra rounds per poll, %d polls"
              % (r[0] / r[1], r[1]))

if os.environ.get("GAB_FDTD_RES_ABLATE", "0") != "0" and hasattr(gab.lib, "gab_debug_fdtd_phases"):
    import ctypes
    ph = (ctypes.c_ulonglong * 128)()
    gab.lib.gab_debug_fdtd_phases(ph)
    total_steps = steps * 4 + min(samples, 8) * 3          # every launch of this run added to the sums
    if any(ph):
        print("resident kernel, one workgroup, clocks from a step's start to: low faces done | barrier A passed | interior pressures done | quads there | face rows + stores done | barrier B passed")
        for w in range(16):
            print("  wave %2d: " % w + "  ".join("%6.0f" % (ph[w * 8 + i] / total_steps) for i in range(6)))
