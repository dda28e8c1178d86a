"""Times the FDTD3D plan: grid n^3, `samples` audio samples (3 leapfrog steps each)."""
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
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
plan.process(x, out, T, samples, 0, samples)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
steps = samples * 3
cells = n ** 3 + 3 * (n + 1) * n * n
alg = 2 * 4 * cells
print("grid %d^3 T=%d: %d steps in %.3f ms -> %.2f us/step, %.0f GB/s algorithmic (%.1f%% of 8 TB/s)"
      % (n, T, steps, ms, ms * 1e3 / steps, alg / (ms * 1e-3 / steps) / 1e9, alg / (ms * 1e-3 / steps) / 8e12 * 100))
