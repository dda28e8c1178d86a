"""Runs the streaming conv kernel N times (for rocprofv3 / quick timing)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import gpuaudiobench_amd as gab
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
T, B, L = (int(sys.argv[2]) if len(sys.argv) > 2 else 1024), 512, 4096
plan = gab.ConvPlan(T, B, L)
plan.set_ir(torch.from_numpy(gab.harness.conv_accel_ir(L, T)).cuda())
xs = [torch.from_numpy(gab.harness.noise(T * B, seed=s)).cuda() for s in range(8)]
out = torch.empty(T * B, device="cuda")
for i in range(50):
    plan.process(xs[i % 8], out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(n):
    plan.process(xs[i % 8], out=out)
e1.record()
torch.cuda.synchronize()
print("T=%d us_per_launch %.3f" % (T, e0.elapsed_time(e1) * 1e3 / n))
