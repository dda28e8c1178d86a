"""Batch mode: n buffers per launch vs one launch per buffer (same results, throughput)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import gpuaudiobench_amd as gab
T, B, L = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), 512, 4096
ir = torch.from_numpy(gab.harness.conv_accel_ir(L, T)).cuda()
N = 64
x = torch.cat([torch.from_numpy(gab.harness.noise(T * B, seed=s)) for s in range(N)]).cuda()
a, b = gab.ConvPlan(T, B, L, scheme="classic"), gab.ConvPlan(T, B, L)   # bit comparison: same cut of the taps
a.set_ir(ir); b.set_ir(ir)
ya = torch.cat([a.process(x[i * T * B:(i + 1) * T * B]).clone() for i in range(N)])
yb = b.process_batch(x, N)
torch.cuda.synchronize()
print("batch == sequential (bits):", bool(torch.equal(ya, yb)))
out = torch.empty_like(x)
for nb in (1, 2, 4, 8, 16, 64):
    reps = max(1, 2048 // nb)
    for _ in range(3):
        b.process_batch(x[:nb * T * B], nb, out=out[:nb * T * B])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.process_batch(x[:nb * T * B], nb, out=out[:nb * T * B])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * nb)
    print("buffers per launch %3d: %.2f us per buffer (%.0f buffers/s, %.0f GB/s algorithmic)"
          % (nb, us, 1e6 / us, 4 * T * (2 * B + 2 * L) / us / 1e3), flush=True)
