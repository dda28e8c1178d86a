"""ONE launch over the same 4032 buffers, as a batch launch or as a doorbell-fed engine launch (for rocprofv3 --pmc passes:
the per-buffer difference of every counter is what the doorbell form costs).
    python3 tools/engine_vs_batch.py batch|engine [channels] [buffers]"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import gpuaudiobench_amd as gab
mode = sys.argv[1]
T, B, L, R = (int(sys.argv[2]) if len(sys.argv) > 2 else 1024), 512, 4096, 64
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4032
ir = torch.from_numpy(gab.harness.conv_accel_ir(L, T)).cuda()
x = torch.cat([torch.from_numpy(gab.harness.noise(T * B, seed=42 + i)) for i in range(R)]).cuda()
warm = gab.ConvPlan(T, B, L, scheme="split")
warm.set_ir(ir)
y = torch.empty_like(x)
for _ in range(300):                                  # the clocks settle (a 64-buffer launch: told apart from the measured one by its size)
    warm.process_batch(x, R, out=y)
torch.cuda.synchronize()
plan = gab.ConvPlan(T, B, L, scheme="split")
plan.set_ir(ir)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
if mode == "batch":
    xx = x.repeat(N // R)                             # the engine's ring contents, laid out end to end (8 GB at 1024 channels)
    yy = torch.empty_like(xx)
    torch.cuda.synchronize()
    e0.record()
    plan.process_batch(xx, N, out=yy)
    e1.record()
    torch.cuda.synchronize()
else:
    side = torch.cuda.Stream()
    in_ring, out_ring = plan.engine_rings(R)
    in_ring.copy_(x.view(R, T * B))
    torch.cuda.synchronize()
    e0.record(side)
    plan.engine_start(R, stream=side)
    plan.engine_feed(N, ahead=16)
    plan.engine_stop()
    e1.record(side)
    side.synchronize()
us = e0.elapsed_time(e1) * 1e3 / N
print("%s: %d buffers in one launch, %.3f us per buffer = %.3f of 8 TB/s" % (mode, N, us, 4 * T * (2 * B + 2 * L) / us / 1e3 / 8000), flush=True)
