"""gab_conv_engine_*: the resident launch fed one buffer per ring of the doorbell (4096 taps x 512-sample buffers).
Device time per buffer (HIP events on the engine's stream around the whole launch) for several `ahead` limits, and the
output ring checked bit for bit against batch launches over the same buffers.
    python tools/engine_conv.py [channels] [passes over the 64-slot ring]"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import gpuaudiobench_amd as gab
T, B, L, R = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), 512, 4096, 64
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 63
alg = 4 * T * (2 * B + 2 * L)
ir = torch.from_numpy(gab.harness.conv_accel_ir(L, T)).cuda()
x = torch.cat([torch.from_numpy(gab.harness.noise(T * B, seed=42 + i)) for i in range(R)]).cuda()
ref = gab.ConvPlan(T, B, L, scheme="split")
ref.set_ir(ir)
for _ in range(passes):
    y = ref.process_batch(x, R)
torch.cuda.synchronize()
y_ref = y.clone()
side = torch.cuda.Stream()
import os
for ahead in ([1000] if os.environ.get('GAB_ENGINE_ONLY_PREPUBLISHED') else
              [int(a) for a in os.environ['ENGINE_AHEADS'].split()] if os.environ.get('ENGINE_AHEADS') else [1000, 48, 32, 24, 16, 12, 8, 6]):
    plan = gab.ConvPlan(T, B, L, scheme="split")
    plan.set_ir(ir)
    in_ring, out_ring = plan.engine_rings(R)
    in_ring.copy_(x.view(R, T * B))
    torch.cuda.synchronize()
    for _ in range(500):                              # the part's clocks settle over the first ~40 ms of load
        ref.process_batch(x, R, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(side)
    plan.engine_start(R, stream=side)
    if ahead >= 1000:
        plan.engine_publish(passes * R)              # everything at once: the launch's own pace, no doorbell traffic to wait for
    else:
        plan.engine_feed(passes * R, ahead=ahead)
    plan.engine_stop()
    e1.record(side)
    side.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (passes * R)
    same = bool(torch.equal(out_ring.reshape(-1).view(torch.int32), y_ref.view(torch.int32)))
    print("ahead %2d: %d buffers, one per ring of the doorbell: %.3f us per buffer = %.0f GB/s algorithmic = %.3f of 8 TB/s; output ring bit-identical to batch launches: %s"
          % (ahead, passes * R, us, alg / us / 1e3, alg / us / 1e3 / 8000, same), flush=True)
    plan.close()
