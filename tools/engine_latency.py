"""gab_conv_engine_* with ONE buffer in flight: submit(1, flush) -> wait(k + 1), host clock per buffer (the input ring is
resident: what is timed is doorbell -> cold burst -> drain -> completion word), back to back and paced at 512/48000 s;
every output compared bit for bit with one gab_conv_process launch per buffer; then the pipelined rate for comparison.
    python tools/engine_latency.py [channels] [buffers]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import gpuaudiobench_amd as gab
T, B, L, R = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), 512, 4096, 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
alg = 4 * T * (2 * B + 2 * L)
ir = torch.from_numpy(gab.harness.conv_accel_ir(L, T)).cuda()
x = torch.cat([torch.from_numpy(gab.harness.noise(T * B, seed=42 + i)) for i in range(R)]).cuda()
ref = gab.ConvPlan(T, B, L, scheme="split")
ref.set_ir(ir)
want = [ref.process(x[(k % R) * T * B:(k % R + 1) * T * B]).clone() for k in range(2 * R)]   # two laps of the ring
torch.cuda.synchronize()
plan = gab.ConvPlan(T, B, L, scheme="split")
plan.set_ir(ir)
in_ring, out_ring = plan.engine_rings(R)
in_ring.copy_(x.view(R, T * B))
torch.cuda.synchronize()
side = torch.cuda.Stream()
plan.engine_start(R, stream=side)
ts, bad = [], 0
h = torch.empty(T * B).pin_memory()
cur = torch.cuda.current_stream()
for k in range(N):
    t0 = time.perf_counter()
    plan.engine_submit(1, flush=True)
    plan.engine_wait(k + 1, timeout=5.0)
    ts.append((time.perf_counter() - t0) * 1e6)
    if k < 2 * R:                                    # (copy ENGINES read the slot: the launch holds every compute unit)
        h.copy_(out_ring[k % R], non_blocking=True)
        cur.synchronize()
        bad += not torch.equal(h.view(torch.int32), want[k].cpu().view(torch.int32))
ts = np.array(ts[50:])
print("one in flight, back to back: %d buffers  p50 %.1f us  p95 %.1f  min %.1f  max %.1f  (publish -> completed, host clock); first %d outputs bit-identical to one launch per buffer: %s"
      % (len(ts), np.percentile(ts, 50), np.percentile(ts, 95), ts.min(), ts.max(), 2 * R, bad == 0), flush=True)
daw = gab.harness.DawSim(buffer_seconds=float(B) / 48000, mode="spin")
paced = []
k0 = N
for i in range(105):
    daw.wait()
    t0 = time.perf_counter()
    plan.engine_submit(1, flush=True)
    plan.engine_wait(k0 + i + 1, timeout=5.0)
    if i >= 5:
        paced.append((time.perf_counter() - t0) * 1e6)
paced = np.array(paced)
print("one in flight, one buffer per 10.667 ms slot: %d buffers  p50 %.1f us  p95 %.1f  max %.1f" % (len(paced), np.percentile(paced, 50), np.percentile(paced, 95), paced.max()), flush=True)
# pipelined on the same launch: 16 in flight for 4032 buffers
M = 4032 if T >= 1024 else 512
t0 = time.perf_counter()
plan.engine_feed(M, ahead=12)
plan.engine_wait(k0 + 105 + M - 6, timeout=20.0)
t1 = time.perf_counter()
print("pipelined on the same launch (12 in flight): %.2f us per buffer by the host clock = %.3f of 8 TB/s" % ((t1 - t0) * 1e6 / M, alg / ((t1 - t0) * 1e6 / M) / 1e3 / 8000), flush=True)
plan.engine_stop()
plan.close()
