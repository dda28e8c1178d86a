"""Stateless windowed streaming: parity with the ring-based plan, and throughput with consecutive
buffers alternating between two streams (raw ctypes calls so the host is not the limit)."""
import sys, time
import ctypes as C
import numpy as np, torch
sys.path.insert(0, ".")
import gpuaudiobench_amd as gab
T, B, L = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), 512, 4096
NBUF = 16
ir = torch.from_numpy(gab.harness.conv_accel_ir(L, T)).cuda()
xs = [torch.from_numpy(gab.harness.noise(T * B, seed=s)).cuda() for s in range(NBUF)]
zeros = torch.zeros(T * B, device="cuda")
a, b = gab.ConvPlan(T, B, L, scheme="classic"), gab.ConvPlan(T, B, L)   # bit comparison: same cut of the taps
a.set_ir(ir); b.set_ir(ir)
# parity: 20 buffers from a cold start
ok = True
for n in range(20):
    ya = a.process(xs[n % NBUF]).clone()
    prev = [xs[(n - k) % NBUF] if n - k >= 0 else zeros for k in range(1, 9)]
    yb = b.process_windowed(xs[n % NBUF], prev)
    ok &= bool(torch.equal(ya, yb))
print("windowed == ring-based (bits):", ok)
# throughput
fn = gab.lib.gab_conv_process_windowed
h = b._h
outs = [torch.empty(T * B, device="cuda") for _ in range(4)]
def args_for(n):
    prev = (C.c_void_p * 8)(*[xs[(n - k) % NBUF].data_ptr() for k in range(1, 9)])
    return C.c_void_p(xs[n % NBUF].data_ptr()), prev
prepared = [args_for(n) for n in range(NBUF)]
def run(n_steps, streams):
    sp = [C.c_void_p(s.cuda_stream) for s in streams]
    po = [C.c_void_p(o.data_ptr()) for o in outs]
    k = len(streams)
    for i in range(n_steps):
        x, prev = prepared[i % NBUF]
        fn(h, x, prev, po[i % k], sp[i % k])
def timeit(streams, n=4000):
    run(300, streams); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(n, streams); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6 / n
cur = torch.cuda.current_stream()
s = [torch.cuda.Stream() for _ in range(4)]
for rep in range(2):
    print("1 stream : %.2f us per buffer" % timeit([cur]))
    print("2 streams: %.2f us per buffer" % timeit(s[:2]))
    print("3 streams: %.2f us per buffer" % timeit(s[:3]))
    print("4 streams: %.2f us per buffer" % timeit(s[:4]), flush=True)
