"""Round trip of one buffer (pinned host in -> conv -> pinned host out), two ways:
copies (H2D, kernel, D2H on one stream) vs zero-copy (the kernel reads/writes pinned memory)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import gpuaudiobench_amd as gab
T, B, L = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), 512, 4096
plan = gab.ConvPlan(T, B, L)
plan.set_ir(torch.from_numpy(gab.harness.conv_accel_ir(L, T)).cuda())
h_in = torch.from_numpy(gab.harness.noise(T * B, seed=7)).pin_memory()
h_out = torch.empty(T * B).pin_memory()
h_out2 = torch.empty(T * B).pin_memory()
d_in = torch.empty(T * B, device="cuda")
d_out = torch.empty(T * B, device="cuda")
s = torch.cuda.current_stream()
def copies():
    d_in.copy_(h_in, non_blocking=True)
    plan.process(d_in, out=d_out)
    h_out.copy_(d_out, non_blocking=True)
    s.synchronize()
def zero_copy():
    plan.process(h_in, out=h_out2)
    s.synchronize()
def zero_in():
    plan.process(h_in, out=d_out)
    h_out2.copy_(d_out, non_blocking=True)
    s.synchronize()
def zero_out():
    d_in.copy_(h_in, non_blocking=True)
    plan.process(d_in, out=h_out2)
    s.synchronize()
for name, fn in (("copies", copies), ("zero-copy in+out", zero_copy), ("zero-copy in", zero_in), ("zero-copy out", zero_out)):
    plan.reset(); torch.cuda.synchronize()
    ts = []
    for i in range(220):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e6)
    ts = np.array(ts[20:])
    print("%-18s p50 %.1f us  p95 %.1f us" % (name, np.percentile(ts, 50), np.percentile(ts, 95)))
# same results both ways (same history sequence after reset)
plan.reset(); copies(); a = h_out.clone(); plan.reset(); zero_copy(); print("identical:", bool(torch.equal(a, h_out2)))
