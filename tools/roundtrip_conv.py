"""Round trip of one buffer (pinned host in -> conv -> pinned host out), host clock around each call:
  copies            H2D copy, kernel, D2H copy on one stream, stream synchronize (the reference's iteration)
  zero-copy         the kernel reads and writes the pinned buffers itself (GAB_CONV_STREAMING_HOST_IO)
  overlapped        gab_conv_round_trip: engine upload consumed as it lands, outputs drained per channel group
Usage: python tools/roundtrip_conv.py [channels] [iterations]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import gpuaudiobench_amd as gab
T, B, L = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), 512, 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 420
ir = torch.from_numpy(gab.harness.conv_accel_ir(L, T)).cuda()
h_in = torch.from_numpy(gab.harness.noise(T * B, seed=7)).pin_memory()
h_out = torch.empty(T * B).pin_memory()
d_in = torch.empty(T * B, device="cuda")
d_out = torch.empty(T * B, device="cuda")
s = torch.cuda.current_stream()
results = {}
for scheme in ("split", "classic"):
    plan = gab.ConvPlan(T, B, L, scheme=scheme)
    plan.set_ir(ir)
    def copies():
        d_in.copy_(h_in, non_blocking=True)
        plan.process(d_in, out=d_out)
        h_out.copy_(d_out, non_blocking=True)
        s.synchronize()
    def zero_copy():
        plan.process(h_in, out=h_out)
        s.synchronize()
    rt_args = plan.prepare_round_trip(h_in, h_out)
    def overlapped():
        plan.launch_round_trip(rt_args)
    legs = [("copies", copies), ("zero-copy", zero_copy)] + ([("overlapped", overlapped)] if scheme == "classic" else [])
    for name, fn in legs:
        plan.reset(); torch.cuda.synchronize()
        ts = []
        for i in range(N):
            t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e6)
        ts = np.array(ts[20:])
        results[(scheme, name)] = ts
        print("%-8s %-12s p50 %6.1f us  p95 %6.1f us  min %6.1f  max %7.1f" % (scheme, name, np.percentile(ts, 50), np.percentile(ts, 95), ts.min(), ts.max()), flush=True)
    # same bits whichever way the buffer travels (same history sequence after a reset)
    outs = []
    for name, fn in legs:
        plan.reset()
        for _ in range(10):
            fn()
        outs.append(h_out.clone())
    print("%-8s identical across legs: %s" % (scheme, all(torch.equal(outs[0].view(torch.int32), o.view(torch.int32)) for o in outs[1:])), flush=True)
    plan.close()
