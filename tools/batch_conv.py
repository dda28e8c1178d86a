"""gab_conv_process_batch at BASELINE C3 (4096 taps x 1024 channels x 512): us per buffer for n buffers
per launch, on the plan's own cut (split: conv_split_batch_kernel, both roles of a duo in one resident
workgroup) and on the classic cut (conv_batch_kernel), plus parity of the two paths with per-buffer launches."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import gpuaudiobench_amd as gab
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B, L = 512, 4096
ir = torch.from_numpy(gab.harness.conv_accel_ir(L, T)).cuda()
alg = 4 * T * (2 * B + 2 * L)
for scheme in ("split", "classic"):
    a, b = gab.ConvPlan(T, B, L, scheme=scheme), gab.ConvPlan(T, B, L, scheme=scheme)
    a.set_ir(ir); b.set_ir(ir)
    xs = torch.cat([torch.from_numpy(gab.harness.noise(T * B, seed=60 + i)) for i in range(8)]).cuda()
    seq = torch.cat([a.process(xs[i * T * B:(i + 1) * T * B].contiguous()) for i in range(8)])
    y = b.process_batch(xs, 8)
    torch.cuda.synchronize()
    same = bool(torch.equal(seq.view(torch.int32), y.view(torch.int32)))
    for n in (4, 16, 64):
        x = torch.cat([xs] * (n // 8)) if n >= 8 else xs[:n * T * B].contiguous()
        out = torch.empty_like(x)
        for _ in range(5):
            b.process_batch(x, n, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(4, 2000 // n)
        e0.record()
        for _ in range(reps):
            b.process_batch(x, n, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (reps * n)
        print("T=%d %s cut, %d buffers per launch: %.3f us per buffer, %.0f GB/s algorithmic (%.3f of 8 TB/s)%s"
              % (T, scheme, n, us, alg / us / 1e3, alg / us / 1e3 / 8000, "" if same else "  PARITY MISMATCH"), flush=True)
    a.close(); b.close()
