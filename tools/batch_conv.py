"""gab_conv_process_batch (4096 taps x 512-sample buffers) across channel counts: us per buffer, algorithmic GB/s
and fraction of 8 TB/s, with the batch checked bit for bit against per-buffer launches on a sampled channel range.

    python tools/batch_conv.py [channels ...]          (default: 1024 4096 16384 65536)
"""
import os
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import gpuaudiobench_amd as gab
B, L = 512, 4096
for T in [int(a) for a in sys.argv[1:]] or [1024, 4096, 16384, 65536]:
    # buffers per launch: the caller's choice (bench.py: 64).  A launch's first window, drain and boundary are paid once,
    # so few buffers per launch cost most where a launch is several rounds of workgroups (65 536 channels: 64 rounds):
    # 16 / 16 / 8 buffers gave 0.84 / 0.84 / 0.74 at 4 096 / 16 384 / 65 536 channels, 64 / 64 / 32 give 0.90 / 0.88 / 0.86.
    n = 64 if T <= 16384 else 32
    n = int(os.environ.get("NBUF", n))                           # NBUF=4032: one launch as long as the engine's bench leg
    alg = 4 * T * (2 * B + 2 * L)
    ir = torch.from_numpy(gab.harness.conv_accel_ir(L, T)).cuda()
    a, b = gab.ConvPlan(T, B, L, scheme=os.environ.get("SCHEME")), gab.ConvPlan(T, B, L, scheme=os.environ.get("SCHEME"))   # SCHEME=classic: the other cut
    a.set_ir(ir); b.set_ir(ir)
    del ir
    x = torch.empty(n * T * B, device="cuda").uniform_(-1, 1)
    y = torch.empty_like(x)
    # parity: the first 9 buffers, per-buffer launches vs one batch launch
    seq = torch.cat([a.process(x[i * T * B:(i + 1) * T * B]) for i in range(min(n, 9))])
    b.process_batch(x[:min(n, 9) * T * B], min(n, 9), out=y[:min(n, 9) * T * B])
    torch.cuda.synchronize()
    same = bool(torch.equal(seq.view(torch.int32), y[:seq.numel()].view(torch.int32)))
    del seq
    a.close()
    args = b.prepare_batch(x, n, y)
    reps = max(3 if n > 1000 else 6, 6000 // (n * max(1, T // 1024)))
    for _ in range(max(3, reps // 2)):
        b.launch_batch(args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.launch_batch(args)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * n)
    print("T=%-6d %2d buffers per launch: %9.3f us per buffer, %6.0f GB/s algorithmic = %.3f of 8 TB/s, state %.0f MB, bit-identical to per-buffer launches: %s"
          % (T, n, us, alg / us / 1e3, alg / us / 1e3 / 8000, sum(b.state_bytes()) / 1e6, same), flush=True)
    b.close()
    del x, y
    torch.cuda.empty_cache()
