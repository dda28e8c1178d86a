#!/usr/bin/env python3
"""Device time of the secondary streaming kernels at scale, without the per-launch event floor:
N back-to-back launches between two HIP events on the launch stream.

    python tools/secondary_bench.py [--reps 400]      -> one JSON line per kernel
"""
import argparse, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpuaudiobench_amd as gab

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=400)
ap.add_argument("--only", default="")
args = ap.parse_args()
PEAK = 8000.0


def timed(fn, reps):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def report(name, cfg, us, nbytes):
    gbps = nbytes / us / 1e3
    print(json.dumps({"kernel": name, "config": cfg, "us": round(us, 3), "alg_bytes": nbytes,
                      "alg_GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / PEAK, 3)}), flush=True)


B = 512
want = lambda n: not args.only or args.only in n
rng = np.random.default_rng(0)
if want("iir"):
    for T in (8192, 65536):
        x = torch.from_numpy(rng.standard_normal(T * B).astype(np.float32)).cuda()
        st = torch.zeros(2 * T, device="cuda")
        c = [0.292875, 0.585750, 0.292875, 5.12078699e-08, 0.171500]
        for seq in (False, True):
            us = timed(lambda: gab.iir(x, c, st, T, B, sequential=seq), args.reps if not seq else max(20, args.reps // 8))
            report("iir_biquad_kernel (sequential, bit-exact)" if seq else "iir_scan_kernel", {"tracks": T, "bufsize": B}, us, 2 * T * B * 4 + 16 * T)
if want("fft"):
    for T in (8192, 65536):
        x = torch.from_numpy(rng.standard_normal(T * 1024).astype(np.float32)).cuda()
        us = timed(lambda: gab.fft_r2c_1024(x, T), args.reps)
        report("fft_r2c_1024_kernel", {"tracks": T}, us, T * (1024 * 4 + 513 * 8))
if want("rndmem"):
    pool = torch.from_numpy(rng.standard_normal(1 << 27).astype(np.float32)).cuda()
    for T in (8192, 65536):
        ph = torch.from_numpy(rng.integers(0, (1 << 27) - B, T).astype(np.int32)).cuda()
        us = timed(lambda: gab.rndmem(pool, ph, T, B), args.reps)
        report("rndmem_kernel", {"tracks": T, "bufsize": B, "pool_MiB": 512}, us, 2 * T * B * 4)
    del pool
if want("gain"):
    for T in (8192, 65536):
        x = torch.from_numpy(rng.standard_normal(T * B).astype(np.float32)).cuda()
        out = torch.empty_like(x)
        us = timed(lambda: gab.gain(x, 2.0, out=out), args.reps)
        report("scale_vec4_kernel (gain)", {"tracks": T, "bufsize": B}, us, 2 * T * B * 4)
        us = timed(lambda: gab.gainstats(x, T, B), args.reps)
        report("gainstats_kernel", {"tracks": T, "bufsize": B}, us, 2 * T * B * 4 + 8 * T)
