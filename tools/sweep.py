#!/usr/bin/env python3
"""Track-count sweeps: the shape of the reference poster's tables.

    python tools/sweep.py [--out DIR] [--max-tracks N] [rndmem] [gain] [iir] [conv] [modal]

For every benchmark and track count: median wall latency of one buffer through the harness
(copies included, as the reference reports it), median device time, the share of the buffer's
real-time deadline (bufferSize / fs = 10.667 ms) and whether validation passed.  The conv sweep
runs the streaming FFT convolution plan directly (4096 taps) up to 65 536 channels and reports
how many channels fit the deadline.  Writes <DIR>/sweep.csv, sweep.json and sweep.md.
"""
import argparse
import csv
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FS, B = 48000, 512
DEADLINE_MS = 1e3 * B / FS


def harness_sweep(gab, name, tracks_list, rows, iterations=30, **cfg):
    for T in tracks_list:
        b = gab.Benchmark(name, n_tracks=T, buffer_size=B, **cfg)
        b.setup()
        r = b.run(iterations=iterations, warmup=3)
        v, _ = b.validate()
        rows.append(dict(benchmark=name, tracks=T, wall_median_ms=r.median_ms, wall_p95_ms=r.p95_ms,
                         device_median_ms=r.gpu_median_ms, deadline_share=r.median_ms / DEADLINE_MS,
                         algorithmic_bytes=b.algorithmic_bytes(), valid=(v.status == 0)))
        print("%-16s T=%-6d wall %.4f ms  device %.4f ms  %.2f%% of deadline  valid=%s"
              % (name, T, r.median_ms, r.gpu_median_ms, 100 * r.median_ms / DEADLINE_MS, v.status == 0), flush=True)
        b.close()


def shard_checker(gab, taps):
    """The tool's own row check (tools may not touch the oracle): the sampled channels again on a small
    plan that holds the same rows of the global impulse-response bank — a shape the GPU tests pin to
    the oracle; the columns must agree bit for bit.  tests/test_sweep_gpu.py runs the sweep with an
    oracle-based checker instead."""
    import numpy as np
    import torch

    def check(T, lo, n_s, ir_rows, x_rows, got):
        shard = gab.ConvPlan(n_s, B, taps)
        shard.set_ir(torch.from_numpy(np.ascontiguousarray(ir_rows).ravel()).cuda())
        ok = True
        for x, y in zip(x_rows, got):
            ys = shard.process(torch.from_numpy(np.ascontiguousarray(x).ravel()).cuda()).view(B, n_s).cpu().numpy()
            ok = ok and np.array_equal(y.view(np.uint32), ys.view(np.uint32)) and bool(np.isfinite(y).all())
        shard.close()
        return ok
    return check


def conv_sweep(gab, tracks_list, rows, taps=4096, steps=400, checker=None):
    """Streaming FFT convolution, `taps` taps.  Every row is VALIDATED: after the timed loop the
    plan is reset, ten buffers go through it and a sampled range of 64 channels is handed to
    `checker(T, lo, n, ir_rows, x_rows, got)` (default: shard_checker)."""
    import numpy as np
    import torch
    rng = np.random.default_rng(taps)
    for T in tracks_list:
        t0 = time.time()
        ir_host = gab.harness.conv_accel_ir(taps, T).reshape(T, taps)
        plan = gab.ConvPlan(T, B, taps)
        plan.set_ir(torch.from_numpy(ir_host.ravel()).cuda())
        xs = [torch.from_numpy(gab.harness.noise(T * B, seed=s)).cuda() for s in range(4)]
        out = torch.empty(T * B, device="cuda")
        prepared = [plan.prepare(x, out) for x in xs]
        for i in range(200):
            plan.launch(prepared[i % 4])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(steps):
            plan.launch(prepared[i % 4])
        e1.record()
        torch.cuda.synchronize()
        dev_ms = e0.elapsed_time(e1) / steps
        # one buffer in flight, pinned host -> device -> kernel -> device -> pinned host
        h_in = torch.from_numpy(gab.harness.noise(T * B, seed=9)).pin_memory()
        h_out = torch.empty(T * B).pin_memory()
        d_in = torch.empty(T * B, device="cuda")
        rt = []
        for i in range(60):
            t1 = time.perf_counter()
            d_in.copy_(h_in, non_blocking=True)
            plan.process(d_in, out=out)
            h_out.copy_(out, non_blocking=True)
            torch.cuda.synchronize()
            if i >= 10:
                rt.append((time.perf_counter() - t1) * 1e3)
        wall = float(np.median(rt))
        # ---- validation: ten buffers from a reset, a sampled channel range
        n_s = min(64, T)
        lo = 4 * int(rng.integers(0, (T - n_s) // 4 + 1))
        plan.reset()
        x_rows, got = [], []
        for i in range(10):
            x = xs[i % 4]
            got.append(plan.process(x, out=out).view(B, T)[:, lo:lo + n_s].contiguous().cpu().numpy())
            x_rows.append(x.view(T, B)[lo:lo + n_s].contiguous().cpu().numpy())
        valid = bool((checker or shard_checker(gab, taps))(T, lo, n_s, ir_host[lo:lo + n_s], x_rows, got))
        alg = 4 * T * (2 * B + 2 * taps)
        rows.append(dict(benchmark="Conv1D_accel_stream_%d" % taps, tracks=T, wall_median_ms=wall,
                         wall_p95_ms=float(np.percentile(rt, 95)), device_median_ms=dev_ms,
                         deadline_share=wall / DEADLINE_MS, algorithmic_bytes=alg, valid=valid))
        print("conv %d taps      T=%-6d round trip %.4f ms  device %.4f ms (%.0f GB/s algorithmic)  %.2f%% of deadline"
              "  valid=%s  [setup %.1f s]" % (taps, T, wall, dev_ms, alg / dev_ms / 1e6, 100 * wall / DEADLINE_MS,
                                            valid, time.time() - t0), flush=True)
        plan.close()
        del xs, out, h_in, h_out, d_in
        torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="*", default=["rndmem", "gain", "iir", "conv", "modal"])
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sweep"))
    ap.add_argument("--max-tracks", type=int, default=65536)
    args = ap.parse_args()
    import gpuaudiobench_amd as gab

    pow2 = [t for t in (32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536) if t <= args.max_tracks]
    rows = []
    if "rndmem" in args.which:
        harness_sweep(gab, "RndMemRead", pow2, rows)
    if "gain" in args.which:
        harness_sweep(gab, "gain", pow2, rows)
        harness_sweep(gab, "GainStats", pow2, rows)
    if "iir" in args.which:
        harness_sweep(gab, "IIRFilter", [t for t in pow2 if t <= 16384], rows)
    if "modal" in args.which:
        harness_sweep(gab, "ModalFilterBank", [32, 128, 512, 1024], rows, iterations=10, modal_mode=1)
    if "conv" in args.which:
        conv_sweep(gab, [t for t in (128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536)
                         if t <= args.max_tracks], rows)
        # beyond one 4096-point window: the uniform-partition kernel (cuda/bench_conv1d_accel.cu:49-53
        # takes any ir_length)
        conv_sweep(gab, [t for t in (256, 1024, 4096, 16384) if t <= args.max_tracks], rows, taps=8192, steps=200)

    os.makedirs(args.out, exist_ok=True)
    keys = ["benchmark", "tracks", "wall_median_ms", "wall_p95_ms", "device_median_ms", "deadline_share",
            "algorithmic_bytes", "valid"]
    with open(os.path.join(args.out, "sweep.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=keys)
        w.writeheader()
        w.writerows(rows)
    with open(os.path.join(args.out, "sweep.json"), "w") as f:
        json.dump(dict(fs=FS, buffer_size=B, deadline_ms=DEADLINE_MS, rows=rows), f, indent=1)
    with open(os.path.join(args.out, "sweep.md"), "w") as f:
        f.write("Track-count sweep, buffer %d @ %d Hz (deadline %.3f ms), MI355X.\n\n" % (B, FS, DEADLINE_MS))
        for name in dict.fromkeys(r["benchmark"] for r in rows):
            sub = [r for r in rows if r["benchmark"] == name]
            f.write("**%s**\n\n| tracks | wall median ms | wall p95 ms | device ms | %% of deadline | alg. GB/s (device) | valid |\n"
                    "|---|---|---|---|---|---|---|\n" % name)
            for r in sub:
                gbs = r["algorithmic_bytes"] / r["device_median_ms"] / 1e6 if r["device_median_ms"] else float("nan")
                f.write("| %d | %.4f | %.4f | %.4f | %.2f | %.0f | %s |\n" % (
                    r["tracks"], r["wall_median_ms"], r["wall_p95_ms"], r["device_median_ms"],
                    100 * r["deadline_share"], gbs, r["valid"]))
            ok = [r for r in sub if r["deadline_share"] <= 1.0]
            if ok:
                last = ok[-1]
                est = int(last["tracks"] / max(last["deadline_share"], 1e-9))
                f.write("\nLargest measured size inside the deadline: %d tracks (%.2f %% of it); linear extrapolation "
                        "of that point: ~%d tracks.\n\n" % (last["tracks"], 100 * last["deadline_share"], est))
    print("wrote", args.out)


if __name__ == "__main__":
    main()
