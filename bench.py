#!/usr/bin/env python3
"""bench.py — headline benchmark: streaming 4096-tap FIR convolution
(bench_conv1d_accel) x 1024 channels x 512-sample buffers @ 48 kHz on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over the resident batch of synthetic input:
BUFFERS_PER_STEP = 32 consecutive audio buffers (512 new samples for every one of a
rank's 1024 channels), each pushed through the fused overlap-save kernel with
carried history — one launch (or one set of channel-range launches) per buffer, each
depending on the one before.  Inputs are resident in HBM before the timed region.

With N > 1 every rank owns a 1024-channel shard of an N*1024-channel job: rank 0
generates the whole impulse-response bank, it is broadcast over RCCL/xGMI once,
each rank transforms its slice; there is no per-buffer collective (channels are
independent), so scaling is weak.  A plain `python bench.py --gpus N` starts the N
ranks itself (torch.distributed.run) before touching any GPU and relays rank 0's line.

One JSON line on rank 0:
  value        = 1024-channel buffers per second, whole job (N * 32 K / max-rank time)
  roofline     = algorithmic bytes per buffer / device period per buffer (HIP events
                 on the launch streams over the timed region), against 8 TB/s HBM
  cpu_baseline = the CPU oracle (the reference golden extended with history), timed on a
                 bounded sample of the same workload on this box's cores (N = 1 only)
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TRACKS_PER_GPU = 1024
BUFSIZE = 512
TAPS = 4096
FS = 48000
BUFFERS_PER_STEP = 32           # the resident input batch: distinct buffers cycled through (64 MiB)
CLOCK_WARM_BUFFERS = 3000       # untimed, besides --warmup: clocks and caches at their running state
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
DEFAULT_STREAMS = 2             # channel ranges per buffer, each on its own stream (1 = one launch per buffer)
TRAFFIC_SOURCE = "profiles/r02b_conv_split_pmc_means.json"


def cpu_threads():
    """Threads for the CPU baseline: GAB_BENCH_CPU_THREADS, else the smallest of the affinity mask,
    the cgroup CPU quota and 16 (a one-GPU box's CPU share)."""
    env = os.environ.get("GAB_BENCH_CPU_THREADS")
    if env:
        return max(1, int(env))
    n = min(len(os.sched_getaffinity(0)), 16)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def algorithmic_bytes(tracks, bufsize, taps):
    # new input + output + every tap + every history sample the taps reach
    return 4 * tracks * (2 * bufsize + 2 * taps)


def launch_ranks(args, argv):
    """`bench.py --gpus N` as a plain command: start N fresh rank processes (one per GPU) with
    torch.distributed.run, relay rank 0's JSON line, return the launcher's exit code.  Runs before
    this process has imported torch or touched a GPU."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            sys.stderr.write(ln + "\n")
    if line is not None:
        print(line, flush=True)
    if proc.returncode == 0 and line is None:
        sys.stderr.write("bench.py: the ranks exited cleanly but printed no result line\n")
        return 1
    return proc.returncode


def dry_run(args, rank, world):
    """GAB_BENCH_DRYRUN=1: the launcher, rendezvous, bank broadcast and timing collectives on CPU
    tensors over gloo, with NO device work — what the CPU test suite can run of the N > 1 path.
    The line says so and carries no measurement."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from gpuaudiobench_amd import sharding
    T, L = int(os.environ.get("GAB_BENCH_DRYRUN_TRACKS", "8")), 64
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    t0 = time.perf_counter()
    ir = sharding.broadcast_ir_bank(L, T * world, rank, world, torch.device("cpu"), dist if world > 1 else None)
    bcast_ms = (time.perf_counter() - t0) * 1e3
    lo, hi = sharding.shard_range(rank, world, T * world)
    ok = bool(np.array_equal(ir.numpy().ravel(), __import__("gpuaudiobench_amd").harness.conv_accel_ir(L, hi - lo, lo, T * world)))
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    elapsed = time.perf_counter() - t0 + 1e-9
    if world > 1:
        dist.barrier()
        t = torch.tensor([elapsed, 1.0 if ok else 0.0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t2 = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64)
        dist.all_reduce(t2, op=dist.ReduceOp.MIN)
        ok = bool(t2.item() == 1.0)
    if rank == 0:
        print(json.dumps({"metric": "audio_buffers_per_sec", "value": None, "unit": "buffers/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                          "data": "DRY RUN (GAB_BENCH_DRYRUN=1): rendezvous, bank broadcast and collectives "
                                  "on CPU over gloo; no device work, no measurement",
                          "config": {"workload": "dry run", "ir_broadcast_ms": bcast_ms,
                                     "ir_slices_match_global_bank": ok}}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150, help="timed steps of %d buffers each" % BUFFERS_PER_STEP)
    ap.add_argument("--warmup", type=int, default=15, help="untimed steps (besides the fixed clock warm-up)")
    ap.add_argument("--streams", type=int, default=DEFAULT_STREAMS,
                    help="channel ranges per buffer, one HIP stream and host thread each (1 = one launch per buffer)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-legs", action="store_true", help="only the timed region (profiling runs)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, sys.argv[1:])

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if os.environ.get("GAB_BENCH_DRYRUN") == "1":
        return dry_run(args, rank, world)

    import numpy as np
    import torch
    import torch.distributed as dist

    # GAB_BENCH_REHEARSE=1: every rank on device 0 with gloo, to walk the N>1 code path on a one-GPU
    # box (ranks then share the device, so the rate means nothing; the line says so in `data`)
    rehearse = os.environ.get("GAB_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import gpuaudiobench_amd as gab           # raises if libgab_hip.so is missing

    T, B, L = TRACKS_PER_GPU, BUFSIZE, TAPS
    T_total = T * world
    NB = BUFFERS_PER_STEP

    # ---- impulse-response bank: generated once, broadcast over RCCL -------------
    from gpuaudiobench_amd import sharding
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ir_dev = sharding.broadcast_ir_bank(L, T_total, rank, world, dev, dist if world > 1 else None)
    torch.cuda.synchronize()
    bcast_ms = (time.perf_counter() - t0) * 1e3 if world > 1 else None

    plan = gab.ConvPlan(T, B, L)
    plan.set_ir(ir_dev)
    spectra_bytes, history_bytes = plan.state_bytes()
    R = args.streams if plan.scheme == "split" else 1
    if R < 1 or T % (4 * R):
        raise SystemExit("--streams must cut %d channels into ranges that are multiples of 4" % T)

    # ---- synthetic input: the reference's noise generator, this rank's channels --
    inputs = [torch.from_numpy(sharding.shard_noise(T_total, B, rank, world, seed=42 + i)).to(dev)
              for i in range(NB)]
    out = torch.empty(T * B, dtype=torch.float32, device=dev)
    main_stream = torch.cuda.current_stream()
    if R == 1:
        streams = [main_stream]
        step_args = [plan.prepare(x, out, gab.CONV_STREAMING) for x in inputs]
        launch_one = plan.launch

        def run_buffers(n):            # one gab_conv_process call per buffer, arguments prepared once
            for i in range(n):
                launch_one(step_args[i % NB])
    else:
        streams = [torch.cuda.Stream() for _ in range(R)]
        ranges = [(r * (T // R), T // R) for r in range(R)]

        def run_buffers(n):            # the library's launch loop: one host thread and one stream per range
            plan.stream_ranges(inputs, out, ranges, streams, n)

    def run_steps(k):
        run_buffers(k * NB)

    run_buffers(CLOCK_WARM_BUFFERS)
    run_steps(args.warmup)

    # ---- timed region -------------------------------------------------------------
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in streams]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s, (a, _) in zip(streams, ev):
        a.record(s)
    run_steps(args.steps)
    for s, (_, b) in zip(streams, ev):
        b.record(s)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    n_buffers = args.steps * NB
    region_ms = max(a.elapsed_time(b) for a, b in ev)      # device view of the same launches
    period_us = region_ms * 1e3 / n_buffers

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    wall_us = elapsed * 1e6 / n_buffers

    alg = algorithmic_bytes(T, B, L)
    side = {}
    if not args.no_side_legs:
        side = side_legs(gab, plan, inputs, out, main_stream, R, dev, T, B, L, np, torch)

    traffic = None
    pmc_file = os.path.join(ROOT, TRAFFIC_SOURCE)
    if os.path.exists(pmc_file):
        try:
            traffic = json.load(open(pmc_file)).get("hbm_traffic_bytes_per_launch") if plan.scheme == "split" else None
        except Exception:
            traffic = None

    achieved = alg / (period_us * 1e-6) / 1e9
    result = {
        "metric": "audio_buffers_per_sec",
        "value": world * n_buffers / elapsed,
        "unit": "buffers/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3 / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic" if not rehearse else "synthetic; REHEARSAL: all ranks share device 0 over gloo",
        "config": {
            "workload": "bench_conv1d_accel streaming overlap-save: %d-tap IR x %d channels x "
                        "%d-sample buffers @ %d Hz per GPU (BASELINE configs[2]%s); one step = %d consecutive buffers"
                        % (L, T, B, FS, "; configs[4]-style channel sharding, %d channels total" % T_total
                           if world > 1 else "", NB),
            "taps": L, "channels_per_gpu": T, "channels_total": T_total, "buffer_size": B, "fs": FS,
            "buffers_per_step": NB, "us_per_buffer": wall_us,
            "mode": "streaming", "tap_cut": plan.scheme,
            "launches_per_buffer": R,
            "launch_path": ("gab_conv_process, one launch per buffer" if R == 1 else
                            "gab_conv_stream_ranges: %d channel ranges per buffer, one stream and host thread each" % R),
            "clock_warm_buffers": CLOCK_WARM_BUFFERS,
            "realtime_factor": (world * n_buffers / elapsed) * B / FS,
            "ir_broadcast_ms": bcast_ms,
            "state_bytes": {"spectra": spectra_bytes, "history": history_bytes},
        },
        "roofline": {
            "bound": "hbm",
            "kernel": ("conv_split_kernel" if R == 1 else "conv_split_range_kernel x%d per buffer" % R)
                      if plan.scheme == "split" else "conv_overlap_save_kernel<true,true>",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "frac_wall": alg / (wall_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": TRAFFIC_SOURCE + " (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes of an earlier run over one-launch-per-buffer conv_split_kernel; not measured in this run)"
                              if traffic is not None else None,
            "algorithmic_bytes_per_buffer": alg,
            "launches_in_flight": R,
            "how": ("one launch per buffer: achieved = algorithmic bytes per launch / launch period (HIP events on the launch stream)"
                    if R == 1 else
                    "%d concurrent channel-range launches per buffer: achieved = algorithmic bytes per BUFFER / device period "
                    "per buffer (HIP events on every range stream, slowest stream); a kernel tracer serialises the streams, so "
                    "rocprofv3's per-kernel average describes a range kernel running alone — the figure it can confirm is "
                    "config.one_launch_per_buffer_single_stream" % R),
            "device_period_us_per_buffer": period_us,
            "wall_period_us_per_buffer": wall_us,
        },
    }
    result["config"].update(side)

    # ---- CPU baseline: the oracle on this box's cores, bounded sample (rank 0, N = 1) ----
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args, ir_dev, inputs, T, B, L, np)
    elif rank == 0:
        result["cpu_baseline"] = None

    plan.close()
    # ---- library yardstick (tools only, a child process; never on the product path): the
    # reference's pipeline on rocFFT, same GPU, same shape (tools/ubench/library_baseline.cpp)
    if world == 1 and rank == 0 and not args.no_side_legs:
        result["library_baseline"] = library_baseline(T, B, L)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def library_baseline(T, B, L):
    exe = os.path.join(ROOT, "tools", "ubench", "bin", "library_baseline")
    if not os.path.exists(exe):
        return None
    try:
        r = subprocess.run([exe, str(T), str(B), str(L), "1000"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True, timeout=120)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": (r.stderr or r.stdout)[-300:]}
        d = json.loads(line[-1])
        d["what"] = ("cuda/bench_conv1d_accel.cu:258-304 restated on hipFFT/rocFFT (tools/ubench/library_baseline.cpp): "
                     "stateless, one N = nextpow2(L+B-1) transform pair per buffer; a yardstick, not the product")
        return d
    except Exception as e:              # a missing FFT library must not cost the bench line
        return {"error": repr(e)[:300]}


def side_legs(gab, plan, inputs, out, stream, R, dev, T, B, L, np, torch):
    """Rates that are NOT `value`, each with a fixed iteration count (independent of --steps)."""
    NB = len(inputs)
    alg = algorithmic_bytes(T, B, L)
    res = {}

    # ---- the same buffers as ONE launch per buffer on one stream, HIP events around 4000 launches
    plan.reset()
    step_args = [plan.prepare(x, out, gab.CONV_STREAMING) for x in inputs]
    for i in range(500):
        plan.launch(step_args[i % NB])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for i in range(4000):
        plan.launch(step_args[i % NB])
    e1.record(stream)
    torch.cuda.synchronize()
    one_us = e0.elapsed_time(e1) * 1e3 / 4000
    res["one_launch_per_buffer_single_stream"] = {
        "kernel": "conv_split_kernel" if plan.scheme == "split" else "conv_overlap_save_kernel<true,true>",
        "us_per_buffer": one_us, "buffers_per_sec": 1e6 / one_us, "alg_GBps": alg / one_us / 1e3,
        "frac": alg / one_us / 1e3 / HBM_PEAK_GBS, "launches": 4000}

    # ---- batch mode: 32 buffers per launch, for callers that hold the input ahead of time (on a split
    # plan: conv_split_batch_kernel, both roles of a duo in one resident workgroup, no kernel boundary)
    nb = 32
    xb = torch.cat([inputs[i % NB] for i in range(nb)])
    yb = torch.empty_like(xb)
    for _ in range(20):
        plan.process_batch(xb, nb, out=yb)
    eb0, eb1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eb0.record(stream)
    for _ in range(100):
        plan.process_batch(xb, nb, out=yb)
    eb1.record(stream)
    torch.cuda.synchronize()
    batch_us = eb0.elapsed_time(eb1) * 1e3 / (100 * nb)
    res["batch_mode_32_buffers_per_launch"] = {"us_per_buffer": batch_us, "buffers_per_sec": 1e6 / batch_us,
                                               "alg_GBps": alg / batch_us / 1e3, "frac": alg / batch_us / 1e3 / HBM_PEAK_GBS,
                                               "launches": 100, "tap_cut": plan.scheme}
    del xb, yb
    plan.reset()

    # ---- p50 round trip: pinned host -> HBM -> kernel -> HBM -> pinned host, one buffer in flight
    h_in = torch.from_numpy(gab.harness.noise(T * B, seed=7)).pin_memory()
    h_out = torch.empty(T * B, dtype=torch.float32).pin_memory()
    d_in = torch.empty(T * B, dtype=torch.float32, device=dev)
    rt = []
    for i in range(320):
        t1 = time.perf_counter()
        d_in.copy_(h_in, non_blocking=True)
        plan.process(d_in, out=out, mode=gab.CONV_STREAMING)
        h_out.copy_(out, non_blocking=True)
        stream.synchronize()
        if i >= 20:
            rt.append((time.perf_counter() - t1) * 1e6)
    rt = np.array(rt)
    res["p50_round_trip_us"] = float(np.percentile(rt, 50))
    res["p95_round_trip_us"] = float(np.percentile(rt, 95))

    # ---- zero-copy round trip: the kernel reads the pinned input and writes the pinned output itself
    h_out_zc = torch.empty(T * B, dtype=torch.float32).pin_memory()
    zc = []
    for i in range(220):
        t1 = time.perf_counter()
        plan.process(h_in, out=h_out_zc, mode=gab.CONV_STREAMING)
        stream.synchronize()
        if i >= 20:
            zc.append((time.perf_counter() - t1) * 1e6)
    zc = np.array(zc)
    plan.reset()          # host-io launches use the classic cut; back to the plan's own
    res["p50_round_trip_zero_copy_us"] = float(np.percentile(zc, 50))
    res["p95_round_trip_zero_copy_us"] = float(np.percentile(zc, 95))

    # ---- the same round trip under DAW pacing: one buffer per 512/48000 s slot, device idle between
    paced = []
    daw = gab.harness.DawSim(buffer_seconds=float(B) / FS, mode="spin")
    for i in range(105):
        daw.wait()
        t1 = time.perf_counter()
        d_in.copy_(h_in, non_blocking=True)
        plan.process(d_in, out=out, mode=gab.CONV_STREAMING)
        h_out.copy_(out, non_blocking=True)
        stream.synchronize()
        if i >= 5:
            paced.append((time.perf_counter() - t1) * 1e6)
    paced = np.array(paced)
    waits, missed = daw.stats()
    daw.close()
    res["paced_10p667ms"] = {"p50_round_trip_us": float(np.percentile(paced, 50)),
                             "p95_round_trip_us": float(np.percentile(paced, 95)),
                             "max_round_trip_us": float(paced.max()),
                             "slots": int(waits), "missed_slots": int(missed)}
    return res


def cpu_baseline(args, ir_dev, inputs, T, B, L, np):
    import oracle                     # checker / baseline only — never the product path
    from concurrent.futures import ThreadPoolExecutor
    from gpuaudiobench_amd import sharding
    NB = len(inputs)
    ir_host = ir_dev.cpu().numpy().reshape(T, L)
    xs = [inputs[i].cpu().numpy().reshape(T, B) for i in range(NB)]
    budget = args.cpu_baseline_seconds / 2.0

    # (i) the reference's golden as it runs it: scalar loops, one thread
    hist = np.zeros(T * L, np.float32)
    t1 = time.perf_counter()
    oracle.conv_accel_stream(xs[0].ravel(), ir_host.ravel(), hist, L, B, T)
    first = time.perf_counter() - t1
    n_one = max(1, min(20, int(budget / first) - 1))
    t1 = time.perf_counter()
    for i in range(n_one):
        oracle.conv_accel_stream(xs[(i + 1) % NB].ravel(), ir_host.ravel(), hist, L, B, T)
    dt_one = time.perf_counter() - t1

    # (ii) the same loops with the channels cut over every core this process may use (the
    # library call releases the interpreter lock; channels are independent)
    cores = max(1, min(cpu_threads(), T))
    cuts = [sharding.shard_range(k, cores, T) for k in range(cores)]
    hists = [np.zeros((hi - lo) * L, np.float32) for lo, hi in cuts]
    irs = [np.ascontiguousarray(ir_host[lo:hi]).ravel() for lo, hi in cuts]
    xcut = [[np.ascontiguousarray(x[lo:hi]).ravel() for lo, hi in cuts] for x in xs]

    def one(k, i):
        lo, hi = cuts[k]
        oracle.conv_accel_stream(xcut[i % NB][k], irs[k], hists[k], L, B, hi - lo)

    with ThreadPoolExecutor(max_workers=cores) as pool:
        list(pool.map(lambda k: one(k, 0), range(cores)))            # warm
        n_all = max(2, min(200, int(budget * cores / first * 0.8)))
        t1 = time.perf_counter()
        for i in range(n_all):
            list(pool.map(lambda k, i=i: one(k, i + 1), range(cores)))
        dt_all = time.perf_counter() - t1
    return {
        "value": n_all / dt_all,
        "unit": "buffers/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d buffers of the full %d-channel x %d-tap workload, direct-form fp32 "
                  "(oracle/gab_oracle.c orc_conv_accel_stream), channels cut over %d threads, %.1f s; "
                  "single thread (as the reference runs its golden): %d buffers in %.1f s"
                  % (n_all, T, L, cores, dt_all, n_one, dt_one),
        "single_thread_value": n_one / dt_one,
    }


if __name__ == "__main__":
    sys.exit(main())
