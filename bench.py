#!/usr/bin/env python3
"""bench.py — headline benchmark: streaming 4096-tap FIR convolution
(bench_conv1d_accel) x 1024 channels x 512-sample buffers @ 48 kHz on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one audio buffer (512 new samples for every one of a rank's 1024
channels) pushed through the fused overlap-save kernel with carried history.
Inputs are resident in HBM before the timed region.  With N > 1 (launched by
torch.distributed.run, one rank per GPU) every rank owns a 1024-channel shard
of an N*1024-channel job: rank 0 generates the whole impulse-response bank, it
is broadcast over RCCL/xGMI once, each rank transforms its slice; there is no
per-buffer collective (channels are independent), so scaling is weak.

One JSON line on rank 0:
  value      = 1024-channel buffers per second, whole job (N * K / max-rank time)
  roofline   = algorithmic bytes per launch / average kernel duration (HIP events
               on the launch stream), against the 8 TB/s HBM peak
  cpu_baseline = the CPU oracle (reference golden extended with history), one
               core, timed on a bounded sample of the same workload (N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TRACKS_PER_GPU = 1024
BUFSIZE = 512
TAPS = 4096
FS = 48000
N_INPUT_BUFFERS = 16            # distinct input buffers cycled through (32 MiB of HBM)
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--roundtrip-iters", type=int, default=300)
    ap.add_argument("--paced-iters", type=int, default=150)
    # these launches run at PCIe speed; they use the kernel's host-io name, so a rocprofv3 --stats
    # average of the timed kernel is not skewed by them
    ap.add_argument("--zero-copy-iters", type=int, default=300)
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run "
                     "--nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
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

    # ---- synthetic input: the reference's noise generator, this rank's channels --
    inputs = [torch.from_numpy(sharding.shard_noise(T_total, B, rank, world, seed=42 + i)).to(dev)
              for i in range(N_INPUT_BUFFERS)]
    out = torch.empty(T * B, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream()

    # one gab_conv_process call per buffer; the ctypes arguments are built once per input buffer so
    # that the timed loop stays device-bound (a launch costs the host ~4 us this way, ~7.5 us through
    # ConvPlan.process, against ~9 us of device time)
    step_args = [plan.prepare(x, out, gab.CONV_STREAMING) for x in inputs]
    launch_one = plan.launch

    def step(i):
        launch_one(step_args[i % N_INPUT_BUFFERS])

    for i in range(args.warmup):
        step(i)

    # ---- timed region -------------------------------------------------------------
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(stream)
    for i in range(args.steps):
        step(i)
    ev1.record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    region_ms = ev0.elapsed_time(ev1)                 # device view of the same K launches

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-launch kernel duration: one event pair per launch (outside `value`) ---
    n_pairs = min(args.steps, 400)
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
             for _ in range(n_pairs)]
    for i, (a, b) in enumerate(pairs):
        a.record(stream)
        step(i)
        b.record(stream)
    torch.cuda.synchronize()
    kernel_us = float(np.mean([a.elapsed_time(b) for a, b in pairs]) * 1e3)
    period_us = region_ms * 1e3 / args.steps

    # ---- batch mode (not `value`): 16 buffers per launch, for callers that have the input ahead
    # of time; same results, no kernel boundary between buffers ------------------------------
    nb = 16
    xb = torch.cat([inputs[i % N_INPUT_BUFFERS] for i in range(nb)])
    yb = torch.empty_like(xb)
    for _ in range(20):
        plan.process_batch(xb, nb, out=yb)
    eb0, eb1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    eb0.record(stream)
    nrep = max(1, min(args.steps, 3200) // nb)
    for _ in range(nrep):
        plan.process_batch(xb, nb, out=yb)
    eb1.record(stream)
    torch.cuda.synchronize()
    batch_us = eb0.elapsed_time(eb1) * 1e3 / (nrep * nb)
    del xb, yb
    plan.reset()          # a batch call moves the history ring with the classic cut; back to the plan's own

    # ---- pipelined mode (not `value`): the stateless kernel (history = the caller's last eight
    # input buffers), consecutive buffers alternating between two streams, so the device overlaps
    # the end of one launch with the start of the next.  One launch per buffer, same bits. ------
    side = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs2 = [torch.empty(T * B, dtype=torch.float32, device=dev) for _ in side]
    prepared = []
    for i in range(N_INPUT_BUFFERS * len(side)):           # (input index, stream) repeats with this period
        prev = [inputs[(i - k) % N_INPUT_BUFFERS] for k in range(1, 9)]
        j = i % len(side)
        prepared.append(plan.prepare_windowed(inputs[i % N_INPUT_BUFFERS], prev, outs2[j], side[j]))
    launch = plan.launch_prepared
    n_pipe = min(args.steps, 4000)
    for i in range(200):
        launch(prepared[i % len(prepared)])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(n_pipe):
        launch(prepared[i % len(prepared)])
    torch.cuda.synchronize()
    pipe_us = (time.perf_counter() - t1) * 1e6 / n_pipe

    # ---- channel ranges on two streams (not `value`): the same buffers, each launched as two
    # halves of the channels on two streams.  Channels are independent, so every stream is its own
    # chain of dependent launches and the chains overlap each other's kernel boundaries.  Same bits.
    two_us = None
    if plan.scheme == "split" and T % 8 == 0:
        plan.reset()
        halves = [(0, T // 2), (T // 2, T // 2)]
        plan.stream_ranges(inputs, out, halves, side, 200)
        torch.cuda.synchronize()
        n_two = min(args.steps, 3000)
        t1 = time.perf_counter()
        plan.stream_ranges(inputs, out, halves, side, n_two)     # the launch loop runs in the library
        torch.cuda.synchronize()
        two_us = (time.perf_counter() - t1) * 1e6 / n_two
        plan.reset()

    # ---- p50 round trip: pinned host -> HBM -> kernel -> HBM -> pinned host ----------
    h_in = torch.from_numpy(gab.harness.noise(T * B, seed=7)).pin_memory()
    h_out = torch.empty(T * B, dtype=torch.float32).pin_memory()
    d_in = torch.empty(T * B, dtype=torch.float32, device=dev)
    rt = []
    for i in range(args.roundtrip_iters + 20):
        t1 = time.perf_counter()
        d_in.copy_(h_in, non_blocking=True)
        plan.process(d_in, out=out, mode=gab.CONV_STREAMING)
        h_out.copy_(out, non_blocking=True)
        stream.synchronize()
        if i >= 20:
            rt.append((time.perf_counter() - t1) * 1e6)
    rt = np.array(rt)

    # ---- zero-copy round trip: the kernel reads the pinned input and writes the pinned output
    # itself (no copy commands); same buffer, same history sequence ----------------------------
    zc = None
    if args.zero_copy_iters > 0:
        h_out_zc = torch.empty(T * B, dtype=torch.float32).pin_memory()
        zc = []
        for i in range(args.zero_copy_iters + 20):
            t1 = time.perf_counter()
            plan.process(h_in, out=h_out_zc, mode=gab.CONV_STREAMING)
            stream.synchronize()
            if i >= 20:
                zc.append((time.perf_counter() - t1) * 1e6)
        zc = np.array(zc)
        plan.reset()      # host-io launches use the classic cut; back to the plan's own

    # ---- the same round trip under DAW pacing: one buffer per 512/48000 s slot, device idle in
    # between (the Metal port's DAWSimulator) --------------------------------------
    paced = []
    daw = gab.harness.DawSim(buffer_seconds=float(B) / FS, mode="spin")
    for i in range(args.paced_iters + 5):
        daw.wait()
        t1 = time.perf_counter()
        d_in.copy_(h_in, non_blocking=True)
        plan.process(d_in, out=out, mode=gab.CONV_STREAMING)
        h_out.copy_(out, non_blocking=True)
        stream.synchronize()
        if i >= 5:
            paced.append((time.perf_counter() - t1) * 1e6)
    paced = np.array(paced) if paced else np.array([float("nan")])
    paced_waits, paced_missed = daw.stats()
    daw.close()

    alg = algorithmic_bytes(T, B, L)
    # achieved is priced on the launch period of the timed region (kernel time plus
    # the inter-launch gap): that is what back-to-back buffers actually cost.
    achieved = alg / (period_us * 1e-6) / 1e9
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_file):
        try:
            traffic = json.load(open(pmc_file)).get(
                "conv_split_kernel_bytes_per_launch" if plan.scheme == "split" else "conv_overlap_save_kernel_bytes_per_launch")
        except Exception:
            traffic = None

    result = {
        "metric": "audio_buffers_per_sec",
        "value": world * args.steps / elapsed,
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
                        "%d-sample buffers @ %d Hz per GPU (BASELINE configs[2]%s)"
                        % (L, T, B, FS, "; configs[4]-style channel sharding, %d channels total" % T_total
                           if world > 1 else ""),
            "taps": L, "channels_per_gpu": T, "channels_total": T_total, "buffer_size": B, "fs": FS,
            "mode": "streaming", "tap_cut": plan.scheme,
            "realtime_factor": (world * args.steps / elapsed) * B / FS,
            "p50_round_trip_us": float(np.percentile(rt, 50)),
            "p95_round_trip_us": float(np.percentile(rt, 95)),
            "p50_round_trip_zero_copy_us": float(np.percentile(zc, 50)) if zc is not None else None,
            "p95_round_trip_zero_copy_us": float(np.percentile(zc, 95)) if zc is not None else None,
            "paced_10p667ms": {"p50_round_trip_us": float(np.percentile(paced, 50)),
                               "p95_round_trip_us": float(np.percentile(paced, 95)),
                               "max_round_trip_us": float(paced.max()),
                               "slots": int(paced_waits), "missed_slots": int(paced_missed)},
            "batch_mode_16_buffers_per_launch": {"us_per_buffer": batch_us, "buffers_per_sec": 1e6 / batch_us,
                                                 "alg_GBps": algorithmic_bytes(T, B, L) / batch_us / 1e3},
            "pipelined_two_streams_stateless_kernel": {"us_per_buffer": pipe_us, "buffers_per_sec": 1e6 / pipe_us,
                                                       "alg_GBps": algorithmic_bytes(T, B, L) / pipe_us / 1e3},
            "channel_halves_on_two_streams": None if two_us is None else {
                "us_per_buffer": two_us, "buffers_per_sec": 1e6 / two_us,
                "alg_GBps": algorithmic_bytes(T, B, L) / two_us / 1e3},
            "ir_broadcast_ms": bcast_ms,
            "state_bytes": {"spectra": spectra_bytes, "history": history_bytes},
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "conv_split_kernel" if plan.scheme == "split" else "conv_overlap_save_kernel<true,true>",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "algorithmic_bytes_per_launch": alg,
            "launch_period_us": period_us,
            "kernel_us_event_pairs": kernel_us,
        },
    }

    # ---- CPU baseline: the oracle, one core, bounded sample (rank 0, N = 1) ------------
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        import oracle                     # checker / baseline only — never the product path
        from concurrent.futures import ThreadPoolExecutor
        ir_host = ir_dev.cpu().numpy().reshape(T, L)
        xs = [inputs[i].cpu().numpy().reshape(T, B) for i in range(N_INPUT_BUFFERS)]
        budget = args.cpu_baseline_seconds / 2.0

        # (i) the reference's golden as it runs it: scalar loops, one thread
        hist = np.zeros(T * L, np.float32)
        t1 = time.perf_counter()
        oracle.conv_accel_stream(xs[0].ravel(), ir_host.ravel(), hist, L, B, T)
        first = time.perf_counter() - t1
        n_one = max(1, min(20, int(budget / first) - 1))
        t1 = time.perf_counter()
        for i in range(n_one):
            oracle.conv_accel_stream(xs[(i + 1) % N_INPUT_BUFFERS].ravel(), ir_host.ravel(), hist, L, B, T)
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
            oracle.conv_accel_stream(xcut[i % N_INPUT_BUFFERS][k], irs[k], hists[k], L, B, hi - lo)

        with ThreadPoolExecutor(max_workers=cores) as pool:
            list(pool.map(lambda k: one(k, 0), range(cores)))            # warm
            n_all = max(2, min(200, int(budget * cores / first * 0.8)))
            t1 = time.perf_counter()
            for i in range(n_all):
                list(pool.map(lambda k, i=i: one(k, i + 1), range(cores)))
            dt_all = time.perf_counter() - t1
        result["cpu_baseline"] = {
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
    elif rank == 0:
        result["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(result))
    plan.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
