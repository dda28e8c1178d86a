#!/usr/bin/env python3
"""bench.py — headline benchmark: streaming 4096-tap FIR convolution
(bench_conv1d_accel) x 1024 channels x 512-sample buffers @ 48 kHz on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over the resident batch of synthetic input:
BUFFERS_PER_STEP = 128 consecutive audio buffers (512 new samples for every one of a
rank's 1024 channels) pushed through overlap-save with carried history by ONE launch
of gab_conv_process_batch (conv_split_batch12_kernel: a workgroup of twelve waves owns four channels for
the whole launch and walks the 128 buffers in order).  The inputs are resident in HBM
before the timed region starts — the precondition the metric is quoted under — so
nothing has to cross a kernel boundary between buffers; history carries over from one
step to the next (the stream never restarts).  Results are bit-identical to one
gab_conv_process launch per buffer (tests/), whose rate is reported beside `value`.

With N > 1 every rank owns a 1024-channel shard of an N*1024-channel job: rank 0
generates the whole impulse-response bank, it is broadcast over RCCL/xGMI once (or, --ir-distribution slices,
sent as per-rank rows), each rank transforms its slice; there is no per-buffer collective (channels are
independent), so scaling is weak.  A plain `python bench.py --gpus N` starts the N
ranks itself (torch.distributed.run) before touching any GPU and relays rank 0's line.

One JSON line on rank 0:
  value          = 1024-channel buffers per second, whole job (N * 128 K / max-rank time)
  roofline       = algorithmic bytes per launch / average launch duration (HIP events on the
                   launch stream over the timed region), against 8 TB/s HBM
  parity_checked = after the timed region: sampled channels of the last step's first and last
                   buffer against the float64 direct form (the CPU oracle); mismatch => exit 1
  round_trip     = (config) p50 / p95 of one buffer pinned host -> GPU -> pinned host through gab_conv_round_trip
                   (returns on the launch's own completion event), against the link floor measured in the same run
  one_buffer_per_doorbell = (config) the resident engine: pipelined rate, and under in_flight_1 the time from doorbell
                   to `completed` with ONE buffer in flight (the reference's iteration), back to back and paced
  a bit-identity flag of either leg that is False => exit 1
  cpu_baseline   = the CPU oracle (the reference golden extended with history), timed on a
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
BUFFERS_PER_STEP = 128          # the resident input batch (256 MiB), one launch (64: 1.5 % slower per buffer; 192: the same)
CLOCK_WARM_STEPS = 500          # untimed, besides --warmup: ~0.2 s of the same launches; the part's clocks settle over the first ~40 ms of sustained load
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
PARITY_CHANNELS = 16            # sampled per checked buffer
PARITY_TOL = 1e-5               # of the stream's peak (north_star: 1e-5 relative for float DSP)
TRAFFIC_SOURCE = "profiles/r06_conv_batch_pmc_means.json"


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
    # (GAB_BENCH_DRYRUN_TRACKS / _TAPS: 1024 / 4096 walk BASELINE configs[4]'s shape arithmetic with 8 ranks — 8192
    # channels, a 128 MiB bank, 1024-channel slices at global indices: tests/test_bench_launcher.py)
    T, L = int(os.environ.get("GAB_BENCH_DRYRUN_TRACKS", "8")), int(os.environ.get("GAB_BENCH_DRYRUN_TAPS", "64"))
    gran = sharding.shard_granule("Conv1D_accel")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    t0 = time.perf_counter()
    ir = sharding.broadcast_ir_bank(L, T * world, rank, world, torch.device("cpu"), dist if world > 1 else None,
                                    granule=gran, distribution=args.ir_distribution)
    bcast_ms = (time.perf_counter() - t0) * 1e3
    lo, hi = sharding.shard_range(rank, world, T * world, gran)
    ok = bool(hi - lo == T and np.array_equal(ir.numpy().ravel(), __import__("gpuaudiobench_amd").harness.conv_accel_ir(L, hi - lo, lo, T * world)))
    # the rank's input rows and its output columns at the job's global indices (no device: the rows stand in for outputs)
    x = sharding.shard_noise(T * world, 4, rank, world, seed=42, granule=gran)
    ok = ok and x.shape == (T, 4)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    elapsed = time.perf_counter() - t0 + 1e-9
    received = sharding.bank_bytes_received(L, T * world, rank, world, gran, args.ir_distribution)
    per_rank = sharding.gather_per_rank([0.0, bcast_ms, received], rank, world, dist if world > 1 else None)
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
                          "config": {"workload": "dry run", "ir_broadcast_ms": bcast_ms, "ir_distribution": args.ir_distribution,
                                     "channels_total": T * world, "taps": L, "channels_per_rank": T,
                                     "ir_slices_match_global_bank": ok,
                                     "per_rank": {"us_per_buffer": [None] * world,
                                                  "ir_broadcast_ms": [r[1] for r in per_rank],
                                                  "ir_bytes_received": [int(r[2]) for r in per_rank]}}}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps of %d buffers each" % BUFFERS_PER_STEP)
    ap.add_argument("--warmup", type=int, default=10, help="untimed steps (besides the fixed clock warm-up)")
    ap.add_argument("--buffers-per-step", type=int, default=BUFFERS_PER_STEP, help="buffers per launch (>= 9)")
    ap.add_argument("--clock-warm-steps", type=int, default=CLOCK_WARM_STEPS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-legs", action="store_true", help="only the timed region (profiling runs)")
    ap.add_argument("--batch-sizes", action="store_true",
                    help="also time 8 / 16 / 32 buffers per launch (off by default: those launches carry the headline "
                         "kernel's name and would mix into a profiler's per-kernel average of the 128-buffer launches)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    ap.add_argument("--ir-distribution", choices=("broadcast", "slices"), default="broadcast",
                    help="N > 1: every rank receives the whole impulse-response bank (north_star's RCCL broadcast) or only its rows")
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
    # Under a launcher (WORLD_SIZE set) the process group is created even for ONE rank, so that a one-GPU
    # box walks the RCCL path too: communicator, bank broadcast, barrier, all-reduce of the timing.
    grouped = "WORLD_SIZE" in os.environ
    backend = None
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        backend = "gloo" if rehearse else "nccl"
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import gpuaudiobench_amd as gab           # raises if libgab_hip.so is missing

    T, B, L = TRACKS_PER_GPU, BUFSIZE, TAPS
    T_total = T * world
    NB = max(9, args.buffers_per_step)

    # ---- impulse-response bank: generated once, broadcast over RCCL -------------
    from gpuaudiobench_amd import sharding
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gran = sharding.shard_granule("Conv1D_accel")
    ir_dev = sharding.broadcast_ir_bank(L, T_total, rank, world, dev, dist if grouped else None, granule=gran,
                                        distribution=args.ir_distribution)
    torch.cuda.synchronize()
    bcast_ms = (time.perf_counter() - t0) * 1e3 if grouped else None

    plan = gab.ConvPlan(T, B, L)
    plan.set_ir(ir_dev)
    spectra_bytes, history_bytes = plan.state_bytes()
    if plan.scheme != "split":
        raise SystemExit("bench.py: the headline shape must run the split cut")

    # ---- synthetic input: the reference's noise generator, this rank's channels --
    host_in = [sharding.shard_noise(T_total, B, rank, world, seed=42 + i, granule=gran) for i in range(NB)]
    xb = torch.cat([torch.from_numpy(x).reshape(-1) for x in host_in]).to(dev)        # [NB][T*B], resident
    yb = torch.empty_like(xb)                                             # [NB][B*T]
    stream = torch.cuda.current_stream()
    batch_args = plan.prepare_batch(xb, NB, yb)

    def run_steps(k):                 # one launch per step
        for _ in range(k):
            plan.launch_batch(batch_args)

    run_steps(args.clock_warm_steps)
    run_steps(args.warmup)

    # ---- timed region -------------------------------------------------------------
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record(stream)
    run_steps(args.steps)
    e1.record(stream)
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    n_buffers = args.steps * NB
    launch_us = e0.elapsed_time(e1) * 1e3 / args.steps       # average launch period, device clock

    # every rank's own launch period, bank-distribution time and bytes received, on rank 0's line: the first real N > 1 run
    # then names a straggler rank or a slow broadcast by itself (the job's `value` is the max-over-ranks time below)
    received = sharding.bank_bytes_received(L, T_total, rank, world, gran, args.ir_distribution)
    per_rank = sharding.gather_per_rank([launch_us / NB, bcast_ms or 0.0, received], rank, world, dist if grouped else None,
                                        device="cpu" if rehearse or not grouped else dev)
    if grouped:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    wall_us = elapsed * 1e6 / n_buffers

    # ---- in-run parity: the timed launches' own output against the float64 direct form --------
    parity = parity_check(plan, host_in, yb, ir_dev, T, B, L, NB, np, torch) if rank == 0 else None

    alg = algorithmic_bytes(T, B, L)
    side = {}
    power_state = None
    if not args.no_side_legs and world == 1:           # the one-GPU real-time path: reported at N = 1 only
        power_state = power_state_under_load(run_steps, torch)
        side = side_legs(gab, plan, ir_dev, host_in, xb, stream, dev, T, B, L, np, torch, args.batch_sizes)

    traffic, pmc = None, {}
    pmc_file = os.path.join(ROOT, TRAFFIC_SOURCE)
    if os.path.exists(pmc_file):
        try:
            pmc = json.load(open(pmc_file))
            traffic = pmc.get("hbm_traffic_bytes_per_launch")
        except Exception:
            traffic, pmc = None, {}
    cnt = pmc.get("counters_mean_per_launch", {})
    valu_share = (cnt["SQ_ACTIVE_INST_VALU"] / cnt["SQ_WAVE_CYCLES"]) if cnt.get("SQ_WAVE_CYCLES") and cnt.get("SQ_ACTIVE_INST_VALU") else None

    achieved = alg * NB / (launch_us * 1e-6) / 1e9
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
        "value_is": "throughput with %d buffers resident in HBM per launch (gab_conv_process_batch); the rate with ONE launch per "
                    "buffer and the pinned-host round trip of one buffer are config.one_launch_per_buffer / config.round_trip" % NB,
        "dtype": "f32",
        "data": "synthetic" if not rehearse else "synthetic; REHEARSAL: all ranks share device 0 over gloo",
        "config": {
            "workload": "bench_conv1d_accel streaming overlap-save: %d-tap IR x %d channels x "
                        "%d-sample buffers @ %d Hz per GPU (BASELINE configs[2]%s); one step = %d consecutive buffers, "
                        "resident in HBM, one launch"
                        % (L, T, B, FS, "; configs[4]-style channel sharding, %d channels total" % T_total
                           if world > 1 else "", NB),
            "taps": L, "channels_per_gpu": T, "channels_total": T_total, "buffer_size": B, "fs": FS,
            "buffers_per_step": NB, "us_per_buffer": wall_us,
            "mode": "streaming", "tap_cut": plan.scheme,
            "launch_path": "gab_conv_process_batch: %d buffers per launch, history carried across launches" % NB,
            "working_set": "38 MB of plan state (Infinity-Cache-resident) + %d MiB of input and %d MiB of output per step "
                           "streamed through HBM" % (NB * T * B * 4 >> 20, NB * T * B * 4 >> 20),
            "clock_warm_steps": args.clock_warm_steps,
            "realtime_factor": (world * n_buffers / elapsed) * B / FS,
            "ir_broadcast_ms": bcast_ms, "ir_distribution": args.ir_distribution if grouped else None, "collective_backend": backend,
            "per_rank": {"us_per_buffer": [r[0] for r in per_rank],          # device clock (HIP events), each rank's own launches
                         "ir_broadcast_ms": [r[1] for r in per_rank] if grouped else None,
                         "ir_bytes_received": [int(r[2]) for r in per_rank]},
            "state_bytes": {"spectra": spectra_bytes, "history": history_bytes},
            # clocks and socket power under the timed launches (N = 1): the launch draws the board's power limit, the shader clock
            # a box reaches there is what its line's value follows (profiles/r06_box_clocks.txt)
            "power_state": power_state,
            "env": {k: v for k, v in sorted(os.environ.items()) if k.startswith("GAB_")},
        },
        "roofline": {
            # `bound` names the roofline the contract prices against (algorithmic bytes over the 8 TB/s HBM peak);
            # what actually limits the kernel is `limited_by` / `bound_measured`: not memory
            "bound": "hbm",
            "limited_by": "instruction issue and barrier waits of the far role's transform chains (two four-wave groups, three waves per SIMD; "
                          "every barrier's last arrival is a far wave: profiles/r06_batch12_stamps.txt) at the board's power limit (config.power_state), not HBM: see bound_measured and hbm_frac_measured",
            "kernel": "conv_split_batch12_kernel",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "frac_wall": alg / (wall_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "traffic": traffic,
            # what the memory system really carried, against the same peak: `frac` prices the ALGORITHMIC bytes (SURVEY 8d:
            # every tap and every history sample a 4096-tap FIR depends on), most of which this kernel keeps in LDS, registers
            # and the Infinity Cache, so frac is a rate of useful work, not HBM utilisation
            "hbm_frac_measured": (traffic / (launch_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if traffic is not None and NB == BUFFERS_PER_STEP else None,
            "bound_measured": ("barriers and latency of three resident waves per SIMD: VALU busy %.0f %% of wave-cycles, HBM-side traffic %.2f x "
                               "algorithmic (counters of %s)" % (100 * valu_share, traffic / (alg * BUFFERS_PER_STEP), TRAFFIC_SOURCE))
                              if traffic is not None and valu_share is not None else None,
            "traffic_source": TRAFFIC_SOURCE + " (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes over the same "
                              "launch; not measured in this run)" if traffic is not None else None,
            "algorithmic_bytes_per_buffer": alg,
            "algorithmic_bytes_per_launch": alg * NB,
            "buffers_per_launch": NB,
            "launch_us": launch_us,
            "how": "achieved = algorithmic bytes per launch (%d buffers x 4*T*(2B+2L)) / average launch period "
                   "(two HIP events on the launch stream around the %d timed launches); rocprofv3's average "
                   "duration of conv_split_batch12_kernel is the same quantity" % (NB, args.steps),
        },
        "parity_checked": parity,
    }
    result["config"].update(side)

    # ---- CPU baseline: the oracle on this box's cores, bounded sample (rank 0, N = 1) ----
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args, ir_dev, host_in, T, B, L, np)
    elif rank == 0:
        result["cpu_baseline"] = None

    plan.close()
    # ---- library yardstick (tools only, a child process; never on the product path): the
    # reference's pipeline on rocFFT, same GPU, same shape (tools/ubench/library_baseline.cpp)
    if world == 1 and rank == 0 and not args.no_side_legs:
        result["library_baseline"] = library_baseline(T, B, L)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if grouped:
        dist.destroy_process_group()
    if parity is not None and not parity["ok"]:
        sys.stderr.write("bench.py: PARITY CHECK FAILED: %s\n" % json.dumps(parity))
        return 1
    # the side legs compare their outputs bit for bit with the launches `value` is measured on: a False there is a wrong
    # result on the real-time path, not a footnote
    wrong = [leg + "." + k for leg in ("round_trip", "one_buffer_per_doorbell") for k, v in (side.get(leg) or {}).items()
             if k.startswith("bit_identical") and v is False]
    if wrong:
        sys.stderr.write("bench.py: BIT-IDENTITY CHECK FAILED: %s\n" % ", ".join(wrong))
        return 1
    return 0


def parity_check(plan, host_in, yb, ir_dev, T, B, L, NB, np, torch):
    """Checker use of the oracle (float64 direct form): PARITY_CHANNELS sampled channels of the LAST
    buffer of the last timed step (its window lies inside the step) and of its FIRST buffer (its
    window reaches back into the previous launch: the history carried across launches).  A 4096-tap
    FIR sees 8 blocks of history, so the direct form is fed exactly the 9 blocks that matter."""
    import oracle
    blocks = L // B
    rng = np.random.default_rng(2024)
    chans = np.sort(rng.choice(T, size=min(PARITY_CHANNELS, T), replace=False))
    ir = ir_dev.view(T, L)[torch.from_numpy(chans).to(ir_dev.device)].cpu().numpy()
    out = yb.view(NB, B, T)
    worst, peak = 0.0, 0.0
    for last in (NB - 1, 0):
        hist = np.zeros(len(chans) * L, np.float32)
        ref = None
        for k in range(last - blocks, last + 1):
            x = host_in[k % NB].reshape(T, B)[chans]
            ref = oracle.conv_accel_stream(np.ascontiguousarray(x).ravel(), ir.ravel(), hist, L, B, len(chans), f64=True)
        got = out[last][:, torch.from_numpy(chans).to(yb.device)].cpu().numpy().astype(np.float64)
        ref = ref.reshape(B, len(chans))
        worst = max(worst, float(np.abs(got - ref).max()))
        peak = max(peak, float(np.abs(ref).max()))
    err = worst / peak if peak > 0 else float("inf")
    return {"channels": int(len(chans)), "buffers": "first and last buffer of the last timed step",
            "against": "oracle orc_conv_accel_stream_f64 (float64 direct form) fed the 9 input blocks each output depends on",
            "max_err_of_peak": err, "tolerance": PARITY_TOL, "ok": bool(err <= PARITY_TOL and peak > 0)}


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


def power_state_under_load(run_steps, torch, limit_s=8.0):
    """Shader / memory clock, socket power and junction temperature WHILE the timed launches run: rocm-smi as a child process
    beside an untimed loop of the same launches.  `value` sits at the board's power limit (profiles/r06_box_clocks.txt), so the
    clock a box reaches there is what separates one box's line from another's.  None where rocm-smi is missing or says nothing."""
    import re
    import shutil
    import subprocess
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return None
    # Not under a profiler (its tool library, preloaded, would come up again in the child of a process that holds the GPU): the
    # field is None there.  Otherwise the child inherits the environment as it is, and the interpreter of rocm-smi's script is
    # named so that no `env` hop stands between.
    profiler = ("rocprof", "roctracer", "rocprofiler")
    if any(t in os.environ.get("LD_PRELOAD", "").lower() for t in profiler) or os.environ.get("HSA_TOOLS_LIB") or \
            any(k.startswith(("ROCP_", "ROCPROF", "ROCTRACER")) for k in os.environ):
        return None
    cmd = [os.path.realpath(exe), "--showclocks", "--showpower", "--showtemp"]
    try:
        with open(cmd[0], "rb") as f:
            if f.read(2) == b"#!":
                cmd.insert(0, sys.executable)
    except OSError:
        return None
    try:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 5.0:            # the load is up and the sensor's power average has settled (it lags by seconds)
            run_steps(8)
            torch.cuda.synchronize()
        child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        t0 = time.perf_counter()
        while child.poll() is None and time.perf_counter() - t0 < limit_s:
            run_steps(8)
            torch.cuda.synchronize()
        if child.poll() is None:
            child.kill()
        text = child.communicate(timeout=2)[0]
    except Exception:
        return None
    gpu0 = "\n".join(l for l in text.splitlines() if l.startswith("GPU[0]"))

    def num(pattern):
        m = re.search(pattern, gpu0)
        return float(m.group(1)) if m else None
    got = {"sclk_mhz": num(r"sclk clock level: *\d+: *\((\d+)Mhz\)"), "mclk_mhz": num(r"mclk clock level: *\d+: *\((\d+)Mhz\)"),
           "socket_power_w": num(r"Power \(W\): *([\d.]+)"), "junction_c": num(r"Sensor junction\) \(C\): *([\d.]+)")}
    if all(v is None for v in got.values()):
        return None
    got["how"] = "rocm-smi (GPU[0]) sampled once, five seconds into an untimed loop of the timed launches, after the timed region"
    return got


def side_legs(gab, plan, ir_dev, host_in, xb, stream, dev, T, B, L, np, torch, batch_sizes=False):
    """Rates that are NOT `value`, each with a fixed iteration count (independent of --steps)."""
    NB = len(host_in)
    alg = algorithmic_bytes(T, B, L)
    res = {}
    out = torch.empty(T * B, dtype=torch.float32, device=dev)
    inputs = [xb[i * T * B:(i + 1) * T * B] for i in range(NB)]

    # ---- the same buffers as ONE launch per buffer (what a caller that receives its audio one
    # buffer at a time does), HIP events around 4000 launches
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
    res["one_launch_per_buffer"] = {
        "kernel": "conv_split_kernel",
        "us_per_buffer": one_us, "buffers_per_sec": 1e6 / one_us, "alg_GBps": alg / one_us / 1e3,
        "frac": alg / one_us / 1e3 / HBM_PEAK_GBS, "launches": 4000}

    # ---- the same buffers ARRIVING one at a time at a resident launch (gab_conv_engine_*): the host rings the doorbell once
    # per buffer and keeps at most `ahead` in flight; the ring is the resident batch.  HIP events around start .. stop on a side
    # stream around the whole launch; the output ring checked bit for bit against batch launches over the same buffers.
    ahead, passes = 16, 64
    # The engine's ring is HALF a step (64 slots: 128 MiB in, 128 MiB out), and every batch launch of this leg — the clock
    # warm-up, the reference — is a launch of a whole step's size over the ring's contents twice: rocprofv3's per-kernel
    # average of the command then averages launches of ONE size (conv_split_batch12_kernel: `value`'s).
    NB_step, NB = NB, (NB // 2 if NB // 2 > ahead else NB)
    xb_step, xb = xb, xb[:NB * T * B]
    per = NB_step // NB                                # ring passes per reference launch (2, or 1 for a small step)
    x2 = torch.cat([xb] * per) if per > 1 else xb
    y2 = torch.empty_like(x2)
    ref = gab.ConvPlan(T, B, L, scheme="split")
    ref.set_ir(ir_dev)
    eplan = gab.ConvPlan(T, B, L, scheme="split")
    eplan.set_ir(ir_dev)
    in_ring, out_ring = eplan.engine_rings(NB)
    in_ring.copy_(xb.view(NB, T * B))
    warm = plan.prepare_batch(x2, per * NB, y2)       # the clocks settle over ~40 ms of sustained load: the same warm-up as `value`
    for _ in range(300 // per):
        plan.launch_batch(warm)
    torch.cuda.synchronize()
    for _ in range(passes // per):                     # (the warm-up used the buffer: the reference again)
        ref.process_batch(x2, per * NB, out=y2)
    ref.close()
    torch.cuda.synchronize()
    yref = y2[(per - 1) * NB * T * B:]                 # what the last pass over the ring leaves
    side = torch.cuda.Stream()
    # an untimed first launch, as `value` has its warm-up steps: the first resident launch of a process runs ~10 % slower
    # than the following ones (tools/engine_conv.py: 6.67 us per buffer, then 6.05-6.14)
    eng_warm_passes = passes                          # (a 256-buffer first launch leaves 6.45 us; one of the same length 6.1)
    eplan.engine_start(NB, stream=side)
    eplan.engine_feed(eng_warm_passes * NB, ahead=ahead)
    eplan.engine_stop()
    side.synchronize()
    eplan.reset()                                     # the timed launch starts the stream anew, as the reference did
    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g0.record(side)
    eplan.engine_start(NB, stream=side)
    eplan.engine_feed(passes * NB, ahead=ahead)
    eplan.engine_stop()
    g1.record(side)
    side.synchronize()
    eng_us = g0.elapsed_time(g1) * 1e3 / (passes * NB)
    res["one_buffer_per_doorbell"] = {
        "entry": "gab_conv_engine_start / _feed / _stop: one resident launch (conv_split_engine12_kernel), the host publishes ONE "
                 "buffer per ring of the doorbell and keeps at most %d in flight" % ahead,
        "us_per_buffer": eng_us, "buffers_per_sec": 1e6 / eng_us, "alg_GBps": alg / eng_us / 1e3,
        "frac": alg / eng_us / 1e3 / HBM_PEAK_GBS, "buffers": passes * NB, "ahead": ahead, "ring_slots": NB,
        "untimed_first_launch_buffers": eng_warm_passes * NB,
        "bit_identical_to_batch_launches": bool(torch.equal(out_ring.reshape(-1).view(torch.int32), yref.view(torch.int32)))}
    # The same engine with ONE buffer in flight (the reference's iteration takes one buffer and returns it,
    # cuda/bench_conv1d_accel.cu:258-304): ring the doorbell with the flush rung, wait for `completed` to count THAT buffer,
    # only then the next.  Host clock around each { submit, wait } (gab_conv_engine_feed_one_in_flight); the same buffers,
    # so the output ring must again be what the batch launches left.  Then paced: one buffer per 512/48000 s slot.
    eplan.reset()
    out_ring.zero_()
    torch.cuda.synchronize()
    lat = np.zeros(passes * NB, np.float32)
    eplan.engine_start(NB, stream=side)
    eplan.engine_feed_one_in_flight(passes * NB, lat)
    eplan.engine_stop()
    side.synchronize()
    lone_same = bool(torch.equal(out_ring.reshape(-1).view(torch.int32), yref.view(torch.int32)))
    eplan.engine_start(NB, stream=side)
    daw1 = gab.harness.DawSim(buffer_seconds=float(B) / FS, mode="spin")
    paced1 = np.zeros(1, np.float32)
    paced_lat = []
    for i in range(65):
        daw1.wait()
        eplan.engine_feed_one_in_flight(1, paced1)
        if i >= 5:
            paced_lat.append(float(paced1[0]))
    eplan.engine_stop()
    side.synchronize()
    daw1.close()
    lat = lat[100:]
    res["one_buffer_per_doorbell"]["in_flight_1"] = {
        "what": "ONE buffer in flight: doorbell with the flush rung -> completed counts that buffer -> next (input and output in the "
                "engine's device rings; the link is config.round_trip's business); us from publish to completed, host clock",
        "p50_us": float(np.percentile(lat, 50)), "p95_us": float(np.percentile(lat, 95)), "max_us": float(lat.max()),
        "buffers": int(len(lat)),
        "paced_10p667ms": {"p50_us": float(np.percentile(paced_lat, 50)), "max_us": float(np.max(paced_lat)), "buffers": len(paced_lat)},
    }
    res["one_buffer_per_doorbell"]["bit_identical_in_flight_1"] = lone_same
    # The composed real-time path through the engine (gab_conv_engine_round_trip): pinned host -> ring slot (engine copy),
    # doorbell with the flush rung, ring slot -> pinned host (engine copy), ONE buffer in flight — the reference's whole
    # iteration (cuda/bench_base.cu:30-42 around bench_conv1d_accel.cu:258-304) on the resident launch.  Its two link legs
    # do not overlap with the transform, so config.round_trip (gab_conv_round_trip) stays the faster call; this is the
    # per-buffer engine's stated number.  Outputs against one ordinary launch per buffer, bit for bit.
    eplan.reset()
    chk = gab.ConvPlan(T, B, L, scheme="split")
    chk.set_ir(ir_dev)
    # (the reference outputs BEFORE the engine takes the device: at 1024 channels no other kernel runs beside it)
    wants = [chk.process(xb_step[(i % NB_step) * T * B:(i % NB_step + 1) * T * B]).cpu() for i in range(24)]
    chk.close()
    torch.cuda.synchronize()
    h_in, h_out = torch.empty(T * B).pin_memory(), torch.empty(T * B).pin_memory()
    eplan.engine_start(NB, stream=side)
    rt_lat, rt_same = [], True
    for i in range(260):
        h_in.copy_(torch.from_numpy(host_in[i % NB_step]).reshape(-1))
        t0 = time.perf_counter()
        eplan.engine_round_trip(h_in, h_out)
        rt_lat.append((time.perf_counter() - t0) * 1e6)
        if i < len(wants):                            # (the first buffers, incl. the history window filling; later ones are timing only)
            rt_same = rt_same and bool(torch.equal(wants[i].view(torch.int32), h_out.view(torch.int32)))
    eplan.engine_stop()
    side.synchronize()
    rt_lat = np.array(rt_lat[60:])
    res["one_buffer_per_doorbell"]["round_trip"] = {
        "entry": "gab_conv_engine_round_trip: pinned host -> ring slot -> doorbell (flush) -> completed -> ring slot -> pinned host, "
                 "ONE buffer in flight; host clock around the call",
        "p50_us": float(np.percentile(rt_lat, 50)), "p95_us": float(np.percentile(rt_lat, 95)), "max_us": float(rt_lat.max()),
        "calls": int(len(rt_lat))}
    res["one_buffer_per_doorbell"]["bit_identical_round_trip"] = rt_same
    eplan.close()
    plan.reset()
    NB, xb = NB_step, xb_step

    # ---- batch size: how the per-launch cost (first window, drain, boundary) amortises
    sizes = {}
    for nb in ((8, 16, 32) if batch_sizes else ()):
        x = xb[:nb * T * B]
        y = torch.empty_like(x)
        a = plan.prepare_batch(x, nb, y)
        for _ in range(10):
            plan.launch_batch(a)
        eb0, eb1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 1600 // nb
        eb0.record(stream)
        for _ in range(reps):
            plan.launch_batch(a)
        eb1.record(stream)
        torch.cuda.synchronize()
        us = eb0.elapsed_time(eb1) * 1e3 / (reps * nb)
        sizes[str(nb)] = {"us_per_buffer": us, "frac": alg / us / 1e3 / HBM_PEAK_GBS, "launches": reps}
    if batch_sizes:
        res["batch_buffers_per_launch"] = sizes
    plan.reset()

    # ---- round trip, one buffer in flight, host clock around each call (the second half of the metric).
    # The product's real-time entry is gab_conv_round_trip on a classic-cut plan: one engine upload consumed by the
    # kernel as it lands, the outputs drained to the pinned buffer per channel group while later groups still arrive.
    h_in = torch.from_numpy(gab.harness.noise(T * B, seed=7)).pin_memory()
    h_out = torch.empty(T * B, dtype=torch.float32).pin_memory()
    d_in = torch.empty(T * B, dtype=torch.float32, device=dev)
    rplan = gab.ConvPlan(T, B, L, scheme="classic")
    rplan.set_ir(ir_dev)
    rt_args = rplan.prepare_round_trip(h_in, h_out)

    def timed(fn, n, skip):
        ts = []
        for i in range(n):
            t1 = time.perf_counter()
            fn()
            if i >= skip:
                ts.append((time.perf_counter() - t1) * 1e6)
        return np.array(ts)

    def by_copies():            # the reference's iteration: H2D copy, kernel, D2H copy, synchronize
        d_in.copy_(h_in, non_blocking=True)
        rplan.process(d_in, out=out, mode=gab.CONV_STREAMING)
        h_out.copy_(out, non_blocking=True)
        stream.synchronize()

    def by_kernel():            # the kernel reads and writes the pinned buffers itself
        rplan.process(h_in, out=h_out, mode=gab.CONV_STREAMING)
        stream.synchronize()

    rt = timed(lambda: rplan.launch_round_trip(rt_args), 520, 20)
    # the call's own output against device-buffer launches of the same cut on the same stream of inputs
    cplan = gab.ConvPlan(T, B, L, scheme="classic")
    cplan.set_ir(ir_dev)
    rplan.reset()
    same = True
    for i in range(10):
        rplan.launch_round_trip(rt_args)
        ref = cplan.process(h_in.to(dev), mode=gab.CONV_STREAMING)
        same = same and bool(torch.equal(ref.cpu().view(torch.int32), h_out.view(torch.int32)))
    cplan.close()
    cp = timed(by_copies, 220, 20)
    zc = timed(by_kernel, 220, 20)

    # link rate per direction (engine copies of 64 MiB, device clock): the floor a duplex round trip cannot beat
    big_h = torch.empty(16 << 20, dtype=torch.float32).pin_memory()
    big_d = torch.empty(16 << 20, dtype=torch.float32, device=dev)
    rates = {}
    for name, (dst, src) in (("h2d", (big_d, big_h)), ("d2h", (big_h, big_d))):
        dst.copy_(src, non_blocking=True)
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea.record(stream)
        for _ in range(4):
            dst.copy_(src, non_blocking=True)
        eb.record(stream)
        torch.cuda.synchronize()
        rates[name] = 4 * big_h.numel() * 4 / (ea.elapsed_time(eb) * 1e-3) / 1e9
    del big_h, big_d
    bytes_each_way = T * B * 4
    floor_us = bytes_each_way / (min(rates.values()) * 1e3)
    p50 = float(np.percentile(rt, 50))
    res["p50_round_trip_us"] = p50
    res["p95_round_trip_us"] = float(np.percentile(rt, 95))
    res["round_trip"] = {
        "entry": "gab_conv_round_trip (classic cut; conv_round_trip_kernel)",
        "p50_us": p50, "p95_us": float(np.percentile(rt, 95)), "max_us": float(rt.max()), "calls": int(len(rt)),
        "bytes_each_way": bytes_each_way,
        "link_GBps": {"h2d": rates["h2d"], "d2h": rates["d2h"],
                      "how": "4 engine copies of 64 MiB per direction, HIP events, this run"},
        "floor_us": floor_us,
        "floor": "bytes each way / the slower direction's rate: both directions fully overlapped, no launch, no arithmetic",
        "frac": floor_us / p50,
        "bit_identical_to_device_buffer_launches": same,
        "by_copies": {"p50_us": float(np.percentile(cp, 50)), "p95_us": float(np.percentile(cp, 95)),
                      "what": "H2D copy, kernel, D2H copy on one stream, synchronize (the reference's iteration)"},
        "by_kernel": {"p50_us": float(np.percentile(zc, 50)), "p95_us": float(np.percentile(zc, 95)),
                      "what": "the kernel reads and writes the pinned buffers itself, nothing overlapped"},
    }

    # ---- the same call under DAW pacing: one buffer per 512/48000 s slot, device idle between
    def paced_calls():
        ts = []
        daw = gab.harness.DawSim(buffer_seconds=float(B) / FS, mode="spin")
        for i in range(105):
            daw.wait()
            t1 = time.perf_counter()
            rplan.launch_round_trip(rt_args)
            if i >= 5:
                ts.append((time.perf_counter() - t1) * 1e6)
        waits, missed = daw.stats()
        daw.close()
        ts = np.array(ts)
        return {"p50_round_trip_us": float(np.percentile(ts, 50)), "p95_round_trip_us": float(np.percentile(ts, 95)),
                "max_round_trip_us": float(ts.max()), "slots": int(waits), "missed_slots": int(missed)}

    res["paced_10p667ms"] = paced_calls()
    # the same slots with eight idle waves left on the device between calls (gab_conv_round_trip_keep_warm: opt-in)
    rplan.round_trip_keep_warm(True)
    rplan.launch_round_trip(rt_args)
    res["paced_10p667ms"]["keep_warm"] = dict(paced_calls(), what="gab_conv_round_trip_keep_warm(plan, 1): every call ends by kicking a resident "
                                              "launch of eight sleeping waves (one per XCD), which ends by itself about 0.1 s (eight buffer periods) after the last call")
    # where the eight waves landed (gab_keep_warm_placement): a run in which they buy nothing classifies itself
    res["paced_10p667ms"]["keep_warm"]["placement"] = gab.ops.placement_summary(rplan.round_trip_keep_warm_placement())
    rplan.round_trip_keep_warm(False)
    rplan.close()
    return res


def cpu_baseline(args, ir_dev, host_in, T, B, L, np):
    import oracle                     # checker / baseline only — never the product path
    from concurrent.futures import ThreadPoolExecutor
    from gpuaudiobench_amd import sharding
    NB = 8                              # distinct buffers the sample cycles through
    ir_host = ir_dev.cpu().numpy().reshape(T, L)
    xs = [host_in[i].reshape(T, B) for i in range(NB)]
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
