"""The harness end to end on the GPU: every registry name sets up, runs and
validates against its CPU golden; the gpubench driver speaks the reference CLI."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "gpuaudiobench_amd", "gpubench")

NAMES = ["NoOp", "gain", "GainStats", "datacopy0199", "datacopy2080", "datacopy5050",
         "datacopy8020", "datacopy9901", "FFT1D", "IIRFilter", "Conv1D", "Conv1D_accel",
         "ModalFilterBank", "DWG1DNaive", "DWG1DAccel", "FDTD3D", "RndMemRead"]


@pytest.fixture(scope="module")
def gab():
    import torch
    assert torch.cuda.is_available()
    import gpuaudiobench_amd as g
    return g


@pytest.mark.parametrize("name", NAMES)
def test_every_registered_benchmark_validates(gab, name):
    cfg = dict(n_tracks=128, buffer_size=512)
    if name == "FDTD3D":
        cfg.update(n_tracks=16, buffer_size=64)       # 52^3 x 192 steps; host check of every sample
    b = gab.Benchmark(name, **cfg)
    b.setup()
    r = b.run(iterations=4, warmup=2)
    assert r.iterations == 4 and r.mean_ms > 0 and r.min_ms <= r.median_ms <= r.max_ms
    v, text = b.validate()
    assert v.status == 0, (name, text, v.max_error)
    assert b.algorithmic_bytes() > 0
    lat = b.latencies()
    assert lat.size == 4 and np.all(lat > 0)
    b.close()


@pytest.mark.parametrize("name", NAMES)
def test_every_registered_benchmark_paced_with_keep_warm(gab, name):
    """--dawsim --keepWarm: eight idle waves stay on the device for the length of the run (gab_keep_warm, kicked after
    every iteration).  Nothing an iteration does may wait for them: every benchmark still validates, and a run lasts
    its slots, not its slots plus an idle limit per iteration."""
    import time
    cfg = dict(n_tracks=128, buffer_size=512)
    if name == "FDTD3D":
        cfg.update(n_tracks=16, buffer_size=64)
    b = gab.Benchmark(name, **cfg)
    b.setup()
    slot = 0.004
    b.set_dawsim(buffer_seconds=slot, mode="spin")
    b.set_keep_warm(True)
    b.run(iterations=2, warmup=1)                       # (first-use costs of this benchmark's kernels)
    t0 = time.perf_counter()
    r = b.run(iterations=6, warmup=2)
    elapsed = time.perf_counter() - t0
    v, text = b.validate()
    assert v.status == 0, (name, text, v.max_error)
    assert r.median_ms < 40.0, (name, r.median_ms)      # an iteration that waited for the idle limit (50 ms) would show here
    waits, missed = b.dawsim_stats()
    assert waits == 8
    assert elapsed < 8 * max(slot, r.max_ms * 1e-3) + 0.5, (name, elapsed)
    b.close()


@pytest.mark.parametrize("mode", ["stream", "stateless"])
def test_conv_accel_c3_through_the_harness(gab, mode):
    b = gab.Benchmark("Conv1D_accel", n_tracks=1024, buffer_size=512, ir_length=4096,
                      conv_mode=gab.CONV_STREAMING if mode == "stream" else gab.CONV_STATELESS)
    b.setup()
    r = b.run(iterations=20, warmup=3)
    v, text = b.validate()
    assert v.status == 0 and v.max_error <= 1e-5, text
    T, B, L = 1024, 512, 4096
    expect = 4 * T * (2 * B + 2 * L) if mode == "stream" else 4 * T * 3 * B
    assert b.algorithmic_bytes() == expect                   # 37 748 736 / 6 291 456
    assert r.gpu_median_ms > 0
    b.close()


def test_fdtd_128_grid_prefix_validates(gab):
    b = gab.Benchmark("FDTD3D", n_tracks=8, buffer_size=16, fdtd_grid=128)
    b.setup()
    b.run(iterations=1, warmup=0)
    v, text = b.validate()
    assert v.status == 0, text
    assert v.max_error == 0.0, text                          # same arithmetic on both sides
    b.close()


def run_driver(*args):
    return subprocess.run([DRIVER, *args], capture_output=True, text=True, timeout=600)


def test_driver_list_and_help():
    r = run_driver("--list")
    assert r.returncode == 0
    assert r.stdout.split("\n")[2:19] == NAMES
    assert "Usage: gpubench [options]" in run_driver("--help").stdout


def test_driver_runs_gain_and_writes_reference_outputs(tmp_path):
    csv = str(tmp_path / "r.csv")
    r = run_driver("--benchmark", "gain", "--nRuns", "7", "--nTracks", "64", "--outputfile", csv)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Validation passed for gain" in r.stdout and "gain benchmark completed successfully!" in r.stdout
    assert "=== Gain Benchmark Results ===" in r.stdout           # benchmark_name_, not the registry key
    assert os.path.exists("/tmp/Gain_latencies.txt")              # cuda/bench_base.cu:120-127
    rows = open(csv).read().splitlines()
    assert rows[0].startswith("benchmark,fs,bufferSize,nTracks,nRuns,") and rows[1].startswith("gain,48000,512,64,7,")


def test_driver_dawsim_keep_warm_round_trip():
    """gpubench --dawsim --keepWarm on the round-trip iteration: validates, keeps its slots, and exits at once (the resident
    launch is stopped at the end of the run, not waited out)."""
    import time
    t0 = time.perf_counter()
    r = run_driver("--benchmark", "Conv1D_accel", "--nTracks", "256", "--irLength", "4096", "--convMode", "roundtrip",
                   "--nRuns", "30", "--cpu-threads", "0", "--dawsim", "--keepWarm")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "DAW simulation: 33 slots, " in r.stdout, r.stdout[-1500:]
    assert "Validation passed for Conv1D_accel" in r.stdout
    assert time.perf_counter() - t0 < 60
    assert "--keepWarm" in run_driver("--help").stdout


def test_driver_json_and_unknown_name():
    r = run_driver("--benchmark", "IIRFilter", "--nRuns", "5", "--json")
    assert r.returncode == 0
    start = r.stdout.index('{\n  "benchmark"')
    d = json.loads(r.stdout[start:r.stdout.index("\n}\n", start) + 3])
    assert d["benchmark"] == "IIRFilter" and d["configuration"]["nRuns"] == 5
    r = run_driver("--benchmark", "nope")
    assert r.returncode == 1 and "Unknown benchmark 'nope'" in r.stdout
    assert run_driver("--benchmark").returncode == 1


def test_dawsim_paces_the_harness_loop(gab):
    """--dawsim: every warm-up and timed iteration is followed by a wait for the next slot
    (metal-swift Core/GPUABenchmark.swift:358-392), so a run lasts iterations x slot."""
    import time
    b = gab.Benchmark("gain", n_tracks=128)
    b.setup()
    slot = 0.005
    b.set_dawsim(buffer_seconds=slot, mode="spin")
    t0 = time.perf_counter()
    r = b.run(iterations=20, warmup=3)
    elapsed = time.perf_counter() - t0
    waits, missed = b.dawsim_stats()
    assert waits == 23 and missed <= 4                  # a descheduled host may overrun a slot or two
    assert 23 * slot - 1e-4 <= elapsed < 23 * slot + 0.5
    assert r.median_ms < slot * 1e3                     # latencies exclude the wait
    v, _ = b.validate()
    assert v.status == 0
    b.set_dawsim(enable=False)
    t0 = time.perf_counter()
    b.run(iterations=20, warmup=3)
    unpaced = time.perf_counter() - t0
    w2, _ = b.dawsim_stats()
    assert w2 == 0 and unpaced < elapsed                # unpaced again: no slots were waited for
    b.close()


@pytest.mark.parametrize("tracks", [8, 128])
def test_modal_real_bank_through_the_harness(gab, tracks):
    """modal_mode=1: the Metal port's bank (min(1024*tracks, 2^20) modes onto min(tracks, 32)
    rows) validates against its golden; the default stays the CUDA port's placeholder."""
    b = gab.Benchmark("ModalFilterBank", n_tracks=tracks, buffer_size=128, modal_mode=1)
    b.setup()
    r = b.run(iterations=3, warmup=1)
    v, text = b.validate()
    assert v.status == 0, text
    assert r.gpu_median_ms > 0
    assert b.algorithmic_bytes() == min(1024 * tracks, 1 << 20) * 32 + min(tracks, 32) * 128 * 4
    b.close()


def _json_of(stdout):
    start = stdout.index('{\n  "benchmark"')
    return json.loads(stdout[start:stdout.index("\n}\n", start) + 3])


def test_driver_conv_accel_on_devices_with_rccl_broadcast():
    """gpubench --gpus 1 goes through the whole multi-device path on one device: ncclCommInitAll,
    the impulse-response bank uploaded to device 0 and broadcast with ncclBroadcast, a host thread
    per device, the rank transforming its rows of the bank; validation (against the golden made
    from the host formula) cross-checks the broadcast bank.  N > 1 is unmeasured (one-GPU boxes)."""
    r = run_driver("--benchmark", "Conv1D_accel", "--irLength", "4096", "--nTracks", "1024", "--gpus", "1",
                   "--nRuns", "20", "--json")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    d = _json_of(r.stdout)
    m = d["multi_gpu"]
    assert m["gpus"] == 1 and m["total_tracks"] == 1024 and "ncclBroadcast" in m["collective"]
    assert m["ir_bank_bytes"] == 1024 * 4096 * 4 and m["ir_broadcast_ms"] is not None and m["ir_broadcast_ms"] >= 0
    assert m["ranks"] == [dict(m["ranks"][0], device=0, first_track=0, tracks=1024, valid=True)]
    assert m["ranks"][0]["max_error"] <= 1e-5 and m["ranks"][0]["algorithmic_bytes"] == 4 * 1024 * (2 * 512 + 2 * 4096)
    assert m["job_median_ms"] > 0 and m["tracks_per_second"] > 0
    # asking for more devices than the box has is refused with a clear message
    r = run_driver("--benchmark", "Conv1D_accel", "--gpus", "64")
    assert r.returncode == 1 and "HIP device(s) present" in r.stdout
    # replicas: a benchmark without channel structure
    r = run_driver("--benchmark", "DWG1DAccel", "--gpus", "1", "--nRuns", "5")
    assert r.returncode == 0 and "replicas only" in r.stdout


def test_driver_json_carries_roofline_and_cpu_golden():
    r = run_driver("--benchmark", "Conv1D_accel", "--irLength", "4096", "--nTracks", "256", "--nRuns", "10", "--json",
                   "--cpu-threads", "4")
    assert r.returncode == 0, r.stdout[-2000:]
    d = _json_of(r.stdout)
    assert d["roofline"]["algorithmic_bytes"] == 4 * 256 * (2 * 512 + 2 * 4096)
    assert d["roofline"]["device_median_ms"] > 0 and 0 < d["roofline"]["frac"] < 1
    assert d["cpu_golden"]["threads"] == 4 and d["cpu_golden"]["ms"] > 0
    assert d["validation"]["passed"] is True and d["validation"]["max_error"] <= 1e-5
    r = run_driver("--benchmark", "gain", "--nRuns", "5", "--json", "--cpu-threads", "0")
    assert _json_of(r.stdout)["cpu_golden"] is None


def test_driver_validate_only_and_fdtd_steps():
    r = run_driver("--benchmark", "FDTD3D", "--fdtdGrid", "32", "--fdtdSteps", "30", "--validate-only")
    assert r.returncode == 0, r.stdout[-2000:]
    assert "30 steps asked for -> 10 samples x 3 steps = 30 steps" in r.stdout
    assert "Validation passed for FDTD3D" in r.stdout and "Running FDTD3D benchmark (" not in r.stdout


def test_driver_throughput_mode_reproduces_the_headline_through_the_harness():
    """gpubench --convBatch 64: an iteration is ONE gab_conv_process_batch launch over 64 HBM-resident
    buffers (bench.py's `value` path behind the reference's driver); the first buffer of a batch from
    reset validates against the golden, and the JSON's roofline prices 64 buffers per launch."""
    r = run_driver("--benchmark", "Conv1D_accel", "--irLength", "4096", "--nTracks", "1024", "--convBatch", "64",
                   "--nRuns", "150", "--json", "--cpu-threads", "0")
    assert r.returncode == 0, r.stdout[-3000:]
    assert "Validation passed for Conv1D_accel" in r.stdout
    d = _json_of(r.stdout)
    assert d["roofline"]["algorithmic_bytes"] == 64 * 4 * 1024 * (2 * 512 + 2 * 4096)
    assert d["validation"]["passed"] is True and d["validation"]["max_error"] <= 1e-5
    assert 0.25 < d["roofline"]["device_median_ms"] < 0.6         # 64 buffers in one launch: 0.31-0.33 ms warm (twelve-wave launch)
    assert d["roofline"]["frac"] > 0.6
    r = run_driver("--benchmark", "Conv1D_accel", "--convBatch", "0")
    assert r.returncode == 1 and "Error: --convBatch must be >= 1" in r.stdout


def test_harness_conv_batch_config(gab):
    """gab_bench_config.conv_batch: the throughput mode through the C harness API."""
    b = gab.Benchmark("Conv1D_accel", n_tracks=64, ir_length=4096, conv_batch=12)
    b.setup()
    r = b.run(iterations=5, warmup=1)
    v, text = b.validate()
    assert v.status == 0 and v.max_error <= 1e-5, text
    assert b.algorithmic_bytes() == 12 * 4 * 64 * (2 * 512 + 2 * 4096) and r.gpu_median_ms > 0
    b.close()


@pytest.mark.parametrize("name,total,cfg,iters", [
    ("gain", 37, {}, 1),
    ("GainStats", 37, {}, 1),
    ("IIRFilter", 37, {}, 3),                               # state carries from iteration to iteration
    ("FFT1D", 21, {}, 1),
    ("RndMemRead", 29, {}, 4),                              # the playheads advance and wrap between iterations
    ("Conv1D", 11, {"ir_length": 1024}, 1),                 # a halo of two tracks
    ("Conv1D", 9, {"ir_length": 256}, 1),                   # one track
    ("Conv1D", 7, {"ir_length": 1500, "buffer_size": 128}, 1),   # twelve tracks of history: more than some shards have before them
    ("Conv1D_accel", 24, {"ir_length": 4096}, 3),
    ("Conv1D_accel", 52, {"ir_length": 2000}, 2),           # 13 duos over 2 / 3 ranks
])
@pytest.mark.parametrize("world", [2, 3])
def test_channel_shards_side_by_side_are_the_unsharded_results(gab, name, total, cfg, iters, world):
    """SURVEY 8e on one device: every benchmark with independent tracks, run as `world` contiguous shards of the
    job (gab_bench_set_shard: the job's input rows, banks with the global track index, playheads of the global
    tracks, Conv1D's halo rows), leaves — side by side — exactly what the unsharded benchmark leaves: every result
    array bit for bit (outputs, GainStats' statistics, the IIR state, RndMem after its playheads have advanced),
    and every shard validates against its own CPU golden.  FFT1D and Conv1D_accel pack two / four tracks into one
    transform, so their shards are cut at multiples of that (sharding.shard_granule) — with any other cut the values
    still agree to rounding, not to the bit."""
    from gpuaudiobench_amd import sharding
    whole = gab.Benchmark(name, n_tracks=total, **cfg)
    whole.setup()
    whole.run(iterations=iters, warmup=0)
    want = {k: v[0] for k, v in whole.results().items()}
    whole.close()
    parts = []
    for rank in range(world):
        b = sharding.shard_benchmark(name, rank, world, total, **cfg)
        b.setup()
        b.run(iterations=iters, warmup=0)
        parts.append(b.results())
        v, text = b.validate()
        assert v.status == 0, (rank, text)
        b.close()
    got = sharding.join_results(parts)
    assert set(got) == set(want)
    for key in want:
        assert got[key].shape == want[key].shape, key
        assert np.array_equal(got[key].view(np.uint32), want[key].view(np.uint32)), (name, key)


def test_benchmarks_that_reduce_into_shared_outputs_refuse_a_shard(gab):
    for name in ("DWG1DAccel", "ModalFilterBank", "FDTD3D", "NoOp"):
        b = gab.Benchmark(name, n_tracks=8)
        with pytest.raises(gab.GabError):
            b.set_shard(0, 16)
        b.close()
    b = gab.Benchmark("gain", n_tracks=8)
    with pytest.raises(gab.GabError):
        b.set_shard(12, 16)                                # [12, 20) does not lie inside 16 tracks
    b.close()


@pytest.mark.parametrize("name", ["Conv1D", "RndMemRead", "IIRFilter", "DWG1DAccel"])
def test_driver_gpus_says_which_partition_ran(name):
    """gpubench --gpus 1 walks the multi-device path (host thread, shard, setup under the lock) for every
    benchmark; --json names the partition: channel shards for independent tracks, replicas otherwise."""
    r = run_driver("--benchmark", name, "--gpus", "1", "--nTracks", "16", "--nRuns", "3", "--json", "--cpu-threads", "0")
    assert r.returncode == 0, r.stdout[-2000:]
    j = _json_of(r.stdout)
    part = j["multi_gpu"]["partition"]
    if name == "DWG1DAccel":
        assert part == "replicas only"
    else:
        assert part.startswith("contiguous channel shards")
        assert ("halo" in part) == (name == "Conv1D") and ("pool" in part) == (name == "RndMemRead")
    assert j["multi_gpu"]["ranks"][0]["valid"] is True and j["multi_gpu"]["ranks"][0]["tracks"] == 16


def test_conv_accel_round_trip_mode_through_the_harness_and_the_driver(gab):
    """--convMode roundtrip / conv_mode = 2: every iteration of the reference-shaped loop is ONE gab_conv_round_trip
    call (upload, kernel and download overlapped) instead of copy, kernel, copy: validates against the same golden
    and the iteration's wall time drops (C3: ~63 us against ~112 us; asserted loosely: below 0.8 of it)."""
    cfg = dict(n_tracks=1024, ir_length=4096)
    lat = {}
    for mode in (1, 2):
        b = gab.Benchmark("Conv1D_accel", conv_mode=mode, **cfg)
        b.setup()
        r = b.run(iterations=200, warmup=20)
        v, text = b.validate()
        assert v.status == 0 and v.max_error <= 1e-5, text
        lat[mode] = r.median_ms
        b.close()
    assert lat[2] < 0.8 * lat[1], lat
    r = run_driver("--benchmark", "Conv1D_accel", "--irLength", "4096", "--nTracks", "1024", "--convMode", "roundtrip",
                   "--nRuns", "100", "--json", "--cpu-threads", "0")
    assert r.returncode == 0, r.stdout[-2000:]
    j = _json_of(r.stdout)
    assert j["validation"]["passed"] is True and j["statistics"]["p50_ms"] < 0.09


def test_datacopy_overlap_and_sequential_leave_the_same_bits_and_overlap_is_faster(gab):
    """--datacopyMode: the default (ONE engine upload beside a kernel that writes the pinned output as the input lands)
    and the reference's H2D -> kernel -> D2H both validate against the golden; their input-independent tails are the
    same bits (each instance draws its own input from rand(); bit-identity on one input is test_gpu_parity's); at the
    even split the iteration's wall time drops (~126 us against ~219 us; asserted loosely: below 0.8 of it)."""
    for name, n_in in (("datacopy0199", 26214), ("datacopy5050", 1310720), ("datacopy9901", 2595225)):
        out, lat = {}, {}
        for mode in (0, 1):
            b = gab.Benchmark(name, datacopy_mode=mode)
            b.setup()
            r = b.run(iterations=100, warmup=10)
            v, text = b.validate()
            assert v.status == 0 and v.max_error <= 1e-5, text
            out[mode] = b.results()["output"][0]
            lat[mode] = r.median_ms
            b.close()
        assert out[0].size == out[1].size
        assert np.array_equal(out[0][n_in:].view(np.uint32), out[1][n_in:].view(np.uint32)), name
        if name == "datacopy5050":
            assert lat[0] < 0.8 * lat[1], lat
    r = run_driver("--benchmark", "datacopy2080", "--datacopyMode", "sequential", "--nRuns", "50", "--json", "--cpu-threads", "0")
    assert r.returncode == 0, r.stdout[-2000:]
    assert _json_of(r.stdout)["validation"]["passed"] is True
    assert run_driver("--benchmark", "datacopy2080", "--datacopyMode", "sideways").returncode != 0
