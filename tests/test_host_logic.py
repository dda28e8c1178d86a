"""Host-side logic of the harness against the oracle: generators, statistics,
legacy writers, registry.  No GPU needed."""
import json
import os

import numpy as np
import pytest

import gpuaudiobench_amd as gab
from gpuaudiobench_amd import harness as H


def test_registry_names_and_order():
    # cuda/main.cu:84-100
    assert H.benchmark_names() == [
        "NoOp", "gain", "GainStats", "datacopy0199", "datacopy2080", "datacopy5050",
        "datacopy8020", "datacopy9901", "FFT1D", "IIRFilter", "Conv1D", "Conv1D_accel",
        "ModalFilterBank", "DWG1DNaive", "DWG1DAccel", "FDTD3D", "RndMemRead"]


@pytest.mark.parametrize("n,seed", [(1, 42), (65536, 42), (1000, 7)])
def test_noise_generator_matches_oracle(orc, n, seed):
    assert np.array_equal(H.noise(n, seed), orc.noise(n, seed))


def test_noise_pin():
    x = H.noise(65536)
    assert float("%.9g" % x[0]) == -0.250919759 and float("%.9g" % x[65535]) == -0.69250834


@pytest.mark.parametrize("L,T,off,total", [(256, 256, 0, 256), (4096, 16, 1000, 8192), (512, 3, 0, 3)])
def test_ir_generators_match_oracle(orc, L, T, off, total):
    assert np.array_equal(H.conv_accel_ir(L, T, off, total), orc.conv_accel_ir(L, T, off, total))
    if off == 0:
        assert np.array_equal(H.conv1d_ir(L, T), orc.conv1d_ir(L, T))


def test_sharded_ir_equals_slice_of_global_bank():
    full = H.conv_accel_ir(512, 64).reshape(64, 512)
    for r in range(4):
        part = H.conv_accel_ir(512, 16, track_offset=16 * r, total_tracks=64).reshape(16, 512)
        assert np.array_equal(part, full[16 * r:16 * r + 16])


@pytest.mark.parametrize("lat", [[1, 2, 3, 4, 10], [5.0], [0.25, 0.5, 0.125, 7, 3, 3, 3, 9.5],
                                 list(np.linspace(0.1, 11, 100))])
def test_statistics_match_oracle(orc, lat):
    a, b = H.statistics(lat), orc.statistics(np.array(lat, np.float32))
    for f in ("mean", "median", "min_val", "max_val", "p95", "p99", "count"):
        assert getattr(a, f) == getattr(b, f), f
    if len(lat) > 1:
        assert a.std_dev == b.std_dev


def test_json_results_format():
    H.set_globals(fs=48000, buffer_size=512, n_tracks=128, n_runs=100)
    lat = [1.0, 2.0, 3.0, 12.0]
    txt = H.json_results(lat, "Conv1D_accel")
    d = json.loads(txt)
    assert d["benchmark"] == "Conv1D_accel"
    assert d["configuration"] == {"fs": 48000, "bufferSize": 512, "nTracks": 128, "nRuns": 4}
    # nearest-rank percentiles of the legacy writer: index = int(n*p)  (cuda/globals.cu:86-88)
    assert d["statistics"]["p50_ms"] == 3.0 and d["statistics"]["p99_ms"] == 12.0
    assert abs(d["deadline"]["threshold_ms"] - 1000.0 * 512 / 48000) < 1e-5
    assert d["deadline"]["meets_deadline"] is False
    assert txt.startswith('{\n  "benchmark": "Conv1D_accel",\n  "configuration": {\n    "fs": 48000,')


def test_csv_results_format(tmp_path):
    H.set_globals(fs=44100, buffer_size=256, n_tracks=64, n_runs=3)
    f = str(tmp_path / "out.csv")
    H.write_csv_results([0.5, 0.25, 1.0], "gain", f)
    H.write_csv_results([0.5, 0.25, 9.0], "gain", f)
    lines = open(f).read().splitlines()
    assert lines[0] == ("benchmark,fs,bufferSize,nTracks,nRuns,min_ms,max_ms,avg_ms,p50_ms,p95_ms,"
                        "p99_ms,threshold_ms,meets_deadline")
    assert len(lines) == 3                                # header written once
    cols = lines[1].split(",")
    assert cols[:5] == ["gain", "44100", "256", "64", "3"]
    assert cols[-1] == "true" and lines[2].split(",")[-1] == "false"
    H.set_globals()


def test_bench_config_validation():
    with pytest.raises(gab.GabError):
        gab.Benchmark("gain", n_tracks=0)
    with pytest.raises(gab.GabError):
        gab.Benchmark("not-a-benchmark")
    with pytest.raises(TypeError):
        gab.Benchmark("gain", bogus=1)


# ---- DAW-style pacing (metal-swift Core/BenchmarkUtilities.swift:140-178) ----------
# Slots of 10-20 ms and generous upper bounds: these run on shared CI hosts.
@pytest.mark.parametrize("mode", ["spin", "sleep"])
def test_dawsim_keeps_the_buffer_grid(mode):
    import time
    slot = 0.010
    sim = gab.harness.DawSim(buffer_seconds=slot, mode=mode)
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        sim.wait()                       # first call fixes the grid: returns at t0 + slot
    elapsed = time.perf_counter() - t0
    assert elapsed >= n * slot - 1e-4    # never early: that is the contract
    assert elapsed < n * slot + 0.25     # and not absurdly late
    waits, missed = sim.stats()
    assert waits == n and missed <= 1    # a descheduled host may overrun one slot
    sim.close()


def test_dawsim_counts_missed_slots_and_stays_on_grid():
    import time
    slot = 0.020
    sim = gab.harness.DawSim(buffer_seconds=slot, mode="spin")
    sim.wait()                           # returns at t0 + slot; slots 2, 3, ... follow on the grid
    t_grid = time.perf_counter()
    time.sleep(2.5 * slot)               # an "iteration" that overruns slots 2 and 3
    t0 = time.perf_counter()
    sim.wait()                           # slot 2 is already past: returns at once, counted as missed
    assert time.perf_counter() - t0 < slot
    sim.wait()                           # slot 3 too
    sim.wait()                           # slot 4 = t_grid + 3 slots: waited for
    assert time.perf_counter() - t_grid >= 3 * slot - 1e-3
    waits, missed = sim.stats()
    assert waits == 4 and missed >= 2
    sim.close()


def test_dawsim_jitter_is_bounded_and_rejects_bad_arguments():
    import time
    slot, jit_us = 0.010, 2000.0
    sim = gab.harness.DawSim(buffer_seconds=slot, mode="spin", jitter_us=jit_us)
    t0 = time.perf_counter()
    stamps = []
    for _ in range(12):
        sim.wait()
        stamps.append(time.perf_counter() - t0)
    # the k-th return is never earlier than its grid point minus the jitter
    for k, t in enumerate(stamps):
        assert t >= (k + 1) * slot - jit_us * 1e-6 - 5e-4
    assert stamps[-1] < 12 * slot + 0.25
    sim.close()
    with pytest.raises(gab.GabError):
        gab.harness.DawSim(buffer_seconds=0.0)
    with pytest.raises(gab.GabError):
        gab.harness.DawSim(buffer_seconds=0.01, jitter_us=-1.0)


def test_trace_period_tool_on_a_synthetic_trace(tmp_path):
    """tools/trace_period.py: the per-buffer period and the kernels-in-flight figure from a
    rocprofv3 kernel-trace CSV — two range kernels per buffer, overlapping on two streams."""
    import csv, json, subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    f = tmp_path / "t_kernel_trace.csv"
    with open(f, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kind", "Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        for k in range(100):                              # buffer k: two 7 us kernels, 8 us period, offset by 3 us
            for r in range(2):
                t0 = 1000000 + 8000 * k + 3000 * r
                w.writerow(["KERNEL_DISPATCH", "gab::(anonymous namespace)::conv_split_range_kernel(float const*)", t0, t0 + 7000])
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "trace_period.py"), str(f), "--kernel", "conv_split_range",
                          "--last", "200", "--per-buffer", "2"], capture_output=True, text=True, check=True).stdout
    d = json.loads(out)
    assert d["kernel"] == "conv_split_range_kernel" and d["launches_per_buffer"] == 2 and d["buffers"] == 100
    assert abs(d["avg_kernel_duration_us"] - 7.0) < 1e-9
    assert abs(d["period_us_per_buffer"] - (8000 * 99 + 3000 + 7000) / 100 / 1e3) < 1e-9
    assert 1.7 < d["avg_kernels_in_flight"] < 1.76


def test_driver_shard_arithmetic_and_flags_without_a_gpu():
    """gpubench --print-shards (no device is touched) and gab_shard_range agree with the Python side's
    sharding.shard_range; flag validation happens before any device call."""
    import ctypes as C
    import subprocess
    from gpuaudiobench_amd import sharding, _capi
    exe = os.path.join(os.path.dirname(os.path.abspath(gab.__file__)), "gpubench")
    for world, total in [(1, 128), (3, 1000), (8, 8192), (5, 7), (4, 4)]:
        r = subprocess.run([exe, "--print-shards", "--gpus", str(world), "--nTracks", str(total)],
                           capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stdout
        got = [ln for ln in r.stdout.splitlines() if ln.startswith("shard ")]
        assert len(got) == world
        covered = 0
        for rank, ln in enumerate(got):
            lo, hi = sharding.shard_range(rank, world, total)
            assert ln == "shard %d: tracks [%d, %d) = %d" % (rank, lo, hi, hi - lo)        # (the default benchmark, RndMemRead, shards)
            a, b = C.c_size_t(0), C.c_size_t(0)
            assert _capi.lib.gab_shard_range(rank, world, total, C.byref(a), C.byref(b)) == 0
            assert (a.value, b.value) == (lo, hi) and lo == covered
            covered = hi
        assert covered == total
    a, b = C.c_size_t(0), C.c_size_t(0)
    assert _capi.lib.gab_shard_range(3, 3, 10, C.byref(a), C.byref(b)) == _capi.GAB_ERR_INVALID_ARG
    assert _capi.lib.gab_shard_range(0, 0, 10, C.byref(a), C.byref(b)) == _capi.GAB_ERR_INVALID_ARG
    for bad in (["--gpus", "0"], ["--gpus"], ["--cpu-threads", "-2"], ["--print-shards", "--gpus", "9", "--nTracks", "4"]):
        r = subprocess.run([exe] + bad, capture_output=True, text=True, timeout=60)
        assert r.returncode == 1 and "Error" in r.stdout, (bad, r.stdout)
    help_text = subprocess.run([exe, "--help"], capture_output=True, text=True, timeout=60).stdout
    for flag in ("--gpus", "--fdtdSteps", "--validate-only", "--cpu-threads", "--print-shards", "--convBatch"):
        assert flag in help_text


def test_conv1d_shards_carry_their_halo_rows_and_shared_outputs_stay_replicas():
    """gpubench --print-shards for Conv1D names the input rows a shard needs in front of its own (its golden
    convolves the FLAT input: the preceding ceil((L-1)/B) tracks, fewer at the job's start) — the same arithmetic
    as sharding.conv1d_halo_tracks; benchmarks that reduce into shared outputs are replicas."""
    import subprocess
    from gpuaudiobench_amd import sharding
    exe = os.path.join(os.path.dirname(os.path.abspath(gab.__file__)), "gpubench")
    for world, total, L, B in [(3, 10, 1024, 512), (8, 256, 256, 512), (4, 9, 4000, 128), (2, 5, 1, 512)]:
        r = subprocess.run([exe, "--benchmark", "Conv1D", "--print-shards", "--gpus", str(world), "--nTracks", str(total),
                            "--irLength", str(L), "--bufferSize", str(B)], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stdout
        got = [ln for ln in r.stdout.splitlines() if ln.startswith("shard ")]
        for rank, ln in enumerate(got):
            lo, hi = sharding.shard_range(rank, world, total)
            halo = sharding.conv1d_halo_tracks(lo, L, B)
            assert halo == min(lo, -(-(L - 1) // B))
            assert ln == "shard %d: tracks [%d, %d) = %d, input rows from track %d (halo %d)" % (rank, lo, hi, hi - lo, lo - halo, halo)
    r = subprocess.run([exe, "--benchmark", "FDTD3D", "--print-shards", "--gpus", "4"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "replicas only" in r.stdout
    assert set(sharding.SHARDABLE) == {"gain", "GainStats", "IIRFilter", "FFT1D", "RndMemRead", "Conv1D", "Conv1D_accel"}
    # shards of benchmarks that pack tracks into one transform are cut at multiples of the pack
    import ctypes as C
    from gpuaudiobench_amd import _capi
    for name, g in (("FFT1D", 2), ("Conv1D_accel", 4), ("gain", 1), ("Conv1D", 1)):
        assert sharding.shard_granule(name) == g == _capi.lib.gab_shard_granule(name.encode())
    for world, total, g in [(2, 21, 2), (3, 52, 4), (8, 8192, 4), (3, 7, 4), (4, 10, 1)]:
        covered = 0
        for rank in range(world):
            lo, hi = sharding.shard_range(rank, world, total, g)
            a, b = C.c_size_t(0), C.c_size_t(0)
            assert _capi.lib.gab_shard_range_aligned(rank, world, total, g, C.byref(a), C.byref(b)) == 0
            assert (a.value, b.value) == (lo, hi) and lo == covered and (lo % g == 0 or lo == total)
            covered = hi
        assert covered == total
    r = subprocess.run([exe, "--benchmark", "FFT1D", "--print-shards", "--gpus", "2", "--nTracks", "21"], capture_output=True, text=True, timeout=60)
    assert "shard 0: tracks [0, 12) = 12" in r.stdout and "shard 1: tracks [12, 21) = 9" in r.stdout
    # results side by side: rows for track-major arrays, columns for sample-major ones
    a = {"o": (np.arange(6, dtype=np.float32), 0, 3), "s": (np.arange(4, dtype=np.float32).reshape(2, 2).ravel(), 1, 2)}
    b = {"o": (np.arange(6, 9, dtype=np.float32), 0, 3), "s": (np.array([9, 8], np.float32), 1, 2)}
    j = sharding.join_results([a, b])
    assert j["o"].tolist() == list(range(9)) and j["s"].tolist() == [0, 1, 9, 2, 3, 8]


def test_private_rand_streams_are_the_platforms_and_can_be_entered_anywhere():
    """BenchmarkUtils::GlibcRand (the harness draws FFT1D's input and RndMemRead's pool / playheads from it) gives the
    very stream the platform's srand(seed) + rand() gives, and `skip` enters it where a channel shard's first track starts."""
    import ctypes as C
    from gpuaudiobench_amd import _capi
    libc = C.CDLL(None)
    for seed in (1, 42, 12345):
        libc.srand(seed)
        ref = np.array([libc.rand() for _ in range(3000)], np.int32)
        got = np.empty(3000, np.int32)
        assert _capi.lib.gab_glibc_rand(seed, 0, got.ctypes.data_as(C.c_void_p), got.size) == 0
        assert np.array_equal(got, ref)
        part = np.empty(500, np.int32)
        assert _capi.lib.gab_glibc_rand(seed, 1234, part.ctypes.data_as(C.c_void_p), part.size) == 0
        assert np.array_equal(part, ref[1234:1734])


def test_round_trip_mismatch_report_classifies_each_kind_of_wrong_word():
    """tests/rt_diag.py (the report a failing round-trip comparison prints) on synthetic failures: a row that had not
    arrived (NaN pieces in an otherwise right pair), a pair poisoned by a sentinel taken for a sample (all NaN), a stale
    output word, a consumed block that differs from h_in (with the sentinel and with the previous call's word) — and
    nothing to say when the bits agree."""
    import numpy as np
    import rt_diag
    T, B = 16, 512
    rng = np.random.default_rng(5)
    want = rng.standard_normal((B, T)).astype(np.float32)
    h_in = rng.standard_normal((T, B)).astype(np.float32)
    prev_in = rng.standard_normal((T, B)).astype(np.float32)
    prev_out = rng.standard_normal((B, T)).astype(np.float32)
    assert rt_diag.classify(want, want.copy(), T, B, prev_out=prev_out, h_in=h_in, consumed=h_in.copy(), prev_in=prev_in) == ""
    got = want.copy()
    got[7, 4:8] = np.nan                                   # a 16-byte piece of one row never arrived
    got[:, 10:12] = np.nan                                 # channel pair 5 poisoned
    got[100, 0] = prev_out[100, 0]                         # a stale word
    consumed = h_in.copy()
    consumed.view(np.uint32)[3, 17] = rt_diag.SENTINEL     # the kernel took the sentinel for a sample
    consumed[9, 200] = prev_in[9, 200]                     # ... and a word of the previous call's input
    text = rt_diag.classify(want, got, T, B, prev_out=prev_out, h_in=h_in, consumed=consumed, prev_in=prev_in, label="call 3")
    assert text.startswith("call 3:") and "poisoned-pair: [5]" in text and "missing-row: 4" in text
    assert "stale-output (finite, equal to the previous call's output there): 1 of 1" in text
    assert "differs from h_in in 2 words" in text and "the sentinel: 1" in text and "previous call's input word: 1" in text
    # outputs right but the consumed block wrong is still a finding
    assert "consumed block differs" in rt_diag.classify(want, want.copy(), T, B, h_in=h_in, consumed=consumed)
