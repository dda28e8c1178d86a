"""tools/sweep.py on a three-point grid, every convolution row checked against the
oracle's float64 direct form (impulse responses taken at the GLOBAL channel index), every harness
row against the harness's own validate()."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu
TOL = 1e-5          # of the stream's peak (north_star: 1e-5 relative for float DSP)


@pytest.fixture(scope="module")
def gab():
    import gpuaudiobench_amd
    return gpuaudiobench_amd


@pytest.fixture(scope="module")
def orc():
    import oracle
    oracle.build()
    return oracle


def test_sweep_rows_are_checked_against_the_oracle(gab, orc):
    import sweep
    B = sweep.B
    seen = []

    def oracle_checker(taps):
        def check(T, lo, n, ir_rows, x_rows, got):
            # the tool's bank rows must be the reference formula at the global index ...
            ir = orc.conv_accel_ir(taps, n, track_offset=lo, total_tracks=T)
            assert np.array_equal(np.ascontiguousarray(ir_rows).ravel().view(np.uint32), ir.view(np.uint32))
            # ... and the ten streamed buffers the float64 direct form with carried history
            hist = np.zeros(n * taps, np.float32)
            worst, peak = 0.0, 0.0
            for x, y in zip(x_rows, got):
                ref = orc.conv_accel_stream(np.ascontiguousarray(x).ravel(), ir, hist, taps, B, n, f64=True).reshape(B, n)
                worst = max(worst, float(np.abs(y - ref).max()))
                peak = max(peak, float(np.abs(ref).max()))
            seen.append((taps, T, lo, worst / peak))
            return peak > 0 and worst / peak <= TOL
        return check

    rows = []
    grid = [128, 1024, 4096]
    sweep.conv_sweep(gab, grid, rows, taps=4096, steps=50, checker=oracle_checker(4096))
    sweep.conv_sweep(gab, [256], rows, taps=8192, steps=20, checker=oracle_checker(8192))
    sweep.harness_sweep(gab, "gain", grid, rows, iterations=5)
    sweep.harness_sweep(gab, "RndMemRead", [128, 1024], rows, iterations=5)
    assert len(seen) == 4 and [s[1] for s in seen] == grid + [256]
    assert len(rows) == 9
    for r in rows:
        assert r["valid"] is True, r
        assert r["device_median_ms"] > 0 and r["algorithmic_bytes"] > 0
    # the tool's default checker (a small shard plan, bit equality) accepts the same rows
    rows2 = []
    sweep.conv_sweep(gab, [128], rows2, taps=4096, steps=10)
    assert rows2[0]["valid"] is True
