"""Pins the CPU oracle against the known-answer values of SURVEY.md §8c.

Those values (first samples, sums, FNV-1a-64 of whole arrays) were captured at
survey time from the reference's own CPU golden functions; the reference ships
no fixtures of its own.  Every assertion here is on oracle output only.
"""
import ctypes
import numpy as np
import pytest


def f9(x):
    return float("%.9g" % float(x))


def test_noise_mt19937(orc):
    x = orc.noise(65536, 42)
    assert [f9(v) for v in x[:4]] == [-0.250919759, 0.593086004, 0.90142858, -0.633130431]
    assert f9(x[65535]) == -0.69250834
    assert orc.fnv_survey(x) == "9d27214af78a9d9d"


def test_glibc_rand_restatement_matches_libc(orc):
    libc = ctypes.CDLL("libc.so.6")
    libc.rand.restype = ctypes.c_int
    for seed in (1, 42, 123456789):
        libc.srand(seed)
        r = orc.Rand(seed)
        ref = [libc.rand() for _ in range(2000)]
        got = [r.next() for _ in range(2000)]
        assert ref == got
    r = orc.Rand(1)
    assert r.next() == 1804289383
    assert f9(orc.Rand(1).unit(1)[0]) == 0.840187728
    assert f9(orc.Rand(1).bipolar(1)[0]) == 0.680375457


def test_gain_c1(orc):
    x = orc.noise(128 * 512)
    g = orc.gain(x, 2.0)
    assert f9(g[0]) == -0.501839519 and f9(g[1]) == 1.18617201
    assert orc.fnv_survey(g) == "74d41e0b3a202944"
    assert f9(g.astype(np.float64).sum()) == 64.4601512


def test_gainstats(orc):
    x = orc.noise(128 * 512)
    out, stats = orc.gainstats(x, 128, 512)
    assert orc.fnv_survey(out) == "56a0b9708f4b7926"
    assert [f9(v) for v in stats[:4]] == [-0.0038294245, 0.995481014, 0.00223832252, 0.991862535]
    assert orc.fnv_survey(stats) == "636fdedd26124c44"


def test_iir(orc):
    c = orc.iir_coeffs(0.25)
    assert abs(c[0] - 0.292875) < 5e-7 and abs(c[1] - 0.585750) < 5e-7
    assert c[0] == c[2]
    assert f9(c[3]) == 5.12078699e-08
    assert abs(c[4] - 0.171500) < 5e-7
    x = orc.noise(128 * 512)
    state = np.zeros(256, np.float32)
    y = orc.iir(x, c, state, 128, 512)
    assert f9(y[0]) == -0.0734880939 and f9(y[1]) == 0.0267238021
    assert orc.fnv_survey(y) == "fad0d0724cb98566"
    assert orc.fnv_survey(state) == "14f9b23a8a88e1ba"


def test_conv1d_c2(orc):
    ir = orc.conv1d_ir(256, 256)
    assert f9(np.abs(ir).max()) == 0.00390611379
    assert f9(ir[0]) == -3.69544091e-06
    assert orc.fnv_survey(ir) == "09852172fa5603d5"
    x = orc.noise(256 * 512)
    y = orc.conv1d(x, ir, 256, 512, 256)
    assert f9(y[0]) == 9.27259123e-07 and f9(y[1]) == -1.25552117e-06
    assert f9(y[-1]) == -0.00624387199
    assert f9(np.abs(y).max()) == 0.0194899812
    assert orc.fnv_survey(y) == "f66260025b0fa20c"


@pytest.mark.slow
def test_conv_accel_c3(orc):
    ir = orc.conv_accel_ir(4096, 1024)
    assert f9(np.abs(ir).max()) == 0.000244140596
    assert f9(ir[0]) == -1.44351588e-08
    assert orc.fnv_survey(ir) == "9cc7c2749286dfe6"
    x = orc.noise(1024 * 512)
    y = orc.conv_accel(x, ir, 4096, 512, 1024)
    assert f9(y[0]) == 3.62206665e-09 and f9(y[1]) == -7.77838949e-09
    assert f9(np.abs(y).max()) == 9.12381267e-07
    assert abs(float(np.abs(y).astype(np.float64).sum()) - 0.0457849466) < 1e-8
    assert orc.fnv_survey(y) == "6931c469f45f4d0e"


def test_conv_accel_stream_first_buffer_equals_golden(orc):
    T, B, L = 8, 64, 256
    ir = orc.conv_accel_ir(L, T)
    x = orc.noise(T * B)
    hist = np.zeros(T * L, np.float32)
    y_stream = orc.conv_accel_stream(x, ir, hist, L, B, T)
    y_gold = orc.conv_accel(x, ir, L, B, T)
    # zero history: the extra products are x*0 added to the same running sum
    assert np.array_equal(y_stream, y_gold)
    # history now holds the buffer at its tail
    assert np.array_equal(hist.reshape(T, L)[:, -B:], x.reshape(T, B))


def test_fft_golden(orc):
    x = orc.fft_input(orc.Rand(1), 128, 512)
    assert f9(x[0]) == 0.680375457
    re, im = orc.fft_golden(x, 128)
    assert f9(re[0]) == 10.1018934 and f9(re[1]) == 11.670352
    assert f9(im[0]) == 0.0 and f9(im[1]) == -5.83395052
    assert orc.fnv_survey(re) == "8384498b709f2cdc"
    assert orc.fnv_survey(im) == "2592c045d5d99256"
    # the golden itself is ~3e-3 from the true DFT
    tr, ti = orc.fft_truth(x, 128)
    d = np.abs(re - tr) + np.abs(im - ti)
    assert 1e-3 < d.max() < 5e-3


def test_datatransfer(orc):
    sizes = {0.01: 26214, 0.20: 524288, 0.50: 1310720, 0.80: 2097152, 0.99: 2595225}
    for r, n in sizes.items():
        assert orc.datatransfer_size(r) == n
    n_in, n_out = sizes[0.20], sizes[0.80]
    x = orc.Rand(1).unit(n_in)
    y = orc.datatransfer(x, n_out)
    assert f9(y[-1]) == 0.00475528836
    assert orc.fnv_survey(y) == "791d28ed182a29c3"


@pytest.mark.slow
def test_rndmem(orc):
    ph, st, en = orc.rndmem_playheads(128, 512)
    assert list(ph[:4]) == [71876168, 6739505, 40299364, 24826860]
    assert ph[127] == 84904032
    pool = orc.rndmem_pool()
    assert [f9(v) for v in pool[:4]] == [0.0334699489, 0.329964221, 0.690635681, 0.422486693]
    y = orc.rndmem(pool, ph, 512)
    assert orc.fnv_survey(y) == "b59ca490d48ee02c"


def test_dwg(orc):
    wg, x = orc.dwg_init(128, 512)
    assert wg[0]["length"] == 1166
    assert f9(wg[0]["gain"]) == 0.396967798
    assert f9(wg[0]["reflection"]) == 0.991906345
    assert f9(wg[0]["damping"]) == 0.999892235
    assert wg["length"].min() == 116
    assert int((wg["length"] <= 512).sum()) == 32
    fwd = np.zeros(128 * 2000, np.float32)
    bwd = np.zeros(128 * 2000, np.float32)
    y = orc.dwg(wg, fwd, bwd, x, 512)
    # output is identically zero, the delay lines are not
    assert not y.any()
    assert fwd.any() and bwd.any()
    for _ in range(5):
        y = orc.dwg(wg, fwd, bwd, x, 512)
        assert not y.any()


def test_modal(orc):
    p = orc.modal_params(64)          # first 64 modes draw the same stream prefix
    y = orc.modal(p, 64, 512)
    assert f9(y[0]) == 0.0484272987
    assert f9(y[-1]) == 0.116782375
    assert orc.fnv_survey(y) == "4061a3942e534783"


def test_modal_bank_golden_restatement(orc):
    """Real bank (Metal golden, ModalFilterBankBenchmark.swift:73-101): analytic single modes,
    round-robin track mapping, and agreement with its float64-accumulated twin."""
    q = np.zeros(8, np.float32)
    q[0], q[1], q[3] = 0.5, 0.125, 1.0                       # amp, freq, re (im = 0)
    y = orc.modal_bank(q, 1, 16, 1)
    want = 0.5 * np.cos(2 * np.pi * 0.125 * np.arange(1, 17))
    assert np.abs(y - want).max() < 2e-7
    # mode m lands on track m % tracks
    p = np.zeros(5 * 8, np.float32)
    for m in range(5):
        p[8 * m + 0] = m + 1.0
        p[8 * m + 3] = 1.0                                    # freq 0: state stays (1, 0)
    y = orc.modal_bank(p, 5, 4, 3).reshape(3, 4)
    assert np.array_equal(y[:, 0], np.array([1.0 + 4.0, 2.0 + 5.0, 3.0], np.float32))
    p = orc.modal_params(20000)
    a, b = orc.modal_bank(p, 20000, 64, 32), orc.modal_bank_f64acc(p, 20000, 64, 32)
    assert np.abs(a - b).max() <= 3e-6 * np.abs(b).max()


def test_fdtd_track_positions_restatement(orc):
    """orc_fdtd_tracks: with one track at the shared cells it IS orc_fdtd; a track's output depends
    only on what reaches its own receiver; tracks sharing a source cell add up in track order."""
    n, B = 20, 12
    P = orc.fdtd_params(n)
    shared_src = np.array([[P.src_x, P.src_y, P.src_z]], np.int32)
    shared_rcv = np.array([[P.rcv_x, P.rcv_y, P.rcv_z]], np.int32)
    x = orc.Rand(1).bipolar(B)
    g1, g2 = orc.fdtd_grids(P), orc.fdtd_grids(P)
    o1, o2 = np.zeros(B, np.float32), np.zeros(B, np.float32)
    orc.fdtd(P, g1, x, o1, 1, B, 0, B)
    orc.fdtd_tracks(P, g2, x, o2, 1, B, 0, B, shared_src, shared_rcv)
    assert np.array_equal(o1, o2) and np.array_equal(g1[0], g2[0]) and np.abs(g1[0]).max() > 0
    # two tracks on one source cell: the field is that of their sample-wise sum (track order)
    x2 = orc.Rand(2).bipolar(2 * B)
    src = np.repeat(shared_src, 2, axis=0)
    rcv = np.array([[P.src_x + 1, P.src_y, P.src_z], [P.src_x, P.src_y + 2, P.src_z]], np.int32)
    g3 = orc.fdtd_grids(P)
    o3 = np.zeros(2 * B, np.float32)
    orc.fdtd_tracks(P, g3, x2, o3, 2, B, 0, B, src, rcv)
    o3 = o3.reshape(2, B)
    assert np.abs(o3[0]).max() > 0 and np.abs(o3[1]).max() > 0 and not np.array_equal(o3[0], o3[1])
    # the nearer receiver hears the source first
    first = lambda v: int(np.flatnonzero(v)[0])
    assert first(o3[0]) <= first(o3[1])


def test_statistics(orc):
    lat = np.array([1.0, 2.0, 3.0, 4.0, 10.0], np.float32)
    s = orc.statistics(lat)
    assert s.count == 5 and s.mean == 4.0 and s.median == 3.0
    assert s.min_val == 1.0 and s.max_val == 10.0
    assert abs(s.std_dev - np.std(lat, ddof=1)) < 1e-6
    assert abs(s.p95 - np.percentile(lat.astype(np.float64), 95)) < 1e-5
    assert abs(s.p99 - np.percentile(lat.astype(np.float64), 99)) < 1e-5


def test_fdtd_source_cell_orders(orc):
    """orc_fdtd_trackwise — one order of the reference's atomicAdd at the source cell (cuda/bench_fdtd3d.cu:101-120):
    with ONE track it is orc_fdtd; with all tracks on the shared cells it is orc_fdtd_tracks (both accumulate into the
    cell in ascending track order); against orc_fdtd's own grouping (the tracks' sum added once) any order stays within
    rounding of the output's peak, and an order is really applied (a reversed one may change bits, never the scale)."""
    n, T, B = 20, 8, 24
    P = orc.fdtd_params(n)
    x1 = orc.Rand(1).bipolar(B)
    g1, g2 = orc.fdtd_grids(P), orc.fdtd_grids(P)
    o1, o2 = np.zeros(B, np.float32), np.zeros(B, np.float32)
    orc.fdtd(P, g1, x1, o1, 1, B, 0, B)
    orc.fdtd_trackwise(P, g2, x1, o2, 1, B, 0, B)
    assert np.array_equal(o1, o2) and np.array_equal(g1[0], g2[0])
    x = orc.Rand(3).bipolar(T * B)
    cells = lambda a, b, c: np.repeat(np.array([[a, b, c]], np.int32), T, axis=0)
    ga, gb, gc, gd = (orc.fdtd_grids(P) for _ in range(4))
    oa, ob, oc, od = (np.zeros(T * B, np.float32) for _ in range(4))
    orc.fdtd_trackwise(P, ga, x, oa, T, B, 0, B)
    orc.fdtd_tracks(P, gb, x, ob, T, B, 0, B, cells(P.src_x, P.src_y, P.src_z), cells(P.rcv_x, P.rcv_y, P.rcv_z))
    assert np.array_equal(oa, ob) and np.array_equal(ga[0], gb[0])
    orc.fdtd_trackwise(P, gc, x, oc, T, B, 0, B, order=list(range(T - 1, -1, -1)))
    orc.fdtd(P, gd, x, od, T, B, 0, B)
    peak = float(np.abs(od).max())
    assert peak > 0
    for o in (oa, oc):
        assert float(np.abs(o - od).max()) <= 1e-5 * peak
    with pytest.raises(AssertionError):
        orc.fdtd_trackwise(P, orc.fdtd_grids(P), x, np.zeros(T * B, np.float32), T, B, 0, B, order=[0] * T)
