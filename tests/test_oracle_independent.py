"""The CPU oracle against TEXTBOOK implementations of the same mathematics (numpy / scipy, float64).

tests/test_oracle_pins.py pins the oracle to numbers captured from the reference's own golden functions; this file asks
a different question — is the restated algorithm the signal-processing operation it claims to be? — with code that
shares nothing with oracle/gab_oracle.c: scipy.signal.lfilter, scipy.signal.fftconvolve, numpy.convolve, numpy.fft.  Tolerances are those of float32 accumulation against float64 (stated per test); the oracle's bit-level
behaviour is the business of the pins and of the GPU parity tests.  CPU only, seconds.
"""
import numpy as np
import pytest

scipy_signal = pytest.importorskip("scipy.signal")


def test_iir_is_the_direct_form_biquad_lfilter_computes(orc):
    """bench_iir.cu:170-197 as restated: w = x - a1 z1 - a2 z2, y = b0 w + b1 z1 + b2 z2 is H(z) = B(z)/A(z) with
    a0 = 1; the state carried between buffers is lfilter's continuation (checked by filtering two buffers in turn
    against ONE lfilter call over both)."""
    T, B = 16, 512
    c = orc.iir_coeffs(0.25)
    b, a = c[:3].astype(np.float64), np.array([1.0, c[3], c[4]], np.float64)
    x = np.concatenate([orc.noise(T * B, seed=5).reshape(T, B), orc.noise(T * B, seed=6).reshape(T, B)], axis=1)   # [T][2B]
    want = scipy_signal.lfilter(b, a, x.astype(np.float64), axis=1)
    state = np.zeros(2 * T, np.float32)
    y0 = orc.iir(np.ascontiguousarray(x[:, :B]).reshape(-1), c, state, T, B).reshape(T, B)
    y1 = orc.iir(np.ascontiguousarray(x[:, B:]).reshape(-1), c, state, T, B).reshape(T, B)
    got = np.concatenate([y0, y1], axis=1)
    assert np.abs(got - want).max() < 2e-6 * max(1.0, np.abs(want).max())      # float32 recursion, poles at radius 0.41


def test_conv1d_is_a_linear_convolution_over_the_flat_stream(orc):
    """bench_conv1d.cu:188-208 as restated: track t's taps run over the FLAT sample stream (the range test is on the flat
    index, so a track's first outputs reach into the previous track's samples — SURVEY 2.3); that is numpy.convolve of
    the stream up to the track's end with the track's taps."""
    T, B, L = 6, 128, 96
    ir = orc.conv1d_ir(L, T).reshape(T, L).astype(np.float64)
    x = orc.noise(T * B, seed=11)
    got = orc.conv1d(x, ir.astype(np.float32).reshape(-1), L, B, T).reshape(T, B)
    flat = x.astype(np.float64)
    for t in range(T):
        want = np.convolve(flat[:(t + 1) * B], ir[t])[t * B:(t + 1) * B]
        assert np.abs(got[t] - want).max() < 1e-6 * max(1e-3, np.abs(want).max()) + 1e-9, t


def test_conv_accel_is_the_first_block_of_a_linear_convolution_sample_major(orc):
    """bench_conv1d_accel.cu:208-246 as restated: per track the first B samples of x * h (zero history), written
    sample-major (out[T s + t])."""
    T, B, L = 8, 64, 256
    ir = orc.conv_accel_ir(L, T)
    x = orc.noise(T * B, seed=3)
    got = orc.conv_accel(x, ir, L, B, T).reshape(B, T)
    for t in range(T):
        want = np.convolve(x[t * B:(t + 1) * B].astype(np.float64), ir[t * L:(t + 1) * L].astype(np.float64))[:B]
        assert np.abs(got[:, t] - want).max() < 1e-6 * np.abs(want).max() + 1e-12, t


def test_conv_accel_stream_is_overlap_save_of_the_whole_signal(orc):
    """The streaming extension (history carried between buffers): N buffers in turn give the first N B samples of the
    linear convolution of the whole input with the impulse response — what scipy.signal.fftconvolve computes by FFT."""
    T, B, L, N = 4, 64, 256, 7
    ir = orc.conv_accel_ir(L, T)
    xs = [orc.noise(T * B, seed=40 + n) for n in range(N)]
    hist = np.zeros(T * L, np.float32)
    got = np.concatenate([orc.conv_accel_stream(x, ir, hist, L, B, T).reshape(B, T) for x in xs], axis=0)    # [N B][T]
    for t in range(T):
        whole = np.concatenate([x[t * B:(t + 1) * B] for x in xs]).astype(np.float64)
        want = scipy_signal.fftconvolve(whole, ir[t * L:(t + 1) * L].astype(np.float64))[:N * B]
        assert np.abs(got[:, t] - want).max() < 2e-6 * np.abs(want).max(), t
    # and the float64 form of the same oracle (the yardstick of the GPU tests' 1e-5 bound) agrees with it more closely still
    hist64 = np.zeros(T * L, np.float32)
    got64 = np.concatenate([orc.conv_accel_stream(x, ir, hist64, L, B, T, f64=True).reshape(B, T) for x in xs], axis=0)
    for t in range(T):
        whole = np.concatenate([x[t * B:(t + 1) * B] for x in xs]).astype(np.float64)
        want = scipy_signal.fftconvolve(whole, ir[t * L:(t + 1) * L].astype(np.float64))[:N * B]
        assert np.abs(got64[:, t] - want).max() < 1e-12 + 1e-9 * np.abs(want).max(), t


def test_fft_truth_is_numpys_real_fft(orc):
    """orc_fft_truth (the float64 DFT the FFT kernels are held to within 1e-5 of peak) against numpy.fft.rfft; the
    reference's own golden (its twiddle recurrence in float) stays ~3e-3 away, as SURVEY 2.3 records."""
    T = 6
    x = orc.fft_input(orc.Rand(7), T, 512)
    re, im = orc.fft_truth(x, T)
    want = np.fft.rfft(x.reshape(T, 1024).astype(np.float64), axis=1)
    assert np.abs(re.reshape(T, 513) - want.real).max() < 1e-9
    assert np.abs(im.reshape(T, 513) - want.imag).max() < 1e-9
    gre, gim = orc.fft_golden(x, T)
    d = np.abs(gre.reshape(T, 513) - want.real) + np.abs(gim.reshape(T, 513) - want.imag)
    assert 1e-4 < d.max() < 1e-2


def test_gain_and_gainstats_are_what_numpy_says(orc):
    """bench_gain.cu / bench_gainstats.cu:120-144 as restated: out = g x (g = 2 and 0.5); per track the mean and the
    maximum of the INPUT samples (float32 running sum against numpy's float64 mean)."""
    T, B = 12, 256
    x = orc.noise(T * B, seed=21)
    assert np.array_equal(orc.gain(x, 2.0), (x * np.float32(2.0)).astype(np.float32))
    out, stats = orc.gainstats(x, T, B)
    assert np.array_equal(out, (x * np.float32(0.5)).astype(np.float32))
    rows = x.reshape(T, B).astype(np.float64)
    st = stats.reshape(T, 2)
    assert np.abs(st[:, 0] - rows.mean(axis=1)).max() < 1e-6
    assert np.array_equal(st[:, 1], x.reshape(T, B).max(axis=1))


def test_datatransfer_is_a_copy_with_a_synthetic_tail(orc):
    """bench_datatransfer.cu:139-147 as restated: the input copied, and past its end 0.5 + 0.5 sin(0.001 i) in float."""
    n_in, n_out = orc.datatransfer_size(0.20), orc.datatransfer_size(0.80)
    x = orc.Rand(3).unit(n_in)
    y = orc.datatransfer(x, n_out)
    assert np.array_equal(y[:n_in], x)
    i = np.arange(n_in, n_out, dtype=np.float32)
    tail = 0.5 + 0.5 * np.sin((i * np.float32(0.001)).astype(np.float64))
    assert np.abs(y[n_in:] - tail).max() < 2e-7                   # sinf against float64 sin of the same float argument
    x2 = orc.Rand(4).unit(n_out)
    assert np.array_equal(orc.datatransfer(x2, n_in), x2[:n_in])


def test_modal_bank_is_a_sum_of_rotating_phasors(orc):
    """ModalFilterBankBenchmark.swift:73-101 as restated: every mode is a unit-modulus rotation of its complex state by
    2 pi f per sample, its real part times the amplitude added into track (mode % tracks) — in closed form
    amp Re((re0 + i im0) e^{i (n+1) theta}); the float32 recurrence drifts from it by rounding only."""
    n_modes, B, tracks = 96, 256, 8
    p = orc.modal_params(n_modes).reshape(n_modes, 8)
    got = orc.modal_bank(p.reshape(-1), n_modes, B, tracks).reshape(tracks, B).astype(np.float64)
    theta = (np.float32(2.0) * np.float32(np.pi) * p[:, 1]).astype(np.float64)      # the angle as the float it is formed in
    z0 = p[:, 3].astype(np.float64) + 1j * p[:, 4].astype(np.float64)
    n = np.arange(1, B + 1, dtype=np.float64)
    per_mode = p[:, 0:1].astype(np.float64) * (z0[:, None] * np.exp(1j * theta[:, None] * n[None, :])).real
    want = np.zeros((tracks, B))
    for m in range(n_modes):
        want[m % tracks] += per_mode[m]
    scale = np.abs(p[:, 0]).astype(np.float64).sum() / tracks
    assert np.abs(got - want).max() < 2e-4 * scale


def test_fdtd_is_the_staggered_grid_leapfrog_of_the_acoustic_wave_equation(orc):
    """bench_fdtd3d.cu:14-139, 384-438 as restated: velocities on the interior faces from the pressure gradient, pressure in
    the interior cells from the velocity divergence, the six boundary layers of cells multiplied by (1 - absorption), the
    tracks' scaled samples added at the source cell, the receiver cell read out.  The same scheme written with whole-array
    numpy slices in float64; the float32 recursion stays within rounding of it over 24 samples x 3 steps."""
    nx, ny, nz, tracks, B, n = 14, 12, 10, 3, 32, 24
    P = orc.fdtd_params(nx, ny, nz)
    x = orc.noise(tracks * B, seed=9)
    got = orc.fdtd(P, orc.fdtd_grids(P), x, np.zeros(tracks * B, np.float32), tracks, B, 0, n, fused=True).reshape(tracks, B)
    c1, c2, damp = float(P.dt_over_rho_dx), float(P.rho_c2_dt_over_dx), 1.0 - float(np.float32(P.absorption))
    p = np.zeros((nz, ny, nx))
    vx, vy, vz = np.zeros((nz, ny, nx + 1)), np.zeros((nz, ny + 1, nx)), np.zeros((nz + 1, ny, nx))
    edge = np.ones((nz, ny, nx), bool)
    edge[1:-1, 1:-1, 1:-1] = False
    xin = x.reshape(tracks, B).astype(np.float64)
    want = np.zeros(n)
    for s in range(n):
        p[P.src_z, P.src_y, P.src_x] += (xin[:, s] * float(np.float32(0.1))).sum()
        for _ in range(P.steps_per_sample):
            vx[:, :, 1:nx] -= c1 * (p[:, :, 1:] - p[:, :, :-1])
            vy[:, 1:ny, :] -= c1 * (p[:, 1:, :] - p[:, :-1, :])
            vz[1:nz, :, :] -= c1 * (p[1:, :, :] - p[:-1, :, :])
            div = (vx[:, :, 1:] - vx[:, :, :-1]) + (vy[:, 1:, :] - vy[:, :-1, :]) + (vz[1:, :, :] - vz[:-1, :, :])
            p = np.where(edge, p * damp, p - c2 * div)
        want[s] = p[P.rcv_z, P.rcv_y, P.rcv_x] * float(np.float32(0.1))
    assert np.abs(want).max() > 0                                   # (the wave has reached the receiver: the comparison is not 0 = 0)
    for t in range(tracks):                                         # every track carries the receiver's sample
        assert np.abs(got[t, :n] - want).max() < 2e-5 * np.abs(want).max()


def test_dwg_serial_walk_equals_independent_cell_recurrences_bit_for_bit(orc):
    """bench_dwg.cu:10-141, 356-399 as restated walks a waveguide's samples in order.  With writePos fixed, sample s touches
    the cell pair (fwd[p], bwd[(p + L/2) % L]), p = (writePos + s) % L, and no other sample position touches that pair: the
    512-step loop is min(L, B) independent two-cell recurrences — the cut the HIP kernels are built on.  Written here as
    such in float32 numpy (cells side by side, visits k = 0, 1, ... of every cell in turn, the output mix in waveguide
    order): bit-identical to the oracle's serial walk over two consecutive buffers, delay lines included."""
    n_wg, B, max_len = 48, 512, 2000
    wg, x = orc.dwg_init(n_wg, B)
    wg["writePos"][::3] = 37                                        # (the harness never advances it; any fixed value must do)
    assert (wg["length"] < B).any() and (wg["length"] > B).any()    # both kinds of line: visited more than once, and not at all in part
    rng = np.random.default_rng(5)
    fwd0 = rng.uniform(-1, 1, n_wg * max_len).astype(np.float32)
    bwd0 = rng.uniform(-1, 1, n_wg * max_len).astype(np.float32)
    f_o, b_o = fwd0.copy(), bwd0.copy()
    f_n, b_n = fwd0.copy(), bwd0.copy()
    half = np.float32(0.5)
    for rep in range(2):
        want = orc.dwg(wg, f_o, b_o, x, B, max_len)
        got = np.zeros(B, np.float32)
        for g in range(n_wg):
            L, tin, tout, wp = (int(wg[k][g]) for k in ("length", "inputTapPos", "outputTapPos", "writePos"))
            gain, refl, damp = wg["gain"][g], wg["reflection"][g], wg["damping"][g]
            j = np.arange(min(L, B))
            cur = (wp + j) % L
            fp, bp = g * max_len + cur, g * max_len + (cur + L // 2) % L
            f, b = f_n[fp].copy(), b_n[bp].copy()
            k = 0
            while k * L < B:
                act = j + k * L < B
                s = (j + k * L)[act]
                xs = x[s] * gain
                fa, ba = f[act] * damp, b[act] * damp
                inj = cur[act] == tin
                fa, ba = np.where(inj, fa + xs, fa), np.where(inj, ba + xs, ba)
                f[act], b[act] = ba * refl, fa * refl
                tap = cur[act] == tout
                got[s[tap]] = got[s[tap]] + (fa[tap] + ba[tap]) * half
                k += 1
            f_n[fp], b_n[bp] = f, b
        assert np.array_equal(got.view(np.int32), want.view(np.int32)), rep
        assert np.array_equal(f_n.view(np.int32), f_o.view(np.int32)) and np.array_equal(b_n.view(np.int32), b_o.view(np.int32)), rep
    assert np.abs(want).max() > 0


def test_noise_is_mt19937_through_the_standard_librarys_uniform_distribution(orc):
    """The benchmarks' input (bench_utils.cuh: std::mt19937(seed) into std::uniform_real_distribution<float>(-1, 1)) from
    numpy's own MT19937: the raw 32-bit draws, canonical value float(draw) / 2^32 (one below 1 where that rounds to 1),
    then 2 u - 1 in float — bit for bit."""
    for seed in (42, 7, 123456):
        raw = np.random.RandomState(seed).randint(0, 2 ** 32, size=20000, dtype=np.uint64).astype(np.uint32)
        u = raw.astype(np.float32) / np.float32(4294967296.0)
        u = np.where(u >= 1.0, np.nextafter(np.float32(1), np.float32(0)), u)
        want = (u * np.float32(2.0) + np.float32(-1.0)).astype(np.float32)
        assert np.array_equal(orc.noise(20000, seed).view(np.int32), want.view(np.int32)), seed


def test_rndmem_is_a_gather_of_contiguous_runs_written_sample_major(orc):
    """bench_rndmem.cu:194-205 as restated: track t plays bufsize consecutive pool samples from its playhead, the output is
    sample-major (out[tracks i + t]); a numpy fancy-index gather says the same."""
    tracks, B, n = 24, 64, 1 << 16
    pool = orc.rndmem_pool(n)
    ph, st, en = orc.rndmem_playheads(tracks, B, pool_elems=n, min_loop=100, max_loop=4000)
    assert ph.min() >= 0 and ph.max() + B <= n
    got = orc.rndmem(pool, ph, B).reshape(B, tracks)
    want = pool[ph[None, :].astype(np.int64) + np.arange(B)[:, None]]
    assert np.array_equal(got, want)
