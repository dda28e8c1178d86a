"""ctypes binding of the CPU oracle (oracle/libgab_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by bench.py's cpu_baseline leg and
by __graft_entry__.smoke() as the checker.  Nothing under gpuaudiobench_amd/
may import this package.  See oracle/gab_oracle.h for provenance and pinning.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgab_oracle.so")


def build(force=False):
    override = os.environ.get("GAB_ORACLE_LIB")       # e.g. the sanitizer build (see the Makefile)
    if override:
        return os.path.abspath(override)
    src = [os.path.join(_HERE, f) for f in ("gab_oracle.c", "gab_oracle.h", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class RandState(C.Structure):
    _fields_ = [("r", C.c_uint32 * 34), ("f", C.c_int), ("b", C.c_int)]


class IIRCoeffs(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("b0", "b1", "b2", "a1", "a2")]


class FDTDParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "nx", "ny", "nz", "src_x", "src_y", "src_z", "rcv_x", "rcv_y", "rcv_z",
        "steps_per_sample")] + [(n, C.c_float) for n in (
            "dt_over_rho_dx", "rho_c2_dt_over_dx", "absorption")]


class Stats(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "mean", "median", "std_dev", "min_val", "max_val", "p95", "p99")] + [
            ("count", C.c_size_t)]


WG_DTYPE = np.dtype([("length", "<i4"), ("inputTapPos", "<i4"), ("outputTapPos", "<i4"),
                     ("writePos", "<i4"), ("gain", "<f4"), ("reflection", "<f4"),
                     ("damping", "<f4"), ("padding", "<f4")])

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_fnv1a64.restype = C.c_uint64
        _lib.orc_fnv1a64_survey.restype = C.c_uint64
        _lib.orc_rand.restype = C.c_int
        _lib.orc_datatransfer_size.restype = C.c_int
        _lib.orc_datatransfer_size.argtypes = [C.c_float]
        _lib.orc_iir_butterworth.restype = IIRCoeffs
        _lib.orc_iir_butterworth.argtypes = [C.c_float]
        _lib.orc_fdtd_default_params.restype = FDTDParams
        _lib.orc_statistics.restype = Stats
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def fnv(a):
    a = np.ascontiguousarray(a)
    return "%016x" % lib().orc_fnv1a64(_p(a), C.c_size_t(a.nbytes))


def fnv_survey(a):
    """FNV-1a-64 with the offset basis the SURVEY §8c pins were recorded with."""
    a = np.ascontiguousarray(a)
    return "%016x" % lib().orc_fnv1a64_survey(_p(a), C.c_size_t(a.nbytes))


# ---- RNG -------------------------------------------------------------------
def noise(n, seed=42):
    out = np.empty(n, np.float32)
    lib().orc_noise_mt19937(_p(out), C.c_size_t(n), C.c_uint32(seed))
    return out


class Rand:
    """glibc rand() stream; Rand(1) is a process that never called srand()."""

    def __init__(self, seed=1):
        self.st = RandState()
        lib().orc_srand(C.byref(self.st), C.c_uint(seed))

    def next(self):
        return lib().orc_rand(C.byref(self.st))

    def unit(self, n):
        out = np.empty(n, np.float32)
        lib().orc_rand_unit(C.byref(self.st), _p(out), C.c_size_t(n))
        return out

    def bipolar(self, n):
        out = np.empty(n, np.float32)
        lib().orc_rand_bipolar(C.byref(self.st), _p(out), C.c_size_t(n))
        return out


# ---- simple kernels ----------------------------------------------------------
def gain(x, g=2.0):
    x = _f32(x)
    out = np.empty_like(x)
    lib().orc_gain(_p(x), _p(out), C.c_size_t(x.size), C.c_float(g))
    return out


def gainstats(x, tracks, bufsize):
    x = _f32(x)
    out = np.empty_like(x)
    stats = np.empty(2 * tracks, np.float32)
    lib().orc_gainstats(_p(x), _p(out), _p(stats), C.c_size_t(tracks), C.c_size_t(bufsize))
    return out, stats


def noop(x):
    x = _f32(x)
    out = np.empty_like(x)
    lib().orc_noop(_p(x), _p(out), C.c_size_t(x.size))
    return out


def datatransfer_size(ratio):
    return lib().orc_datatransfer_size(C.c_float(ratio))


def datatransfer(x, out_size):
    x = _f32(x)
    out = np.empty(out_size, np.float32)
    lib().orc_datatransfer(_p(x), _p(out), C.c_int(x.size), C.c_int(out_size))
    return out


# ---- FFT ---------------------------------------------------------------------
def fft_input(rand, tracks, bufsize):
    x = np.empty(tracks * 1024, np.float32)
    lib().orc_fft_input(C.byref(rand.st), _p(x), C.c_size_t(tracks), C.c_size_t(bufsize))
    return x


def fft_golden(x, tracks):
    x = _f32(x)
    re = np.empty(tracks * 513, np.float32)
    im = np.empty(tracks * 513, np.float32)
    lib().orc_fft_golden(_p(x), _p(re), _p(im), C.c_size_t(tracks))
    return re, im


def fft_truth(x, tracks):
    x = _f32(x)
    re = np.empty(tracks * 513, np.float64)
    im = np.empty(tracks * 513, np.float64)
    lib().orc_fft_truth(_p(x), _p(re), _p(im), C.c_size_t(tracks))
    return re, im


# ---- IIR ---------------------------------------------------------------------
def iir_coeffs(nf=0.25):
    c = lib().orc_iir_butterworth(C.c_float(nf))
    return np.array([c.b0, c.b1, c.b2, c.a1, c.a2], np.float32)


def iir(x, coeffs, state, tracks, bufsize):
    """state (tracks*2) is updated in place, like the reference's d_state."""
    x = _f32(x)
    out = np.empty_like(x)
    c = IIRCoeffs(*[float(v) for v in coeffs])
    assert state.dtype == np.float32 and state.size == 2 * tracks
    lib().orc_iir(_p(x), _p(out), C.byref(c), _p(state), C.c_int(tracks), C.c_int(bufsize))
    return out


# ---- conv1d ------------------------------------------------------------------
def conv1d_ir(ir_len, tracks):
    ir = np.empty(tracks * ir_len, np.float32)
    lib().orc_conv1d_ir(_p(ir), C.c_int(ir_len), C.c_size_t(tracks))
    return ir


def conv1d(x, ir, ir_len, bufsize, tracks):
    x, ir = _f32(x), _f32(ir)
    out = np.empty(tracks * bufsize, np.float32)
    lib().orc_conv1d(_p(x), _p(ir), _p(out), C.c_int(ir_len), C.c_int(bufsize), C.c_int(tracks))
    return out


def conv_accel_ir(ir_len, tracks, track_offset=0, total_tracks=None):
    total = tracks if total_tracks is None else total_tracks
    ir = np.empty(tracks * ir_len, np.float32)
    lib().orc_conv_accel_ir(_p(ir), C.c_int(ir_len), C.c_size_t(track_offset),
                            C.c_size_t(tracks), C.c_size_t(total))
    return ir


def conv_accel(x, ir, ir_len, bufsize, tracks):
    x, ir = _f32(x), _f32(ir)
    out = np.empty(tracks * bufsize, np.float32)
    lib().orc_conv_accel(_p(x), _p(ir), _p(out), C.c_int(ir_len), C.c_int(bufsize), C.c_int(tracks))
    return out


def conv_accel_stream(x, ir, hist, ir_len, bufsize, tracks, f64=False):
    """hist (tracks*ir_len float32) is updated in place."""
    x, ir = _f32(x), _f32(ir)
    assert hist.dtype == np.float32 and hist.size == tracks * ir_len
    out = np.empty(tracks * bufsize, np.float64 if f64 else np.float32)
    fn = lib().orc_conv_accel_stream_f64 if f64 else lib().orc_conv_accel_stream
    fn(_p(x), _p(ir), _p(out), _p(hist), C.c_int(ir_len), C.c_int(bufsize), C.c_int(tracks))
    return out


# ---- modal -------------------------------------------------------------------
def modal_params(n_modes):
    p = np.empty(n_modes * 8, np.float32)
    lib().orc_modal_params(_p(p), C.c_int(n_modes))
    return p


def modal_bank(params, n_modes, bufsize, out_tracks=32):
    """The real bank: Metal golden (ModalFilterBankBenchmark.swift:73-101), fp32, mode order."""
    params = np.ascontiguousarray(params, np.float32)
    out = np.zeros(out_tracks * bufsize, np.float32)
    lib().orc_modal_bank(_p(params), _p(out), C.c_int(n_modes), C.c_int(bufsize), C.c_int(out_tracks))
    return out


def modal_bank_f64acc(params, n_modes, bufsize, out_tracks=32):
    params = np.ascontiguousarray(params, np.float32)
    out = np.zeros(out_tracks * bufsize, np.float64)
    lib().orc_modal_bank_f64acc(_p(params), _p(out), C.c_int(n_modes), C.c_int(bufsize), C.c_int(out_tracks))
    return out


def modal(params, n_modes, bufsize, out_tracks=32):
    params = _f32(params)
    out = np.empty(bufsize * out_tracks, np.float32)
    lib().orc_modal(_p(params), _p(out), C.c_int(n_modes), C.c_int(bufsize), C.c_int(out_tracks))
    return out


# ---- DWG ---------------------------------------------------------------------
def dwg_init(n_wg, bufsize):
    wg = np.zeros(n_wg, WG_DTYPE)
    x = np.empty(bufsize, np.float32)
    lib().orc_dwg_init(_p(wg), _p(x), C.c_int(n_wg), C.c_int(bufsize))
    return wg, x


def dwg(wg, fwd, bwd, x, bufsize, max_len=2000, out_tracks=None):
    """fwd/bwd (n_wg*max_len float32) are updated in place."""
    n_wg = wg.size
    out = np.empty(bufsize, np.float32)
    ot = n_wg if out_tracks is None else out_tracks
    lib().orc_dwg(_p(wg), _p(fwd), _p(bwd), _p(_f32(x)), _p(out), C.c_int(n_wg),
                  C.c_int(bufsize), C.c_int(max_len), C.c_int(ot))
    return out


# ---- FDTD3D ------------------------------------------------------------------
def fdtd_params(nx, ny=None, nz=None):
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    return lib().orc_fdtd_default_params(C.c_int(nx), C.c_int(ny), C.c_int(nz))


def fdtd_grids(P):
    return (np.zeros(P.nx * P.ny * P.nz, np.float32),
            np.zeros((P.nx + 1) * P.ny * P.nz, np.float32),
            np.zeros(P.nx * (P.ny + 1) * P.nz, np.float32),
            np.zeros(P.nx * P.ny * (P.nz + 1), np.float32))


def fdtd_placeholder(x, tracks, bufsize):
    x = _f32(x)
    out = np.empty_like(x)
    lib().orc_fdtd_placeholder(_p(x), _p(out), C.c_int(tracks), C.c_int(bufsize))
    return out


def fdtd(P, grids, x, out, tracks, bufsize, first_sample, n_samples, fused=True):
    p, vx, vy, vz = grids
    lib().orc_fdtd(C.byref(P), _p(p), _p(vx), _p(vy), _p(vz), _p(_f32(x)), _p(out),
                   C.c_int(tracks), C.c_int(bufsize), C.c_int(first_sample),
                   C.c_int(n_samples), C.c_int(1 if fused else 0))
    return out


def fdtd_trackwise(P, grids, x, out, tracks, bufsize, first_sample, n_samples, fused=True, order=None):
    """orc_fdtd with the source cell as one order of the reference's atomicAdd leaves it: the tracks' samples
    accumulate into the cell one by one (order: a permutation of range(tracks); None = ascending)."""
    p, vx, vy, vz = grids
    o = None
    if order is not None:
        o = np.ascontiguousarray(order, np.int32)
        assert sorted(o.tolist()) == list(range(tracks))
    lib().orc_fdtd_trackwise(C.byref(P), _p(p), _p(vx), _p(vy), _p(vz), _p(_f32(x)), _p(out),
                             C.c_int(tracks), C.c_int(bufsize), C.c_int(first_sample),
                             C.c_int(n_samples), C.c_int(1 if fused else 0), _p(o) if o is not None else None)
    return out


def fdtd_tracks(P, grids, x, out, tracks, bufsize, first_sample, n_samples, src_xyz, rcv_xyz, fused=True):
    """Track-dependent source / receiver cells (tracks x 3 int32 each, (x, y, z))."""
    p, vx, vy, vz = grids
    src = np.ascontiguousarray(src_xyz, np.int32)
    rcv = np.ascontiguousarray(rcv_xyz, np.int32)
    assert src.shape == (tracks, 3) and rcv.shape == (tracks, 3)
    lib().orc_fdtd_tracks(C.byref(P), _p(p), _p(vx), _p(vy), _p(vz), _p(_f32(x)), _p(out),
                          C.c_int(tracks), C.c_int(bufsize), C.c_int(first_sample),
                          C.c_int(n_samples), C.c_int(1 if fused else 0), _p(src), _p(rcv))
    return out


# ---- rndmem ------------------------------------------------------------------
RNDMEM_POOL_ELEMS = 512 * 1024 * 1024 // 4


def rndmem_pool(n=RNDMEM_POOL_ELEMS):
    pool = np.empty(n, np.float32)
    lib().orc_rndmem_pool(_p(pool), C.c_size_t(n))
    return pool


def rndmem_playheads(tracks, bufsize, pool_elems=RNDMEM_POOL_ELEMS, min_loop=1000, max_loop=48000):
    ph = np.empty(tracks, np.int32)
    st = np.empty(tracks, np.float32)
    en = np.empty(tracks, np.float32)
    lib().orc_rndmem_playheads(_p(ph), _p(st), _p(en), C.c_int(tracks), C.c_int(bufsize),
                               C.c_size_t(pool_elems), C.c_int(min_loop), C.c_int(max_loop))
    return ph, st, en


def rndmem_advance(ph, st, en, bufsize):
    lib().orc_rndmem_advance(_p(ph), _p(st), _p(en), C.c_int(ph.size), C.c_int(bufsize))


def rndmem(pool, ph, bufsize):
    tracks = ph.size
    out = np.empty(tracks * bufsize, np.float32)
    lib().orc_rndmem(_p(pool), _p(ph), _p(out), C.c_int(bufsize), C.c_int(tracks))
    return out


# ---- statistics ----------------------------------------------------------------
def statistics(lat):
    lat = _f32(lat)
    return lib().orc_statistics(_p(lat), C.c_size_t(lat.size))
