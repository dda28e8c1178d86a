"""Kernel- and plan-level calls of the C ABI on torch device tensors.

Every function forwards raw device pointers and the current torch stream to
libgab_hip.so; outputs are allocated with torch (device memory plumbing only).
Argument names and meaning follow the reference kernels (see gab_c_api.h).
"""
import ctypes as C

import torch

from ._capi import (lib, check, WaveguideState, FdtdParams, CONV_STATELESS, CONV_STREAMING,
                    CONV_STREAMING_HOST_IO,
                    DWG_NAIVE, DWG_ACCEL)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError("expected a CUDA/HIP tensor")
    if t.dtype != dtype:
        raise TypeError("expected dtype %s, got %s" % (dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


def _acc(t, dtype=torch.float32):
    """Device-ACCESSIBLE memory: a device tensor, or a pinned host tensor (hipHostMalloc memory is
    mapped into the device's address space, so a kernel can read and write it over PCIe)."""
    if isinstance(t, torch.Tensor) and not t.is_cuda and t.is_pinned():
        if t.dtype != dtype:
            raise TypeError("expected dtype %s, got %s" % (dtype, t.dtype))
        if not t.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return C.c_void_p(t.data_ptr())
    return _dev(t, dtype)


def _view(ptr, rows, cols):
    """A torch tensor over library-owned device memory (no ownership: keep the plan alive)."""
    class _Mem:
        pass
    m = _Mem()
    m.__cuda_array_interface__ = {"shape": (rows, cols), "typestr": "<f4", "data": (ptr, False), "version": 2}
    return torch.as_tensor(m, device="cuda")


def device_count():
    n = C.c_int(0)
    check(lib.gab_device_count(C.byref(n)))
    return n.value


def noop(x, out=None):
    out = torch.empty_like(x) if out is None else out
    check(lib.gab_noop(_dev(x), _dev(out), x.numel(), _stream()))
    return out


def gain(x, g=2.0, out=None):
    out = torch.empty_like(x) if out is None else out
    check(lib.gab_gain(_dev(x), _dev(out), x.numel(), g, _stream()))
    return out


def gainstats(x, tracks, bufsize, g=0.5):
    out = torch.empty_like(x)
    stats = torch.empty(2 * tracks, dtype=torch.float32, device=x.device)
    check(lib.gab_gainstats(_dev(x), _dev(out), _dev(stats), tracks, bufsize, g, _stream()))
    return out, stats


def datatransfer(x, out_size):
    out = torch.empty(out_size, dtype=torch.float32, device=x.device)
    check(lib.gab_datatransfer(_dev(x) if x.numel() else None, _dev(out) if out_size else None,
                               x.numel(), out_size, _stream()))
    return out


def _placement(fn, handle):
    import numpy as np
    cap = 256
    hw, xc = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
    n = C.c_int(0)
    check(fn(handle, hw.ctypes.data_as(C.c_void_p), xc.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
    return [dict(xcc=int(x & 0xf), se=int((h >> 13) & 7), sa=int((h >> 12) & 1), cu=int((h >> 8) & 0xf),
                 simd=int((h >> 4) & 3), slot=int(h & 0xf)) for h, x in zip(hw[:n.value], xc[:n.value])]


def placement_summary(places):
    """'8 waves on 8 XCDs: x0/se1/cu3 ...' — one line for a measurement's record."""
    if not places:
        return "no wave has started"
    return "%d waves on %d XCDs: %s" % (len(places), len({p["xcc"] for p in places}),
                                        " ".join("x%d/se%d/cu%d" % (p["xcc"], p["se"], p["cu"]) for p in places[:16]) +
                                        (" ..." if len(places) > 16 else ""))


class KeepWarm:
    """gab_keep_warm: a small resident launch that keeps the device from going idle between real-time slots
    (kick() once per slot; it ends by itself idle_seconds after the last kick)."""

    def __init__(self, workgroups=8, idle_seconds=0.25):
        h = C.c_void_p()
        check(lib.gab_keep_warm_create(C.byref(h), int(workgroups), float(idle_seconds)))
        self._h = h

    def kick(self):
        check(lib.gab_keep_warm_kick(self._h))

    def running(self):
        v = C.c_int(0)
        check(lib.gab_keep_warm_running(self._h, C.byref(v)))
        return bool(v.value)

    def placement(self):
        """Where the waves of the current (or last) launch landed: a list of dicts (xcc, se, sa, cu, simd, slot) for every
        wave that has started (gab_keep_warm_placement; HW_ID / XCC_ID as the hardware reports them)."""
        return _placement(lib.gab_keep_warm_placement, self._h)

    def close(self):
        if self._h:
            lib.gab_keep_warm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LinkPlan:
    """gab_link_plan: the staging buffer, words and upload stream of datatransfer with both link directions
    busy at once (gab_datatransfer_round_trip)."""

    def __init__(self, max_in_size):
        h = C.c_void_p()
        check(lib.gab_link_plan_create(int(max_in_size), C.byref(h)))
        self._h, self.max_in_size = h, int(max_in_size)

    def round_trip(self, h_in, h_out, stream=None):
        """Pinned host tensor in -> pinned host tensor out; returns when h_out is complete and h_in uploaded."""
        if h_in.is_cuda or h_out.is_cuda or (h_out.numel() and not h_out.is_pinned()):
            raise TypeError("round_trip takes host tensors; the output must be pinned")
        if h_in.dtype != torch.float32 or h_out.dtype != torch.float32 or not h_in.is_contiguous() or not h_out.is_contiguous():
            raise TypeError("round_trip takes contiguous float32 tensors")
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        check(lib.gab_datatransfer_round_trip(self._h, C.c_void_p(h_in.data_ptr()) if h_in.numel() else None,
                                              C.c_void_p(h_out.data_ptr()) if h_out.numel() else None,
                                              h_in.numel(), h_out.numel(), st))
        return h_out

    def check(self):
        """The verdict of the check launch behind the last round trip (gab_datatransfer_round_trip_check)."""
        check(lib.gab_datatransfer_round_trip_check(self._h))

    def close(self):
        if self._h:
            lib.gab_link_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def iir(x, coeffs, state, tracks, bufsize, sequential=False):
    """state (tracks*2, device) is updated in place.  sequential=True forces the
    lane-per-track kernel that is bit-identical to the golden."""
    out = torch.empty_like(x)
    c = (C.c_float * 5)(*[float(v) for v in coeffs])
    fn = lib.gab_iir_sequential if sequential else lib.gab_iir
    check(fn(_dev(x), _dev(out), c, _dev(state), tracks, bufsize, _stream()))
    return out


def conv1d(x, ir, ir_len, tracks, bufsize):
    out = torch.empty(tracks * bufsize, dtype=torch.float32, device=x.device)
    check(lib.gab_conv1d(_dev(x), _dev(out), _dev(ir), ir_len, tracks, bufsize, _stream()))
    return out


def rndmem(pool, playheads, tracks, bufsize):
    out = torch.empty(tracks * bufsize, dtype=torch.float32, device=pool.device)
    check(lib.gab_rndmem(_dev(pool), _dev(playheads, torch.int32), _dev(out), tracks, bufsize,
                         _stream()))
    return out


def modal(params, n_modes, bufsize, out_tracks=32):
    out = torch.zeros(out_tracks * bufsize, dtype=torch.float32, device=params.device)
    check(lib.gab_modal(_dev(params), _dev(out), n_modes, bufsize, out_tracks, _stream()))
    return out


def modal_bank_workspace(n_modes, bufsize, out_tracks=32, device="cuda"):
    nbytes = lib.gab_modal_bank_workspace_bytes(n_modes, out_tracks, bufsize)
    return torch.empty(max(nbytes, 4) // 4, dtype=torch.float32, device=device)


def modal_bank(params, n_modes, bufsize, out_tracks=32, out=None, workspace=None):
    """The real bank (Metal kernel semantics) on 8-float mode records; returns [out_tracks*bufsize]."""
    if out is None:
        out = torch.empty(out_tracks * bufsize, dtype=torch.float32, device=params.device)
    if workspace is None:
        workspace = modal_bank_workspace(n_modes, bufsize, out_tracks, params.device)
    check(lib.gab_modal_bank(_dev(params), _dev(out), n_modes, bufsize, out_tracks, _dev(workspace), _stream()))
    return out


def dwg(wg_bytes, fwd, bwd, x, bufsize, max_len=2000, out_tracks=None, variant=DWG_ACCEL):
    """wg_bytes: uint8 device tensor holding n_wg WaveguideState records (32 B each)."""
    n_wg = wg_bytes.numel() // C.sizeof(WaveguideState)
    out = torch.empty(bufsize, dtype=torch.float32, device=x.device)
    ws = torch.empty(lib.gab_dwg_workspace_bytes(n_wg, bufsize), dtype=torch.uint8, device=x.device)
    ot = n_wg if out_tracks is None else out_tracks
    check(lib.gab_dwg(_dev(wg_bytes, torch.uint8), _dev(fwd), _dev(bwd), _dev(x), _dev(out),
                      _dev(ws, torch.uint8), n_wg, bufsize, max_len, ot, variant, _stream()))
    return out


def fft_r2c_1024(x, tracks):
    """x: tracks*1024 real -> (tracks, 513, 2) interleaved complex."""
    out = torch.empty(tracks * 513 * 2, dtype=torch.float32, device=x.device)
    check(lib.gab_fft_r2c_1024(_dev(x), _dev(out), tracks, _stream()))
    return out.view(tracks, 513, 2)


class ConvPlan:
    """Conv1DAccelBenchmark's device side: spectra bank + history + process()."""

    def __init__(self, tracks, bufsize, ir_len, scheme=None):
        """scheme: None (the library's default for this shape), "classic" or "split"
        (gab_conv_set_scheme)."""
        self.tracks, self.bufsize, self.ir_len = tracks, bufsize, ir_len
        self._h = C.c_void_p()
        check(lib.gab_conv_create(C.byref(self._h), tracks, bufsize, ir_len))
        if scheme is not None:
            self.set_scheme(scheme)

    def set_scheme(self, scheme):
        check(lib.gab_conv_set_scheme(self._h, {"classic": 0, "split": 1}[scheme]))

    @property
    def scheme(self):
        v = C.c_int(0)
        check(lib.gab_conv_get_scheme(self._h, C.byref(v)))
        return "split" if v.value == 1 else "classic"

    def set_ir(self, ir):
        assert ir.numel() == self.tracks * self.ir_len
        check(lib.gab_conv_set_ir(self._h, _dev(ir), _stream()))

    def reset(self):
        check(lib.gab_conv_reset(self._h, _stream()))

    def process(self, x, out=None, mode=CONV_STREAMING):
        assert x.numel() == self.tracks * self.bufsize
        if out is None:
            out = torch.empty(self.tracks * self.bufsize, dtype=torch.float32,
                              device=x.device if x.is_cuda else "cuda")
        # x / out may also be pinned host tensors: the kernel then streams the buffer over PCIe
        # itself (zero-copy), without separate copy commands — under its own kernel name
        if mode == CONV_STREAMING and not x.is_cuda and not out.is_cuda:
            mode = CONV_STREAMING_HOST_IO
        check(lib.gab_conv_process(self._h, _acc(x), _acc(out), mode, _stream()))
        return out

    def round_trip(self, h_in, h_out, stream=None):
        """One buffer, pinned host tensor in -> pinned host tensor out; returns when h_out is complete
        (gab_conv_round_trip)."""
        assert h_in.numel() == self.tracks * self.bufsize == h_out.numel()
        if h_in.is_cuda or h_out.is_cuda or not h_out.is_pinned():
            raise TypeError("round_trip takes host tensors; the output must be pinned")
        if h_in.dtype != torch.float32 or h_out.dtype != torch.float32 or not h_in.is_contiguous() or not h_out.is_contiguous():
            raise TypeError("round_trip takes contiguous float32 tensors")
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        check(lib.gab_conv_round_trip(self._h, C.c_void_p(h_in.data_ptr()), C.c_void_p(h_out.data_ptr()), st))
        return h_out

    def round_trip_check(self):
        """The verdict of the check launch behind the last round trip (gab_conv_round_trip_check): raises if a word the
        kernel consumed early is not what the completed upload left."""
        check(lib.gab_conv_round_trip_check(self._h))

    def round_trip_set_check(self, mode):
        """0 ignore the verdict, 1 (default) read it at the next call / round_trip_check(), 2 read it in the call."""
        check(lib.gab_conv_round_trip_set_check(self._h, int(mode)))

    def round_trip_keep_warm(self, on=True):
        """Every later round trip of this plan ends with a keep-warm kick (gab_conv_round_trip_keep_warm)."""
        check(lib.gab_conv_round_trip_keep_warm(self._h, 1 if on else 0))

    def round_trip_keep_warm_placement(self):
        """KeepWarm.placement() of the plan's own keep-warm launch ([] if it has none)."""
        return _placement(lib.gab_conv_round_trip_keep_warm_placement, self._h)

    def newest_block(self):
        """The block the plan consumed last ([tracks*512], the input's layout), from its history ring
        (gab_conv_newest_block): what a round trip's upload hand-off is checked against."""
        out = torch.empty(self.tracks * self.bufsize, dtype=torch.float32, device="cuda")
        check(lib.gab_conv_newest_block(self._h, _dev(out), _stream()))
        return out

    def prepare_round_trip(self, h_in, h_out, stream=None):
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        return (self._h, C.c_void_p(h_in.data_ptr()), C.c_void_p(h_out.data_ptr()), st)

    @staticmethod
    def launch_round_trip(args):
        check(lib.gab_conv_round_trip(*args))

    # ---- the doorbell-fed resident engine (gab_conv_engine_*) ----
    def engine_rings(self, ring_buffers):
        """The engine's rings without starting it (to fill resident input before the launch takes the device)."""
        a, b = C.c_void_p(), C.c_void_p()
        check(lib.gab_conv_engine_rings(self._h, ring_buffers, C.byref(a), C.byref(b)))
        n = self.tracks * self.bufsize
        return _view(a.value, ring_buffers, n), _view(b.value, ring_buffers, n)

    def engine_start(self, ring_buffers, stream=None):
        """Launches the resident engine; returns (in_ring, out_ring) as device tensors [ring][T*B] / [ring][B*T]
        viewing the plan's rings."""
        a, b = C.c_void_p(), C.c_void_p()
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        check(lib.gab_conv_engine_start(self._h, ring_buffers, C.byref(a), C.byref(b), st))
        n = self.tracks * self.bufsize
        return _view(a.value, ring_buffers, n), _view(b.value, ring_buffers, n)

    def engine_publish(self, n_more=1):
        check(lib.gab_conv_engine_publish(self._h, n_more))

    def engine_submit(self, n_more=1, flush=True):
        """Publish n_more buffers; flush=True also rings the flush rung: finish what is published without
        waiting for more (the real-time form, one buffer in flight)."""
        check(lib.gab_conv_engine_submit(self._h, n_more, 1 if flush else 0))

    def engine_wait(self, count, timeout=10.0):
        """Spins until `count` buffers are reported complete (gab_conv_engine_wait)."""
        check(lib.gab_conv_engine_wait(self._h, count, float(timeout)))

    def engine_running(self):
        """True while the resident launch is still on the device (gab_conv_engine_running)."""
        v = C.c_int(0)
        check(lib.gab_conv_engine_running(self._h, C.byref(v)))
        return bool(v.value)

    def engine_completed(self):
        v = C.c_int(0)
        check(lib.gab_conv_engine_completed(self._h, C.byref(v)))
        return v.value

    def engine_feed(self, n_buffers, ahead=4):
        check(lib.gab_conv_engine_feed(self._h, n_buffers, ahead))

    def engine_feed_one_in_flight(self, n_buffers, latencies=None):
        """n_buffers times { doorbell with the flush rung; wait for that buffer } on resident rings; latencies: a
        float32 numpy array of n_buffers (host clock, us) or None."""
        ptr = None
        if latencies is not None:
            assert latencies.dtype.name == "float32" and latencies.size >= n_buffers and latencies.flags["C_CONTIGUOUS"]
            ptr = latencies.ctypes.data_as(C.c_void_p)
        check(lib.gab_conv_engine_feed_one_in_flight(self._h, n_buffers, ptr))

    def engine_stop(self):
        check(lib.gab_conv_engine_stop(self._h))

    def engine_round_trip(self, h_in, h_out):
        """Pinned host -> ring slot -> engine (flush rung, ONE buffer in flight) -> ring slot -> pinned host
        (gab_conv_engine_round_trip): the reference's iteration through the resident engine."""
        assert h_in.is_pinned() and h_out.is_pinned() and h_in.numel() == h_out.numel() == self.tracks * self.bufsize
        check(lib.gab_conv_engine_round_trip(self._h, C.c_void_p(h_in.data_ptr()), C.c_void_p(h_out.data_ptr())))

    def engine_set_idle_limit(self, seconds):
        """How long a stalled engine waits for the doorbell before it ends by itself (taken at the next start)."""
        check(lib.gab_conv_engine_set_idle_limit(self._h, float(seconds)))

    def process_batch(self, x, n_buffers, out=None):
        """n_buffers consecutive buffers ([n][T*B] in, [n][B*T] out) in one launch."""
        assert x.numel() == n_buffers * self.tracks * self.bufsize
        if out is None:
            out = torch.empty(n_buffers * self.tracks * self.bufsize, dtype=torch.float32, device=x.device)
        check(lib.gab_conv_process_batch(self._h, _dev(x), _dev(out), n_buffers, _stream()))
        return out

    def prepare_batch(self, x, n_buffers, out, stream=None):
        """The ctypes arguments of process_batch(), built once for a loop over the same resident batch."""
        assert x.numel() == n_buffers * self.tracks * self.bufsize == out.numel()
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        return (self._h, _dev(x), _dev(out), n_buffers, st)

    @staticmethod
    def launch_batch(args):
        check(lib.gab_conv_process_batch(*args))

    def prepare(self, x, out, mode=CONV_STREAMING, stream=None):
        """The ctypes arguments of process(), built once for a loop that cycles through a fixed set
        of buffers; `launch(args)` then costs ~4 us of host time instead of ~7.5."""
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        return (self._h, _dev(x), _dev(out), mode, st)

    @staticmethod
    def launch(args):
        check(lib.gab_conv_process(*args))

    def state_bytes(self):
        a, b = C.c_size_t(0), C.c_size_t(0)
        check(lib.gab_conv_state_bytes(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def close(self):
        if self._h:
            lib.gab_conv_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fdtd_default_params(nx, ny=None, nz=None):
    P = FdtdParams()
    check(lib.gab_fdtd_default_params(nx, nx if ny is None else ny, nx if nz is None else nz,
                                      C.byref(P)))
    return P


class FdtdPlan:
    def __init__(self, params):
        self.params = params
        self._h = C.c_void_p()
        check(lib.gab_fdtd_create(C.byref(self._h), C.byref(params)))

    def reset(self):
        check(lib.gab_fdtd_reset(self._h, _stream()))

    def process(self, x, out, tracks, bufsize, first_sample, n_samples):
        check(lib.gab_fdtd_process(self._h, _dev(x), _dev(out), tracks, bufsize, first_sample,
                                   n_samples, _stream()))
        return out

    def set_form(self, form):
        """"auto" (resident in LDS where the room fits) or "step" (one launch per step): gab_fdtd_set_form."""
        check(lib.gab_fdtd_set_form(self._h, {"auto": 0, "step": 1}[form]))

    def status(self):
        """Synchronises the current stream; raises GabError if the last process() call's resident launch
        gave up waiting for a neighbour workgroup: gab_fdtd_status."""
        check(lib.gab_fdtd_status(self._h, _stream()))

    def resident(self):
        """(takes the LDS-resident whole-buffer kernel, its workgroups): gab_fdtd_resident."""
        r, w = C.c_int(0), C.c_int(0)
        check(lib.gab_fdtd_resident(self._h, C.byref(r), C.byref(w)))
        return bool(r.value), w.value

    def set_track_positions(self, src_xyz, rcv_xyz):
        """Per-track source and receiver cells: two (tracks, 3) integer arrays of (x, y, z);
        pass None, None to return to the shared cells of the params."""
        import numpy as np
        if src_xyz is None:
            check(lib.gab_fdtd_set_track_positions(self._h, None, None, 0))
            return
        src = np.ascontiguousarray(src_xyz, np.int32)
        rcv = np.ascontiguousarray(rcv_xyz, np.int32)
        if src.ndim != 2 or src.shape[1] != 3 or rcv.shape != src.shape:
            raise ValueError("positions must be two (tracks, 3) arrays")
        ip = C.POINTER(C.c_int)
        check(lib.gab_fdtd_set_track_positions(self._h, src.ctypes.data_as(ip), rcv.ctypes.data_as(ip),
                                               src.shape[0]))

    def pressure(self):
        """A copy of the pressure grid as a torch tensor (nz, ny, nx)."""
        P = self.params
        out = torch.empty(P.nx * P.ny * P.nz, dtype=torch.float32, device="cuda")
        check(lib.gab_fdtd_copy_pressure(self._h, _dev(out), _stream()))
        return out.view(P.nz, P.ny, P.nx)

    def close(self):
        if self._h:
            lib.gab_fdtd_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
