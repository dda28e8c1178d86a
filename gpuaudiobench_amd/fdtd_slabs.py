"""FDTD3D cut into z-slabs.

The reference runs the room as one grid on one device and only remarks that it could be
tiled (metal-swift kernels_fdtd3d.metal:184,217).  Here the grid is cut along z — the slowest
axis, so a halo is one contiguous nx*ny plane — into slabs that each own planes
[z_begin, z_end) and keep one ghost pressure plane on either side and one ghost layer of vz
faces above.  One leapfrog step of a slab reads, besides its own planes,

    p(z_begin - 1)            from the slab below   (its top plane)
    p(z_end), vz(z_end)       from the slab above   (its bottom plane and bottom faces)

so the schedule of one audio sample is

    inject (the source's owner adds the sample's sum into its cell)
    steps_per_sample times:  exchange the three planes  ->  step every slab
    (the last step stores 0.1 * p[receiver] on the receiver's owner)

which performs exactly the arithmetic of the single-grid plan, cell for cell, so the result is
bit-identical to it for any cut.  The exchange is the only communication: device-to-device
copies inside one process (`exchange_local`), point-to-point sends between neighbour ranks
otherwise (`exchange_ranks`; RCCL over xGMI on GPUs, gloo in the CPU tests) — a chain, no
collective.  Per step a rank moves 3 planes out and 3 in: 24 * nx * ny bytes.
"""
import ctypes as C

import torch

from ._capi import lib, check
from .sharding import shard_range

SEND_DOWN_P, SEND_DOWN_VZ, SEND_UP_P, RECV_DOWN_P, RECV_UP_P, RECV_UP_VZ = range(6)


def slab_ranges(nz, parts):
    """[z_begin, z_end) per slab: contiguous, the remainder planes go to the low slabs."""
    if parts < 1 or parts > nz:
        raise ValueError("cannot cut %d planes into %d slabs" % (nz, parts))
    return [shard_range(k, parts, nz) for k in range(parts)]


class _DevicePlane:
    """A device address as something torch.as_tensor can wrap without copying."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = dict(shape=(n,), typestr="<f4", data=(ptr, False), version=2)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class FdtdSlab:
    """One slab of the room on the current device (gab_fdtd_create_slab and friends)."""

    def __init__(self, params, z_begin, z_end):
        self.params, self.z_begin, self.z_end = params, z_begin, z_end
        self._h = C.c_void_p()
        check(lib.gab_fdtd_create_slab(C.byref(self._h), C.byref(params), z_begin, z_end))
        a, b = C.c_int(0), C.c_int(0)
        check(lib.gab_fdtd_owns(self._h, C.byref(a), C.byref(b)))
        self.owns_source, self.owns_receiver = bool(a.value), bool(b.value)

    def reset(self):
        check(lib.gab_fdtd_reset(self._h, _stream()))

    def source_sums(self, x, tracks, bufsize):
        check(lib.gab_fdtd_source_sums(self._h, C.c_void_p(x.data_ptr()), tracks, bufsize, _stream()))

    def inject(self, sample):
        check(lib.gab_fdtd_inject(self._h, sample, _stream()))

    def step(self, strip_sample=-1):
        check(lib.gab_fdtd_step(self._h, strip_sample, _stream()))

    def halo(self, which):
        """Zero-copy view (nx*ny floats) of a halo plane of the CURRENT fields; the fields swap
        every step, so ask again after each one."""
        p, n = C.c_void_p(), C.c_size_t(0)
        check(lib.gab_fdtd_halo(self._h, which, C.byref(p), C.byref(n)))
        return torch.as_tensor(_DevicePlane(p.value, n.value), device="cuda")

    def emit(self, out, tracks, bufsize):
        check(lib.gab_fdtd_emit(self._h, C.c_void_p(out.data_ptr()), tracks, bufsize, _stream()))
        return out

    def strip(self):
        p, cap = C.c_void_p(), C.c_int(0)
        check(lib.gab_fdtd_strip(self._h, C.byref(p), C.byref(cap)))
        return torch.as_tensor(_DevicePlane(p.value, cap.value), device="cuda")

    def pressure(self):
        """A copy of the slab's own planes, (z_end - z_begin, ny, nx)."""
        P = self.params
        nzl = self.z_end - self.z_begin
        out = torch.empty(P.nx * P.ny * nzl, dtype=torch.float32, device="cuda")
        check(lib.gab_fdtd_copy_pressure(self._h, C.c_void_p(out.data_ptr()), _stream()))
        return out.view(nzl, P.ny, P.nx)

    def close(self):
        if self._h:
            lib.gab_fdtd_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- the schedule; works on anything with halo()/inject()/step() (the tests drive it with a
# ---- CPU stand-in under gloo) -----------------------------------------------------------------

def exchange_local(slabs):
    """All slabs in one process, bottom to top: neighbour planes are copied device to device."""
    for lo, hi in zip(slabs[:-1], slabs[1:]):
        hi.halo(RECV_DOWN_P).copy_(lo.halo(SEND_UP_P))
        lo.halo(RECV_UP_P).copy_(hi.halo(SEND_DOWN_P))
        lo.halo(RECV_UP_VZ).copy_(hi.halo(SEND_DOWN_VZ))


def exchange_ranks(slab, rank, world, dist):
    """One slab per rank, rank k below rank k+1: one batch of point-to-point sends and
    receives with the two neighbours.  Device planes go as they are over RCCL; a backend that
    only moves host memory (gloo, used to rehearse the multi-rank path on one device) gets them
    staged through host copies."""
    plan = []                                              # (is_send, plane, peer)
    if rank + 1 < world:
        plan += [(True, slab.halo(SEND_UP_P), rank + 1), (False, slab.halo(RECV_UP_P), rank + 1),
                 (False, slab.halo(RECV_UP_VZ), rank + 1)]
    if rank > 0:
        plan += [(False, slab.halo(RECV_DOWN_P), rank - 1), (True, slab.halo(SEND_DOWN_P), rank - 1),
                 (True, slab.halo(SEND_DOWN_VZ), rank - 1)]
    if not plan:
        return
    staged = plan[0][1].is_cuda and dist.get_backend() != "nccl"
    bufs = [(t.cpu() if snd else torch.empty(t.shape, dtype=t.dtype)) if staged else t for snd, t, _ in plan]
    ops = [dist.P2POp(dist.isend if snd else dist.irecv, b, peer) for (snd, _, peer), b in zip(plan, bufs)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    if staged:
        for (snd, t, _), b in zip(plan, bufs):
            if not snd:
                t.copy_(b)


def run_buffer(slabs, bufsize, steps_per_sample, exchange):
    """The per-buffer schedule over this process's slabs; `exchange()` refreshes every ghost."""
    for smp in range(bufsize):
        for s in slabs:
            s.inject(smp)
        for k in range(steps_per_sample):
            exchange()
            closes = k == steps_per_sample - 1
            for s in slabs:
                s.step(smp if closes else -1)


def process_local(slabs, x, out, tracks, bufsize):
    """One buffer through slabs that all live on this device; same result as FdtdPlan.process."""
    for s in slabs:
        s.source_sums(x, tracks, bufsize)
    run_buffer(slabs, bufsize, slabs[0].params.steps_per_sample, lambda: exchange_local(slabs))
    return next(s for s in slabs if s.owns_receiver).emit(out, tracks, bufsize)


def process_ranks(slab, x, out, tracks, bufsize, rank, world, dist):
    """One buffer with one slab per rank.  Every rank passes the same input buffer; the rank
    that owns the receiver broadcasts the B recorded samples so every rank returns the output."""
    slab.source_sums(x, tracks, bufsize)
    run_buffer([slab], bufsize, slab.params.steps_per_sample, lambda: exchange_ranks(slab, rank, world, dist))
    ranges = slab_ranges(slab.params.nz, world)
    owner = next(r for r, (a, b) in enumerate(ranges) if a <= slab.params.receiver_z < b)
    strip = slab.strip()[:bufsize]
    if world > 1:
        if strip.is_cuda and dist.get_backend() != "nccl":
            host = strip.cpu()
            dist.broadcast(host, src=owner)
            strip.copy_(host)
        else:
            dist.broadcast(strip, src=owner)
    out.view(tracks, bufsize)[:] = strip
    return out
