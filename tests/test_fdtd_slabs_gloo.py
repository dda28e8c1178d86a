"""The z-slab schedule of gpuaudiobench_amd.fdtd_slabs on CPU: three gloo ranks,
each with a torch stand-in for the device slab, exchange planes with their neighbours and must
reproduce the uncut grid exactly.  What is under test is the schedule and the halo bookkeeping
(which plane goes where, and when); the device arithmetic itself is checked in the GPU suite."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gpuaudiobench_amd import fdtd_slabs as fs


class TorchSlab:
    """Planes [z_begin, z_end) of the staggered-grid scheme (cuda/bench_fdtd3d.cu:9-160 as the
    oracle restates it), with the same ghost layout and halo selectors as the device slab."""

    def __init__(self, params, z_begin, z_end):
        P = self.params = params
        self.z_begin, self.z_end = z_begin, z_end
        nzl = z_end - z_begin
        f = dict(dtype=torch.float32)
        self.p = torch.zeros(nzl + 2, P.ny, P.nx, **f)          # planes z_begin-1 .. z_end
        self.vz = torch.zeros(nzl + 1, P.ny, P.nx, **f)         # faces  z_begin .. z_end
        self.vx = torch.zeros(nzl, P.ny, P.nx + 1, **f)
        self.vy = torch.zeros(nzl, P.ny + 1, P.nx, **f)
        self.owns_source = z_begin <= P.source_z < z_end
        self.owns_receiver = z_begin <= P.receiver_z < z_end
        self._strip = None
        zs = torch.arange(z_begin, z_end).view(-1, 1, 1)
        ys = torch.arange(P.ny).view(1, -1, 1)
        xs = torch.arange(P.nx).view(1, 1, -1)
        self.interior = ((zs > 0) & (zs < P.nz - 1) & (ys > 0) & (ys < P.ny - 1) & (xs > 0) & (xs < P.nx - 1))

    def source_sums(self, x, tracks, bufsize):
        acc = torch.zeros(bufsize)
        for t in range(tracks):                                  # track order, one rounding per add
            acc = acc + x.view(tracks, bufsize)[t] * np.float32(0.1)
        self.inj = acc
        self._strip = torch.zeros(bufsize)

    def inject(self, smp):
        P = self.params
        if self.owns_source:
            self.p[P.source_z - self.z_begin + 1, P.source_y, P.source_x] += self.inj[smp]

    def step(self, strip_sample=-1):
        P, p = self.params, self.p
        c1, c2 = np.float32(P.dt_over_rho_dx), np.float32(P.rho_c2_dt_over_dx)
        own = p[1:-1]
        vx, vy, vz = self.vx.clone(), self.vy.clone(), self.vz.clone()
        vx[:, :, 1:P.nx] = self.vx[:, :, 1:P.nx] - c1 * (own[:, :, 1:] - own[:, :, :-1])
        vy[:, 1:P.ny, :] = self.vy[:, 1:P.ny, :] - c1 * (own[:, 1:, :] - own[:, :-1, :])
        low = self.vz[:-1] - c1 * (own - p[:-2])
        if self.z_begin == 0:
            low[0] = self.vz[0]                                   # the floor's face is carried
        vz[:-1] = low
        top = self.vz[-1] - c1 * (p[-1] - own[-1])                # the upper neighbour's face, recomputed
        hz = torch.cat([low[1:], top[None]])
        div = ((vx[:, :, 1:] - vx[:, :, :-1]) + (vy[:, 1:, :] - vy[:, :-1, :])) + (hz - low)
        new = torch.where(self.interior, own - c2 * div, own * np.float32(1.0 - P.absorption_coeff))
        if strip_sample >= 0 and self.owns_receiver:
            self._strip[strip_sample] = new[P.receiver_z - self.z_begin, P.receiver_y, P.receiver_x] * np.float32(0.1)
        p[1:-1] = new
        self.vx, self.vy, self.vz = vx, vy, vz

    def halo(self, which):
        nzl = self.z_end - self.z_begin
        return {fs.SEND_DOWN_P: self.p[1], fs.SEND_DOWN_VZ: self.vz[0], fs.SEND_UP_P: self.p[nzl],
                fs.RECV_DOWN_P: self.p[0], fs.RECV_UP_P: self.p[nzl + 1], fs.RECV_UP_VZ: self.vz[nzl]}[which].view(-1)

    def strip(self):
        return self._strip

    def pressure(self):
        return self.p[1:-1]


def _params(n):
    from gpuaudiobench_amd import fdtd_default_params
    return fdtd_default_params(n)


def _whole(n, x, T, B, buffers):
    P = _params(n)
    s = TorchSlab(P, 0, P.nz)
    outs = []
    for _ in range(buffers):
        out = torch.zeros(T * B)
        fs.process_ranks(s, x, out, T, B, 0, 1, None)
        outs.append(out)
    return outs, s.pressure().clone()


def test_local_exchange_matches_the_uncut_grid(orc):
    n, T, B = 20, 3, 45
    x = torch.from_numpy(orc.Rand(5).bipolar(T * B))
    want, field = _whole(n, x, T, B, 1)
    P = _params(n)
    slabs = [TorchSlab(P, a, b) for a, b in fs.slab_ranges(P.nz, 3)]
    for s in slabs:
        s.source_sums(x, T, B)
    fs.run_buffer(slabs, B, P.steps_per_sample, lambda: fs.exchange_local(slabs))
    got = next(s for s in slabs if s.owns_receiver).strip()
    assert torch.equal(got, want[0].view(T, B)[0])
    assert torch.equal(torch.cat([s.pressure() for s in slabs]), field)
    assert got.abs().max() > 0
    # the stand-in against the oracle's grid: same scheme (it contracts nothing, the oracle uses fmaf)
    op = orc.fdtd_params(n)
    grids = orc.fdtd_grids(op)
    ref = np.zeros(T * B, np.float32)
    orc.fdtd(op, grids, x.numpy(), ref, T, B, 0, B, fused=True)
    assert np.abs(ref[:B] - got.numpy()).max() <= 1e-4 * np.abs(ref).max()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, T, B, buffers, seed, q):
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = _params(n)
        a, b = fs.slab_ranges(P.nz, world)[rank]
        slab = TorchSlab(P, a, b)
        x = torch.from_numpy(oracle.Rand(seed).bipolar(T * B))
        outs = []
        for _ in range(buffers):
            out = torch.zeros(T * B)
            fs.process_ranks(slab, x, out, T, B, rank, world, dist)
            outs.append(out.numpy().copy())
        gathered = [None] * world
        dist.all_gather_object(gathered, (a, b, outs, slab.pressure().numpy().copy()))
        if rank == 0:
            q.put(gathered)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_rank_exchange_matches_the_uncut_grid(orc, world):
    n, T, B, buffers, seed = 20, 2, 24, 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, T, B, buffers, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want, field = _whole(n, torch.from_numpy(orc.Rand(seed).bipolar(T * B)), T, B, buffers)
    assert [g[:2] for g in gathered] == fs.slab_ranges(n, world)
    for r in range(world):                                       # every rank returns the whole output
        for k in range(buffers):
            assert np.array_equal(gathered[r][2][k].view(np.uint32), want[k].numpy().view(np.uint32))
    assert np.array_equal(np.concatenate([g[3] for g in gathered]), field.numpy())
    assert np.abs(want[-1].numpy()).max() > 0


def _gpu_worker(rank, world, port, n, T, B, buffers, seed, q):
    import oracle
    import gpuaudiobench_amd as gab
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)                                 # every rank on the one device
        P = gab.fdtd_default_params(n)
        a, b = fs.slab_ranges(P.nz, world)[rank]
        slab = fs.FdtdSlab(P, a, b)
        x = torch.from_numpy(oracle.Rand(seed).bipolar(T * B)).cuda()
        outs = []
        for _ in range(buffers):
            out = torch.zeros(T * B, device="cuda")
            fs.process_ranks(slab, x, out, T, B, rank, world, dist)
            outs.append(out.cpu().numpy())
        gathered = [None] * world
        dist.all_gather_object(gathered, (a, b, outs, slab.pressure().cpu().numpy()))
        if rank == 0:
            q.put(gathered)
        slab.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_device_slabs_on_two_ranks_match_the_single_grid(orc):
    """The multi-rank path with the real kernels: two processes, one device slab each (both on
    this box's one GPU, planes staged through the host because gloo moves host memory), against
    the single-grid plan and the oracle, bit for bit."""
    import gpuaudiobench_amd as gab
    n, T, B, buffers, seed, world = 52, 4, 16, 2, 11, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, n, T, B, buffers, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    x = orc.Rand(seed).bipolar(T * B)
    P = orc.fdtd_params(n)
    grids = orc.fdtd_grids(P)
    plan = gab.FdtdPlan(gab.fdtd_default_params(n))
    for k in range(buffers):
        ref = np.zeros(T * B, np.float32)
        orc.fdtd(P, grids, x, ref, T, B, 0, B, fused=True)
        out = torch.zeros(T * B, device="cuda")
        plan.process(torch.from_numpy(x).cuda(), out, T, B, 0, B)
        assert np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32))
        for r in range(world):
            assert np.array_equal(gathered[r][2][k].view(np.uint32), ref.view(np.uint32))
    field = np.concatenate([g[3] for g in gathered]).ravel()
    assert np.array_equal(field.view(np.uint32), grids[0].view(np.uint32))
    plan.close()
