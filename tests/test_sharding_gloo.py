"""The N>1 path on CPU: two gloo ranks broadcast the IR bank, slice their
channels, and (with the oracle standing in for the device kernel) reproduce the
unsharded result column for column."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gpuaudiobench_amd import sharding, harness


def test_shard_ranges_cover_everything():
    for total, world in ((8192, 8), (1024, 1), (10, 4), (7, 8)):
        spans = [sharding.shard_range(r, world, total) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    with pytest.raises(ValueError):
        sharding.shard_range(3, 2, 10)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, T, B, L, q, granule=1, distribution="broadcast"):
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ir = sharding.broadcast_ir_bank(L, T, rank, world, torch.device("cpu"), dist, granule=granule, distribution=distribution)
        lo, hi = sharding.shard_range(rank, world, T, granule)
        # the slice a rank receives is what it would have generated for itself
        assert np.array_equal(ir.numpy().ravel(), harness.conv_accel_ir(L, hi - lo, lo, T))
        outs = []
        hist = np.zeros((hi - lo) * L, np.float32)
        for n in range(3):
            x = sharding.shard_noise(T, B, rank, world, seed=42 + n, granule=granule)
            outs.append(oracle.conv_accel_stream(x.ravel(), ir.numpy().ravel(), hist, L, B, hi - lo))
        # no data-path collective: results only meet here, for the check
        gathered = [None] * world
        dist.all_gather_object(gathered, (lo, hi, outs))
        if rank == 0:
            q.put(gathered)
    finally:
        dist.destroy_process_group()


def _run_sharded(orc, T, B, L, world, granule=1, distribution="broadcast", buffers=3):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, T, B, L, q, granule, distribution)) for r in range(world)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ir = orc.conv_accel_ir(L, T)
    hist = np.zeros(T * L, np.float32)
    for n in range(buffers):
        x = orc.noise(T * B, seed=42 + n)
        full = orc.conv_accel_stream(x, ir, hist, L, B, T)
        glob = np.zeros(T * B, np.float32)
        for r, (lo, hi, outs) in enumerate(gathered):
            assert (lo, hi) == sharding.shard_range(r, world, T, granule)
            sharding.scatter_columns(glob, outs[n], r, world, T, B, granule)
        assert np.array_equal(glob.view(np.uint32), full.view(np.uint32))
    return gathered


def test_two_rank_sharding_matches_unsharded(orc):
    _run_sharded(orc, T=12, B=64, L=200, world=2)


def test_helpers_cut_where_the_shards_cut():
    """broadcast_ir_bank / shard_noise / scatter_columns take the shard's granule: with a total that is no multiple of
    world x granule (1026 tracks, 4 ranks, duos of 4) the aligned ranges differ from the granule-1 ranges, and helpers
    cutting at granule 1 would hand a shard the wrong rows."""
    T, world, g = 1026, 4, sharding.shard_granule("Conv1D_accel")
    aligned = [sharding.shard_range(r, world, T, g) for r in range(world)]
    plain = [sharding.shard_range(r, world, T) for r in range(world)]
    assert aligned != plain
    assert aligned[0][0] == 0 and aligned[-1][1] == T and all(a[1] == b[0] for a, b in zip(aligned, aligned[1:]))
    assert all(lo % g == 0 for lo, _ in aligned)
    B = 4
    flat = harness.noise(T * B, 42).reshape(T, B)
    glob = np.zeros(T * B, np.float32)
    for r, (lo, hi) in enumerate(aligned):
        x = sharding.shard_noise(T, B, r, world, seed=42, granule=g)
        assert np.array_equal(x, flat[lo:hi])
        ir = sharding.broadcast_ir_bank(8, T, r, world, torch.device("cpu"), None, granule=g)
        assert np.array_equal(ir.numpy().ravel(), harness.conv_accel_ir(8, hi - lo, lo, T))
        sharding.scatter_columns(glob, np.ascontiguousarray(x.T), r, world, T, B, granule=g)    # rows stand in for outputs
    assert np.array_equal(glob.reshape(B, T), flat.T)


@pytest.mark.parametrize("distribution", ["broadcast", "slices"])
def test_four_ranks_uneven_duo_aligned_shards(orc, distribution):
    """42 tracks over 4 ranks cut at duos (12, 12, 12, 6): uneven shards, the bank as
    a broadcast and as per-rank slices (point-to-point when the shards are uneven) — same slices, same columns."""
    _run_sharded(orc, T=42, B=64, L=200, world=4, granule=4, distribution=distribution, buffers=2)


def test_eight_rank_rehearsal_of_c5_shapes(orc):
    """BASELINE configs[4] on the CPU path: 8192 channels over 8 gloo ranks, 1024 per rank, the 4096-tap bank (128 MiB)
    distributed as per-rank slices at GLOBAL indices, each rank's rows of the one flat noise stream, the first buffer
    convolved by the oracle standing in for the device (reference semantics: taps k <= s), columns scattered with the
    global stride — equal to the unsharded golden on every channel.  No GPU: what `bench.py --gpus 8` does around its
    launches, at the real sizes."""
    T, B, L, world = 8192, 512, 4096, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_c5_worker, args=(r, world, port, T, B, L, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    x = orc.noise(T * B, seed=42)
    full = orc.conv_accel(x, orc.conv_accel_ir(L, T), L, B, T)
    glob = np.zeros(T * B, np.float32)
    for r, (lo, hi, y) in enumerate(gathered):
        assert (lo, hi) == (1024 * r, 1024 * (r + 1))
        sharding.scatter_columns(glob, y, r, world, T, B, granule=4)
    assert np.array_equal(glob.view(np.uint32), full.view(np.uint32))


def _c5_worker(rank, world, port, T, B, L, q):
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = sharding.shard_granule("Conv1D_accel")
        ir = sharding.broadcast_ir_bank(L, T, rank, world, torch.device("cpu"), dist, granule=g, distribution="slices")
        lo, hi = sharding.shard_range(rank, world, T, g)
        assert ir.shape == (hi - lo, L)
        x = sharding.shard_noise(T, B, rank, world, seed=42, granule=g)
        y = oracle.conv_accel(x.ravel(), ir.numpy().ravel(), L, B, hi - lo)
        gathered = [None] * world
        dist.all_gather_object(gathered, (lo, hi, y))
        if rank == 0:
            q.put(gathered)
    finally:
        dist.destroy_process_group()
