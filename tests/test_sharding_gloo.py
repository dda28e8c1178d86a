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


def _worker(rank, world, port, T, B, L, q):
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ir = sharding.broadcast_ir_bank(L, T, rank, world, torch.device("cpu"), dist)
        lo, hi = sharding.shard_range(rank, world, T)
        # the slice a rank receives is what it would have generated for itself
        assert np.array_equal(ir.numpy().ravel(), harness.conv_accel_ir(L, hi - lo, lo, T))
        outs = []
        hist = np.zeros((hi - lo) * L, np.float32)
        for n in range(3):
            x = sharding.shard_noise(T, B, rank, world, seed=42 + n)
            outs.append(oracle.conv_accel_stream(x.ravel(), ir.numpy().ravel(), hist, L, B, hi - lo))
        # no data-path collective: results only meet here, for the check
        gathered = [None] * world
        dist.all_gather_object(gathered, (lo, hi, outs))
        if rank == 0:
            q.put(gathered)
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_matches_unsharded(orc):
    T, B, L, world = 12, 64, 200, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, T, B, L, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ir = orc.conv_accel_ir(L, T)
    hist = np.zeros(T * L, np.float32)
    for n in range(3):
        x = orc.noise(T * B, seed=42 + n)
        full = orc.conv_accel_stream(x, ir, hist, L, B, T)
        glob = np.zeros(T * B, np.float32)
        for r, (lo, hi, outs) in enumerate(gathered):
            sharding.scatter_columns(glob, outs[n], r, world, T, B)
        assert np.array_equal(glob.view(np.uint32), full.view(np.uint32))
