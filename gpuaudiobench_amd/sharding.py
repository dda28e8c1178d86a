"""Channel sharding for the many-independent-channel benchmarks.

Tracks are independent, so a job of `total` tracks is cut into contiguous
shards, one per rank, and there is NO per-buffer collective.  The only exchange
is one-time: the impulse-response bank, whose formula needs the GLOBAL track
index and count, is generated once on rank 0 and broadcast (RCCL over xGMI on
GPUs; gloo in the CPU tests), then every rank keeps its slice.
"""
import numpy as np
import torch

from . import harness


def shard_range(rank, world, total_tracks):
    """Contiguous [lo, hi) of this rank; the remainder goes to the low ranks."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    base, extra = divmod(total_tracks, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def broadcast_ir_bank(ir_len, total_tracks, rank, world, device, dist=None, src=0):
    """Returns this rank's slice (tracks x ir_len, contiguous, on `device`) of the
    global conv1d_accel bank.  Rank `src` generates the whole bank."""
    lo, hi = shard_range(rank, world, total_tracks)
    if dist is None:                # no process group: one rank makes its own bank
        return torch.from_numpy(harness.conv_accel_ir(ir_len, hi - lo, lo, total_tracks)).to(device)
    bank = torch.empty(total_tracks * ir_len, dtype=torch.float32, device=device)
    if rank == src:
        bank.copy_(torch.from_numpy(harness.conv_accel_ir(ir_len, total_tracks)))
    dist.broadcast(bank, src=src)
    return bank.view(total_tracks, ir_len)[lo:hi].contiguous()


def shard_noise(total_tracks, bufsize, rank, world, seed=42):
    """The reference's noise is one flat track-major stream over ALL tracks
    (cuda/bench_utils.cu:238-245); a rank takes the rows of its tracks."""
    lo, hi = shard_range(rank, world, total_tracks)
    flat = harness.noise(total_tracks * bufsize, seed)
    return np.ascontiguousarray(flat.reshape(total_tracks, bufsize)[lo:hi])


def scatter_columns(global_out, shard_out, rank, world, total_tracks, bufsize):
    """Places a rank's sample-major result [s][local t] into the global
    sample-major buffer [s][T_total] (out[T*s + t] with the GLOBAL stride)."""
    lo, hi = shard_range(rank, world, total_tracks)
    global_out.reshape(bufsize, total_tracks)[:, lo:hi] = shard_out.reshape(bufsize, hi - lo)
    return global_out
