"""Channel sharding for the many-independent-channel benchmarks.

Tracks are independent, so a job of `total` tracks is cut into contiguous
shards, one per rank, and there is NO per-buffer collective.  The only exchange
is one-time: the Conv1D_accel impulse-response bank, whose formula needs the
GLOBAL track index and count, is generated once on rank 0 and broadcast (RCCL
over xGMI on GPUs; gloo in the CPU tests), then every rank keeps its slice.

Every benchmark with independent tracks shards the same way through the harness
(`shard_benchmark`): gain, GainStats, IIRFilter and FFT1D by rows; RndMemRead by
tracks with the 512 MiB pool on every rank; Conv1D with the ceil((L-1)/B)
preceding tracks' input rows as a halo (its golden convolves the FLAT input:
cuda/bench_conv1d.cu:188-208).  DWG, modal and FDTD3D reduce into shared
outputs: replicas only.
"""
import numpy as np
import torch

from . import harness


def shard_range(rank, world, total_tracks, granule=1):
    """Contiguous [lo, hi) of this rank, cut at multiples of `granule` tracks; the remainder goes to
    the low ranks.  (gab_shard_range_aligned; the C++ driver's gab::shardRange.)"""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    units = -(-total_tracks // granule)
    base, extra = divmod(units, world)
    lo = min(total_tracks, (rank * base + min(rank, extra)) * granule)
    return lo, min(total_tracks, lo + (base + (1 if rank < extra else 0)) * granule)


# tracks a kernel packs into one transform must stay in one shard for the bits to be the unsharded ones
GRANULE = {"FFT1D": 2, "Conv1D_accel": 4}


def shard_granule(name):
    return GRANULE.get(name, 1)


def bank_bytes_received(ir_len, total_tracks, rank, world, granule=1, distribution="broadcast", src=0):
    """Bytes of impulse-response bank that arrive at `rank` over the collective of broadcast_ir_bank: the whole bank
    under "broadcast", its own rows under "slices", nothing at `src` (which generates it) or without a group."""
    if world <= 1 or rank == src:
        return 0
    if distribution == "broadcast":
        return 4 * total_tracks * ir_len
    lo, hi = shard_range(rank, world, total_tracks, granule)
    return 4 * (hi - lo) * ir_len


def gather_per_rank(values, rank, world, dist=None, device="cpu"):
    """Every rank's list of floats, as a list of lists on every rank (all_gather over the job's group; a job without a
    group is one rank).  For rank 0's result line: a straggler rank or a slow bank distribution then names itself."""
    if dist is None or world <= 1:
        return [list(map(float, values))]
    t = torch.tensor(list(map(float, values)), dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [[float(v) for v in o.cpu()] for o in out]


def broadcast_ir_bank(ir_len, total_tracks, rank, world, device, dist=None, src=0, granule=1, distribution="broadcast"):
    """Returns this rank's slice (tracks x ir_len, contiguous, on `device`) of the
    global conv1d_accel bank.  Rank `src` generates the whole bank.

    granule: the SAME cut as the shard's data (shard_granule(name): 4 for Conv1D_accel) — ranges cut with different
    granules disagree whenever total_tracks is no multiple of world * granule, and slices would land on the wrong rows.
    distribution: "broadcast" — every rank receives the whole bank and keeps its rows (north_star's RCCL broadcast;
    total_tracks x ir_len x 4 bytes per rank: 128 MiB at C5) — or "slices": `src` sends every rank ITS rows only
    (dist.scatter: send/recv pairs, one xGMI link per peer; 16 MiB per rank at C5, and no rank but `src` ever holds
    the whole bank).  Same slices either way (test_sharding_gloo.py)."""
    lo, hi = shard_range(rank, world, total_tracks, granule)
    if dist is None:                # no process group: one rank makes its own bank
        return torch.from_numpy(harness.conv_accel_ir(ir_len, hi - lo, lo, total_tracks)).to(device)
    if distribution == "slices":
        final_device = device
        if dist.get_backend() == "gloo":          # gloo scatters and sends host tensors only (CPU tests, the one-GPU rehearsal)
            device = torch.device("cpu")
        mine = torch.empty((hi - lo) * ir_len, dtype=torch.float32, device=device)
        parts = None
        if rank == src:
            bank = torch.from_numpy(harness.conv_accel_ir(ir_len, total_tracks)).to(device).view(total_tracks, ir_len)
            parts = []
            for r in range(world):
                a, b = shard_range(r, world, total_tracks, granule)
                parts.append(bank[a:b].reshape(-1).contiguous())
        if all(shard_range(r, world, total_tracks, granule)[1] - shard_range(r, world, total_tracks, granule)[0] == hi - lo
               for r in range(world)):
            dist.scatter(mine, parts, src=src)
        else:                        # uneven shards: scatter wants equal sizes — point-to-point instead
            if rank == src:
                reqs = [dist.isend(parts[r], dst=r) for r in range(world) if r != src]
                mine.copy_(parts[src])
                for q in reqs:
                    q.wait()
            else:
                dist.recv(mine, src=src)
        return mine.view(hi - lo, ir_len).to(final_device)
    if distribution != "broadcast":
        raise ValueError("distribution must be 'broadcast' or 'slices'")
    bank = torch.empty(total_tracks * ir_len, dtype=torch.float32, device=device)
    if rank == src:
        bank.copy_(torch.from_numpy(harness.conv_accel_ir(ir_len, total_tracks)))
    dist.broadcast(bank, src=src)
    return bank.view(total_tracks, ir_len)[lo:hi].contiguous()


def shard_noise(total_tracks, bufsize, rank, world, seed=42, granule=1):
    """The reference's noise is one flat track-major stream over ALL tracks
    (cuda/bench_utils.cu:238-245); a rank takes the rows of its tracks (granule: as broadcast_ir_bank)."""
    lo, hi = shard_range(rank, world, total_tracks, granule)
    flat = harness.noise(total_tracks * bufsize, seed)
    return np.ascontiguousarray(flat.reshape(total_tracks, bufsize)[lo:hi])


def scatter_columns(global_out, shard_out, rank, world, total_tracks, bufsize, granule=1):
    """Places a rank's sample-major result [s][local t] into the global
    sample-major buffer [s][T_total] (out[T*s + t] with the GLOBAL stride; granule: as broadcast_ir_bank)."""
    lo, hi = shard_range(rank, world, total_tracks, granule)
    global_out.reshape(bufsize, total_tracks)[:, lo:hi] = shard_out.reshape(bufsize, hi - lo)
    return global_out


# registry names whose tracks are independent (gab_bench_set_shard accepts them)
SHARDABLE = ("gain", "GainStats", "IIRFilter", "FFT1D", "RndMemRead", "Conv1D", "Conv1D_accel")


def conv1d_halo_tracks(first_track, ir_len, bufsize):
    """Input rows a Conv1D shard needs in front of its first track: the golden's flat indexing lets a
    track's first L-1 outputs read the end of the preceding track(s)."""
    return min(first_track, (ir_len - 1 + bufsize - 1) // bufsize)


def shard_benchmark(name, rank, world, total_tracks, **cfg):
    """The harness benchmark `name` for this rank's tracks of a total_tracks job (not yet set up)."""
    if name not in SHARDABLE:
        raise ValueError("%s has no independent tracks: replicas only" % name)
    lo, hi = shard_range(rank, world, total_tracks, shard_granule(name))
    b = harness.Benchmark(name, n_tracks=hi - lo, **cfg)
    b.set_shard(lo, total_tracks)
    return b


def join_results(parts):
    """Shard result dicts (Benchmark.results()) in rank order -> the whole job's arrays: track-major arrays
    concatenate by rows, sample-major ones by columns."""
    out = {}
    for key in parts[0]:
        layout, per = parts[0][key][1], parts[0][key][2]
        if layout == 0:
            out[key] = np.concatenate([p[key][0] for p in parts])
        else:
            out[key] = np.concatenate([p[key][0].reshape(per, -1) for p in parts], axis=1).ravel()
    return out
