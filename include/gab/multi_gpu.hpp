// multi_gpu.hpp — one node, several GPUs, behind the reference driver (additive: the reference runs
// one device; BASELINE configs[4] shards 8192 channels of Conv1D_accel over 8).
//
// One host thread per device.  Tracks are independent, so gain, GainStats, IIRFilter, FFT1D, RndMemRead,
// Conv1D and Conv1D_accel are cut into contiguous channel shards (the remainder goes to the low ranks)
// and there is NO per-buffer collective: every rank takes its rows of the job's one input stream; Conv1D
// also takes the few preceding tracks' rows its flat-index history reaches; RndMemRead keeps the whole
// pool on every rank.  The one exchange is one-time and Conv1D_accel's: the impulse-response bank —
// its formula needs the GLOBAL track index and count (cuda/bench_conv1d_accel.cu:152-173) — is
// generated once, uploaded to device 0 and
// broadcast with RCCL's C API (ncclCommInitAll + ncclBroadcast inside one group, xGMI between the
// devices); every rank then transforms its own rows.  Benchmarks that reduce into shared outputs or
// have no channel structure run as N independent replicas ("replicas only").
//
// RCCL is resolved at run time (dlopen of librccl.so.1) the first time a multi-device run asks for
// it: the library has no link-time dependency on it, and a process that already carries another
// copy of RCCL (PyTorch bundles one) does not get two sets of symbols.
#pragma once

#include <cstddef>
#include <string>
#include <vector>

#include "bench_base.hpp"

namespace gab {

struct ShardRange {
    size_t lo = 0, hi = 0;
    size_t count() const { return hi - lo; }
};
// Contiguous [lo, hi) of `rank` out of `world`; the remainder goes to the low ranks.  Throws
// std::invalid_argument for a rank outside the world or a world < 1.
// `granule`: tracks that must stay together — FFT1D transforms two tracks in one complex transform and Conv1D_accel
// four channels in one workgroup's, so their bits depend on which tracks share one: shards of those are cut at
// multiples of shardGranule() and then reproduce the unsharded results bit for bit.
ShardRange shardRange(int rank, int world, size_t total_tracks, size_t granule = 1);
size_t shardGranule(const std::string& benchmark);

struct MultiGpuConfig {
    std::string benchmark;       // registry name
    int gpus = 1;
    int iterations = 100;
    int warmup = 3;
    bool validate_only = false;  // setup + validate, no timed loop
};

struct RankReport {
    int device = 0;
    ShardRange tracks;           // the rank's channels (sharded) or [0, NTRACKS) (replica)
    RunResult result;            // empty latencies when validate_only
    ValidationReport validation;
    size_t algorithmic_bytes = 0;
    std::string error;           // non-empty: the rank failed with this message
};

struct MultiGpuReport {
    int gpus = 0;
    bool sharded = false;            // channel shards (every benchmark with independent tracks) vs replicas
    std::string partition;           // which form ran, in words (the "partition" member of --json)
    size_t total_tracks = 0;
    size_t ir_bank_bytes = 0;        // what the one-time broadcast moved (0: none)
    double ir_broadcast_ms = -1.0;   // ncclBroadcast of the bank incl. stream sync; < 0: no broadcast
    std::string collective;          // "rccl ncclBroadcast" | "none"
    std::vector<RankReport> ranks;
    // whole job: every rank processes one buffer of its tracks per iteration, concurrently
    double job_median_ms = 0.0;      // slowest rank's median iteration
    double job_device_median_ms = 0.0;
    double tracks_per_second = 0.0;  // total tracks (buffers of one track) per second at job_median_ms
    bool ok() const;
};

// Runs `cfg.benchmark` on devices 0 .. cfg.gpus-1 with the process globals (NTRACKS = the job's
// total tracks for a sharded benchmark, the per-replica count otherwise).  Throws std::runtime_error
// when fewer devices are present or RCCL cannot be loaded for a sharded run on more than... any
// device count (a one-device run still goes through ncclCommInitAll + ncclBroadcast, so the path
// is exercised on a one-GPU box).
MultiGpuReport runOnDevices(const MultiGpuConfig& cfg);

// "multi_gpu": {...} as a JSON member (no trailing comma), for the driver's --json output.
std::string multiGpuJson(const MultiGpuReport& r);

}  // namespace gab
