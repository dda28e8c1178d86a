// h_multi.cpp — one node, several GPUs: one host thread per device, contiguous channel shards,
// the impulse-response bank broadcast once with RCCL's C API (see include/gab/multi_gpu.hpp).
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <mutex>
#include <stdexcept>
#include <thread>

#include "gab/benchmarks.hpp"
#include "gab/multi_gpu.hpp"
#include "gab_common.hpp"

namespace gab {

ShardRange shardRange(int rank, int world, size_t total_tracks, size_t granule) {
    if (world < 1) throw std::invalid_argument("shardRange: world must be >= 1");
    if (rank < 0 || rank >= world) throw std::invalid_argument("shardRange: rank outside the world");
    if (granule < 1) throw std::invalid_argument("shardRange: granule must be >= 1");
    // whole granules (the tracks a kernel packs into one transform must stay together), remainder to the low ranks
    const size_t units = (total_tracks + granule - 1) / granule;
    const size_t base = units / world, extra = units % world, r = static_cast<size_t>(rank);
    ShardRange s;
    s.lo = std::min(total_tracks, (r * base + std::min(r, extra)) * granule);
    s.hi = std::min(total_tracks, s.lo + (base + (r < extra ? 1 : 0)) * granule);
    return s;
}

size_t shardGranule(const std::string& benchmark) {
    if (benchmark == "FFT1D") return 2;             // two tracks share one complex transform
    if (benchmark == "Conv1D_accel") return 4;      // a duo of channel pairs shares a workgroup's transforms
    return 1;
}

bool MultiGpuReport::ok() const {
    if (ranks.empty()) return false;
    for (const auto& r : ranks)
        if (!r.error.empty() || r.validation.status != ValidationOutcome::SUCCESS) return false;
    return true;
}

namespace {

// ---- the five RCCL entry points this file needs, resolved at run time ------------------------
// (signatures from rccl.h; ncclComm_t is an opaque pointer, ncclFloat = 7, ncclSuccess = 0)
struct Rccl {
    using Comm = void*;
    int (*CommInitAll)(Comm*, int, const int*) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;            // non-empty: could not be loaded

    static const Rccl& get() {
        static const Rccl r = load();
        return r;
    }

private:
    static Rccl load() {
        Rccl r;
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) {
            r.why = std::string("librccl.so.1 could not be loaded: ") + dlerror();
            return r;
        }
        auto sym = [&](const char* n) {
            void* p = dlsym(h, n);
            if (!p && r.why.empty()) r.why = std::string("librccl lacks ") + n;
            return p;
        };
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        return r;
    }
};

void rcclCheck(const Rccl& n, int rc, const char* what) {
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + (n.GetErrorString ? n.GetErrorString(rc) : "RCCL error"));
}

constexpr int kNcclFloat = 7;

// threads meet here between setup and the timed loop, so the ranks run their iterations concurrently
class Rendezvous {
public:
    explicit Rendezvous(int n) : want_(n) {}
    void arrive() {
        std::unique_lock<std::mutex> lock(mu_);
        const int gen = generation_;
        if (++count_ == want_) {
            count_ = 0;
            ++generation_;
            cv_.notify_all();
        } else {
            cv_.wait(lock, [&] { return gen != generation_; });
        }
    }

private:
    std::mutex mu_;
    std::condition_variable cv_;
    int want_, count_ = 0, generation_ = 0;
};

// The bank of `total` tracks on every device: generated once on the host, uploaded to device 0,
// broadcast to the others.  Returns the per-device base pointers (caller frees).
struct SharedBank {
    std::vector<float*> d_bank;
    std::vector<hipStream_t> streams;
    std::vector<Rccl::Comm> comms;
    double broadcast_ms = -1.0;
    size_t bytes = 0;

    void release() {
        const Rccl& n = Rccl::get();
        for (size_t r = 0; r < d_bank.size(); ++r) {
            (void)hipSetDevice(static_cast<int>(r));
            if (r < streams.size() && streams[r]) (void)hipStreamDestroy(streams[r]);
            if (d_bank[r]) (void)hipFree(d_bank[r]);
        }
        for (auto c : comms)
            if (c && n.CommDestroy) (void)n.CommDestroy(c);
        d_bank.clear();
        streams.clear();
        comms.clear();
    }
};

SharedBank broadcastConvAccelBank(int gpus, int ir_len, size_t total_tracks) {
    const Rccl& n = Rccl::get();
    if (!n.why.empty()) throw std::runtime_error("multi-GPU run needs RCCL: " + n.why);
    SharedBank b;
    const size_t count = total_tracks * static_cast<size_t>(ir_len);
    b.bytes = count * sizeof(float);
    b.d_bank.assign(gpus, nullptr);
    b.streams.assign(gpus, nullptr);
    b.comms.assign(gpus, nullptr);
    try {
        std::vector<int> devs(gpus);
        for (int r = 0; r < gpus; ++r) devs[r] = r;
        rcclCheck(n, n.CommInitAll(b.comms.data(), gpus, devs.data()), "ncclCommInitAll");
        for (int r = 0; r < gpus; ++r) {
            HIP_CHECK(hipSetDevice(r));
            HIP_CHECK(hipMalloc(&b.d_bank[r], b.bytes));
            HIP_CHECK(hipStreamCreateWithFlags(&b.streams[r], hipStreamNonBlocking));
        }
        {
            std::vector<float> host(count);
            BenchmarkUtils::generateConvAccelImpulseResponses(host.data(), ir_len, 0, total_tracks, total_tracks);
            HIP_CHECK(hipSetDevice(0));
            HIP_CHECK(hipMemcpy(b.d_bank[0], host.data(), b.bytes, hipMemcpyHostToDevice));
        }
        const auto t0 = std::chrono::steady_clock::now();
        rcclCheck(n, n.GroupStart(), "ncclGroupStart");
        for (int r = 0; r < gpus; ++r)
            rcclCheck(n, n.Broadcast(b.d_bank[r], b.d_bank[r], count, kNcclFloat, 0, b.comms[r], b.streams[r]),
                      "ncclBroadcast");
        rcclCheck(n, n.GroupEnd(), "ncclGroupEnd");
        for (int r = 0; r < gpus; ++r) {
            HIP_CHECK(hipSetDevice(r));
            HIP_CHECK(hipStreamSynchronize(b.streams[r]));
        }
        b.broadcast_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    } catch (...) {
        b.release();
        throw;
    }
    return b;
}

}  // namespace

MultiGpuReport runOnDevices(const MultiGpuConfig& cfg) {
    if (cfg.gpus < 1) throw std::invalid_argument("runOnDevices: gpus must be >= 1");
    int present = 0;
    HIP_CHECK(hipGetDeviceCount(&present));
    if (present < cfg.gpus)
        throw std::runtime_error("--gpus " + std::to_string(cfg.gpus) + " asked for, " + std::to_string(present) +
                                 " HIP device(s) present");
    const auto& names = benchmarkNames();
    if (std::find(names.begin(), names.end(), cfg.benchmark) == names.end())
        throw std::invalid_argument("Unknown benchmark: " + cfg.benchmark);

    MultiGpuReport rep;
    rep.gpus = cfg.gpus;
    rep.sharded = benchmarkShards(cfg.benchmark);
    const bool bank = cfg.benchmark == "Conv1D_accel";          // the one benchmark with a one-time exchange
    rep.total_tracks = rep.sharded ? static_cast<size_t>(NTRACKS) : static_cast<size_t>(NTRACKS) * cfg.gpus;
    rep.collective = "none";
    rep.ranks.resize(cfg.gpus);
    if (rep.sharded && shardRange(cfg.gpus - 1, cfg.gpus, rep.total_tracks, shardGranule(cfg.benchmark)).count() == 0)
        throw std::invalid_argument("more GPUs than tracks");

    const int ir_len = IR_LENGTH > 0 ? IR_LENGTH : Conv1DAccelBenchmark::DEFAULT_IR_LEN;
    SharedBank shared;
    if (bank) {
        shared = broadcastConvAccelBank(cfg.gpus, ir_len, rep.total_tracks);
        rep.ir_bank_bytes = shared.bytes;
        rep.ir_broadcast_ms = shared.broadcast_ms;
        rep.collective = "rccl ncclCommInitAll + ncclBroadcast (once, before the first buffer)";
    }
    if (rep.sharded) {
        if (cfg.benchmark == "Conv1D")
            rep.partition = "contiguous channel shards + the ceil((L-1)/B) preceding tracks' input rows as a halo from the host, no collective";
        else if (cfg.benchmark == "RndMemRead")
            rep.partition = "contiguous channel shards, the 512 MiB pool on every rank (generated once, uploaded by each), no collective";
        else if (bank)
            rep.partition = "contiguous channel shards, impulse-response bank broadcast once, no per-buffer collective";
        else
            rep.partition = "contiguous channel shards, no collective";
    } else {
        rep.partition = "replicas only";
    }

    const bool quiet_before = GAB_QUIET;
    if (cfg.gpus > 1) GAB_QUIET = true;           // the ranks' progress chatter would interleave
    Rendezvous meet(cfg.gpus);
    std::mutex setup_mu;
    auto rank_body = [&](int r) {
        RankReport& out = rep.ranks[r];
        out.device = r;
        bool arrived = false;
        try {
            HIP_CHECK(hipSetDevice(r));
            std::unique_ptr<GPUABenchmark> b;
            if (rep.sharded) {
                out.tracks = shardRange(r, cfg.gpus, rep.total_tracks, shardGranule(cfg.benchmark));
                b = createBenchmarkShard(cfg.benchmark, out.tracks.count());
                b->setShard(out.tracks.lo, rep.total_tracks);
                if (bank)
                    static_cast<Conv1DAccelBenchmark*>(b.get())->shareImpulseResponses(
                        shared.d_bank[r] + out.tracks.lo * static_cast<size_t>(ir_len));
            } else {
                out.tracks.lo = 0;
                out.tracks.hi = static_cast<size_t>(NTRACKS);
                b = createBenchmark(cfg.benchmark);
            }
            {
                // one rank at a time: the replicas' generators draw from the process-wide rand() (DWG, modal, FDTD3D,
                // datacopy), whose stream concurrent threads would interleave
                std::lock_guard<std::mutex> lock(setup_mu);
                b->setupBenchmark();
            }
            out.algorithmic_bytes = b->algorithmicBytes();
            meet.arrive();
            arrived = true;
            if (!cfg.validate_only) out.result = b->runBenchmark(cfg.iterations, cfg.warmup);
            b->validate(out.validation);
        } catch (const std::exception& e) {
            out.error = e.what();
            out.validation.status = ValidationOutcome::FATAL;
            if (!arrived) meet.arrive();          // the others must not wait for a rank that failed in setup
        }
    };
    std::vector<std::thread> pool;
    for (int r = 1; r < cfg.gpus; ++r) pool.emplace_back(rank_body, r);
    rank_body(0);
    for (auto& t : pool) t.join();
    GAB_QUIET = quiet_before;
    if (bank) shared.release();
    (void)hipSetDevice(0);

    for (const auto& r : rep.ranks) {
        if (!r.result.latencies.empty()) {
            rep.job_median_ms = std::max(rep.job_median_ms, static_cast<double>(r.result.statistics.median));
            rep.job_device_median_ms = std::max(rep.job_device_median_ms, static_cast<double>(r.result.gpu_statistics.median));
        }
    }
    if (rep.job_median_ms > 0.0) rep.tracks_per_second = rep.total_tracks / (rep.job_median_ms * 1e-3);
    return rep;
}

std::string multiGpuJson(const MultiGpuReport& r) {
    char buf[512];
    std::string j = "  \"multi_gpu\": {\n";
    snprintf(buf, sizeof buf, "    \"gpus\": %d,\n    \"partition\": \"%s\",\n    \"total_tracks\": %zu,\n", r.gpus,
             r.partition.c_str(), r.total_tracks);
    j += buf;
    j += "    \"collective\": \"" + r.collective + "\",\n";
    snprintf(buf, sizeof buf, "    \"ir_bank_bytes\": %zu,\n    \"ir_broadcast_ms\": %s,\n", r.ir_bank_bytes,
             r.ir_broadcast_ms >= 0 ? std::to_string(r.ir_broadcast_ms).c_str() : "null");
    j += buf;
    snprintf(buf, sizeof buf, "    \"job_median_ms\": %.6f,\n    \"job_device_median_ms\": %.6f,\n    \"tracks_per_second\": %.1f,\n",
             r.job_median_ms, r.job_device_median_ms, r.tracks_per_second);
    j += buf;
    j += "    \"ranks\": [\n";
    for (size_t i = 0; i < r.ranks.size(); ++i) {
        const RankReport& k = r.ranks[i];
        snprintf(buf, sizeof buf,
                 "      {\"device\": %d, \"first_track\": %zu, \"tracks\": %zu, \"median_ms\": %.6f, \"device_median_ms\": %.6f, "
                 "\"algorithmic_bytes\": %zu, \"valid\": %s, \"max_error\": %.3g%s}%s\n",
                 k.device, k.tracks.lo, k.tracks.count(), k.result.latencies.empty() ? 0.0 : k.result.statistics.median,
                 k.result.gpu_latencies.empty() ? 0.0 : k.result.gpu_statistics.median, k.algorithmic_bytes,
                 k.validation.status == ValidationOutcome::SUCCESS ? "true" : "false", k.validation.max_error,
                 k.error.empty() ? "" : ", \"error\": \"rank failed, see stdout\"", i + 1 < r.ranks.size() ? "," : "");
        j += buf;
    }
    j += "    ]\n  }";
    return j;
}

}  // namespace gab

extern "C" int gab_shard_range_aligned(int rank, int world, size_t total_tracks, size_t granule, size_t* lo, size_t* hi) {
    if (!lo || !hi) return gab::bad_arg("gab_shard_range: null pointer");
    if (world < 1 || rank < 0 || rank >= world || granule < 1) return gab::bad_arg("gab_shard_range: rank outside the world");
    const gab::ShardRange s = gab::shardRange(rank, world, total_tracks, granule);
    *lo = s.lo;
    *hi = s.hi;
    return GAB_OK;
}

extern "C" int gab_shard_range(int rank, int world, size_t total_tracks, size_t* lo, size_t* hi) {
    return gab_shard_range_aligned(rank, world, total_tracks, 1, lo, hi);
}

extern "C" size_t gab_shard_granule(const char* benchmark) { return benchmark ? gab::shardGranule(benchmark) : 1; }
