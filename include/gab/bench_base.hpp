// bench_base.hpp — GPUABenchmark, the plugin interface of the benchmark suite.
//
// This is the drop-in boundary: the class keeps the public and protected
// surface of the reference's cuda/bench_base.cuh:18-139 (same type, method and
// member names, same defaults, same exception behaviour), so a main.cu-style
// driver and benchmark subclasses written against the reference compile against
// it.  Differences are additive:
//   * every benchmark owns a HIP stream; host<->device copies are
//     hipMemcpyAsync on pinned memory (the reference uses synchronous
//     cudaMemcpy on the default stream);
//   * resetState() + runValidationIteration(): state-carrying benchmarks (IIR,
//     DWG, FDTD3D, RndMem, streaming convolution) are validated on "reset ->
//     one iteration -> compare", which is the iteration their CPU golden
//     describes (the reference compares its LAST timed
//     iteration against a golden of the FIRST);
//   * algorithmicBytes(): the roofline numerator for the benchmark's kernel.
#pragma once

#include <cstdio>
#include <functional>
#include <iostream>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "bench_utils.hpp"
#include "globals.hpp"

namespace gab {

// What a run returns (bench_base.cuh:20-34) and what validate() fills in (:36-47).  Defined at
// namespace scope and re-exported from the class under the reference's nested names.
struct RunResult {
    std::vector<float> latencies;          // wall ms per iteration (copies included)
    std::vector<float> gpu_latencies;      // device ms per iteration, when recorded
    BenchmarkUtils::Statistics statistics;
    BenchmarkUtils::Statistics gpu_statistics;
    std::string benchmark_name;
    size_t buffer_size;
    size_t track_count;
    int iterations;
    double throughput_gbps;                // bytes_processed / mean latency, GiB/s
    double samples_per_sec;
    size_t bytes_processed;
    float mean_latency_ms;
    // DAW pacing of this run (additive; zero when no simulator was set)
    unsigned long long daw_waits = 0;
    unsigned long long daw_missed_slots = 0;
};

enum class ValidationOutcome { SUCCESS = 0, FAILURE = 1, FATAL = -1 };

struct ValidationReport {
    ValidationOutcome status = ValidationOutcome::SUCCESS;
    std::vector<std::string> messages;
    float max_error = 0.0f;
    float mean_error = 0.0f;
};

// Two pinned host buffers and two device buffers of element_count floats (bench_base.cuh:50-74).
struct IoBuffers {
    float* h_input = nullptr;
    float* h_output = nullptr;
    float* d_input = nullptr;
    float* d_output = nullptr;
    size_t element_count = 0;
    size_t size_bytes = 0;

    ~IoBuffers() { cleanup(); }
    void cleanup();
};

}  // namespace gab

// The four stages every benchmark supplies (bench_base.cuh:94-97), as an interface of their own so
// that tooling can drive a benchmark without knowing the harness around it.
struct BenchmarkStages {
    virtual ~BenchmarkStages() = default;
    virtual void setupBenchmark() = 0;                                // allocate, generate inputs, goldens
    virtual void runKernel() = 0;                                     // device work only
    virtual void performBenchmarkIteration() = 0;                     // H2D + device work + D2H
    virtual void validate(gab::ValidationReport& validation_data) = 0;
};

class GPUABenchmark : public BenchmarkStages {
public:
    // the reference's nested names
    using BenchmarkResult = gab::RunResult;
    using ValidationStatus = gab::ValidationOutcome;
    using ValidationData = gab::ValidationReport;

    GPUABenchmark(const std::string& name, size_t buffer_size = BUFSIZE, size_t track_count = NTRACKS);
    ~GPUABenchmark() override;
    GPUABenchmark(const GPUABenchmark&) = delete;
    GPUABenchmark& operator=(const GPUABenchmark&) = delete;

    // ---- the harness (bench_base.cuh:103-110) ---------------------------------------------------
    BenchmarkResult runBenchmark(int iterations = NRUNS, int warmupIterations = 3);        // whole iterations
    BenchmarkResult runKernelBenchmark(int iterations = NRUNS, int warmupIterations = 3);  // runKernel() only
    void allocateBuffers(size_t element_count);
    void generateTestData(unsigned int seed = 42);
    void transferToDevice();
    void transferToHost();
    void printResults(const BenchmarkResult& result);
    void writeResults(const BenchmarkResult& result, const std::string& filename = "");

    // ---- additive hooks -------------------------------------------------------------------------
    virtual void resetState() {}
    virtual void runValidationIteration() { resetState(); performBenchmarkIteration(); }
    virtual size_t algorithmicBytes() const { return 2 * getTotalElements() * sizeof(float); }
    // true when the kernel keeps its working set on chip for the whole iteration, so that the HBM roofline
    // the algorithmic bytes are priced against does not bound it (FDTD3D rooms that fit the LDS)
    virtual bool workingSetOnChip() const { return false; }
    // The CPU golden of tracks [first, first + count) recomputed into the benchmark's own reference
    // buffer (what setupBenchmark already did once); false when the golden cannot be cut by track.
    // timeCpuGolden() runs it over `threads` host threads (1 when it cannot be cut) and returns the
    // wall time in ms, or a negative value when the benchmark offers no golden to time.
    virtual bool cpuGoldenSlice(size_t first_track, size_t count) { (void)first_track; (void)count; return false; }
    virtual bool cpuGoldenWhole() { return false; }
    double timeCpuGolden(int threads, int* threads_used = nullptr);

    // Channel shard (additive): this instance computes tracks [first_track, first_track + getTrackCount())
    // of a job of total_tracks.  Inputs, impulse responses, playheads and goldens are then the GLOBAL job's rows — the
    // one flat noise stream, the bank formulas with the global track index — so that the shards' results, put side
    // by side, are the unsharded results bit for bit.  Call before setupBenchmark().  Benchmarks whose tracks are not
    // independent (DWG, modal, FDTD3D reduce into shared outputs) throw std::invalid_argument from setupBenchmark().
    void setShard(size_t first_track, size_t total_tracks);
    bool isShard() const { return shard_total_ != 0; }
    size_t shardFirstTrack() const { return shard_first_; }
    size_t jobTracks() const { return shard_total_ ? shard_total_ : track_count_; }
    virtual bool shardable() const { return false; }
    // What an iteration leaves on the host, by name: `layout` 0 = track-major rows of per_track values, 1 = sample-major
    // (value [per_track index][track]).  Shards' arrays concatenate by rows / by columns into the unsharded arrays.
    struct ResultArray { const char* name; const float* data; size_t count; int layout; size_t per_track; };
    virtual std::vector<ResultArray> resultArrays() const;

    // DAW-style pacing (metal-swift Core/GPUABenchmark.swift:90,358-392): when set, every
    // warm-up and timed iteration is followed by a wait for the next buffer slot.
    void setDawSimulator(const BenchmarkUtils::DAWSimulator& sim) { daw_simulator_ = sim; daw_enabled_ = true; }
    void clearDawSimulator() { daw_enabled_ = false; }
    bool hasDawSimulator() const { return daw_enabled_; }
    const BenchmarkUtils::DAWSimulator& dawSimulator() const { return daw_simulator_; }
    // Keep-warm (additive; gab_keep_warm in gab_c_api.h): for the length of a run, a resident launch of eight sleeping
    // waves is kicked after every iteration, so the device does not go idle while the loop waits for the next slot
    // (at C3 a paced round trip answers ~9 us sooner: profiles/r05_paced_keep_warm.txt).  Off by default.
    void setKeepWarm(bool on) { keep_warm_enabled_ = on; }
    bool keepWarm() const { return keep_warm_enabled_; }

    // ---- identity and shape (bench_base.cuh:116-119) --------------------------------------------
    const std::string& getName() const { return benchmark_name_; }
    size_t getBufferSize() const { return buffer_size_; }
    size_t getTrackCount() const { return track_count_; }
    size_t getTotalElements() const { return buffer_size_ * track_count_; }
    hipStream_t getStream() const { return stream_; }
    const float* hostInput() const { return buffers.h_input; }      // read-only views for tests and bindings
    const float* hostOutput() const { return buffers.h_output; }

protected:
    using BufferSet = gab::IoBuffers;

    // ---- helpers for subclasses (bench_base.cuh:126-138) ----------------------------------------
    BenchmarkResult runWithIteration(int iterations, int warmupIterations, const std::function<void()>& iterationBody);
    ValidationData compareWithReference(const float* cpu_reference, float tolerance = 1e-5f);
    BenchmarkUtils::BenchmarkParams makeBenchmarkParams(float gainValue = 0.0f) const;
    std::pair<int, int> calculateGridDimensions(int desired_threads_per_block = 256) const;
    void synchronizeAndCheck();
    void resetGpuIterationMetrics();
    void recordGpuDuration(float milliseconds);
    float* getHostInput() { return buffers.h_input; }
    float* getHostOutput() { return buffers.h_output; }
    float* getDeviceInput() { return buffers.d_input; }
    float* getDeviceOutput() { return buffers.d_output; }
    // compare any pair of host arrays with the same bookkeeping as compareWithReference
    static ValidationData compareArrays(const float* got, const float* expected, size_t n, float tolerance);
    // throws std::runtime_error carrying gab_last_error() when a C-ABI call fails
    static void checkGab(int rc, const char* what);

    // ---- state ----------------------------------------------------------------------------------
    BufferSet buffers;
    BenchmarkUtils::BenchmarkTimer timer;
    std::string benchmark_name_;
    size_t buffer_size_;
    size_t track_count_;
    size_t shard_first_ = 0, shard_total_ = 0;      // shard_total_ == 0: the whole job
    float current_iteration_gpu_ms_ = 0.0f;
    hipStream_t stream_ = nullptr;             // all of this benchmark's device work
    BenchmarkUtils::DAWSimulator daw_simulator_;
    bool daw_enabled_ = false;
    bool keep_warm_enabled_ = false;
};
