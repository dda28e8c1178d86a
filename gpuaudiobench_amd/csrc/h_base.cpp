// h_base.cpp — GPUABenchmark: buffers, the warm-up + timed loop, statistics,
// validation bookkeeping.  Behavioural reference: cuda/bench_base.cu.
#include <algorithm>
#include <cmath>
#include <iomanip>
#include <thread>

#include "gab/bench_base.hpp"
#include "gab_c_api.h"

void gab::IoBuffers::cleanup() {
    // drain the device before memory it may still be using goes away
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess)
        fprintf(stderr, "Warning: hipDeviceSynchronize before cleanup failed: %s\n", hipGetErrorString(e));
    BenchmarkUtils::freeHostBuffers({h_input, h_output});
    BenchmarkUtils::freeDeviceBuffers({d_input, d_output});
    h_input = h_output = nullptr;
    d_input = d_output = nullptr;
}

GPUABenchmark::GPUABenchmark(const std::string& name, size_t buffer_size, size_t track_count)
    : benchmark_name_(name), buffer_size_(buffer_size), track_count_(track_count) {
    HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
}

GPUABenchmark::~GPUABenchmark() {
    if (stream_) {
        (void)hipStreamSynchronize(stream_);
        (void)hipStreamDestroy(stream_);
        stream_ = nullptr;
    }
}

void GPUABenchmark::allocateBuffers(size_t element_count) {
    if (element_count == 0) throw std::invalid_argument("allocateBuffers requires element_count > 0");
    buffers.element_count = element_count;
    buffers.size_bytes = element_count * sizeof(float);
    buffers.h_input = BenchmarkUtils::allocateHostBuffer<float>(element_count, benchmark_name_ + " host input buffer");
    buffers.h_output = BenchmarkUtils::allocateHostBuffer<float>(element_count, benchmark_name_ + " host output buffer");
    buffers.d_input = BenchmarkUtils::allocateDeviceBuffer<float>(element_count, benchmark_name_ + " device input buffer");
    buffers.d_output = BenchmarkUtils::allocateDeviceBuffer<float>(element_count, benchmark_name_ + " device output buffer");
}

// Asynchronous on the benchmark's stream: the kernel that follows is ordered
// behind the copy by the stream, not by a host-side wait.
void GPUABenchmark::transferToDevice() {
    if (!buffers.d_input || !buffers.h_input)
        throw std::runtime_error("transferToDevice called before input buffers were allocated");
    HIP_CHECK(hipMemcpyAsync(buffers.d_input, buffers.h_input, buffers.size_bytes, hipMemcpyHostToDevice, stream_));
}

// The host reads the result next, so this one does wait.
void GPUABenchmark::transferToHost() {
    if (!buffers.d_output || !buffers.h_output)
        throw std::runtime_error("transferToHost called before output buffers were allocated");
    HIP_CHECK(hipMemcpyAsync(buffers.h_output, buffers.d_output, buffers.size_bytes, hipMemcpyDeviceToHost, stream_));
    HIP_CHECK(hipStreamSynchronize(stream_));
}

void GPUABenchmark::generateTestData(unsigned int seed) {
    if (!buffers.h_input) throw std::runtime_error("generateTestData called before host input buffer allocation");
    if (isShard())     // the rows of this shard's tracks out of the one flat stream over all tracks
        BenchmarkUtils::generateRandomAudioDataFrom(buffers.h_input, buffers.element_count, seed,
                                                    static_cast<unsigned long long>(shard_first_) * buffer_size_);
    else
        BenchmarkUtils::generateRandomAudioData(buffers.h_input, buffers.element_count, seed);
}

void GPUABenchmark::setShard(size_t first_track, size_t total_tracks) {
    if (total_tracks == 0 || first_track + track_count_ > total_tracks)
        throw std::invalid_argument("setShard: tracks [first, first + count) must lie inside the job's total_tracks");
    if (!shardable())
        throw std::invalid_argument("setShard: " + benchmark_name_ + " has no independent tracks to shard (replicas only)");
    if (buffers.h_input) throw std::invalid_argument("setShard must be called before setupBenchmark()");
    shard_first_ = first_track;
    shard_total_ = total_tracks;
}

std::vector<GPUABenchmark::ResultArray> GPUABenchmark::resultArrays() const {
    return {{"output", buffers.h_output, buffers.element_count, 0, buffer_size_}};
}

GPUABenchmark::BenchmarkResult GPUABenchmark::runKernelBenchmark(int iterations, int warmupIterations) {
    return runWithIteration(iterations, warmupIterations, [this]() { this->runKernel(); });
}

GPUABenchmark::BenchmarkResult GPUABenchmark::runBenchmark(int iterations, int warmupIterations) {
    return runWithIteration(iterations, warmupIterations, [this]() { this->performBenchmarkIteration(); });
}

// The measurement loop behind runBenchmark / runKernelBenchmark.  What it must produce is the
// reference's contract (cuda/bench_base.cu:59-118): warm-ups whose failures are reported and
// skipped, one wall-clock latency per timed iteration, the device time a body reported through
// recordGpuDuration, statistics over both, and throughput priced on tracks x buffer x 4 bytes per
// mean latency.  How it gets there is this harness's own: three small stages sharing a pacing clock.
namespace {

struct PacedLoop {
    GPUABenchmark* bench;
    bool paced;
    BenchmarkUtils::DAWSimulator* daw;
    BenchmarkUtils::DAWSimulationState clock;
    gab_keep_warm* warm = nullptr;              // setKeepWarm: there for the length of the run, kicked after every pass
    void slot() {
        if (warm) (void)gab_keep_warm_kick(warm);
        if (paced) daw->wait(clock);
    }
    ~PacedLoop() { if (warm) (void)gab_keep_warm_destroy(warm); }
};

void announce(const char* fmt, int a, int b = 0) {
    if (!GAB_QUIET) printf(fmt, a, b);
}

}  // namespace

GPUABenchmark::BenchmarkResult GPUABenchmark::runWithIteration(int iterations, int warmupIterations,
                                                              const std::function<void()>& body) {
    PacedLoop loop{this, daw_enabled_, &daw_simulator_, {}};
    // (the launch must outlive the wait for the next slot: four slots, at least 50 ms)
    if (keep_warm_enabled_ && gab_keep_warm_create(&loop.warm, 8, std::min(10.0, std::max(0.05, daw_enabled_ ? 4.0 * daw_simulator_.bufferDuration : 0.05))) != GAB_OK)
        throw std::runtime_error(std::string("keep-warm: ") + gab_last_error());

    // stage 1: untimed passes; an exception costs that pass only
    if (warmupIterations > 0) announce("Running %d warmup iterations...\n", warmupIterations);
    for (int pass = 1; pass <= warmupIterations; ++pass) {
        resetGpuIterationMetrics();
        try {
            body();
            announce("  Warmup %d/%d completed\n", pass, warmupIterations);
        } catch (const std::exception& e) {
            printf("  Warmup iteration %d failed: %s\n", pass, e.what());
        }
        loop.slot();
    }
    if (warmupIterations > 0 && !GAB_QUIET) printf("Warmup complete, starting timed iterations...\n");

    // stage 2: the timed passes
    const size_t want = iterations > 0 ? static_cast<size_t>(iterations) : 0;
    std::vector<float> wall_ms(want), device_ms(want);
    bool device_seen = false;
    for (size_t i = 0; i < want; ++i) {
        resetGpuIterationMetrics();
        wall_ms[i] = static_cast<float>(BenchmarkUtils::BenchmarkTimer::measureKernel(body));
        device_ms[i] = current_iteration_gpu_ms_;
        device_seen = device_seen || device_ms[i] > 0.0f;
        loop.slot();
    }

    // stage 3: the record
    BenchmarkResult r;
    r.benchmark_name = benchmark_name_;
    r.buffer_size = buffer_size_;
    r.track_count = track_count_;
    r.iterations = iterations;
    r.daw_waits = loop.clock.waits;
    r.daw_missed_slots = loop.clock.late;
    r.latencies = std::move(wall_ms);
    r.statistics = BenchmarkUtils::calculateStatistics(r.latencies);
    r.gpu_statistics = {};
    if (device_seen) {
        r.gpu_latencies = std::move(device_ms);
        r.gpu_statistics = BenchmarkUtils::calculateStatistics(r.gpu_latencies);
    }
    const double samples = static_cast<double>(buffer_size_) * static_cast<double>(track_count_);
    const double seconds = r.statistics.mean / 1000.0;
    r.mean_latency_ms = r.statistics.mean;
    r.bytes_processed = buffer_size_ * track_count_ * sizeof(float);
    r.throughput_gbps = (r.bytes_processed / (1024.0 * 1024.0 * 1024.0)) / seconds;   // GiB/s, as the reference prints it
    r.samples_per_sec = samples / seconds;
    return r;
}

// Times the CPU golden on the host: the tracks cut over `threads` threads where the golden is
// separable by track, otherwise the whole golden on one thread (as the reference runs it).
double GPUABenchmark::timeCpuGolden(int threads, int* threads_used) {
    const size_t T = track_count_;
    size_t n = static_cast<size_t>(threads < 1 ? 1 : threads);
    if (n > T) n = T;
    auto t0 = std::chrono::steady_clock::now();
    bool ok = false;
    if (cpuGoldenSlice(0, 0)) {                       // a zero-track probe: "can be cut"
        std::vector<std::thread> pool;
        std::vector<char> done(n, 0);
        t0 = std::chrono::steady_clock::now();
        for (size_t k = 0; k < n; ++k) {
            const size_t lo = T * k / n, hi = T * (k + 1) / n;
            auto work = [this, lo, hi, k, &done]() { done[k] = cpuGoldenSlice(lo, hi - lo) ? 1 : 0; };
            if (k + 1 < n) pool.emplace_back(work); else work();
        }
        for (auto& th : pool) th.join();
        ok = std::all_of(done.begin(), done.end(), [](char c) { return c != 0; });
    } else {
        n = 1;
        t0 = std::chrono::steady_clock::now();
        ok = cpuGoldenWhole();
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (threads_used) *threads_used = static_cast<int>(n);
    return ok ? ms : -1.0;
}

void GPUABenchmark::writeResults(const BenchmarkResult& result, const std::string& filename) {
    std::string path = filename.empty() ? "/tmp/" + result.benchmark_name + "_latencies.txt" : filename;
    BenchmarkUtils::writeLatenciesToFile(result.latencies, path);
}

void GPUABenchmark::printResults(const BenchmarkResult& result) {
    BenchmarkUtils::printStatistics(result.latencies, result.benchmark_name);
    if (!result.gpu_latencies.empty()) {
        BenchmarkUtils::Statistics g = result.gpu_statistics;
        if (g.count == 0) g = BenchmarkUtils::calculateStatistics(result.gpu_latencies);
        std::cout << std::fixed << std::setprecision(3);
        std::cout << "GPU Median:  " << g.median << " ms" << std::endl;
        std::cout << "GPU P95:     " << g.p95 << " ms" << std::endl;
        std::cout << "GPU Mean:    " << g.mean << " ms" << std::endl;
    }
    std::cout << "\nPerformance Metrics:" << std::endl;
    std::cout << std::fixed << std::setprecision(3);
    std::cout << "Throughput:        " << result.throughput_gbps << " GB/s" << std::endl;
    std::cout << "Samples/sec:       " << std::fixed << std::setprecision(0) << result.samples_per_sec << std::endl;
    std::cout << "Bytes processed:   " << result.bytes_processed << std::endl;
}

BenchmarkUtils::BenchmarkParams GPUABenchmark::makeBenchmarkParams(float gainValue) const {
    return BenchmarkUtils::makeBenchmarkParams(buffer_size_, track_count_, gainValue);
}

void GPUABenchmark::resetGpuIterationMetrics() { current_iteration_gpu_ms_ = 0.0f; }

void GPUABenchmark::recordGpuDuration(float milliseconds) {
    if (milliseconds > 0.0f) current_iteration_gpu_ms_ += milliseconds;
}

std::pair<int, int> GPUABenchmark::calculateGridDimensions(int desired_threads_per_block) const {
    int tpb = std::max(std::min(desired_threads_per_block, 512), 32);
    int blocks = (static_cast<int>(track_count_) + tpb - 1) / tpb;
    return std::make_pair(blocks, tpb);
}

void GPUABenchmark::synchronizeAndCheck() {
    HIP_CHECK(hipStreamSynchronize(stream_));
    HIP_CHECK(hipGetLastError());
}

// Element-wise comparison against a golden array.  Contract (cuda/bench_base.cu:181-225): absolute
// difference per element against `tolerance` (a NaN difference fails), max and mean difference, the
// first ten offenders as messages behind a one-line summary.  Built here as a scan that only
// remembers WHERE the first offenders are, and a report written afterwards.
GPUABenchmark::ValidationData GPUABenchmark::compareArrays(const float* got, const float* expected,
                                                          size_t n, float tolerance) {
    ValidationData report;
    if (got == nullptr || expected == nullptr) {
        report.status = ValidationStatus::FATAL;
        report.messages.emplace_back("Null pointer in validation comparison");
        return report;
    }
    constexpr size_t kShown = 10;
    size_t first_bad[kShown];
    size_t shown = 0, offenders = 0;
    float worst = 0.0f, total = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        const float miss = std::abs(got[i] - expected[i]);
        total += miss;
        if (miss > worst) worst = miss;
        if (!(miss <= tolerance)) {                       // also true for NaN
            if (shown < kShown) first_bad[shown++] = i;
            ++offenders;
        }
    }
    report.max_error = worst;
    report.mean_error = n != 0 ? total / static_cast<float>(n) : 0.0f;
    if (offenders == 0) return report;                    // status stays SUCCESS
    report.status = ValidationStatus::FAILURE;
    report.messages.reserve(shown + 1);
    report.messages.push_back("Validation failed: " + std::to_string(offenders) + " out of " + std::to_string(n) +
                              " elements exceeded tolerance");
    for (size_t k = 0; k < shown; ++k) {
        const size_t i = first_bad[k];
        report.messages.push_back("Error at index " + std::to_string(i) + ": expected " + std::to_string(expected[i]) +
                                  ", got " + std::to_string(got[i]) + ", diff " +
                                  std::to_string(std::abs(got[i] - expected[i])));
    }
    return report;
}

GPUABenchmark::ValidationData GPUABenchmark::compareWithReference(const float* cpu_reference, float tolerance) {
    return compareArrays(buffers.h_output, cpu_reference, buffers.element_count, tolerance);
}

void GPUABenchmark::checkGab(int rc, const char* what) {
    if (rc != GAB_OK) throw std::runtime_error(std::string(what) + ": " + gab_last_error());
}
