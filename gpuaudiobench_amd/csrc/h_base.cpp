// h_base.cpp — GPUABenchmark: buffers, the warm-up + timed loop, statistics,
// validation bookkeeping.  Behavioural reference: cuda/bench_base.cu.
#include <algorithm>
#include <cmath>
#include <iomanip>

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
    BenchmarkUtils::generateRandomAudioData(buffers.h_input, buffers.element_count, seed);
}

GPUABenchmark::BenchmarkResult GPUABenchmark::runKernelBenchmark(int iterations, int warmupIterations) {
    return runWithIteration(iterations, warmupIterations, [this]() { this->runKernel(); });
}

GPUABenchmark::BenchmarkResult GPUABenchmark::runBenchmark(int iterations, int warmupIterations) {
    return runWithIteration(iterations, warmupIterations, [this]() { this->performBenchmarkIteration(); });
}

GPUABenchmark::BenchmarkResult GPUABenchmark::runWithIteration(int iterations, int warmupIterations,
                                                              const std::function<void()>& body) {
    BenchmarkResult result;
    result.benchmark_name = benchmark_name_;
    result.buffer_size = buffer_size_;
    result.track_count = track_count_;
    result.iterations = iterations;

    BenchmarkUtils::DAWSimulationState daw_state;
    if (warmupIterations > 0) {
        if (!GAB_QUIET) printf("Running %d warmup iterations...\n", warmupIterations);
        for (int i = 0; i < warmupIterations; ++i) {
            try {
                resetGpuIterationMetrics();
                body();
                if (!GAB_QUIET) printf("  Warmup %d/%d completed\n", i + 1, warmupIterations);
            } catch (const std::exception& e) {
                // a failing warm-up is reported and skipped, as in the reference
                printf("  Warmup iteration %d failed: %s\n", i + 1, e.what());
            }
            if (daw_enabled_) daw_simulator_.wait(daw_state);
        }
        if (!GAB_QUIET) printf("Warmup complete, starting timed iterations...\n");
    }

    result.latencies.reserve(std::max(iterations, 0));
    std::vector<float> gpu;
    gpu.reserve(std::max(iterations, 0));
    for (int i = 0; i < iterations; ++i) {
        resetGpuIterationMetrics();
        double ms = BenchmarkUtils::BenchmarkTimer::measureKernel(body);
        result.latencies.push_back(static_cast<float>(ms));
        gpu.push_back(current_iteration_gpu_ms_);
        if (daw_enabled_) daw_simulator_.wait(daw_state);
    }
    result.daw_waits = daw_state.waits;
    result.daw_missed_slots = daw_state.late;
    result.statistics = BenchmarkUtils::calculateStatistics(result.latencies);

    const bool any_gpu = std::any_of(gpu.begin(), gpu.end(), [](float v) { return v > 0.0f; });
    if (any_gpu) {
        result.gpu_latencies = std::move(gpu);
        result.gpu_statistics = BenchmarkUtils::calculateStatistics(result.gpu_latencies);
    } else {
        result.gpu_statistics = {};
    }

    const size_t total = buffer_size_ * track_count_;
    result.bytes_processed = total * sizeof(float);
    result.mean_latency_ms = result.statistics.mean;
    const double sec = result.mean_latency_ms / 1000.0;
    result.throughput_gbps = (result.bytes_processed / (1024.0 * 1024.0 * 1024.0)) / sec;
    result.samples_per_sec = total / sec;
    return result;
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

GPUABenchmark::ValidationData GPUABenchmark::compareArrays(const float* got, const float* expected,
                                                          size_t n, float tolerance) {
    ValidationData v;
    if (!got || !expected) {
        v.status = ValidationStatus::FATAL;
        v.messages.push_back("Null pointer in validation comparison");
        return v;
    }
    float sum = 0.0f, mx = 0.0f;
    int bad = 0;
    for (size_t i = 0; i < n; ++i) {
        float d = std::abs(got[i] - expected[i]);
        sum += d;
        mx = std::max(mx, d);
        if (d > tolerance || std::isnan(d)) {
            ++bad;
            if (v.messages.size() < 10)
                v.messages.push_back("Error at index " + std::to_string(i) + ": expected " +
                                     std::to_string(expected[i]) + ", got " + std::to_string(got[i]) +
                                     ", diff " + std::to_string(d));
        }
    }
    v.mean_error = n ? sum / static_cast<float>(n) : 0.0f;
    v.max_error = mx;
    if (bad > 0) {
        v.status = ValidationStatus::FAILURE;
        v.messages.insert(v.messages.begin(), "Validation failed: " + std::to_string(bad) + " out of " +
                                                  std::to_string(n) + " elements exceeded tolerance");
    }
    return v;
}

GPUABenchmark::ValidationData GPUABenchmark::compareWithReference(const float* cpu_reference, float tolerance) {
    return compareArrays(buffers.h_output, cpu_reference, buffers.element_count, tolerance);
}

void GPUABenchmark::checkGab(int rc, const char* what) {
    if (rc != GAB_OK) throw std::runtime_error(std::string(what) + ": " + gab_last_error());
}
