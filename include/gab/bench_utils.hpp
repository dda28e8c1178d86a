// bench_utils.hpp — memory, timing, data generation, launch and statistics
// helpers of the harness.  Mirrors the public surface of the reference's
// cuda/bench_utils.cuh (namespace BenchmarkUtils) on the HIP runtime:
// cudaMallocHost -> hipHostMalloc, cudaEvent -> hipEvent, CUDA_CHECK ->
// HIP_CHECK (CUDA_CHECK stays as an alias).  All templates are header-only, so
// every element type works (the reference instantiates only float/int/
// cufftComplex in its .cu and therefore fails to link — SURVEY §2.3-2).
#pragma once

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstring>
#include <functional>
#include <initializer_list>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace BenchmarkUtils {

// ---- kernel parameter block (cuda/bench_utils.cuh:22-31) -------------------
struct BenchmarkParams {
    uint32_t bufferSize = 0;
    uint32_t trackCount = 0;
    uint32_t totalSamples = 0;
    float gainValue = 0.0f;
};

BenchmarkParams makeBenchmarkParams(size_t bufferSize, size_t trackCount, float gainValue = 0.0f);

// ---- errors (cuda/bench_utils.cuh:246-254) ----------------------------------
void checkHipError(hipError_t error, const std::string& message);
inline void checkCudaError(hipError_t error, const std::string& message) { checkHipError(error, message); }

#define HIP_CHECK(call)                                                   \
    do {                                                                  \
        hipError_t gab_err_ = (call);                                     \
        if (gab_err_ != hipSuccess) BenchmarkUtils::checkHipError(gab_err_, #call); \
    } while (0)
#define CUDA_CHECK(call) HIP_CHECK(call)

// ---- memory (cuda/bench_utils.cuh:140-154, bench_utils.cu:101-171) -----------
template <typename T>
T* allocateDeviceBuffer(size_t count, const std::string& name = "device buffer") {
    T* ptr = nullptr;
    const size_t bytes = count * sizeof(T);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&ptr), bytes);
    if (e != hipSuccess)
        throw std::runtime_error("Failed to allocate " + name + " (" + std::to_string(bytes) +
                                 " bytes): " + hipGetErrorString(e));
    return ptr;
}

// Pinned (page-locked) host memory: what hipMemcpyAsync needs to be truly async.
template <typename T>
T* allocateHostBuffer(size_t count, const std::string& name = "host buffer") {
    T* ptr = nullptr;
    const size_t bytes = count * sizeof(T);
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&ptr), bytes, hipHostMallocDefault);
    if (e != hipSuccess)
        throw std::runtime_error("Failed to allocate pinned " + name + " (" + std::to_string(bytes) +
                                 " bytes): " + hipGetErrorString(e));
    return ptr;
}

template <typename T>
void copyToDevice(T* dst, const T* src, size_t count, hipStream_t stream = nullptr) {
    if (dst == nullptr || src == nullptr) throw std::invalid_argument("copyToDevice received null pointer");
    const size_t bytes = count * sizeof(T);
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess)
        throw std::runtime_error("Failed to copy " + std::to_string(bytes) + " bytes to device: " +
                                 hipGetErrorString(e));
}

template <typename T>
void copyToHost(T* dst, const T* src, size_t count, hipStream_t stream = nullptr) {
    if (dst == nullptr || src == nullptr) throw std::invalid_argument("copyToHost received null pointer");
    const size_t bytes = count * sizeof(T);
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess)
        throw std::runtime_error("Failed to copy " + std::to_string(bytes) + " bytes to host: " +
                                 hipGetErrorString(e));
}

void freeDeviceBuffers(std::initializer_list<void*> buffers);
void freeHostBuffers(std::initializer_list<void*> buffers);

// ---- timing (cuda/bench_utils.cuh:160-207) -------------------------------------
class BenchmarkTimer {
public:
    void start();
    void stop();
    double elapsed_ms() const;          // microsecond resolution, like the reference
    static double measureKernel(std::function<void()> kernel);
    void reset();

private:
    std::chrono::high_resolution_clock::time_point start_time;
    std::chrono::high_resolution_clock::time_point end_time;
    bool is_running = false;
};

class HipEventTimer {
public:
    HipEventTimer();
    ~HipEventTimer();
    HipEventTimer(const HipEventTimer&) = delete;
    HipEventTimer& operator=(const HipEventTimer&) = delete;
    HipEventTimer(HipEventTimer&& other) noexcept;
    HipEventTimer& operator=(HipEventTimer&& other) noexcept;

    void start(hipStream_t stream = nullptr);
    float stop(hipStream_t stream = nullptr);     // ms; 0 if not running
    void reset();
    bool isRunning() const { return running; }

private:
    hipEvent_t start_event = nullptr;
    hipEvent_t stop_event = nullptr;
    bool running = false;
    void destroy();
};
using CudaEventTimer = HipEventTimer;

void collectLatencies(std::vector<float>& latencies, std::function<void()> benchmark, int iterations);

// ---- DAW-style pacing ------------------------------------------------------------
// The CUDA reference only declares the knobs (cuda/globals.cuh:27-30 ENABLE_DAWSIM_SLEEP /
// SLEEP_MS / ENABLE_DAWSIM_SPIN, cuda/bench_utils.cuh:57-58,95 --dawsim); the scheduler itself
// exists in its Metal port (metal-swift/MetalSwiftBench/Core/BenchmarkUtilities.swift:140-178):
// iteration k+1 may not start before t0 + (k+1)*bufferDuration (+/- jitter), reached by
// sleeping or spinning, so latencies are measured with the device idle between buffers the
// way a DAW's audio callback leaves it.
enum class DAWSimulationMode { SPIN, SLEEP };

struct DAWSimulationState {
    bool started = false;
    double next_start = 0.0;          // seconds on the steady clock
    unsigned long long waits = 0;     // calls so far
    unsigned long long late = 0;      // calls that found their slot already past (a missed deadline)
    unsigned int rng = 0x9E3779B9u;   // jitter source (xorshift32); fixed seed: runs are repeatable
};

struct DAWSimulator {
    double bufferDuration = 512.0 / 48000.0;   // seconds: BUFSIZE / FS
    DAWSimulationMode mode = DAWSimulationMode::SPIN;
    double jitterSeconds = 0.0;

    static double now();                       // steady clock, seconds
    void wait(DAWSimulationState& state) const;
};

// ---- harness configuration (cuda/bench_utils.cuh:36-132) ---------------------------
// Declared by the reference, not read by its main.cu; kept so code written against it compiles.
struct BenchmarkConfig {
    int bufferSize = 512;
    int trackCount = 128;
    int sampleRate = 48000;
    int iterations = 100;
    int warmupIterations = 3;
    bool enableValidation = false;
    bool enableProfiling = false;
    bool verboseOutput = false;
    std::string outputDirectory = "/tmp";
    std::string outputPrefix = "";
    bool writeToFile = true;
    bool printStatistics = true;
    int preferredBlockSize = 256;
    bool useOptimalOccupancy = false;
    bool enableDAWSimulation = false;
    int dawSleepMs = 90;

    static BenchmarkConfig fromCommandLine(int argc, char** argv);
    bool validate() const;            // same limits and messages as the reference (:104-131)
    void print() const;
};

// ---- data generation (cuda/bench_utils.cuh:213-240) ----------------------------
void generateRandomAudioData(float* buffer, size_t samples, unsigned int seed = 42);
// The same stream from its `skip`-th value on: a channel shard takes its rows of the one flat
// track-major stream the reference draws over ALL tracks (additive).
void generateRandomAudioDataFrom(float* buffer, size_t samples, unsigned int seed, unsigned long long skip);

// glibc's rand() (random_r.c TYPE_3: x^31 + x^3 + 1 seeded through the 16807 generator, 310 values discarded,
// result = state >> 1) as an object: the stream srand(seed) starts — seed 1 is what an unseeded rand() gives a fresh
// process, which is how the reference draws its FFT input (cuda/bench_fft.cu:37: one benchmark per process) — that
// does not depend on what else in the process has called rand(), can be shared by no other thread, and can be
// entered at any position (a channel shard skips the draws of the tracks before its own).  Additive.
class GlibcRand {
public:
    static constexpr int kMax = 2147483647;
    explicit GlibcRand(unsigned int seed = 1);
    int next();
    void discard(unsigned long long n) { while (n--) (void)next(); }
private:
    unsigned int r_[31];
    int f_, b_;
};

enum class WindowType { RECTANGULAR, HAMMING, HANN, BLACKMAN };
void generateImpulseResponse(float* buffer, int length, float frequency,
                             WindowType window_type = WindowType::HAMMING);

enum class TestPattern { ZEROS, ONES, RAMP, SINE_WAVE, WHITE_NOISE };
void initializeTestPattern(float* buffer, size_t samples, TestPattern pattern);

struct BiquadCoefficients { float b0, b1, b2, a1, a2; };
BiquadCoefficients generateLowpassCoefficients(float cutoff_freq, float q = 0.707f);

// IR banks of the two convolution benchmarks.  The formulas use the GLOBAL
// track index and count, so a shard passes its offset and the global total.
// conv1d: cuda/bench_conv1d.cu:159-181 (float pi); accel: cuda/bench_conv1d_accel.cu:152-173
// (double M_PI intermediates).
void generateConv1DImpulseResponses(float* ir, int ir_len, size_t track_offset, size_t n_tracks,
                                    size_t total_tracks);
void generateConvAccelImpulseResponses(float* ir, int ir_len, size_t track_offset, size_t n_tracks,
                                       size_t total_tracks);

// ---- launch helpers (cuda/bench_utils.cuh:260-329) ------------------------------
// Same contract: launch, check the launch, synchronise, check execution.
template <typename KernelFunc, typename... Args>
void launchKernel(KernelFunc kernel, dim3 gridDim, dim3 blockDim, Args&&... args) {
    hipLaunchKernelGGL(kernel, gridDim, blockDim, 0, nullptr, std::forward<Args>(args)...);
    hipError_t launchError = hipGetLastError();
    if (launchError != hipSuccess)
        throw std::runtime_error("HIP kernel launch failed: " + std::string(hipGetErrorString(launchError)));
    hipError_t syncError = hipDeviceSynchronize();
    if (syncError != hipSuccess)
        throw std::runtime_error("HIP kernel execution failed: " + std::string(hipGetErrorString(syncError)));
}

template <typename KernelFunc, typename... Args>
void launchKernel1D(KernelFunc kernel, size_t totalThreads, int preferredBlockSize = 256, Args&&... args) {
    int blockSize = preferredBlockSize < static_cast<int>(totalThreads) ? preferredBlockSize
                                                                        : static_cast<int>(totalThreads);
    if (blockSize < 64) blockSize = 64;           // one wavefront minimum on gfx950
    int gridSize = static_cast<int>((totalThreads + blockSize - 1) / blockSize);
    launchKernel(kernel, dim3(gridSize), dim3(blockSize), std::forward<Args>(args)...);
}

// cuda/bench_utils.cuh:298-318: block size from the occupancy calculator, 256 if it declines.
template <typename KernelFunc, typename... Args>
void launchKernelOptimal(KernelFunc kernel, size_t totalThreads, Args&&... args) {
    int minGridSize = 0, blockSize = 0;
    if (hipOccupancyMaxPotentialBlockSize(&minGridSize, &blockSize, kernel, 0, 0) != hipSuccess || blockSize <= 0) {
        launchKernel1D(kernel, totalThreads, 256, std::forward<Args>(args)...);
        return;
    }
    const int gridSize = static_cast<int>((totalThreads + blockSize - 1) / blockSize);
    launchKernel(kernel, dim3(gridSize), dim3(blockSize), std::forward<Args>(args)...);
}

template <typename KernelFunc, typename... Args>
double launchKernelTimed(KernelFunc kernel, dim3 gridDim, dim3 blockDim, Args&&... args) {
    HipEventTimer timer;
    timer.start();
    launchKernel(kernel, gridDim, blockDim, std::forward<Args>(args)...);
    return static_cast<double>(timer.stop());
}

// Times any callable that enqueues work on `stream` with an event pair.
double timeOnStream(hipStream_t stream, const std::function<void()>& enqueue);

// ---- statistics (cuda/bench_utils.cuh:335-350) ----------------------------------
struct Statistics {
    float mean;
    float median;
    float std_dev;
    float min_val;
    float max_val;
    float p95;
    float p99;
    size_t count;
};

Statistics calculateStatistics(const std::vector<float>& latencies);
void writeLatenciesToFile(const std::vector<float>& latencies, const std::string& filename);
void printStatistics(const std::vector<float>& latencies, const std::string& benchmark_name);

}  // namespace BenchmarkUtils
