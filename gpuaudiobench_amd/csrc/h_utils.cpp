// h_utils.cpp — BenchmarkUtils and the CLI globals / legacy writers.
// Behavioural reference: cuda/bench_utils.cu, cuda/globals.cu.
#include <algorithm>
#include <cmath>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits>
#include <numeric>
#include <random>
#include <thread>
#include <chrono>
#include <cstdlib>

#include "gab/bench_utils.hpp"
#include "gab/globals.hpp"

// ---------------------------------------------------------------------------
// globals (cuda/globals.cu:4-9)
// ---------------------------------------------------------------------------
int FS = 48000;
int NTRACKS = 128;
int BUFSIZE = 512;
int NRUNS = 100;
std::string OUTPUT_FILE = "";
bool JSON_OUTPUT = false;
int IR_LENGTH = 0;
int FDTD_GRID = 0;
int CONV_STREAMING = 1;
int MODAL_REAL = 0;
int CONV_BATCH = 0;
int FDTD_STEPS = 0;
int FDTD_FORM = 0;
int DATACOPY_SEQUENTIAL = 0;
int CPU_THREADS = 0;
bool GAB_QUIET = false;

namespace {

// min / max / avg / nearest-rank percentiles as the legacy writers compute them
// (cuda/globals.cu:73-90): note max starts from FLT_MIN and the percentile index
// is size*p truncated.
struct LegacyStats {
    float min, max, avg, p50, p95, p99, threshold;
    bool meets;
};

LegacyStats legacy_stats(const std::vector<float>& vec) {
    LegacyStats s{};
    float sum = 0.0f;
    s.min = std::numeric_limits<float>::max();
    s.max = std::numeric_limits<float>::min();
    for (float v : vec) {
        sum += v;
        if (v < s.min) s.min = v;
        if (v > s.max) s.max = v;
    }
    s.avg = sum / vec.size();
    std::vector<float> sorted = vec;
    std::sort(sorted.begin(), sorted.end());
    s.p50 = sorted[static_cast<size_t>(sorted.size() * 0.50)];
    s.p95 = sorted[static_cast<size_t>(sorted.size() * 0.95)];
    s.p99 = sorted[static_cast<size_t>(sorted.size() * 0.99)];
    s.threshold = 1000.0f * BUFSIZE / FS;          // ms per buffer: the real-time deadline
    s.meets = s.p99 <= s.threshold;
    return s;
}

}  // namespace

void writeVectorToFile(const std::vector<float>& vec, const std::string& filename) {
    std::ofstream file(filename);
    for (float v : vec) file << v << std::endl;
}

void printVectorStats(const std::vector<float>& vec) {
    if (vec.empty()) return;
    LegacyStats s = legacy_stats(vec);
    std::cout << "Min: " << s.min << " Max: " << s.max << " Avg: " << s.avg << std::endl;
    std::cout << "p50: " << s.p50 << " p95: " << s.p95 << " p99: " << s.p99 << std::endl;
    std::cout << "Latency threshold (" << FS << "Hz" << "): " << s.threshold << " ms" << std::endl;
    if (s.p50 > s.threshold) std::cout << "WARNING: p50 exceeds threshold" << std::endl;
    else if (s.p95 > s.threshold) std::cout << "WARNING: p95 exceeds threshold" << std::endl;
    else if (s.p99 > s.threshold) std::cout << "WARNING: p99 exceeds threshold" << std::endl;
    else std::cout << "OK: Measured latencies within threshold. Please consider a margin of safety." << std::endl;
}

void writeCSVResults(const std::vector<float>& vec, const std::string& benchmarkName,
                     const std::string& filename) {
    if (filename.empty() || vec.empty()) return;
    LegacyStats s = legacy_stats(vec);
    bool writeHeader = !std::ifstream(filename).good();
    std::ofstream file(filename, std::ios::app);
    if (writeHeader)
        file << "benchmark,fs,bufferSize,nTracks,nRuns,min_ms,max_ms,avg_ms,p50_ms,p95_ms,p99_ms,"
                "threshold_ms,meets_deadline\n";
    file << benchmarkName << "," << FS << "," << BUFSIZE << "," << NTRACKS << "," << vec.size() << ","
         << s.min << "," << s.max << "," << s.avg << "," << s.p50 << "," << s.p95 << "," << s.p99 << ","
         << s.threshold << "," << (s.meets ? "true" : "false") << "\n";
    file.close();
    std::cout << "Results saved to: " << filename << std::endl;
}

std::string generateJSONResults(const std::vector<float>& vec, const std::string& benchmarkName) {
    LegacyStats s = vec.empty() ? LegacyStats{} : legacy_stats(vec);
    std::string j = "{\n";
    j += "  \"benchmark\": \"" + benchmarkName + "\",\n";
    j += "  \"configuration\": {\n";
    j += "    \"fs\": " + std::to_string(FS) + ",\n";
    j += "    \"bufferSize\": " + std::to_string(BUFSIZE) + ",\n";
    j += "    \"nTracks\": " + std::to_string(NTRACKS) + ",\n";
    j += "    \"nRuns\": " + std::to_string((int)vec.size()) + "\n";
    j += "  },\n";
    j += "  \"statistics\": {\n";
    j += "    \"min_ms\": " + std::to_string(s.min) + ",\n";
    j += "    \"max_ms\": " + std::to_string(s.max) + ",\n";
    j += "    \"avg_ms\": " + std::to_string(s.avg) + ",\n";
    j += "    \"p50_ms\": " + std::to_string(s.p50) + ",\n";
    j += "    \"p95_ms\": " + std::to_string(s.p95) + ",\n";
    j += "    \"p99_ms\": " + std::to_string(s.p99) + "\n";
    j += "  },\n";
    j += "  \"deadline\": {\n";
    j += "    \"threshold_ms\": " + std::to_string(s.threshold) + ",\n";
    j += "    \"meets_deadline\": " + std::string(s.meets ? "true" : "false") + "\n";
    j += "  }\n";
    j += "}\n";
    return j;
}

std::string generateJSONResultsWith(const std::vector<float>& vec, const std::string& benchmarkName,
                                    const std::string& extra_members) {
    std::string j = generateJSONResults(vec, benchmarkName);
    if (extra_members.empty()) return j;
    const size_t close = j.rfind("  }\n}");              // the end of "deadline" and of the object
    if (close == std::string::npos) return j;
    return j.substr(0, close) + "  },\n" + extra_members + "\n}\n";
}

void writeJSONResults(const std::vector<float>& vec, const std::string& benchmarkName,
                      const std::string& filename) {
    if (filename.empty()) {
        std::cout << generateJSONResults(vec, benchmarkName) << std::endl;
        return;
    }
    std::ofstream file(filename);
    file << generateJSONResults(vec, benchmarkName);
    file.close();
    std::cout << "JSON results saved to: " << filename << std::endl;
}

// ---------------------------------------------------------------------------
namespace BenchmarkUtils {

BenchmarkParams makeBenchmarkParams(size_t bufferSize, size_t trackCount, float gainValue) {
    BenchmarkParams p;
    p.bufferSize = static_cast<uint32_t>(bufferSize);
    p.trackCount = static_cast<uint32_t>(trackCount);
    p.totalSamples = static_cast<uint32_t>(bufferSize * trackCount);
    p.gainValue = gainValue;
    return p;
}

void checkHipError(hipError_t error, const std::string& message) {
    if (error != hipSuccess) throw std::runtime_error(message + ": " + hipGetErrorString(error));
}

void freeDeviceBuffers(std::initializer_list<void*> buffers) {
    for (void* b : buffers)
        if (b != nullptr) (void)hipFree(b);
}

void freeHostBuffers(std::initializer_list<void*> buffers) {
    for (void* b : buffers)
        if (b != nullptr) (void)hipHostFree(b);
}

// ---- timers ------------------------------------------------------------------
void BenchmarkTimer::start() {
    start_time = std::chrono::high_resolution_clock::now();
    is_running = true;
}

void BenchmarkTimer::stop() {
    end_time = std::chrono::high_resolution_clock::now();
    is_running = false;
}

double BenchmarkTimer::elapsed_ms() const {
    auto until = is_running ? std::chrono::high_resolution_clock::now() : end_time;
    auto us = std::chrono::duration_cast<std::chrono::microseconds>(until - start_time);
    return us.count() / 1000.0;
}

double BenchmarkTimer::measureKernel(std::function<void()> kernel) {
    BenchmarkTimer t;
    t.start();
    kernel();
    t.stop();
    return t.elapsed_ms();
}

void BenchmarkTimer::reset() { is_running = false; }

HipEventTimer::HipEventTimer() {
    HIP_CHECK(hipEventCreate(&start_event));
    HIP_CHECK(hipEventCreate(&stop_event));
}

HipEventTimer::~HipEventTimer() { destroy(); }

HipEventTimer::HipEventTimer(HipEventTimer&& o) noexcept
    : start_event(o.start_event), stop_event(o.stop_event), running(o.running) {
    o.start_event = nullptr;
    o.stop_event = nullptr;
    o.running = false;
}

HipEventTimer& HipEventTimer::operator=(HipEventTimer&& o) noexcept {
    if (this != &o) {
        destroy();
        start_event = o.start_event;
        stop_event = o.stop_event;
        running = o.running;
        o.start_event = nullptr;
        o.stop_event = nullptr;
        o.running = false;
    }
    return *this;
}

void HipEventTimer::destroy() {
    if (start_event) { (void)hipEventDestroy(start_event); start_event = nullptr; }
    if (stop_event) { (void)hipEventDestroy(stop_event); stop_event = nullptr; }
    running = false;
}

void HipEventTimer::start(hipStream_t stream) {
    HIP_CHECK(hipEventRecord(start_event, stream));
    running = true;
}

float HipEventTimer::stop(hipStream_t stream) {
    if (!running) return 0.0f;
    HIP_CHECK(hipEventRecord(stop_event, stream));
    HIP_CHECK(hipEventSynchronize(stop_event));
    float ms = 0.0f;
    HIP_CHECK(hipEventElapsedTime(&ms, start_event, stop_event));
    running = false;
    return ms;
}

void HipEventTimer::reset() { running = false; }

double timeOnStream(hipStream_t stream, const std::function<void()>& enqueue) {
    HipEventTimer t;
    t.start(stream);
    enqueue();
    return static_cast<double>(t.stop(stream));
}

void collectLatencies(std::vector<float>& latencies, std::function<void()> benchmark, int iterations) {
    latencies.clear();
    latencies.reserve(iterations);
    for (int i = 0; i < iterations; ++i)
        latencies.push_back(static_cast<float>(BenchmarkTimer::measureKernel(benchmark)));
}

// ---- DAW-style pacing ------------------------------------------------------------
double DAWSimulator::now() {
    using clk = std::chrono::steady_clock;
    return std::chrono::duration<double>(clk::now().time_since_epoch()).count();
}

void DAWSimulator::wait(DAWSimulationState& st) const {
    const double t = now();
    if (!st.started) {
        st.started = true;
        st.next_start = t + bufferDuration;
    }
    double jitter = 0.0;
    if (jitterSeconds > 0.0) {
        unsigned int x = st.rng;                       // xorshift32 -> U(-j, +j)
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        st.rng = x;
        jitter = ((double)x / 4294967295.0 * 2.0 - 1.0) * jitterSeconds;
    }
    const double target = st.next_start + jitter;
    ++st.waits;
    if (target > t) {
        if (mode == DAWSimulationMode::SLEEP)
            std::this_thread::sleep_for(std::chrono::duration<double>(target - t));
        else
            while (now() < target) {}
    } else {
        ++st.late;
    }
    st.next_start += bufferDuration;                   // slots stay on the grid, late or not
}

// ---- harness configuration ---------------------------------------------------------
BenchmarkConfig BenchmarkConfig::fromCommandLine(int argc, char** argv) {
    BenchmarkConfig c;
    struct IntFlag { const char* name; int BenchmarkConfig::*field; };
    struct BoolFlag { const char* name; bool BenchmarkConfig::*field; bool value; };
    static const IntFlag ints[] = {
        {"--buffersize", &BenchmarkConfig::bufferSize}, {"--ntracks", &BenchmarkConfig::trackCount},
        {"--fs", &BenchmarkConfig::sampleRate},         {"--nruns", &BenchmarkConfig::iterations},
        {"--warmup", &BenchmarkConfig::warmupIterations}, {"--blocksize", &BenchmarkConfig::preferredBlockSize},
    };
    static const BoolFlag bools[] = {
        {"--validate", &BenchmarkConfig::enableValidation, true},
        {"--profile", &BenchmarkConfig::enableProfiling, true},
        {"--verbose", &BenchmarkConfig::verboseOutput, true},
        {"--no-file", &BenchmarkConfig::writeToFile, false},
        {"--quiet", &BenchmarkConfig::printStatistics, false},
        {"--optimal-occupancy", &BenchmarkConfig::useOptimalOccupancy, true},
        {"--dawsim", &BenchmarkConfig::enableDAWSimulation, true},
    };
    for (int i = 1; i < argc; ++i) {
        const std::string a(argv[i]);
        const bool more = i + 1 < argc;
        bool taken = false;
        for (const IntFlag& f : ints)
            if (a == f.name && more) { c.*(f.field) = std::atoi(argv[++i]); taken = true; break; }
        if (taken) continue;
        for (const BoolFlag& f : bools)
            if (a == f.name) { c.*(f.field) = f.value; taken = true; break; }
        if (taken) continue;
        if (a == "--output" && more) c.outputDirectory = argv[++i];
        else if (a == "--prefix" && more) c.outputPrefix = argv[++i];
    }
    return c;
}

bool BenchmarkConfig::validate() const {
    struct Range { const char* what; int value, lo, hi; };
    const Range checks[] = {
        {"buffer size", bufferSize, 1, 8192},
        {"track count", trackCount, 1, 2048},
        {"iterations", iterations, 1, 10000},
        {"block size", preferredBlockSize, 32, 1024},
    };
    for (const Range& r : checks)
        if (r.value < r.lo || r.value > r.hi) {
            std::cerr << "Invalid " << r.what << ": " << r.value << std::endl;
            return false;
        }
    return true;
}

void BenchmarkConfig::print() const {
    std::cout << "=== Benchmark Configuration ===\n"
              << "Buffer Size: " << bufferSize << " samples\n"
              << "Track Count: " << trackCount << "\n"
              << "Sample Rate: " << sampleRate << " Hz\n"
              << "Iterations: " << iterations << "\n"
              << "Warmup: " << warmupIterations << "\n"
              << "Validation: " << (enableValidation ? "ON" : "OFF") << "\n"
              << "Profiling: " << (enableProfiling ? "ON" : "OFF") << "\n"
              << "===============================" << std::endl;
}

// ---- generators -----------------------------------------------------------------
void generateRandomAudioData(float* buffer, size_t samples, unsigned int seed) {
    std::mt19937 gen(seed);
    std::uniform_real_distribution<float> dist(-1.0f, 1.0f);
    for (size_t i = 0; i < samples; ++i) buffer[i] = dist(gen);
}

void generateRandomAudioDataFrom(float* buffer, size_t samples, unsigned int seed, unsigned long long skip) {
    std::mt19937 gen(seed);
    gen.discard(skip);                         // one engine draw per float (generate_canonical<float,24>)
    std::uniform_real_distribution<float> dist(-1.0f, 1.0f);
    for (size_t i = 0; i < samples; ++i) buffer[i] = dist(gen);
}

GlibcRand::GlibcRand(unsigned int seed) : f_(3), b_(0) {
    int word = static_cast<int>(seed ? seed : 1u);
    r_[0] = static_cast<unsigned int>(word);
    for (int i = 1; i < 31; ++i) {                       // word = 16807 * word mod (2^31 - 1), Schrage's split
        const long hi = word / 127773, lo = word % 127773;
        long w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        word = static_cast<int>(w);
        r_[i] = static_cast<unsigned int>(word);
    }
    for (int i = 0; i < 310; ++i) (void)next();
}

int GlibcRand::next() {
    r_[f_] += r_[b_];
    const unsigned int out = r_[f_] >> 1;
    if (++f_ >= 31) f_ = 0;
    if (++b_ >= 31) b_ = 0;
    return static_cast<int>(out);
}

void generateImpulseResponse(float* buffer, int length, float frequency, WindowType window_type) {
    for (int i = 0; i < length; ++i) {
        float t = static_cast<float>(i) - static_cast<float>(length) / 2.0f;
        float sinc_val = 1.0f;
        if (t != 0.0f) {
            float arg = 2.0f * M_PI * frequency * t;
            sinc_val = std::sin(arg) / arg;
        }
        float n = static_cast<float>(i) / static_cast<float>(length - 1);
        float w = 1.0f;
        switch (window_type) {
            case WindowType::RECTANGULAR: w = 1.0f; break;
            case WindowType::HAMMING: w = 0.54f - 0.46f * std::cos(2.0f * M_PI * n); break;
            case WindowType::HANN: w = 0.5f * (1.0f - std::cos(2.0f * M_PI * n)); break;
            case WindowType::BLACKMAN:
                w = 0.42f - 0.5f * std::cos(2.0f * M_PI * n) + 0.08f * std::cos(4.0f * M_PI * n);
                break;
        }
        buffer[i] = sinc_val * w;
    }
    float sum = 0.0f;
    for (int i = 0; i < length; ++i) sum += std::abs(buffer[i]);
    if (sum > 0.0f)
        for (int i = 0; i < length; ++i) buffer[i] /= sum;
}

void initializeTestPattern(float* buffer, size_t samples, TestPattern pattern) {
    switch (pattern) {
        case TestPattern::ZEROS: std::fill(buffer, buffer + samples, 0.0f); break;
        case TestPattern::ONES: std::fill(buffer, buffer + samples, 1.0f); break;
        case TestPattern::RAMP:
            for (size_t i = 0; i < samples; ++i)
                buffer[i] = static_cast<float>(i) / static_cast<float>(samples - 1);
            break;
        case TestPattern::SINE_WAVE:
            for (size_t i = 0; i < samples; ++i)
                buffer[i] = std::sin(2.0f * M_PI * static_cast<float>(i) / 64.0f);
            break;
        case TestPattern::WHITE_NOISE: generateRandomAudioData(buffer, samples, 42); break;
    }
}

BiquadCoefficients generateLowpassCoefficients(float cutoff_freq, float q) {
    float omega = 2.0f * M_PI * cutoff_freq;
    float sn = std::sin(omega), cs = std::cos(omega);
    float alpha = sn / (2.0f * q);
    float a0 = 1.0f + alpha;
    BiquadCoefficients c;
    c.b0 = ((1.0f - cs) / 2.0f) / a0;
    c.b1 = (1.0f - cs) / a0;
    c.b2 = ((1.0f - cs) / 2.0f) / a0;
    c.a1 = (-2.0f * cs) / a0;
    c.a2 = (1.0f - alpha) / a0;
    return c;
}

void generateConv1DImpulseResponses(float* ir, int L, size_t track_offset, size_t n_tracks,
                                    size_t total_tracks) {
    const float PI = 3.14159265358979323846f;
    for (size_t lt = 0; lt < n_tracks; ++lt) {
        const size_t track = track_offset + lt;
        const float freq = 0.1f + 0.05f * static_cast<float>(track) / static_cast<float>(total_tracks);
        for (int i = 0; i < L; ++i) {
            float t = static_cast<float>(i) - static_cast<float>(L) / 2.0f;
            float window = 0.54f - 0.46f * cosf(2.0f * PI * static_cast<float>(i) / static_cast<float>(L - 1));
            float sinc = (t == 0.0f) ? 1.0f : sinf(2.0f * PI * freq * t) / (2.0f * PI * freq * t);
            ir[lt * static_cast<size_t>(L) + i] = window * sinc / static_cast<float>(L);
        }
    }
}

void generateConvAccelImpulseResponses(float* ir, int L, size_t track_offset, size_t n_tracks,
                                       size_t total_tracks) {
    // M_PI is a double: trig arguments are formed in double and narrowed at the call
    for (size_t lt = 0; lt < n_tracks; ++lt) {
        const size_t track = track_offset + lt;
        const float freq = 0.1f + 0.05f * (float)track / (float)total_tracks;
        for (int i = 0; i < L; ++i) {
            float t = (float)i - (float)L / 2.0f;
            float window = 0.54f - 0.46f * cosf(2.0f * M_PI * (float)i / (float)(L - 1));
            float sinc = (t == 0.0f) ? 1.0f : sinf(2.0f * M_PI * freq * t) / (2.0f * M_PI * freq * t);
            ir[lt * static_cast<size_t>(L) + i] = window * sinc / (float)L;
        }
    }
}

// ---- statistics --------------------------------------------------------------------
Statistics calculateStatistics(const std::vector<float>& latencies) {
    if (latencies.empty()) return {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0};
    Statistics s;
    s.count = latencies.size();
    std::vector<float> sorted = latencies;
    std::sort(sorted.begin(), sorted.end());
    s.min_val = sorted.front();
    s.max_val = sorted.back();
    s.mean = std::accumulate(latencies.begin(), latencies.end(), 0.0f) / static_cast<float>(latencies.size());
    const size_t mid = latencies.size() / 2;
    s.median = (latencies.size() % 2 == 0) ? (sorted[mid - 1] + sorted[mid]) / 2.0f : sorted[mid];
    float var = 0.0f;
    for (float v : latencies) var += (v - s.mean) * (v - s.mean);
    var /= static_cast<float>(latencies.size() - 1);       // sample variance, as the reference
    s.std_dev = std::sqrt(var);
    auto percentile = [&](float p) {
        float index = p / 100.0f * static_cast<float>(sorted.size() - 1);
        size_t lo = static_cast<size_t>(std::floor(index)), hi = static_cast<size_t>(std::ceil(index));
        if (lo == hi) return sorted[lo];
        float w = index - static_cast<float>(lo);
        return sorted[lo] * (1.0f - w) + sorted[hi] * w;
    };
    s.p95 = percentile(95.0f);
    s.p99 = percentile(99.0f);
    return s;
}

void writeLatenciesToFile(const std::vector<float>& latencies, const std::string& filename) {
    Statistics s = calculateStatistics(latencies);
    std::ofstream file(filename);
    if (!file.is_open()) throw std::runtime_error("Failed to open file for writing: " + filename);
    file << std::fixed << std::setprecision(3);
    file << "# Latency Statistics (ms)\n";
    file << "# Count: " << s.count << "\n";
    file << "# Mean: " << s.mean << "\n";
    file << "# Median: " << s.median << "\n";
    file << "# Std Dev: " << s.std_dev << "\n";
    file << "# Min: " << s.min_val << "\n";
    file << "# Max: " << s.max_val << "\n";
    file << "# P95: " << s.p95 << "\n";
    file << "# P99: " << s.p99 << "\n";
    file << "#\n";
    file << "# Raw latencies:\n";
    for (float v : latencies) file << v << "\n";
}

void printStatistics(const std::vector<float>& latencies, const std::string& benchmark_name) {
    Statistics s = calculateStatistics(latencies);
    std::cout << "\n=== " << benchmark_name << " Benchmark Results ===\n";
    std::cout << std::fixed << std::setprecision(3);
    std::cout << "Iterations: " << s.count << "\n";
    std::cout << "Mean:       " << s.mean << " ms\n";
    std::cout << "Median:     " << s.median << " ms\n";
    std::cout << "Std Dev:    " << s.std_dev << " ms\n";
    std::cout << "Min:        " << s.min_val << " ms\n";
    std::cout << "Max:        " << s.max_val << " ms\n";
    std::cout << "P95:        " << s.p95 << " ms\n";
    std::cout << "P99:        " << s.p99 << " ms\n";
    std::cout << "==========================================\n\n";
}

}  // namespace BenchmarkUtils
