// gpubench_main.cpp — command-line driver, flag-compatible with the reference's
// cuda/main.cu:236-328 (--help --list --json --benchmark --fs --bufferSize
// --nTracks --nRuns --outputfile; default benchmark RndMemRead; exit code 0/1),
// plus the flags the BASELINE configs need: --irLength --fdtdGrid --convMode.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "gab/benchmarks.hpp"

namespace {

void printBenchmarkList() {
    printf("Available benchmarks:\n");
    for (const auto& n : gab::benchmarkNames()) printf("%s\n", n.c_str());
}

void printHelp() {
    printf("GPU Audio Benchmark Suite (HIP / MI355X)\n");
    printf("========================================\n");
    printf("Real-time GPGPU audio processing benchmarks\n\n");
    printf("Usage: gpubench [options]\n\n");
    printf("Options:\n");
    printf("  --help              Print this help message\n");
    printf("  --list              List all available benchmarks\n");
    printf("  --benchmark [name]  Run specific benchmark (see list below)\n");
    printf("  --fs [rate]         Set sampling rate (default: 48000)\n");
    printf("  --bufferSize [size] Set buffer size (default: 512)\n");
    printf("  --nTracks [count]   Set number of tracks (default: 128)\n");
    printf("  --nRuns [count]     Set number of iterations (default: 100)\n");
    printf("  --outputfile [file] Save results to CSV file\n");
    printf("  --json              Output results in JSON format\n");
    printf("  --irLength [taps]   Impulse-response length for Conv1D / Conv1D_accel\n");
    printf("  --fdtdGrid [n]      FDTD3D grid edge including the boundary shell (default: 52)\n");
    printf("  --convMode [m]      Conv1D_accel: stream (carried history, default) | stateless\n");
    printf("  --modalMode [m]     ModalFilterBank: placeholder (the CUDA port, default) | bank (real phasor bank)\n");
    printf("  --dawsim            Pace iterations to one buffer slot each (bufferSize / fs)\n");
    printf("  --dawsim-mode [m]   spin | sleep (default: spin)\n");
    printf("  --dawsim-jitter-us [us]  Uniform jitter on each slot (default: 0)\n");
    printf("\nAvailable Benchmarks:\n=====================\n");
    printf("\nData Transfer:\n");
    printf("  datacopy0199     - 1%% input, 99%% output transfer\n");
    printf("  datacopy2080     - 20%% input, 80%% output transfer\n");
    printf("  datacopy5050     - 50%% input, 50%% output transfer\n");
    printf("  datacopy8020     - 80%% input, 20%% output transfer\n");
    printf("  datacopy9901     - 99%% input, 1%% output transfer\n");
    printf("\nBasic Audio Processing:\n");
    printf("  NoOp             - No-operation baseline\n");
    printf("  gain             - Simple gain/volume control\n");
    printf("  GainStats        - Gain with statistical analysis\n");
    printf("\nDigital Signal Processing:\n");
    printf("  IIRFilter        - Infinite Impulse Response filter\n");
    printf("  Conv1D           - 1D convolution\n");
    printf("  Conv1D_accel     - Accelerated 1D convolution\n");
    printf("  ModalFilterBank  - Modal synthesis filter bank\n");
    printf("  FFT1D            - 1D Fast Fourier Transform\n");
    printf("\nPhysical Modeling:\n");
    printf("  DWG1DNaive       - 1D Digital Waveguide (naive)\n");
    printf("  DWG1DAccel       - 1D Digital Waveguide (accelerated)\n");
    printf("  FDTD3D           - 3D Finite Difference Time Domain\n");
    printf("\nMemory Performance:\n");
    printf("  RndMemRead       - Random memory access pattern\n");
    printf("\nExamples:\n");
    printf("  gpubench --benchmark gain\n");
    printf("  gpubench --benchmark Conv1D_accel --irLength 4096 --nTracks 1024\n");
    printf("  gpubench --benchmark FDTD3D --fdtdGrid 128 --nRuns 5\n\n");
}

// --dawsim flags (metal-swift/MetalSwiftBench/main.swift:110-133)
static bool g_dawsim = false;
static BenchmarkUtils::DAWSimulationMode g_dawsim_mode = BenchmarkUtils::DAWSimulationMode::SPIN;
static double g_dawsim_jitter_us = 0.0;

void runSelectedBenchmark(std::unique_ptr<GPUABenchmark> benchmark, const std::string& name) {
    try {
        printf("Setting up %s benchmark...\n", name.c_str());
        benchmark->setupBenchmark();
        if (g_dawsim) {
            BenchmarkUtils::DAWSimulator sim;
            sim.bufferDuration = (double)BUFSIZE / (double)FS;
            sim.mode = g_dawsim_mode;
            sim.jitterSeconds = g_dawsim_jitter_us * 1e-6;
            benchmark->setDawSimulator(sim);
            printf("DAW simulation: %s, buffer slot %.3f ms, jitter +/-%.1f us\n",
                   g_dawsim_mode == BenchmarkUtils::DAWSimulationMode::SPIN ? "spin" : "sleep",
                   sim.bufferDuration * 1e3, g_dawsim_jitter_us);
        }
        printf("Running %s benchmark (%d iterations with %d warmup)...\n", name.c_str(), NRUNS, 3);
        auto result = benchmark->runBenchmark(NRUNS, 3);

        printf("Validating %s benchmark results...\n", name.c_str());
        GPUABenchmark::ValidationData validation;
        benchmark->validate(validation);
        if (validation.status != GPUABenchmark::ValidationStatus::SUCCESS)
            printf("Validation failed for %s:\n", name.c_str());
        else
            printf("Validation passed for %s\n", name.c_str());
        for (const auto& msg : validation.messages) printf("  %s\n", msg.c_str());

        if (JSON_OUTPUT) {
            writeJSONResults(result.latencies, name, OUTPUT_FILE);
        } else {
            benchmark->printResults(result);
            benchmark->writeResults(result);
            if (!OUTPUT_FILE.empty()) writeCSVResults(result.latencies, name, OUTPUT_FILE);
            if (!result.gpu_latencies.empty() && result.gpu_statistics.median > 0.0f) {
                const double bytes = static_cast<double>(benchmark->algorithmicBytes());
                const double gbs = bytes / (result.gpu_statistics.median * 1e-3) / 1e9;
                printf("Roofline: %.0f algorithmic bytes / %.3f ms device time = %.1f GB/s (%.1f%% of 8 TB/s)\n",
                       bytes, result.gpu_statistics.median, gbs, 100.0 * gbs / 8000.0);
            }
            if (g_dawsim)
                printf("DAW simulation: %llu slots, %llu missed (iteration longer than %.3f ms)\n",
                       result.daw_waits, result.daw_missed_slots, 1e3 * (double)BUFSIZE / (double)FS);
        }
        printf("%s benchmark completed successfully!\n", name.c_str());
    } catch (const std::exception& e) {
        printf("Benchmark %s failed: %s\n", name.c_str(), e.what());
    }
}

}  // namespace

int main(int argc, char** argv) {
    printf("GPGPU Audio Benchmark\n");
    std::string which = "RndMemRead";

    for (int i = 1; i < argc; i++) {
        const bool hasNext = i + 1 < argc;
        auto need = [&](const char* flag) {
            if (!hasNext) printf("Error: %s requires an argument\n", flag);
            return hasNext;
        };
        if (strcmp(argv[i], "--help") == 0) { printHelp(); return 0; }
        else if (strcmp(argv[i], "--list") == 0) { printBenchmarkList(); return 0; }
        else if (strcmp(argv[i], "--json") == 0) { JSON_OUTPUT = true; }
        else if (strcmp(argv[i], "--benchmark") == 0) { if (!need("--benchmark")) return 1; which = argv[++i]; }
        else if (strcmp(argv[i], "--fs") == 0) { if (!need("--fs")) return 1; FS = atoi(argv[++i]); }
        else if (strcmp(argv[i], "--bufferSize") == 0) {
            if (!need("--bufferSize")) return 1;
            BUFSIZE = atoi(argv[++i]);
            printf("Buffer size set to: %d\n", BUFSIZE);
        } else if (strcmp(argv[i], "--nTracks") == 0) {
            if (!need("--nTracks")) return 1;
            NTRACKS = atoi(argv[++i]);
            printf("Number of tracks set to: %d\n", NTRACKS);
        } else if (strcmp(argv[i], "--nRuns") == 0) { if (!need("--nRuns")) return 1; NRUNS = atoi(argv[++i]); }
        else if (strcmp(argv[i], "--outputfile") == 0) {
            if (!need("--outputfile")) return 1;
            OUTPUT_FILE = argv[++i];
            printf("Output file set to: %s\n", OUTPUT_FILE.c_str());
        } else if (strcmp(argv[i], "--irLength") == 0) { if (!need("--irLength")) return 1; IR_LENGTH = atoi(argv[++i]); }
        else if (strcmp(argv[i], "--fdtdGrid") == 0) { if (!need("--fdtdGrid")) return 1; FDTD_GRID = atoi(argv[++i]); }
        else if (strcmp(argv[i], "--convMode") == 0) {
            if (!need("--convMode")) return 1;
            CONV_STREAMING = strcmp(argv[++i], "stateless") == 0 ? 0 : 1;
        } else if (strcmp(argv[i], "--modalMode") == 0) {
            if (!need("--modalMode")) return 1;
            MODAL_REAL = strcmp(argv[++i], "bank") == 0 ? 1 : 0;
        } else if (strcmp(argv[i], "--dawsim") == 0) {
            g_dawsim = true;
        } else if (strcmp(argv[i], "--dawsim-mode") == 0) {
            if (!need("--dawsim-mode")) return 1;
            const char* m = argv[++i];
            if (strcmp(m, "spin") == 0) g_dawsim_mode = BenchmarkUtils::DAWSimulationMode::SPIN;
            else if (strcmp(m, "sleep") == 0) g_dawsim_mode = BenchmarkUtils::DAWSimulationMode::SLEEP;
            else { printf("Error: Unknown DAW simulation mode '%s'. Expected spin | sleep.\n", m); return 1; }
        } else if (strcmp(argv[i], "--dawsim-jitter-us") == 0) {
            if (!need("--dawsim-jitter-us")) return 1;
            g_dawsim_jitter_us = atof(argv[++i]);
            if (g_dawsim_jitter_us < 0.0) { printf("Error: --dawsim-jitter-us must be >= 0\n"); return 1; }
        } else {
            printf("Warning: Unparsed argument: %s\n", argv[i]);
        }
    }
    if (BUFSIZE <= 0 || NTRACKS <= 0 || NRUNS <= 0 || FS <= 0) {
        printf("Error: --fs, --bufferSize, --nTracks and --nRuns must be positive\n");
        return 1;
    }

    int deviceCount = 0;
    hipError_t err = hipGetDeviceCount(&deviceCount);
    if (err != hipSuccess) {
        printf("Failed to get HIP device count: %s\n", hipGetErrorString(err));
        return 1;
    }
    printf("Found %d HIP device(s)\n", deviceCount);

    std::unique_ptr<GPUABenchmark> instance;
    try {
        instance = gab::createBenchmark(which);
    } catch (const std::exception& e) {
        printf("Benchmark %s failed: %s\n", which.c_str(), e.what());
        return 1;
    }
    if (instance) {
        printf("Running %s benchmark...\n", which.c_str());
        runSelectedBenchmark(std::move(instance), which);
        printf("Done\n");
        return 0;
    }
    printf("Error: Unknown benchmark '%s'\n", which.c_str());
    printf("Use --list to see available benchmarks.\n");
    return 1;
}
