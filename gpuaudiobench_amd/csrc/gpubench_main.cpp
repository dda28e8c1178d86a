// gpubench_main.cpp — command-line driver, flag-compatible with the reference's
// cuda/main.cu:236-328 (--help --list --json --benchmark --fs --bufferSize
// --nTracks --nRuns --outputfile; default benchmark RndMemRead; exit code 0/1),
// plus the flags the BASELINE configs need: --irLength --fdtdGrid --fdtdSteps
// --convMode --gpus --validate-only --cpu-threads.  With --json the reference's object gains a
// "roofline", a "cpu_golden" and (with --gpus) a "multi_gpu" member.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "gab/benchmarks.hpp"
#include "gab/multi_gpu.hpp"

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
    printf("  --fdtdSteps [n]     FDTD3D: leapfrog steps per iteration (3 per sample: sets the buffer to ceil(n/3) samples)\n");
    printf("  --fdtdForm [f]      FDTD3D: auto (the room resident in LDS where it fits, one launch per buffer; default) |\n");
    printf("                      step (one launch per step: for a device shared with other work)\n");
    printf("  --datacopyMode [m]  datacopy*: overlap (ONE engine upload beside a kernel that writes the pinned output as the input\n");
    printf("                      lands: both link directions at once; default) | sequential (H2D -> kernel -> D2H, as the reference)\n");
    printf("  --convMode [m]      Conv1D_accel: stream (carried history, default) | stateless | roundtrip (stream, with the\n");
    printf("                      iteration's upload, kernel and download overlapped in one call: gab_conv_round_trip)\n");
    printf("  --convBatch [n]     Conv1D_accel: an iteration is ONE launch over n HBM-resident buffers (throughput mode,\n");
    printf("                      no per-iteration copies); default: one buffer per iteration with its copies\n");
    printf("  --gpus [n]          Run on n devices, one host thread each: gain, GainStats, IIRFilter, FFT1D, RndMemRead, Conv1D\n");
    printf("                      and Conv1D_accel as contiguous channel shards of --nTracks (Conv1D_accel's impulse-response\n");
    printf("                      bank broadcast once over RCCL; Conv1D with its halo rows; RndMemRead with the pool on every\n");
    printf("                      device); DWG, modal, FDTD3D, datacopy, NoOp as replicas\n");
    printf("  --print-shards      Print the channel shards --gpus / --nTracks give and exit (no device is touched)\n");
    printf("  --validate-only     Set up, validate against the CPU golden, exit 0/1; no timed loop\n");
    printf("  --cpu-threads [n]   Threads for the timed CPU golden (default: all hardware threads; 0 = skip it)\n");
    printf("  --modalMode [m]     ModalFilterBank: placeholder (the CUDA port, default) | bank (real phasor bank)\n");
    printf("  --dawsim            Pace iterations to one buffer slot each (bufferSize / fs)\n");
    printf("  --dawsim-mode [m]   spin | sleep (default: spin)\n");
    printf("  --dawsim-jitter-us [us]  Uniform jitter on each slot (default: 0)\n");
    printf("  --keepWarm          Leave eight idle waves on the device between iterations (with --dawsim: the device\n");
    printf("                      does not go idle while the loop waits for the next slot)\n");
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
static bool g_keep_warm = false;
static bool g_validate_only = false;
static int g_gpus = 0;              // 0: the reference's single-device path, no RCCL
static bool g_validation_failed = false;
static bool g_skip_cpu_golden = false;   // --cpu-threads 0

// "roofline": {...} and "cpu_golden": {...} members for --json
std::string extraMembers(GPUABenchmark& b, const GPUABenchmark::BenchmarkResult* result,
                         const GPUABenchmark::ValidationData& validation) {
    char buf[640];
    std::string j;
    const double bytes = static_cast<double>(b.algorithmicBytes());
    // a kernel that keeps its fields in LDS moves none of the algorithmic bytes through HBM: say so, frac may pass 1
    const char* bound = b.workingSetOnChip() ? "lds-resident (hbm figures for comparison only)" : "hbm";
    const double dev_ms = (result && !result->gpu_latencies.empty()) ? result->gpu_statistics.median : 0.0;
    if (dev_ms > 0.0) {
        const double gbs = bytes / (dev_ms * 1e-3) / 1e9;
        snprintf(buf, sizeof buf,
                 "  \"roofline\": {\n    \"bound\": \"%s\",\n    \"algorithmic_bytes\": %.0f,\n    \"device_median_ms\": %.6f,\n"
                 "    \"achieved_GBps\": %.1f,\n    \"peak_GBps\": 8000.0,\n    \"frac\": %.4f\n  },\n", bound, bytes, dev_ms, gbs, gbs / 8000.0);
    } else {
        snprintf(buf, sizeof buf, "  \"roofline\": {\n    \"bound\": \"%s\",\n    \"algorithmic_bytes\": %.0f,\n"
                                  "    \"device_median_ms\": null\n  },\n", bound, bytes);
    }
    j += buf;
    int used = 0;
    const int want = CPU_THREADS > 0 ? CPU_THREADS : (int)std::max(1u, std::thread::hardware_concurrency());
    const double cpu_ms = g_skip_cpu_golden ? -1.0 : b.timeCpuGolden(want, &used);
    if (cpu_ms >= 0.0)
        snprintf(buf, sizeof buf, "  \"cpu_golden\": {\n    \"ms\": %.4f,\n    \"threads\": %d,\n    \"speedup_vs_device\": %s\n  },\n",
                 cpu_ms, used, dev_ms > 0.0 ? std::to_string(cpu_ms / dev_ms).c_str() : "null");
    else
        snprintf(buf, sizeof buf, "  \"cpu_golden\": null,\n");
    j += buf;
    snprintf(buf, sizeof buf, "  \"validation\": {\n    \"passed\": %s,\n    \"max_error\": %.6g\n  }",
             validation.status == GPUABenchmark::ValidationStatus::SUCCESS ? "true" : "false", validation.max_error);
    j += buf;
    return j;
}

void runSelectedBenchmark(std::unique_ptr<GPUABenchmark> benchmark, const std::string& name) {
    try {
        printf("Setting up %s benchmark...\n", name.c_str());
        benchmark->setupBenchmark();
        benchmark->setKeepWarm(g_keep_warm);
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
        GPUABenchmark::BenchmarkResult result;
        if (!g_validate_only) {
            printf("Running %s benchmark (%d iterations with %d warmup)...\n", name.c_str(), NRUNS, 3);
            result = benchmark->runBenchmark(NRUNS, 3);
        }

        printf("Validating %s benchmark results...\n", name.c_str());
        GPUABenchmark::ValidationData validation;
        benchmark->validate(validation);
        if (validation.status != GPUABenchmark::ValidationStatus::SUCCESS) {
            printf("Validation failed for %s:\n", name.c_str());
            g_validation_failed = true;
        } else {
            printf("Validation passed for %s\n", name.c_str());
        }
        for (const auto& msg : validation.messages) printf("  %s\n", msg.c_str());

        if (JSON_OUTPUT) {
            const std::string j = generateJSONResultsWith(result.latencies, name,
                                                          extraMembers(*benchmark, g_validate_only ? nullptr : &result, validation));
            if (OUTPUT_FILE.empty()) {
                printf("%s\n", j.c_str());
            } else {
                FILE* f = fopen(OUTPUT_FILE.c_str(), "w");
                if (f) { fputs(j.c_str(), f); fclose(f); printf("JSON results saved to: %s\n", OUTPUT_FILE.c_str()); }
            }
        } else if (g_validate_only) {
            // nothing was timed
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

// --gpus N: one host thread per device (include/gab/multi_gpu.hpp)
int runOnSeveralDevices(const std::string& name) {
    try {
        gab::MultiGpuConfig cfg;
        cfg.benchmark = name;
        cfg.gpus = g_gpus;
        cfg.iterations = NRUNS;
        cfg.warmup = 3;
        cfg.validate_only = g_validate_only;
        const gab::MultiGpuReport rep = gab::runOnDevices(cfg);
        printf("%s on %d device(s): %s; %zu tracks in all\n", name.c_str(), rep.gpus,
               rep.partition.c_str(), rep.total_tracks);
        if (rep.ir_broadcast_ms >= 0.0)
            printf("Impulse-response bank: %zu bytes broadcast with RCCL in %.3f ms (%.1f GB/s)\n", rep.ir_bank_bytes,
                   rep.ir_broadcast_ms, rep.ir_bank_bytes / (rep.ir_broadcast_ms * 1e-3) / 1e9);
        for (const auto& r : rep.ranks) {
            printf("  device %d: tracks [%zu, %zu)  median %.4f ms  device %.4f ms  validation %s (max error %.3g)%s%s\n", r.device,
                   r.tracks.lo, r.tracks.hi, r.result.latencies.empty() ? 0.0 : r.result.statistics.median,
                   r.result.gpu_latencies.empty() ? 0.0 : r.result.gpu_statistics.median,
                   r.validation.status == GPUABenchmark::ValidationStatus::SUCCESS ? "passed" : "FAILED", r.validation.max_error,
                   r.error.empty() ? "" : "  error: ", r.error.c_str());
            for (const auto& msg : r.validation.messages) printf("    %s\n", msg.c_str());
        }
        if (rep.job_median_ms > 0.0)
            printf("Whole job: %.4f ms per buffer of all tracks (slowest rank), %.0f track-buffers/s\n", rep.job_median_ms,
                   rep.tracks_per_second);
        if (JSON_OUTPUT) {
            const std::vector<float> none;
            const std::string j = generateJSONResultsWith(rep.ranks.empty() ? none : rep.ranks[0].result.latencies, name,
                                                          gab::multiGpuJson(rep));
            if (OUTPUT_FILE.empty()) printf("%s\n", j.c_str());
            else if (FILE* f = fopen(OUTPUT_FILE.c_str(), "w")) { fputs(j.c_str(), f); fclose(f); }
        }
        return rep.ok() ? 0 : 1;
    } catch (const std::exception& e) {
        printf("Benchmark %s failed: %s\n", name.c_str(), e.what());
        return 1;
    }
}

}  // namespace

int main(int argc, char** argv) {
    printf("GPGPU Audio Benchmark\n");
    std::string which = "RndMemRead";
    bool print_shards = false;

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
        else if (strcmp(argv[i], "--fdtdSteps") == 0) { if (!need("--fdtdSteps")) return 1; FDTD_STEPS = atoi(argv[++i]); }
        else if (strcmp(argv[i], "--fdtdForm") == 0) {
            if (!need("--fdtdForm")) return 1;
            const char* m = argv[++i];
            if (strcmp(m, "auto") == 0) FDTD_FORM = 0;
            else if (strcmp(m, "step") == 0) FDTD_FORM = 1;
            else { printf("Error: --fdtdForm takes auto or step\n"); return 1; }
        }
        else if (strcmp(argv[i], "--datacopyMode") == 0) {
            if (!need("--datacopyMode")) return 1;
            const char* m = argv[++i];
            if (strcmp(m, "overlap") == 0) DATACOPY_SEQUENTIAL = 0;
            else if (strcmp(m, "sequential") == 0) DATACOPY_SEQUENTIAL = 1;
            else { printf("Error: --datacopyMode takes overlap or sequential\n"); return 1; }
        }
        else if (strcmp(argv[i], "--gpus") == 0) {
            if (!need("--gpus")) return 1;
            g_gpus = atoi(argv[++i]);
            if (g_gpus < 1) { printf("Error: --gpus must be >= 1\n"); return 1; }
        }
        else if (strcmp(argv[i], "--print-shards") == 0) { print_shards = true; }
        else if (strcmp(argv[i], "--validate-only") == 0) { g_validate_only = true; }
        else if (strcmp(argv[i], "--cpu-threads") == 0) {
            if (!need("--cpu-threads")) return 1;
            CPU_THREADS = atoi(argv[++i]);
            if (CPU_THREADS < 0) { printf("Error: --cpu-threads must be >= 0\n"); return 1; }
            g_skip_cpu_golden = CPU_THREADS == 0;
        }
        else if (strcmp(argv[i], "--convBatch") == 0) {
            if (!need("--convBatch")) return 1;
            CONV_BATCH = atoi(argv[++i]);
            if (CONV_BATCH < 1) { printf("Error: --convBatch must be >= 1\n"); return 1; }
        }
        else if (strcmp(argv[i], "--convMode") == 0) {
            if (!need("--convMode")) return 1;
            const char* m = argv[++i];
            CONV_STREAMING = strcmp(m, "stateless") == 0 ? 0 : (strcmp(m, "roundtrip") == 0 ? 2 : 1);
        } else if (strcmp(argv[i], "--modalMode") == 0) {
            if (!need("--modalMode")) return 1;
            MODAL_REAL = strcmp(argv[++i], "bank") == 0 ? 1 : 0;
        } else if (strcmp(argv[i], "--dawsim") == 0) {
            g_dawsim = true;
        } else if (strcmp(argv[i], "--keepWarm") == 0) {
            g_keep_warm = true;
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

    if (FDTD_STEPS > 0 && which == "FDTD3D") {
        // the reference advances 3 leapfrog steps per audio sample (bench_fdtd3d.cuh:41)
        BUFSIZE = (FDTD_STEPS + 2) / 3;
        printf("FDTD3D: %d steps asked for -> %d samples x 3 steps = %d steps per iteration\n", FDTD_STEPS, BUFSIZE, 3 * BUFSIZE);
    }
    if (print_shards) {
        const int world = g_gpus > 0 ? g_gpus : 1;
        if ((size_t)world > (size_t)NTRACKS) { printf("Error: more GPUs than tracks\n"); return 1; }
        if (!gab::benchmarkShards(which)) {
            printf("%s: replicas only (its tracks reduce into shared outputs): %d x %d tracks\n", which.c_str(), world, NTRACKS);
            return 0;
        }
        const int L = IR_LENGTH > 0 ? IR_LENGTH : Conv1DBenchmark::DEFAULT_IR_LEN;
        for (int r = 0; r < world; ++r) {
            const gab::ShardRange s = gab::shardRange(r, world, (size_t)NTRACKS, gab::shardGranule(which));
            if (which == "Conv1D") {          // its golden convolves the flat input: the preceding tracks' rows come along
                const size_t halo = std::min(s.lo, ((size_t)L - 1 + (size_t)BUFSIZE - 1) / (size_t)BUFSIZE);
                printf("shard %d: tracks [%zu, %zu) = %zu, input rows from track %zu (halo %zu)\n", r, s.lo, s.hi, s.count(),
                       s.lo - halo, halo);
            } else {
                printf("shard %d: tracks [%zu, %zu) = %zu\n", r, s.lo, s.hi, s.count());
            }
        }
        return 0;
    }

    int deviceCount = 0;
    hipError_t err = hipGetDeviceCount(&deviceCount);
    if (err != hipSuccess) {
        printf("Failed to get HIP device count: %s\n", hipGetErrorString(err));
        return 1;
    }
    printf("Found %d HIP device(s)\n", deviceCount);

    if (g_gpus > 0) {
        printf("Running %s benchmark on %d device(s)...\n", which.c_str(), g_gpus);
        const int rc = runOnSeveralDevices(which);
        printf("Done\n");
        return rc;
    }

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
        return (g_validate_only && g_validation_failed) ? 1 : 0;
    }
    printf("Error: Unknown benchmark '%s'\n", which.c_str());
    printf("Use --list to see available benchmarks.\n");
    return 1;
}
