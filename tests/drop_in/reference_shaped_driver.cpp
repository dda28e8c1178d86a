// A driver written the way code against the REFERENCE is written: the reference's header names
// (cuda/main.cu:53,59-70), its factory-table idiom over the benchmark classes (cuda/main.cu:77-100:
// default constructors, DataTransferBenchmark(in, out), DWGBenchmark::Variant) and its run sequence
// (cuda/main.cu:117-164: setupBenchmark / runBenchmark / validate / writeJSONResults, the globals
// NTRACKS / NRUNS / JSON_OUTPUT / OUTPUT_FILE).  tests/test_drop_in_headers.py compiles it with hipcc
// against include/ and links it with libgab_hip.so: a maintainer's translation unit needs no edit
// beyond dropping <cuda_runtime.h> / <helper_cuda.h>.
#include <algorithm>
#include <functional>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "globals.cuh"

#include "bench_noop.cuh"
#include "bench_gain.cuh"
#include "bench_gainstats.cuh"
#include "bench_datatransfer.cuh"
#include "bench_fft.cuh"
#include "bench_iir.cuh"
#include "bench_conv1d.cuh"
#include "bench_conv1d_accel.cuh"
#include "bench_modal.cuh"
#include "bench_dwg.cuh"
#include "bench_fdtd3d.cuh"
#include "bench_rndmem.cuh"
#include "bench_base.cuh"
#include "bench_utils.cuh"
#include "benchmark_constants.cuh"
#include "thread_config.cuh"

using Factory = std::function<std::unique_ptr<GPUABenchmark>()>;

static const std::vector<std::pair<const char*, Factory>>& table() {
    static const std::vector<std::pair<const char*, Factory>> t = {
        {"NoOp", [] { return std::make_unique<NoOpBenchmark>(); }},
        {"gain", [] { return std::make_unique<GainBenchmark>(); }},
        {"GainStats", [] { return std::make_unique<GainStatsBenchmark>(); }},
        {"datacopy2080", [] { return std::make_unique<DataTransferBenchmark>(0.20f, 0.80f); }},
        {"FFT1D", [] { return std::make_unique<FFTBenchmark>(); }},
        {"IIRFilter", [] { return std::make_unique<IIRBenchmark>(); }},
        {"Conv1D", [] { return std::make_unique<Conv1DBenchmark>(); }},
        {"Conv1D_accel", [] { return std::make_unique<Conv1DAccelBenchmark>(); }},
        {"ModalFilterBank", [] { return std::make_unique<ModalBenchmark>(); }},
        {"DWG1DNaive", [] { return std::make_unique<DWGBenchmark>(DWGBenchmark::Variant::NAIVE); }},
        {"DWG1DAccel", [] { return std::make_unique<DWGBenchmark>(DWGBenchmark::Variant::ACCELERATED); }},
        {"FDTD3D", [] { return std::make_unique<FDTD3DBenchmark>(); }},
        {"RndMemRead", [] { return std::make_unique<RndMemBenchmark>(); }},
    };
    return t;
}

int main(int argc, char** argv) {
    const std::string want = argc > 1 ? argv[1] : "--list";
    if (want == "--list") {
        for (const auto& e : table()) cout << e.first << endl;      // cout / endl: globals.cuh's using-declarations
        cout << "FS=" << FS << " BUFSIZE=" << BUFSIZE << " NTRACKS=" << NTRACKS << " NRUNS=" << NRUNS
             << " numElements=" << numElements << endl;
        return 0;
    }
    const auto it = std::find_if(table().begin(), table().end(),
                                 [&](const auto& e) { return want == e.first; });
    if (it == table().end()) return 2;
    NTRACKS = 64;
    NRUNS = 5;
    JSON_OUTPUT = true;
    std::unique_ptr<GPUABenchmark> benchmark = it->second();
    // the sequence of cuda/main.cu:124-153
    benchmark->setupBenchmark();
    auto result = benchmark->runBenchmark(NRUNS, 3);
    GPUABenchmark::ValidationData validation;
    benchmark->validate(validation);
    if (validation.status != GPUABenchmark::ValidationStatus::SUCCESS) {
        for (const auto& msg : validation.messages) printf("  %s\n", msg.c_str());
        return 1;
    }
    if (JSON_OUTPUT) {
        writeJSONResults(result.latencies, want, OUTPUT_FILE);
    } else {
        benchmark->printResults(result);
        benchmark->writeResults(result);
        if (!OUTPUT_FILE.empty()) writeCSVResults(result.latencies, want, OUTPUT_FILE);
    }
    return 0;
}
