// benchmark_constants.hpp — the suite's numeric parameters as ONE table.
//
// Every entry keeps the name and value it has in the reference (cuda/benchmark_constants.cuh:4-37),
// so code written against `BenchmarkConstants::NAME` compiles unchanged; the table form adds what
// the reference's flat list cannot: the entries can be enumerated (BenchmarkConstants::kTable, used
// by the host-logic tests and by tooling that dumps a run's parameterisation).
//
//   X(name, value, where the reference uses it)
#pragma once

#include <cstddef>

#define GAB_BENCHMARK_CONSTANT_TABLE(X)                                                            \
    X(GAIN_VALUE,                        2.0f,    "bench_gain.cu:56")                              \
    X(GAINSTATS_GAIN,                    0.5f,    "bench_gainstats.cu:20")                         \
    X(FDTD3D_SOURCE_SCALE,               0.1f,    "bench_fdtd3d.cu inject")                        \
    X(FDTD3D_OUTPUT_SCALE,               0.1f,    "bench_fdtd3d.cu extract")                       \
    X(FDTD3D_CPU_REF_FREQ,               0.01f,   "bench_fdtd3d.cu:445-459 placeholder golden")    \
    X(WAVEGUIDE_MIX_FACTOR,              0.5f,    "bench_dwg.cu output mix")                       \
    X(WAVEGUIDE_GAIN_MIN,                0.1f,    "bench_dwg.cu:325-348")                          \
    X(WAVEGUIDE_GAIN_RANGE,              0.9f,    "bench_dwg.cu:325-348")                          \
    X(WAVEGUIDE_REFLECTION_PERTURBATION, 0.01f,   "bench_dwg.cu:325-348")                          \
    X(WAVEGUIDE_DAMPING_PERTURBATION,    0.0001f, "bench_dwg.cu:325-348")                          \
    X(HAMMING_WINDOW_A0,                 0.54f,   "bench_conv1d.cu:166-176")                       \
    X(HAMMING_WINDOW_A1,                 0.46f,   "bench_conv1d.cu:166-176")                       \
    X(CONV1D_IR_BASE_FREQ,               0.1f,    "bench_conv1d.cu:166-176")                       \
    X(CONV1D_IR_FREQ_RANGE,              0.05f,   "bench_conv1d.cu:166-176")                       \
    X(DATATRANSFER_SIGNAL_OFFSET,        0.5f,    "bench_datatransfer.cu:139-147")                 \
    X(DATATRANSFER_SIGNAL_AMPLITUDE,     0.5f,    "bench_datatransfer.cu:139-147")                 \
    X(DATATRANSFER_SIGNAL_FREQ,          0.001f,  "bench_datatransfer.cu:139-147")                 \
    X(RANDOM_SIGNAL_SCALE,               2.0f,    "rand()/RAND_MAX * 2 - 1 inputs")                \
    X(MODAL_STATE_INIT_REAL,             0.5f,    "bench_modal.cu:5-13")                           \
    X(MODAL_STATE_INIT_IMAG,             0.5f,    "bench_modal.cu:5-13")

namespace BenchmarkConstants {

#define GAB_DEFINE_CONSTANT(name, value, where) constexpr float name = value;
GAB_BENCHMARK_CONSTANT_TABLE(GAB_DEFINE_CONSTANT)
#undef GAB_DEFINE_CONSTANT

struct Entry {
    const char* name;
    float value;
    const char* used_by;
};

#define GAB_TABLE_ROW(name, value, where) {#name, value, where},
constexpr Entry kTable[] = {GAB_BENCHMARK_CONSTANT_TABLE(GAB_TABLE_ROW)};
#undef GAB_TABLE_ROW
constexpr std::size_t kTableSize = sizeof(kTable) / sizeof(kTable[0]);

}  // namespace BenchmarkConstants
