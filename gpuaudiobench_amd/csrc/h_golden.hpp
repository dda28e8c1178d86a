// h_golden.hpp — the CPU golden functions the harness validates against.
//
// VALIDATION ONLY.  These are the host-side checks the reference runs inside
// each benchmark's setupBenchmark()/validate() (its `calculateCPUReference`
// members).  No device output is ever produced, patched or replaced by them:
// benchmarks call them to fill `cpu_reference` buffers and to time the CPU
// baseline, nothing else.  They are independent of oracle/ (which is test
// infrastructure and never linked into this library).
#pragma once

#include <cstddef>

#include "gab_c_api.h"

struct IIRCoefficients;
struct WaveguideState;
struct DWGParams;

namespace gab {
namespace golden {

void gain(const float* in, float* out, size_t n, float g);                       // bench_gain.cu:85-95
void gainstats(const float* in, float* out, float* stats, size_t T, size_t B);   // bench_gainstats.cu:120-144
void datatransfer(const float* in, float* out, int in_size, int out_size);       // bench_datatransfer.cu:139-147
void dft1024(const float* in, float* re, float* im, size_t tracks);              // bench_fft.cu:149-168
void dft1024_f64(const float* in, double* re, double* im, size_t tracks);        // truth for the FFT gate
void iir(const float* in, float* out, const IIRCoefficients* c, float* state, int T, int B);  // bench_iir.cu:170-197
void conv1d(const float* in, const float* ir, float* out, int L, int B, int T);  // bench_conv1d.cu:188-208
void conv_accel(const float* in, const float* ir, float* out, int L, int B, int T);  // bench_conv1d_accel.cu:234-252
// the same loops for tracks [t_lo, t_hi) only (the CPU baseline cuts the tracks over threads)
void conv1d_rows(const float* in, const float* ir, float* out, int L, int B, int t_lo, int t_hi, int T);
void conv1d_shard_rows(const float* in_with_halo, const float* ir, float* out, int L, int B, int t_lo, int t_hi, int T,
                       int halo);
void conv_accel_rows(const float* in, const float* ir, float* out, int L, int B, int t_lo, int t_hi, int T);
void modal(const float* params, float* out, int n_modes, int B, int out_tracks); // bench_modal.cu:152-179
// the real bank: metal-swift/MetalSwiftBench/Benchmarks/ModalFilterBankBenchmark.swift:73-101
void modal_bank(const float* params, float* out, int n_modes, int B, int out_tracks);
void dwg(const WaveguideState* wg, float* fwd, float* bwd, const float* in, float* out,
         const DWGParams* p);                                                      // bench_dwg.cu:356-399
void fdtd_placeholder(const float* in, float* out, size_t T, size_t B);          // bench_fdtd3d.cu:445-459
// the four FDTD kernels (bench_fdtd3d.cu:14-139) run on the host, grids caller-owned
void fdtd3d(const gab_fdtd_params& P, float* p, float* vx, float* vy, float* vz, const float* in,
            float* out, int T, int B, int first_sample, int n_samples);
void rndmem(const float* pool, const int* playheads, float* out, int B, int T);  // bench_rndmem.cu:194-205

}  // namespace golden
}  // namespace gab
