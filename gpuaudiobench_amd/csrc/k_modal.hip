// k_modal.hip — the real modal filter bank.
//
// The reference's CUDA port is a placeholder (cuda/bench_modal.cu:15-36: 32 constants);
// the bank itself is its Metal kernel BenchmarkModalFilterBank
// (metal-swift/MetalSwiftBench/Metal/kernels_benchmark_staging.metal:121-162) with the
// golden of Benchmarks/ModalFilterBankBenchmark.swift:73-101: mode m is a complex phasor
// (re, im) turned by 2*pi*freq every sample, out[(m % tracks)*B + i] += amp * re_i.
// There: one thread per mode, 512 float atomics each, into 32 rows.  Here:
//   * a lane keeps J modes of ONE output track in registers and walks the buffer in
//     16-sample chunks, summing its modes' contributions into 16 accumulators;
//   * lanes of a wave that serve the same track are folded with shuffles, the waves of
//     a workgroup through LDS, workgroups through a partials array and a second
//     kernel — every sum in a fixed order, no atomics, so a run is reproducible;
//   * the phasor recurrence is the golden's own (four products, one difference, one
//     sum, unfused), three packed instructions per sample, so each mode's sequence is
//     bit-identical to the oracle's and only the summation order differs.
// VALU-bound: 2^20 modes x 512 samples x 5 instructions (9 flops, none of them fusable).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "gab_common.hpp"
#include "gab_fft.hpp"

namespace gab {
namespace {

using fft::cf;
using fft::mk;

constexpr int kMbThreads = 512;
constexpr int kMbWaves = kMbThreads / 64;
constexpr int kMbChunk = 16;

// (st.x * cs.x, st.y * cs.x)  and  (st.x * cs.y, st.y * cs.y)
__device__ __forceinline__ cf mul_by_lo(cf st, cf cs) {
    cf r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(st), "v"(cs));
    return r;
}
__device__ __forceinline__ cf mul_by_hi(cf st, cf cs) {
    cf r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(st), "v"(cs));
    return r;
}

// Layout of the launch:
//   slots = 64 / tracks lanes of a wave serve each track (lane = slot*tracks + track);
//   a "row" is `tracks` consecutive modes; wave w of workgroup g owns rows
//   [(g*waves + w) * slots * J, +slots*J): lane (slot, track) takes row
//   first + j*slots + slot, j < J, i.e. mode (row * tracks + track).
// TRACKS > 0 fixes the output-track count at compile time (32 is the reference's shape: every
// loop over it unrolls and the wave fold is one cross-half add); TRACKS == 0 takes it at run time.
// (Measured and not kept: two modes per packed register (RE, IM, C, S, AMP pairs), 4.5 instead
// of 5 instructions per mode-sample but half as many lanes busy: 144 vs 125 us at 2^20 modes —
// with two or more waves per SIMD unpacked fp32 already issues at the packed rate.  Round 4: sample-outer / mode-inner
// order, so that no instruction waits for the one before it: 112 against 113 us, the same.  What the 5-instruction
// floor (71 us at 2.4 GHz: tools/kernel_table.py) does not hold: the double-precision sincos per mode (~12 %), the folds (~20 %), the clock
// under packed-fp32 load (~2.2 GHz).)
// (Run-time track counts, TRACKS == 0: three waves per SIMD and at most four modes per lane — the fold over a track count that is
// not a power of two holds three chunks of samples at once; under four waves' 128 registers four modes per lane spilled 16 bytes
// per lane and eight 612, and eight still spill with 256 registers: modal_launch stops at four there.)
template <int J, int TRACKS>
__global__ __launch_bounds__(kMbThreads, TRACKS > 0 ? 4 : 2) void modal_bank_kernel(const float* __restrict__ params,
                                                               float* __restrict__ partial,
                                                               int n_modes, int tracks_rt, int B) {
    const int tracks = TRACKS > 0 ? TRACKS : tracks_rt;
    // [parity][wave][i * (tracks + 1) + track]: lanes write consecutive words; the fold below
    // reads it transposed (16 consecutive samples of a track per 16 threads) two-way at worst
    __shared__ float fold[2][kMbWaves][65 * kMbChunk];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int waves = kMbWaves;                  // every launch is kMbThreads wide (modal_launch)
    const int slots = 64 / tracks;
    const int track = lane % tracks, slot = lane / tracks;
    const bool serving = slot < slots;
    const long first_row = ((long)blockIdx.x * waves + w) * slots * J;

    cf st[J], cs[J];
    float amp[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const long m = (first_row + (long)j * slots + slot) * tracks + track;
        if (serving && m < n_modes) {
            const float4 a = *reinterpret_cast<const float4*>(params + m * 8);        // amp, freq, -, re
            const float im = params[m * 8 + 4];
            const float ang = 2.0f * 3.14159265358979323846f * a.y;
            double sn, cn;
            sincos((double)ang, &sn, &cn);
            cs[j] = mk((float)cn, (float)sn);
            st[j] = mk(a.w, im);
            amp[j] = a.x;
        } else {
            cs[j] = mk(1.0f, 0.0f);
            st[j] = mk(0.0f, 0.0f);
            amp[j] = 0.0f;
        }
    }

    float* const mine = partial + (size_t)blockIdx.x * tracks * B;
    const int n_chunks = (B + kMbChunk - 1) / kMbChunk;
    for (int c = 0; c < n_chunks; ++c) {
        float acc[kMbChunk];
#pragma unroll
        for (int i = 0; i < kMbChunk; ++i) acc[i] = 0.0f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
#pragma unroll
            for (int i = 0; i < kMbChunk; ++i) {
                // (re*c, im*c), (re*s, im*s) -> (re*c - im*s, im*c + re*s): each rounded on its own
                st[j] = fft::addpi(mul_by_lo(st[j], cs[j]), mul_by_hi(st[j], cs[j]));
                acc[i] = __fadd_rn(acc[i], __fmul_rn(amp[j], st[j].x));
            }
        }
        // lanes serving the same track: slot 0 collects slots 1.. in order
        if (slots > 1) {
            if ((tracks & (tracks - 1)) == 0) {
                for (int off = 32; off >= tracks; off >>= 1) {
                    float other[kMbChunk];
#pragma unroll
                    for (int i = 0; i < kMbChunk; ++i) other[i] = __shfl_xor(acc[i], off);     // all in flight
#pragma unroll
                    for (int i = 0; i < kMbChunk; ++i) acc[i] = __fadd_rn(acc[i], other[i]);
                }
            } else {
                float sum[kMbChunk];
#pragma unroll
                for (int i = 0; i < kMbChunk; ++i) sum[i] = acc[i];
                for (int k = 1; k < slots; ++k) {
                    float other[kMbChunk];
#pragma unroll
                    for (int i = 0; i < kMbChunk; ++i) other[i] = __shfl(acc[i], lane + k * tracks);   // lanes >= 64 wrap; unused there
#pragma unroll
                    for (int i = 0; i < kMbChunk; ++i) sum[i] = __fadd_rn(sum[i], other[i]);
                }
#pragma unroll
                for (int i = 0; i < kMbChunk; ++i) acc[i] = sum[i];
            }
        }
        float* const f = fold[c & 1][w];
        if (lane < tracks) {
#pragma unroll
            for (int i = 0; i < kMbChunk; ++i) f[i * (tracks + 1) + lane] = acc[i];
        }
        // One barrier per chunk (the next chunk writes the other parity).  LDS-only: a full
        // __syncthreads() also waits for the previous chunk's global stores (vmcnt).
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int e = tid; e < tracks * kMbChunk; e += kMbThreads) {
            const int t = e / kMbChunk, ii = e % kMbChunk;
            const int at = ii * (tracks + 1) + t;
            float part[kMbWaves];
#pragma unroll
            for (int ww = 0; ww < kMbWaves; ++ww) part[ww] = fold[c & 1][ww][at];      // all in flight, then wave order
            float sum = part[0];
#pragma unroll
            for (int ww = 1; ww < kMbWaves; ++ww) sum = __fadd_rn(sum, part[ww]);
            const int i = c * kMbChunk + ii;
            if (i < B) mine[(size_t)t * B + i] = sum;
        }
    }
}

// out[e] = sum over workgroups of partial[g][e], in a fixed order: a block owns 64
// consecutive outputs, its 16 waves each sum a contiguous range of g (loads independent,
// eight in flight), wave 0 adds the 16 range sums in range order.
constexpr int kMrWaves = 16;
__global__ __launch_bounds__(64 * kMrWaves) void modal_bank_reduce_kernel(const float* __restrict__ partial,
                                                                         float* __restrict__ out,
                                                                         int n_partials, size_t n_out) {
    __shared__ float part[kMrWaves][64];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t e = (size_t)blockIdx.x * 64 + lane;
    const int per = (n_partials + kMrWaves - 1) / kMrWaves;
    const int g0 = w * per, g1 = min(n_partials, g0 + per);
    float sum = 0.0f;
    if (e < n_out) {
        int g = g0;
        for (; g + 8 <= g1; g += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = partial[(size_t)(g + k) * n_out + e];
#pragma unroll
            for (int k = 0; k < 8; ++k) sum = __fadd_rn(sum, v[k]);
        }
        for (; g < g1; ++g) sum = __fadd_rn(sum, partial[(size_t)g * n_out + e]);
    }
    part[w][lane] = sum;
    __syncthreads();
    if (w == 0 && e < n_out) {
        float total = part[0][lane];
#pragma unroll
        for (int k = 1; k < kMrWaves; ++k) total = __fadd_rn(total, part[k][lane]);
        out[e] = total;
    }
}

struct ModalLaunch {
    int J, threads, grid;      // modes per lane, workgroup size, workgroups
};

// Enough workgroups to cover the chip (one per CU: two waves per SIMD already saturate packed fp32) before a lane takes
// more modes; up to eight modes per lane (2^20 modes: 256 workgroups x 8 — 108.7 + 4.2 us against 113.6 + 6.0 with
// 512 x 4: half the folds, half the partials).
ModalLaunch modal_launch(int n_modes, int tracks) {
    const int slots = 64 / tracks;
    const long rows = ((long)n_modes + tracks - 1) / tracks;
    const long rows_per_wg1 = (long)kMbWaves * slots;
    const int Jmax = tracks == 32 ? 8 : 4;          // (the compile-time track count's instantiations go to eight: see the kernel)
    int J = 1;
    while (J < Jmax && (rows + rows_per_wg1 * J - 1) / (rows_per_wg1 * J) > 256) J *= 2;
    const long grid = (rows + rows_per_wg1 * J - 1) / (rows_per_wg1 * J);
    return {J, kMbThreads, (int)(grid < 1 ? 1 : grid)};
}

}  // namespace
}  // namespace gab

extern "C" {

size_t gab_modal_bank_workspace_bytes(int n_modes, int out_tracks, int bufsize) {
    if (n_modes <= 0 || out_tracks <= 0 || out_tracks > 64 || bufsize <= 0) return 0;
    const gab::ModalLaunch L = gab::modal_launch(n_modes, out_tracks);
    return sizeof(float) * (size_t)L.grid * out_tracks * bufsize;
}

int gab_modal_bank(const float* d_params, float* d_out, int n_modes, int bufsize, int out_tracks,
                   float* d_workspace, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!d_params || !d_out || !d_workspace) return gab::bad_arg("gab_modal_bank: null pointer");
        if (n_modes <= 0 || bufsize <= 0) return gab::bad_arg("gab_modal_bank: n_modes and bufsize must be > 0");
        if (out_tracks <= 0 || out_tracks > 64)
            return gab::bad_arg("gab_modal_bank: out_tracks must be in 1..64");
        if (reinterpret_cast<uintptr_t>(d_params) & 15u)
            return gab::bad_arg("gab_modal_bank: d_params must be 16-byte aligned");
        hipStream_t s = gab::as_stream(stream);
        const gab::ModalLaunch L = gab::modal_launch(n_modes, out_tracks);
#define GAB_MODAL_LAUNCH(JJ)                                                                          \
    do {                                                                                              \
        if (out_tracks == 32)                                                                         \
            gab::modal_bank_kernel<JJ, 32><<<L.grid, L.threads, 0, s>>>(d_params, d_workspace,         \
                                                                       n_modes, out_tracks, bufsize); \
        else                                                                                          \
            gab::modal_bank_kernel<JJ, 0><<<L.grid, L.threads, 0, s>>>(d_params, d_workspace,          \
                                                                      n_modes, out_tracks, bufsize);  \
    } while (0)
        switch (L.J) {
            case 1: GAB_MODAL_LAUNCH(1); break;
            case 2: GAB_MODAL_LAUNCH(2); break;
            case 8:                                  // 32 tracks only (modal_launch)
                gab::modal_bank_kernel<8, 32><<<L.grid, L.threads, 0, s>>>(d_params, d_workspace, n_modes, out_tracks, bufsize);
                break;
            default: GAB_MODAL_LAUNCH(4); break;
        }
#undef GAB_MODAL_LAUNCH
        int rc = gab::launch_status("modal_bank_kernel");
        if (rc) return rc;
        const size_t n_out = (size_t)out_tracks * bufsize;
        gab::modal_bank_reduce_kernel<<<(unsigned)((n_out + 63) / 64), 64 * gab::kMrWaves, 0, s>>>(d_workspace, d_out,
                                                                                               L.grid, n_out);
        return gab::launch_status("modal_bank_reduce_kernel");
    });
}

}  // extern "C"
