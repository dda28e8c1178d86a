// k_recursive.hip — kernels with a serial dependence per track: IIR biquad,
// time-domain FIR (ordered accumulation) and the 1-D digital waveguide.
//
// The reference gives each track to one thread that walks global memory with a
// stride of one buffer (cuda/bench_iir.cu:10-44, cuda/bench_conv1d.cu:7-27,
// cuda/bench_dwg.cu:10-141).  Here the serial order the CPU goldens define is
// kept (so results are bit-identical to them) while global traffic is staged
// through LDS in coalesced rows, and the waveguide is re-cut into independent
// delay-line cells.
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>

#include <cstdint>

#include "gab_common.hpp"

namespace gab {
namespace {

// ---------------------------------------------------------------------------
// IIR: DF-II biquad, one lane per track, 64 tracks x 64 samples LDS tiles.
// Operation order and rounding follow iirFilterCPUReference exactly
// (cuda/bench_iir.cu:170-197): no fused multiply-adds.
// ---------------------------------------------------------------------------
struct BiquadCoeffs { float b0, b1, b2, a1, a2; };

constexpr int kIirTracks = 64;    // tracks per workgroup = one wavefront of recurrences
constexpr int kIirChunk = 64;     // samples staged per step
constexpr int kIirPitch = 68;     // row pitch in floats: 16-byte aligned rows, conflict-free b128 access

// Wave 0 owns the recurrences (lane = track); all four waves move rows between
// HBM and LDS with 256-byte coalesced accesses.  A lane pulls its 64-sample row
// chunk into registers with 16 ds_read_b128, runs the 64 steps with nothing but
// VALU on the dependency chain, and writes the chunk back — an LDS round trip
// inside the loop costs ~10x the arithmetic.  Tiles are double-buffered so the
// next chunk streams in while this one is filtered.
__global__ __launch_bounds__(256) void iir_biquad_kernel(const float* __restrict__ in,
                                                        float* __restrict__ out,
                                                        float* __restrict__ state,
                                                        BiquadCoeffs c, int T, int B) {
    __shared__ __attribute__((aligned(16))) float tile[2][kIirTracks][kIirPitch];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int t0 = blockIdx.x * kIirTracks;
    const int my_track = t0 + lane;
    const bool owner = (w == 0) && (my_track < T);
    float z1 = 0.0f, z2 = 0.0f;
    if (owner) {
        z1 = state[2 * my_track];
        z2 = state[2 * my_track + 1];
    }
    // Row movers: 16 rows per wave, all requests issued before the first is consumed.
    auto load_rows = [&](int buf, int s0, int first, int step) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            int r = first + k * step, t = t0 + r, s = s0 + lane;
            v[k] = (r < kIirTracks && t < T && s < B) ? in[(size_t)t * B + s] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            int r = first + k * step;
            if (r < kIirTracks) tile[buf][r][lane] = v[k];
        }
    };
    auto store_rows = [&](int buf, int s0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            int r = w + 4 * k, t = t0 + r, s = s0 + lane;
            if (t < T && s < B) out[(size_t)t * B + s] = tile[buf][r][lane];
        }
    };
    const int nchunks = (B + kIirChunk - 1) / kIirChunk;
    load_rows(0, 0, w, 4);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1, s0 = ch * kIirChunk;
        const int n = (B - s0) < kIirChunk ? (B - s0) : kIirChunk;     // valid samples in this chunk
        if (w == 0) {
            if (owner) {
                float x[kIirChunk];
                float4* row = reinterpret_cast<float4*>(&tile[buf][lane][0]);
#pragma unroll
                for (int i = 0; i < kIirChunk / 4; ++i) {
                    float4 v = row[i];
                    x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
                }
#pragma unroll
                for (int i = 0; i < kIirChunk; ++i) {
                    if (i < n) {     // wave-uniform: a ragged last chunk must not run on padding
                        // golden's operation order, one rounding per operation (bench_iir.cu:170-197)
                        float wv = __fsub_rn(__fsub_rn(x[i], __fmul_rn(c.a1, z1)), __fmul_rn(c.a2, z2));
                        float y = __fadd_rn(__fadd_rn(__fmul_rn(c.b0, wv), __fmul_rn(c.b1, z1)),
                                            __fmul_rn(c.b2, z2));
                        z2 = z1;
                        z1 = wv;
                        x[i] = y;
                    }
                }
#pragma unroll
                for (int i = 0; i < kIirChunk / 4; ++i)
                    row[i] = make_float4(x[4 * i], x[4 * i + 1], x[4 * i + 2], x[4 * i + 3]);
            }
        } else if (ch + 1 < nchunks) {
            // the other three waves fetch the next chunk meanwhile (rows w-1, w+2, ...)
            load_rows(buf ^ 1, s0 + kIirChunk, w - 1, 3);
            load_rows(buf ^ 1, s0 + kIirChunk, w - 1 + 48, 3);
        }
        __syncthreads();
        store_rows(buf, s0);
        __syncthreads();
    }
    if (owner) {
        state[2 * my_track] = z1;
        state[2 * my_track + 1] = z2;
    }
}

// ---------------------------------------------------------------------------
// IIR, wave-scan form: one wavefront per track, lane l owns samples [l*M, l*M+M).
//   1. every lane runs the recurrence over its M samples from ZERO state (lane 0
//      from the carried state), giving w_local and its outgoing state c_l;
//   2. the state entering lane l is S_l = A^M S_{l-1} + c_{l-1}: an affine scan
//      with one constant matrix, done in 6 Kogge-Stone steps of __shfl_up with
//      the precomputed powers A^(M*2^k);
//   3. w[i] = w_local[i] + (row i of the homogeneous response) . S_l;
//   4. y[n] = b0 w[n] + b1 w[n-1] + b2 w[n-2] element-wise.
// 512 samples cost ~130 VALU ops per lane instead of a 512-step serial chain.
// Rounding differs from the sequential golden by re-association only; the filter's
// poles (radius sqrt(a2) = 0.41) make A^M tiny, so the result stays within ~1e-7.
// ---------------------------------------------------------------------------
struct IirScanConsts {
    float alpha[16], beta[16];   // w-response of sample i to an incoming state (z1, z2)
    float p[6][4];               // (A^M)^(2^k), row-major 2x2, acting on (z1, z2)
};

// H > 1: the buffer is H segments of 64*M samples scanned one after the other, the state running from one to
// the next.  With M = 4 every load and store instruction of a wave then covers one contiguous KiB (lane l: 16
// bytes at 16 l) instead of every other 16 bytes of two — the same bytes in half the cache-line visits; all H
// segments are requested before the first is scanned.
template <int M, int H = 1>
__global__ __launch_bounds__(256) void iir_scan_kernel(const float* __restrict__ in,
                                                      float* __restrict__ out,
                                                      float* __restrict__ state, BiquadCoeffs c,
                                                      IirScanConsts k, int T) {
    constexpr int S = 64 * M;                    // samples per segment
    constexpr int B = S * H;
    const int lane = threadIdx.x & 63;
    const int track = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (track >= T) return;
    const float* x = in + (size_t)track * B + lane * M;
    float xs[H][M];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        if constexpr (M % 4 == 0) {
#pragma unroll
            for (int i = 0; i < M / 4; ++i) {
                float4 v = reinterpret_cast<const float4*>(x + h * S)[i];
                xs[h][4 * i] = v.x; xs[h][4 * i + 1] = v.y; xs[h][4 * i + 2] = v.z; xs[h][4 * i + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < M; ++i) xs[h][i] = x[h * S + i];
        }
    }
    float in1 = state[2 * track], in2 = state[2 * track + 1];          // the state entering the segment (uniform)
    float* o = out + (size_t)track * B + lane * M;
#pragma unroll
    for (int h = 0; h < H; ++h) {
        float w[M];
        // 1. local pass
        float z1 = 0.0f, z2 = 0.0f;
        if (lane == 0) { z1 = in1; z2 = in2; }
        const float z1_in0 = z1, z2_in0 = z2;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            float wv = xs[h][i] - c.a1 * z1 - c.a2 * z2;
            z2 = z1; z1 = wv; w[i] = wv;
        }
        // 2. inclusive scan of outgoing states: E_l = c_l + A^M E_{l-1}
        float e1 = z1, e2 = z2;
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const int d = 1 << s;
            float u1 = __shfl_up(e1, d, 64), u2 = __shfl_up(e2, d, 64);
            if (lane >= d) {
                e1 += k.p[s][0] * u1 + k.p[s][1] * u2;
                e2 += k.p[s][2] * u1 + k.p[s][3] * u2;
            }
        }
        // state entering this lane (lane 0 already started from the carried state)
        float s1 = __shfl_up(e1, 1, 64), s2 = __shfl_up(e2, 1, 64);
        if (lane == 0) { s1 = 0.0f; s2 = 0.0f; }
        // 3. homogeneous correction
#pragma unroll
        for (int i = 0; i < M; ++i) w[i] += k.alpha[i] * s1 + k.beta[i] * s2;
        // 4. output taps need w[n-1], w[n-2]: the previous lane's last two (or the carried state)
        float p1 = __shfl_up(w[M - 1], 1, 64);
        float p2 = (M >= 2) ? __shfl_up(w[M >= 2 ? M - 2 : 0], 1, 64) : __shfl_up(w[0], 2, 64);
        const float carried_z1 = __shfl(z1_in0, 0, 64);      // all lanes: lane 0's incoming z1
        if (lane == 0) { p1 = z1_in0; p2 = z2_in0; }
        if (M == 1 && lane == 1) p2 = carried_z1;
        float y[M];
#pragma unroll
        for (int i = 0; i < M; ++i) {
            const float wm1 = (i >= 1) ? w[i - 1] : p1;
            const float wm2 = (i >= 2) ? w[i - 2] : (i == 1 ? p1 : p2);
            y[i] = c.b0 * w[i] + c.b1 * wm1 + c.b2 * wm2;
        }
        if constexpr (M % 4 == 0) {
#pragma unroll
            for (int i = 0; i < M / 4; ++i)
                reinterpret_cast<float4*>(o + h * S)[i] = make_float4(y[4 * i], y[4 * i + 1], y[4 * i + 2], y[4 * i + 3]);
        } else {
#pragma unroll
            for (int i = 0; i < M; ++i) o[h * S + i] = y[i];
        }
        // the state leaving the segment: the last lane's last two w
        const float out1 = w[M - 1], out2 = (M >= 2) ? w[M >= 2 ? M - 2 : 0] : p1;
        in1 = __shfl(out1, 63, 64);
        in2 = __shfl(out2, 63, 64);
    }
    if (lane == 63) {
        state[2 * track] = in1;
        state[2 * track + 1] = in2;
    }
}

// Host: constants of the scan for M samples per lane, computed in float64.
inline IirScanConsts make_scan_consts(const BiquadCoeffs& c, int M) {
    IirScanConsts k{};
    // state map per sample on (z1, z2): z1' = -a1 z1 - a2 z2 (+x), z2' = z1
    double A[4] = {-(double)c.a1, -(double)c.a2, 1.0, 0.0};
    double P[4] = {1, 0, 0, 1};
    auto mul = [](const double* X, const double* Y, double* Z) {
        double r[4] = {X[0] * Y[0] + X[1] * Y[2], X[0] * Y[1] + X[1] * Y[3],
                       X[2] * Y[0] + X[3] * Y[2], X[2] * Y[1] + X[3] * Y[3]};
        for (int i = 0; i < 4; ++i) Z[i] = r[i];
    };
    for (int i = 0; i < M; ++i) {
        mul(A, P, P);                       // P = A^(i+1); its first row is w[i]'s response
        k.alpha[i] = (float)P[0];
        k.beta[i] = (float)P[1];
    }
    double Q[4] = {P[0], P[1], P[2], P[3]}; // A^M
    for (int s = 0; s < 6; ++s) {
        for (int i = 0; i < 4; ++i) k.p[s][i] = (float)Q[i];
        mul(Q, Q, Q);
    }
    return k;
}

// ---------------------------------------------------------------------------
// conv1d: y[t*B+i] = sum_j h[t*L+j] * xflat[t*B+i-j], j ascending, with the
// golden's range test on the FLAT index (cuda/bench_conv1d.cu:188-208).  One
// workgroup = 256 consecutive outputs of one track; the input window is staged
// in LDS in chunks of kTapChunk taps.
//
// The sum is a strictly ordered chain of roundings (that is what makes it
// bit-identical to the golden): an output is ONE chain, a wave has 64 of them
// and nothing more, so a tile lasts (taps) x (what one chain step costs the
// wave).  One wave issues a dependent add every 5.25 clocks and any other fp32
// instruction in ~4 (tools/ubench/dep_chain), so add + product is ~9 clocks per
// tap; the rest is kept off the wave's instruction stream or out of its waits:
// the taps are wave-uniform scalar loads (no LDS cycles: at C2, 8 waves per CU,
// the window reads alone are 16 of the LDS's clocks per tap); requests run one
// group of 32 taps ahead in a hand-fixed order (tap_group_*); the tiles at the
// very start of the stream — where the golden SKIPS the taps that reach before
// sample 0 — need no select with finite taps (see the kernel).
// Measured (profiles/r04_conv1d_stamps.txt): 29 -> 11.6 clocks per tap at 1024 taps
// (one wave per SIMD: 2.3 instructions per tap at ~4.6 clocks each), 30 -> 17
// at C2 (two per SIMD, LDS-bound).
// ---------------------------------------------------------------------------
#ifdef GAB_ABLATE
// diagnostic builds: s_memtime of workgroup (1, 7), thread 0 — [0] tile entered, [1] window staged, [2] chain done, [3] back
__device__ unsigned long long g_c1_stamps[4];
#define GAB_C1_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 1 && blockIdx.y == 7) g_c1_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GAB_C1_STAMP(i) do {} while (0)
#endif
constexpr int kTapChunk = 1024;
constexpr int kConvTile = 256;
constexpr int kTapGroup = 16;
constexpr int kWinPad = 32;            // >= the largest prefetch group

// One group of 16 taps in flight: the lane's 16 window samples (x[k] = w[-at-k], as pairs) and the track's 16 taps
// (wave-uniform, SGPRs).  Requested and waited for by hand: scalar loads return out of order, so the only wait that
// covers them is lgkmcnt(0), which takes every LDS read in flight with it — the compiler, which places its waits at the
// first use, therefore waits for the NEXT group's requests as well (29 clocks per tap measured, against ~9 for the
// chain itself: tools/ubench/dep_chain).  Here the order is fixed: wait for the group about to be used, request the
// next, then run the 16 steps, so that requests have a whole group's chain to come back in.
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int kTapSet = 32;            // taps per hand-ordered group: the lead a request has is one group's chain (~270 clocks)
struct TapGroup {
    f2v x[16];
    f16v h0, h1;
};
// lds_lowest: LDS byte address of w[-at-31]; taps: &h[at] (wave-uniform).
// (`acc` passes through both statements: the chain is the only thing the compiler could otherwise move across them)
__device__ __forceinline__ void tap_group_request(TapGroup& g, unsigned lds_lowest, const float* taps, float& acc) {
    asm volatile(
        "ds_read2_b32 %0, %19 offset0:31 offset1:30\n\t"
        "ds_read2_b32 %1, %19 offset0:29 offset1:28\n\t"
        "ds_read2_b32 %2, %19 offset0:27 offset1:26\n\t"
        "ds_read2_b32 %3, %19 offset0:25 offset1:24\n\t"
        "ds_read2_b32 %4, %19 offset0:23 offset1:22\n\t"
        "ds_read2_b32 %5, %19 offset0:21 offset1:20\n\t"
        "ds_read2_b32 %6, %19 offset0:19 offset1:18\n\t"
        "ds_read2_b32 %7, %19 offset0:17 offset1:16\n\t"
        "ds_read2_b32 %8, %19 offset0:15 offset1:14\n\t"
        "ds_read2_b32 %9, %19 offset0:13 offset1:12\n\t"
        "ds_read2_b32 %10, %19 offset0:11 offset1:10\n\t"
        "ds_read2_b32 %11, %19 offset0:9 offset1:8\n\t"
        "ds_read2_b32 %12, %19 offset0:7 offset1:6\n\t"
        "ds_read2_b32 %13, %19 offset0:5 offset1:4\n\t"
        "ds_read2_b32 %14, %19 offset0:3 offset1:2\n\t"
        "ds_read2_b32 %15, %19 offset0:1 offset1:0\n\t"
        "s_load_dwordx16 %16, %20, 0x0\n\t"
        "s_load_dwordx16 %17, %20, 0x40"
        : "=&v"(g.x[0]), "=&v"(g.x[1]), "=&v"(g.x[2]), "=&v"(g.x[3]), "=&v"(g.x[4]), "=&v"(g.x[5]), "=&v"(g.x[6]), "=&v"(g.x[7]),
          "=&v"(g.x[8]), "=&v"(g.x[9]), "=&v"(g.x[10]), "=&v"(g.x[11]), "=&v"(g.x[12]), "=&v"(g.x[13]), "=&v"(g.x[14]), "=&v"(g.x[15]),
          "=&s"(g.h0), "=&s"(g.h1), "+v"(acc)
        : "v"(lds_lowest), "s"(taps)
        : "memory");
}
// Everything requested so far has arrived.  The group passes THROUGH the statement, so nothing reads it earlier.
__device__ __forceinline__ void tap_group_arrive(TapGroup& g, float& acc) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(g.x[0]), "+v"(g.x[1]), "+v"(g.x[2]), "+v"(g.x[3]), "+v"(g.x[4]), "+v"(g.x[5]), "+v"(g.x[6]), "+v"(g.x[7]),
                   "+v"(g.x[8]), "+v"(g.x[9]), "+v"(g.x[10]), "+v"(g.x[11]), "+v"(g.x[12]), "+v"(g.x[13]), "+v"(g.x[14]), "+v"(g.x[15]),
                   "+s"(g.h0), "+s"(g.h1), "+v"(acc)
                 :
                 : "memory");
}
// (products as two-wide vector multiplies: one v_pk_mul_f32 per pair of taps — one wave issues an instruction of ANY kind
// about every 4.6 clocks, so the instruction count per tap is the chain's cost)
__device__ __forceinline__ float tap_group_chain(const TapGroup& g, float acc) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const f2v hp = {g.h0[2 * p], g.h0[2 * p + 1]};
        const f2v pr = hp * g.x[p];
        acc = __fadd_rn(acc, pr.x);
        acc = __fadd_rn(acc, pr.y);
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const f2v hp = {g.h1[2 * p], g.h1[2 * p + 1]};
        const f2v pr = hp * g.x[8 + p];
        acc = __fadd_rn(acc, pr.x);
        acc = __fadd_rn(acc, pr.y);
    }
    return acc;
}

template <bool START>
__device__ __forceinline__ float conv1d_tile(const float* __restrict__ in, const float* __restrict__ h, int L,
                                             long flat0, long total, float* win) {
    GAB_C1_STAMP(0);
    const int tid = threadIdx.x;
    float acc = 0.0f;
    for (int j0 = 0; j0 < L; j0 += kTapChunk) {
        const int nj = (L - j0) < kTapChunk ? (L - j0) : kTapChunk;
        // win[kWinPad + m] = xflat[wbase + m], m in [0, nj-1+tile); the kWinPad floats in front of it
        // are only ever prefetched, never used
        const long wbase = flat0 - j0 - (nj - 1);
        const float* const hs = h + j0;                         // wave-uniform
        for (int m = tid; m < nj - 1 + kConvTile; m += kConvTile) {
            const long g = wbase + m;
            win[kWinPad + m] = (g >= 0 && g < total) ? in[g] : 0.0f;   // a ragged last tile overhangs
        }
        if (tid < kWinPad) win[tid] = 0.0f;
        __syncthreads();
        GAB_C1_STAMP(1);
        const float* w = win + kWinPad + (nj - 1) + tid;      // w[-jj]: this output's sample at tap jj
        const int valid = (int)(flat0 + tid - j0 < nj ? flat0 + tid - j0 : nj);   // START: taps 0..valid are in range
        int jj = 0;
        if constexpr (!START) {
            // every tap is in range (or meets a zero of the window: see the kernel)
            const unsigned w_lds = (unsigned)(uintptr_t)w;          // the low half of a flat LDS address is the LDS offset
            auto request = [&](TapGroup& g, int at) {
                const int hat = at + kTapSet <= nj ? at : nj - kTapSet;         // past the last whole group: re-read it, unused
                tap_group_request(g, w_lds - 4u * (unsigned)(at + kTapSet - 1), hs + hat, acc);
            };
            if (nj >= kTapSet) {
                TapGroup A, B;
                request(A, 0);
                for (; jj + 2 * kTapSet <= nj; jj += 2 * kTapSet) {
                    tap_group_arrive(A, acc);
                    request(B, jj + kTapSet);
                    acc = tap_group_chain(A, acc);
                    tap_group_arrive(B, acc);
                    request(A, jj + 2 * kTapSet);                       // at the end: the pads
                    acc = tap_group_chain(B, acc);
                }
                tap_group_arrive(A, acc);                               // nothing stays in flight
                if (jj + kTapSet <= nj) { acc = tap_group_chain(A, acc); jj += kTapSet; }
            }
            for (; jj < nj; ++jj) acc = __fadd_rn(acc, __fmul_rn(hs[jj], w[-jj]));
        } else {
            // the golden skips a tap that reaches before sample 0, it does not add zero: a select on the chain
            float xa[kTapGroup], xb[kTapGroup], ha[kTapGroup], hb[kTapGroup];
            auto fetch = [&](float (&x)[kTapGroup], float (&hh)[kTapGroup], int at) {
#pragma unroll
                for (int k = 0; k < kTapGroup; ++k) x[k] = w[-at - k];
                const int hat = at + kTapGroup <= nj ? at : nj - kTapGroup;
#pragma unroll
                for (int k = 0; k < kTapGroup; ++k) hh[k] = hs[hat + k];
            };
            auto chain = [&](const float (&x)[kTapGroup], const float (&hh)[kTapGroup], int at) {
#pragma unroll
                for (int k = 0; k < kTapGroup; ++k) {
                    const float next = __fadd_rn(acc, __fmul_rn(hh[k], x[k]));
                    acc = (at + k <= valid) ? next : acc;
                }
            };
            if (nj >= kTapGroup) {
                fetch(xa, ha, 0);
                for (; jj + 2 * kTapGroup <= nj; jj += 2 * kTapGroup) {
                    fetch(xb, hb, jj + kTapGroup);
                    chain(xa, ha, jj);
                    fetch(xa, ha, jj + 2 * kTapGroup);                  // at the end: the pads
                    chain(xb, hb, jj + kTapGroup);
                }
                if (jj + kTapGroup <= nj) { chain(xa, ha, jj); jj += kTapGroup; }
            }
            for (; jj < nj; ++jj) {
                const float next = __fadd_rn(acc, __fmul_rn(hs[jj], w[-jj]));
                acc = (jj <= valid) ? next : acc;
            }
        }
        GAB_C1_STAMP(2);
        __syncthreads();
    }
    return acc;
}

__global__ __launch_bounds__(kConvTile) void conv1d_direct_kernel(const float* __restrict__ in,
                                                                 float* __restrict__ out,
                                                                 const float* __restrict__ ir,
                                                                 int L, int T, int B, int halo) {
    __shared__ float win[kWinPad + kTapChunk + kConvTile];
    const int t = blockIdx.y;
    const int i0 = blockIdx.x * kConvTile;
    // `in` starts `halo` tracks before the first track computed here (a channel shard's input; 0 for a whole job):
    // the flat index runs over those rows too, `ir` and `out` hold the computed tracks only
    const long flat0 = (long)(halo + t) * B + i0;  // flat index of this tile's first output
    const long total = (long)(halo + T) * B;
    const float* h = ir + (size_t)t * L;
    // Tiles at the very start of the stream: the golden SKIPS the taps that reach before sample 0.  The window holds
    // zeros there, a finite tap times zero is +-0, and a round-to-nearest sum that started at +0 is never -0, so adding
    // it changes no bit: with finite taps the skip needs no code at all.  (A kernel lasts as long as its longest chain:
    // a select on the sum here would set the duration of the whole launch.)  Only a track whose response holds an
    // infinity or a NaN takes the chain with the select.
    bool select_chain = false;
    if (flat0 < L - 1) {
        int finite = 1;
        for (int m = threadIdx.x; m < L; m += kConvTile) finite &= (__float_as_uint(h[m]) & 0x7f800000u) != 0x7f800000u;
        select_chain = !__syncthreads_and(finite);
    }
    const float acc = select_chain ? conv1d_tile<true>(in, h, L, flat0, total, win)
                                   : conv1d_tile<false>(in, h, L, flat0, total, win);
    GAB_C1_STAMP(3);
    if (i0 + (int)threadIdx.x < B) out[(size_t)t * B + i0 + threadIdx.x] = acc;
}

// ---------------------------------------------------------------------------
// Digital waveguide.  With writePos fixed, sample s touches exactly the cell
// pair (fwd[p], bwd[(p+L/2)%L]) with p = (writePos+s) % L, and no other sample
// position ever touches that pair: the 512-step "serial" loop is really
// min(L,B) independent two-cell recurrences of length <= ceil(B/L).
//   NAIVE: one thread per waveguide walks the samples in order (reference form).
//   ACCEL: one thread per (waveguide, cell); both cells live in registers.
// Output-tap contributions go to workspace[g*B + s]; a second kernel adds them
// per sample in waveguide order — the golden's order, no float atomics.
// ---------------------------------------------------------------------------
struct WG { int length, inTap, outTap, writePos; float gain, reflection, damping, pad; };

__device__ __forceinline__ void dwg_step(float& f, float& b, float x, bool inject, const WG& wg,
                                         float& mix) {
    f = __fmul_rn(f, wg.damping);
    b = __fmul_rn(b, wg.damping);
    if (inject) { f = __fadd_rn(f, x); b = __fadd_rn(b, x); }
    mix = __fmul_rn(__fadd_rn(f, b), 0.5f);          // WAVEGUIDE_MIX_FACTOR
    float nf = __fmul_rn(b, wg.reflection);
    float nb = __fmul_rn(f, wg.reflection);
    f = nf;
    b = nb;
}

// Where sample s of waveguide g reaches the output tap: s = first[g] + k * period[g].  The mix kernel
// reads this pair instead of the 32-byte record and a modulo per (sample, waveguide).
__device__ __forceinline__ void dwg_publish_hits(const WG& wg, int g, int2* __restrict__ hits) {
    int f = wg.outTap - wg.writePos % wg.length;
    if (f < 0) f += wg.length;
    // a tap outside the line is never reached
    hits[g] = make_int2((wg.outTap >= 0 && wg.outTap < wg.length) ? f : 0x7fffffff, wg.length);
}

// The reference form: one thread per waveguide walks the samples in order.  Consecutive samples touch
// consecutive cells, so a run of up to 16 samples (fewer than the line is long) reads 16 DIFFERENT
// cell pairs: their loads are issued together instead of one dependent round trip per sample.  Lanes
// of a wave walk DIFFERENT lines (8 KB apart), every access is its own cache line, so a workgroup is
// 16 threads: the waveguides spread over four times as many CUs' address paths (140 -> 61 us at 128
// waveguides with the runs alone).  Same operations in the same order per cell: bit-identical.
__global__ __launch_bounds__(64) void dwg_naive_kernel(const WG* __restrict__ wgs,
                                                      float* __restrict__ fwd, float* __restrict__ bwd,
                                                      const float* __restrict__ input,
                                                      float* __restrict__ ws, int2* __restrict__ hits, int n_wg, int B,
                                                      int max_len, int* __restrict__ mix_count) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (mix_count && blockIdx.x == 0) for (int i = threadIdx.x; i < B; i += blockDim.x) mix_count[i] = 0;   // for dwg_gather_kernel
    if (g >= n_wg) return;
    const WG wg = wgs[g];
    dwg_publish_hits(wg, g, hits);
    float* F = fwd + (size_t)g * max_len;
    float* Bk = bwd + (size_t)g * max_len;
    const int L = wg.length;
    constexpr int U = 16;
    int cur = wg.writePos % L;
    int bp = (cur + L / 2) % L;
    int s = 0;
    if (L >= 2 * U) {                        // a run never meets its own cells again (forward or backward line)
        for (; s + U <= B; s += U) {
            float f[U], b[U], x[U];
            int c[U], d[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                c[j] = cur; d[j] = bp;
                f[j] = F[cur]; b[j] = Bk[bp];
                x[j] = input[s + j];
                cur = cur + 1 == L ? 0 : cur + 1;
                bp = bp + 1 == L ? 0 : bp + 1;
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                float mix;
                dwg_step(f[j], b[j], __fmul_rn(x[j], wg.gain), c[j] == wg.inTap, wg, mix);
                F[c[j]] = f[j];
                Bk[d[j]] = b[j];
                if (c[j] == wg.outTap) ws[(size_t)g * B + s + j] = mix;
            }
        }
    }
    for (; s < B; ++s) {
        float f = F[cur], b = Bk[bp], mix;
        float x = __fmul_rn(input[s], wg.gain);
        dwg_step(f, b, x, cur == wg.inTap, wg, mix);
        F[cur] = f;
        Bk[bp] = b;
        if (cur == wg.outTap) ws[(size_t)g * B + s] = mix;
        cur = cur + 1 == L ? 0 : cur + 1;
        bp = bp + 1 == L ? 0 : bp + 1;
    }
}

// writePos % L without the divide in the usual case (the harness never advances writePos: the reference's own defect,
// SURVEY 2.3); the value is uniform over the workgroup either way
__device__ __forceinline__ int wrap_once(int writePos, int L) {
    return (unsigned)writePos < (unsigned)L ? writePos : writePos % L;
}

__global__ __launch_bounds__(256) void dwg_cells_kernel(const WG* __restrict__ wgs,
                                                       float* __restrict__ fwd, float* __restrict__ bwd,
                                                       const float* __restrict__ input,
                                                       float* __restrict__ ws, int2* __restrict__ hits, int n_wg, int B,
                                                       int max_len, int* __restrict__ mix_count) {
    __shared__ float xin[2048];                       // staged input (B <= 2048), else global
    const int g = blockIdx.y;
    if (mix_count && blockIdx.x == 0 && g == 0) for (int i = threadIdx.x; i < B; i += blockDim.x) mix_count[i] = 0;   // for dwg_gather_kernel
    const WG wg = wgs[g];
    const bool staged = B <= 2048;
    if (staged) {
        for (int i = threadIdx.x; i < B; i += blockDim.x) xin[i] = input[i];
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) dwg_publish_hits(wg, g, hits);
    // thread j owns the j-th cell the buffer visits: cell (writePos + j) % L, first touched by sample j
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= wg.length || s >= B) return;
    // (writePos % L + s) % L and (p + L/2) % L with both addends below L: one conditional subtraction each — the
    // general modulo is ~40 VALU instructions by a run-time divisor, three of them per cell were most of this kernel
    const int wp = wrap_once(wg.writePos, wg.length);
    int p = wp + s;
    if (p >= wg.length) p -= wg.length;
    int bp = p + wg.length / 2;
    if (bp >= wg.length) bp -= wg.length;
    float* F = fwd + (size_t)g * max_len + p;
    float* Bk = bwd + (size_t)g * max_len + bp;
    float f = *F, b = *Bk, mix;
    const bool inject = (p == wg.inTap), tap = (p == wg.outTap);
    for (; s < B; s += wg.length) {
        float x = __fmul_rn(staged ? xin[s] : input[s], wg.gain);
        dwg_step(f, b, x, inject, wg, mix);
        if (tap) ws[(size_t)g * B + s] = mix;
    }
    *F = f;
    *Bk = b;
}

// The same for banks of a thousand waveguides and more: a workgroup of the one-line kernel moves 4 KB and costs a dispatch,
// and at 8 192 lines there are 16 384 of them (19.7 us for 67 MB).  Here a thread owns cell j of U consecutive lines: the
// input is staged once per U lines, and the 2 U cell loads of a thread are all requested before the first is used.
// Same operations per cell, same order: bit-identical.  Round 4, measured at 8 192 lines: 19.7 us (one line per workgroup,
// three run-time modulos per cell) -> 16.9 us; the counters of that launch (profiles/r04_dwg_accel_8192_pmc_means.txt): 32.7 MB
// read + 30.3 MB written, the waves alive ~12 us of the launch's 17 = 5 TB/s while they run, VALU busy 14 % — the delay
// lines' 2 KB pieces at DRAM rate plus a launch's ramp.  Four cells per thread as 16-byte pieces (a quarter of the
// memory instructions): 17.7 us, not kept.
template <int U>
__global__ __launch_bounds__(256) void dwg_cells_multi_kernel(const WG* __restrict__ wgs,
                                                             float* __restrict__ fwd, float* __restrict__ bwd,
                                                             const float* __restrict__ input,
                                                             float* __restrict__ ws, int2* __restrict__ hits, int n_wg, int B,
                                                             int max_len, int* __restrict__ mix_count) {
    __shared__ float xin[2048];
    const int g0 = blockIdx.y * U;
    if (mix_count && blockIdx.x == 0 && blockIdx.y == 0) for (int i = threadIdx.x; i < B; i += blockDim.x) mix_count[i] = 0;
    const bool staged = B <= 2048;
    if (staged) {
        for (int i = threadIdx.x; i < B; i += blockDim.x) xin[i] = input[i];
        __syncthreads();
    }
    const int s0 = blockIdx.x * blockDim.x + threadIdx.x;
    WG wg[U];
    float f[U], b[U];
    float* F[U];
    float* Bk[U];
    bool live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int g = g0 + u;
        live[u] = false;
        if (g >= n_wg) continue;
        wg[u] = wgs[g];
        if (blockIdx.x == 0 && threadIdx.x == 0) dwg_publish_hits(wg[u], g, hits);
        live[u] = s0 < wg[u].length && s0 < B;
        int p = wrap_once(wg[u].writePos, wg[u].length) + s0;      // both below L (live threads): conditional subtractions
        if (p >= wg[u].length) p -= wg[u].length;
        int bp = p + wg[u].length / 2;
        if (bp >= wg[u].length) bp -= wg[u].length;
        F[u] = fwd + (size_t)g * max_len + p;
        Bk[u] = bwd + (size_t)g * max_len + bp;
        wg[u].pad = __int_as_float(p);                     // the cell, kept where the record has room
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (live[u]) { f[u] = *F[u]; b[u] = *Bk[u]; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (!live[u]) continue;
        const int g = g0 + u;
        const int p = __float_as_int(wg[u].pad);
        const bool inject = (p == wg[u].inTap), tap = (p == wg[u].outTap);
        float mix;
        for (int s = s0; s < B; s += wg[u].length) {
            float x = __fmul_rn(staged ? xin[s] : input[s], wg[u].gain);
            dwg_step(f[u], b[u], x, inject, wg[u], mix);
            if (tap) ws[(size_t)g * B + s] = mix;
        }
        *F[u] = f[u];
        *Bk[u] = b[u];
    }
}

// The ordered per-sample mix, sparse form.  A waveguide reaches its output tap at samples first + k * length:
// a handful per buffer, against the (samples x waveguides) pairs a scan of every waveguide per sample tests.
//   dwg_gather_kernel: one thread per waveguide appends (g, its tap value) to the list of every sample it hits
//                      (an atomic counter per sample; lists of kMixCap entries);
//   dwg_mix_kernel:    one wavefront per sample sorts its list by g (bitonic, in registers) and adds the values
//                      in that order — the golden's ordered sum; a sample with more than kMixCap entries (every
//                      line the same length and phase, say) falls back to the scan below.
constexpr int kMixCap = 64;
__global__ __launch_bounds__(256) void dwg_gather_kernel(const int2* __restrict__ hits, const float* __restrict__ ws,
                                                        int* __restrict__ mix_count, int2* __restrict__ mix_list,
                                                        int n, int B) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const int2 h = hits[g];                                // h.x: first sample on the tap (never: 0x7fffffff), h.y: the line's length
    for (long long s = h.x; s < B; s += h.y) {
        const int idx = atomicAdd(&mix_count[s], 1);
        if (idx < kMixCap) mix_list[(size_t)s * kMixCap + idx] = make_int2(g, __float_as_int(ws[(size_t)g * B + s]));
    }
}

// The scan form (kept for crowded samples): lanes test 64 waveguides at a time for "does sample s land on
// your output tap" — s = first[g] + k * period[g], two coalesced words per waveguide, a modulo only
// for lines shorter than the buffer — and the (few) that do are added in waveguide order via ballot
// + readlane, so the sum is the golden's ordered sum without a serial scan over every waveguide.
__global__ __launch_bounds__(256) void dwg_mix_kernel(const int2* __restrict__ hits,
                                                     const float* __restrict__ ws,
                                                     float* __restrict__ out, int n_wg, int B,
                                                     int out_tracks, const int* mix_count,
                                                     const int2* __restrict__ mix_list, bool zero_counts) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (s >= B) return;
    float acc = 0.0f;
    const int n = n_wg < out_tracks ? n_wg : out_tracks;
    const int c = mix_count ? mix_count[s] : kMixCap + 1;  // wave-uniform; no lists: the scan
    if (zero_counts && lane == 0 && c != 0) const_cast<int*>(mix_count)[s] = 0;    // (c is in hand: the load is through) the pool's slot goes back to zero
    if (c <= kMixCap) {
        int key = 0x7fffffff;
        float val = 0.0f;
        if (lane < c) {
            const int2 e = mix_list[(size_t)s * kMixCap + lane];
            key = e.x;
            val = __int_as_float(e.y);
        }
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                const int pk = __shfl_xor(key, j, 64);
                const float pv = __shfl_xor(val, j, 64);
                const bool keep_min = ((lane & j) == 0) == ((lane & k) == 0);
                const bool take = keep_min ? pk < key : pk > key;
                key = take ? pk : key;
                val = take ? pv : val;
            }
        }
        for (int i = 0; i < c; ++i) acc = __fadd_rn(acc, __shfl(val, i, 64));     // ascending g: the golden's order
        if (lane == 0) out[s] = acc;
        return;
    }
    constexpr int U = 8;                               // chunks of 64 waveguides requested together:
    for (int g0 = 0; g0 < n; g0 += 64 * U) {           // the scan is a chain of dependent loads otherwise
        int2 h[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int g = g0 + 64 * k + lane;
            h[k] = g < n ? hits[g] : make_int2(0x7fffffff, 1);
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int g = g0 + 64 * k + lane;
            const int d = s - h[k].x;                  // h.x: first sample on the tap, h.y: the line's length
            const bool hit = d == 0 || (d >= h[k].y && (d % h[k].y) == 0);
            unsigned long long m = __ballot(hit);
            if (m) {                                   // wave-uniform
                const float v = hit ? ws[(size_t)g * B + s] : 0.0f;
                while (m) {
                    const int src = __ffsll((long long)m) - 1;
                    acc = __fadd_rn(acc, __shfl(v, src, 64));
                    m &= m - 1;
                }
            }
        }
    }
    if (lane == 0) out[s] = acc;
}

// ---- banks of two thousand mixed waveguides and more: the cells kernel appends to the hit lists itself (round 5) ---------
// Round 4 took three launches at 8 192 lines: cells 17.2 us + gather 6.2 + mix 4.3.  The gather — one thread per waveguide
// appending (g, tap value) to the list of every sample it reaches — was a second pass over values the cells kernel had just
// produced, and a kernel boundary.  Here the thread that owns the output tap's cell appends as it goes.  (Its atomics
// requested BEFORE the cell loads, so that their round trip runs beside the loads': 27 more registers per thread, five
// waves per SIMD instead of seven, 29.4 us instead of 17 — not kept.)  The per-sample counters must be ZERO before the first append, which no
// workgroup of the same launch can guarantee for the others: they live in a zero-initialised array of the code object
// (one slot per call in flight, kDwgSlots of them, picked round robin by the host) and the mix kernel puts every counter
// back to zero behind itself.  (All three steps in ONE launch — the launch's last workgroups waiting for every other
// one's appends, then mixing — was built and measured: 27.7 us, the three launches' time; profiles/r05_dwg_one_launch.txt.)
// Same operations per cell in the same order as dwg_cells_multi_kernel, same ordered sum: bit-identical.
constexpr int kDwgSlots = 16;
constexpr int kDwgMaxB = 2048;
__device__ int g_dwg_count[kDwgSlots][kDwgMaxB];      // hits per sample; zero between calls

// STAGED = true is round 5's form (U = 8): the buffer's input staged in LDS behind a barrier, 2048 workgroups at 8192 lines —
// EIGHT per compute unit where seven fit (61 registers), so the launch ran two rounds of a chain of three dependent round
// trips (input -> LDS -> barrier; the lines' records; their cells): 20.1 us for 67 MB.  STAGED = false (round 6): no LDS and
// no barrier — a thread's first sample x[s0] is one coalesced load requested beside the records (every line of 512 cells and
// more uses no other; shorter lines fetch their later samples from the L1/L2 as they go) — and U = 16 lines per workgroup:
// 1024 workgroups, four per compute unit, ONE round, twice the cell loads in flight per thread.  Same operations per cell in
// the same order: bit-identical (test_dwg_large_bank_*).
template <int U, bool STAGED>
__global__ __launch_bounds__(256) void dwg_cells_append_kernel(const WG* __restrict__ wgs,
                                                              float* __restrict__ fwd, float* __restrict__ bwd,
                                                              const float* __restrict__ input,
                                                              float* __restrict__ ws, int2* __restrict__ hits,
                                                              int2* __restrict__ mix_list, int n_wg, int n_mix, int B,
                                                              int max_len, int slot) {
    __shared__ float xin[STAGED ? kDwgMaxB : 1];
    int* const count = g_dwg_count[slot];
    const int g0 = blockIdx.y * U;
    const int s0 = blockIdx.x * blockDim.x + threadIdx.x;
    float x0 = 0.0f;
    if constexpr (STAGED) {
        for (int i = threadIdx.x; i < B; i += blockDim.x) xin[i] = input[i];
        __syncthreads();
    } else {
        if (s0 < B) x0 = input[s0];
    }
    WG wg[U];
    float f[U], b[U];
    float* F[U];
    float* Bk[U];
    bool live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int g = g0 + u;
        live[u] = false;
        if (g >= n_wg) continue;
        wg[u] = wgs[g];
        if (blockIdx.x == 0 && threadIdx.x == 0) dwg_publish_hits(wg[u], g, hits);
        live[u] = s0 < wg[u].length && s0 < B;
        int p = wrap_once(wg[u].writePos, wg[u].length) + s0;      // both below L (live threads): conditional subtractions
        if (p >= wg[u].length) p -= wg[u].length;
        int bp = p + wg[u].length / 2;
        if (bp >= wg[u].length) bp -= wg[u].length;
        F[u] = fwd + (size_t)g * max_len + p;
        Bk[u] = bwd + (size_t)g * max_len + bp;
        wg[u].pad = __int_as_float(p);                     // the cell, kept where the record has room
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (live[u]) { f[u] = *F[u]; b[u] = *Bk[u]; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (!live[u]) continue;
        const int g = g0 + u;
        const int p = __float_as_int(wg[u].pad);
        const bool inject = (p == wg[u].inTap), tap = (p == wg[u].outTap);
        float mix;
        for (int s = s0; s < B; s += wg[u].length) {
            float x = __fmul_rn(STAGED ? xin[s] : (s == s0 ? x0 : input[s]), wg[u].gain);
            dwg_step(f[u], b[u], x, inject, wg[u], mix);
            if (tap) {
                ws[(size_t)g * B + s] = mix;                        // (the crowded-sample scan's copy)
                if (g < n_mix) {
                    const int idx = atomicAdd(&count[s], 1);
                    if (idx < kMixCap) mix_list[(size_t)s * kMixCap + idx] = make_int2(g, __float_as_int(mix));
                }
            }
        }
        *F[u] = f[u];
        *Bk[u] = b[u];
    }
}

// ---- round 6: the same cells, a quarter of the instructions ---------------------------------------------------------
// Counters of the kernel above at 8192 lines (profiles/r06_dwg_pmc.md): 523 vector and 417 scalar instructions per WAVE for
// 16 loads, 16 stores and 80 floating-point operations — 64-bit pointer arithmetic per cell, an exec-mask region per
// `if (live)` per loop, a divide in the tap bookkeeping — and a wave alive 4 650 issue slots of which it issues in 1 100:
// the launch is bound by instruction issue (the scalar unit is one per compute unit: 13 000 scalar instructions each), not
// by its 61 MB.  Here:
//   * the delay lines are BUFFER resources: a cell's address is one v_lshl_add (line base + 4 p, 32 bits), and a lane that
//     owns no cell of a line carries the offset 0xffffffff — its load returns zero and its store is dropped by the
//     resource's range check, so no exec-mask region is needed around either;
//   * wrap-arounds as min(p, p - L) on unsigned values (two instructions);
//   * lines of B cells and more (four in five of the reference's 100..1999) take ONE step per lane: straight-line code, the
//     loop over further samples behind a scalar branch on L < B;
//   * the output tap (at most one lane per line) is remembered as a bit and a value and handled for all U lines in one
//     region after the stores, which most waves skip.
// Same floating-point operations per cell in the same order (the injection stays a select: f + 0 is not f for f = -0):
// bit-identical to the forms above (test_dwg_large_bank_*).
template <int U>
__global__ __launch_bounds__(256) void dwg_cells_lean_kernel(const WG* __restrict__ wgs,
                                                            float* __restrict__ fwd, float* __restrict__ bwd,
                                                            const float* __restrict__ input,
                                                            float* __restrict__ ws, int2* __restrict__ hits,
                                                            int2* __restrict__ mix_list, int n_wg, int n_mix, int B,
                                                            int max_len, int slot, unsigned line_bytes_total) {
    int* const count = g_dwg_count[slot];
    const int g0 = blockIdx.y * U;
    const unsigned s0 = blockIdx.x * blockDim.x + threadIdx.x;
    const auto rf = __builtin_amdgcn_make_buffer_rsrc(fwd, 0, (int)line_bytes_total, 0x00020000);
    const auto rb = __builtin_amdgcn_make_buffer_rsrc(bwd, 0, (int)line_bytes_total, 0x00020000);
    const float x0 = s0 < (unsigned)B ? input[s0] : 0.0f;
    // all U records requested at once, nothing conditional between them (a line beyond the bank re-reads the last record
    // and owns no cell below): one wait for the scalar loads instead of one per line
    WG wg[U];
#pragma unroll
    for (int u = 0; u < U; ++u) wg[u] = wgs[min(g0 + u, n_wg - 1)];
    if (blockIdx.x == 0 && threadIdx.x < U && g0 + (int)threadIdx.x < n_wg)      // lane u of the first wave: where line u's tap is reached
        dwg_publish_hits(wgs[g0 + threadIdx.x], g0 + threadIdx.x, hits);
    unsigned of[U], ob[U];                                     // byte offsets of this lane's two cells; 0xffffffff: none
    unsigned cell[U];
    unsigned livebits = 0;                                     // bit u: this lane owns a cell of line u
    float f[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int g = g0 + u;
        const unsigned L = (unsigned)wg[u].length;
        const unsigned wp = (unsigned)wrap_once(wg[u].writePos, wg[u].length);
        const unsigned lim = g < n_wg ? (L < (unsigned)B ? L : (unsigned)B) : 0u;   // the buffer visits min(L, B) cells of the line
        unsigned p = wp + s0;
        p = min(p, p - L);                                      // both addends below L (live lanes): one wrap
        unsigned q = p + (L >> 1);
        q = min(q, q - L);
        const unsigned base = (unsigned)g * (unsigned)max_len * 4u;
        const bool live = s0 < lim;
        of[u] = live ? base + 4u * p : 0xffffffffu;
        ob[u] = live ? base + 4u * q : 0xffffffffu;
        cell[u] = p;
        livebits |= live ? (1u << u) : 0u;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        f[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, of[u], 0, 0));
        b[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, ob[u], 0, 0));
    }
    unsigned tapbits = 0;
    float tapval[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const bool live = (livebits >> u) & 1u;
        const bool inject = cell[u] == (unsigned)wg[u].inTap;                 // (a lane without a cell: its values go nowhere)
        const bool tap = live && cell[u] == (unsigned)wg[u].outTap;
        float mix;
        dwg_step(f[u], b[u], __fmul_rn(x0, wg[u].gain), inject, wg[u], mix);
        tapval[u] = mix;
        tapbits |= tap ? (1u << u) : 0u;
        if (wg[u].length < B) {                                 // (uniform) a short line: this lane's cell is visited again
            const int g = g0 + u;
            if (tap && g < n_mix) {                             // the first visit's tap value goes out here: the loop overwrites it
                ws[(size_t)g * B + s0] = mix;
                const int idx = atomicAdd(&count[s0], 1);
                if (idx < kMixCap) mix_list[(size_t)s0 * kMixCap + idx] = make_int2(g, __float_as_int(mix));
            } else if (tap) {
                ws[(size_t)g * B + s0] = mix;
            }
            tapbits &= ~(1u << u);
            if (live) {
                for (unsigned s = s0 + (unsigned)wg[u].length; s < (unsigned)B; s += (unsigned)wg[u].length) {
                    dwg_step(f[u], b[u], __fmul_rn(input[s], wg[u].gain), inject, wg[u], mix);
                    if (tap) {
                        ws[(size_t)g * B + s] = mix;
                        if (g < n_mix) {
                            const int idx = atomicAdd(&count[s], 1);
                            if (idx < kMixCap) mix_list[(size_t)s * kMixCap + idx] = make_int2(g, __float_as_int(mix));
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, f[u]), rf, of[u], 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, b[u]), rb, ob[u], 0, 0);
    }
    if (tapbits) {                                              // at most one lane per line: most waves skip this
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!((tapbits >> u) & 1u)) continue;
            const int g = g0 + u;
            ws[(size_t)g * B + s0] = tapval[u];                 // (the crowded-sample scan's copy)
            if (g < n_mix) {
                const int idx = atomicAdd(&count[s0], 1);
                if (idx < kMixCap) mix_list[(size_t)s0 * kMixCap + idx] = make_int2(g, __float_as_int(tapval[u]));
            }
        }
    }
}

}  // namespace
}  // namespace gab

extern "C" {

int gab_iir_sequential(const float* d_in, float* d_out, const float* coeffs, float* d_state,
                       int tracks, int bufsize, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!d_in || !d_out || !coeffs || !d_state) return gab::bad_arg("gab_iir_sequential: null pointer");
        if (tracks <= 0 || bufsize <= 0) return gab::bad_arg("gab_iir_sequential: tracks and bufsize must be > 0");
        gab::BiquadCoeffs c{coeffs[0], coeffs[1], coeffs[2], coeffs[3], coeffs[4]};
        dim3 grid((tracks + gab::kIirTracks - 1) / gab::kIirTracks);
        gab::iir_biquad_kernel<<<grid, 256, 0, gab::as_stream(stream)>>>(d_in, d_out, d_state, c,
                                                                           tracks, bufsize);
        return gab::launch_status("iir_biquad_kernel");
    });
}

int gab_iir(const float* d_in, float* d_out, const float* coeffs, float* d_state, int tracks,
            int bufsize, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!d_in || !d_out || !coeffs || !d_state) return gab::bad_arg("gab_iir: null pointer");
        if (tracks <= 0 || bufsize <= 0) return gab::bad_arg("gab_iir: tracks and bufsize must be > 0");
        const int m = bufsize / 64;
        const bool scan_ok = bufsize % 64 == 0 && (m == 1 || m == 2 || m == 4 || m == 8 || m == 16) &&
                             (reinterpret_cast<uintptr_t>(d_in) & 15u) == 0 &&
                             (reinterpret_cast<uintptr_t>(d_out) & 15u) == 0;
        if (!scan_ok) return gab_iir_sequential(d_in, d_out, coeffs, d_state, tracks, bufsize, stream);
        gab::BiquadCoeffs c{coeffs[0], coeffs[1], coeffs[2], coeffs[3], coeffs[4]};
        // 512 and 1024 samples, many tracks (bandwidth-bound): segments of 256, one contiguous KiB per load
        // instruction — 65 536 x 512: 43.3 us against 48.0 (0.78 against 0.70 of 8 TB/s); at 8 192 tracks the two
        // scans in a row cost more (10.3 against 10.0 us) than the tidier accesses save, so fewer tracks keep one scan
        const bool segments = m >= 8 && tracks >= 16384;
        const gab::IirScanConsts k = gab::make_scan_consts(c, segments ? 4 : m);
        dim3 grid((tracks + 3) / 4);
        hipStream_t s = gab::as_stream(stream);
        switch (m) {
            case 1: gab::iir_scan_kernel<1><<<grid, 256, 0, s>>>(d_in, d_out, d_state, c, k, tracks); break;
            case 2: gab::iir_scan_kernel<2><<<grid, 256, 0, s>>>(d_in, d_out, d_state, c, k, tracks); break;
            case 4: gab::iir_scan_kernel<4><<<grid, 256, 0, s>>>(d_in, d_out, d_state, c, k, tracks); break;
            case 8:
                if (segments) gab::iir_scan_kernel<4, 2><<<grid, 256, 0, s>>>(d_in, d_out, d_state, c, k, tracks);
                else gab::iir_scan_kernel<8><<<grid, 256, 0, s>>>(d_in, d_out, d_state, c, k, tracks);
                break;
            default:
                if (segments) gab::iir_scan_kernel<4, 4><<<grid, 256, 0, s>>>(d_in, d_out, d_state, c, k, tracks);
                else gab::iir_scan_kernel<16><<<grid, 256, 0, s>>>(d_in, d_out, d_state, c, k, tracks);
                break;
        }
        return gab::launch_status("iir_scan_kernel");
    });
}

#ifdef GAB_ABLATE
int gab_debug_conv1d_stamps(unsigned long long* h_out) {
    (void)hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(h_out, HIP_SYMBOL(gab::g_c1_stamps), sizeof(unsigned long long) * 4);
}
#endif

int gab_conv1d_shard(const float* d_in, float* d_out, const float* d_ir, int ir_len, int tracks,
                     int bufsize, int halo_tracks, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!d_in || !d_out || !d_ir) return gab::bad_arg("gab_conv1d: null pointer");
        if (ir_len <= 0 || tracks <= 0 || bufsize <= 0 || halo_tracks < 0) return gab::bad_arg("gab_conv1d: sizes must be > 0");
        dim3 grid((bufsize + gab::kConvTile - 1) / gab::kConvTile, tracks);
        gab::conv1d_direct_kernel<<<grid, gab::kConvTile, 0, gab::as_stream(stream)>>>(
            d_in, d_out, d_ir, ir_len, tracks, bufsize, halo_tracks);
        return gab::launch_status("conv1d_direct_kernel");
    });
}

int gab_conv1d(const float* d_in, float* d_out, const float* d_ir, int ir_len, int tracks,
               int bufsize, gab_stream_t stream) {
    return gab_conv1d_shard(d_in, d_out, d_ir, ir_len, tracks, bufsize, 0, stream);
}

size_t gab_dwg_workspace_bytes(int n_waveguides, int bufsize) {
    if (n_waveguides <= 0 || bufsize <= 0) return 0;
    // tap contributions [n][B] (padded to 8 bytes), then where each waveguide's output tap is reached: (first, period)[n]
    // ... then the per-sample hit counters [B] (padded to 8 bytes) and hit lists [B][kMixCap] of (g, value)
    return sizeof(float) * ((((size_t)n_waveguides * (size_t)bufsize) + 1) & ~(size_t)1) + 2 * sizeof(int) * (size_t)n_waveguides +
           sizeof(int) * (((size_t)bufsize + 1) & ~(size_t)1) + sizeof(int2) * (size_t)gab::kMixCap * (size_t)bufsize;
}

int gab_dwg(const gab_waveguide_state* d_wg, float* d_fwd, float* d_bwd, const float* d_in,
            float* d_out, void* d_workspace, int n_waveguides, int bufsize, int max_len,
            int out_tracks, int variant, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!d_wg || !d_fwd || !d_bwd || !d_in || !d_out || !d_workspace)
            return gab::bad_arg("gab_dwg: null pointer");
        if (n_waveguides <= 0 || bufsize <= 0 || max_len <= 0) return gab::bad_arg("gab_dwg: sizes must be > 0");
        if (variant != GAB_DWG_NAIVE && variant != GAB_DWG_ACCEL) return gab::bad_arg("gab_dwg: unknown variant");
        static_assert(sizeof(gab_waveguide_state) == sizeof(gab::WG), "WaveguideState layout");
        hipStream_t s = gab::as_stream(stream);
        const gab::WG* wgs = reinterpret_cast<const gab::WG*>(d_wg);
        float* ws = static_cast<float*>(d_workspace);
        int2* hits = reinterpret_cast<int2*>(ws + (((size_t)n_waveguides * bufsize + 1) & ~(size_t)1));
        // the sparse mix pays from about two thousand waveguides on (8 192: 30.2 -> 10.7 us for the mix; at 128 its
        // extra launch costs more than the scan it replaces: 3.3 -> 8.8 us)
        const int n_mix = n_waveguides < out_tracks ? n_waveguides : out_tracks;
        const bool sparse = n_mix >= 2048;
        int* mix_count = sparse ? reinterpret_cast<int*>(hits + n_waveguides) : nullptr;
        int2* mix_list = reinterpret_cast<int2*>(reinterpret_cast<int*>(hits + n_waveguides) + (((size_t)bufsize + 1) & ~(size_t)1));
        if (variant == GAB_DWG_NAIVE) {
            gab::dwg_naive_kernel<<<(n_waveguides + 15) / 16, 16, 0, s>>>(wgs, d_fwd, d_bwd, d_in, ws, hits,
                                                                         n_waveguides, bufsize, max_len, mix_count);
        } else {
            int cells = max_len < bufsize ? max_len : bufsize;   // a buffer visits min(L, B) cells of a line
            constexpr int U = 8;
            if (sparse && bufsize <= gab::kDwgMaxB) {            // large banks: the cells kernel appends to the hit lists itself
                // The per-sample counters live in a pool of kDwgSlots zeroed arrays in the code object; a call takes the next
                // slot and its mix kernel zeroes it again.  What keeps a slot from being taken twice at once (more than
                // kDwgSlots calls in flight over several streams) is an event per slot: a call on ANOTHER stream goes behind
                // the slot's last user; a mix launch that fails zeroes the slot itself.
                struct Slot { hipEvent_t busy = nullptr; hipStream_t last = nullptr; bool used = false; };
                static Slot slots[gab::kDwgSlots];
                static std::mutex slots_mu;
                static unsigned next_slot = 0;
                // 0: round 5's (8 lines per workgroup, staged input)   1, 2: no staging, 8 / 16 lines (r06: no faster — it was
                // never the staging)   3, 4, 5, 6, 7: the lean kernel, 8 / 16 / 4 / 2 / 1 lines per workgroup: 15.5 / 19.6 / 14.0 /
                // 14.6 / 16.9 us against 19.6 (profiles/r06_dwg.md); delay lines of 4 GiB and more per array: form 0
                int form = 5;
#ifdef GAB_ABLATE
                if (getenv("GAB_DWG_FORM")) form = atoi(getenv("GAB_DWG_FORM"));     // diagnostic builds: A/B on one box
#endif
                int* pool = nullptr;
                GAB_HIP_CHECK(hipGetSymbolAddress(reinterpret_cast<void**>(&pool), HIP_SYMBOL(gab::g_dwg_count)));
                std::lock_guard<std::mutex> lock(slots_mu);
                const int slot = (int)(next_slot++ % (unsigned)gab::kDwgSlots);
                Slot& sl = slots[slot];
                if (!sl.busy) GAB_HIP_CHECK(hipEventCreateWithFlags(&sl.busy, hipEventDisableTiming));
                if (sl.used && sl.last != s) GAB_HIP_CHECK(hipStreamWaitEvent(s, sl.busy, 0));
                const unsigned long long line_bytes = 4ull * (unsigned long long)n_waveguides * (unsigned long long)max_len;
                if (line_bytes >= 0xffffff00ull && form >= 3) form = 0;      // (a buffer resource's range is 32 bits)
                const int UU = (form == 2 || form == 4) ? 16 : form == 5 ? 4 : form == 6 ? 2 : form == 7 ? 1 : U;
                dim3 grid((cells + 255) / 256, (n_waveguides + UU - 1) / UU);
#define GAB_DWG_APPEND(UV, ST) gab::dwg_cells_append_kernel<UV, ST><<<grid, 256, 0, s>>>(wgs, d_fwd, d_bwd, d_in, ws, hits, mix_list, \
                                                                                n_waveguides, n_mix, bufsize, max_len, slot)
#define GAB_DWG_LEAN(UV) gab::dwg_cells_lean_kernel<UV><<<grid, 256, 0, s>>>(wgs, d_fwd, d_bwd, d_in, ws, hits, mix_list, n_waveguides, n_mix, \
                                                                      bufsize, max_len, slot, (unsigned)line_bytes)
                if (form == 5) GAB_DWG_LEAN(4);
#ifdef GAB_ABLATE                                                // diagnostic builds: the forms that were measured and not kept
                else if (form == 7) GAB_DWG_LEAN(1);
                else if (form == 6) GAB_DWG_LEAN(2);
                else if (form == 4) GAB_DWG_LEAN(16);
                else if (form == 3) GAB_DWG_LEAN(U);
                else if (form == 2) GAB_DWG_APPEND(16, false);
                else if (form == 1) GAB_DWG_APPEND(U, false);
#endif
                else GAB_DWG_APPEND(U, true);
#undef GAB_DWG_LEAN
#undef GAB_DWG_APPEND
                int rc = gab::launch_status("dwg_cells_append_kernel");
                if (rc) return rc;                               // (nothing was appended: the slot is still zero)
                gab::dwg_mix_kernel<<<(bufsize + 3) / 4, 256, 0, s>>>(hits, ws, d_out, n_waveguides, bufsize, out_tracks,
                                                                      pool + (size_t)slot * gab::kDwgMaxB, mix_list, true);
                rc = gab::launch_status("dwg_mix_kernel");
                if (rc) (void)hipMemsetAsync(pool + (size_t)slot * gab::kDwgMaxB, 0, sizeof(int) * gab::kDwgMaxB, s);   // the appends stand: nobody else zeroes them
                sl.used = true;
                sl.last = s;
                GAB_HIP_CHECK(hipEventRecord(sl.busy, s));
                return rc;
            }
            if (n_waveguides >= 1024) {                          // U lines per workgroup (see the kernel)
                dim3 grid((cells + 255) / 256, (n_waveguides + U - 1) / U);
                gab::dwg_cells_multi_kernel<U><<<grid, 256, 0, s>>>(wgs, d_fwd, d_bwd, d_in, ws, hits, n_waveguides,
                                                                    bufsize, max_len, mix_count);
            } else {
                dim3 grid((cells + 255) / 256, n_waveguides);
                gab::dwg_cells_kernel<<<grid, 256, 0, s>>>(wgs, d_fwd, d_bwd, d_in, ws, hits, n_waveguides,
                                                           bufsize, max_len, mix_count);
            }
        }
        int rc = gab::launch_status("dwg kernel");
        if (rc) return rc;
        if (sparse) {
            gab::dwg_gather_kernel<<<(n_mix + 255) / 256, 256, 0, s>>>(hits, ws, mix_count, mix_list, n_mix, bufsize);
            rc = gab::launch_status("dwg_gather_kernel");
            if (rc) return rc;
        }
        gab::dwg_mix_kernel<<<(bufsize + 3) / 4, 256, 0, s>>>(hits, ws, d_out, n_waveguides, bufsize,
                                                              out_tracks, mix_count, mix_list, false);
        return gab::launch_status("dwg_mix_kernel");
    });
}

}  // extern "C"
