// k_conv_accel.hip — FFT convolution for gfx950 (the north-star pipeline).
//
// Replaces Conv1DAccelBenchmark's device pipeline (cuda/bench_conv1d_accel.cu):
//   cudaMemset + T x cudaMemcpy D2D (:265-274) -> cufftExecR2C (:276) ->
//   ComplexMultiplyKernel (:9-30, :281-285) -> cufftExecC2R (:289) ->
//   ExtractRealPartKernel (:32-47, :294-298)
// and precomputeImpulseResponseFFTs (:175-228), with ONE kernel per buffer.
// Result contract = the CPU golden conv1DCPUReference (:234-252):
//   y[T*s + t] = sum_{k<=s, k<L} x[t*B + s-k] * h[t*L + k]   (sample-major out)
// which is what the streaming kernel produces on its first buffer after a
// reset; later buffers carry the history the reference computes
// (overlap_size_, :53) but never uses.
//
// Algorithm (B = 512): two-partition overlap-save.
//   partition A: taps [0,512)      N=1024 window = [previous block | new block]
//   partition B: taps [512,4096)   N=4096 window = the 8 previous blocks
// Both keep their LAST 512 outputs; partition B does not depend on the new
// block at all.  State per channel is the time-domain ring of the last 8 blocks
// (16 KiB) — no frequency-domain delay line — so HBM traffic per buffer is
//   history 16 KiB (read) + 2 KiB (write) + new 2 KiB + out 2 KiB
//   + spectra (513 + 2049) bins * 8 B = 20 KiB            per channel
// = 42.5 KiB against the 36 KiB algorithmic figure (in + out + L taps + L history).
//
// Two channels share one complex transform: z = x_a + i x_b.  With Z' =
// conj(Z[N-k]) the spectrum of the packed output is W = Z*P + Z'*M where
// P = (Ha+Hb)/2, M = (Ha-Hb)/2 (1/N folded in); real(ifft W) = y_a,
// imag = y_b.  One workgroup (256 threads) = one channel pair; the 4096-point
// transform is three radix-16 passes, the 1024-point one five radix-4 passes.
//
// Streaming plans at the headline shape use a different cut of the same taps,
// conv_split_kernel below ("split roles"): taps [0,512) + [512,1024) on one
// workgroup, taps [1024,4096) every other buffer and one buffer ahead on another.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdlib>
#include <algorithm>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include "gab_common.hpp"
#include "gab_fft.hpp"

namespace gab {
namespace fft {

void build_twiddles(float* t) {
    for (int m = 0; m < kTwiddleN; ++m) {
        double a = -2.0 * M_PI * (double)m / (double)kTwiddleN;
        t[2 * m] = (float)std::cos(a);
        t[2 * m + 1] = (float)std::sin(a);
    }
}

// One table per device, created on first use, never freed (process lifetime).
const cf* device_twiddles() {
    static std::mutex mu;
    static std::vector<cf*> per_device(64, nullptr);
    int dev = 0;
    GAB_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (dev >= (int)per_device.size()) per_device.resize(dev + 1, nullptr);
    if (!per_device[dev]) {
        std::vector<float> host(2 * kTwiddleN);
        build_twiddles(host.data());
        cf* d = nullptr;
        GAB_HIP_CHECK(hipMalloc(&d, sizeof(cf) * kTwiddleN));
        GAB_HIP_CHECK(hipMemcpy(d, host.data(), sizeof(cf) * kTwiddleN, hipMemcpyHostToDevice));
        per_device[dev] = d;
    }
    return per_device[dev];
}

}  // namespace fft

using fft::cf;
using fft::mk;

namespace {

constexpr int kB = 512;            // block (buffer) size of the fused path
constexpr int kNA = 1024;          // partition A transform
constexpr int kNB = 4096;          // partition B transform
constexpr int kSlots = 8;          // history ring: kNB / kB blocks
constexpr int kBinsA = kNA / 2 + 1;
constexpr int kBinsB = kNB / 2 + 1;
constexpr int kThreads = 256;

using PadB = fft::Pad<16>;
constexpr int kLdsHalf = PadB::size(kNB);       // cf entries per LDS buffer

// Partner exchange: every thread publishes its R bins (k = tid + r*N/R) and
// fetches conj(Z[(N-k) mod N]).  Threads beyond N/R only meet the barrier.
template <int N, int R, bool RAW = false>
__device__ __forceinline__ void partner_exchange(const cf (&z)[R], cf (&zp)[R],
                                                 cf* __restrict__ lds, int tid, bool active = true) {
    constexpr int NT = N / R;
    if (active) {
#pragma unroll
        for (int r = 0; r < R; ++r) lds[tid + r * NT] = z[r];
    }
    __syncthreads();
    if (active) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int k = tid + r * NT;
            cf v = lds[(N - k) & (N - 1)];
            zp[r] = RAW ? v : fft::conj(v);      // RAW: the caller folds the conjugate into its product
        }
    }
}

// The bank stores k <= N/2 only; the upper half is the conjugate (P and M come
// from Hermitian spectra).  Loads are split from the arithmetic so a caller can
// issue them early and let HBM stream underneath a transform.
// Bin k = tid + r*NT lies in the stored half (k <= N/2) for r < R/2, in the
// conjugate half for r > R/2; only r == R/2 depends on the thread (k = N/2 for tid 0).
template <int N, int R>
__device__ __forceinline__ void load_spectra(float4 (&c)[R], const float4* __restrict__ pm, int tid) {
    constexpr int NT = N / R;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int k = tid + r * NT;
        if (r < R / 2) c[r] = pm[k];
        else if (r > R / 2) c[r] = pm[N - k];
        else c[r] = pm[tid == 0 ? k : N - k];
    }
}

// W[k] = Z[k]*P[k] + Z'[k]*M[k], Z' = conj(zraw) (zraw = Z[N-k] as fetched, unconjugated)
template <int N, int R>
__device__ __forceinline__ void spectral_product(cf (&z)[R], const cf (&zraw)[R], const float4 (&c)[R],
                                                 int tid) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        cf P = mk(c[r].x, c[r].y), M = mk(c[r].z, c[r].w);
        if (r > R / 2 || (r == R / 2 && tid != 0)) {       // conjugate half: z*conj(P) + conj(zraw*M)
            z[r] = fft::cfma_cjcj(zraw[r], M, fft::cmulc(z[r], P));
        } else {
            z[r] = fft::cfma_cj(zraw[r], M, fft::cmul(z[r], P));
        }
    }
}

template <typename T>
__device__ __forceinline__ void keep_alive(const T& v) {
    const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) asm volatile("" ::"v"(f[i]));
}

#ifdef GAB_ABLATE
// split kernel (debug bit 64): [launch parity][block][8] of s_memrealtime; slot 6 = HW_ID, 7 = XCC_ID.
// Bit 128 additionally drains the wave's loads (s_waitcnt vmcnt(0)) before slot 1, so that slot 1
// reads "all of this wave's data has arrived" (that changes the timing it measures).
__device__ unsigned long long g_split_stamps[2 * 8 * 8192];
#define GAB_SSTAMP(i)                                                                          \
    do {                                                                                       \
        if (sp.debug & 64) {                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            if (threadIdx.x == 0)                                                              \
                g_split_stamps[((head & 1) * 8192 + (sp.debug >> 20) + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
            __builtin_amdgcn_sched_barrier(0);                                                 \
        }                                                                                      \
    } while (0)
#define GAB_SDRAIN() do { if (sp.debug & 128) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } while (0)
#else
#define GAB_SSTAMP(i) do {} while (0)
#define GAB_SDRAIN() do {} while (0)
#endif

// (Tried and measured slower, so not kept: running partition B first in half of
// the workgroups to de-phase the two workgroups that share a CU, 14.8 vs 14.0 us;
// a 512-thread radix-8 form of this kernel (8 values per thread, 4 passes), 16.6 us:
// the transforms are bound by LDS write bandwidth and VALU throughput, not by the
// per-thread instruction chain, and radix-8 needs a third exchange.)
// One buffer of one channel pair; the body of the kernels below.
// LOOPED: the caller walks several buffers through this body in one launch (conv_batch_kernel).  Everything that depends
// on the thread index alone (both partitions' twiddle powers, 84 registers; addresses) is loop-invariant there and the
// compiler kept it across the whole body — 256 registers and 24 bytes of scratch per lane.  With an opaque copy of the
// thread index per buffer the body is the single-buffer launch's (197 registers, the powers re-formed per buffer as
// there) and nothing spills.
template <bool STREAM, bool TAIL, bool LOOPED = false>
__device__ __forceinline__ void conv_one_buffer(
    const float* __restrict__ in, float* __restrict__ out, float* __restrict__ hist,
    const float4* __restrict__ pmA, const float4* __restrict__ pmB,
    const cf* __restrict__ tw, int T, int head, cf* __restrict__ lds) {
    cf* const lds0 = lds;
    cf* const lds1 = lds + kLdsHalf;

    int tid_ = threadIdx.x;
    if constexpr (LOOPED) asm volatile("" : "+v"(tid_));
    const int tid = tid_;
    const int q = xcd_contiguous(blockIdx.x, gridDim.x);
    const int ta = 2 * q, tb = 2 * q + 1;
    const bool hasb = tb < T;

    // history ring of this pair: [slot][sample] complex (channel a, channel b)
    cf* const hp = reinterpret_cast<cf*>(hist) + (size_t)q * kSlots * kB;
    using FA = fft::BlockFFT<kNA, 4, false>;
    using FAi = fft::BlockFFT<kNA, 4, true>;
    using FB = fft::BlockFFT<kNB, 16, false>;
    using FBi = fft::BlockFFT<kNB, 16, true>;

    cf zb[16];
    cf za[4];
    float4 ca[4];
    typename FA::Twiddles twa;      // all powers precomputed while the first loads are in flight
    typename FB::Bases twb_base;    // partition B's are expanded after A (60 registers)

    // ---- requests, in need order: partition A's inputs (new block, previous block,
    // its spectra) first, then the older history partition B works on, so A's
    // transform runs while the rest of the window streams in.
    {
        const float* xa = in + (size_t)ta * kB;
        const float* xb = in + (size_t)tb * kB;
        za[2] = mk(xa[tid], hasb ? xb[tid] : 0.0f);
        za[3] = mk(xa[tid + kThreads], hasb ? xb[tid + kThreads] : 0.0f);
    }
    if constexpr (STREAM) {
        const int off = ((head + kSlots - 1) & (kSlots - 1)) * kB + tid;
        za[0] = hp[off];
        za[1] = hp[off + kThreads];
    } else {
        za[0] = mk(0.0f, 0.0f);
        za[1] = mk(0.0f, 0.0f);
    }
    __builtin_amdgcn_sched_barrier(0);      // keep the request order: a wave's loads return in order
    load_spectra<kNA, 4>(ca, pmA + (size_t)q * kBinsA, tid);
    __builtin_amdgcn_sched_barrier(0);
    typename FA::Bases twa_base;
    FA::load_twiddles(twa_base, tw, tid);
    if constexpr (STREAM && TAIL) {
        FB::load_twiddles(twb_base, tw, tid);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 14; ++r) {
            zb[r] = hp[((head + (r >> 1)) & (kSlots - 1)) * kB + (r & 1) * kThreads + tid];      // oldest first
        }
        zb[14] = za[0];
        zb[15] = za[1];
    }
    __builtin_amdgcn_sched_barrier(0);
    FA::expand_twiddles(twa_base, twa);      // while the first loads are in flight
    if constexpr (STREAM) {
        // overwrite the oldest block (already requested into zb[0..1]) with the new one
        hp[head * kB + tid] = za[2];
        hp[head * kB + kThreads + tid] = za[3];
    }
    // Two of these workgroups share a CU and the one dispatched second (blockIdx + grid/2
    // on a 512-workgroup grid) gets its first data ~1 us later and then loses most issue
    // arbitration to the older waves: it finished 2.6 us after its neighbour.  Raising its
    // priority evens the pair out (11.8 -> 11.55 us per launch); delaying either one by any
    // amount only cost time.
    if (blockIdx.x >= gridDim.x / 2) __builtin_amdgcn_s_setprio(1);

    // ---- stages.  LDS hand-over is barrier-free by construction: a stage called
    // with (X, Y) first writes X and makes its last reads from X, so the next
    // stage is called with (Y, X) — Y's last readers are behind a barrier by then.
    float ya0 = 0.f, yb0 = 0.f, ya1 = 0.f, yb1 = 0.f;
    // (Partition B's spectra are requested when B starts.  Asked for up front they queue
    // in the CU's request path ahead of other waves' FIRST loads: first data +1 us, 14.0 us
    // per launch; trickled out two loads per pass of A: 12.1 us; at B's start: 11.8 us.)
    auto part_a = [&](cf* X, cf* Y) {
        FA::run(za, X, Y, twa, tid);                         // 5 passes: last reads Y
        cf zpa[4];
        partner_exchange<kNA, 4, true>(za, zpa, X, tid);
        spectral_product<kNA, 4>(za, zpa, ca, tid);
        FAi::run(za, Y, X, twa, tid);                        // last reads X
        ya0 += za[2].x; yb0 += za[2].y;
        ya1 += za[3].x; yb1 += za[3].y;
    };
    auto part_b = [&](cf* X, cf* Y) {
        if constexpr (STREAM && TAIL) {
            // request the spectra now; the scheduling barrier keeps the loads ahead of
            // the transform so HBM streams underneath it
            float4 cb[16];
            load_spectra<kNB, 16>(cb, pmB + (size_t)q * kBinsB, tid);
            __builtin_amdgcn_sched_barrier(0);
            typename FB::Twiddles twb;
            FB::expand_twiddles(twb_base, twb);
            FB::run(zb, X, Y, twb, tid);                     // 3 passes: last reads Y
            cf zpb[16];
            partner_exchange<kNB, 16, true>(zb, zpb, X, tid);
            spectral_product<kNB, 16>(zb, zpb, cb, tid);
            FBi::template run<typename FB::Twiddles, 2>(zb, Y, X, twb, tid);   // last reads X; only [14],[15]
            ya0 += zb[14].x; yb0 += zb[14].y;
            ya1 += zb[15].x; yb1 += zb[15].y;
        }
    };
    part_a(lds0, lds1);
    part_b(lds1, lds0);

    // ---- scatter: sample-major out[T*s + t], s = tid and tid+256 -----------
    float* o0 = out + (size_t)T * tid + ta;
    float* o1 = out + (size_t)T * (tid + kThreads) + ta;
    if (hasb && (T & 1) == 0) {
        *reinterpret_cast<float2*>(o0) = make_float2(ya0, yb0);
        *reinterpret_cast<float2*>(o1) = make_float2(ya1, yb1);
    } else {
        o0[0] = ya0;
        o1[0] = ya1;
        if (hasb) { o0[1] = yb0; o1[1] = yb1; }
    }
}

template <bool STREAM, bool TAIL>
__global__ __launch_bounds__(kThreads, 2) void conv_overlap_save_kernel(
    const float* __restrict__ in, float* __restrict__ out, float* __restrict__ hist,
    const float4* __restrict__ pmA, const float4* __restrict__ pmB,
    const cf* __restrict__ tw, int T, int head) {
    __shared__ cf lds[2 * kLdsHalf];
    conv_one_buffer<STREAM, TAIL>(in, out, hist, pmA, pmB, tw, T, head, lds);
}

// ---- split roles: near and far partitions on different workgroups ---------------------------
// The same convolution with the taps cut so that no workgroup runs a short and a long transform
// one after the other (that chain — first data, near transform, far transform, store — is what a
// launch of conv_overlap_save_kernel lasts):
//   A   taps [0,512)       N=1024, window [block k-1 | block k]
//   A2  taps [512,1024)    N=1024, window [block k-2 | block k-1]; its spectral product joins
//                          A's before ONE inverse
//   F   taps [1024,4096)   N=4096, window [block k-7 .. block k]; keeps its last 1024 outputs,
//                          which are blocks k+1 and k+2 (taps >= 1024 cannot reach further), and
//                          parks them in a four-slot carry ring, slot = block & 3.
// F therefore runs for a channel pair every OTHER buffer, one buffer ahead of its use.  Two
// channel pairs form a duo with two workgroups on one CU (blockIdx b and b + grid/2):
//   near workgroup: A and A2 of BOTH pairs, every launch — four 1024-point transforms held one
//       per wave (no workgroup barrier but one), then the two inverses on waves 0 and 2; adds the
//       parked far share of block k, writes the output and the history ring;
//   far workgroup: F of pair 0 on even launches, of pair 1 on odd launches.
// Nothing in a launch waits for anything else in it.  Bytes per launch: 37 MB instead of 44 MB.
struct ConvSplit {
    const float4* pmA2;    // [pairs][kBinsA]   taps [512,1024)
    const float4* pmF;     // [pairs][kBinsB]   taps [1024,4096)
    cf* carry;             // [pairs][4][512]   F's outputs, slot = block & 3
#ifdef GAB_ABLATE
    int debug;             // diagnostic builds only (GAB_CONV_SPLIT_DEBUG): role / stage ablations, stamps
#endif
};

// Stage ablations exist in diagnostic builds only (-DGAB_ABLATE, libgab_hip_ablate.so): the product
// kernels have no path that skips work.
#ifdef GAB_ABLATE
#define GAB_SDBG(bit) ((sp.debug & (bit)) != 0)
#else
#define GAB_SDBG(bit) false
#endif

constexpr int kCarrySlots = 4;
using PadA16 = fft::Pad<16>;
constexpr int kWaveImg = PadA16::size(kNA);     // LDS image of one wave-held 1024-point transform

__device__ __forceinline__ void conv_split_buffer(
    const float* __restrict__ in, float* __restrict__ out, float* __restrict__ hist,
    const float4* __restrict__ pmA, const ConvSplit& sp, const cf* __restrict__ tw, int T, int head,
    cf* __restrict__ lds) {
    const int tid = threadIdx.x;
    const int duos = gridDim.x / 2;
    // near workgroups first in dispatch order, equal priority: measured best (far first 10.7 us,
    // far at raised priority 10.7, near at raised priority 9.4, as is 9.4)
    // Roles alternate in runs of 256 workgroups (one per CU), so that whatever part of a large
    // grid is resident holds as many near as far workgroups, one of each per CU; a 512-workgroup
    // grid is simply near half, far half.  Grids that are not a multiple of 512: halves.
    bool far;                                                         // uniform over the workgroup
    int slot;                                                         // index within the role
    if ((gridDim.x & 511) == 0) {
        const int run = blockIdx.x >> 8;
        far = ((run & 1) != 0) != GAB_SDBG(256);                      // diagnostic bit 256: far workgroups first
        slot = (run >> 1) * 256 + (blockIdx.x & 255);
    } else {
        far = ((int)blockIdx.x >= duos) != GAB_SDBG(256);
        slot = (int)blockIdx.x >= duos ? blockIdx.x - duos : blockIdx.x;
    }
    const int d = xcd_contiguous(slot, duos);

    GAB_SSTAMP(0);
#ifdef GAB_ABLATE
    if (GAB_SDBG(64) && threadIdx.x == 0) {
        g_split_stamps[((head & 1) * 8192 + (sp.debug >> 20) + blockIdx.x) * 8 + 6] = __builtin_amdgcn_s_getreg(63492);   // HW_ID
        g_split_stamps[((head & 1) * 8192 + (sp.debug >> 20) + blockIdx.x) * 8 + 7] = __builtin_amdgcn_s_getreg(63508);   // XCC_ID
    }
#endif
    if (far) {
        if (GAB_SDBG(4)) return;
        if (GAB_SDBG(512)) __builtin_amdgcn_s_setprio(2);
        // ---- F of one pair: window = the seven newest blocks of the ring + the new block
        const int q = 2 * d + (head & 1);
        const int ta = 2 * q, tb = ta + 1;
        const cf* const hp = reinterpret_cast<const cf*>(hist) + (size_t)q * kSlots * kB;
        cf* const cp = sp.carry + (size_t)q * kCarrySlots * kB;
        using FB = fft::BlockFFT<kNB, 16, false>;
        using FBi = fft::BlockFFT<kNB, 16, true>;
        cf* const X = lds;
        cf* const Y = lds + kLdsHalf;
        cf zb[16];
        typename FB::Bases twb_base;
        {
            const float* xa = in + (size_t)ta * kB;
            const float* xb = in + (size_t)tb * kB;
            zb[14] = mk(xa[tid], xb[tid]);
            zb[15] = mk(xa[tid + kThreads], xb[tid + kThreads]);
        }
        FB::load_twiddles(twb_base, tw, tid);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 14; ++r)                                  // blocks k-7 .. k-1, oldest first
            zb[r] = hp[((head + 1 + (r >> 1)) & (kSlots - 1)) * kB + (r & 1) * kThreads + tid];
        __builtin_amdgcn_sched_barrier(0);
        float4 cb[16];
        load_spectra<kNB, 16>(cb, sp.pmF + (size_t)q * kBinsB, tid);
        __builtin_amdgcn_sched_barrier(0);
        typename FB::Twiddles twb;
        FB::expand_twiddles(twb_base, twb);
        GAB_SDRAIN();
        GAB_SSTAMP(1);
        FB::run(zb, X, Y, twb, tid);
        GAB_SSTAMP(2);
        cf zpb[16];
        partner_exchange<kNB, 16, true>(zb, zpb, X, tid);
        spectral_product<kNB, 16>(zb, zpb, cb, tid);
        GAB_SSTAMP(3);
        FBi::template run<typename FB::Twiddles, 4>(zb, Y, X, twb, tid);     // only [12..15]
        GAB_SSTAMP(4);
        cf* const c1 = cp + ((head + 1) & (kCarrySlots - 1)) * kB;            // block k+1
        cf* const c2 = cp + ((head + 2) & (kCarrySlots - 1)) * kB;            // block k+2
        if (GAB_SDBG(16)) { keep_alive(zb[12]); keep_alive(zb[13]); keep_alive(zb[14]); keep_alive(zb[15]); return; }
        c1[tid] = zb[12];
        c1[tid + kThreads] = zb[13];
        c2[tid] = zb[14];
        c2[tid + kThreads] = zb[15];
        GAB_SSTAMP(5);
        return;
    }

    if (GAB_SDBG(1)) return;
    if (GAB_SDBG(1024)) __builtin_amdgcn_s_setprio(2);
    // ---- near: wave w holds one 1024-point transform: pair (w >> 1) of the duo, window w & 1
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = 2 * d + (w >> 1);
    const int ta = 2 * q, tb = ta + 1;
    const bool second = (w & 1) != 0;
    cf* const hp = reinterpret_cast<cf*>(hist) + (size_t)q * kSlots * kB;
    cf* const img = lds + w * kWaveImg;
    const int s1 = ((head + kSlots - 1) & (kSlots - 1)) * kB;         // block k-1
    const int s2 = ((head + kSlots - 2) & (kSlots - 1)) * kB;         // block k-2
    using WF = fft::WaveFFT1024<false>;
    using WFi = fft::WaveFFT1024<true>;

    WF::Twiddles t;
    WF::load_twiddles_raw(t, tw, lane);          // requested first: a wave's loads return in order, and
                                                 // the forward transform needs these before the spectra
    cf z[16];
    if (!second) {
        const float* xa = in + (size_t)ta * kB;
        const float* xb = in + (size_t)tb * kB;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[8 + j] = mk(xa[lane + 64 * j], xb[lane + 64 * j]);
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = hp[s1 + lane + 64 * j];
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = hp[s2 + lane + 64 * j];
#pragma unroll
        for (int j = 0; j < 8; ++j) z[8 + j] = hp[s1 + lane + 64 * j];
    }
    __builtin_amdgcn_sched_barrier(0);
    float4 c[16];
    load_spectra<kNA, 16>(c, (second ? sp.pmA2 : pmA) + (size_t)q * kBinsA, lane);
    __builtin_amdgcn_sched_barrier(0);
    WF::expand_twiddles(t);
    if (!second && !GAB_SDBG(8)) {                                 // the new block enters the ring
#pragma unroll
        for (int j = 0; j < 8; ++j) hp[head * kB + lane + 64 * j] = z[8 + j];
    }
    GAB_SDRAIN();
    GAB_SSTAMP(1);
    WF::run(z, img, t, lane);
    GAB_SSTAMP(2);
    const unsigned rb = (unsigned)lane + ((unsigned)lane >> 4);      // Pad(lane + 64 r) = rb + 68 r
    {
        cf zp[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) img[rb + 68 * r] = z[r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) zp[r] = img[PadA16::at((kNA - (lane + 64 * r)) & (kNA - 1))];
        spectral_product<kNA, 16>(z, zp, c, lane);
    }
    __builtin_amdgcn_wave_barrier();
    if (second) {
#pragma unroll
        for (int r = 0; r < 16; ++r) img[rb + 68 * r] = z[r];
    }
    __syncthreads();
    GAB_SSTAMP(3);
    cf y[8];
    const bool pieces8 = GAB_SDBG(2048);      // A/B: one float2 per pair and sample, no swap
    if (pieces8 && second) return;
    if (!second) {
        {
            const cf* const other = img + kWaveImg;                   // the A2 transform of the same pair
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = fft::cadd(z[r], other[rb + 68 * r]);
        }
        const cf* const cy = sp.carry + ((size_t)q * kCarrySlots + (head & (kCarrySlots - 1))) * kB;
        cf park[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) park[j] = cy[lane + 64 * j];
        WFi::run(z, img, t, lane);
        GAB_SSTAMP(4);
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = fft::cadd(z[8 + j], park[j]);
        if (pieces8) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                *reinterpret_cast<float2*>(out + (size_t)T * (lane + 64 * j) + ta) = make_float2(y[j].x, y[j].y);
            return;
        }
        // The two pairs of a duo are four neighbouring channels: 16 bytes per sample.  The waves
        // swap halves through LDS so that each stores float4 pieces — half as many partial-line
        // writes into L2 as with one float2 per pair (the output scatter is 8 KiB-strided).
        // Wave 0 (pair 0) keeps samples lane + 64 j, j < 4, wave 2 (pair 1) keeps j >= 4.
        if (w == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) img[lane + 64 * j] = y[4 + j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) img[lane + 64 * j] = y[j];
        }
    }
    __syncthreads();
    if (second) return;
    if (GAB_SDBG(32)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) keep_alive(y[j]);
        return;
    }
    {
        float* const o0 = out + 4 * (size_t)d;
        if (w == 0) {
            const cf* const other = lds + 2 * kWaveImg;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const cf theirs = other[lane + 64 * j];
                *reinterpret_cast<float4*>(o0 + (size_t)T * (lane + 64 * j)) = make_float4(y[j].x, y[j].y, theirs.x, theirs.y);
            }
        } else {
            const cf* const other = lds;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const cf theirs = other[lane + 64 * j];
                *reinterpret_cast<float4*>(o0 + (size_t)T * (lane + 64 * (4 + j))) = make_float4(theirs.x, theirs.y, y[4 + j].x, y[4 + j].y);
            }
        }
    }
    GAB_SSTAMP(5);
}

__global__ __launch_bounds__(kThreads, 2) void conv_split_kernel(
    const float* __restrict__ in, float* __restrict__ out, float* __restrict__ hist,
    const float4* __restrict__ pmA, ConvSplit sp, const cf* __restrict__ tw, int T, int head) {
    __shared__ cf lds[2 * kLdsHalf];
    conv_split_buffer(in, out, hist, pmA, sp, tw, T, head, lds);
}


// The same kernel under its own name for buffers that live in pinned host memory (the kernel then
// moves them over the link itself): such launches run at link speed, and profilers average per
// kernel name — this keeps them out of the figures of the HBM-resident launches.
template <bool TAIL>
__global__ __launch_bounds__(kThreads, 2) void conv_overlap_save_host_io_kernel(
    const float* __restrict__ in, float* __restrict__ out, float* __restrict__ hist,
    const float4* __restrict__ pmA, const float4* __restrict__ pmB,
    const cf* __restrict__ tw, int T, int head) {
    __shared__ cf lds[2 * kLdsHalf];
    conv_one_buffer<true, TAIL>(in, out, hist, pmA, pmB, tw, T, head, lds);
}

__global__ __launch_bounds__(kThreads, 2) void conv_split_host_io_kernel(
    const float* __restrict__ in, float* __restrict__ out, float* __restrict__ hist,
    const float4* __restrict__ pmA, ConvSplit sp, const cf* __restrict__ tw, int T, int head) {
    __shared__ cf lds[2 * kLdsHalf];
    conv_split_buffer(in, out, hist, pmA, sp, tw, T, head, lds);
}

// ---- the real-time round trip: one buffer, pinned host in -> pinned host out, both link directions busy at once ----
// gab_conv_round_trip (classic cut).  What crosses the link is 2 MiB each way at C3; the reference moves them one after
// the other around its kernels (cuda/bench_base.cu:30-42, bench_conv1d_accel.cu:258-304).  Measured on this part
// (tools/ubench/link_duplex, link_modes): a kernel reads pinned memory at 50 GB/s and writes it at 51 GB/s, but doing
// both at once it gets 27-29 GB/s each way; a copy-engine upload beside a kernel that writes pinned memory runs both at
// full rate; every extra engine copy costs ~10 us.  Hence:
//   up    ONE engine copy of the whole input into `stage` (fine-grained device memory, so that a running kernel sees
//         it land), on the plan's own stream, started before the kernel;
//   wait  the kernel is launched at once: every workgroup (one channel pair) first runs the far partition — taps
//         [512,4096) see the eight PREVIOUS blocks only — and then polls for its two rows.  A consumed word is
//         overwritten with a sentinel (a NaN no audio stream carries), so "my words are no longer the sentinel" means
//         they have landed; a buffer that really holds the sentinel is released by the `landed` word the host sets
//         once the copy's event has completed — slower, never wrong;
//   down  outputs are parked sample-major in device memory by write-through stores; the workgroup whose arrival
//         completes a channel group (the copy lands groups in order) drains the group's slab to the pinned output in
//         whole rows of the group's width, so the link carries 256-512-byte pieces while later groups are still landing;
//   done  the call returns when the LAUNCH HAS ENDED (its own stop event: hipExtLaunchKernelGGL + hipEventQuery).  The drain that
//         completes the last group also writes the epoch to a pinned word: a hint that tells the spinning host when to go
//         and ask the event, nothing more.
// What each hand-off rests on (round 5; profiles/r05_roundtrip_protocol.md has the reasoning and the record it answers):
//   upload -> kernel   a naturally aligned 32-bit word is single-copy atomic (HSA), so a word read as "not the sentinel" is
//         the input word — given that the copy writes every word ONCE with its final value (one linear engine copy from
//         pinned memory; inputs that are not pinned are uploaded completely before the launch).  That is the one
//         assumption the overlap needs and it is checkable after the fact: the consumed block is the plan's newest
//         history block (gab_conv_newest_block; tests and tools/roundtrip_stress.py compare it with h_in).  A word
//         that really IS the sentinel is accepted only behind an acquire of `landed`, which the host releases after it has
//         seen the copy's completion event (HSA: a completed copy's writes are visible at system scope).
//   re-arm -> next upload   the sentinel stores belong to a launch that has ended before the call returns, and the next
//         call's copy is submitted after that: ordered by the launch's completion, not by a counter.
//   rows -> host   the rows are the launch's stores to pinned memory; the host reads them after the launch's completion
//         signal (HIP: everything a kernel wrote is visible to the host once the launch's event has completed).
// Same operations in the same order as conv_overlap_save_kernel<true, true>: bit-identical to device-buffer launches.
constexpr unsigned kRtSentinel = 0xffa5c3e1u;       // a negative NaN with a payload
// "Has this word landed?" looks at the word's TOP BYTE only: not the sentinel's 0xff.  An engine copy that is cut inside a word
// (the runtime cuts at 4 MiB - 1 bytes; ours go out in word-aligned pieces, but the limit is the runtime's to change) leaves, for
// a moment, the input's low bytes under the sentinel's high ones — still "not landed" by this test, where a whole-word compare took
// it for a sample (profiles/r05_incident_torn_word.txt).  An input word whose own top byte is 0xff (a negative NaN, -inf, below
// -1.7e38: no audio) waits for the upload's completion like the sentinel itself.
__device__ __forceinline__ bool rt_pending(unsigned w) { return (w >> 24) == (kRtSentinel >> 24); }
constexpr size_t kRtUploadPiece = (size_t(4) << 20) - 256;   // bytes per engine copy of an upload that a kernel consumes as it lands (below the runtime's 4 MiB - 1 packet limit, a multiple of 256)
constexpr int kRtCompletion = 2;                   // how gab_conv_round_trip observes the launch's end (see there)
constexpr int kRtPollLimit = 1 << 21;              // x ~0.5 us of s_sleep: about a second, then the launch gives up
constexpr int kRtMaxGroups = 40;
[[maybe_unused]] constexpr int kEngineWaves = 12;  // gab_conv_engine_start: conv_split_engine12_kernel; diagnostic builds: GAB_ENGINE_WAVES=8 -> round 5's conv_split_engine_kernel
[[maybe_unused]] constexpr int kBatchWaves = 12;   // gab_conv_process_batch on a split plan: conv_split_batch12_kernel; diagnostic builds: GAB_BATCH_WAVES=8 / 6 / 64 / 26
constexpr size_t kBatchChunk = 256;                // buffers per conv_split_batch12_kernel launch at most (see gab_conv_process_batch)
struct ConvRoundTrip {
    unsigned* stage;                  // [T*B] fine-grained device memory
    float* park;                      // [B*T] device memory
    float* h_out;                     // [B*T] pinned host memory
    unsigned* counters;               // device: [32 g + {0,1,2}] arrivals / claims / announced epoch of group g, [32 groups] shares drained
    unsigned* done;                   // pinned host: the epoch, once the last share has been drained (a HINT: the call waits for the launch's end)
    const unsigned* landed;           // pinned host: the epoch, once the host has seen the upload complete
    unsigned* error;                  // pinned host: bit 0 a wait ran out, bit 1 a consumed word is not what the completed upload left (kRtErrTorn)
    unsigned epoch;
    int groups;
    unsigned bound[kRtMaxGroups + 1];   // group g = pairs [bound[g], bound[g + 1]): equal groups, the first and the last cut finer (see gab_conv_round_trip_init)
};

#ifdef GAB_ABLATE
// diagnostic builds: s_memrealtime (100 MHz) per channel group — [g][0] first workgroup enters, [1] last workgroup has
// its rows, [2] last workgroup parked (= the drain starts), [3] drain done; [64][0] the completion word is written
__device__ unsigned long long g_rt_stamps[65 * 4];
#define GAB_RT_STAMP_MIN(g, i) do { if (threadIdx.x == 0) atomicMin(&g_rt_stamps[(g) * 4 + (i)], (unsigned long long)__builtin_amdgcn_s_memrealtime()); } while (0)
#define GAB_RT_STAMP_MAX(g, i) do { if (threadIdx.x == 0) atomicMax(&g_rt_stamps[(g) * 4 + (i)], (unsigned long long)__builtin_amdgcn_s_memrealtime()); } while (0)
#else
#define GAB_RT_STAMP_MIN(g, i) do {} while (0)
#define GAB_RT_STAMP_MAX(g, i) do {} while (0)
#endif

__device__ __forceinline__ unsigned rt_peek(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Round 6: what a workgroup consumed EARLY is checked against what the COMPLETED upload left.  The hand-off above rests on two
// observations — an engine packet writes a naturally aligned word whole, and a word never shows an intermediate value — and
// a violation was silent wrong audio (profiles/r05_incident_torn_word.txt was one: found by a stress run, not by the
// call).  The consumed words stay in the staging buffer — the kernel no longer puts the sentinel back — and a second, small
// launch behind it on the same stream, ordered behind the UPLOAD'S OWN COMPLETION EVENT (hipStreamWaitEvent: a completed
// copy's writes are final and visible), reads every word again, compares it with what the kernel consumed (the plan's
// newest history block: the consumed words, bit for bit) and only then puts the sentinel back: conv_round_trip_check_kernel.
// Why not inside the kernel: the earliest statement "the upload is complete" that reaches a running kernel — the host's
// release of `landed` behind hipEventQuery, or a four-byte copy queued behind the upload's last piece — arrives 14-20 us
// after the last byte (both go through the command processor), and a launch that waits for it ends that much later:
// 87-90 us per call against 69 (profiles/r06_roundtrip_check.txt).  The check launch costs the CALL nothing: the call
// returns on the main launch's end as before; the check's verdict is read by the next call on the plan (which waits for it:
// its upload must not meet the re-arming stores), by gab_conv_round_trip_check, or — gab_conv_round_trip_set_check(plan, 2) —
// by the call itself, which then returns ~10 us later and reports a torn word AT the failing call.
constexpr unsigned kRtErrWait = 1u, kRtErrTorn = 2u;

// the consumed words (the newest block of the history ring, as conv_round_trip_kernel stored them) against the staging buffer
// after the upload's completion; then the sentinel goes back for the next buffer
__global__ __launch_bounds__(kThreads) void conv_round_trip_check_kernel(unsigned* __restrict__ stage, const float* __restrict__ hist,
                                                                        unsigned* __restrict__ verdict, int slot) {
    const int tid = threadIdx.x, q = blockIdx.x;
    const cf* const hp = reinterpret_cast<const cf*>(hist) + ((size_t)q * kSlots + slot) * kB;
    unsigned* const row = stage + (size_t)(2 * q) * kB;                              // the pair's two rows, contiguous
    const cf c0 = hp[tid], c1 = hp[tid + kThreads];                                 // (channel a, channel b) of samples tid, tid + 256
    const unsigned v0 = rt_peek(row + tid), v1 = rt_peek(row + tid + kThreads), v2 = rt_peek(row + kB + tid), v3 = rt_peek(row + kB + tid + kThreads);
    const bool torn = v0 != __float_as_uint(c0.x) || v2 != __float_as_uint(c0.y) || v1 != __float_as_uint(c1.x) || v3 != __float_as_uint(c1.y);
    if (torn) __hip_atomic_fetch_or(verdict, kRtErrTorn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(row + tid, kRtSentinel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(row + tid + kThreads, kRtSentinel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(row + kB + tid, kRtSentinel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(row + kB + tid + kThreads, kRtSentinel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(kThreads, 2) void conv_round_trip_kernel(
    ConvRoundTrip rt, float* __restrict__ hist, const float4* __restrict__ pmA, const float4* __restrict__ pmB,
    const cf* __restrict__ tw, int T, int head) {
    __shared__ cf lds[2 * kLdsHalf];
    __shared__ int s_word;
    cf* const lds0 = lds;
    cf* const lds1 = lds + kLdsHalf;
    const int tid = threadIdx.x;
    const int q = blockIdx.x;                       // pairs in dispatch order: the copy lands them in that order too
    const int ta = 2 * q;
    int g = 0;
    while (q >= (int)rt.bound[g + 1]) ++g;          // (a scalar scan of at most kRtMaxGroups words)
    GAB_RT_STAMP_MIN(g, 0);
    cf* const hp = reinterpret_cast<cf*>(hist) + (size_t)q * kSlots * kB;
    using FA = fft::BlockFFT<kNA, 4, false>;
    using FAi = fft::BlockFFT<kNA, 4, true>;
    using FB = fft::BlockFFT<kNB, 16, false>;
    using FBi = fft::BlockFFT<kNB, 16, true>;

    // ---- far partition: blocks k-8 .. k-1, oldest first (slot `head` holds the oldest until the new block replaces it)
    cf zb[16];
    typename FB::Bases twb_base;
    FB::load_twiddles(twb_base, tw, tid);
#pragma unroll
    for (int r = 0; r < 16; ++r) zb[r] = hp[((head + (r >> 1)) & (kSlots - 1)) * kB + (r & 1) * kThreads + tid];
    const cf prev0 = zb[14], prev1 = zb[15];        // block k-1: the near window's first half
    {
        float4 cb[16];
        load_spectra<kNB, 16>(cb, pmB + (size_t)q * kBinsB, tid);
        __builtin_amdgcn_sched_barrier(0);
        typename FB::Twiddles twb;
        FB::expand_twiddles(twb_base, twb);
        FB::run(zb, lds1, lds0, twb, tid);
        cf zpb[16];
        partner_exchange<kNB, 16, true>(zb, zpb, lds1, tid);
        spectral_product<kNB, 16>(zb, zpb, cb, tid);
        FBi::template run<typename FB::Twiddles, 2>(zb, lds0, lds1, twb, tid);      // only [14], [15]
    }
    const cf far0 = zb[14], far1 = zb[15];

    // everything else the near partition needs is fetched before the wait
    float4 ca[4];
    load_spectra<kNA, 4>(ca, pmA + (size_t)q * kBinsA, tid);
    typename FA::Twiddles twa;
    {
        typename FA::Bases twa_base;
        FA::load_twiddles(twa_base, tw, tid);
        FA::expand_twiddles(twa_base, twa);
    }

    // ---- the pair's two rows: wait until they have landed
    const unsigned* const row = rt.stage + (size_t)ta * kB;                         // 2 x 512 words, contiguous
    if (tid == 0) {
        int tries = 0, bad = 0;
        while (rt_pending(rt_peek(row + 2 * kB - 1))) {                             // the region's last word
            // (a pinned word: looked at every 64th round only — 512 pollers reading it every round would be link traffic)
            if ((++tries & 63) == 0 && rt_peek(rt.landed) == rt.epoch) break;       // the whole upload is in: it IS the sentinel
            if (tries > kRtPollLimit) { bad = 1; break; }
            __builtin_amdgcn_s_sleep(30);
        }
        s_word = bad;
    }
    __syncthreads();
    bool gave_up = s_word != 0;
    unsigned w[4];
    {
        int tries = 0;
        for (;;) {
            w[0] = rt_peek(row + tid);
            w[1] = rt_peek(row + tid + kThreads);
            w[2] = rt_peek(row + kB + tid);
            w[3] = rt_peek(row + kB + tid + kThreads);
            if (gave_up || !(rt_pending(w[0]) || rt_pending(w[1]) || rt_pending(w[2]) || rt_pending(w[3]))) break;
            // A word that is STILL the sentinel counts as a sample only on this chain: the copy's completion signal (its
            // writes are visible at system scope before it: HSA) -> the host's hipEventQuery -> the host's release store
            // to `landed` -> THIS acquire load -> loads issued after it.  One more look behind the acquire is final.
            if ((++tries & 15) == 0 && __hip_atomic_load(rt.landed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == rt.epoch) {
                w[0] = rt_peek(row + tid);
                w[1] = rt_peek(row + tid + kThreads);
                w[2] = rt_peek(row + kB + tid);
                w[3] = rt_peek(row + kB + tid + kThreads);
                break;
            }
            if (tries > kRtPollLimit) { gave_up = true; break; }
            __builtin_amdgcn_s_sleep(10);
        }
    }
    GAB_RT_STAMP_MAX(g, 1);
    if (gave_up) __hip_atomic_store(rt.error, kRtErrWait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // (the words stay where they are: conv_round_trip_check_kernel, behind the upload's completion, compares them with what
    // was taken here and puts the sentinel back)

    // ---- near partition: [block k-1 | block k]
    cf za[4];
    za[0] = prev0;
    za[1] = prev1;
    za[2] = mk(__uint_as_float(w[0]), __uint_as_float(w[2]));
    za[3] = mk(__uint_as_float(w[1]), __uint_as_float(w[3]));
    hp[head * kB + tid] = za[2];                     // the new block replaces the oldest
    hp[head * kB + kThreads + tid] = za[3];
    __syncthreads();                                 // the far partition's last LDS readers are done
    FA::run(za, lds0, lds1, twa, tid);
    {
        cf zpa[4];
        partner_exchange<kNA, 4, true>(za, zpa, lds0, tid);
        spectral_product<kNA, 4>(za, zpa, ca, tid);
    }
    FAi::run(za, lds1, lds0, twa, tid);
    float ya0 = 0.f, yb0 = 0.f, ya1 = 0.f, yb1 = 0.f;           // the sums in conv_one_buffer's order: (0 + near) + far
    ya0 += za[2].x; yb0 += za[2].y; ya1 += za[3].x; yb1 += za[3].y;
    ya0 += far0.x; yb0 += far0.y; ya1 += far1.x; yb1 += far1.y;

    // ---- park (write-through, so that the draining workgroup, on whatever XCD, finds it in memory)
    {
        unsigned long long* const p0 = reinterpret_cast<unsigned long long*>(rt.park + (size_t)T * tid + ta);
        unsigned long long* const p1 = reinterpret_cast<unsigned long long*>(rt.park + (size_t)T * (tid + kThreads) + ta);
        const unsigned long long v0 = ((unsigned long long)__float_as_uint(yb0) << 32) | __float_as_uint(ya0);
        const unsigned long long v1 = ((unsigned long long)__float_as_uint(yb1) << 32) | __float_as_uint(ya1);
        __hip_atomic_store(p0, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p1, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave drains its own stores
    __syncthreads();
    GAB_RT_STAMP_MAX(g, 2);
    const int first = rt.bound[g];
    const int members = (int)rt.bound[g + 1] - first;
    unsigned* const gw = rt.counters + 32 * g;
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(&gw[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int role = 1;                                            // 0 gave up, 1 helper, 2 last arriver
        if (old + 1u == rt.epoch * (unsigned)members) {
            role = 2;
            __hip_atomic_store(&gw[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the claim counter is at zero before anyone is told
            __hip_atomic_store(&gw[2], rt.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            int tries = 0;
            while (__hip_atomic_load(&gw[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != rt.epoch) {
                if (++tries > (1 << 10)) { role = 0; break; }   // a fraction of a millisecond: the last arriver drains what is left
                __builtin_amdgcn_s_sleep(3);
            }
        }
        s_word = role;
    }
    __syncthreads();
    const int role = s_word;
    if (role == 0) return;
    const int row_f4 = members / 2;                              // float4 per row (members is even: T % 4 == 0)
    const auto srd = __builtin_amdgcn_make_buffer_rsrc(rt.park, 0, (int)((size_t)T * kB * 4), 0x00020000);
    const int col0 = 2 * first;
    unsigned drained = 0;
    for (;;) {
        __syncthreads();                                         // s_word's readers of the previous round are done
        if (tid == 0) s_word = (int)__hip_atomic_fetch_add(&gw[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int share = s_word;
        if (share >= members) break;
        const int i = share * kThreads + tid;
        const int sidx = i / row_f4, c = i - sidx * row_f4;
        const auto raw = __builtin_amdgcn_raw_buffer_load_b128(srd, 4u * (unsigned)(T * sidx + col0 + 4 * c), 0, 16);   // sc1
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v val = {__uint_as_float(raw[0]), __uint_as_float(raw[1]), __uint_as_float(raw[2]), __uint_as_float(raw[3])};
        float* const dst = rt.h_out + (size_t)T * sidx + col0 + 4 * c;
        // system-scope write-through: nothing of it stays behind in a cache
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(val) : "memory");
        ++drained;
        if (role != 2) break;                                    // helpers take one share; the last arriver takes what is left
    }
    // every wave waits for its own rows, then its shares count as drained; the workgroup whose count completes the launch
    // writes the hint word (the host goes and waits for the launch's end when it sees it: completion is the stream's)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    GAB_RT_STAMP_MAX(g, 3);
    if (tid == 0 && drained) {
        const unsigned old = __hip_atomic_fetch_add(&rt.counters[32 * rt.groups], drained, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + drained == rt.epoch * (unsigned)gridDim.x) {
            __hip_atomic_store(rt.done, rt.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            GAB_RT_STAMP_MAX(64, 0);
        }
    }
}

// The launch carries n_buffers consecutive buffers (in/out are [n][T*B]); a workgroup walks them
// in order for its pair.  Pairs are independent, so nothing is synchronised between workgroups; a
// thread re-reads from the ring only what it wrote itself; the LDS hand-over rule of the stages
// holds across iterations (the last stage's final reads are from the buffer the next iteration
// writes second).
__global__ __launch_bounds__(kThreads, 2) void conv_batch_kernel(
    const float* __restrict__ in, float* __restrict__ out, float* __restrict__ hist,
    const float4* __restrict__ pmA, const float4* __restrict__ pmB,
    const cf* __restrict__ tw, int T, int head, int n_buffers) {
    __shared__ cf lds[2 * kLdsHalf];
    const size_t step = (size_t)T * kB;
    for (int nb = 0; nb < n_buffers; ++nb)
        conv_one_buffer<true, true, true>(in + nb * step, out + nb * step, hist, pmA, pmB, tw, T,
                                             (head + nb) & (kSlots - 1), lds);
}

// ---- split cut, n buffers per launch: a duo of channel pairs in ONE resident workgroup -------------------
// gab_conv_process_batch on a split plan (bench.py's `value`).  A workgroup owns a duo for the whole launch,
// so the parked far share never leaves it and nothing needs a kernel boundary between buffers.  Same
// arithmetic, same order as conv_split_kernel — bit-identical to n launches of it — with fewer bytes, fewer
// transforms, and every role's work cut to the barrier intervals of the far role's transform:
//  * A2's window at buffer k, [k-2 | k-1], IS A's window at buffer k-1, so its spectrum is the one A's
//    wave left in its LDS image a buffer earlier: the A2 product is formed from that image before the
//    next forward transform overwrites it (the first buffer of a launch gets it from a prologue transform
//    of the ring's blocks).  Same inputs, same operations: bit-identical.  One forward transform per pair
//    and buffer instead of two;
//  * the previous block stays in registers, so the near role loads only the NEW block of every buffer
//    (a per-buffer launch reads every input block as k, k-1 and k-2);
//  * the carry ring of the duo lives in LDS for the whole launch (32 KB): the far share never goes to
//    memory between buffers; it is loaded at entry and written back at exit;
//  * the near role is a pipeline of two waves per pair: the FORWARD wave (A2 product, forward transform,
//    A product, sum) hands the output spectrum of buffer k to the INVERSE wave through LDS, which turns it
//    into samples one period later, while the forward wave is already on buffer k+1.  A launch of n
//    buffers therefore runs n + 1 periods (the far and forward waves idle in the last one).
// (Round 2's form — near role on four waves with both near transforms per pair, carry ring in memory,
// 6.4 us per buffer — is in the history of this file; this one runs 5.3.)
// Round 5's workgroup (k_conv_accel_diag.hpp, diagnostic builds) = 512 threads: waves 0-1 forward (pair 0, pair 1), waves 2-3
// inverse, waves 4-7 far — one near and one far wave per SIMD; the product's is the twelve-wave one further down, which says
// what it changes (two far groups, the carry through memory, spectra in LDS).  Every wave executes kBatchBarriers s_barriers per period; a
// wave-held transform arrives at two of them from inside (WaveFFT1024's hook), so that each role does
// about one transform pass per barrier interval.
constexpr int kBatchBarriers = 6;

#ifdef GAB_ABLATE
// diagnostic bit 64: in period 32 of a launch every wave's lane 0 stamps s_memrealtime (100 MHz) —
// near waves when they ARRIVE at each of the six barriers (slots 0-5) and after the last (6), far waves
// after each barrier RELEASES them (slots 0-5) and at the period's start (6); slot = [block][wave][8].
#define GAB_BSTAMP(i)                                                                                     \
    do {                                                                                                  \
        if (GAB_SDBG(64) && nb == (ENGINE ? 4000 : 32) && lane == 0)     /* the engine: once the clocks have settled */   \
            g_split_stamps[((size_t)blockIdx.x * 8 + w) * 8 + (i)] = __builtin_amdgcn_s_memrealtime();    \
    } while (0)
#else
#define GAB_BSTAMP(i) do {} while (0)
#endif
struct ArriveAtBarrier { __device__ __forceinline__ void operator()(int) const { __syncthreads(); } };

// ENGINE (gab_conv_engine_*, conv_split_engine12_resident below): the batch launch's period code kept on the device; it takes
// buffer nb from slot nb % ring of an input ring when the doorbell word says it has been published, instead of n_buffers known at
// launch.
//   doorbell   [pinned host memory] bits 0-29 buffers published so far, bit 31 STOP (no more will come), bit 30 FLUSH (finish what
//              is published without waiting for more: round 5, ONE buffer in flight).  Only workgroup 0 reads it over the link —
//              256 workgroups asking the host every period cost 20-40 us per period, measured — and passes it on through a word in
//              device memory (`relay`) that one lane of every workgroup (the first inverse wave's) asks for at the start of a
//              period; the answer is needed at the period's end, so a producer that keeps a few buffers ahead never makes the
//              engine wait; otherwise every wave of the workgroup parks at a barrier while the first inverse wave polls (bounded:
//              then the engine stops with an error).
//   period nb  runs when buffer nb + 1 is there too (its operands are requested one period ahead), or — STOP or FLUSH — when
//              buffer nb is the last one published: such a period requests nothing and ends its BURST; a drain period delivers
//              the buffer, the workgroup idles in the gate and the next burst starts cold.  The output of buffer nb leaves one
//              period after its own, as in a batch launch.
//   progress   every inverse wave stores (write-through) how many buffers it has finished — pipelined: one period late, when the
//              rows' own stores have long drained; at a burst's end: behind a wait for them.  The first inverse wave of workgroup 1
//              takes the minimum of all of them (each period, and while it idles) and writes it into `completed` (pinned host word).
// The history ring is written every period (the last eight buffers are not known in advance) and the far role takes
// blocks k-7 .. k-2 from it, as round 3's batch kernel did; everything else is the batch launch's period code: same bits.
struct ConvEngine {
    const unsigned* doorbell;
    unsigned* relay;              // device: the doorbell as workgroup 0 last saw it (zero at launch)
    unsigned* progress;           // device: [2 x workgroups] buffers finished per inverse wave (zero at launch)
    unsigned* completed;          // pinned host
    unsigned* error;              // pinned host
    int ring;                     // buffers in the input and output rings (>= 3)
    int poll_every_period;        // 1; diagnostic builds may turn it off (the word is then read only when the engine stalls)
    unsigned* started;            // device: workgroups that have begun (zero at launch)
    unsigned* resident;           // pinned host: [0] = 1 once the first workgroup runs, [1] = workgroups once the last one does
    unsigned long long idle_ticks;   // of the 100 MHz wall clock: a stalled workgroup gives up after that long without the doorbell moving
};
constexpr double kEngineIdleSeconds = 4.0;     // the default of gab_conv_engine_set_idle_limit

typedef unsigned u4 __attribute__((ext_vector_type(4)));
// experiments only (GAB_EXTRA_FLAGS=-DGAB_ENGV=bits, a build of its own): 1 no progress words / aggregator, 2 plain output
// stores, 4 plain input loads, 8 history from the input ring as a batch launch takes it (needs a ring of >= 9 slots),
// 32 the doorbell is read only when the engine stalls, 64 rings in fine-grained device memory, 256 non-temporal input loads (64 + 256: the engine as it was before the rings became ordinary memory).  Compile-time: a run-time switch at every load perturbs what it measures.
#ifndef GAB_ENGV
#define GAB_ENGV 0
#endif
#define GAB_EABL(bit) ((GAB_ENGV & (bit)) != 0)
// (experiments on the twelve-wave engine's output stores: GAB_ENGSTORE 1 "sc0 sc1", 2 "nt sc1", 3 "nt", 4 plain;
// GAB_ENGWB / GAB_ENGWB8: an L2 write-back by every inverse wave before its progress word / by one wave per XCD and period;
// GAB_ENGPUSH: whole lines written through a second time — profiles/r06_engine12.md)
#ifndef GAB_ENGSTORE
#define GAB_ENGSTORE 0
#endif
#if GAB_ENGSTORE == 1
#define GAB_ENGSTORE_BITS "sc0 sc1"
#elif GAB_ENGSTORE == 2
#define GAB_ENGSTORE_BITS "nt sc1"
#elif GAB_ENGSTORE == 3
#define GAB_ENGSTORE_BITS "nt"
#elif GAB_ENGSTORE == 4
#define GAB_ENGSTORE_BITS ""
#else
#define GAB_ENGSTORE_BITS "sc1"
#endif

// ---- n buffers per launch on TWELVE waves, three per SIMD (round 6) ----------------------------------------------------------
// Round 5's batch launch (k_conv_accel_diag.hpp) holds two waves per SIMD at 244 registers: a near wave and a far wave, and what the period costs
// beyond the far role's own chain is what two waves cannot hide of each other's latency.  Here the far role is
// TWO groups of four waves, group g owning pair g of the duo for the whole launch; a pair's turn comes every other buffer,
// so a group takes TWO periods — twelve barrier intervals — per transform and each SIMD holds a near wave and a wave of
// either group.  Same arithmetic on the same values in the same order as conv_split_batch_kernel (same transform passes,
// same twiddle powers, same products, same sums): bit-identical to it and to n launches of conv_split_kernel.
// What makes 168 registers per wave and 160 KB of LDS enough:
//   * a group transforms in ONE padded LDS image (write, barrier, read, barrier: it has twelve intervals for six exchanges);
//   * the carry ring goes through memory again (8 KB per pair and buffer, written and read within two periods by the same
//     compute unit: L2 hits): its 32 KB of LDS hold the duo's NEAR spectra instead (taps [0,512) and [512,1024) of both
//     pairs, 32.8 KB), loaded once per launch — the forward waves no longer carry 64 registers of spectra from one interval
//     to the next, and no longer ask the L2 for 33 KB per duo and period;
//   * the far waves keep the LAST pass's twiddle powers in registers (30) and read the middle pass's (16 distinct sets)
//     from a 1.9 KB table in LDS, formed at entry by the same powers_of;
//   * the A2 share waits in the hand-over image (written behind barrier 1, summed into behind barrier 5) instead of in
//     32 registers; spectral products run in halves of eight bins.
// The far role's schedule per transform (window nb, periods nb and nb + 1; the other group is one period out of step, so a
// heavy interval of one meets a light one of the other):
//   period nb:      pass 0 + write | read, pass 1 | write | read, pass 2 | partner write | partner read, product, inverse pass 0
//   period nb + 1:  (next window's requests) | write | read, pass 1 | write | read, last pass (4 of 16), carry out | -
constexpr int kB12Threads = 768;
constexpr int kB12SpecCf = 4 * kBinsA * 2;                         // [A pair 0, A pair 1, A2 pair 0, A2 pair 1][513] float4, in cf entries
constexpr int kB12Tw1Cf = 16 * 15;
constexpr int kB12Lds = 6 * kWaveImg + 2 * kLdsHalf + kB12SpecCf + kB12Tw1Cf;     // cf entries (156 608 bytes)

// bins [R0, R1) of load_spectra / spectral_product (same index rule, same operations per bin)
template <int N, int R, int R0, int R1, class PM>
__device__ __forceinline__ void load_spectra_part(float4 (&c)[R1 - R0], PM pm, int tid) {
    constexpr int NT = N / R;
#pragma unroll
    for (int r = R0; r < R1; ++r) {
        const int k = tid + r * NT;
        if (r < R / 2) c[r - R0] = pm[k];
        else if (r > R / 2) c[r - R0] = pm[N - k];
        else c[r - R0] = pm[tid == 0 ? k : N - k];
    }
}
template <int N, int R, int R0, int R1>
__device__ __forceinline__ void spectral_product_part(cf (&z)[R], const cf (&zraw)[R1 - R0], const float4 (&c)[R1 - R0], int tid) {
#pragma unroll
    for (int r = R0; r < R1; ++r) {
        cf P = mk(c[r - R0].x, c[r - R0].y), M = mk(c[r - R0].z, c[r - R0].w);
        if (r > R / 2 || (r == R / 2 && tid != 0)) z[r] = fft::cfma_cjcj(zraw[r - R0], M, fft::cmulc(z[r], P));
        else z[r] = fft::cfma_cj(zraw[r - R0], M, fft::cmul(z[r], P));
    }
}

#ifdef GAB_ABLATE
// diagnostic bit 64: in periods 32 and 33 of a launch every wave's lane 0 stamps s_memrealtime (100 MHz) when it ARRIVES at each of
// the six barriers: slot = [block][wave (12)][period - 32][barrier]; tools/stamp_batch12.py
#define GAB_B12BAR(period, i)                                                                                         \
    do {                                                                                                              \
        if (GAB_SDBG(64) && ((period) == 32 || (period) == 33) && lane == 0)                                          \
            g_split_stamps[(((size_t)blockIdx.x * 12 + w) * 2 + ((period) - 32)) * 6 + (i)] = __builtin_amdgcn_s_memrealtime(); \
        __syncthreads();                                                                                              \
    } while (0)
#else
#define GAB_B12BAR(period, i) __syncthreads()
#endif
__device__ __forceinline__ void conv_split_batch12_resident(
    const float* __restrict__ in, float* __restrict__ out, float* __restrict__ hist,
    const float4* __restrict__ pmA, const ConvSplit& sp, const cf* __restrict__ tw, int T, int head0, int n_buffers,
    cf* __restrict__ lds) {
    cf* const far_img = lds + 6 * kWaveImg;                           // [group][kLdsHalf]
    float4* const spec = reinterpret_cast<float4*>(far_img + 2 * kLdsHalf);   // [A p0 | A p1 | A2 p0 | A2 p1][513]
    cf* const tw1 = reinterpret_cast<cf*>(spec + 4 * kBinsA);          // [16][15]: W256^(r k), k = thread & 15 (the middle pass)
    const int tid = threadIdx.x;
    const int d = xcd_contiguous(blockIdx.x, gridDim.x);
    const size_t step = (size_t)T * kB;
    {
        const float4* const gA = pmA + (size_t)(2 * d) * kBinsA;       // the duo's two pairs are contiguous
        const float4* const gA2 = sp.pmA2 + (size_t)(2 * d) * kBinsA;
        for (int i = tid; i < 2 * kBinsA; i += kB12Threads) { spec[i] = gA[i]; spec[2 * kBinsA + i] = gA2[i]; }
        if (tid < 16) {
            cf pw[15];
            fft::powers_of<16>(tw[tid * (fft::kTwiddleN / 256)], pw);
#pragma unroll
            for (int r = 0; r < 15; ++r) tw1[tid * 15 + r] = pw[r];
        }
    }
    __syncthreads();
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned rb = (unsigned)lane + ((unsigned)lane >> 4);      // Pad(lane + 64 r) = rb + 68 r
    auto in_slot = [&](int slot) -> const float* { return in + (size_t)slot * step; };
    auto idle_period = [&]() { for (int i = 0; i < kBatchBarriers; ++i) __syncthreads(); };

    if (w >= 4) {
        // ---- far waves: group g turns pair g of the duo, one transform per two periods
        const int g = __builtin_amdgcn_readfirstlane((w - 4) >> 2);
        const int ft = (tid - kThreads) & (kThreads - 1);
#ifdef GAB_ABLATE
        if (GAB_SDBG(2048)) __builtin_amdgcn_s_setprio(2);            // diagnostic builds: the far waves ahead of the near ones
        if (GAB_SDBG(4096) && g == 1) __builtin_amdgcn_s_setprio(1);  // ... or only the younger group (the SIMD favours the older)
#endif
        cf* const img = far_img + g * kLdsHalf;
        const int q = 2 * d + g;
        const cf* const hp = reinterpret_cast<const cf*>(hist) + (size_t)q * kSlots * kB;
        cf* const cg = sp.carry + (size_t)q * kCarrySlots * kB;
        const size_t ca = (size_t)(2 * q) * kB, cb_ = ca + kB;
        const float4* const pf = sp.pmF + (size_t)q * kBinsB;
        using B16 = fft::Butterfly<16, false>;
        using B16i = fft::Butterfly<16, true>;
        cf tw2[15];                                                   // the last pass's powers: W4096^(r t)
        fft::powers_of<16>(tw[ft], tw2);
        // LDS positions are formed where they are used, from an opaque copy of the thread index: as loop invariants they
        // would be kept (and spilled) across both halves of a transform
        auto opaque_t = [&]() -> unsigned { unsigned v = (unsigned)ft; asm volatile("" : "+v"(v)); return v; };
        auto w0_of = [](unsigned t) -> unsigned { return t * 17u; };                       // pass-0 writes: Pad(16 t + r) = 17 t + r
        auto rd_of = [](unsigned t) -> unsigned { return t + (t >> 4); };                  // linear reads: Pad(t + 256 r) = rd + 272 r
        auto w1_of = [](unsigned t) -> unsigned { const unsigned b1 = (t >> 4) * 256u + (t & 15u); return b1 + (b1 >> 4); };   // pass-1 writes: + 17 r
        auto tw1row_of = [&](unsigned t) -> const cf* { return tw1 + (t & 15u) * 15u; };
        // window of buffer nb: blocks k-7 .. k — from the input buffers where they lie inside the launch, else the ring
        // PART 0: the seven older blocks' first halves + the newest block (16 requests); PART 1: the rest (16 requests) — the
        // requests of one window go out in two intervals: thirty-two in one made that interval the period's longest
        auto load_window = [&](auto part_tag, int nb, cf (&z)[16]) {
            constexpr int PART = decltype(part_tag)::value;
            const int head = (head0 + nb) & (kSlots - 1);
            int fo = ft;
            asm volatile("" : "+v"(fo));                              // (addresses formed here, not hoisted and spilled)
            if constexpr (PART == 0) {
                const float* const cur = in_slot(nb);
                z[14] = mk(cur[ca + fo], cur[cb_ + fo]);
                z[15] = mk(cur[ca + fo + kThreads], cur[cb_ + fo + kThreads]);
            }
            constexpr int BL0 = PART == 0 ? 4 : 0, BL1 = PART == 0 ? 7 : 4;       // blocks k-7+bl
            if (nb >= kSlots - 1) {
#pragma unroll
                for (int bl = BL0; bl < BL1; ++bl) {
                    const float* const src = in_slot(nb - (7 - bl));
                    z[2 * bl] = mk(src[ca + fo], src[cb_ + fo]);
                    z[2 * bl + 1] = mk(src[ca + fo + kThreads], src[cb_ + fo + kThreads]);
                }
            } else {
#pragma unroll
                for (int bl = BL0; bl < BL1; ++bl) {                  // block k-7+bl = buffer nb-7+bl (uniform branch)
                    if (nb - 7 + bl >= 0) {
                        const float* const src = in_slot(nb - (7 - bl));
                        z[2 * bl] = mk(src[ca + fo], src[cb_ + fo]);
                        z[2 * bl + 1] = mk(src[ca + fo + kThreads], src[cb_ + fo + kThreads]);
                    } else {
                        const int s = ((head + 1 + bl) & (kSlots - 1)) * kB;
                        z[2 * bl] = hp[s + fo];
                        z[2 * bl + 1] = hp[s + kThreads + fo];
                    }
                }
            }
        };
        cf z[16], zn[16];
        // The requests ride on the lighter steps (every instruction of a step costs the wave about a dozen clocks beside two
        // others on its SIMD: thirty-two requests in one interval made it the period's longest), the wait for the carry's stores
        // stands in the idle interval.  (Measured and not kept, profiles/r06_batch12_stamps.txt: steps cut as read + twiddle |
        // butterfly + write, so that no step is a bare write — 5.08 against 5.01 us per buffer.)
        // first half of a transform (period nb):   pass 0, write | read, twiddle, pass 1 | write (+ spectra 0-7) |
        //                                          read, twiddle (+ spectra 8-15), pass 2 | partner write | partner read, product, inverse pass 0
        auto first_half = [&](int nb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = zn[r];
            B16::run(z);
            {
                const unsigned w0 = w0_of(opaque_t());
#pragma unroll
                for (int r = 0; r < 16; ++r) img[w0 + r] = z[B16::out_slot(r)];
            }
            GAB_B12BAR(nb, 0);                                          // 1
            {
                const unsigned t = opaque_t(), rd = rd_of(t);
                const cf* const tw1row = tw1row_of(t);
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = img[rd + 272 * r];
#pragma unroll
                for (int r = 1; r < 16; ++r) z[r] = fft::cmul(z[r], tw1row[r - 1]);
            }
            B16::run(z);
            GAB_B12BAR(nb, 1);                                          // 2: every wave has read
            {
                const unsigned w1 = w1_of(opaque_t());
#pragma unroll
                for (int r = 0; r < 16; ++r) img[w1 + 17 * r] = z[B16::out_slot(r)];
            }
            __builtin_amdgcn_sched_barrier(0);
            float4 clo[8], chi[8];                                    // the far spectra: bins r < 8 asked for here, r >= 8 an interval later
            {
                int fo = ft;
                asm volatile("" : "+v"(fo));
                load_spectra_part<kNB, 16, 0, 8>(clo, pf, fo);
            }
            __builtin_amdgcn_sched_barrier(0);
            GAB_B12BAR(nb, 2);                                          // 3
            {
                const unsigned rd = rd_of(opaque_t());
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = img[rd + 272 * r];
            }
#pragma unroll
            for (int r = 1; r < 16; ++r) z[r] = fft::cmul(z[r], tw2[r - 1]);
            __builtin_amdgcn_sched_barrier(0);
            {
                int fo = ft;
                asm volatile("" : "+v"(fo));
                load_spectra_part<kNB, 16, 8, 16>(chi, pf, fo);
            }
            __builtin_amdgcn_sched_barrier(0);
            B16::run(z);
            GAB_B12BAR(nb, 3);                                          // 4
            {
                const unsigned t = opaque_t();
#pragma unroll
                for (int r = 0; r < 16; ++r) img[t + 256u * r] = z[B16::out_slot(r)];   // Z[t + 256 r]: the partner exchange, raw
            }
            GAB_B12BAR(nb, 4);                                          // 5
            {
                // the thread's own bins back in order, and Z[(N - k) mod N], k = t + 256 r: one base and constant offsets for
                // r >= 1 (N - k > 0 there); bin k = t alone wraps
                cf o[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = z[B16::out_slot(r)];
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = o[r];            // z[r] = Z[t + 256 r]
                const unsigned t = opaque_t();
                const cf* const pb = img + (kNB - 256 * 15) - t;      // pb[256 (15 - r)] = img[N - t - 256 r]
                cf zp[8];
                zp[0] = img[(kNB - t) & (kNB - 1)];
#pragma unroll
                for (int r = 1; r < 8; ++r) zp[r] = pb[256 * (15 - r)];
                spectral_product_part<kNB, 16, 0, 8>(z, zp, clo, ft);
#pragma unroll
                for (int r = 0; r < 8; ++r) zp[r] = pb[256 * (7 - r)];
                spectral_product_part<kNB, 16, 8, 16>(z, zp, chi, ft);
            }
            B16i::run(z);
            GAB_B12BAR(nb, 5);                                          // 6
        };
        // second half (period nb + 1):   (half of the next window's requests) | write | read, twiddle, pass 1 |
        //                                write (+ the other half) | read, twiddle, last pass (4 of 16), carry out | (the carry's stores leave)
        auto second_half = [&](int period, int nb_done, int nb_next) {
            if (nb_next >= 0) {
                load_window(std::integral_constant<int, 0>{}, nb_next, zn);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) zn[r] = mk(0.0f, 0.0f);  // (no value survives from the last window: registers)
            }
            GAB_B12BAR(period, 0);                                          // 1
            if (nb_done >= 0) {
                const unsigned w0 = w0_of(opaque_t());
#pragma unroll
                for (int r = 0; r < 16; ++r) img[w0 + r] = z[B16i::out_slot(r)];
            }
            GAB_B12BAR(period, 1);                                          // 2
            if (nb_done >= 0) {
                const unsigned t = opaque_t(), rd = rd_of(t);
                const cf* const tw1row = tw1row_of(t);
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = img[rd + 272 * r];
#pragma unroll
                for (int r = 1; r < 16; ++r) z[r] = fft::cmulc(z[r], tw1row[r - 1]);
                B16i::run(z);
            }
            GAB_B12BAR(period, 2);                                          // 3
            if (nb_done >= 0) {
                const unsigned w1 = w1_of(opaque_t());
#pragma unroll
                for (int r = 0; r < 16; ++r) img[w1 + 17 * r] = z[B16i::out_slot(r)];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (nb_next >= 0) load_window(std::integral_constant<int, 1>{}, nb_next, zn);      // the window's other half, beside the light write
            __builtin_amdgcn_sched_barrier(0);
            GAB_B12BAR(period, 3);                                          // 4
            if (nb_done >= 0) {
                const unsigned rd = rd_of(opaque_t());
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = img[rd + 272 * r];
#pragma unroll
                for (int r = 1; r < 16; ++r) z[r] = fft::cmulc(z[r], tw2[r - 1]);
                cf x12, x13, x14, x15;
                B16i::run_last4(z, x12, x13, x14, x15);
                const int head = (head0 + nb_done) & (kSlots - 1);
                cf* const c1 = cg + ((head + 1) & (kCarrySlots - 1)) * kB;    // block k+1
                cf* const c2 = cg + ((head + 2) & (kCarrySlots - 1)) * kB;    // block k+2
                c1[ft] = x12;
                c1[ft + kThreads] = x13;
                c2[ft] = x14;
                c2[ft + kThreads] = x15;
            }
            GAB_B12BAR(period, 4);                                          // 5
            // the inverse wave of this pair asks for block k+1's share behind the period's closing barrier: the stores must
            // have left this wave by then (waited for HERE, in the group's idle interval, not on its last pass)
            if (nb_done >= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            GAB_B12BAR(period, 5);                                          // 6
        };
        const int first = (g - head0) & 1;                             // this group's first window
        int nb = 0;
        if (first == 1) {
            second_half(0, -1, 1 < n_buffers ? 1 : -1);                   // period 0: nothing to finish, window 1 asked for
            nb = 1;
        } else if (n_buffers > 0) {
            load_window(std::integral_constant<int, 0>{}, 0, zn);
            load_window(std::integral_constant<int, 1>{}, 0, zn);
        }
        for (;;) {                                                     // nb: a period in which a transform of this group starts
            if (nb > n_buffers) break;
            if (nb < n_buffers) first_half(nb); else idle_period();
            ++nb;
            if (nb > n_buffers) break;
            second_half(nb, nb - 1 < n_buffers ? nb - 1 : -1, nb + 1 < n_buffers ? nb + 1 : -1);
            ++nb;
        }
    } else if (w < 2) {
        // ---- forward waves: wave w holds pair w of the duo
        const int q = 2 * d + w;
        cf* const hp = reinterpret_cast<cf*>(hist) + (size_t)q * kSlots * kB;
        cf* const img = lds + w * kWaveImg;                           // the transform's exchanges, then its spectrum
        cf* const hand = lds + (2 + w) * kWaveImg;                    // A2 share, then the output spectrum for the inverse wave
        const float4* const sa = spec + w * kBinsA;                   // taps [0,512)
        const float4* const sa2 = spec + (2 + w) * kBinsA;            // taps [512,1024)
        using WF = fft::WaveFFT1024<false>;
        WF::Lean t;
        WF::load_twiddles(t, tw, lane);
        const size_t xoff = (size_t)(2 * q) * kB;                     // channel a of a buffer; channel b is kB further
        cf z[16], prev[8], nxt[8];
        // W = Z x spectra (from LDS) in halves of eight bins: partner values and spectra of a half live in registers at a time
        auto product_from_image = [&](cf (&v)[16], const float4* sp_lds) {
            {
                cf zp[8];
                float4 ch[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) zp[r] = img[PadA16::at((kNA - (lane + 64 * r)) & (kNA - 1))];
                load_spectra_part<kNA, 16, 0, 8>(ch, sp_lds, lane);
                spectral_product_part<kNA, 16, 0, 8>(v, zp, ch, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                cf zp[8];
                float4 ch[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) zp[r] = img[PadA16::at((kNA - (lane + 64 * (8 + r))) & (kNA - 1))];
                load_spectra_part<kNA, 16, 8, 16>(ch, sp_lds, lane);
                spectral_product_part<kNA, 16, 8, 16>(v, zp, ch, lane);
            }
        };
        {   // prologue: the spectrum of the ring's blocks [k-2 | k-1] into the image
            const int s1 = ((head0 + kSlots - 1) & (kSlots - 1)) * kB, s2 = ((head0 + kSlots - 2) & (kSlots - 1)) * kB;
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = hp[s2 + lane + 64 * j];
#pragma unroll
            for (int j = 0; j < 8; ++j) prev[j] = hp[s1 + lane + 64 * j];
            if (n_buffers > 0) {
                const float* const x0 = in_slot(0) + xoff;
#pragma unroll
                for (int j = 0; j < 8; ++j) nxt[j] = mk(x0[lane + 64 * j], x0[kB + lane + 64 * j]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) z[8 + j] = prev[j];
            WF::run(z, img, t, lane, WF::NoHook());
#pragma unroll
            for (int r = 0; r < 16; ++r) img[rb + 68 * r] = z[r];
            __builtin_amdgcn_wave_barrier();
        }
        for (int nb = 0; nb < n_buffers; ++nb) {
            // A2 share | hand-over + window + pass 0 | pass 1 | pass 2 | spectrum + A product | sum into the hand-over
            const int head = (head0 + nb) & (kSlots - 1);
            {
                cf share[16];                                         // taps [512,1024): last period's spectrum x its spectra
#pragma unroll
                for (int r = 0; r < 16; ++r) share[r] = img[rb + 68 * r];
                product_from_image(share, sa2);
                GAB_B12BAR(nb, 0);                                   // barrier 1: the inverse wave has read the hand-over image
#pragma unroll
                for (int r = 0; r < 16; ++r) hand[rb + 68 * r] = share[r];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 8; ++j) { z[j] = prev[j]; z[8 + j] = nxt[j]; }
            if (nb + kSlots >= n_buffers) {                           // the ring only has to hold the launch's LAST eight blocks
#pragma unroll
                for (int j = 0; j < 8; ++j) hp[head * kB + lane + 64 * j] = nxt[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) prev[j] = nxt[j];
            if (nb + 1 < n_buffers) {                                 // the next buffer's block: needed a period from now
                int lo = lane;
                asm volatile("" : "+v"(lo));
                const float* const xa = in_slot(nb + 1) + xoff + lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) nxt[j] = mk(xa[64 * j], xa[kB + 64 * j]);
            }
            WF::run(z, img, t, lane, [&](int i) { GAB_B12BAR(nb, 1 + i); });   // barriers 2, 3 from inside
            GAB_B12BAR(nb, 3);                                       // barrier 4
#pragma unroll
            for (int r = 0; r < 16; ++r) img[rb + 68 * r] = z[r];     // the spectrum stays here for the next period
            __builtin_amdgcn_wave_barrier();
            product_from_image(z, sa);
            GAB_B12BAR(nb, 4);                                       // barrier 5
#pragma unroll
            for (int r = 0; r < 16; ++r) hand[rb + 68 * r] = fft::cadd(z[r], hand[rb + 68 * r]);   // A product + A2 share
            GAB_B12BAR(nb, 5);                                       // barrier 6 closes the period
        }
        idle_period();                                                // the pipeline's last period
    } else {
        // ---- inverse waves: wave 2 + p turns the output spectrum of pair p into samples, one period later
        const int pr = w - 2;
        const cf* const hand = lds + (2 + pr) * kWaveImg;
        cf* const img = lds + (4 + pr) * kWaveImg;                    // the transform's exchanges, then the output swap
        const cf* const other = lds + (4 + (1 - pr)) * kWaveImg;
        const cf* const cg = sp.carry + (size_t)(2 * d + pr) * kCarrySlots * kB;
        using WFi = fft::WaveFFT1024<true>;
        WFi::Lean t;
        WFi::load_twiddles(t, tw, lane);
        idle_period();                                                // first period: nothing to turn yet
        for (int nb = 1; nb <= n_buffers; ++nb) {
            // hand-over read + far share asked for | pass 0 | pass 1 | pass 2 + far share | swap | stores
            const int b = nb - 1;                                     // the buffer whose spectrum was handed over last period
            const int head = (head0 + b) & (kSlots - 1);
            float* const outb = out + (size_t)b * step;
            cf z[16], y[8], park[8];
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = hand[rb + 68 * r];
            {
                int lo = lane;
                asm volatile("" : "+v"(lo));
                // written by this workgroup's far waves one to three periods ago, behind barriers; agent-scope loads (answered
                // by the L2, never by a line this compute unit's L1 kept from the slot's previous use four buffers ago)
                const unsigned long long* const cy = reinterpret_cast<const unsigned long long*>(cg + (head & (kCarrySlots - 1)) * kB + lo);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned long long v = __hip_atomic_load(cy + 64 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    park[j] = mk(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
                }
            }
            GAB_B12BAR(nb, 0);                                       // barrier 1
            WFi::run(z, img, t, lane, [&](int i) { GAB_B12BAR(nb, 1 + i); });  // barriers 2, 3 from inside
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = fft::cadd(z[8 + j], park[j]);
            GAB_B12BAR(nb, 3);                                       // barrier 4
            if (pr == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) img[lane + 64 * j] = y[4 + j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) img[lane + 64 * j] = y[j];
            }
            GAB_B12BAR(nb, 4);                                       // barrier 5: the swapped halves are in LDS
            {
                float* const o0 = outb + 4 * (size_t)d;
                if (pr == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const cf theirs = other[lane + 64 * j];
                        *reinterpret_cast<float4*>(o0 + (size_t)T * (lane + 64 * j)) = make_float4(y[j].x, y[j].y, theirs.x, theirs.y);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const cf theirs = other[lane + 64 * j];
                        *reinterpret_cast<float4*>(o0 + (size_t)T * (lane + 64 * (4 + j))) = make_float4(theirs.x, theirs.y, y[4 + j].x, y[4 + j].y);
                    }
                }
            }
            GAB_B12BAR(nb, 5);                                       // barrier 6 closes the period
        }
    }
}

#ifdef GAB_ABLATE      // diagnostic builds only: round 5's eight-wave launches (GAB_BATCH_WAVES / GAB_ENGINE_WAVES = 8) and the
                       // six-wave forms (GAB_BATCH_WAVES = 6 / 64 / 26) that were measured against the product's and not kept
#include "k_conv_accel_diag.hpp"
#endif

__global__ __launch_bounds__(kB12Threads) void conv_split_batch12_kernel(
    const float* __restrict__ in, float* __restrict__ out, float* __restrict__ hist,
    const float4* __restrict__ pmA, ConvSplit sp, const cf* __restrict__ tw, int T, int head0, int n_buffers) {
    __shared__ __attribute__((aligned(16))) cf lds[kB12Lds];         // (the spectra in it are read as 16-byte pieces)
    conv_split_batch12_resident(in, out, hist, pmA, sp, tw, T, head0, n_buffers, lds);
}

// ---- the doorbell-fed engine on the twelve-wave period code (round 6) ----------------------------------------------------------
// conv_split_batch12_resident's roles behind conv_split_engine_resident's doorbell: the same gate (every wave asks it once per
// period, in step), bursts that start cold and end with a drain period, ring slots, system-scope input loads, write-through
// outputs, per-wave progress words.  A far group's transform spans two periods, so a transform begun in a burst's LAST period is
// finished in the burst's drain period; the window of a group's next turn is asked for a period ahead where that buffer is
// published (block k-1 then comes from the input ring), else at the turn itself (block k-1 from the history ring).
#ifdef GAB_ABLATE
// diagnostic bit 64: every wave's lane 0 stamps its ARRIVAL at each barrier of periods 4000 and 4001 (tools/stamp_engine12.py)
#define GAB_E12BAR(i)                                                                                                 \
    do {                                                                                                              \
        if (GAB_SDBG(64) && (nb == 4000 || nb == 4001) && lane == 0)                                                  \
            g_split_stamps[(((size_t)blockIdx.x * 12 + w) * 2 + (nb - 4000)) * 6 + (i)] = __builtin_amdgcn_s_memrealtime(); \
        __syncthreads();                                                                                              \
    } while (0)
#else
#define GAB_E12BAR(i) __syncthreads()
#endif
__device__ __forceinline__ void conv_split_engine12_resident(
    const float* __restrict__ in, float* __restrict__ out, float* __restrict__ hist,
    const float4* __restrict__ pmA, const ConvSplit& sp, const cf* __restrict__ tw, int T, int head0,
    const ConvEngine& eng, cf* __restrict__ lds, unsigned* __restrict__ s_door) {
    constexpr bool ENGINE = true;
    constexpr int n_buffers = 0;
    (void)ENGINE; (void)n_buffers;
    cf* const far_img = lds + 6 * kWaveImg;                           // [group][kLdsHalf]
    float4* const spec = reinterpret_cast<float4*>(far_img + 2 * kLdsHalf);   // [A p0 | A p1 | A2 p0 | A2 p1][513]
    cf* const tw1 = reinterpret_cast<cf*>(spec + 4 * kBinsA);          // [16][15]: the middle pass's twiddle powers
    const int tid = threadIdx.x;
    const int d = xcd_contiguous(blockIdx.x, gridDim.x);
    const size_t step = (size_t)T * kB;
    {
        const float4* const gA = pmA + (size_t)(2 * d) * kBinsA;
        const float4* const gA2 = sp.pmA2 + (size_t)(2 * d) * kBinsA;
        for (int i = tid; i < 2 * kBinsA; i += kB12Threads) { spec[i] = gA[i]; spec[2 * kBinsA + i] = gA2[i]; }
        if (tid < 16) {
            cf pw[15];
            fft::powers_of<16>(tw[tid * (fft::kTwiddleN / 256)], pw);
#pragma unroll
            for (int r = 0; r < 15; ++r) tw1[tid * 15 + r] = pw[r];
        }
    }
    constexpr int kPoller = 2 * 64;                                   // lane 0 of the first inverse wave
    // the doorbell as this workgroup may read it: workgroup 0 asks the host and passes the answer on, the others ask the relay
    auto read_door = [&]() -> unsigned {
        if (blockIdx.x == 0) {
            const unsigned v = __hip_atomic_load(eng.doorbell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(eng.relay, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return v;
        }
        return __hip_atomic_load(eng.relay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (tid == 0) {
        s_door[2] = 0;
        // has the launch become resident?  The first and the last workgroup to begin say so in host words: a wait that
        // runs out can then tell "never started" (something ahead of it on its hardware queue) and "some workgroups are
        // kept out" (waves of another launch hold registers or LDS on their compute units) from a silent producer
        const unsigned before = __hip_atomic_fetch_add(eng.started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (before == 0) __hip_atomic_store(&eng.resident[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (before + 1 == gridDim.x) __hip_atomic_store(&eng.resident[1], gridDim.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned rb = (unsigned)lane + ((unsigned)lane >> 4);      // Pad(lane + 64 r) = rb + 68 r
    // The aggregator — the first inverse wave of workgroup 1 (workgroup 0 where there is only one) — takes every inverse wave's count
    // of finished buffers (8 per lane, sc1 loads) and writes the minimum into `completed` (pinned host word), from the period loop
    // and from the idle loop below.  Not workgroup 0: that one's idle loop reads the doorbell over the link, two microseconds a look,
    // and a count that waits behind such a look reaches the host that much later.
    const bool aggregator = blockIdx.x == (gridDim.x > 1 ? 1u : 0u);
    unsigned reported = 0;                                            // (meaningful in that wave only)
    auto aggregate_request = [&](u4& a, u4& b) {
        const auto srd = __builtin_amdgcn_make_buffer_rsrc(eng.progress, 0, (int)(8u * gridDim.x), 0x00020000);
        a = __builtin_amdgcn_raw_buffer_load_b128(srd, 32u * (unsigned)lane, 0, 16);          // sc1; beyond the end: zeros dropped below
        b = __builtin_amdgcn_raw_buffer_load_b128(srd, 32u * (unsigned)lane + 16u, 0, 16);
    };
    auto aggregate_report = [&](const u4& a, const u4& b) {
        const unsigned words = 2u * gridDim.x;                        // lanes beyond the array read zeros: mask them out
        auto pick = [&](unsigned v, unsigned idx) { return idx < words ? v : 0xffffffffu; };
        unsigned m = min(min(min(pick(a[0], 8u * lane), pick(a[1], 8u * lane + 1)), min(pick(a[2], 8u * lane + 2), pick(a[3], 8u * lane + 3))),
                         min(min(pick(b[0], 8u * lane + 4), pick(b[1], 8u * lane + 5)), min(pick(b[2], 8u * lane + 6), pick(b[3], 8u * lane + 7))));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = min(m, (unsigned)__shfl_xor((int)m, o));
        if (m != reported) {
            reported = m;
            if (lane == 0) __hip_atomic_store(eng.completed, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    };
    // How many buffers may be touched, asked by EVERY wave — at the top of period nb of a burst (same answer in all of them:
    // it is read from LDS, written before the previous period's closing barrier), and with idle = true between bursts.
    // Batch launches: n_buffers.  The engine's doorbell word: bits 0-29 buffers published so far, bit 31 STOP (no more will
    // come), bit 30 FLUSH (finish what is published without waiting for more).  A period runs buffer nb when buffer nb + 1
    // is there too (its operands are requested one period ahead) — or, on STOP or FLUSH, when nb is the last one published:
    // that period requests nothing, the burst ends behind it with a drain period, and the workgroup idles here until the
    // doorbell moves (the next burst starts cold: a real-time caller with ONE buffer in flight rings FLUSH with every
    // buffer).  Returns the count published (> nb), or -1: the stop rung with nothing pending — the launch ends.
    auto gate = [&](int nb, bool idle) -> int {
        {
            if (idle && __builtin_amdgcn_readfirstlane(s_door[2]) != 0) return -1;   // the doorbell ran out of time in this burst: no further wait
            bool look = !idle;                                        // an idle gate asks first: the word in LDS is the one the last burst ended on
            for (;;) {
                if (look) {
                    // (the same word in every lane: said so, or every test on it becomes an exec-masked region — the far
                    // role's request burst under a divergent branch took 2.4 instead of 1.4 us of its barrier interval)
                    const unsigned D = __builtin_amdgcn_readfirstlane(s_door[nb & 1]);
                    const int pub = (int)(D & 0x3fffffffu);
                    const bool stop = (D >> 31) != 0, flush = ((D >> 30) & 1u) != 0;
                    if (pub >= nb + 2 || ((stop || flush) && pub >= nb + 1)) return pub;
                    if (stop) return -1;                              // nothing more will come
                }
                look = true;
                __syncthreads();                                      // every wave has read the word
                if (w == 2) {                                         // the first inverse wave polls (lane 0 asks; workgroup 0's also aggregates)
                    unsigned v = 0;
                    int tries = 0;
                    const unsigned long long t_poll = __builtin_amdgcn_s_memrealtime();
                    for (;;) {
                        u4 pa, pb;
                        if (aggregator && !GAB_EABL(1)) aggregate_request(pa, pb);
                        unsigned mine = 0;
                        if (lane == 0) mine = read_door();
                        v = __builtin_amdgcn_readfirstlane(mine);
                        if (aggregator && !GAB_EABL(1)) aggregate_report(pa, pb);
                        const int p2 = (int)(v & 0x3fffffffu);
                        if (p2 >= nb + 2 || (v >> 31) || (((v >> 30) & 1u) && p2 >= nb + 1)) break;
                        if ((++tries & 255) == 0 && __builtin_amdgcn_s_memrealtime() - t_poll > eng.idle_ticks) {   // the producer is gone: stop here, say so
                            if (lane == 0) {
                                __hip_atomic_store(eng.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                s_door[2] = 1;
                            }
                            v = 0x80000000u | (unsigned)(p2 < nb ? p2 : nb);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(6);                      // (~0.2 us between looks: with 20, a quarter of a microsecond more from doorbell to count)
                    }
                    if (lane == 0) s_door[nb & 1] = v;
                }
                __syncthreads();
            }
        }
    };
    // buffer nb lives in slot nb % ring of the engine's rings (a batch launch: buffer nb itself); callers walk the slots
    // with next_slot() instead of dividing
    auto in_slot = [&](int slot) -> const float* { return in + (size_t)slot * step; };
    auto next_slot = [&](int slot) -> int { return (ENGINE && slot + 1 == eng.ring) ? 0 : slot + 1; };
    // The engine's input ring is rewritten while the launch runs (by copy engines): its loads are system-scope loads,
    // answered by memory and never by a line an L1 or an L2 kept.  The rings are ORDINARY device memory (round 4, measured
    // at 1024 channels: fine-grained rings read by non-temporal loads 6.85 us per buffer, ordinary rings read by
    // system- or agent-scope loads 6.09-6.13, by plain loads — which may be stale — 6.2-6.3).
    auto ld = [](const float* p) -> float {
        if constexpr (ENGINE)
            return GAB_EABL(4)     ? *p
                   : GAB_EABL(256) ? __builtin_nontemporal_load(p)
                                   : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // sc0 sc1
        else return *p;
    };

    auto idle_period = [&]() { for (int i = 0; i < kBatchBarriers; ++i) __syncthreads(); };

    if (w >= 4) {
        // ---- far waves: group g turns pair g of the duo, one transform per two periods
        const int g = __builtin_amdgcn_readfirstlane((w - 4) >> 2);
        const int ft = (tid - kThreads) & (kThreads - 1);
        cf* const img = far_img + g * kLdsHalf;
        const int q = 2 * d + g;
        const cf* const hp = reinterpret_cast<const cf*>(hist) + (size_t)q * kSlots * kB;
        cf* const cg = sp.carry + (size_t)q * kCarrySlots * kB;
        const size_t ca = (size_t)(2 * q) * kB, cb_ = ca + kB;
        const float4* const pf = sp.pmF + (size_t)q * kBinsB;
        using B16 = fft::Butterfly<16, false>;
        using B16i = fft::Butterfly<16, true>;
        // the last pass's powers, W4096^(r t), are re-formed from their base where they are used (the same powers_of: the same
        // values; kept in 30 registers, as the batch launch keeps them, the engine's waves spilled)
        const cf tw2_base = tw[ft];
        auto tw2_of = [&](cf (&w)[15]) {
            cf b = tw2_base;
            asm volatile("" : "+v"(b.x), "+v"(b.y));                 // pinned to the pass
            fft::powers_of<16>(b, w);
        };
        // LDS positions are formed where they are used, from an opaque copy of the thread index: as loop invariants they
        // would be kept (and spilled) across both halves of a transform
        auto opaque_t = [&]() -> unsigned { unsigned v = (unsigned)ft; asm volatile("" : "+v"(v)); return v; };
        auto w0_of = [](unsigned t) -> unsigned { return t * 17u; };                       // pass-0 writes: Pad(16 t + r) = 17 t + r
        auto rd_of = [](unsigned t) -> unsigned { return t + (t >> 4); };                  // linear reads: Pad(t + 256 r) = rd + 272 r
        auto w1_of = [](unsigned t) -> unsigned { const unsigned b1 = (t >> 4) * 256u + (t & 15u); return b1 + (b1 >> 4); };   // pass-1 writes: + 17 r
        auto tw1row_of = [&](unsigned t) -> const cf* { return tw1 + (t & 15u) * 15u; };
        // window of buffer nb: blocks k-7 .. k — from the input buffers where they lie inside the launch, else the ring
        // The window of buffer nb in two halves of sixteen requests (PART 0: blocks k-3 .. k, PART 1: blocks k-7 .. k-4).  The newest
        // block comes from the input ring (system-scope loads: a copy engine wrote it); block k-1 from the input ring too where
        // the window is asked for a period ahead (its copy in the history ring is being written in that very period), from the
        // history ring at a COLD start (FIRST: the previous burst's last buffer; its ring slot may be the producer's again);
        // everything older from the history ring, which the forward waves write every period.
        auto load_window = [&](auto part_tag, auto first_tag, int nb, int slot, int slot_before, cf (&z)[16]) {
            constexpr int PART = decltype(part_tag)::value;
            constexpr bool FIRST = decltype(first_tag)::value;
            const int head = (head0 + nb) & (kSlots - 1);
            int fo = ft;
            asm volatile("" : "+v"(fo));                              // (addresses formed here, not hoisted and spilled)
            if constexpr (PART == 0) {
                const float* const cur = in_slot(slot);
                z[14] = mk(ld(cur + ca + fo), ld(cur + cb_ + fo));
                z[15] = mk(ld(cur + ca + fo + kThreads), ld(cur + cb_ + fo + kThreads));
                if constexpr (!FIRST) {
                    const float* const prv = in_slot(slot_before);
                    z[12] = mk(ld(prv + ca + fo), ld(prv + cb_ + fo));
                    z[13] = mk(ld(prv + ca + fo + kThreads), ld(prv + cb_ + fo + kThreads));
                } else {
                    const int s6 = ((head + kSlots - 1) & (kSlots - 1)) * kB;
                    z[12] = hp[s6 + fo];
                    z[13] = hp[s6 + kThreads + fo];
                }
#pragma unroll
                for (int bl = 4; bl < 6; ++bl) {
                    const int sb = ((head + 1 + bl) & (kSlots - 1)) * kB;
                    z[2 * bl] = hp[sb + fo];
                    z[2 * bl + 1] = hp[sb + kThreads + fo];
                }
            } else {
#pragma unroll
                for (int bl = 0; bl < 4; ++bl) {
                    const int sb = ((head + 1 + bl) & (kSlots - 1)) * kB;
                    z[2 * bl] = hp[sb + fo];
                    z[2 * bl + 1] = hp[sb + kThreads + fo];
                }
            }
        };
        cf z[16], zn[16];
        // The requests ride on the lighter steps (every instruction of a step costs the wave about a dozen clocks beside two
        // others on its SIMD: thirty-two requests in one interval made it the period's longest), the wait for the carry's stores
        // stands in the idle interval.  (Measured and not kept, profiles/r06_batch12_stamps.txt: steps cut as read + twiddle |
        // butterfly + write, so that no step is a bare write — 5.08 against 5.01 us per buffer.)
        // first half of a transform (period nb):   pass 0, write | read, twiddle, pass 1 | write (+ spectra 0-7) |
        //                                          read, twiddle (+ spectra 8-15), pass 2 | partner write | partner read, product, inverse pass 0
        auto first_half = [&](int nb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = zn[r];
            B16::run(z);
            {
                const unsigned w0 = w0_of(opaque_t());
#pragma unroll
                for (int r = 0; r < 16; ++r) img[w0 + r] = z[B16::out_slot(r)];
            }
            GAB_E12BAR(0);                                          // 1
            {
                const unsigned t = opaque_t(), rd = rd_of(t);
                const cf* const tw1row = tw1row_of(t);
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = img[rd + 272 * r];
#pragma unroll
                for (int r = 1; r < 16; ++r) z[r] = fft::cmul(z[r], tw1row[r - 1]);
            }
            B16::run(z);
            GAB_E12BAR(1);                                          // 2: every wave has read
            {
                const unsigned w1 = w1_of(opaque_t());
#pragma unroll
                for (int r = 0; r < 16; ++r) img[w1 + 17 * r] = z[B16::out_slot(r)];
            }
            __builtin_amdgcn_sched_barrier(0);
            float4 clo[8], chi[8];                                    // the far spectra: bins r < 8 asked for here, r >= 8 an interval later
            {
                int fo = ft;
                asm volatile("" : "+v"(fo));
                load_spectra_part<kNB, 16, 0, 8>(clo, pf, fo);
            }
            __builtin_amdgcn_sched_barrier(0);
            GAB_E12BAR(2);                                          // 3
            {
                const unsigned rd = rd_of(opaque_t());
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = img[rd + 272 * r];
            }
            {
                cf tw2[15];
                tw2_of(tw2);
#pragma unroll
                for (int r = 1; r < 16; ++r) z[r] = fft::cmul(z[r], tw2[r - 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                int fo = ft;
                asm volatile("" : "+v"(fo));
                load_spectra_part<kNB, 16, 8, 16>(chi, pf, fo);
            }
            __builtin_amdgcn_sched_barrier(0);
            B16::run(z);
            GAB_E12BAR(3);                                          // 4
            {
                const unsigned t = opaque_t();
#pragma unroll
                for (int r = 0; r < 16; ++r) img[t + 256u * r] = z[B16::out_slot(r)];   // Z[t + 256 r]: the partner exchange, raw
            }
            GAB_E12BAR(4);                                          // 5
            {
                // the thread's own bins back in order, and Z[(N - k) mod N], k = t + 256 r: one base and constant offsets for
                // r >= 1 (N - k > 0 there); bin k = t alone wraps
                cf o[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = z[B16::out_slot(r)];
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = o[r];            // z[r] = Z[t + 256 r]
                const unsigned t = opaque_t();
                const cf* const pb = img + (kNB - 256 * 15) - t;      // pb[256 (15 - r)] = img[N - t - 256 r]
                cf zp[8];
                zp[0] = img[(kNB - t) & (kNB - 1)];
#pragma unroll
                for (int r = 1; r < 8; ++r) zp[r] = pb[256 * (15 - r)];
                spectral_product_part<kNB, 16, 0, 8>(z, zp, clo, ft);
#pragma unroll
                for (int r = 0; r < 8; ++r) zp[r] = pb[256 * (7 - r)];
                spectral_product_part<kNB, 16, 8, 16>(z, zp, chi, ft);
            }
            B16i::run(z);
            GAB_E12BAR(5);                                          // 6
        };
        // second half (period nb + 1):   (half of the next window's requests) | write | read, twiddle, pass 1 |
        //                                write (+ the other half) | read, twiddle, last pass (4 of 16), carry out | (the carry's stores leave)
        auto second_half = [&](int period, int nb_done, int nb_next, int slot_next, int slot_now) {
            const int nb = period;                                    // (the stamps' period)
            (void)nb;
            if (nb_next >= 0) {
                load_window(std::integral_constant<int, 0>{}, std::false_type{}, nb_next, slot_next, slot_now, zn);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) zn[r] = mk(0.0f, 0.0f);  // (no value survives from the last window: registers)
            }
            GAB_E12BAR(0);                                          // 1
            if (nb_done >= 0) {
                const unsigned w0 = w0_of(opaque_t());
#pragma unroll
                for (int r = 0; r < 16; ++r) img[w0 + r] = z[B16i::out_slot(r)];
            }
            GAB_E12BAR(1);                                          // 2
            if (nb_done >= 0) {
                const unsigned t = opaque_t(), rd = rd_of(t);
                const cf* const tw1row = tw1row_of(t);
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = img[rd + 272 * r];
#pragma unroll
                for (int r = 1; r < 16; ++r) z[r] = fft::cmulc(z[r], tw1row[r - 1]);
                B16i::run(z);
            }
            GAB_E12BAR(2);                                          // 3
            if (nb_done >= 0) {
                const unsigned w1 = w1_of(opaque_t());
#pragma unroll
                for (int r = 0; r < 16; ++r) img[w1 + 17 * r] = z[B16i::out_slot(r)];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (nb_next >= 0) load_window(std::integral_constant<int, 1>{}, std::false_type{}, nb_next, slot_next, slot_now, zn);      // the window's other half, beside the light write
            __builtin_amdgcn_sched_barrier(0);
            GAB_E12BAR(3);                                          // 4
            if (nb_done >= 0) {
                const unsigned rd = rd_of(opaque_t());
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = img[rd + 272 * r];
                {
                    cf tw2[15];
                    tw2_of(tw2);
#pragma unroll
                    for (int r = 1; r < 16; ++r) z[r] = fft::cmulc(z[r], tw2[r - 1]);
                }
                cf x12, x13, x14, x15;
                B16i::run_last4(z, x12, x13, x14, x15);
                const int head = (head0 + nb_done) & (kSlots - 1);
                cf* const c1 = cg + ((head + 1) & (kCarrySlots - 1)) * kB;    // block k+1
                cf* const c2 = cg + ((head + 2) & (kCarrySlots - 1)) * kB;    // block k+2
                c1[ft] = x12;
                c1[ft + kThreads] = x13;
                c2[ft] = x14;
                c2[ft + kThreads] = x15;
            }
            GAB_E12BAR(4);                                          // 5
            // the inverse wave of this pair asks for block k+1's share behind the period's closing barrier: the stores must
            // have left this wave by then (waited for HERE, in the group's idle interval, not on its last pass)
            if (nb_done >= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            GAB_E12BAR(5);                                          // 6
        };
        // A group's state from period to period: a transform begun last period (to be finished in this one), a window asked for
        // last period (for this one).  A burst starts cold; the group whose turn it is loads its window there and then.
        bool in_flight = false, have_window = false;
        int nb = 0, slot = 0;
        for (;;) {                                                    // bursts
            int avail = gate(nb, true);
            if (nb >= avail) break;
            for (;;) {                                                // the burst's periods
                const bool mine = ((head0 + nb) & 1) == g;
                if (mine) {
                    if (!have_window) {
                        load_window(std::integral_constant<int, 0>{}, std::true_type{}, nb, slot, 0, zn);
                        load_window(std::integral_constant<int, 1>{}, std::true_type{}, nb, slot, 0, zn);
                    }
                    first_half(nb);
                    in_flight = true;
                    have_window = false;
                } else {
                    const bool can = nb + 1 < avail;                  // buffer nb + 1 is published: its window can be asked for
                    second_half(nb, in_flight ? nb - 1 : -1, can ? nb + 1 : -1, next_slot(slot), slot);
                    have_window = can;
                    in_flight = false;
                }
                ++nb;
                slot = next_slot(slot);
                if (nb >= avail) break;                               // nothing was published for buffer nb: the burst ends here
                avail = gate(nb, false);
                if (nb >= avail) break;                               // (the doorbell ran out of time)
            }
            // the burst's drain period: a transform begun in the burst's last period is finished here
            if (in_flight) second_half(nb, nb - 1, -1, 0, 0);
            else idle_period();
            in_flight = false;
            have_window = false;
        }
    } else if (w < 2) {
        // ---- forward waves: wave w holds pair w of the duo
        const int q = 2 * d + w;
        cf* const hp = reinterpret_cast<cf*>(hist) + (size_t)q * kSlots * kB;
        cf* const img = lds + w * kWaveImg;                           // the transform's exchanges, then its spectrum
        cf* const hand = lds + (2 + w) * kWaveImg;                    // A2 share, then the output spectrum for the inverse wave
        const float4* const sa = spec + w * kBinsA;                   // taps [0,512)
        const float4* const sa2 = spec + (2 + w) * kBinsA;            // taps [512,1024)
        using WF = fft::WaveFFT1024<false>;
        WF::Lean t;
        WF::load_twiddles(t, tw, lane);
        const size_t xoff = (size_t)(2 * q) * kB;                     // channel a of a buffer; channel b is kB further
        cf z[16], prev[8], nxt[8];
        // W = Z x spectra (from LDS) in halves of eight bins: partner values and spectra of a half live in registers at a time
        auto product_from_image = [&](cf (&v)[16], const float4* sp_lds) {
            {
                cf zp[8];
                float4 ch[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) zp[r] = img[PadA16::at((kNA - (lane + 64 * r)) & (kNA - 1))];
                load_spectra_part<kNA, 16, 0, 8>(ch, sp_lds, lane);
                spectral_product_part<kNA, 16, 0, 8>(v, zp, ch, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                cf zp[8];
                float4 ch[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) zp[r] = img[PadA16::at((kNA - (lane + 64 * (8 + r))) & (kNA - 1))];
                load_spectra_part<kNA, 16, 8, 16>(ch, sp_lds, lane);
                spectral_product_part<kNA, 16, 8, 16>(v, zp, ch, lane);
            }
        };
        auto fwd_period = [&](int nb, int slot, int avail) {
            // A2 share | hand-over + window + pass 0 | pass 1 | pass 2 | spectrum + A product | sum into the hand-over
            const int head = (head0 + nb) & (kSlots - 1);
            {
                cf share[16];                                         // taps [512,1024): last period's spectrum x its spectra
#pragma unroll
                for (int r = 0; r < 16; ++r) share[r] = img[rb + 68 * r];
                product_from_image(share, sa2);
                GAB_E12BAR(0);                                   // barrier 1: the inverse wave has read the hand-over image
#pragma unroll
                for (int r = 0; r < 16; ++r) hand[rb + 68 * r] = share[r];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 8; ++j) { z[j] = prev[j]; z[8 + j] = nxt[j]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) hp[head * kB + lane + 64 * j] = nxt[j];   // the history ring, every period (the last eight buffers are not known in advance)
#pragma unroll
            for (int j = 0; j < 8; ++j) prev[j] = nxt[j];
            if (nb + 1 < avail) {                                     // the next buffer's block: needed a period from now
                int lo = lane;
                asm volatile("" : "+v"(lo));
                const float* const xa = in_slot(next_slot(slot)) + xoff + lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) nxt[j] = mk(ld(xa + 64 * j), ld(xa + kB + 64 * j));
            }
            WF::run(z, img, t, lane, [&](int i) { GAB_E12BAR(1 + i); });   // barriers 2, 3 from inside
            GAB_E12BAR(3);                                       // barrier 4
#pragma unroll
            for (int r = 0; r < 16; ++r) img[rb + 68 * r] = z[r];     // the spectrum stays here for the next period
            __builtin_amdgcn_wave_barrier();
            product_from_image(z, sa);
            GAB_E12BAR(4);                                       // barrier 5
#pragma unroll
            for (int r = 0; r < 16; ++r) hand[rb + 68 * r] = fft::cadd(z[r], hand[rb + 68 * r]);   // A product + A2 share
            GAB_E12BAR(5);                                       // barrier 6 closes the period
        };
        int nb = 0, slot = 0;                                         // the next buffer and its ring slot
        bool primed = false;                                          // the image holds the spectrum of blocks [k-2 | k-1]
        for (;;) {                                                    // bursts
            int avail = gate(nb, true);
            if (nb >= avail) break;
            if (!primed) {
                // the launch's first burst: the spectrum of the ring's blocks [k-2 | k-1] into the image (later bursts find it there)
                const int s1 = ((head0 + kSlots - 1) & (kSlots - 1)) * kB, s2 = ((head0 + kSlots - 2) & (kSlots - 1)) * kB;
                int lo = lane;
                asm volatile("" : "+v"(lo));
#pragma unroll
                for (int j = 0; j < 8; ++j) z[j] = hp[s2 + lo + 64 * j];
#pragma unroll
                for (int j = 0; j < 8; ++j) prev[j] = hp[s1 + lo + 64 * j];
#pragma unroll
                for (int j = 0; j < 8; ++j) z[8 + j] = prev[j];
                WF::run(z, img, t, lane, WF::NoHook());
#pragma unroll
                for (int r = 0; r < 16; ++r) img[rb + 68 * r] = z[r];
                __builtin_amdgcn_wave_barrier();
                primed = true;
            }
            {                                                         // the burst's first block
                int lo = lane;
                asm volatile("" : "+v"(lo));
                const float* const xa = in_slot(slot) + xoff + lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) nxt[j] = mk(ld(xa + 64 * j), ld(xa + kB + 64 * j));
            }
            for (;;) {
                fwd_period(nb, slot, avail);
                ++nb;
                slot = next_slot(slot);
                if (nb >= avail) break;
                avail = gate(nb, false);
                if (nb >= avail) break;
            }
            idle_period();                                            // the burst's drain period
        }
    } else {
        // ---- inverse waves: wave 2 + p turns the output spectrum of pair p into samples, one period later
        const int pr = w - 2;
        const cf* const hand = lds + (2 + pr) * kWaveImg;
        cf* const img = lds + (4 + pr) * kWaveImg;                    // the transform's exchanges, then the output swap
        const cf* const other = lds + (4 + (1 - pr)) * kWaveImg;
        const cf* const cg = sp.carry + (size_t)(2 * d + pr) * kCarrySlots * kB;   // the far share comes through memory (agent-scope loads)
        using WFi = fft::WaveFFT1024<true>;
        WFi::Lean t;
        WFi::load_twiddles(t, tw, lane);
        int oslot = 0;                                                // of the next buffer to leave
        int nb = 0, avail = 0, base = 0;                              // the period, buffers that may be touched, the burst's first buffer
        bool boundary = true;                                         // between bursts (the launch's start is a boundary)
        for (;;) {                                                    // periods base .. end of every burst; in a burst's last one only this role works (the drain)
            if (boundary) {
                avail = gate(nb, true);
                if (nb >= avail) break;
                base = nb;                                            // nothing to turn in a burst's first period
                boundary = false;
            } else if (nb < avail) {
                avail = gate(nb, false);                              // (nb == avail: the drain period — the other roles ask nothing either)
            }
            const bool more = nb < avail;                         // the other roles work on buffer nb in this period
            unsigned door_next = s_door[nb & 1];                  // (no per-period poll: the word as last seen)
            u4 prog_a = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, prog_b = prog_a;
            if constexpr (ENGINE) {
                if (nb >= base + 2 && !GAB_EABL(1)) {
                    // this wave's rows of buffer nb - 2 were stored a period ago: drained by now, so the wait is free,
                    // and the count of finished buffers can go out (write-through, nobody waits for it)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef GAB_ENGWB
                    asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
#endif
#ifdef GAB_ENGWB8
                    if (blockIdx.x < 8 && pr == 1) asm volatile("buffer_wbl2 sc1" ::: "memory");
#endif
                    if (lane == 0) __hip_atomic_store(&eng.progress[2 * blockIdx.x + pr], (unsigned)(nb - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (tid == kPoller && eng.poll_every_period)      // asked now, needed at the period's end
                    door_next = blockIdx.x == 0 ? __hip_atomic_load(eng.doorbell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                                : __hip_atomic_load(eng.relay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (aggregator && pr == 0 && !GAB_EABL(1)) aggregate_request(prog_a, prog_b);
            }
            auto close_period = [&]() {                           // before the closing barrier
                if constexpr (ENGINE) {
                    if (aggregator && pr == 0 && !GAB_EABL(1)) aggregate_report(prog_a, prog_b);
                    if (tid == kPoller) {
                        s_door[(nb + 1) & 1] = door_next;
                        if (blockIdx.x == 0 && eng.poll_every_period)
                            __hip_atomic_store(eng.relay, door_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            };
            bool idle_period = nb == base;                        // a burst's first period: nothing to turn yet
#ifdef GAB_ABLATE
            if (GAB_SDBG(1)) idle_period = true;                  // diagnostic builds: near role idle
#endif
            if (idle_period) {
                for (int i = 0; i < kBatchBarriers - 1; ++i) __syncthreads();
                close_period();
                __syncthreads();
            } else {
                // One piece per barrier interval: hand-over read | pass 0 | pass 1 | pass 2 + far share | swap | stores
                const int b = nb - 1;                             // the buffer whose spectrum was handed over last period
                const int head = (head0 + b) & (kSlots - 1);
                float* const outb = out + (size_t)(ENGINE ? oslot : b) * step;
                oslot = next_slot(oslot);
                cf z[16], y[8], park[8];
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = hand[rb + 68 * r];    // the forward wave writes the next one in interval 6
                {
                    int lo = lane;
                    asm volatile("" : "+v"(lo));
                    const unsigned long long* const cy = reinterpret_cast<const unsigned long long*>(cg + (head & (kCarrySlots - 1)) * kB + lo);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const unsigned long long v = __hip_atomic_load(cy + 64 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        park[j] = mk(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
                    }
                }
                GAB_E12BAR(0);                                    // barrier 1
                WFi::run(z, img, t, lane, [&](int i) { GAB_E12BAR(1 + i); });   // barriers 2, 3 from inside
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] = fft::cadd(z[8 + j], park[j]);
                GAB_E12BAR(3);                                    // barrier 4
#ifdef GAB_ENGPUSH
                // experiment (the cost alone, no protocol): this wave's share of the eight-workgroup line group's rows of the buffer
                // stored FOUR periods ago is read back from L2 here and written through as whole 128-byte lines behind the stores
                u4 pl[4] = {};
                const bool push = nb >= base + 5;
                char* pbase = nullptr;
                if (push) {
                    int ps = oslot - 4;
                    if (ps < 0) ps += eng.ring;
                    pbase = reinterpret_cast<char*>(out + (size_t)ps * step) + (size_t)(d >> 3) * 128 +
                            (size_t)((((d & 7) * 2 + pr) * 32) + (lane >> 3)) * T * 4 + (lane & 7) * 16;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(pl[j]) : "v"(pbase + (size_t)8 * j * T * 4) : "memory");
                }
#endif
                // the two pairs of a duo are four neighbouring channels: the waves swap halves through LDS
                // so that each stores float4 pieces (pair 0 keeps samples lane + 64 j, j < 4, pair 1 j >= 4)
                if (pr == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) img[lane + 64 * j] = y[4 + j];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) img[lane + 64 * j] = y[j];
                }
                GAB_E12BAR(4);                                    // barrier 5: the swapped halves are in LDS
                {
                    float* const o0 = outb + 4 * (size_t)d;
                    auto put = [&](float* dst, float a, float b2, float c2, float d2) {
                        if (ENGINE && !GAB_EABL(2)) {             // write-through: in memory before `completed` says so
                            typedef float f4v __attribute__((ext_vector_type(4)));
                            const f4v val = {a, b2, c2, d2};
                            asm volatile("global_store_dwordx4 %0, %1, off " GAB_ENGSTORE_BITS ::"v"(dst), "v"(val) : "memory");
                        } else {
                            *reinterpret_cast<float4*>(dst) = make_float4(a, b2, c2, d2);
                        }
                    };
                    if (pr == 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const cf theirs = other[lane + 64 * j];
                            put(o0 + (size_t)T * (lane + 64 * j), y[j].x, y[j].y, theirs.x, theirs.y);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const cf theirs = other[lane + 64 * j];
                            put(o0 + (size_t)T * (lane + 64 * (4 + j)), theirs.x, theirs.y, y[4 + j].x, y[4 + j].y);
                        }
                    }
                }
#ifdef GAB_ENGPUSH
                if (push) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (loads and stores may come back out of order with each other)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(pbase + (size_t)8 * j * T * 4), "v"(pl[j]) : "memory");
                }
#endif
                close_period();
                GAB_E12BAR(5);                                    // barrier 6 closes the period
            }
            if (!more) {
                // the burst is through: this wave's last rows must be in memory before its count says so (the one wait for
                // stores on this path: a workgroup that goes idle has nothing to hide it behind)
                if (!GAB_EABL(1)) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(&eng.progress[2 * blockIdx.x + pr], (unsigned)nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                boundary = true;
                continue;
            }
            ++nb;
        }
    }
}
#undef GAB_E12BAR

__global__ __launch_bounds__(kB12Threads) void conv_split_engine12_kernel(
    const float* __restrict__ in, float* __restrict__ out, float* __restrict__ hist,
    const float4* __restrict__ pmA, ConvSplit sp, const cf* __restrict__ tw, int T, int head0, ConvEngine eng) {
    __shared__ __attribute__((aligned(16))) cf lds[kB12Lds];
    __shared__ unsigned door[4];          // [0], [1] the doorbell as last seen, by period parity; [2] the engine has given up
    conv_split_engine12_resident(in, out, hist, pmA, sp, tw, T, head0, eng, lds, door);
}

// IR bank -> (P, M) spectra of a near (512 taps from offA) and a far (taps from offB) partition.
// d_ir is T x L track-major.
__global__ __launch_bounds__(kThreads) void conv_ir_spectra_kernel(
    const float* __restrict__ ir, float4* __restrict__ pmA, float4* __restrict__ pmB,
    const cf* __restrict__ tw, int T, int L, int offA, int offB) {
    __shared__ cf lds[2 * kLdsHalf];
    cf* const lds0 = lds;
    cf* const lds1 = lds + kLdsHalf;
    const int tid = threadIdx.x;
    const int q = blockIdx.x;
    const int ta = 2 * q, tb = 2 * q + 1;
    const bool hasb = tb < T;
    const float* ia = ir + (size_t)ta * L;
    const float* ib = ir + (size_t)tb * L;

    auto tap = [&](int j) {
        return (j < L) ? mk(ia[j], hasb ? ib[j] : 0.0f) : mk(0.0f, 0.0f);
    };
    auto emit = [&](cf z, cf zp, float scale, float4* dst) {
        // Ha = (Z + Z')/2, Hb = (Z - Z')/(2i); P = (Ha+Hb)/2, M = (Ha-Hb)/2
        cf s = fft::cadd(z, zp), d = fft::csub(z, zp);
        cf Ha = mk(0.5f * s.x, 0.5f * s.y);
        cf Hb = mk(0.5f * d.y, -0.5f * d.x);
        *dst = make_float4(0.5f * scale * (Ha.x + Hb.x), 0.5f * scale * (Ha.y + Hb.y),
                           0.5f * scale * (Ha.x - Hb.x), 0.5f * scale * (Ha.y - Hb.y));
    };

    {   // near partition: taps [offA, offA+512) zero-padded to 1024
        cf z[4], zp[4];
        z[0] = tap(offA + tid);
        z[1] = tap(offA + tid + kThreads);
        z[2] = mk(0.0f, 0.0f);
        z[3] = mk(0.0f, 0.0f);
        fft::BlockFFT<kNA, 4, false>::Twiddles t;
        fft::BlockFFT<kNA, 4, false>::load_twiddles(t, tw, tid);
        fft::BlockFFT<kNA, 4, false>::run(z, lds0, lds1, t, tid);
        partner_exchange<kNA, 4>(z, zp, lds0, tid);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int k = tid + r * kThreads;
            if (k <= kNA / 2) emit(z[r], zp[r], 1.0f / kNA, pmA + (size_t)q * kBinsA + k);
        }
    }
    __syncthreads();
    if (pmB != nullptr) {   // far partition: taps [offB, 4096) zero-padded to 4096
        cf z[16], zp[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int m = tid + r * kThreads;
            z[r] = (m < kNB - offB) ? tap(offB + m) : mk(0.0f, 0.0f);
        }
        fft::BlockFFT<kNB, 16, false>::Twiddles t;
        fft::BlockFFT<kNB, 16, false>::load_twiddles(t, tw, tid);
        fft::BlockFFT<kNB, 16, false>::run(z, lds0, lds1, t, tid);
        partner_exchange<kNB, 16>(z, zp, lds0, tid);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int k = tid + r * kThreads;
            if (k <= kNB / 2) emit(z[r], zp[r], 1.0f / kNB, pmB + (size_t)q * kBinsB + k);
        }
    }
}

// Batched 1024-point R2C (cuda/bench_fft.cu:63,105): two tracks per complex transform,
// one transform per WAVE (64 lanes x 16 values, three passes, exchanges private to the
// wave: no workgroup barrier anywhere), four waves per workgroup.
__global__ __launch_bounds__(kThreads) void fft_r2c_1024_kernel(
    const float* __restrict__ in, float2* __restrict__ out, const cf* __restrict__ tw, int T) {
    __shared__ cf lds[4 * kWaveImg];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    cf* const img = lds + w * kWaveImg;
    const int q = blockIdx.x * 4 + w;
    const int ta = 2 * q, tb = 2 * q + 1;
    if (ta >= T) return;                                   // whole wave; no barriers below
    const bool hasb = tb < T;
    const float* xa = in + (size_t)ta * kNA;
    const float* xb = in + (size_t)tb * kNA;
    cf z[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = mk(xa[lane + 64 * r], hasb ? xb[lane + 64 * r] : 0.0f);
    using WF = fft::WaveFFT1024<false>;
    WF::Twiddles t;
    WF::load_twiddles(t, tw, lane);
    WF::run(z, img, t, lane);
    // partner Z[(N - k) mod N], k = lane + 64 r: bins 0..512 are r < 8 plus (r = 8, lane 0)
    const unsigned rb = (unsigned)lane + ((unsigned)lane >> 4);
#pragma unroll
    for (int r = 0; r < 16; ++r) img[rb + 68 * r] = z[r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const int k = lane + 64 * r;
        if (k <= kNA / 2) {
            const cf zp = fft::conj(img[PadA16::at((kNA - k) & (kNA - 1))]);
            const cf s = fft::cadd(z[r], zp), d = fft::csub(z[r], zp);
            out[(size_t)ta * kBinsA + k] = make_float2(0.5f * s.x, 0.5f * s.y);
            if (hasb) out[(size_t)tb * kBinsA + k] = make_float2(0.5f * d.y, -0.5f * d.x);
        }
    }
}

// ---- uniform partitions: every other power-of-two buffer size, impulse responses up to 16384 taps ----
// The reference sizes ONE transform as nextpow2(L + B - 1) for any (ir_length, buffer_size)
// (cuda/bench_conv1d_accel.cu:49-53).  Here the taps are cut into J partitions, each convolved with
// its own 4096-sample window of the history by the same 4096-point transform:
//   partition 0      taps [0, B)                          window ends at the newest sample
//   partition p >= 1 taps [B + (p-1) S, B + p S), S = 4096 - B   window ends B + (p-1) S samples earlier
// (S is a multiple of B, so windows start on block boundaries of the stream).  The J spectral
// products are summed before ONE inverse transform, of which the last B outputs are the block.
// Partition 0 is kept to the B taps that can reach a first buffer: every other window then holds
// nothing but history, so after a reset their products are exact zeros and the first buffer is the
// reference's golden to rounding of the small leading taps (the large centre taps of a long
// response would otherwise bury it under their round-off).
// State: a time-domain ring of ring_len samples per channel pair; a sample's ring index is its
// absolute index in the stream modulo ring_len, so no block arithmetic is needed anywhere.
// Bytes per buffer and pair: J x (32 KiB window + 32.8 KB spectra) + 8 B x B in + out + ring.
struct ConvUniform {
    const float4* pm;      // [J][pairs][kBinsB]
    cf* ring;              // [pairs][ring_len]
    int J, S, ring_mask;
    unsigned pos;          // absolute index (mod ring_len) of the new block's first sample
};

template <bool STREAM>
__global__ __launch_bounds__(kThreads, 2) void conv_uniform_kernel(
    const float* __restrict__ in, float* __restrict__ out, ConvUniform u, const cf* __restrict__ tw,
    int T, int B) {
    __shared__ cf lds[2 * kLdsHalf];
    cf* const X = lds;
    cf* const Y = lds + kLdsHalf;
    const int tid = threadIdx.x;
    const int q = blockIdx.x;
    const int ta = 2 * q, tb = ta + 1;
    const bool hasb = tb < T;
    const int pairs = gridDim.x;
    using FB = fft::BlockFFT<kNB, 16, false>;
    using FBi = fft::BlockFFT<kNB, 16, true>;
    typename FB::Bases twb;
    FB::load_twiddles(twb, tw, tid);
    cf* const ring = u.ring + (size_t)q * (u.ring_mask + 1);
    const float* const xa = in + (size_t)ta * B;
    const float* const xb = in + (size_t)tb * B;
    const int newest = kNB - B;                    // window positions >= newest are the new block (partition 0)

    if (STREAM) {
        // The new block enters the ring FIRST (its slot holds the oldest samples, which no window reaches), so
        // that every window — partition 0's too — is sixteen unconditional loads from one place.  (Round 2 read
        // partition 0's newest samples from the input buffer under a per-element condition: each such load was
        // followed by a register merge, i.e. an s_waitcnt vmcnt(0) — sixteen dependent round trips per window.)
        for (int s = tid; s < B; s += kThreads)
            ring[(u.pos + (unsigned)s) & (unsigned)u.ring_mask] = mk(xa[s], hasb ? xb[s] : 0.0f);
        __threadfence_block();                               // the stores are out of this CU's write path ...
        __syncthreads();                                      // ... before any wave of the workgroup reads them back
    }
    cf acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = mk(0.0f, 0.0f);
    // (a run-time 1 in stateless mode: as a compile-time constant the loop is peeled and the allocator spills 1 KB)
    const int J = STREAM ? u.J : (u.J > 0 ? 1 : 0);
    for (int j = 0; j < J; ++j) {
        const unsigned back = j == 0 ? 0u : (unsigned)(B + (j - 1) * u.S);      // how far this window ends before the newest sample
        cf z[16];
        float zero = 0.0f;                                   // opaque: a literal zero history lets the compiler specialise the
        asm volatile("" : "+v"(zero));                       // first pass sixteen ways (1 KB of scratch in the stateless form)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = tid + kThreads * r;
            if (STREAM) {
                z[r] = ring[(u.pos + (unsigned)(n - newest) - back) & (unsigned)u.ring_mask];
            } else if (n >= newest) {                        // stateless: the new block over a zero history
                const int s = n - newest;
                z[r] = mk(xa[s], hasb ? xb[s] : 0.0f);
            } else {
                z[r] = mk(zero, zero);
            }
        }
        float4 c[16];
        {
            int to = tid;                                    // opaque: keeps sixteen 64-bit addresses out of long-lived registers
            asm volatile("" : "+v"(to));
            load_spectra<kNB, 16>(c, u.pm + ((size_t)j * pairs + q) * kBinsB, to);
        }
        FB::run(z, X, Y, twb, tid);                          // last reads Y
        cf zp[16];
        partner_exchange<kNB, 16, true>(z, zp, X, tid);      // writes X, barrier, reads X
        spectral_product<kNB, 16>(z, zp, c, tid);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = fft::cadd(acc[r], z[r]);
        __syncthreads();                                      // X's readers are done before the next forward writes it
    }
    FBi::run(acc, X, Y, twb, tid);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = tid + kThreads * r;
        if (n >= newest) {
            float* o = out + (size_t)T * (n - newest) + ta;
            o[0] = acc[r].x;
            if (hasb) o[1] = acc[r].y;
        }
    }
}

// (P, M) spectra of taps [off, off + count) of every pair, zero-padded to 4096 (1/N folded in).
__global__ __launch_bounds__(kThreads) void conv_ir_spectra_uniform_kernel(
    const float* __restrict__ ir, float4* __restrict__ pm, const cf* __restrict__ tw, int T, int L,
    int off, int count) {
    __shared__ cf lds[2 * kLdsHalf];
    const int tid = threadIdx.x;
    const int q = blockIdx.x;
    const int ta = 2 * q, tb = ta + 1;
    const bool hasb = tb < T;
    const float* ia = ir + (size_t)ta * L;
    const float* ib = ir + (size_t)tb * L;
    cf z[16], zp[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = tid + r * kThreads;
        const int j = off + m;
        z[r] = (m < count && j < L) ? mk(ia[j], hasb ? ib[j] : 0.0f) : mk(0.0f, 0.0f);
    }
    fft::BlockFFT<kNB, 16, false>::Twiddles t;
    fft::BlockFFT<kNB, 16, false>::load_twiddles(t, tw, tid);
    fft::BlockFFT<kNB, 16, false>::run(z, lds, lds + kLdsHalf, t, tid);
    partner_exchange<kNB, 16>(z, zp, lds, tid);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = tid + r * kThreads;
        if (k <= kNB / 2) {
            cf sm = fft::cadd(z[r], zp[r]), df = fft::csub(z[r], zp[r]);
            cf Ha = mk(0.5f * sm.x, 0.5f * sm.y);
            cf Hb = mk(0.5f * df.y, -0.5f * df.x);
            const float sc = 0.5f / kNB;
            pm[(size_t)q * kBinsB + k] = make_float4(sc * (Ha.x + Hb.x), sc * (Ha.y + Hb.y),
                                                     sc * (Ha.x - Hb.x), sc * (Ha.y - Hb.y));
        }
    }
}

// Direct-form fallback for shapes the fused path does not cover (bufsize != 512
// or ir_len > 4096): y[T*s+t] = sum_{k<L} x_hist[s-k] h[k], history ring of
// `hlen` samples per track (hlen >= L-1, multiple of bufsize).  One thread per
// output, k ascending like the golden.
template <bool STREAM>
__global__ void conv_direct_kernel(const float* __restrict__ in, float* __restrict__ out,
                                   const float* __restrict__ hist_in, const float* __restrict__ ir,
                                   int T, int B, int L, int hlen) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    int t = blockIdx.y;
    if (s >= B) return;
    const float* x = in + (size_t)t * B;
    const float* h = ir + (size_t)t * L;
    const float* hs = hist_in + (size_t)t * hlen;   // oldest first, linear (not a ring)
    float acc = 0.0f;
    for (int k = 0; k < L; ++k) {
        int i = s - k;
        float xv;
        if (i >= 0) xv = x[i];
        else if (STREAM && hlen + i >= 0) xv = hs[hlen + i];
        else break;
        acc = __builtin_fmaf(xv, h[k], acc);
    }
    out[(size_t)T * s + t] = acc;
}

// history <- (history << B) | new block     (fallback path only)
__global__ void conv_direct_shift_kernel(const float* __restrict__ in, const float* __restrict__ hist_in,
                                         float* __restrict__ hist_out, int T, int B, int hlen) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int t = blockIdx.y;
    if (i >= hlen) return;
    float v = (i + B < hlen) ? hist_in[(size_t)t * hlen + i + B] : in[(size_t)t * B + (i + B - hlen)];
    hist_out[(size_t)t * hlen + i] = v;
}

}  // namespace
}  // namespace gab

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
#ifdef GAB_ABLATE
// diagnostic builds: the ablation mask of the split kernels comes from the environment
static int gab_split_debug_mask() {
    static const int dbg = getenv("GAB_CONV_SPLIT_DEBUG") ? atoi(getenv("GAB_CONV_SPLIT_DEBUG")) : 0;
    return dbg;
}
#define GAB_SPLIT_DEBUG_ARG , gab_split_debug_mask()
#else
#define GAB_SPLIT_DEBUG_ARG
#endif

// The block the plan consumed last, as the kernels keep it (the newest slot of the history ring), back in the input's
// layout [tracks][512] — for checks of the round trip's upload hand-off: it must equal the h_in of the last call.
__global__ void conv_newest_block_kernel(const float* __restrict__ hist, float* __restrict__ out, int T, int slot) {
    const int q = blockIdx.x;
    const gab::fft::cf* const hp = reinterpret_cast<const gab::fft::cf*>(hist) + ((size_t)q * gab::kSlots + slot) * gab::kB;
    for (int i = threadIdx.x; i < gab::kB; i += blockDim.x) {
        const gab::fft::cf v = hp[i];
        out[(size_t)(2 * q) * gab::kB + i] = v.x;
        if (2 * q + 1 < T) out[(size_t)(2 * q + 1) * gab::kB + i] = v.y;
    }
}

struct gab_conv_plan {
    int tracks = 0, bufsize = 0, ir_len = 0;
    int pairs = 0;
    int head = 0;            // ring slot the next block goes to
    bool fused = false;      // bufsize == 512 && ir_len <= 4096
    bool tail = false;       // ir_len > 512 (partition B present)
    bool ir_set = false;
    float4* pmA = nullptr;
    float4* pmB = nullptr;
    float* hist = nullptr;   // fused: [pairs][8 slots][512] complex (ch a, ch b); fallback: two linear [T][hlen]
    float* hist_alt = nullptr;
    float* ir_copy = nullptr;   // fallback path keeps the time-domain taps
    int hlen = 0;
    size_t spectra_bytes = 0, history_bytes = 0;
    const gab::fft::cf* tw = nullptr;
    // split roles (conv_split_kernel): its spectra and the carry ring.  A plan keeps ONE cut of the
    // taps from its first buffer to the next reset: every streaming entry (one buffer, a batch, pinned
    // host buffers) launches that cut.
    bool split = false;
    float4* pmA2 = nullptr;
    float4* pmF = nullptr;
    gab::fft::cf* carry = nullptr;
    size_t carry_bytes = 0;
    bool fresh = true;        // nothing has run since the last reset
    // gab_conv_round_trip (classic cut): created at its first call
    unsigned* rt_stage = nullptr;       // fine-grained device memory the upload lands in: TWO buffers [2][T*B], taken in turn (call parity) — the
                                        // check launch behind call k re-arms buffer k & 1 while call k + 1 is already filling the other
    float* rt_park = nullptr;           // device: outputs until their channel group is complete
    unsigned* rt_counters = nullptr;    // device: per-group arrivals, groups drained
    unsigned* rt_words = nullptr;       // pinned host: [0] done, [16] landed, [32] error, [48] the check launch's verdict (a 64-byte line each)
    hipEvent_t rt_check_ev[2] = {nullptr, nullptr};   // behind conv_round_trip_check_kernel, per staging buffer (verdicts: rt_words[48], rt_words[56])
    bool rt_check_pending[2] = {false, false};        // a check launch has been queued and its verdict not yet read
    int rt_check_mode = 1;              // gab_conv_round_trip_set_check: 0 the verdict is ignored (the check launch still puts the sentinel back), 1 read at the next call, 2 read in the call
    hipStream_t rt_copy_stream = nullptr;
    hipEvent_t rt_copy_ev = nullptr;
    hipEvent_t rt_done_ev = nullptr;      // the launch's own completion (what gab_conv_round_trip returns on)
    gab_keep_warm* warm = nullptr;        // gab_conv_round_trip_keep_warm: kicked at the end of every round trip while warm_on
    bool warm_on = false;
    unsigned rt_epoch = 0;
    int rt_groups = 0;
    unsigned rt_bound[gab::kRtMaxGroups + 1] = {};
    const void* rt_checked_out = nullptr;
    // gab_conv_engine_* (split cut): one resident launch fed through a doorbell
    float* eng_in = nullptr;            // fine-grained device memory: [ring][T*B], the producer writes it while the launch runs
    float* eng_out = nullptr;           // [ring][B*T]
    unsigned* eng_done = nullptr;       // device: buffers finished per inverse wave [2 x workgroups]
    unsigned* eng_words = nullptr;      // pinned host: [0] doorbell, [16] completed, [32] error, [48] first workgroup runs, [49] all of them do
    int eng_ring = 0;
    double eng_idle_seconds = gab::kEngineIdleSeconds;   // gab_conv_engine_set_idle_limit
    hipStream_t eng_copy_stream = nullptr;   // gab_conv_engine_round_trip: the two link legs (engine copies)
    hipEvent_t eng_copy_ev = nullptr;
    bool eng_running = false;
    unsigned eng_published = 0;
    unsigned eng_seen_completed = 0;
    hipStream_t eng_stream = nullptr;    // the stream the resident launch is on: the plan's own (below)
    hipStream_t eng_own_stream = nullptr; // created at the first start, at the HIGHEST priority: the runtime maps streams onto a few hardware
                                          // queues per priority, and anything that shares a queue with a resident launch stands behind it until
                                          // the stop (round 5: a copy on the default stream did, and the engine starved for its own doorbell)
    hipEvent_t eng_ev = nullptr;
    // uniform partitions (conv_uniform_kernel): other power-of-two buffer sizes / longer responses
    bool uniform = false;
    int uJ = 0, uS = 0, ring_len = 0;
    unsigned upos = 0;
    float4* pmU = nullptr;
    gab::fft::cf* ring = nullptr;
    // Ordering between gab_conv_reset (memsets on the caller's stream) and launches on OTHER streams
    // (channel ranges): the reset waits for every stream that launched since the previous reset, and
    // a stream's first launch after a reset waits for the reset's event.
    int device = 0;
    hipEvent_t reset_ev = nullptr;
    hipStream_t reset_stream = nullptr;
    bool reset_recorded = false;
    std::vector<hipStream_t> used_streams;      // launched on since the last reset
    std::vector<hipStream_t> ordered_streams;   // already wait for reset_ev
    std::mutex order_mu;

    // called before every launch on `s`
    void order_after_reset(hipStream_t s) {
        std::lock_guard<std::mutex> lock(order_mu);
        if (std::find(used_streams.begin(), used_streams.end(), s) == used_streams.end()) used_streams.push_back(s);
        if (!reset_recorded || s == reset_stream) return;
        if (std::find(ordered_streams.begin(), ordered_streams.end(), s) != ordered_streams.end()) return;
        GAB_HIP_CHECK(hipStreamWaitEvent(s, reset_ev, 0));
        ordered_streams.push_back(s);
    }
};

extern "C" {

int gab_conv_create(gab_conv_plan** out, int tracks, int bufsize, int ir_len) {
    return gab::guarded([&]() -> int {
        if (!out) return gab::bad_arg("gab_conv_create: null plan pointer");
        if (int rc = gab::refuse_unsupported_runtime_mode("gab_conv_create")) return rc;
        if (tracks <= 0 || bufsize <= 0 || ir_len <= 0)
            return gab::bad_arg("gab_conv_create: tracks, bufsize and ir_len must be > 0");
        auto* p = new gab_conv_plan;
        p->tracks = tracks; p->bufsize = bufsize; p->ir_len = ir_len;
        p->pairs = (tracks + 1) / 2;
        GAB_HIP_CHECK(hipGetDevice(&p->device));
        GAB_HIP_CHECK(hipEventCreateWithFlags(&p->reset_ev, hipEventDisableTiming));
        p->fused = (bufsize == gab::kB && ir_len <= gab::kNB);
        p->tail = ir_len > gab::kB;
        const bool pow2 = (bufsize & (bufsize - 1)) == 0;
        p->uniform = !p->fused && pow2 && bufsize >= 32 && bufsize <= 2048 && ir_len <= 16384;
        try {
            if (p->uniform) {
                p->tw = gab::fft::device_twiddles();
                p->uS = gab::kNB - bufsize;
                p->uJ = 1 + (ir_len > bufsize ? (ir_len - bufsize + p->uS - 1) / p->uS : 0);
                // oldest sample any window reaches, plus the slot the new block is written to
                int need = gab::kNB + bufsize + (p->uJ > 1 ? bufsize + (p->uJ - 2) * p->uS : 0);
                p->ring_len = gab::kNB;
                while (p->ring_len < need) p->ring_len *= 2;
                p->spectra_bytes = sizeof(float4) * (size_t)p->uJ * p->pairs * gab::kBinsB;
                p->history_bytes = sizeof(gab::fft::cf) * (size_t)p->pairs * p->ring_len;
                GAB_HIP_CHECK(hipMalloc(&p->pmU, p->spectra_bytes));
                GAB_HIP_CHECK(hipMalloc(&p->ring, p->history_bytes));
                GAB_HIP_CHECK(hipMemset(p->ring, 0, p->history_bytes));
            } else if (p->fused) {
                p->tw = gab::fft::device_twiddles();
                size_t a = sizeof(float4) * (size_t)p->pairs * gab::kBinsA;
                size_t b = p->tail ? sizeof(float4) * (size_t)p->pairs * gab::kBinsB : 0;
                GAB_HIP_CHECK(hipMalloc(&p->pmA, a));
                if (b) GAB_HIP_CHECK(hipMalloc(&p->pmB, b));
                p->spectra_bytes = a + b;
                p->history_bytes = sizeof(float) * (size_t)p->pairs * 2 * gab::kSlots * gab::kB;
                GAB_HIP_CHECK(hipMalloc(&p->hist, p->history_bytes));
                GAB_HIP_CHECK(hipMemset(p->hist, 0, p->history_bytes));
                const bool can_split = ir_len > 2 * gab::kB && (tracks % 4) == 0;
                p->split = can_split;                  // gab_conv_set_scheme picks the classic cut
                if (can_split) {
                    GAB_HIP_CHECK(hipMalloc(&p->pmA2, a));
                    GAB_HIP_CHECK(hipMalloc(&p->pmF, b));
                    p->carry_bytes = sizeof(gab::fft::cf) * (size_t)p->pairs * gab::kCarrySlots * gab::kB;
                    GAB_HIP_CHECK(hipMalloc(&p->carry, p->carry_bytes));
                    GAB_HIP_CHECK(hipMemset(p->carry, 0, p->carry_bytes));
                }
            } else {
                int blocks = (ir_len - 1 + bufsize - 1) / bufsize;
                p->hlen = (blocks < 1 ? 1 : blocks) * bufsize;
                p->history_bytes = sizeof(float) * (size_t)tracks * p->hlen;
                p->spectra_bytes = sizeof(float) * (size_t)tracks * ir_len;
                GAB_HIP_CHECK(hipMalloc(&p->hist, p->history_bytes));
                GAB_HIP_CHECK(hipMalloc(&p->hist_alt, p->history_bytes));
                GAB_HIP_CHECK(hipMalloc(&p->ir_copy, p->spectra_bytes));
                GAB_HIP_CHECK(hipMemset(p->hist, 0, p->history_bytes));
            }
        } catch (...) {
            gab_conv_destroy(p);
            throw;
        }
        gab::resident_add(p, p->device, gab::kResidentEngine,
                          [](const void* o) { return static_cast<const gab_conv_plan*>(o)->eng_running; });
        *out = p;
        return GAB_OK;
    });
}

int gab_conv_destroy(gab_conv_plan* p) {
    if (!p) return GAB_OK;
    gab::resident_remove(p);
    if (p->eng_running) {                                            // never leave a resident launch behind: ring the stop rung FIRST
        if (p->eng_words) __atomic_store_n(&p->eng_words[0], p->eng_published | 0x80000000u, __ATOMIC_RELEASE);
        (void)hipStreamSynchronize(p->eng_stream);                   // (a null handle is the default stream)
        p->eng_running = false;
    }
    if (p->warm) { (void)gab_keep_warm_destroy(p->warm); p->warm = nullptr; }   // (a resident launch: before the device-wide wait)
    (void)hipDeviceSynchronize();
    if (p->pmA) (void)hipFree(p->pmA);
    if (p->pmB) (void)hipFree(p->pmB);
    if (p->hist) (void)hipFree(p->hist);
    if (p->hist_alt) (void)hipFree(p->hist_alt);
    if (p->ir_copy) (void)hipFree(p->ir_copy);
    if (p->pmA2) (void)hipFree(p->pmA2);
    if (p->pmF) (void)hipFree(p->pmF);
    if (p->carry) (void)hipFree(p->carry);
    if (p->pmU) (void)hipFree(p->pmU);
    if (p->ring) (void)hipFree(p->ring);
    if (p->reset_ev) (void)hipEventDestroy(p->reset_ev);
    if (p->rt_stage) (void)hipFree(p->rt_stage);
    if (p->rt_park) (void)hipFree(p->rt_park);
    if (p->rt_counters) (void)hipFree(p->rt_counters);
    for (hipEvent_t e : p->rt_check_ev) if (e) (void)hipEventDestroy(e);
    if (p->rt_words) (void)hipHostFree(p->rt_words);
    if (p->eng_in) (void)hipFree(p->eng_in);
    if (p->eng_out) (void)hipFree(p->eng_out);
    if (p->eng_done) (void)hipFree(p->eng_done);
    if (p->eng_words) (void)hipHostFree(p->eng_words);
    if (p->rt_copy_ev) (void)hipEventDestroy(p->rt_copy_ev);
    if (p->eng_ev) (void)hipEventDestroy(p->eng_ev);
    if (p->eng_own_stream) (void)hipStreamDestroy(p->eng_own_stream);
    if (p->eng_copy_ev) (void)hipEventDestroy(p->eng_copy_ev);
    if (p->eng_copy_stream) (void)hipStreamDestroy(p->eng_copy_stream);
    if (p->rt_done_ev) (void)hipEventDestroy(p->rt_done_ev);
    if (p->rt_copy_stream) (void)hipStreamDestroy(p->rt_copy_stream);
    delete p;
    return GAB_OK;
}

int gab_conv_set_ir(gab_conv_plan* p, const float* d_ir, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!p || !d_ir) return gab::bad_arg("gab_conv_set_ir: null argument");
        if (p->eng_running) return gab::bad_arg("gab_conv_set_ir: the plan's engine is running (gab_conv_engine_stop first)");
        hipStream_t s = gab::as_stream(stream);
        if (p->uniform) {
            for (int j = 0; j < p->uJ; ++j)
                gab::conv_ir_spectra_uniform_kernel<<<p->pairs, gab::kThreads, 0, s>>>(
                    d_ir, p->pmU + (size_t)j * p->pairs * gab::kBinsB, p->tw, p->tracks, p->ir_len,
                    j == 0 ? 0 : p->bufsize + (j - 1) * p->uS, j == 0 ? p->bufsize : p->uS);
            int rc = gab::launch_status("conv_ir_spectra_uniform_kernel");
            if (rc) return rc;
        } else if (p->fused) {
            // pmB is null when ir_len <= 512: the kernel then skips partition B
            gab::conv_ir_spectra_kernel<<<p->pairs, gab::kThreads, 0, s>>>(
                d_ir, p->pmA, p->pmB, p->tw, p->tracks, p->ir_len, 0, gab::kB);
            if (p->pmF)
                gab::conv_ir_spectra_kernel<<<p->pairs, gab::kThreads, 0, s>>>(
                    d_ir, p->pmA2, p->pmF, p->tw, p->tracks, p->ir_len, gab::kB, 2 * gab::kB);
            int rc = gab::launch_status("conv_ir_spectra_kernel");
            if (rc) return rc;
        } else {
            GAB_HIP_CHECK(hipMemcpyAsync(p->ir_copy, d_ir, p->spectra_bytes,
                                         hipMemcpyDeviceToDevice, s));
        }
        GAB_HIP_CHECK(hipStreamSynchronize(s));
        p->ir_set = true;
        return GAB_OK;
    });
}

int gab_conv_set_scheme(gab_conv_plan* p, int scheme) {
    if (!p) return gab::bad_arg("gab_conv_set_scheme: null plan");
    if (scheme != GAB_CONV_SCHEME_CLASSIC && scheme != GAB_CONV_SCHEME_SPLIT)
        return gab::bad_arg("gab_conv_set_scheme: unknown scheme");
    if (!p->fresh) return gab::bad_arg("gab_conv_set_scheme: only on a fresh plan (before the first buffer or right after a reset)");
    if (scheme == GAB_CONV_SCHEME_SPLIT && !p->pmF)
        return gab::bad_arg("gab_conv_set_scheme: the split scheme needs 512-sample buffers, 1025..4096 taps and a channel count divisible by 4");
    p->split = scheme == GAB_CONV_SCHEME_SPLIT;
    return GAB_OK;
}

int gab_conv_get_scheme(const gab_conv_plan* p, int* scheme) {
    if (!p || !scheme) return gab::bad_arg("gab_conv_get_scheme: null pointer");
    *scheme = p->split ? GAB_CONV_SCHEME_SPLIT : GAB_CONV_SCHEME_CLASSIC;
    return GAB_OK;
}

int gab_conv_reset(gab_conv_plan* p, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!p) return gab::bad_arg("gab_conv_reset: null plan");
        if (p->eng_running) return gab::bad_arg("gab_conv_reset: the plan's engine is running (gab_conv_engine_stop first)");
        for (int b = 0; b < 2; ++b)                            // (a reset starts the stream anew: the last round trips' verdicts no longer matter)
            if (p->rt_check_pending[b]) {
                (void)hipEventSynchronize(p->rt_check_ev[b]);
                p->rt_check_pending[b] = false;
                p->rt_words[48 + 8 * b] = 0;
            }
        hipStream_t s = gab::as_stream(stream);
        {
            // launches still in flight on other streams read the rings: the memsets go behind them
            std::lock_guard<std::mutex> lock(p->order_mu);
            for (hipStream_t u : p->used_streams) {
                if (u == s) continue;
                GAB_HIP_CHECK(hipEventRecord(p->reset_ev, u));
                GAB_HIP_CHECK(hipStreamWaitEvent(s, p->reset_ev, 0));
            }
            p->used_streams.clear();
            p->ordered_streams.clear();
        }
        if (p->uniform) GAB_HIP_CHECK(hipMemsetAsync(p->ring, 0, p->history_bytes, s));
        else GAB_HIP_CHECK(hipMemsetAsync(p->hist, 0, p->history_bytes, s));
        p->head = 0;
        p->upos = 0;
        if (p->carry) GAB_HIP_CHECK(hipMemsetAsync(p->carry, 0, p->carry_bytes, s));
        {
            std::lock_guard<std::mutex> lock(p->order_mu);
            GAB_HIP_CHECK(hipEventRecord(p->reset_ev, s));
            p->reset_stream = s;
            p->reset_recorded = true;
        }
        p->fresh = true;
        return GAB_OK;
    });
}

int gab_conv_process(gab_conv_plan* p, const float* d_in, float* d_out, int mode,
                     gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!p || !d_in || !d_out) return gab::bad_arg("gab_conv_process: null argument");
        if (!p->ir_set) return gab::bad_arg("gab_conv_process: gab_conv_set_ir has not been called");
        if (p->eng_running) return gab::bad_arg("gab_conv_process: the plan's engine is running and owns its history (gab_conv_engine_stop first)");
        if (mode != GAB_CONV_STATELESS && mode != GAB_CONV_STREAMING && mode != GAB_CONV_STREAMING_HOST_IO)
            return gab::bad_arg("gab_conv_process: unknown mode");
        hipStream_t s = gab::as_stream(stream);
        const bool streaming = mode != GAB_CONV_STATELESS;
        if (streaming) p->order_after_reset(s);
        if (p->uniform) {
            if (mode == GAB_CONV_STREAMING_HOST_IO) mode = GAB_CONV_STREAMING;      // same kernel, host pointers
            gab::ConvUniform u{p->pmU, p->ring, p->uJ, p->uS, p->ring_len - 1, p->upos};
            if (streaming) {
                gab::conv_uniform_kernel<true><<<dim3(p->pairs), dim3(gab::kThreads), 0, s>>>(
                    d_in, d_out, u, p->tw, p->tracks, p->bufsize);
                p->upos = (p->upos + (unsigned)p->bufsize) & (unsigned)(p->ring_len - 1);
                p->fresh = false;
            } else {
                gab::conv_uniform_kernel<false><<<dim3(p->pairs), dim3(gab::kThreads), 0, s>>>(
                    d_in, d_out, u, p->tw, p->tracks, p->bufsize);
            }
            return gab::launch_status("conv_uniform_kernel");
        }
        if (p->fused) {
            dim3 grid(p->pairs), block(gab::kThreads);
#define GAB_CONV_ARGS d_in, d_out, p->hist, p->pmA, p->pmB, p->tw, p->tracks, p->head
            if (streaming) p->fresh = false;
            if (streaming && p->split) {
                // pinned host buffers: the same kernel under its own name (it runs at link speed: the new
                // block crosses the link once per role, 99 us per round trip against the classic cut's 90)
                gab::ConvSplit sp{p->pmA2, p->pmF, p->carry GAB_SPLIT_DEBUG_ARG};
                if (mode == GAB_CONV_STREAMING_HOST_IO)
                    gab::conv_split_host_io_kernel<<<grid, block, 0, s>>>(d_in, d_out, p->hist, p->pmA, sp, p->tw,
                                                                         p->tracks, p->head);
                else
                    gab::conv_split_kernel<<<grid, block, 0, s>>>(d_in, d_out, p->hist, p->pmA, sp, p->tw,
                                                                 p->tracks, p->head);
                int rc = gab::launch_status("conv_split_kernel");
                if (rc) return rc;
                p->head = (p->head + 1) & (gab::kSlots - 1);
                return GAB_OK;
            }
            if (mode == GAB_CONV_STREAMING_HOST_IO) {
                if (p->tail) gab::conv_overlap_save_host_io_kernel<true><<<grid, block, 0, s>>>(GAB_CONV_ARGS);
                else gab::conv_overlap_save_host_io_kernel<false><<<grid, block, 0, s>>>(GAB_CONV_ARGS);
            } else if (!streaming)
                gab::conv_overlap_save_kernel<false, false><<<grid, block, 0, s>>>(GAB_CONV_ARGS);
            else if (!p->tail)
                gab::conv_overlap_save_kernel<true, false><<<grid, block, 0, s>>>(GAB_CONV_ARGS);
            else
                gab::conv_overlap_save_kernel<true, true><<<grid, block, 0, s>>>(GAB_CONV_ARGS);
#undef GAB_CONV_ARGS
            int rc = gab::launch_status("conv_overlap_save_kernel");
            if (rc) return rc;
            if (streaming) p->head = (p->head + 1) & (gab::kSlots - 1);
        } else {
            dim3 block(256), grid((p->bufsize + 255) / 256, p->tracks);
            if (streaming) {
                gab::conv_direct_kernel<true><<<grid, block, 0, s>>>(
                    d_in, d_out, p->hist, p->ir_copy, p->tracks, p->bufsize, p->ir_len, p->hlen);
                dim3 g2((p->hlen + 255) / 256, p->tracks);
                gab::conv_direct_shift_kernel<<<g2, block, 0, s>>>(
                    d_in, p->hist, p->hist_alt, p->tracks, p->bufsize, p->hlen);
                std::swap(p->hist, p->hist_alt);
            } else {
                gab::conv_direct_kernel<false><<<grid, block, 0, s>>>(
                    d_in, d_out, p->hist, p->ir_copy, p->tracks, p->bufsize, p->ir_len, p->hlen);
            }
            int rc = gab::launch_status("conv_direct_kernel");
            if (rc) return rc;
        }
        return GAB_OK;
    });
}

// The staging buffers, counters and the upload stream of gab_conv_round_trip.
static void gab_conv_round_trip_init(gab_conv_plan* p) {
    const size_t n = (size_t)p->tracks * p->bufsize;
    GAB_HIP_CHECK(hipExtMallocWithFlags(reinterpret_cast<void**>(&p->rt_stage), 2 * n * 4, hipDeviceMallocFinegrained));
    GAB_HIP_CHECK(hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(p->rt_stage), (int)gab::kRtSentinel, 2 * n));
    GAB_HIP_CHECK(hipMalloc(&p->rt_park, n * 4));
    int groups = 16;                                  // 64 channels = 256-byte rows at 1024 channels
    int taper = 1;                                    // the first and the last group cut into quarter, quarter, half (round 5)
#ifdef GAB_ABLATE
    if (getenv("GAB_RT_GROUPS")) groups = std::max(1, atoi(getenv("GAB_RT_GROUPS")));
    if (getenv("GAB_RT_TAPER")) taper = atoi(getenv("GAB_RT_TAPER"));      // 0 equal groups, 1 both ends, 2 the head only, 3 the tail only
#endif
    int ppg = (p->pairs + groups - 1) / groups;
    ppg += ppg & 1;                                   // whole float4 columns per row
    // Groups are drained in the order the upload lands them.  The link's downward direction cannot start before the FIRST
    // group is through, and after the LAST rows have landed nothing hides that group's transform and drain: both are cut
    // finer (their rows are narrower — 64-byte pieces for 16 channels — but few).  Boundaries in pairs, every group even.
    std::vector<int> sizes;
    for (int at = 0; at < p->pairs; at += ppg) sizes.push_back(std::min(ppg, p->pairs - at));
    auto cut = [&](int n, bool rising) {             // n pairs -> n/4, n/4, n/2 (rising) or n/2, n/4, n/4
        std::vector<int> v;
        if (n % 8 != 0 || n < 16) return std::vector<int>{n};
        if (rising) v = {n / 4, n / 4, n / 2}; else v = {n / 2, n / 4, n / 4};
        return v;
    };
    std::vector<int> fine;
    for (size_t i = 0; i < sizes.size(); ++i) {
        const bool head = i == 0 && (taper == 1 || taper == 2) && sizes.size() > 2;
        const bool tail = i + 1 == sizes.size() && (taper == 1 || taper == 3) && sizes.size() > 2;
        std::vector<int> v = head ? cut(sizes[i], true) : tail ? cut(sizes[i], false) : std::vector<int>{sizes[i]};
        fine.insert(fine.end(), v.begin(), v.end());
    }
    if ((int)fine.size() > gab::kRtMaxGroups) throw std::runtime_error("gab_conv_round_trip: too many channel groups");
    p->rt_groups = (int)fine.size();
    p->rt_bound[0] = 0;
    for (int g = 0; g < p->rt_groups; ++g) p->rt_bound[g + 1] = p->rt_bound[g] + (unsigned)fine[g];
    GAB_HIP_CHECK(hipMalloc(&p->rt_counters, sizeof(unsigned) * 32 * (p->rt_groups + 1)));      // a 128-byte line per group, one for the shares drained
    GAB_HIP_CHECK(hipMemset(p->rt_counters, 0, sizeof(unsigned) * 32 * (p->rt_groups + 1)));
    for (hipEvent_t& e : p->rt_check_ev) GAB_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    GAB_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&p->rt_words), 64 * sizeof(unsigned), hipHostMallocDefault));
    for (int i = 0; i < 64; ++i) p->rt_words[i] = 0;
    {
        // The upload's stream at the HIGHEST priority: the runtime maps streams onto a few hardware queues per priority, and where
        // the upload's stream shares a queue with the caller's, the main launch stands behind the copy's completion packet — the
        // call then takes upload + kernel, 123-143 us instead of 70 (measured after an engine's streams had come and gone in the
        // process: profiles/r06_roundtrip_check.txt).  Streams of the default priority never share a queue with it.
        int lo = 0, hi = 0;
        GAB_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        GAB_HIP_CHECK(hipStreamCreateWithPriority(&p->rt_copy_stream, hipStreamNonBlocking, hi));
    }
    GAB_HIP_CHECK(hipEventCreateWithFlags(&p->rt_copy_ev, hipEventDisableTiming));
    GAB_HIP_CHECK(hipEventCreateWithFlags(&p->rt_done_ev, hipEventDisableTiming));
    GAB_HIP_CHECK(hipDeviceSynchronize());
    p->rt_epoch = 0;
}

// The verdict of the last check launch (conv_round_trip_check_kernel): waits for it (it runs a few microseconds behind the call
// that queued it), GAB_OK or GAB_ERR_RUNTIME with the message.  The plan's next upload must come behind it either way: the
// check launch is what puts the sentinel back.
static int gab_conv_round_trip_finish_check(gab_conv_plan* p, int b, bool wait, const char* who) {
    if (!p->rt_check_pending[b]) return GAB_OK;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;;) {
        const hipError_t q = hipEventQuery(p->rt_check_ev[b]);
        if (q == hipSuccess) break;
        (void)hipGetLastError();
        if (q != hipErrorNotReady) GAB_HIP_CHECK(q);
        if (!wait) return GAB_OK;                                  // (still running: its verdict is read later)
        if ((++spins & 1023u) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 4.0) {
            gab::set_last_error(std::string(who) + ": the check launch behind an earlier round trip did not end within 4 s");
            return GAB_ERR_RUNTIME;
        }
    }
    p->rt_check_pending[b] = false;
    const unsigned verdict = __atomic_load_n(&p->rt_words[48 + 8 * b], __ATOMIC_ACQUIRE);
    p->rt_words[48 + 8 * b] = 0;
    if (p->rt_check_mode != 0 && (verdict & gab::kRtErrTorn)) {
        gab::set_last_error(std::string(who) + ": a word the round trip's kernel consumed while the upload was still running is not the word the "
                            "completed upload left in the staging buffer (an engine write that landed in pieces or out of order): the output of THAT "
                            "round trip was wrong and so is the plan's carried history (gab_conv_reset before the stream goes on)");
        return GAB_ERR_RUNTIME;
    }
    return GAB_OK;
}

int gab_conv_round_trip_check(gab_conv_plan* p) {
    return gab::guarded([&]() -> int {
        if (!p) return gab::bad_arg("gab_conv_round_trip_check: null plan");
        const int r0 = gab_conv_round_trip_finish_check(p, 0, true, "gab_conv_round_trip_check");
        const int r1 = gab_conv_round_trip_finish_check(p, 1, true, "gab_conv_round_trip_check");
        return r0 ? r0 : r1;
    });
}

int gab_conv_round_trip_set_check(gab_conv_plan* p, int mode) {
    if (!p) return gab::bad_arg("gab_conv_round_trip_set_check: null plan");
    if (mode < 0 || mode > 2) return gab::bad_arg("gab_conv_round_trip_set_check: 0 (ignore the verdict), 1 (read it at the next call / gab_conv_round_trip_check), 2 (read it in the call)");
    p->rt_check_mode = mode;
    return GAB_OK;
}

int gab_conv_round_trip(gab_conv_plan* p, const float* h_in, float* h_out, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!p || !h_in || !h_out) return gab::bad_arg("gab_conv_round_trip: null argument");
        if (!p->ir_set) return gab::bad_arg("gab_conv_round_trip: gab_conv_set_ir has not been called");
        if (p->eng_running) return gab::bad_arg("gab_conv_round_trip: the plan's engine is running and owns its history (gab_conv_engine_stop first)");
        hipStream_t s = gab::as_stream(stream);
        auto mapped = [](const void* ptr) {               // may a kernel touch it?  (pinned host or device memory)
            hipPointerAttribute_t at;
            if (hipPointerGetAttributes(&at, ptr) != hipSuccess || at.devicePointer == nullptr) {
                (void)hipGetLastError();
                return false;
            }
            return true;
        };
        if (!(p->fused && p->tail && !p->split && (p->tracks % 4) == 0)) {
            // every other plan: the kernel moves the buffers over the link itself — both must be mapped into the device
            // (a kernel that dereferences pageable host memory faults)
            if (!mapped(h_in) || !mapped(h_out))
                return gab::bad_arg("gab_conv_round_trip: on this plan (not the classic cut at 512-sample buffers) the kernel reads h_in and "
                                    "writes h_out itself: both must be pinned host memory (hipHostMalloc) or device memory");
            int rc = gab_conv_process(p, h_in, h_out, GAB_CONV_STREAMING_HOST_IO, stream);
            if (rc) return rc;
            GAB_HIP_CHECK(hipStreamSynchronize(s));
            return GAB_OK;
        }
        if (!p->rt_stage) gab_conv_round_trip_init(p);
        // The staging buffers take turns: this call fills buffer (epoch + 1) & 1, whose check launch (queued two calls ago: it
        // puts the sentinel back) must be through — waited for; the OTHER buffer's check (the previous call's) is looked at
        // without waiting: a paced caller finds its verdict here, a back-to-back caller one call later.
        {
            const int mine = (int)((p->rt_epoch + 1) & 1u);
            const int ra = gab_conv_round_trip_finish_check(p, mine, true, "gab_conv_round_trip (an earlier call)");
            const int rb = gab_conv_round_trip_finish_check(p, mine ^ 1, false, "gab_conv_round_trip (the previous call)");
            if (ra || rb) return ra ? ra : rb;
        }
        if (p->rt_checked_out != h_out) {              // the kernel writes h_out itself: it must be mapped into the device
            if (!mapped(h_out))
                return gab::bad_arg("gab_conv_round_trip: h_out must be pinned host memory (hipHostMalloc) or device memory");
            p->rt_checked_out = h_out;
        }
        p->order_after_reset(s);
        const size_t bytes = sizeof(float) * (size_t)p->tracks * p->bufsize;
        // the device counters run on from launch to launch and are compared with epoch x members: the plan's epoch moves
        // only when a launch has really been made
        const unsigned epoch = p->rt_epoch + 1;
        const int buf = (int)(epoch & 1u);                                           // this call's staging buffer
        unsigned* const stage = p->rt_stage + (size_t)buf * p->tracks * p->bufsize;
        volatile unsigned* const done = p->rt_words;
        unsigned* const landed = p->rt_words + 16;
        volatile unsigned* const error = p->rt_words + 32;
        bool upload = true;
        // How the launch's end is observed: kRtCompletion (measured, profiles/r05_roundtrip_completion.txt).
        //   0 hipStreamSynchronize   1 an event recorded behind the launch, queried   2 the launch's own stop event
        //   (hipExtLaunchKernelGGL), queried   3 hipStreamQuery   9 (diagnostic builds only) round 4's rule: the hint word
        int completion = gab::kRtCompletion;
#ifdef GAB_ABLATE
        if (getenv("GAB_RT_SKIP_UPLOAD")) upload = false;   // diagnostic builds: the input never lands — every wait must run out
        if (getenv("GAB_RT_COMPLETION")) completion = atoi(getenv("GAB_RT_COMPLETION"));   // diagnostic builds: to price the rules
#endif
        // The kernel takes a word the moment it is no longer the sentinel and puts the sentinel back: that is only right
        // if the upload writes every word exactly ONCE — one engine copy from pinned (or device) memory does.  What the
        // runtime does with PAGEABLE memory (staging pieces, heads and tails on their own) is its own business: such an
        // input is uploaded completely first and announced as landed before the launch.  (profiles/r04_incident_*)
        // h_in is read by the copy from the moment of this call: it must be complete on the host (or, for device memory,
        // on the device) by then — the upload runs on the plan's own stream and is NOT ordered behind work queued on `stream`.
        bool streamed = true;
        // put the stage back to all-sentinel and the upload stream to rest (after anything that may have left words behind)
        auto rearm_stage = [&]() {
            (void)hipStreamSynchronize(s);
            (void)hipStreamSynchronize(p->rt_copy_stream);
            (void)hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(p->rt_stage), (int)gab::kRtSentinel, 2 * (size_t)p->tracks * p->bufsize);
            p->rt_words[48] = p->rt_words[56] = 0;      // (check launches over words that never landed say nothing)
            p->rt_check_pending[0] = p->rt_check_pending[1] = false;
            (void)hipDeviceSynchronize();
        };
        // The engine copy goes out in pieces of kRtUploadPiece bytes, a multiple of four: the runtime cuts a copy into engine
        // packets of at most 4 MiB - 1 BYTES, so in one big copy the word that straddles a packet boundary is written in two
        // pieces — and a consumer that takes a word the moment it differs from the sentinel took three landed bytes with the
        // sentinel's fourth (round 5, caught by the self-classifying stress at 2052 channels: word 1 048 575, bytes 4 194 300 -
        // 4 194 303, consumed fff4b08e for bef4b08e; the r04 incident at 8192 channels has three such words:
        // profiles/r05_incident_torn_word.txt).  Within a packet the engine writes whole aligned bursts.
        long tear = -1;                                 // diagnostic builds: a word index whose early value is NOT the upload's (below)
#ifdef GAB_ABLATE
        if (getenv("GAB_RT_TEAR") && upload && mapped(h_in)) tear = atol(getenv("GAB_RT_TEAR"));
        if (tear >= (long)(bytes / 4)) tear = -1;
#endif
        auto upload_range = [&](size_t lo, size_t hi) {
            for (size_t off = lo; off < hi; off += gab::kRtUploadPiece)
                GAB_HIP_CHECK(hipMemcpyAsync(reinterpret_cast<char*>(stage) + off, reinterpret_cast<const char*>(h_in) + off,
                                             std::min(gab::kRtUploadPiece, hi - off), hipMemcpyHostToDevice, p->rt_copy_stream));
        };
        auto upload_pieces = [&]() {
            if (tear < 0) { upload_range(0, bytes); return; }
            // Diagnostic builds, GAB_RT_TEAR=<word>: what a torn or reordered engine write would look like to the kernel.  The word is
            // in the staging buffer with ONE BIT WRONG before the launch (the kernel takes it the moment its neighbours land), the
            // upload leaves it out, and the right value goes up only when the rest is through, just before `landed` is released:
            // the consumed word differs from what the completed upload left, and the call must say so.
            unsigned wrong = reinterpret_cast<const unsigned*>(h_in)[tear] ^ 0x00010000u;
            GAB_HIP_CHECK(hipMemcpy(stage + tear, &wrong, 4, hipMemcpyHostToDevice));
            upload_range(0, (size_t)tear * 4);
            upload_range((size_t)tear * 4 + 4, bytes);
        };
        if (upload && !mapped(h_in)) {
            upload_pieces();
            GAB_HIP_CHECK(hipStreamSynchronize(p->rt_copy_stream));
            __atomic_store_n(landed, epoch, __ATOMIC_RELEASE);
            streamed = false;
        } else if (upload) {
            upload_pieces();
        }
        gab::ConvRoundTrip rt{stage, p->rt_park, h_out, p->rt_counters, p->rt_words, p->rt_words + 16, p->rt_words + 32,
                              epoch, p->rt_groups, {}};
        for (int g = 0; g <= p->rt_groups; ++g) rt.bound[g] = p->rt_bound[g];
        if (completion == 2)
            hipExtLaunchKernelGGL(gab::conv_round_trip_kernel, dim3(p->pairs), dim3(gab::kThreads), 0, s, nullptr, p->rt_done_ev, 0,
                                  rt, p->hist, (const float4*)p->pmA, (const float4*)p->pmB, (const gab::fft::cf*)p->tw, p->tracks, p->head);
        else
            gab::conv_round_trip_kernel<<<dim3(p->pairs), dim3(gab::kThreads), 0, s>>>(rt, p->hist, p->pmA, p->pmB, p->tw, p->tracks, p->head);
        int rc = gab::launch_status("conv_round_trip_kernel");
        if (!rc && completion == 1) GAB_HIP_CHECK(hipEventRecord(p->rt_done_ev, s));
        if (rc) {                                       // nothing ran: the epoch, the history and the counters stay as they were
            if (upload) rearm_stage();                  // (the upload did: its words must not pass for the next call's)
            return rc;
        }
        p->rt_epoch = epoch;
        if (upload && streamed) GAB_HIP_CHECK(hipEventRecord(p->rt_copy_ev, p->rt_copy_stream));
        // The check launch: queued on the caller's stream (behind the main launch) by the HOST, once the host has seen the upload's
        // completion event — no wait for that event is ever put into a stream: the runtime maps streams onto a few hardware
        // queues, and a barrier that waits 15-20 us for the event holds up whatever shares its queue (measured: the next call's
        // main launch, 98 against 71 us per call in a process with many streams: profiles/r06_roundtrip_check.txt).  It compares
        // the words the main launch consumed (the history ring's newest block) with what the completed upload left and puts
        // the sentinel back.
        const int consumed_slot = p->head;
        auto queue_check = [&]() {
            gab::conv_round_trip_check_kernel<<<dim3(p->pairs), dim3(gab::kThreads), 0, s>>>(stage, p->hist, p->rt_words + 48 + 8 * buf, consumed_slot);
            if (gab::launch_status("conv_round_trip_check_kernel")) throw std::runtime_error(gab::last_error());
            GAB_HIP_CHECK(hipEventRecord(p->rt_check_ev[buf], s));
            p->rt_check_pending[buf] = true;
        };
        if (upload && !streamed) queue_check();         // (an upload that was complete before the launch)
        p->head = (p->head + 1) & (gab::kSlots - 1);
        p->fresh = false;
        // After a wait that ran out, words may land behind the sentinel the kernel put back and the kernel has taken
        // sentinels for samples: the launch bounds its own waits, so let it end (also before the caller may free the
        // buffers), put the stage back to all-sentinel, and say that the carried history now holds garbage.
        auto after_a_failed_wait = [&](const char* what) {
            rearm_stage();
            *error = 0;
            gab::set_last_error(std::string("gab_conv_round_trip: ") + what +
                                "; the output of this call is invalid and so is the plan's carried history (gab_conv_reset before the stream goes on)");
            return GAB_ERR_RUNTIME;
        };
        // 1. The upload: its completion event releases workgroups whose rows really hold the sentinel (`landed`, a release
        //    store the kernel acquires).  The copy is through well before the kernel (its last group is still to be
        //    transformed and drained), so this costs the call nothing.
        // 2. The hint word, so that the stream is asked once, when the launch is about to end (its waits are bounded: it
        //    ends by itself, with or without the hint).
        // 3. The launch's end (its stop event): the stated point from which h_out is the host's and the staging buffer the next upload's.
        bool told = !upload || !streamed;               // (diagnostic: nothing was uploaded, nothing is announced)
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        bool ended = false;
        while (!told || *done != epoch) {
            if (!told && hipEventQuery(p->rt_copy_ev) == hipSuccess) {
                if (tear >= 0) {                            // (diagnostic builds) the word's right value, late
                    GAB_HIP_CHECK(hipMemcpyAsync(stage + tear, reinterpret_cast<const unsigned*>(h_in) + tear, 4, hipMemcpyHostToDevice, p->rt_copy_stream));
                    GAB_HIP_CHECK(hipStreamSynchronize(p->rt_copy_stream));
                }
                __atomic_store_n(landed, epoch, __ATOMIC_RELEASE);
                told = true;
                queue_check();                              // the upload is complete and the host knows it: the check goes behind the main launch
            }
            if ((++spins & 1023u) == 0) {
                if (told && hipStreamQuery(s) == hipSuccess) { ended = true; break; }      // over without the hint: a wait ran out
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 4.0)
                    return after_a_failed_wait("the launch did not end within 4 s");
            }
        }
        (void)hipGetLastError();                        // (hipStreamQuery's hipErrorNotReady)
        if (ended || completion == 0) {
            GAB_HIP_CHECK(hipStreamSynchronize(s));
        } else if (completion != 9) {
            for (spins = 0;;) {
                const hipError_t q = completion == 3 ? hipStreamQuery(s) : hipEventQuery(p->rt_done_ev);
                if (q == hipSuccess) break;
                (void)hipGetLastError();
                if (q != hipErrorNotReady) GAB_HIP_CHECK(q);
                if ((++spins & 1023u) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 4.0)
                    return after_a_failed_wait("the launch did not end within 4 s");
            }
        }
        if (*error != 0) return after_a_failed_wait("a workgroup waited about a second for its input and gave up");
        if (*done != epoch) return after_a_failed_wait("the launch ended without draining every channel group");
        if (p->rt_check_mode == 2)                               // the check's verdict AT this call (it ends when the upload's event has gone through the command processor)
            if (int rc2 = gab_conv_round_trip_finish_check(p, buf, true, "gab_conv_round_trip")) return rc2;
        if (p->warm_on) (void)gab_keep_warm_kick(p->warm);       // the result is out: keep the device from going idle until the next slot
        return GAB_OK;
    });
}

int gab_conv_round_trip_keep_warm(gab_conv_plan* p, int on) {
    return gab::guarded([&]() -> int {
        if (!p) return gab::bad_arg("gab_conv_round_trip_keep_warm: null plan");
        if (on && !p->warm) {
            // the launch must outlive the gap between two calls: several buffer periods even at a low sampling rate
            if (int rc = gab_keep_warm_create(&p->warm, 8, std::max(0.05, 8.0 * p->bufsize / 44100.0))) return rc;
        }
        p->warm_on = on != 0;
        return GAB_OK;
    });
}

int gab_conv_round_trip_keep_warm_placement(gab_conv_plan* p, unsigned* hw_id, unsigned* xcc_id, int capacity, int* started) {
    if (!p || !started) return gab::bad_arg("gab_conv_round_trip_keep_warm_placement: null argument");
    if (!p->warm) { *started = 0; return GAB_OK; }               // no keep-warm launch has been made for this plan
    return gab_keep_warm_placement(p->warm, hw_id, xcc_id, capacity, started);
}

int gab_conv_newest_block(gab_conv_plan* p, float* d_out, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!p || !d_out) return gab::bad_arg("gab_conv_newest_block: null argument");
        if (!p->fused || p->fresh) return gab::bad_arg("gab_conv_newest_block: a 512-sample plan that has taken at least one buffer");
        if (p->eng_running) return gab::bad_arg("gab_conv_newest_block: the plan's engine is running");
        conv_newest_block_kernel<<<dim3(p->pairs), dim3(256), 0, gab::as_stream(stream)>>>(
            p->hist, d_out, p->tracks, (p->head + gab::kSlots - 1) & (gab::kSlots - 1));
        return gab::launch_status("conv_newest_block_kernel");
    });
}

// ---- the doorbell-fed engine ------------------------------------------------------------------------------------------
int gab_conv_engine_rings(gab_conv_plan* p, int ring_buffers, float** d_in_ring, float** d_out_ring) {
    return gab::guarded([&]() -> int {
        if (!p || !d_in_ring || !d_out_ring) return gab::bad_arg("gab_conv_engine_rings: null argument");
        if (!(p->fused && p->split)) return gab::bad_arg("gab_conv_engine_rings: the engine runs the split cut (512-sample buffers, 1025..4096 taps, channels divisible by 4)");
        if (p->eng_running) return gab::bad_arg("gab_conv_engine_rings: the plan's engine is running");
        if (ring_buffers < 3 || ring_buffers > 4096) return gab::bad_arg("gab_conv_engine_rings: ring_buffers must be 3..4096");
        const size_t n = (size_t)p->tracks * p->bufsize;
        if (!p->eng_own_stream) {                     // the launch's own stream (see gab_conv_plan): made here, ahead of any timed start
            int lo = 0, hi = 0;                       // (numerically lower = higher priority)
            GAB_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
            GAB_HIP_CHECK(hipStreamCreateWithPriority(&p->eng_own_stream, hipStreamNonBlocking, hi));
            GAB_HIP_CHECK(hipEventCreateWithFlags(&p->eng_ev, hipEventDisableTiming));
        }
        if (p->eng_ring != ring_buffers) {
            if (p->eng_in) { (void)hipFree(p->eng_in); p->eng_in = nullptr; }
            if (p->eng_out) { (void)hipFree(p->eng_out); p->eng_out = nullptr; }
            p->eng_ring = 0;
            // ordinary device memory: the launch reads the input ring with system-scope loads and writes the output
            // ring with write-through stores, so what a copy engine writes is seen and what it reads is there
            unsigned flags = (GAB_ENGV & 64) ? hipDeviceMallocFinegrained : hipDeviceMallocDefault;   // (experiments: fine-grained, as before)
            GAB_HIP_CHECK(hipExtMallocWithFlags(reinterpret_cast<void**>(&p->eng_in), n * 4 * ring_buffers, flags));
            GAB_HIP_CHECK(hipExtMallocWithFlags(reinterpret_cast<void**>(&p->eng_out), n * 4 * ring_buffers, flags));
            p->eng_ring = ring_buffers;
        }
        *d_in_ring = p->eng_in;
        *d_out_ring = p->eng_out;
        return GAB_OK;
    });
}

int gab_conv_engine_start(gab_conv_plan* p, int ring_buffers, float** d_in_ring, float** d_out_ring, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!p || !d_in_ring || !d_out_ring) return gab::bad_arg("gab_conv_engine_start: null argument");
        if (!p->ir_set) return gab::bad_arg("gab_conv_engine_start: gab_conv_set_ir has not been called");
        if (p->eng_running) return gab::bad_arg("gab_conv_engine_start: the plan's engine is already running");
#ifdef GAB_ABLATE
        int waves = gab::kEngineWaves;                  // diagnostic builds: round 5's eight-wave engine for A/B on one box (same bits)
        if (getenv("GAB_ENGINE_WAVES")) waves = atoi(getenv("GAB_ENGINE_WAVES"));
#endif
        {
            // Every workgroup of the engine stays on the device until the stop and waits for words other workgroups write
            // (the relayed doorbell): all of them must be resident AT ONCE.  One fits per compute unit (153 KB of LDS).
            int dev = 0, cus = 0, per_cu = 0;
            GAB_HIP_CHECK(hipGetDevice(&dev));
            GAB_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
#ifdef GAB_ABLATE
            if (waves != 12) GAB_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gab::conv_split_engine_kernel, gab::kBatchThreads, 0));
            else
#endif
            GAB_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gab::conv_split_engine12_kernel, gab::kB12Threads, 0));
            const long room = (long)cus * per_cu;
            if ((long)(p->tracks / 4) > room)
                return gab::bad_arg(("gab_conv_engine_start: the engine keeps one workgroup per four channels resident for the whole launch; this device holds " +
                                     std::to_string(room) + " of them (" + std::to_string(4 * room) + " channels), the plan has " +
                                     std::to_string(p->tracks) + " channels — shard the channels (one engine per device) or use gab_conv_process_batch").c_str());
        }
        if (int rc = gab_conv_engine_rings(p, ring_buffers, d_in_ring, d_out_ring)) return rc;
        // Round 5's engine needed every compute unit WHOLE (two waves of 256 registers on each SIMD): with keep-warm waves on
        // eight of them it could not become resident until they had ended (first buffer 484 ms: profiles/r05_paced_keep_warm.txt).
        // The twelve-wave engine (three waves of 160 allocated registers per SIMD) might leave such a wave its sixteen — not
        // relied upon: a launch that PROBABLY becomes resident is the trap this rule closed, and the engine keeps the device
        // awake itself.  The plan's own keep-warm goes; anybody else's is the caller's to end — said here, at the failing call.
        if (p->warm) { (void)gab_keep_warm_destroy(p->warm); p->warm = nullptr; p->warm_on = false; }
        if (const int others = gab::resident_running(p->device, gab::kResidentKeepWarm, nullptr))
            return gab::bad_arg(("gab_conv_engine_start: " + std::to_string(others) + " gab_keep_warm launch" + (others > 1 ? "es are" : " is") +
                                 " resident on this device (a KeepWarm object, or a harness run with keep-warm on); their waves keep the engine's "
                                 "workgroups off their compute units until they end — gab_keep_warm_destroy (or stop kicking and wait out "
                                 "idle_seconds) before the engine is started").c_str());
        hipStream_t caller = gab::as_stream(stream);
        // the launch goes on the plan's own stream, behind whatever the caller's stream holds now
        GAB_HIP_CHECK(hipEventRecord(p->eng_ev, caller));
        GAB_HIP_CHECK(hipStreamWaitEvent(p->eng_own_stream, p->eng_ev, 0));
        hipStream_t s = p->eng_own_stream;
        const size_t prog_words = 2 * (size_t)(p->tracks / 4) + 64;        // + the started count and the relay word, a line of their own each
        if (!p->eng_done) GAB_HIP_CHECK(hipMalloc(&p->eng_done, prog_words * sizeof(unsigned)));
        if (!p->eng_words) {
            GAB_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&p->eng_words), 64 * sizeof(unsigned), hipHostMallocDefault));
        }
        for (int i = 0; i < 64; ++i) p->eng_words[i] = 0;
        p->order_after_reset(s);
        GAB_HIP_CHECK(hipMemsetAsync(p->eng_done, 0, prog_words * sizeof(unsigned), s));
        gab::ConvSplit sp{p->pmA2, p->pmF, p->carry GAB_SPLIT_DEBUG_ARG};
        int poll = (GAB_ENGV & 32) ? 0 : 1;
#ifdef GAB_ABLATE
        if (getenv("GAB_ENGINE_NOPOLL")) poll = 0;      // diagnostic builds: the doorbell is read only when the engine stalls

#endif
        gab::ConvEngine eng{p->eng_words, p->eng_done + prog_words - 1, p->eng_done, p->eng_words + 16, p->eng_words + 32, ring_buffers, poll,
                            p->eng_done + prog_words - 33, p->eng_words + 48, (unsigned long long)(p->eng_idle_seconds * 1e8)};
#ifdef GAB_ABLATE
        if (waves != 12)
            gab::conv_split_engine_kernel<<<dim3(p->tracks / 4), dim3(gab::kBatchThreads), 0, s>>>(
                p->eng_in, p->eng_out, p->hist, p->pmA, sp, p->tw, p->tracks, p->head, eng);
        else
#endif
        gab::conv_split_engine12_kernel<<<dim3(p->tracks / 4), dim3(gab::kB12Threads), 0, s>>>(
            p->eng_in, p->eng_out, p->hist, p->pmA, sp, p->tw, p->tracks, p->head, eng);
        int rc = gab::launch_status("conv_split_engine12_kernel");
        if (rc) return rc;
        p->eng_running = true;
        p->eng_published = 0;
        p->eng_seen_completed = 0;
        p->eng_stream = s;
        p->fresh = false;
        *d_in_ring = p->eng_in;
        *d_out_ring = p->eng_out;
        return GAB_OK;
    });
}

int gab_conv_engine_submit(gab_conv_plan* p, int n_more, int flush) {
    if (!p || !p->eng_running) return gab::bad_arg("gab_conv_engine_submit: no running engine");
    if (n_more < 0) return gab::bad_arg("gab_conv_engine_submit: negative count");
    if (p->eng_published + (unsigned)n_more >= 0x40000000u) return gab::bad_arg("gab_conv_engine_submit: more than 2^30 buffers in one launch (stop and start again)");
    p->eng_published += (unsigned)n_more;
    // the doorbell: buffers published so far; bit 30 = finish them without waiting for more (one store: count and rung together)
    __atomic_store_n(&p->eng_words[0], p->eng_published | (flush ? 0x40000000u : 0u), __ATOMIC_RELEASE);
    return GAB_OK;
}

int gab_conv_engine_publish(gab_conv_plan* p, int n_more) { return gab_conv_engine_submit(p, n_more, 0); }

int gab_conv_engine_running(gab_conv_plan* p, int* running) {
    if (!p || !running) return gab::bad_arg("gab_conv_engine_running: null argument");
    *running = 0;
    if (!p->eng_running) return GAB_OK;
    const hipError_t q = hipStreamQuery(p->eng_stream);            // the launch is the only thing on the plan's own stream
    (void)hipGetLastError();
    *running = q == hipErrorNotReady ? 1 : 0;
    return GAB_OK;
}

int gab_conv_engine_wait(gab_conv_plan* p, int count, double timeout_seconds) {
    return gab::guarded([&]() -> int {
        if (!p || !p->eng_words) return gab::bad_arg("gab_conv_engine_wait: no engine");
        if (count < 0 || (unsigned)count > p->eng_published) return gab::bad_arg("gab_conv_engine_wait: count exceeds what has been published");
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        int done = 0;
        for (;;) {
            gab_conv_engine_completed(p, &done);
            if (done >= count) return GAB_OK;
            if (p->eng_words[32]) {
                gab::set_last_error("gab_conv_engine_wait: the engine gave up waiting for the doorbell");
                return GAB_ERR_RUNTIME;
            }
            if ((++spins & 0xfffu) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_seconds) {
                // was the launch ever there?  (the first / the last workgroup to begin say so: [48], [49])
                const unsigned first = __atomic_load_n(&p->eng_words[48], __ATOMIC_ACQUIRE), all = __atomic_load_n(&p->eng_words[49], __ATOMIC_ACQUIRE);
                int on_device = 0;
                (void)gab_conv_engine_running(p, &on_device);
                std::string where;
                if (!first)
                    where = on_device ? "; the launch has NOT STARTED: no workgroup has run yet — something stands ahead of it on its hardware queue (a kernel or "
                                        "device-to-device copy on another highest-priority stream), or another resident launch fills the device"
                                      : "; the launch is no longer on the device and never ran a workgroup";
                else if (!all)
                    where = "; the launch is only PARTLY resident: some of its " + std::to_string(p->tracks / 4) + " workgroups have not begun — waves of another "
                            "launch hold registers or LDS on their compute units (every workgroup must be resident at once)";
                else
                    where = on_device ? "; every workgroup of the launch is resident and waiting for the doorbell"
                                      : "; the launch was resident and has ended";
                gab::set_last_error("gab_conv_engine_wait: " + std::to_string(done) + " of " + std::to_string(count) +
                                    " buffers reported within the time limit (without the flush rung a buffer is reported once five later ones are published)" + where);
                return GAB_ERR_RUNTIME;
            }
        }
    });
}

int gab_conv_engine_completed(gab_conv_plan* p, int* completed) {
    if (!p || !completed || !p->eng_words) return gab::bad_arg("gab_conv_engine_completed: no engine");
    const unsigned c = __atomic_load_n(&p->eng_words[16], __ATOMIC_ACQUIRE);      // two writers may cross: keep the maximum seen
    if ((int)(c - p->eng_seen_completed) > 0) p->eng_seen_completed = c;
    *completed = (int)p->eng_seen_completed;
    return GAB_OK;
}

int gab_conv_engine_feed(gab_conv_plan* p, int n_buffers, int ahead) {
    return gab::guarded([&]() -> int {
        if (!p || !p->eng_running) return gab::bad_arg("gab_conv_engine_feed: no running engine");
        // a buffer is reported back once FIVE later ones are published (asked for a period early, delivered a period late,
        // its count taken a period after that and passed on by workgroup 0): fewer than six in flight and nothing moves
        if (n_buffers < 0 || ahead < 6 || ahead >= p->eng_ring)
            return gab::bad_arg("gab_conv_engine_feed: n_buffers >= 0 and 6 <= ahead < ring_buffers (a buffer is reported back once five later ones are published)");
        const unsigned first = p->eng_published, last = first + (unsigned)n_buffers;
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        int done = 0;
        // one buffer per ring of the doorbell, never more than `ahead` buffers in front of what has come back
        while (p->eng_published < last) {
            gab_conv_engine_completed(p, &done);
            if ((int)(p->eng_published - (unsigned)done) < ahead) gab_conv_engine_publish(p, 1);
            if ((++spins & 0xfffffu) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 10.0 + 1e-4 * n_buffers) {
                gab::set_last_error("gab_conv_engine_feed: the engine stopped consuming");
                return GAB_ERR_RUNTIME;
            }
            if (p->eng_words[32]) break;
        }
        if (p->eng_words[32]) {
            gab::set_last_error("gab_conv_engine_feed: the engine gave up waiting for the doorbell");
            return GAB_ERR_RUNTIME;
        }
        return GAB_OK;
    });
}

int gab_conv_engine_feed_one_in_flight(gab_conv_plan* p, int n_buffers, float* latency_us) {
    return gab::guarded([&]() -> int {
        if (!p || !p->eng_running) return gab::bad_arg("gab_conv_engine_feed_one_in_flight: no running engine");
        if (n_buffers < 0) return gab::bad_arg("gab_conv_engine_feed_one_in_flight: negative count");
        // the real-time loop for resident rings: ring the doorbell with the flush rung, wait for that very buffer, go on
        for (int i = 0; i < n_buffers; ++i) {
            const auto t0 = std::chrono::steady_clock::now();
            if (int rc = gab_conv_engine_submit(p, 1, 1)) return rc;
            if (int rc = gab_conv_engine_wait(p, (int)p->eng_published, 5.0)) return rc;
            if (latency_us) latency_us[i] = std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - t0).count();
        }
        return GAB_OK;
    });
}

int gab_conv_engine_stop(gab_conv_plan* p) {
    return gab::guarded([&]() -> int {
        if (!p || !p->eng_running) return gab::bad_arg("gab_conv_engine_stop: no running engine");
        __atomic_store_n(&p->eng_words[0], p->eng_published | 0x80000000u, __ATOMIC_RELEASE);
        GAB_HIP_CHECK(hipStreamSynchronize(p->eng_stream));
        p->eng_running = false;
        if (p->eng_words[32]) {
            // The engine had given up (the doorbell silent for the idle limit): it ended its burst with what it had taken, and
            // buffers published behind that (a pipelined burst without the flush rung leaves the last one waiting for its
            // successor) were never consumed.  History ring, carry ring and head go on from what WAS consumed: every inverse
            // wave's own count, read now that the launch is over — not from what was published.
            const size_t waves = 2 * (size_t)(p->tracks / 4);
            std::vector<unsigned> prog(waves);
            GAB_HIP_CHECK(hipMemcpy(prog.data(), p->eng_done, waves * sizeof(unsigned), hipMemcpyDeviceToHost));
            unsigned consumed = p->eng_published;
            for (unsigned v : prog) consumed = std::min(consumed, v);
            p->eng_seen_completed = consumed;
            p->head = (p->head + (int)(consumed & 7u)) & (gab::kSlots - 1);
            gab::set_last_error("gab_conv_engine_stop: the engine had given up waiting for the doorbell (" + std::to_string(p->eng_idle_seconds) +
                                " s without a buffer or the stop: gab_conv_engine_set_idle_limit); it consumed " + std::to_string(consumed) + " of the " +
                                std::to_string(p->eng_published) + " buffers published — the plan's history continues behind buffer " + std::to_string(consumed) +
                                ", later ones were dropped and must be published again after the next start");
            return GAB_ERR_RUNTIME;
        }
        p->eng_seen_completed = p->eng_published;            // the launch has ended: everything published is finished
        p->head = (p->head + (int)(p->eng_published & 7u)) & (gab::kSlots - 1);
        return GAB_OK;
    });
}

int gab_conv_engine_set_idle_limit(gab_conv_plan* p, double seconds) {
    if (!p) return gab::bad_arg("gab_conv_engine_set_idle_limit: null plan");
    if (!(seconds >= 0.5 && seconds <= 3600.0)) return gab::bad_arg("gab_conv_engine_set_idle_limit: 0.5 .. 3600 seconds");
    if (p->eng_running) return gab::bad_arg("gab_conv_engine_set_idle_limit: the plan's engine is running (it takes the limit at its start)");
    p->eng_idle_seconds = seconds;
    return GAB_OK;
}

// The reference's iteration through the engine (cuda/bench_base.cu:30-42 around bench_conv1d_accel.cu:258-304): pinned host ->
// ring slot (engine copy), doorbell with the flush rung, wait for that buffer, ring slot -> pinned host (engine copy).  The two
// link legs do not overlap with the transform (gab_conv_round_trip's do): this is the per-buffer engine as a complete
// replacement of that iteration with a stated number, not the fastest round trip.
int gab_conv_engine_round_trip(gab_conv_plan* p, const float* h_in, float* h_out) {
    return gab::guarded([&]() -> int {
        if (!p || !h_in || !h_out) return gab::bad_arg("gab_conv_engine_round_trip: null argument");
        if (!p->eng_running) return gab::bad_arg("gab_conv_engine_round_trip: no running engine (gab_conv_engine_start first)");
        int done = 0;
        gab_conv_engine_completed(p, &done);
        if ((unsigned)done != p->eng_published)
            return gab::bad_arg("gab_conv_engine_round_trip: buffers are still in flight (ONE buffer at a time: wait for what has been published)");
        if (!p->eng_copy_stream) {
            GAB_HIP_CHECK(hipStreamCreateWithFlags(&p->eng_copy_stream, hipStreamNonBlocking));
            GAB_HIP_CHECK(hipEventCreateWithFlags(&p->eng_copy_ev, hipEventDisableTiming));
        }
        const size_t n = (size_t)p->tracks * p->bufsize;
        const size_t slot = p->eng_published % (unsigned)p->eng_ring;
        auto spin = [&](const char* what) -> int {          // (an event query: hipStreamSynchronize would yield the thread)
            GAB_HIP_CHECK(hipEventRecord(p->eng_copy_ev, p->eng_copy_stream));
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned spins = 0;;) {
                const hipError_t q = hipEventQuery(p->eng_copy_ev);
                if (q == hipSuccess) return GAB_OK;
                (void)hipGetLastError();
                if (q != hipErrorNotReady) GAB_HIP_CHECK(q);
                if ((++spins & 1023u) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 4.0) {
                    gab::set_last_error(std::string("gab_conv_engine_round_trip: ") + what + " did not end within 4 s (pinned host memory keeps it on a copy engine; "
                                        "pageable memory may send it through a kernel, which cannot run beside the engine)");
                    return GAB_ERR_RUNTIME;
                }
            }
        };
        GAB_HIP_CHECK(hipMemcpyAsync(p->eng_in + slot * n, h_in, n * sizeof(float), hipMemcpyHostToDevice, p->eng_copy_stream));
        if (int rc = spin("the upload")) return rc;
        if (int rc = gab_conv_engine_submit(p, 1, 1)) return rc;
        if (int rc = gab_conv_engine_wait(p, (int)p->eng_published, 5.0)) return rc;
        GAB_HIP_CHECK(hipMemcpyAsync(h_out, p->eng_out + slot * n, n * sizeof(float), hipMemcpyDeviceToHost, p->eng_copy_stream));
        return spin("the download");
    });
}

int gab_conv_process_batch(gab_conv_plan* p, const float* d_in, float* d_out, int n_buffers,
                           gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!p || !d_in || !d_out) return gab::bad_arg("gab_conv_process_batch: null argument");
        if (!p->ir_set) return gab::bad_arg("gab_conv_process_batch: gab_conv_set_ir has not been called");
        if (p->eng_running) return gab::bad_arg("gab_conv_process_batch: the plan's engine is running and owns its history (gab_conv_engine_stop first)");
        if (n_buffers <= 0) return gab::bad_arg("gab_conv_process_batch: n_buffers must be > 0");
        hipStream_t s = gab::as_stream(stream);
        if (p->fused && (p->split || p->tail)) {
            // At most kBatchChunk buffers per launch.  Nothing holds the workgroups of a launch together, and the eight duos
            // whose 16-byte pieces make up one 128-byte output line must store within a period or two of each other — the
            // spectra streaming through an XCD's L2 turn it over every two periods — or the line leaves the L2 in pieces.
            // Over many hundred periods they drift apart: at 2048 buffers per launch the L2s send 2.75 x the write requests
            // and 1.9 x the bytes per buffer, and a buffer takes 5.63 us instead of 5.17 (profiles/r05_batch_buffers_per_launch.txt;
            // the classic cut's launch: 10.5 instead of 9.3).  A launch boundary puts them back in step for the price of one
            // cold start per chunk (0.03 us per buffer).
            size_t chunk_max = gab::kBatchChunk;
#ifdef GAB_ABLATE
            if (getenv("GAB_BATCH_CHUNK")) chunk_max = std::max(1, atoi(getenv("GAB_BATCH_CHUNK")));   // diagnostic builds: to measure the above
#endif
            if (p->split) p->order_after_reset(s);
            gab::ConvSplit sp{p->pmA2, p->pmF, p->carry GAB_SPLIT_DEBUG_ARG};
            const size_t step = (size_t)p->tracks * p->bufsize;
            for (int done = 0; done < n_buffers;) {
                const int n = (int)std::min<size_t>(chunk_max, (size_t)(n_buffers - done));
                int rc;
                if (p->split) {
                    // the split cut, both roles of a duo in one resident workgroup: same bits as n split launches.
                    // Twelve waves, three per SIMD (round 6); diagnostic builds keep the other forms for A/B on one box: same bits
#ifdef GAB_ABLATE
                    int waves = gab::kBatchWaves;
                    if (getenv("GAB_BATCH_WAVES")) waves = atoi(getenv("GAB_BATCH_WAVES"));
                    if (waves == 26) {
                        gab::conv_split_batch2x6_kernel<<<dim3(p->tracks / 4), dim3(gab::kB26Threads), 0, s>>>(
                            d_in + done * step, d_out + done * step, p->hist, p->pmA, sp, p->tw, p->tracks, p->head, n);
                        rc = gab::launch_status("conv_split_batch2x6_kernel");
                    } else if (waves == 64) {                   // six waves per pair at four waves per SIMD (128 registers)
                        static bool dyn_set = false;
                        if (!dyn_set) {
                            GAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gab::conv_split_batch6r_kernel),
                                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(gab::kB6Lds * sizeof(gab::fft::cf))));
                            dyn_set = true;
                        }
                        gab::conv_split_batch6r_kernel<<<dim3(p->tracks / 2), dim3(gab::kB6Threads), gab::kB6Lds * sizeof(gab::fft::cf), s>>>(
                            d_in + done * step, d_out + done * step, p->hist, p->pmA, sp, p->tw, p->tracks, p->head, n);
                        rc = gab::launch_status("conv_split_batch6r_kernel");
                    } else if (waves == 6) {
                        gab::conv_split_batch6_kernel<<<dim3(p->tracks / 2), dim3(gab::kB6Threads), 0, s>>>(
                            d_in + done * step, d_out + done * step, p->hist, p->pmA, sp, p->tw, p->tracks, p->head, n);
                        rc = gab::launch_status("conv_split_batch6_kernel");
                    } else if (waves == 8) {                    // round 5's eight-wave workgroup
                        gab::conv_split_batch_kernel<<<dim3(p->tracks / 4), dim3(gab::kBatchThreads), 0, s>>>(
                            d_in + done * step, d_out + done * step, p->hist, p->pmA, sp, p->tw, p->tracks, p->head, n);
                        rc = gab::launch_status("conv_split_batch_kernel");
                    } else
#endif
                    {
                        gab::conv_split_batch12_kernel<<<dim3(p->tracks / 4), dim3(gab::kB12Threads), 0, s>>>(
                            d_in + done * step, d_out + done * step, p->hist, p->pmA, sp, p->tw, p->tracks, p->head, n);
                        rc = gab::launch_status("conv_split_batch12_kernel");
                    }
                } else {
                    gab::conv_batch_kernel<<<dim3(p->pairs), dim3(gab::kThreads), 0, s>>>(
                        d_in + done * step, d_out + done * step, p->hist, p->pmA, p->pmB, p->tw, p->tracks, p->head, n);
                    rc = gab::launch_status("conv_batch_kernel");
                }
                if (rc) return rc;
                p->head = (p->head + n) & (gab::kSlots - 1);
                p->fresh = false;
                done += n;
            }
            return GAB_OK;
        }
        // other shapes: one buffer at a time
        const size_t step = (size_t)p->tracks * p->bufsize;
        for (int n = 0; n < n_buffers; ++n) {
            int rc = gab_conv_process(p, d_in + n * step, d_out + n * step, GAB_CONV_STREAMING, stream);
            if (rc) return rc;
        }
        return GAB_OK;
    });
}

int gab_conv_state_bytes(const gab_conv_plan* p, size_t* spectra, size_t* history) {
    if (!p) return gab::bad_arg("gab_conv_state_bytes: null plan");
    // everything resident for the plan: a plan that can use the split cut holds both sets of spectra
    // (classic for batch / host-io launches) and the carry ring beside the history ring
    const size_t bank = p->pmF ? sizeof(float4) * (size_t)p->pairs * (gab::kBinsA + gab::kBinsB) : 0;
    if (spectra) *spectra = p->spectra_bytes + bank;
    if (history) *history = p->history_bytes + p->carry_bytes;
    return GAB_OK;
}

#ifdef GAB_ABLATE
// diagnostic builds only: how many words of the round trip's staging buffer are NOT the sentinel (between calls: none)
int gab_debug_rt_stage_dirty(gab_conv_plan* p, long long* first_index) {
    if (!p || !p->rt_stage) return -1;
    (void)hipDeviceSynchronize();
    const size_t n = 2 * (size_t)p->tracks * p->bufsize;          // both staging buffers
    std::vector<unsigned> h(n);
    if (hipMemcpy(h.data(), p->rt_stage, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    int dirty = 0;
    if (first_index) *first_index = -1;
    for (size_t i = 0; i < n; ++i)
        if (h[i] != gab::kRtSentinel) {
            if (!dirty && first_index) *first_index = (long long)i;
            ++dirty;
        }
    return dirty;
}
// diagnostic builds only: arm (mins to ~0ull, maxes to 0) / read the round-trip kernel's per-group stamps
int gab_debug_rt_stamps(unsigned long long* h_out, int arm) {
    (void)hipDeviceSynchronize();
    if (arm) {
        unsigned long long init[65 * 4];
        for (int i = 0; i < 65 * 4; ++i) init[i] = (i % 4 == 0 && i < 64 * 4) ? ~0ull : 0ull;
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(gab::g_rt_stamps), init, sizeof init);
    }
    return (int)hipMemcpyFromSymbol(h_out, HIP_SYMBOL(gab::g_rt_stamps), sizeof(unsigned long long) * 65 * 4);
}
// diagnostic builds only: copies the phase stamps of the last stamped split launch
int gab_debug_split_stamps(unsigned long long* h_out, int n) {
    (void)hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(h_out, HIP_SYMBOL(gab::g_split_stamps), sizeof(unsigned long long) * n);
}
#endif

int gab_fft_r2c_1024(const float* d_in, float* d_out, int tracks, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!d_in || !d_out) return gab::bad_arg("gab_fft_r2c_1024: null pointer");
        if (tracks <= 0) return gab::bad_arg("gab_fft_r2c_1024: tracks must be > 0");
        const gab::fft::cf* tw = gab::fft::device_twiddles();
        gab::fft_r2c_1024_kernel<<<(tracks + 7) / 8, gab::kThreads, 0, gab::as_stream(stream)>>>(
            d_in, reinterpret_cast<float2*>(d_out), tw, tracks);
        return gab::launch_status("fft_r2c_1024_kernel");
    });
}

}  // extern "C"
