// k_basic.hip — bandwidth kernels: noop, gain, gainstats, datatransfer, modal,
// rndmem.  Reference kernels are one-thread-per-track serial loops with a
// stride-B (uncoalesced) access pattern (cuda/bench_gain.cu:6-24 etc.); here
// every kernel is element- or wave-parallel with 16-byte coalesced accesses.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <memory>

#include "gab_common.hpp"

namespace gab {
namespace {

constexpr int kBlock = 256;
constexpr int kMaxGrid = 8192;     // 256 CUs x 32 blocks: grid-stride beyond that

inline int grid_for(size_t work_items) {
    size_t g = (work_items + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    return (int)(g > (size_t)kMaxGrid ? kMaxGrid : g);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- noop / gain ------------------------------------------------------------
// out = in * gain when SCALE, else a plain copy.  n4 float4 groups + scalar tail.
template <bool SCALE>
__global__ __launch_bounds__(kBlock) void scale_vec4_kernel(const float4* __restrict__ in,
                                                           float4* __restrict__ out, size_t n4,
                                                           const float* __restrict__ in_tail,
                                                           float* __restrict__ out_tail, int tail,
                                                           float gain) {
    // a workgroup takes kUnroll consecutive 4 KiB rows; every lane has kUnroll 16-byte loads in
    // flight before its first store
    constexpr int kUnroll = 4;
    const size_t stride = (size_t)gridDim.x * kBlock * kUnroll;
    for (size_t base = (size_t)blockIdx.x * kBlock * kUnroll + threadIdx.x; base < n4; base += stride) {
        float4 v[kUnroll];
        if (base + (size_t)(kUnroll - 1) * kBlock < n4) {
            // whole rows: unconditional requests (a load under a per-lane condition is followed by a
            // register merge, i.e. by an s_waitcnt vmcnt(0) BETWEEN the loads: the copy kernel ran 6.2 us
            // at 128 tracks where the gain kernel, whose multiply hides the merge, ran 3.8)
#pragma unroll
            for (int k = 0; k < kUnroll; ++k) v[k] = in[base + (size_t)k * kBlock];
#pragma unroll
            for (int k = 0; k < kUnroll; ++k) {
                if (SCALE) { v[k].x = gain * v[k].x; v[k].y = gain * v[k].y; v[k].z = gain * v[k].z; v[k].w = gain * v[k].w; }
                out[base + (size_t)k * kBlock] = v[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < kUnroll; ++k) {
                if (base + (size_t)k * kBlock < n4) {
                    float4 u = in[base + (size_t)k * kBlock];
                    if (SCALE) { u.x = gain * u.x; u.y = gain * u.y; u.z = gain * u.z; u.w = gain * u.w; }
                    out[base + (size_t)k * kBlock] = u;
                }
            }
        }
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < tail)
        out_tail[threadIdx.x] = SCALE ? gain * in_tail[threadIdx.x] : in_tail[threadIdx.x];
}

template <bool SCALE>
__global__ __launch_bounds__(kBlock) void scale_scalar_kernel(const float* __restrict__ in,
                                                             float* __restrict__ out, size_t n,
                                                             float gain) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (; i < n; i += stride) out[i] = SCALE ? gain * in[i] : in[i];
}

template <bool SCALE>
int launch_scale(const float* d_in, float* d_out, size_t n, float gain, hipStream_t s) {
    if (n == 0) return GAB_OK;
    if (aligned16(d_in) && aligned16(d_out)) {
        size_t n4 = n / 4;
        int tail = (int)(n - 4 * n4);
        scale_vec4_kernel<SCALE><<<grid_for((n4 + 3) / 4), kBlock, 0, s>>>(
            reinterpret_cast<const float4*>(d_in), reinterpret_cast<float4*>(d_out), n4,
            d_in + 4 * n4, d_out + 4 * n4, tail, gain);
    } else {
        scale_scalar_kernel<SCALE><<<grid_for(n), kBlock, 0, s>>>(d_in, d_out, n, gain);
    }
    return launch_status(SCALE ? "gain kernel" : "noop kernel");
}

// ---- gainstats -----------------------------------------------------------------
// One wavefront per track: coalesced loads, out = gain*in, then a 64-lane
// butterfly (DPP/ds_swizzle under __shfl_xor) for sum and max.  The sum order
// differs from the golden's sequential loop; max and the scaled output are exact.
template <bool VEC4>
__global__ __launch_bounds__(kBlock) void gainstats_kernel(const float* __restrict__ in,
                                                          float* __restrict__ out,
                                                          float* __restrict__ stats, int T, int B,
                                                          float gain) {
    const int lane = threadIdx.x & (kWave - 1);
    const int track = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (track >= T) return;
    const float* x = in + (size_t)track * B;
    float* y = out + (size_t)track * B;
    float sum = 0.0f, mx = -1e9f;
    if (VEC4) {
        const float4* x4 = reinterpret_cast<const float4*>(x);
        float4* y4 = reinterpret_cast<float4*>(y);
        for (int i = lane; i < B / 4; i += kWave) {
            float4 v = x4[i];
            sum += (v.x + v.y) + (v.z + v.w);
            mx = fmaxf(fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)), mx);
            y4[i] = make_float4(v.x * gain, v.y * gain, v.z * gain, v.w * gain);
        }
    } else {
        for (int i = lane; i < B; i += kWave) {
            float v = x[i];
            sum += v;
            mx = fmaxf(v, mx);
            y[i] = v * gain;
        }
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        sum += __shfl_xor(sum, off, kWave);
        mx = fmaxf(mx, __shfl_xor(mx, off, kWave));
    }
    if (lane == 0) {
        stats[2 * track + 0] = sum / (float)B;
        stats[2 * track + 1] = mx;
    }
}

// ---- datatransfer ----------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void datatransfer_kernel(const float* __restrict__ in,
                                                             float* __restrict__ out, int in_size,
                                                             int out_size) {
    int i = blockIdx.x * kBlock + threadIdx.x;
    const int stride = gridDim.x * kBlock;
    for (; i < out_size; i += stride)
        out[i] = (i < in_size) ? in[i] : 0.5f + 0.5f * sinf((float)i * 0.001f);
}

// ---- datatransfer, both link directions at once ------------------------------------
// gab_datatransfer_round_trip.  The reference moves the input up, runs its kernel and moves the output
// down, one after the other (cuda/bench_datatransfer.cu:62-75).  Here (same findings as the convolver's
// round trip, tools/ubench/link_modes): ONE engine copy of the input into `stage` (fine-grained device
// memory: a running kernel sees it land) on the plan's own stream, and a kernel launched at once that
// writes the pinned output itself — an engine upload beside shader writes to host memory runs both
// directions at full rate.  The output is cut into 4 KiB chunks: the ones that need no input (the sine
// tail) go first, the dependent ones follow in ascending order, dealt round-robin to the workgroups so
// that they leave in the order the copy lands them.  A consumed input word is overwritten with a sentinel
// (a NaN the benchmark's [0,1] input never holds); "my words are no longer the sentinel" means they have
// landed, and an input that really holds it is released by the `landed` word the host sets once the copy's
// event has completed.  Same expression per word as datatransfer_kernel: bit-identical output.
constexpr unsigned kLinkSentinel = 0xffa5c3e1u;
// landed = the word's top byte is no longer the sentinel's (k_conv_accel.hip, rt_pending: a word cut by an engine-packet boundary
// shows the input's low bytes under the sentinel's high ones for a moment)
__device__ __forceinline__ bool link_pending(unsigned w) { return (w >> 24) == (kLinkSentinel >> 24); }
constexpr int kLinkPollLimit = 1 << 21;            // x ~0.5 us of s_sleep: about a second, then the launch gives up
constexpr int kLinkChunk = 4 * kBlock;             // words per chunk: one float4 per thread
struct LinkRoundTrip {
    unsigned* stage;              // [>= in_size] fine-grained device memory, all sentinel between calls
    float* h_out;                 // [out_size] pinned host memory
    unsigned* counter;            // device: workgroups finished, runs on from call to call
    unsigned* done;               // pinned host: the epoch, once every workgroup has finished (a HINT: the call waits for the launch's end)
    const unsigned* landed;       // pinned host: the epoch, once the host has seen the upload complete
    unsigned* error;              // pinned host: bit 0 a wait ran out, bit 1 a consumed word is not what the completed upload left
    unsigned* consumed;           // device: [>= in_size] the words as the kernel took them, until they have been checked
    unsigned epoch;
    int in_size, out_size;
};
constexpr unsigned kLinkErrWait = 1u, kLinkErrTorn = 2u;

__device__ __forceinline__ unsigned link_peek(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(kBlock) void datatransfer_round_trip_kernel(LinkRoundTrip rt) {
    __shared__ int s_word;
    const int tid = threadIdx.x;
    const int dep = min(rt.in_size, rt.out_size);                    // outputs [0, dep) are input words
    const int chunks = (rt.out_size + kLinkChunk - 1) / kLinkChunk;
    const int dep_chunks = (dep + kLinkChunk - 1) / kLinkChunk;      // chunks that hold at least one input word
    const int free_chunks = chunks - dep_chunks;
    bool gave_up = false;
    for (int pos = blockIdx.x; pos < chunks; pos += gridDim.x) {
        const int chunk = pos < free_chunks ? dep_chunks + pos : pos - free_chunks;
        const int w0 = chunk * kLinkChunk + 4 * tid;                  // this thread's four words
        const int n_in = max(0, min(4, dep - w0));                    // how many of them are input words
        unsigned w[4] = {0, 0, 0, 0};
        if (chunk < dep_chunks) {
            // one lane watches the chunk's last input word; then every lane checks its own (the copy need not
            // land in ascending order)
            if (tid == 0) {
                const unsigned* const last = rt.stage + min(chunk * kLinkChunk + kLinkChunk, dep) - 1;
                int tries = 0, bad = gave_up ? 1 : 0;
                while (!bad && link_pending(link_peek(last))) {
                    if ((++tries & 63) == 0 && link_peek(rt.landed) == rt.epoch) break;     // the upload is in: it IS the sentinel
                    if (tries > kLinkPollLimit) { bad = 1; break; }
                    __builtin_amdgcn_s_sleep(20);
                }
                s_word = bad;
            }
            __syncthreads();
            if (s_word) gave_up = true;
            __syncthreads();                                          // s_word is free for the next chunk
            int tries = 0;
            for (;;) {
                bool all = true;
                for (int k = 0; k < 4; ++k)
                    if (k < n_in) { w[k] = link_peek(rt.stage + w0 + k); all = all && !link_pending(w[k]); }
                if (all || gave_up) break;
                // a word that is STILL the sentinel is a value only behind an acquire of `landed` (released by the host
                // after the copy's completion event): one more look after the acquire is final
                if ((++tries & 15) == 0 && __hip_atomic_load(rt.landed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == rt.epoch) {
                    for (int k = 0; k < 4; ++k)
                        if (k < n_in) w[k] = link_peek(rt.stage + w0 + k);
                    break;
                }
                if (tries > kLinkPollLimit) { gave_up = true; break; }
                __builtin_amdgcn_s_sleep(10);
            }
            // taken — and kept beside the staging buffer until they have been checked against the completed upload (below)
            for (int k = 0; k < 4; ++k)
                if (k < n_in) rt.consumed[w0 + k] = w[k];
        }
        typedef float f4v __attribute__((ext_vector_type(4)));
        f4v val;
        for (int k = 0; k < 4; ++k) {
            const int i = w0 + k;
            val[k] = (k < n_in) ? __uint_as_float(w[k]) : 0.5f + 0.5f * sinf((float)i * 0.001f);
        }
        float* const dst = rt.h_out + w0;
        if (w0 + 4 <= rt.out_size) {
            // system-scope write-through: nothing of it stays behind in a cache
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(val) : "memory");
        } else {
            for (int k = 0; k < 4; ++k)
                if (w0 + k < rt.out_size) __hip_atomic_store(dst + k, val[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    // every wave waits for its rows, then the workgroup counts as finished; the one whose count completes the launch
    // writes the hint word (the host then waits for the launch's END: completion is the stream's, not this word's)
    if (gave_up) __hip_atomic_store(rt.error, kLinkErrWait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(rt.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == rt.epoch * gridDim.x) __hip_atomic_store(rt.done, rt.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Round 6 (see k_conv_accel.hip, conv_round_trip_check_kernel): the words the kernel above took while the upload was still
// running — it keeps a copy in `consumed` and leaves the staging buffer as it is — against what the COMPLETED upload left: a
// second launch behind it, ordered behind the upload's completion event, compares and puts the sentinel back.  Its verdict is
// read by the plan's next call or by gab_datatransfer_round_trip_check.
__global__ __launch_bounds__(kBlock) void datatransfer_round_trip_check_kernel(unsigned* __restrict__ stage, const unsigned* __restrict__ consumed,
                                                                              unsigned* __restrict__ verdict, int dep) {
    bool torn = false;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < dep; i += gridDim.x * kBlock) {
        torn = torn || link_peek(stage + i) != consumed[i];
        __hip_atomic_store(stage + i, kLinkSentinel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (torn) __hip_atomic_fetch_or(verdict, kLinkErrTorn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- modal (placeholder semantics) ---------------------------------------------
// One workgroup per output row; only params[8*i] of the first out_tracks modes
// can reach the output in the reference kernel.
__global__ __launch_bounds__(kBlock) void modal_placeholder_kernel(const float* __restrict__ params,
                                                                  float* __restrict__ out,
                                                                  int n_modes, int B) {
    const int i = blockIdx.x;
    if (i >= n_modes) return;
    const float amp = params[(size_t)i * 8 + 0];
    // Re(exp(0.5 + 0.5i)) = exp(0.5) * cos(0.5)   (cuda/bench_modal.cu:5-13, :24-29)
    const float value = amp * (expf(0.5f) * cosf(0.5f));
    for (int s = threadIdx.x; s < B; s += kBlock) out[(size_t)i * B + s] = value;
}

// ---- rndmem -----------------------------------------------------------------------
// 64x64 tile: rows = tracks read along the sample axis (256 B per wave load,
// arbitrary 4-byte alignment since playheads are random), transposed through
// LDS, written with tracks along the lanes: out[T*i + t].  A wave's 16 rows are
// requested back to back (their playheads are wave-uniform scalar loads), so a
// workgroup pays the pool's miss latency once, not once per batch of four.
// A row read at an arbitrary 4-byte offset straddles three 128-byte lines per 256 bytes, and the
// edge lines are also the neighbouring sample tile's: with the tiles of one row group on eight
// different XCDs every edge line crossed the fabric twice (FETCH_SIZE 1.54x the algorithmic reads).
// The 1-D grid is therefore walked so that the sample tiles of a track tile are CONSECUTIVE
// workgroups of ONE XCD (blocks b and b+8 share an XCD): they meet in that XCD's L2.
// (Round 3, measured and not kept: 16-byte accesses — a lane reading four consecutive samples through
// an unaligned dwordx4 load and storing four neighbouring tracks — 54.8 us against 46.5 at 65 536
// tracks: the unaligned wide loads are split and cost more than the instructions they save.  Wide STORES alone —
// a 256-track x 64-sample tile written to LDS sample-major, each lane storing four neighbouring tracks, one
// contiguous KiB per store instruction, loads as here — 66.5 us against 49.6 on the same box: the 66 KB tile
// leaves 8 waves per CU where this kernel keeps 32, and the gather lives on waves in flight.)
__global__ __launch_bounds__(kBlock) void rndmem_kernel(const float* __restrict__ pool,
                                                       const int* __restrict__ playheads,
                                                       float* __restrict__ out, int T, int B) {
    __shared__ float tile[64][65];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nst = (B + 63) / 64, ntt = (T + 63) / 64;          // sample tiles, track tiles
    int st, tt;
    if ((ntt & 7) == 0) {
        const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
        st = seq % nst;
        tt = (seq / nst) * 8 + xcd;
    } else {
        st = blockIdx.x % nst;
        tt = blockIdx.x / nst;
    }
    const int i0 = st * 64, t0 = tt * 64;
    const bool col_ok = i0 + lane < B;
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int t = t0 + w + 4 * k;                      // wave-uniform
        const size_t ph = t < T ? (size_t)playheads[t] : 0;
        v[k] = (t < T && col_ok) ? pool[ph + i0 + lane] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) tile[w + 4 * k][lane] = v[k];
    __syncthreads();
    const int t = t0 + lane;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int i = i0 + w + 4 * k;
        if (t < T && i < B) out[(size_t)T * i + t] = tile[lane][w + 4 * k];
    }
}

// ---- keep-warm: a resident launch that does nothing ------------------------------------------------------------
// A device left alone for a DAW slot (512 / 48000 s) answers the next call later than one that has just been busy:
// gab_conv_round_trip's p50 is 76-81 us paced against 67-70 back to back on the same box, and 67-68 paced with EIGHT idle
// waves resident beside (one per XCD; one wave buys nothing, 64 and more cost: profiles/r05_paced_keep_warm.txt).  These waves sleep, look at a pinned word, sleep:
// no LDS, no memory traffic but the look.  They end at the stop word, and by themselves `idle_ticks` of the 100 MHz
// wall clock after the last kick — the exit every wave reaches.
constexpr unsigned kKeepWarmStop = 0xffffffffu;
constexpr int kKeepWarmWords = 32;          // [0] the kick count / stop word, [16] set by the launch as it ends; then two words per workgroup

// Workgroup 0 ALONE decides that the launch has been idle long enough: it says so in `gone`, and the other waves leave when
// they see that word or the stop word.  (Round 5 let every wave run its own timer: a kick arriving right at expiry could
// keep some waves and lose others — a launch that looked alive with fewer waves than asked for, or one that looked gone
// with stragglers the next kick then waited idle_seconds for.)  The others' own timer is only the exit they reach if
// workgroup 0 never says anything: twice the limit.
// `where`: [2 b] HW_ID, [2 b + 1] XCC_ID | 1 << 31 of workgroup b's wave, written once at its start — which XCD, shader
// engine and compute unit the wave landed on (profiles/r05_paced_keep_warm.txt has a half-run in which explicitly made
// objects bought nothing; the one fact that would classify it is whether the eight waves sat on eight XCDs).
__global__ __launch_bounds__(64) void keep_warm_kernel(const unsigned* kick, unsigned* gone, unsigned* where,
                                                       unsigned long long idle_ticks, int naps) {
    if (threadIdx.x == 0) {
        where[2 * blockIdx.x] = __builtin_amdgcn_s_getreg(63492);                    // HW_ID
        __hip_atomic_store(&where[2 * blockIdx.x + 1], __builtin_amdgcn_s_getreg(63508) | 0x80000000u,   // XCC_ID, behind the first word
                           __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const bool decides = blockIdx.x == 0;
    unsigned last = link_peek(kick);
    unsigned long long since = wall_clock64();
    for (;;) {
        for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(127);       // 127 x 64 clocks: about 4 us each
        const unsigned k = link_peek(kick);
        if (k == kKeepWarmStop) break;
        if (!decides && link_peek(gone) != 0) break;
        const unsigned long long now = wall_clock64();
        if (k != last) { last = k; since = now; }
        else if (now - since > (decides ? idle_ticks : 2 * idle_ticks)) break;
    }
    if (decides && threadIdx.x == 0) __hip_atomic_store(gone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace
}  // namespace gab

extern "C" {

int gab_noop(const float* d_in, float* d_out, size_t n, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if ((!d_in || !d_out) && n) return gab::bad_arg("gab_noop: null pointer");
        return gab::launch_scale<false>(d_in, d_out, n, 1.0f, gab::as_stream(stream));
    });
}

int gab_gain(const float* d_in, float* d_out, size_t n, float gain, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if ((!d_in || !d_out) && n) return gab::bad_arg("gab_gain: null pointer");
        return gab::launch_scale<true>(d_in, d_out, n, gain, gab::as_stream(stream));
    });
}

int gab_gainstats(const float* d_in, float* d_out, float* d_stats, int tracks, int bufsize,
                  float gain, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!d_in || !d_out || !d_stats) return gab::bad_arg("gab_gainstats: null pointer");
        if (tracks <= 0 || bufsize <= 0) return gab::bad_arg("gab_gainstats: tracks and bufsize must be > 0");
        const int waves_per_block = gab::kBlock / gab::kWave;
        dim3 grid((tracks + waves_per_block - 1) / waves_per_block);
        hipStream_t s = gab::as_stream(stream);
        if ((bufsize % 4) == 0 && gab::aligned16(d_in) && gab::aligned16(d_out))
            gab::gainstats_kernel<true><<<grid, gab::kBlock, 0, s>>>(d_in, d_out, d_stats, tracks, bufsize, gain);
        else
            gab::gainstats_kernel<false><<<grid, gab::kBlock, 0, s>>>(d_in, d_out, d_stats, tracks, bufsize, gain);
        return gab::launch_status("gainstats_kernel");
    });
}

int gab_datatransfer(const float* d_in, float* d_out, int in_size, int out_size,
                     gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (in_size < 0 || out_size < 0) return gab::bad_arg("gab_datatransfer: negative size");
        if (out_size == 0) return GAB_OK;
        if (!d_out || (!d_in && in_size)) return gab::bad_arg("gab_datatransfer: null pointer");
        gab::datatransfer_kernel<<<gab::grid_for((size_t)out_size), gab::kBlock, 0, gab::as_stream(stream)>>>(
            d_in, d_out, in_size, out_size);
        return gab::launch_status("datatransfer_kernel");
    });
}

struct gab_link_plan {
    int max_in = 0;
    int workgroups = 256;
    unsigned* stage = nullptr;        // fine-grained device memory, max_in words
    unsigned* counter = nullptr;      // device
    unsigned* words = nullptr;        // pinned: [0] done, [16] landed, [32] error
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_ev = nullptr;
    hipEvent_t done_ev = nullptr;      // the launch's own completion (hipExtLaunchKernelGGL's stop event)
    unsigned epoch = 0;
    const void* checked_out = nullptr;
    // input words an earlier call uploaded but did not consume (its input was longer than its output): they hold
    // data, not the sentinel, and a later call that reads them must find the sentinel first
    int stale_lo[2] = {0, 0}, stale_hi[2] = {0, 0};     // (per staging buffer)
    size_t side_words = 0;              // words per staging buffer: the plan holds TWO (stage, consumed), taken in turn by call parity, so that
                                        // no call waits for the check launch of the call before it
    unsigned* consumed = nullptr;       // device: the words as the kernel took them (checked against the completed upload)
    hipEvent_t check_ev[2] = {nullptr, nullptr};      // behind datatransfer_round_trip_check_kernel, per staging buffer (verdicts: words[48], words[56])
    bool check_pending[2] = {false, false};
    ~gab_link_plan() {
        if (consumed) (void)hipFree(consumed);
        for (hipEvent_t e : check_ev) if (e) (void)hipEventDestroy(e);
        if (stage) (void)hipFree(stage);
        if (counter) (void)hipFree(counter);
        if (words) (void)hipHostFree(words);
        if (copy_ev) (void)hipEventDestroy(copy_ev);
        if (done_ev) (void)hipEventDestroy(done_ev);
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
    }
};

int gab_link_plan_create(int max_in_size, gab_link_plan** out) {
    return gab::guarded([&]() -> int {
        if (!out) return gab::bad_arg("gab_link_plan_create: null argument");
        *out = nullptr;
        if (max_in_size < 0) return gab::bad_arg("gab_link_plan_create: negative size");
        std::unique_ptr<gab_link_plan> p(new gab_link_plan);
        p->max_in = max_in_size;
        const size_t n = (size_t)std::max(max_in_size, 4);
        p->side_words = (n + 63) & ~(size_t)63;
        GAB_HIP_CHECK(hipExtMallocWithFlags(reinterpret_cast<void**>(&p->stage), 2 * p->side_words * 4, hipDeviceMallocFinegrained));
        GAB_HIP_CHECK(hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(p->stage), (int)gab::kLinkSentinel, 2 * p->side_words));
        GAB_HIP_CHECK(hipMalloc(&p->consumed, 2 * p->side_words * 4));
        for (hipEvent_t& e : p->check_ev) GAB_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        GAB_HIP_CHECK(hipMalloc(&p->counter, 128));
        GAB_HIP_CHECK(hipMemset(p->counter, 0, 128));
        GAB_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&p->words), 64 * sizeof(unsigned), hipHostMallocDefault));
        for (int i = 0; i < 64; ++i) p->words[i] = 0;
        {   // (the upload's stream at the highest priority: never on a hardware queue with the caller's — k_conv_accel.hip, gab_conv_round_trip_init)
            int lo = 0, hi = 0;
            GAB_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
            GAB_HIP_CHECK(hipStreamCreateWithPriority(&p->copy_stream, hipStreamNonBlocking, hi));
        }
        GAB_HIP_CHECK(hipEventCreateWithFlags(&p->copy_ev, hipEventDisableTiming));
        GAB_HIP_CHECK(hipEventCreateWithFlags(&p->done_ev, hipEventDisableTiming));
#ifdef GAB_ABLATE
        if (getenv("GAB_LINK_WGS")) p->workgroups = std::max(1, atoi(getenv("GAB_LINK_WGS")));
#endif
        GAB_HIP_CHECK(hipDeviceSynchronize());
        *out = p.release();
        return GAB_OK;
    });
}

void gab_link_plan_destroy(gab_link_plan* p) {
    if (!p) return;
    (void)hipDeviceSynchronize();
    delete p;
}

static int gab_link_finish_check(gab_link_plan* p, int b, bool wait, const char* who) {
    if (!p->check_pending[b]) return GAB_OK;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;;) {
        const hipError_t q = hipEventQuery(p->check_ev[b]);
        if (q == hipSuccess) break;
        (void)hipGetLastError();
        if (q != hipErrorNotReady) GAB_HIP_CHECK(q);
        if (!wait) return GAB_OK;                                  // (still running: its verdict is read later)
        if ((++spins & 1023u) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 4.0) {
            gab::set_last_error(std::string(who) + ": the check launch behind an earlier round trip did not end within 4 s");
            return GAB_ERR_RUNTIME;
        }
    }
    p->check_pending[b] = false;
    const unsigned verdict = __atomic_load_n(&p->words[48 + 8 * b], __ATOMIC_ACQUIRE);
    p->words[48 + 8 * b] = 0;
    if (verdict & gab::kLinkErrTorn) {
        gab::set_last_error(std::string(who) + ": a word the round trip's kernel consumed while the upload was still running is not the word the completed "
                            "upload left in the staging buffer (an engine write that landed in pieces or out of order): the output of THAT round trip was wrong");
        return GAB_ERR_RUNTIME;
    }
    return GAB_OK;
}

int gab_datatransfer_round_trip_check(gab_link_plan* p) {
    return gab::guarded([&]() -> int {
        if (!p) return gab::bad_arg("gab_datatransfer_round_trip_check: null plan");
        const int r0 = gab_link_finish_check(p, 0, true, "gab_datatransfer_round_trip_check");
        const int r1 = gab_link_finish_check(p, 1, true, "gab_datatransfer_round_trip_check");
        return r0 ? r0 : r1;
    });
}

int gab_datatransfer_round_trip(gab_link_plan* p, const float* h_in, float* h_out, int in_size, int out_size,
                                gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!p) return gab::bad_arg("gab_datatransfer_round_trip: null plan");
        // The staging buffers take turns (k_conv_accel.hip, gab_conv_round_trip): this call's buffer must have been re-armed by
        // the check launch queued two calls ago — waited for; the previous call's check is looked at without waiting.
        const int buf = (int)((p->epoch + 1) & 1u);
        {
            const int ra = gab_link_finish_check(p, buf, true, "gab_datatransfer_round_trip (an earlier call)");
            const int rb = gab_link_finish_check(p, buf ^ 1, false, "gab_datatransfer_round_trip (the previous call)");
            if (ra || rb) return ra ? ra : rb;
        }
        unsigned* const stage = p->stage + (size_t)buf * p->side_words;
        unsigned* const consumed = p->consumed + (size_t)buf * p->side_words;
        int& stale_lo = p->stale_lo[buf];
        int& stale_hi = p->stale_hi[buf];
        if (in_size < 0 || out_size < 0) return gab::bad_arg("gab_datatransfer_round_trip: negative size");
        if (in_size > p->max_in) return gab::bad_arg("gab_datatransfer_round_trip: in_size exceeds the plan's max_in_size");
        if ((!h_in && in_size) || (!h_out && out_size)) return gab::bad_arg("gab_datatransfer_round_trip: null pointer");
        hipStream_t s = gab::as_stream(stream);
        if (out_size && p->checked_out != h_out) {     // the kernel writes h_out itself: it must be mapped into the device
            hipPointerAttribute_t at;
            if (hipPointerGetAttributes(&at, h_out) != hipSuccess || at.devicePointer == nullptr) {
                (void)hipGetLastError();
                return gab::bad_arg("gab_datatransfer_round_trip: h_out must be pinned host memory (hipHostMalloc) or device memory");
            }
            p->checked_out = h_out;
        }
        const int dep = std::min(in_size, out_size);
        if (stale_hi > stale_lo && dep > stale_lo) {
            // this call reads words an earlier one left unconsumed: the sentinel goes back, ahead of the upload on its
            // stream, and the kernel's stream waits for it (steady repeats of one shape never come here)
            GAB_HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(stage + stale_lo), (int)gab::kLinkSentinel,
                                            (size_t)(stale_hi - stale_lo), p->copy_stream));
            GAB_HIP_CHECK(hipEventRecord(p->copy_ev, p->copy_stream));
            GAB_HIP_CHECK(hipStreamWaitEvent(s, p->copy_ev, 0));
            stale_lo = stale_hi = 0;
        }
        bool upload = in_size > 0;
#ifdef GAB_ABLATE
        if (getenv("GAB_RT_SKIP_UPLOAD")) upload = false;   // diagnostic builds: the input never lands — every wait must run out
#endif
        // The kernel takes a word the moment it is no longer the sentinel and puts the sentinel back: only right if the
        // upload writes every word exactly ONCE — one engine copy from pinned (or device) memory does; what the runtime does
        // with pageable memory is its own business: such an input is uploaded completely before the launch.
        bool streamed = true;
        if (upload) {
            hipPointerAttribute_t at;
            if (hipPointerGetAttributes(&at, h_in) != hipSuccess || at.devicePointer == nullptr) {
                (void)hipGetLastError();
                streamed = false;
            }
            // in pieces below the runtime's engine-packet limit (4 MiB - 1 BYTES), each a multiple of four bytes: the word that
            // straddles a packet boundary of ONE big copy lands in two pieces, and the kernel takes words as they stop being
            // the sentinel (k_conv_accel.hip, kRtUploadPiece; profiles/r05_incident_torn_word.txt)
            {
                const size_t bytes = sizeof(float) * (size_t)in_size, piece = (size_t(4) << 20) - 256;
                const size_t watched = sizeof(float) * (size_t)dep;     // the kernel only looks at the first `dep` words: what lies beyond goes up in one copy
                for (size_t off = 0; off < bytes;) {
                    const size_t n = off >= watched ? bytes - off : std::min(piece, bytes - off);
                    GAB_HIP_CHECK(hipMemcpyAsync(reinterpret_cast<char*>(stage) + off, reinterpret_cast<const char*>(h_in) + off, n,
                                                 hipMemcpyHostToDevice, p->copy_stream));
                    off += n;
                }
            }
            if (!streamed) GAB_HIP_CHECK(hipStreamSynchronize(p->copy_stream));
        }
        if (in_size > out_size) {
            stale_lo = stale_hi > stale_lo ? std::min(stale_lo, out_size) : out_size;
            stale_hi = std::max(stale_hi, in_size);
        }
        if (out_size == 0) {                            // nothing comes back: the call is the upload
            GAB_HIP_CHECK(hipStreamSynchronize(p->copy_stream));
            return GAB_OK;
        }
        const unsigned epoch = p->epoch + 1;            // moves only when a launch has really been made (the device counter runs on)
        volatile unsigned* const done = p->words;
        unsigned* const landed = p->words + 16;
        volatile unsigned* const error = p->words + 32;
        gab::LinkRoundTrip rt{stage, h_out, p->counter, p->words, p->words + 16, p->words + 32, consumed, epoch, in_size, out_size};
        // the launch carries its own stop event: what the call returns on (k_conv_accel.hip, kRtCompletion: the cheapest of
        // the stated ways to learn that a launch has ended, profiles/r05_roundtrip_completion.txt)
        hipExtLaunchKernelGGL(gab::datatransfer_round_trip_kernel, dim3(p->workgroups), dim3(gab::kBlock), 0, s, nullptr, p->done_ev, 0, rt);
        int rc = gab::launch_status("datatransfer_round_trip_kernel");
        // after a wait that ran out (or a launch that was not made), words may sit in the stage without a consumer: the
        // launch bounds its own waits, so let it end, then start the next call from an all-sentinel stage
        auto repoison = [&]() {
            (void)hipStreamSynchronize(s);
            (void)hipStreamSynchronize(p->copy_stream);
            (void)hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(p->stage), (int)gab::kLinkSentinel, 2 * p->side_words);
            p->words[48] = p->words[56] = 0;               // (check launches over words that never landed say nothing)
            p->check_pending[0] = p->check_pending[1] = false;
            (void)hipDeviceSynchronize();
            p->stale_lo[0] = p->stale_hi[0] = p->stale_lo[1] = p->stale_hi[1] = 0;
        };
        if (rc) {
            if (upload) repoison();
            return rc;
        }
        p->epoch = epoch;
        if (upload && streamed) GAB_HIP_CHECK(hipEventRecord(p->copy_ev, p->copy_stream));
        // The check launch: queued behind the main launch on its stream by the HOST, once the host has seen the upload's completion
        // event (k_conv_accel.hip, gab_conv_round_trip: no wait for that event is ever put into a stream)
        auto queue_check = [&]() {
            if (!(upload && dep > 0)) return;
            gab::datatransfer_round_trip_check_kernel<<<dim3(std::min(256, (dep + gab::kBlock - 1) / gab::kBlock)), dim3(gab::kBlock), 0, s>>>(
                stage, consumed, p->words + 48 + 8 * buf, dep);
            if (gab::launch_status("datatransfer_round_trip_check_kernel")) throw std::runtime_error(gab::last_error());
            GAB_HIP_CHECK(hipEventRecord(p->check_ev[buf], s));
            p->check_pending[buf] = true;
        };
        if (upload && !streamed) queue_check();            // (an upload that was complete before the launch)
        // The upload's event releases workgroups whose words really hold the sentinel (`landed`: a release store the kernel
        // acquires); the hint word says when the launch is about to end; the call returns when the launch HAS ended
        // (its stop event) and the upload is through (an input longer than the output is still landing when the last
        // output has left) — the same completion rule as gab_conv_round_trip, see k_conv_accel.hip.
        bool told = !upload || !streamed;               // nothing to announce (diagnostic: nothing was uploaded, nothing is said)
        if (in_size == 0 || (upload && !streamed)) __atomic_store_n(landed, epoch, __ATOMIC_RELEASE);
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        bool ended = false;
        while (*done != epoch || !told) {
            if (!told && hipEventQuery(p->copy_ev) == hipSuccess) { __atomic_store_n(landed, epoch, __ATOMIC_RELEASE); told = true; queue_check(); }
            if ((++spins & 1023u) == 0) {
                if (told && hipStreamQuery(s) == hipSuccess) { ended = true; break; }
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 4.0) {
                    repoison();
                    gab::set_last_error("gab_datatransfer_round_trip: the launch did not end within 4 s; the output of this call is invalid");
                    return GAB_ERR_RUNTIME;
                }
            }
        }
        (void)hipGetLastError();
        if (ended) {
            GAB_HIP_CHECK(hipStreamSynchronize(s));
        } else {
            for (spins = 0;;) {
                const hipError_t q = hipEventQuery(p->done_ev);
                if (q == hipSuccess) break;
                (void)hipGetLastError();
                if (q != hipErrorNotReady) GAB_HIP_CHECK(q);
                if ((++spins & 1023u) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 4.0) {
                    repoison();
                    gab::set_last_error("gab_datatransfer_round_trip: the launch did not end within 4 s; the output of this call is invalid");
                    return GAB_ERR_RUNTIME;
                }
            }
        }
        if (*error != 0 || *done != epoch) {
            *error = 0;
            repoison();
            gab::set_last_error("gab_datatransfer_round_trip: a workgroup waited about a second for its input and gave up; the output of this call is invalid");
            return GAB_ERR_RUNTIME;
        }
        return GAB_OK;
    });
}

// ---- keep-warm ---------------------------------------------------------------------------------------------------
struct gab_keep_warm {
    hipStream_t stream = nullptr;      // its own, highest priority: a resident launch holds its hardware queue, and streams
                                       // of the default priority never share one with it (profiles/r05_incident_engine_queue_sharing.txt)
    unsigned* words = nullptr;         // pinned host: [0] the kick count / stop word, [16] set by the launch as it ends,
                                       // [32 + 2 b], [33 + 2 b] where workgroup b's wave landed (HW_ID, XCC_ID | 1 << 31)
    unsigned count = 0;
    int workgroups = 1;
    int device = 0;
    int naps = 16;                     // a look every ~64 us: the looks cross the link the round trip uses (64 waves looking every 4 us cost it 7 us)
    double idle_seconds = 0.25;
    bool launched = false;
    bool running() const { return launched && __atomic_load_n(&words[16], __ATOMIC_ACQUIRE) == 0; }
    ~gab_keep_warm() {
        gab::resident_remove(this);
        if (words && launched) __atomic_store_n(&words[0], gab::kKeepWarmStop, __ATOMIC_RELEASE);
        if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
        if (words) (void)hipHostFree(words);
    }
};

int gab_keep_warm_create(gab_keep_warm** out, int workgroups, double idle_seconds) {
    return gab::guarded([&]() -> int {
        if (!out) return gab::bad_arg("gab_keep_warm_create: null argument");
        if (workgroups < 1 || workgroups > 256) return gab::bad_arg("gab_keep_warm_create: 1..256 workgroups (one wave each)");
        if (!(idle_seconds > 0.0 && idle_seconds <= 10.0)) return gab::bad_arg("gab_keep_warm_create: idle_seconds must be in (0, 10]");
        auto k = std::make_unique<gab_keep_warm>();
        k->workgroups = workgroups;
        k->idle_seconds = idle_seconds;
#ifdef GAB_ABLATE
        if (getenv("GAB_KEEP_WARM_NAPS")) k->naps = std::max(1, atoi(getenv("GAB_KEEP_WARM_NAPS")));   // diagnostic builds: how often the waves look
#endif
        int lo = 0, hi = 0;
        GAB_HIP_CHECK(hipGetDevice(&k->device));
        GAB_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        GAB_HIP_CHECK(hipStreamCreateWithPriority(&k->stream, hipStreamNonBlocking, hi));
        const int words = gab::kKeepWarmWords + 2 * workgroups;
        GAB_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&k->words), words * sizeof(unsigned), hipHostMallocDefault));
        for (int i = 0; i < words; ++i) k->words[i] = 0;
        gab::resident_add(k.get(), k->device, gab::kResidentKeepWarm,
                          [](const void* o) { return static_cast<const gab_keep_warm*>(o)->running(); });
        *out = k.release();
        return GAB_OK;
    });
}

int gab_keep_warm_kick(gab_keep_warm* k) {
    return gab::guarded([&]() -> int {
        if (!k) return gab::bad_arg("gab_keep_warm_kick: null argument");
        if (++k->count == gab::kKeepWarmStop) k->count = 1;
        if (k->running()) {
            __atomic_store_n(&k->words[0], k->count, __ATOMIC_RELEASE);          // the launch is there: push its end out
            return GAB_OK;
        }
        // A doorbell-fed engine on this device holds every compute unit's registers until its stop: the launch could only
        // queue up behind it (and would then keep the NEXT engine start out).  The engine keeps the device awake itself.
        if (gab::resident_running(k->device, gab::kResidentEngine, nullptr) > 0)
            return gab::bad_arg("gab_keep_warm_kick: a gab_conv_engine launch is resident on this device and fills its compute units; "
                                "the keep-warm launch cannot start beside it (and is not needed: the engine keeps the device awake) — "
                                "gab_conv_engine_stop first");
        if (k->launched) {
            // workgroup 0 said the launch was ending: the stop word makes every straggler leave at its next look
            __atomic_store_n(&k->words[0], gab::kKeepWarmStop, __ATOMIC_RELEASE);
            GAB_HIP_CHECK(hipStreamSynchronize(k->stream));
        }
        for (int i = 0; i < 2 * k->workgroups; ++i) k->words[gab::kKeepWarmWords + i] = 0;
        __atomic_store_n(&k->words[0], k->count, __ATOMIC_RELEASE);
        __atomic_store_n(&k->words[16], 0u, __ATOMIC_RELEASE);
        gab::keep_warm_kernel<<<dim3(k->workgroups), dim3(64), 0, k->stream>>>(k->words, k->words + 16, k->words + gab::kKeepWarmWords,
                                                                               (unsigned long long)(k->idle_seconds * 1e8), k->naps);
        int rc = gab::launch_status("keep_warm_kernel");
        if (rc) return rc;
        k->launched = true;
        return GAB_OK;
    });
}

int gab_keep_warm_running(gab_keep_warm* k, int* running) {
    if (!k || !running) return gab::bad_arg("gab_keep_warm_running: null argument");
    *running = k->running() ? 1 : 0;
    return GAB_OK;
}

int gab_keep_warm_placement(gab_keep_warm* k, unsigned* hw_id, unsigned* xcc_id, int capacity, int* started) {
    if (!k || !started) return gab::bad_arg("gab_keep_warm_placement: null argument");
    if (capacity < 0 || (capacity > 0 && (!hw_id || !xcc_id))) return gab::bad_arg("gab_keep_warm_placement: capacity without arrays");
    int n = 0;
    for (int b = 0; b < k->workgroups; ++b) {
        const unsigned x = __atomic_load_n(&k->words[gab::kKeepWarmWords + 2 * b + 1], __ATOMIC_ACQUIRE);
        if (!(x >> 31)) continue;                                                // this wave has not started (yet)
        if (n < capacity) {
            hw_id[n] = __atomic_load_n(&k->words[gab::kKeepWarmWords + 2 * b], __ATOMIC_RELAXED);
            xcc_id[n] = x & 0x7fffffffu;
        }
        ++n;
    }
    *started = n;
    return GAB_OK;
}

int gab_keep_warm_destroy(gab_keep_warm* k) {
    delete k;
    return GAB_OK;
}

int gab_modal(const float* d_params, float* d_out, int n_modes, int bufsize, int out_tracks,
              gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!d_params || !d_out) return gab::bad_arg("gab_modal: null pointer");
        if (n_modes <= 0 || bufsize <= 0 || out_tracks <= 0) return gab::bad_arg("gab_modal: sizes must be > 0");
        int rows = n_modes < out_tracks ? n_modes : out_tracks;
        gab::modal_placeholder_kernel<<<rows, gab::kBlock, 0, gab::as_stream(stream)>>>(
            d_params, d_out, n_modes, bufsize);
        return gab::launch_status("modal_placeholder_kernel");
    });
}

int gab_rndmem(const float* d_pool, const int* d_playheads, float* d_out, int tracks, int bufsize,
               gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!d_pool || !d_playheads || !d_out) return gab::bad_arg("gab_rndmem: null pointer");
        if (tracks <= 0 || bufsize <= 0) return gab::bad_arg("gab_rndmem: tracks and bufsize must be > 0");
        dim3 grid(((bufsize + 63) / 64) * ((tracks + 63) / 64));
        gab::rndmem_kernel<<<grid, gab::kBlock, 0, gab::as_stream(stream)>>>(d_pool, d_playheads, d_out,
                                                                               tracks, bufsize);
        return gab::launch_status("rndmem_kernel");
    });
}

}  // extern "C"
