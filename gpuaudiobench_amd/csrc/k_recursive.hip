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

__global__ __launch_bounds__(256) void iir_biquad_kernel(const float* __restrict__ in,
                                                        float* __restrict__ out,
                                                        float* __restrict__ state,
                                                        BiquadCoeffs c, int T, int B) {
    __shared__ float tile[kIirTracks][kIirChunk + 1];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int t0 = blockIdx.x * kIirTracks;
    const int my_track = t0 + lane;
    float z1 = 0.0f, z2 = 0.0f;
    if (w == 0 && my_track < T) {
        z1 = state[2 * my_track];
        z2 = state[2 * my_track + 1];
    }
    for (int s0 = 0; s0 < B; s0 += kIirChunk) {
        // stage: each wave loads rows (tracks), lanes run along samples
        for (int r = w; r < kIirTracks; r += 4) {
            int t = t0 + r, s = s0 + lane;
            if (t < T && s < B) tile[r][lane] = in[(size_t)t * B + s];
        }
        __syncthreads();
        if (w == 0 && my_track < T) {
            const int n = (B - s0) < kIirChunk ? (B - s0) : kIirChunk;
            for (int i = 0; i < n; ++i) {
                float x = tile[lane][i];
                float wv = __fsub_rn(__fsub_rn(x, __fmul_rn(c.a1, z1)), __fmul_rn(c.a2, z2));
                float y = __fadd_rn(__fadd_rn(__fmul_rn(c.b0, wv), __fmul_rn(c.b1, z1)),
                                    __fmul_rn(c.b2, z2));
                z2 = z1;
                z1 = wv;
                tile[lane][i] = y;
            }
        }
        __syncthreads();
        for (int r = w; r < kIirTracks; r += 4) {
            int t = t0 + r, s = s0 + lane;
            if (t < T && s < B) out[(size_t)t * B + s] = tile[r][lane];
        }
        __syncthreads();
    }
    if (w == 0 && my_track < T) {
        state[2 * my_track] = z1;
        state[2 * my_track + 1] = z2;
    }
}

// ---------------------------------------------------------------------------
// conv1d: y[t*B+i] = sum_j h[t*L+j] * xflat[t*B+i-j], j ascending, with the
// golden's range test on the FLAT index (cuda/bench_conv1d.cu:188-208).  One
// workgroup = 256 consecutive outputs of one track; taps and the input window
// are staged in LDS in chunks of kTapChunk taps.
// ---------------------------------------------------------------------------
constexpr int kTapChunk = 1024;
constexpr int kConvTile = 256;

__global__ __launch_bounds__(kConvTile) void conv1d_direct_kernel(const float* __restrict__ in,
                                                                 float* __restrict__ out,
                                                                 const float* __restrict__ ir,
                                                                 int L, int T, int B) {
    __shared__ float taps[kTapChunk];
    __shared__ float win[kTapChunk + kConvTile];
    const int t = blockIdx.y;
    const int i0 = blockIdx.x * kConvTile;
    const int tid = threadIdx.x;
    const long flat0 = (long)t * B + i0;           // flat index of this tile's first output
    const long total = (long)T * B;
    float acc = 0.0f;
    for (int j0 = 0; j0 < L; j0 += kTapChunk) {
        const int nj = (L - j0) < kTapChunk ? (L - j0) : kTapChunk;
        // window element m holds xflat[flat0 - j0 - (nj-1) + m], m in [0, nj-1+tile)
        const long wbase = flat0 - j0 - (nj - 1);
        for (int m = tid; m < nj; m += kConvTile) taps[m] = ir[(size_t)t * L + j0 + m];
        for (int m = tid; m < nj - 1 + kConvTile; m += kConvTile) {
            long g = wbase + m;
            win[m] = (g >= 0 && g < total) ? in[g] : 0.0f;
        }
        __syncthreads();
        // output i0+tid at tap j0+jj reads flat index flat0+tid-j0-jj = wbase + (nj-1) + tid - jj
        const long first_valid = flat0 + tid - j0;   // flat index at jj = 0
        const int top = nj - 1 + tid;
        for (int jj = 0; jj < nj; ++jj) {
            if (first_valid - jj >= 0)               // golden skips taps that fall before sample 0
                acc = __fadd_rn(acc, __fmul_rn(taps[jj], win[top - jj]));
        }
        __syncthreads();
    }
    if (i0 + tid < B) out[(size_t)t * B + i0 + tid] = acc;
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

__global__ __launch_bounds__(64) void dwg_naive_kernel(const WG* __restrict__ wgs,
                                                      float* __restrict__ fwd, float* __restrict__ bwd,
                                                      const float* __restrict__ input,
                                                      float* __restrict__ ws, int n_wg, int B,
                                                      int max_len) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_wg) return;
    const WG wg = wgs[g];
    float* F = fwd + (size_t)g * max_len;
    float* Bk = bwd + (size_t)g * max_len;
    for (int s = 0; s < B; ++s) {
        int cur = (wg.writePos + s) % wg.length;
        int bp = (cur + wg.length / 2) % wg.length;
        float f = F[cur], b = Bk[bp], mix;
        float x = __fmul_rn(input[s], wg.gain);
        dwg_step(f, b, x, cur == wg.inTap, wg, mix);
        F[cur] = f;
        Bk[bp] = b;
        if (cur == wg.outTap) ws[(size_t)g * B + s] = mix;
    }
}

__global__ __launch_bounds__(256) void dwg_cells_kernel(const WG* __restrict__ wgs,
                                                       float* __restrict__ fwd, float* __restrict__ bwd,
                                                       const float* __restrict__ input,
                                                       float* __restrict__ ws, int n_wg, int B,
                                                       int max_len) {
    __shared__ float xin[2048];                       // staged input (B <= 2048), else global
    const int g = blockIdx.y;
    const WG wg = wgs[g];
    const bool staged = B <= 2048;
    if (staged) {
        for (int i = threadIdx.x; i < B; i += blockDim.x) xin[i] = input[i];
        __syncthreads();
    }
    const int p = blockIdx.x * blockDim.x + threadIdx.x;       // delay-line cell
    if (p >= wg.length) return;
    // first sample that lands on cell p: (writePos + s) % L == p
    int s = p - (wg.writePos % wg.length);
    if (s < 0) s += wg.length;
    if (s >= B) return;
    const int bp = (p + wg.length / 2) % wg.length;
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

__global__ __launch_bounds__(256) void dwg_mix_kernel(const WG* __restrict__ wgs,
                                                     const float* __restrict__ ws,
                                                     float* __restrict__ out, int n_wg, int B,
                                                     int out_tracks) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B) return;
    float acc = 0.0f;
    const int n = n_wg < out_tracks ? n_wg : out_tracks;
    for (int g = 0; g < n; ++g) {
        const int L = wgs[g].length;
        if ((wgs[g].writePos + s) % L == wgs[g].outTap) acc = __fadd_rn(acc, ws[(size_t)g * B + s]);
    }
    out[s] = acc;
}

}  // namespace
}  // namespace gab

extern "C" {

int gab_iir(const float* d_in, float* d_out, const float* coeffs, float* d_state, int tracks,
            int bufsize, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!d_in || !d_out || !coeffs || !d_state) return gab::bad_arg("gab_iir: null pointer");
        if (tracks <= 0 || bufsize <= 0) return gab::bad_arg("gab_iir: tracks and bufsize must be > 0");
        gab::BiquadCoeffs c{coeffs[0], coeffs[1], coeffs[2], coeffs[3], coeffs[4]};
        dim3 grid((tracks + gab::kIirTracks - 1) / gab::kIirTracks);
        gab::iir_biquad_kernel<<<grid, 256, 0, gab::as_stream(stream)>>>(d_in, d_out, d_state, c,
                                                                           tracks, bufsize);
        return gab::launch_status("iir_biquad_kernel");
    });
}

int gab_conv1d(const float* d_in, float* d_out, const float* d_ir, int ir_len, int tracks,
               int bufsize, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!d_in || !d_out || !d_ir) return gab::bad_arg("gab_conv1d: null pointer");
        if (ir_len <= 0 || tracks <= 0 || bufsize <= 0) return gab::bad_arg("gab_conv1d: sizes must be > 0");
        dim3 grid((bufsize + gab::kConvTile - 1) / gab::kConvTile, tracks);
        gab::conv1d_direct_kernel<<<grid, gab::kConvTile, 0, gab::as_stream(stream)>>>(
            d_in, d_out, d_ir, ir_len, tracks, bufsize);
        return gab::launch_status("conv1d_direct_kernel");
    });
}

size_t gab_dwg_workspace_bytes(int n_waveguides, int bufsize) {
    if (n_waveguides <= 0 || bufsize <= 0) return 0;
    return sizeof(float) * (size_t)n_waveguides * (size_t)bufsize;
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
        if (variant == GAB_DWG_NAIVE) {
            gab::dwg_naive_kernel<<<(n_waveguides + 63) / 64, 64, 0, s>>>(wgs, d_fwd, d_bwd, d_in, ws,
                                                                         n_waveguides, bufsize, max_len);
        } else {
            int cells = max_len < bufsize ? max_len : bufsize;   // a cell >= B is never visited
            dim3 grid((cells + 255) / 256, n_waveguides);
            gab::dwg_cells_kernel<<<grid, 256, 0, s>>>(wgs, d_fwd, d_bwd, d_in, ws, n_waveguides,
                                                       bufsize, max_len);
        }
        int rc = gab::launch_status("dwg kernel");
        if (rc) return rc;
        gab::dwg_mix_kernel<<<(bufsize + 255) / 256, 256, 0, s>>>(wgs, ws, d_out, n_waveguides, bufsize,
                                                                  out_tracks);
        return gab::launch_status("dwg_mix_kernel");
    });
}

}  // extern "C"
