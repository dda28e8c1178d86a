// far_transform — the far role's 4096-point transform chain of conv_split_batch_kernel, two ways:
//
//   A  today's form: three radix-16 Stockham passes through padded LDS on four waves (gab::fft::BlockFFT<4096, 16>), partner
//      exchange, spectral product, inverse pruned to its last quarter — SIX workgroup barriers per period;
//   B  the form with ONE cross-wave exchange per direction: a lane-local radix-4 pass over elements 1024 apart (the first layer
//      of the radix-16 butterfly) times W4096^(n1 q), one exchange across the four waves, then four WAVE-HELD 1024-point
//      transforms (gab::fft::WaveFFT1024: their two exchanges are private to the wave, no workgroup barrier); bins k = 4 m + q live
//      in wave q, so partners N - k sit in waves 0<->0, 1<->3, 2<->2; the inverse mirrors it and keeps the last 1024 samples
//      (n2 = 3: three adds per output) — FOUR workgroup barriers per period.
//
// Both run on waves 4-7 of a 512-thread workgroup, one workgroup per compute unit (256 of them), `periods` periods back to back on
// data that stays on chip; waves 0-3 either only meet the barriers ("alone") or run a wave-held forward + inverse 1024-point
// transform per period beside them, arriving at the same barriers ("beside": the near role's share of each SIMD).  B's outputs
// are checked against A's before anything is timed.
//
//   tools/ubench/bin/far_transform [periods]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gab_fft.hpp"

using gab::fft::cf;
using gab::fft::mk;
namespace fft = gab::fft;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kN = 4096, kFar = 256, kWG = 512;
constexpr int kHalf = fft::Pad<16>::size(kN);          // 4352 entries: one padded 4096-point image = four wave images of 1088
constexpr int kImg = fft::Pad<16>::size(1024);         // 1088

// the "spectra": any function of the bin index will do (both forms evaluate it at THEIR bin)
__device__ __forceinline__ cf specP(int k) { return mk(0.9f + 1e-4f * (float)(k & 255), 1e-3f * (float)((k >> 3) & 63)); }
__device__ __forceinline__ cf specM(int k) { return mk(2e-3f * (float)(k & 31), 0.05f - 1e-4f * (float)(k & 127)); }

struct Arrive { __device__ __forceinline__ void operator()(int) const { __syncthreads(); } };

template <int FORM, bool BESIDE>
__global__ __launch_bounds__(kWG) void far_kernel(const cf* __restrict__ tw, const cf* __restrict__ in, cf* __restrict__ out, int periods) {
    __shared__ cf lds[2 * kHalf + 4 * kImg];
    cf* const bufA = lds;
    cf* const bufB = lds + kHalf;
    cf* const near_img = lds + 2 * kHalf;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int BARRIERS = FORM == 0 ? 6 : 4;
    if (w < 4) {
        // ---- the near role's stand-in
        cf v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = mk(1e-3f * (float)(lane + 64 * r), 0.f);
        fft::WaveFFT1024<false>::Lean tf;
        fft::WaveFFT1024<false>::load_twiddles(tf, tw, lane);
        for (int p = 0; p < periods; ++p) {
            if (BESIDE) {
                fft::WaveFFT1024<false>::run(v, near_img + w * kImg, tf, lane, Arrive());        // barriers 1, 2 from inside
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] *= (1.0f / 1024.0f);                          // (stays finite; the inverse waves' transform costs the same)
                for (int i = 2; i < BARRIERS; ++i) __syncthreads();
            } else {
                for (int i = 0; i < BARRIERS; ++i) __syncthreads();
            }
        }
        if (v[0].x == 123.456f) out[0] = v[3];           // keep the work alive
        return;
    }
    const int ft = tid - kFar, wv = w - 4, j = lane;
    cf z[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = in[(size_t)blockIdx.x * kN + ft + kFar * r];
    if constexpr (FORM == 0) {
        using FB = fft::BlockFFT<kN, 16, false>;
        using FBi = fft::BlockFFT<kN, 16, true>;
        typename FB::Twiddles twb;
        FB::load_twiddles(twb, tw, ft);
        cf P[16], M[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { P[r] = specP(ft + kFar * r); M[r] = specM(ft + kFar * r); }
        cf keep[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) keep[r] = z[r];
        for (int p = 0; p < periods; ++p) {
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = keep[r];
            FB::run(z, bufA, bufB, twb, ft);                                   // barriers 1, 2
            cf zp[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bufA[ft + r * kFar] = z[r];             // partner exchange, raw (unpadded, as the product kernel's)
            __syncthreads();                                                    // barrier 3
#pragma unroll
            for (int r = 0; r < 16; ++r) zp[r] = bufA[(kN - (ft + r * kFar)) & (kN - 1)];
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = fft::cfma_cj(zp[r], M[r], fft::cmul(z[r], P[r]));
            FBi::template run<typename FB::Twiddles, 4>(z, bufB, bufA, twb, ft);     // barriers 4, 5; only [12..15]
            __syncthreads();                                                    // barrier 6 closes the period
#pragma unroll
            for (int r = 12; r < 16; ++r) keep[r - 12] = keep[r - 12] + 1e-6f * z[r];      // the outputs feed the next period (a chain, as the carry is)
        }
    } else {
        using WF = fft::WaveFFT1024<false>;
        using WFi = fft::WaveFFT1024<true>;
        WF::Lean t;
        WF::load_twiddles(t, tw, j);
        const cf w1 = tw[ft], w2 = tw[(2 * ft) & (kN - 1)], w3 = tw[(3 * ft) & (kN - 1)];      // W4096^(ft q)
        cf P[16], M[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int k = 4 * (j + 64 * i) + wv; P[i] = specP(k); M[i] = specM(k); }
        cf* const E = bufA;                                  // [q][1088]: cross exchange, then wave q's private image
        cf* const PX = bufB;                                 // [q][1024]: the partner exchange
        cf keep[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) keep[r] = z[r];
        for (int p = 0; p < periods; ++p) {
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = keep[r];
            // ---- forward: the radix-4 pass over n2 (elements 1024 apart are v[r'], v[r'+4], v[r'+8], v[r'+12]) and W4096^(n1 q)
#pragma unroll
            for (int rp = 0; rp < 4; ++rp) fft::bfly4<false>(z[rp], z[rp + 4], z[rp + 8], z[rp + 12]);
#pragma unroll
            for (int rp = 0; rp < 4; ++rp) {
                z[rp + 4] = fft::cmul(z[rp + 4], w1);
                z[rp + 8] = fft::cmul(z[rp + 8], w2);
                z[rp + 12] = fft::cmul(z[rp + 12], w3);
            }
            // W16^(r' q): r' q in {1,2,3 | 2,4,6 | 3,6,9}
            z[1 + 4] = fft::tw16<false, 1>(z[1 + 4]); z[1 + 8] = fft::tw16<false, 2>(z[1 + 8]); z[1 + 12] = fft::tw16<false, 3>(z[1 + 12]);
            z[2 + 4] = fft::tw16<false, 2>(z[2 + 4]); z[2 + 8] = fft::tw16<false, 4>(z[2 + 8]); z[2 + 12] = fft::tw16<false, 6>(z[2 + 12]);
            z[3 + 4] = fft::tw16<false, 3>(z[3 + 4]); z[3 + 8] = fft::tw16<false, 6>(z[3 + 8]); z[3 + 12] = fft::tw16<false, 9>(z[3 + 12]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int rp = 0; rp < 4; ++rp) E[q * kImg + ft + kFar * rp] = z[rp + 4 * q];
            __syncthreads();                                                    // barrier 1: the one cross-wave exchange
            cf v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = E[wv * kImg + j + 64 * i];
            WF::run(v, E + wv * kImg, t, j);                                    // wave-held: no workgroup barrier
            // ---- partners: bin k = 4 m + wv; N - k lives in wave (4 - wv) & 3 at m' = 1023 - m (wave 0: (1024 - m) & 1023)
#pragma unroll
            for (int i = 0; i < 16; ++i) PX[wv * 1024 + j + 64 * i] = v[i];
            __syncthreads();                                                    // barrier 2
            const int qp = (4 - wv) & 3;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = j + 64 * i;
                const int mp = wv == 0 ? ((1024 - m) & 1023) : 1023 - m;
                const cf zp = PX[qp * 1024 + mp];
                v[i] = fft::cfma_cj(zp, M[i], fft::cmul(v[i], P[i]));
            }
            WFi::run(v, E + wv * kImg, t, j);                                   // wave-held inverse: u_q[n1], n1 = j + 64 i
#pragma unroll
            for (int i = 0; i < 16; ++i) E[wv * kImg + j + 64 * i] = v[i];       // (the wave's own region: its transform is through)
            __syncthreads();                                                    // barrier 3
            // ---- the last quarter of the output: x[n1 + 3072] = (t0 - t2) - i (t1 - t3), t_q = conj(W4096^(n1 q)) u_q[n1]
#pragma unroll
            for (int rp = 0; rp < 4; ++rp) {
                const int n1 = ft + kFar * rp;
                cf t0 = E[0 * kImg + n1], t1 = E[1 * kImg + n1], t2 = E[2 * kImg + n1], t3 = E[3 * kImg + n1];
                t1 = fft::cmulc(t1, w1); t2 = fft::cmulc(t2, w2); t3 = fft::cmulc(t3, w3);
                if (rp == 1) { t1 = fft::tw16<true, 1>(t1); t2 = fft::tw16<true, 2>(t2); t3 = fft::tw16<true, 3>(t3); }
                if (rp == 2) { t1 = fft::tw16<true, 2>(t1); t2 = fft::tw16<true, 4>(t2); t3 = fft::tw16<true, 6>(t3); }
                if (rp == 3) { t1 = fft::tw16<true, 3>(t1); t2 = fft::tw16<true, 6>(t2); t3 = fft::tw16<true, 9>(t3); }
                z[12 + rp] = fft::addmi(fft::csub(t0, t2), fft::csub(t1, t3));
            }
            __syncthreads();                                                    // barrier 4 closes the period
#pragma unroll
            for (int r = 12; r < 16; ++r) keep[r - 12] = keep[r - 12] + 1e-6f * z[r];
        }
    }
#pragma unroll
    for (int r = 12; r < 16; ++r) out[(size_t)blockIdx.x * 1024 + ft + kFar * (r - 12)] = z[r];
}

// ---- twelve waves: two far groups (form B, 149 VGPRs: three waves per SIMD fit), each owning one pair and taking TWO periods per
// transform, beside four near waves that run TWO wave-held transforms per iteration.  One iteration = two periods of the product kernel.
// The partner exchange reuses the cross exchange's buffer behind one more barrier (two transforms must fit the LDS): five barriers.
template <bool BESIDE>
__global__ __launch_bounds__(768) void far12_kernel(const cf* __restrict__ tw, const cf* __restrict__ in, cf* __restrict__ out, int iters) {
    __shared__ cf lds[2 * kHalf + 4 * kImg];
    cf* const near_img = lds + 2 * kHalf;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int BARRIERS = 5;
    if (w < 4) {
        cf v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = mk(1e-3f * (float)(lane + 64 * r), 0.f);
        fft::WaveFFT1024<false>::Lean tf;
        fft::WaveFFT1024<false>::load_twiddles(tf, tw, lane);
        for (int p = 0; p < iters; ++p) {
            if (BESIDE) {
                fft::WaveFFT1024<false>::run(v, near_img + w * kImg, tf, lane, Arrive());        // barriers 1, 2
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] *= (1.0f / 1024.0f);
                fft::WaveFFT1024<false>::run(v, near_img + w * kImg, tf, lane, Arrive());        // barriers 3, 4 (the second period's near work)
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] *= (1.0f / 1024.0f);
                __syncthreads();                                                               // barrier 5
            } else {
                for (int i = 0; i < BARRIERS; ++i) __syncthreads();
            }
        }
        if (v[0].x == 123.456f) out[0] = v[3];
        return;
    }
    const int grp = (w - 4) >> 2;                            // far group 0 / 1: its own pair, its own LDS half
    const int ft = (tid - 256) & 255, wv = (w - 4) & 3, j = lane;
    cf* const E = lds + grp * kHalf;
    cf z[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = in[((size_t)blockIdx.x * 2 + grp) * kN % ((size_t)gridDim.x * kN) + ft + kFar * r];
    using WF = fft::WaveFFT1024<false>;
    using WFi = fft::WaveFFT1024<true>;
    WF::Lean t;
    WF::load_twiddles(t, tw, j);
    const cf w1 = tw[ft], w2 = tw[(2 * ft) & (kN - 1)], w3 = tw[(3 * ft) & (kN - 1)];
    cf P[16], M[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int k = 4 * (j + 64 * i) + wv; P[i] = specP(k); M[i] = specM(k); }
    cf keep[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) keep[r] = z[r];
    for (int p = 0; p < iters; ++p) {
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = keep[r];
#pragma unroll
        for (int rp = 0; rp < 4; ++rp) fft::bfly4<false>(z[rp], z[rp + 4], z[rp + 8], z[rp + 12]);
#pragma unroll
        for (int rp = 0; rp < 4; ++rp) {
            z[rp + 4] = fft::cmul(z[rp + 4], w1);
            z[rp + 8] = fft::cmul(z[rp + 8], w2);
            z[rp + 12] = fft::cmul(z[rp + 12], w3);
        }
        z[1 + 4] = fft::tw16<false, 1>(z[1 + 4]); z[1 + 8] = fft::tw16<false, 2>(z[1 + 8]); z[1 + 12] = fft::tw16<false, 3>(z[1 + 12]);
        z[2 + 4] = fft::tw16<false, 2>(z[2 + 4]); z[2 + 8] = fft::tw16<false, 4>(z[2 + 8]); z[2 + 12] = fft::tw16<false, 6>(z[2 + 12]);
        z[3 + 4] = fft::tw16<false, 3>(z[3 + 4]); z[3 + 8] = fft::tw16<false, 6>(z[3 + 8]); z[3 + 12] = fft::tw16<false, 9>(z[3 + 12]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int rp = 0; rp < 4; ++rp) E[q * kImg + ft + kFar * rp] = z[rp + 4 * q];
        __syncthreads();                                                    // barrier 1
        cf v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = E[wv * kImg + j + 64 * i];
        WF::run(v, E + wv * kImg, t, j);
        __syncthreads();                                                    // barrier 2: every wave's transform is through, E is free
#pragma unroll
        for (int i = 0; i < 16; ++i) E[wv * kImg + j + 64 * i] = v[i];       // partner exchange in the same buffer
        __syncthreads();                                                    // barrier 3
        const int qp = (4 - wv) & 3;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = j + 64 * i;
            const int mp = wv == 0 ? ((1024 - m) & 1023) : 1023 - m;
            const cf zp = E[qp * kImg + mp];
            v[i] = fft::cfma_cj(zp, M[i], fft::cmul(v[i], P[i]));
        }
        __syncthreads();                                                    // barrier 4: partners read, the images may be written again
        WFi::run(v, E + wv * kImg, t, j);
#pragma unroll
        for (int i = 0; i < 16; ++i) E[wv * kImg + j + 64 * i] = v[i];
        __syncthreads();                                                    // barrier 5
#pragma unroll
        for (int rp = 0; rp < 4; ++rp) {
            const int n1 = ft + kFar * rp;
            cf t0 = E[0 * kImg + n1], t1 = E[1 * kImg + n1], t2 = E[2 * kImg + n1], t3 = E[3 * kImg + n1];
            t1 = fft::cmulc(t1, w1); t2 = fft::cmulc(t2, w2); t3 = fft::cmulc(t3, w3);
            if (rp == 1) { t1 = fft::tw16<true, 1>(t1); t2 = fft::tw16<true, 2>(t2); t3 = fft::tw16<true, 3>(t3); }
            if (rp == 2) { t1 = fft::tw16<true, 2>(t1); t2 = fft::tw16<true, 4>(t2); t3 = fft::tw16<true, 6>(t3); }
            if (rp == 3) { t1 = fft::tw16<true, 3>(t1); t2 = fft::tw16<true, 6>(t2); t3 = fft::tw16<true, 9>(t3); }
            z[12 + rp] = fft::addmi(fft::csub(t0, t2), fft::csub(t1, t3));
        }
#pragma unroll
        for (int r = 12; r < 16; ++r) keep[r - 12] = keep[r - 12] + 1e-6f * z[r];
    }
#pragma unroll
    for (int r = 12; r < 16; ++r) out[(size_t)blockIdx.x * 1024 + ft + kFar * (r - 12)] = z[r];
}

template <bool BESIDE>
static double run12(const cf* tw, const cf* in, cf* out, int iters, int grid) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    far12_kernel<BESIDE><<<grid, 768>>>(tw, in, out, 50);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    far12_kernel<BESIDE><<<grid, 768>>>(tw, in, out, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / iters / 2.0;                       // per PERIOD: an iteration is two
}

template <int FORM, bool BESIDE>
static double run(const cf* tw, const cf* in, cf* out, int periods, int grid) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    far_kernel<FORM, BESIDE><<<grid, kWG>>>(tw, in, out, 50);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    far_kernel<FORM, BESIDE><<<grid, kWG>>>(tw, in, out, periods);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / periods;
}

int main(int argc, char** argv) {
    const int periods = argc > 1 ? atoi(argv[1]) : 4000;
    const int grid = 256;
    std::vector<float> twh(2 * fft::kTwiddleN);
    for (int m = 0; m < fft::kTwiddleN; ++m) {
        const double a = -2.0 * M_PI * (double)m / (double)fft::kTwiddleN;
        twh[2 * m] = (float)std::cos(a);
        twh[2 * m + 1] = (float)std::sin(a);
    }
    std::vector<float> inh((size_t)grid * kN * 2);
    unsigned s = 12345u;
    for (auto& v : inh) { s = s * 1664525u + 1013904223u; v = (float)((int)(s >> 9) - (1 << 22)) / (float)(1 << 22); }
    cf *tw, *in, *outA, *outB;
    CK(hipMalloc(&tw, twh.size() * 4));
    CK(hipMalloc(&in, inh.size() * 4));
    CK(hipMalloc(&outA, (size_t)grid * 1024 * 8));
    CK(hipMalloc(&outB, (size_t)grid * 1024 * 8));
    CK(hipMemcpy(tw, twh.data(), twh.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(in, inh.data(), inh.size() * 4, hipMemcpyHostToDevice));
    // ---- B against A, one period
    far_kernel<0, false><<<grid, kWG>>>(tw, in, outA, 1);
    far_kernel<1, false><<<grid, kWG>>>(tw, in, outB, 1);
    CK(hipDeviceSynchronize());
    std::vector<float> a((size_t)grid * 2048), b((size_t)grid * 2048);
    CK(hipMemcpy(a.data(), outA, a.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), outB, b.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, peak = 0;
    for (size_t i = 0; i < a.size(); ++i) { worst = std::fmax(worst, std::fabs((double)a[i] - b[i])); peak = std::fmax(peak, std::fabs((double)a[i])); }
    printf("{\"check\": \"form B against form A, last 1024 outputs of 256 transforms\", \"max_abs_diff\": %.3e, \"peak\": %.3e, \"rel\": %.2e, \"ok\": %s}\n",
           worst, peak, worst / peak, worst / peak <= 1e-5 ? "true" : "false");
    if (!(worst / peak <= 1e-5)) return 1;
    const double a_alone = run<0, false>(tw, in, outA, periods, grid), b_alone = run<1, false>(tw, in, outB, periods, grid);
    const double a_beside = run<0, true>(tw, in, outA, periods, grid), b_beside = run<1, true>(tw, in, outB, periods, grid);
    const double a_alone2 = run<0, false>(tw, in, outA, periods, grid), b_alone2 = run<1, false>(tw, in, outB, periods, grid);
    const double a_beside2 = run<0, true>(tw, in, outA, periods, grid), b_beside2 = run<1, true>(tw, in, outB, periods, grid);
    printf("{\"us_per_period\": {\"A_three_radix16_passes_6_barriers\": {\"alone\": [%.3f, %.3f], \"beside_near_waves\": [%.3f, %.3f]}, "
           "\"B_radix4_plus_wave_held_1024_4_barriers\": {\"alone\": [%.3f, %.3f], \"beside_near_waves\": [%.3f, %.3f]}}, \"periods\": %d, \"workgroups\": %d}\n",
           a_alone, a_alone2, a_beside, a_beside2, b_alone, b_alone2, b_beside, b_beside2, periods, grid);
    const double c_alone = run12<false>(tw, in, outB, periods / 2, grid), c_beside = run12<true>(tw, in, outB, periods / 2, grid);
    const double c_alone2 = run12<false>(tw, in, outB, periods / 2, grid), c_beside2 = run12<true>(tw, in, outB, periods / 2, grid);
    printf("{\"us_per_period\": {\"C_twelve_waves_two_far_groups_of_form_B_two_periods_per_transform\": {\"alone\": [%.3f, %.3f], \"beside_near_waves\": [%.3f, %.3f]}}, "
           "\"note\": \"one iteration = two periods (each far group turns one transform, the near waves two); 768 threads, three waves per SIMD\"}\n",
           c_alone, c_alone2, c_beside, c_beside2);
    return 0;
}
