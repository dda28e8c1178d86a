// mfma_dft — is the matrix pipe a faster home for a radix-16 FFT pass than the vector butterfly?
//
// One radix-16 DFT of complex fp32 data, two ways, timed per DFT on one wave per SIMD:
//   VALU : gab::fft::Butterfly<16> (the packed-fp32 butterfly the convolution kernels use),
//          one DFT per lane -> 64 DFTs per wave and iteration;
//   MFMA : the same DFT as a dense real product  [Re; Im](32 x 32 columns) = M(32 x 32) x [Re; Im],
//          M = [[C, S], [-S, C]], with v_mfma_f32_32x32x2_f32 (exact fp32; gfx950 has no xf32):
//          16 instructions (K = 32) -> 32 DFTs per wave and iteration.  Operands stay in registers
//          and the result is fed straight back as the next input (no layout conversion between
//          passes, no twiddles, no LDS): the matrix pipe's best case.
// Both are checked against a float64 DFT on the host before they are timed.
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I gpuaudiobench_amd/csrc -I include \
//         tools/ubench/mfma_dft.hip -o tools/ubench/bin/mfma_dft
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gab_fft.hpp"

using gab::fft::cf;
typedef float v16f __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// data: [wave][16][64] complex, lane-major inside a value index
__global__ __launch_bounds__(256) void valu_kernel(cf* data, long long* cycles, int iters) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    cf* p = data + (size_t)wave * 16 * 64;
    cf v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = p[r * 64 + lane];
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        gab::fft::Butterfly<16, false>::run(v);
        cf o[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) o[k] = v[gab::fft::Butterfly<16, false>::out_slot(k)] * 0.25f;   // unitary scaling
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = o[k];
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r * 64 + lane] = v[r];
    if (lane == 0) cycles[wave] = t1 - t0;
}

// One wave: B operand of step s (K rows 2s, 2s+1) in b[s]: lane l holds row 2s + l/32, column l%32.
// Rows 0..15 = Re x[n], rows 16..31 = Im x[n].  A operand of step s: lane l holds M[l%32][2s + l/32].
// D (32x32): acc[v], row = 8*(v/4) + 4*(l/32) + v%4, column = l%32.
// data: [wave][32 rows][32 columns] real
__global__ __launch_bounds__(256) void mfma_kernel(float* data, const float* M, long long* cycles, int iters) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float* p = data + (size_t)wave * 32 * 32;
    float a[16], b[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        a[s] = M[(lane & 31) * 32 + 2 * s + (lane >> 5)];
        b[s] = p[(2 * s + (lane >> 5)) * 32 + (lane & 31)];
    }
    v16f acc;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
        // fed straight back (timing): the D layout is not the B layout, so after the first
        // iteration the "columns" are scrambled — the arithmetic per iteration is the same
#pragma unroll
        for (int s = 0; s < 16; ++s) b[s] = acc[s];
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int v = 0; v < 16; ++v) p[(8 * (v / 4) + 4 * (lane >> 5) + (v & 3)) * 32 + (lane & 31)] = acc[v];
    if (lane == 0) cycles[wave] = t1 - t0;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const int blocks = 256, waves = blocks * 4;
    // ---- reference DFT-16 (unitary) in float64
    auto dft = [](const double* re, const double* im, double* ore, double* oim) {
        for (int k = 0; k < 16; ++k) {
            double sr = 0, si = 0;
            for (int n = 0; n < 16; ++n) {
                double a = -2.0 * M_PI * k * n / 16.0;
                sr += re[n] * cos(a) - im[n] * sin(a);
                si += re[n] * sin(a) + im[n] * cos(a);
            }
            ore[k] = sr * 0.25;
            oim[k] = si * 0.25;
        }
    };
    srand(3);
    // ---- VALU
    std::vector<float> hv((size_t)waves * 16 * 64 * 2);
    for (auto& x : hv) x = (float)rand() / (float)RAND_MAX - 0.5f;
    cf* dv;
    long long* dc;
    CK(hipMalloc(&dv, hv.size() * 4));
    CK(hipMalloc(&dc, sizeof(long long) * waves));
    CK(hipMemcpy(dv, hv.data(), hv.size() * 4, hipMemcpyHostToDevice));
    valu_kernel<<<blocks, 256>>>(dv, dc, 1);
    std::vector<float> ov(hv.size());
    CK(hipMemcpy(ov.data(), dv, hv.size() * 4, hipMemcpyDeviceToHost));
    double ev = 0;
    for (int lane = 0; lane < 64; ++lane) {
        double re[16], im[16], ore[16], oim[16];
        for (int r = 0; r < 16; ++r) { re[r] = hv[(r * 64 + lane) * 2]; im[r] = hv[(r * 64 + lane) * 2 + 1]; }
        dft(re, im, ore, oim);
        for (int k = 0; k < 16; ++k)
            ev = fmax(ev, fmax(fabs(ore[k] - ov[(k * 64 + lane) * 2]), fabs(oim[k] - ov[(k * 64 + lane) * 2 + 1])));
    }
    // ---- MFMA
    std::vector<float> hM(32 * 32), hm((size_t)waves * 32 * 32);
    for (int k = 0; k < 16; ++k)
        for (int n = 0; n < 16; ++n) {
            double a = -2.0 * M_PI * k * n / 16.0, c = 0.25 * cos(a), s = 0.25 * sin(a);
            hM[k * 32 + n] = (float)c;            hM[k * 32 + 16 + n] = (float)-s;        // Re out = c Re - s Im
            hM[(16 + k) * 32 + n] = (float)s;     hM[(16 + k) * 32 + 16 + n] = (float)c;  // Im out = s Re + c Im
        }
    for (auto& x : hm) x = (float)rand() / (float)RAND_MAX - 0.5f;
    float *dm, *dM;
    CK(hipMalloc(&dm, hm.size() * 4));
    CK(hipMalloc(&dM, hM.size() * 4));
    CK(hipMemcpy(dm, hm.data(), hm.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dM, hM.data(), hM.size() * 4, hipMemcpyHostToDevice));
    mfma_kernel<<<blocks, 256>>>(dm, dM, dc, 1);
    std::vector<float> om(hm.size());
    CK(hipMemcpy(om.data(), dm, hm.size() * 4, hipMemcpyDeviceToHost));
    double em = 0;
    for (int col = 0; col < 32; ++col) {
        double re[16], im[16], ore[16], oim[16];
        for (int n = 0; n < 16; ++n) { re[n] = hm[n * 32 + col]; im[n] = hm[(16 + n) * 32 + col]; }
        dft(re, im, ore, oim);
        for (int k = 0; k < 16; ++k)
            em = fmax(em, fmax(fabs(ore[k] - om[k * 32 + col]), fabs(oim[k] - om[(16 + k) * 32 + col])));
    }
    // ---- timing
    auto run = [&](bool mfma) {
        std::vector<long long> hc(waves);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int w = 0; w < 2; ++w) {
            CK(hipEventRecord(e0));
            if (mfma) mfma_kernel<<<blocks, 256>>>(dm, dM, dc, iters);
            else valu_kernel<<<blocks, 256>>>(dv, dc, iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
        }
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(hc.data(), dc, sizeof(long long) * waves, hipMemcpyDeviceToHost));
        double cyc = 0;
        for (auto c : hc) cyc += (double)c;
        cyc /= waves;
        const double per_wave_iter = mfma ? 32.0 : 64.0;
        printf("{\"variant\": \"%s\", \"dfts_per_wave_iteration\": %.0f, \"memtime_ticks_per_dft16\": %.3f, "
               "\"ns_per_dft16_per_simd\": %.3f, \"chip_dft16_per_us\": %.0f, \"max_abs_err_vs_f64\": %.3g}\n",
               mfma ? "MFMA v_mfma_f32_32x32x2_f32, dense 32x32 real product" : "VALU packed-fp32 Butterfly<16>",
               per_wave_iter, cyc / iters / per_wave_iter, ms * 1e6 / iters / per_wave_iter,
               per_wave_iter * iters * waves / (ms * 1e3), mfma ? em : ev);
    };
    run(false);
    run(true);
    return 0;
}
