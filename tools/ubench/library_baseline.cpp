// library_baseline — the reference's intended FFT-convolution pipeline on rocFFT (through hipFFT),
// timed on the same GPU as a yardstick.  TOOLS ONLY: nothing in the product links an FFT library.
//
// Restates Conv1DAccelBenchmark::performBenchmarkIteration (cuda/bench_conv1d_accel.cu:258-304)
// without its host copies: zero-pad every track into a stride-N buffer (N = nextpow2(L + B - 1),
// :49-53) -> batched R2C -> spectral multiply by the precomputed IR spectra (:9-30) -> batched C2R
// -> first B real samples, scaled by 1/N, sample-major (the intent of ExtractRealPartKernel :32-47
// as the Metal sibling states it).  The reference's T separate device-to-device copies (:267-274)
// are replaced by ONE strided copy; everything else is one library call or kernel per stage, as there.
// Semantics are the reference's: zero history every buffer (stateless), all L taps transformed.
//
//   hipcc -O2 tools/ubench/library_baseline.cpp -o tools/ubench/bin/library_baseline -lhipfft
//   tools/ubench/bin/library_baseline [T=1024] [B=512] [L=4096] [iters=2000]
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define FK(x) do { hipfftResult r_ = (x); if (r_ != HIPFFT_SUCCESS) { fprintf(stderr, "%s: hipfft error %d\n", #x, (int)r_); exit(1); } } while (0)

__global__ void multiply_kernel(const float2* __restrict__ X, const float2* __restrict__ H, float2* __restrict__ Y, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) {
        float2 a = X[i], b = H[i];
        Y[i] = make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
    }
}

// out[T*s + t] = y[t*N + s] / N, s < B: a 32 x 32 tile through LDS so both sides are coalesced
__global__ void extract_kernel(const float* __restrict__ y, float* __restrict__ out, int T, int B, int N, float scale) {
    __shared__ float tile[32][33];
    int t0 = blockIdx.y * 32, s0 = blockIdx.x * 32;
    int tx = threadIdx.x, ty = threadIdx.y;
    for (int r = ty; r < 32; r += 8)
        if (t0 + r < T && s0 + tx < B) tile[r][tx] = y[(size_t)(t0 + r) * N + s0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (s0 + r < B && t0 + tx < T) out[(size_t)T * (s0 + r) + t0 + tx] = tile[tx][r] * scale;
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 1024, B = argc > 2 ? atoi(argv[2]) : 512;
    const int L = argc > 3 ? atoi(argv[3]) : 4096, iters = argc > 4 ? atoi(argv[4]) : 2000;
    int N = 1;
    while (N < L + B - 1) N *= 2;
    const int bins = N / 2 + 1;
    std::vector<float> h_in((size_t)T * B), h_ir((size_t)T * L);
    srand(7);
    for (auto& v : h_in) v = (float)rand() / (float)RAND_MAX * 2.0f - 1.0f;
    for (int t = 0; t < T; ++t)
        for (int j = 0; j < L; ++j) {      // the reference's bank shape: Hamming-windowed sinc / L
            double f = 0.1 + 0.05 * t / T, x = j - L / 2.0;
            double sinc = x == 0 ? 1.0 : sin(2 * M_PI * f * x) / (2 * M_PI * f * x);
            h_ir[(size_t)t * L + j] = (float)((0.54 - 0.46 * cos(2 * M_PI * j / (L - 1))) * sinc / L);
        }
    float *d_in, *d_pad, *d_y, *d_out;
    float2 *d_X, *d_H, *d_Y;
    CK(hipMalloc(&d_in, sizeof(float) * T * B));
    CK(hipMalloc(&d_pad, sizeof(float) * (size_t)T * N));
    CK(hipMalloc(&d_y, sizeof(float) * (size_t)T * N));
    CK(hipMalloc(&d_out, sizeof(float) * T * B));
    CK(hipMalloc(&d_X, sizeof(float2) * (size_t)T * bins));
    CK(hipMalloc(&d_H, sizeof(float2) * (size_t)T * bins));
    CK(hipMalloc(&d_Y, sizeof(float2) * (size_t)T * bins));
    CK(hipMemcpy(d_in, h_in.data(), sizeof(float) * T * B, hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipfftHandle fwd, inv;
    int n[1] = {N};
    FK(hipfftPlanMany(&fwd, 1, n, nullptr, 1, N, nullptr, 1, bins, HIPFFT_R2C, T));
    FK(hipfftPlanMany(&inv, 1, n, nullptr, 1, bins, nullptr, 1, N, HIPFFT_C2R, T));
    FK(hipfftSetStream(fwd, st));
    FK(hipfftSetStream(inv, st));
    // precomputeImpulseResponseFFTs (:175-228)
    CK(hipMemsetAsync(d_pad, 0, sizeof(float) * (size_t)T * N, st));
    CK(hipMemcpy2DAsync(d_pad, sizeof(float) * N, h_ir.data(), sizeof(float) * L, sizeof(float) * L, T, hipMemcpyHostToDevice, st));
    FK(hipfftExecR2C(fwd, d_pad, (hipfftComplex*)d_H));
    CK(hipStreamSynchronize(st));

    const size_t nb = (size_t)T * bins;
    auto one = [&]() {
        CK(hipMemsetAsync(d_pad, 0, sizeof(float) * (size_t)T * N, st));
        CK(hipMemcpy2DAsync(d_pad, sizeof(float) * N, d_in, sizeof(float) * B, sizeof(float) * B, T, hipMemcpyDeviceToDevice, st));
        FK(hipfftExecR2C(fwd, d_pad, (hipfftComplex*)d_X));
        multiply_kernel<<<(unsigned)((nb + 255) / 256), 256, 0, st>>>(d_X, d_H, d_Y, nb);
        FK(hipfftExecC2R(inv, (hipfftComplex*)d_Y, d_y));
        extract_kernel<<<dim3((B + 31) / 32, (T + 31) / 32), dim3(32, 8), 0, st>>>(d_y, d_out, T, B, N, 1.0f / N);
    };
    for (int i = 0; i < 50; ++i) one();
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) one();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));

    // sanity: the reference golden (:234-252) on a few outputs
    std::vector<float> h_out((size_t)T * B);
    CK(hipMemcpy(h_out.data(), d_out, sizeof(float) * T * B, hipMemcpyDeviceToHost));
    double worst = 0, peak = 0;
    for (int t = 0; t < T; t += (T > 8 ? T / 8 : 1))
        for (int s = 0; s < B; s += 37) {
            double acc = 0;
            for (int k = 0; k <= s && k < L; ++k) acc += (double)h_in[(size_t)t * B + s - k] * h_ir[(size_t)t * L + k];
            worst = fmax(worst, fabs(acc - h_out[(size_t)T * s + t]));
            peak = fmax(peak, fabs(acc));
        }
    const double us = ms * 1e3 / iters;
    // bytes the pipeline itself moves: pad write + strided copy + R2C r/w + multiply r/r/w + C2R r/w + extract
    const double moved = 4.0 * T * N + 8.0 * T * B + (4.0 * T * N + 8.0 * nb) + 24.0 * nb + (8.0 * nb + 4.0 * T * N) + (4.0 * T * B * 2);
    printf("{\"library\": \"hipFFT/rocFFT\", \"tracks\": %d, \"buffer_size\": %d, \"ir_length\": %d, \"fft_size\": %d, "
           "\"stages\": \"memset + strided copy + R2C + multiply + C2R + extract\", \"semantics\": \"stateless (zero history), as the reference\", "
           "\"us_per_buffer\": %.3f, \"buffers_per_sec\": %.1f, \"pipeline_bytes_per_buffer\": %.0f, \"pipeline_GBps\": %.1f, "
           "\"max_abs_err_vs_direct\": %.3g, \"golden_peak\": %.3g, \"iterations\": %d}\n",
           T, B, L, N, us, 1e6 / us, moved, moved / us / 1e3, worst, peak, iters);
    return 0;
}
