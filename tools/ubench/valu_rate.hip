// Microbenchmark: VALU issue cost of wave64 v_fma_f32 vs v_pk_fma_f32 on gfx950,
// for 1, 2 and 4 resident waves per SIMD.  Prints cycles per instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {1, 2}, p5 = {3, 4}, p6 = {5, 6}, p7 = {7, 8};
    const float c = 1.0001f, d = 0.5f;
    const v2f pc = {c, c}, pd = {d, d};
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                             "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc), "v"(pd));
            }
        }
    }
    long long t1 = clock64();
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    if (r == 12345.678f) out[0] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = (float)(t1 - t0);
}
template <int MODE> void run(const char* name, int waves_per_simd) {
    float* d; hipMalloc(&d, 16);
    int iters = 2000;
    // one block of 256 threads = 1 wave per SIMD on its CU; launch waves_per_simd blocks per CU
    int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 10); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    double instr_per_wave = (double)iters * 64;
    double cyc_per_instr_wave = h[1] / instr_per_wave;                       // shader clocks per instr as seen by one wave
    double ghz = h[1] / (ms * 1e6);                                            // clock64 ticks per ns (approx, whole kernel)
    printf("%-14s waves/SIMD=%d: %.2f clk/instr per wave -> %.2f clk/instr per SIMD  (kernel %.3f ms, ~%.2f GHz tick)\n",
           name, waves_per_simd, cyc_per_instr_wave, cyc_per_instr_wave / waves_per_simd, ms, ghz);
    hipFree(d);
}

// Mixed: a 512-thread workgroup = two waves per SIMD; waves 0-3 issue packed fp32, waves 4-7 plain fp32 (MIX = 1), or
// both halves the same kind (MIX = 0 packed + packed, MIX = 2 plain + plain).  Each half does `iters` x 64 instructions;
// out[2 + half] = clocks of wave 0 / wave 4 of block 0.
template <int MIX>
__global__ __launch_bounds__(512) void kmix(float* out, int iters) {
    const int half = threadIdx.x >> 8;
    const bool packed = MIX == 0 || (MIX == 1 && half == 0);
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {1, 2}, p5 = {3, 4}, p6 = {5, 6}, p7 = {7, 8};
    const float c = 1.0001f, d = 0.5f;
    const v2f pc = {c, c}, pd = {d, d};
    __syncthreads();
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (!packed) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                             "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc), "v"(pd));
        }
    }
    long long t1 = clock64();
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    if (r == 12345.678f) out[0] = r;
    if ((threadIdx.x & 255) == 0 && blockIdx.x == 0) out[2 + half] = (float)(t1 - t0);
}
template <int MIX> void runmix(const char* name) {
    float* d; hipMalloc(&d, 16);
    const int iters = 2000;
    kmix<MIX><<<256, 512>>>(d, 10); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); kmix<MIX><<<256, 512>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("two waves per SIMD, %-16s: first wave %.2f clk/instr, its SIMD mate %.2f clk/instr (kernel %.3f ms)\n", name,
           h[2] / (iters * 64.0), h[3] / (iters * 64.0), ms);
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4, 8}) run<0>("v_fma_f32", w);
    for (int w : {1, 2, 4, 8}) run<1>("v_pk_fma_f32", w);
    runmix<0>("packed + packed"); runmix<1>("packed + plain"); runmix<2>("plain + plain");
    return 0;
}
