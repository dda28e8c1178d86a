// dep_chain — what separates two DEPENDENT fp32 instructions of one wave on gfx950 (tools only).  conv1d's sum is a
// strictly ordered chain of adds, one chain per output: the kernel lasts (taps) x (this number).  Modes, 64 chain steps
// per loop iteration, per wave, for 1 and 2 waves per SIMD:
//   add        a = a + c                                  (the bare chain)
//   mul_add    p = x * c ; a = a + p                      (each add waits for the product issued just before it)
//   add_mulnext a = a + p_k ; p_k' = x * c  (independent) (products one step ahead, between the adds)
//   add_salu   a = a + c ; s_add                          (a scalar instruction between the adds)
//   add_sgpr   a = a + s                                  (operand from an SGPR, as with scalar-loaded taps)
// Prints shader clocks per chain step as one wave sees them (s_memtime) and ns per step on the event clock.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float cc) {
    float a = threadIdx.x, x = 1.0f + threadIdx.x * 1e-3f, p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
    const float c = cc;
    int s = 0;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0)
                asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));
            else if (MODE == 1)
                asm volatile("v_mul_f32 %1, %2, %3\n v_add_f32 %0, %0, %1\n v_mul_f32 %1, %2, %3\n v_add_f32 %0, %0, %1\n"
                             "v_mul_f32 %1, %2, %3\n v_add_f32 %0, %0, %1\n v_mul_f32 %1, %2, %3\n v_add_f32 %0, %0, %1"
                             : "+v"(a), "+v"(p0) : "v"(x), "v"(c));
            else if (MODE == 2)
                asm volatile("v_add_f32 %0, %0, %1\n v_mul_f32 %1, %5, %6\n v_add_f32 %0, %0, %2\n v_mul_f32 %2, %5, %6\n"
                             "v_add_f32 %0, %0, %3\n v_mul_f32 %3, %5, %6\n v_add_f32 %0, %0, %4\n v_mul_f32 %4, %5, %6"
                             : "+v"(a), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(x), "v"(c));
            else if (MODE == 3)
                asm volatile("v_add_f32 %0, %0, %2\n s_add_i32 %1, %1, 1\n v_add_f32 %0, %0, %2\n s_add_i32 %1, %1, 1\n"
                             "v_add_f32 %0, %0, %2\n s_add_i32 %1, %1, 1\n v_add_f32 %0, %0, %2\n s_add_i32 %1, %1, 1"
                             : "+v"(a), "+s"(s) : "v"(c));
            else
                asm volatile("v_add_f32 %0, %1, %0\n v_add_f32 %0, %1, %0\n v_add_f32 %0, %1, %0\n v_add_f32 %0, %1, %0" : "+v"(a) : "s"(c));
        }
    }
    long long t1 = clock64();
    float r = a + p0 + p1 + p2 + p3 + (float)s;
    if (r == 12345.678f) out[0] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = (float)(t1 - t0);
}

template <int MODE> void run(const char* name, int waves_per_simd) {
    float* d; hipMalloc(&d, 16);
    const int iters = 4000;
    const int blocks = 256 * waves_per_simd;           // 256 threads = one wave per SIMD of a CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) k<MODE><<<blocks, 256>>>(d, iters, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, iters, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double steps = (double)iters * 64;
    printf("{\"mode\": \"%s\", \"waves_per_simd\": %d, \"memtime_ticks_per_step\": %.2f, \"ns_per_step\": %.2f, \"kernel_ms\": %.3f}\n",
           name, waves_per_simd, h[1] / steps, ms * 1e6 / steps, ms);
    hipFree(d);
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        run<0>("add", w); run<1>("mul_add", w); run<2>("add_mulnext", w); run<3>("add_salu", w); run<4>("add_sgpr", w);
    }
    return 0;
}
