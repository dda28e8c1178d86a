// dep_chain — what separates two DEPENDENT fp32 instructions of one wave on gfx950 (tools only).  conv1d's sum is a
// strictly ordered chain of adds, one chain per output: the kernel lasts (taps) x (this number).  Modes, 64 chain steps
// per loop iteration, per wave, for 1 and 2 waves per SIMD:
//   add        a = a + c                                  (the bare chain)
//   mul_add    p = x * c ; a = a + p                      (each add waits for the product issued just before it)
//   add_mulnext a = a + p_k ; p_k' = x * c  (independent) (products one step ahead, between the adds)
//   add_salu   a = a + c ; s_add                          (a scalar instruction between the adds)
//   add_sgpr   a = a + s                                  (operand from an SGPR, as with scalar-loaded taps)
//   four_independent_adds  four chains, round robin       (what an instruction costs when nothing waits)
//   add_ldsread a = a + c ; ds_read_b32                   (an LDS request between the adds, waited for every 4)
//   add_waitcnt a = a + c ; s_waitcnt lgkmcnt(0)          (a wait with nothing outstanding)
//   add_nop     a = a + c ; s_nop 0
//   add_ldsread_late_wait  as add_ldsread, one wait per 64 steps (the request's own issue cost)
// Prints shader clocks per chain step as one wave sees them (s_memtime) and ns per step on the event clock.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float cc) {
    float a = threadIdx.x, x = 1.0f + threadIdx.x * 1e-3f, p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
    const float c = cc;
    __shared__ float pad[256];
    pad[threadIdx.x] = cc;
    __syncthreads();
    const unsigned lds_addr = (unsigned)(size_t)&pad[threadIdx.x];
    unsigned s = 0;
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
                asm volatile("v_add_f32 %0, %0, %2\n s_add_u32 %1, %1, 1\n v_add_f32 %0, %0, %2\n s_add_u32 %1, %1, 1\n"
                             "v_add_f32 %0, %0, %2\n s_add_u32 %1, %1, 1\n v_add_f32 %0, %0, %2\n s_add_u32 %1, %1, 1"
                             : "+v"(a), "+s"(s) : "v"(c) : "scc");
            else if (MODE == 5)      // four independent chains
                asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c));
            else if (MODE == 7)      // a wait with nothing outstanding between the adds
                asm volatile("v_add_f32 %0, %0, %1\n s_waitcnt lgkmcnt(0)\n v_add_f32 %0, %0, %1\n s_waitcnt lgkmcnt(0)\n"
                             "v_add_f32 %0, %0, %1\n s_waitcnt lgkmcnt(0)\n v_add_f32 %0, %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(c));
            else if (MODE == 8)      // s_nop between the adds
                asm volatile("v_add_f32 %0, %0, %1\n s_nop 0\n v_add_f32 %0, %0, %1\n s_nop 0\n"
                             "v_add_f32 %0, %0, %1\n s_nop 0\n v_add_f32 %0, %0, %1\n s_nop 0" : "+v"(a) : "v"(c));
            else if (MODE == 9)      // an LDS read between the adds, waited for only at the end of the 64 steps
                asm volatile("v_add_f32 %0, %0, %2\n ds_read_b32 %1, %3\n v_add_f32 %0, %0, %2\n ds_read_b32 %1, %3\n"
                             "v_add_f32 %0, %0, %2\n ds_read_b32 %1, %3\n v_add_f32 %0, %0, %2\n ds_read_b32 %1, %3"
                             : "+v"(a), "=&v"(p1) : "v"(c), "v"(lds_addr) : "memory");
            else if (MODE == 6)      // the chain with an LDS read between the adds (results unused)
                asm volatile("v_add_f32 %0, %0, %2\n ds_read_b32 %1, %3\n v_add_f32 %0, %0, %2\n ds_read_b32 %1, %3\n"
                             "v_add_f32 %0, %0, %2\n ds_read_b32 %1, %3\n v_add_f32 %0, %0, %2\n ds_read_b32 %1, %3\n s_waitcnt lgkmcnt(0)"
                             : "+v"(a), "=&v"(p1) : "v"(c), "v"(lds_addr) : "memory");
            else
                asm volatile("v_add_f32 %0, %1, %0\n v_add_f32 %0, %1, %0\n v_add_f32 %0, %1, %0\n v_add_f32 %0, %1, %0" : "+v"(a) : "s"(c));
        }
        if (MODE == 9) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("add", w); run<1>("mul_add", w); run<2>("add_mulnext", w); run<3>("add_salu", w); run<4>("add_sgpr", w); run<5>("four_independent_adds", w); run<6>("add_ldsread", w); run<7>("add_waitcnt", w); run<8>("add_nop", w); run<9>("add_ldsread_late_wait", w);
    }
    return 0;
}
