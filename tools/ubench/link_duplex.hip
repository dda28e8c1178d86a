// link_duplex — what the PCIe link gives a KERNEL that moves one audio buffer each way (tools only).
//
// The real-time path of Conv1D_accel moves 2 MiB up and 2 MiB down per buffer (1024 channels x 512 samples);
// this measures the floor of that data path without any arithmetic:
//   rd         workgroups read pinned host memory (track-major rows, 4 B or 16 B per lane)
//   wr         workgroups write pinned host memory in contiguous rows of `row` bytes
//   wr_scatter 8 / 16 B pieces at a 4 KiB stride (a channel pair / duo writing sample-major output directly)
//   duplex     rd and wr at once in one launch (different workgroups)
//   pipe       the planned pipeline: G channel groups in dispatch order, reads gated to `depth` groups in
//              flight, results parked in device memory, the group's last workgroup drains its slab to the host
//   copies     hipMemcpyAsync H2D, D2H, and both on two streams
//   sync       launch -> host-visible completion: hipStreamSynchronize against polling a pinned word
// Prints one JSON object per line.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kT = 1024, kB = 512;
constexpr size_t kBytes = (size_t)kT * kB * 4;

template <int VEC>
__global__ __launch_bounds__(256) void rd_kernel(const float* __restrict__ h, float* __restrict__ sink, size_t n_floats) {
    // every workgroup takes a contiguous share; lanes read VEC floats each, a wave 256*VEC/4.. contiguous bytes
    const size_t per = n_floats / gridDim.x;
    const float* p = h + per * blockIdx.x;
    float acc = 0.f;
    if (VEC == 4) {
        const float4* q = reinterpret_cast<const float4*>(p);
        for (size_t i = threadIdx.x; i < per / 4; i += 256) { float4 v = q[i]; acc += v.x + v.y + v.z + v.w; }
    } else {
        for (size_t i = threadIdx.x; i < per; i += 256) acc += p[i];
    }
    if (acc == 123.456f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void wr_kernel(float* __restrict__ h, size_t n_floats, float v) {
    const size_t per = n_floats / gridDim.x;
    float4* q = reinterpret_cast<float4*>(h + per * blockIdx.x);
    for (size_t i = threadIdx.x; i < per / 4; i += 256) q[i] = make_float4(v, v, v, v);
}

// piece = 2 (float2) or 4 (float4) floats at stride kT floats: workgroup b owns columns [piece*b, piece*b+piece)
template <int PIECE>
__global__ __launch_bounds__(256) void wr_scatter_kernel(float* __restrict__ h, float v) {
    for (int s = threadIdx.x; s < kB; s += 256) {
        float* o = h + (size_t)kT * s + PIECE * blockIdx.x;
        if (PIECE == 2) *reinterpret_cast<float2*>(o) = make_float2(v, v);
        else *reinterpret_cast<float4*>(o) = make_float4(v, v, v, v);
    }
}

__global__ __launch_bounds__(256) void duplex_kernel(const float* __restrict__ hin, float* __restrict__ hout, float* __restrict__ sink,
                                                     size_t n_floats, int n_rd) {
    if ((int)blockIdx.x < n_rd) {
        const size_t per = n_floats / n_rd;
        const float4* q = reinterpret_cast<const float4*>(hin + per * blockIdx.x);
        float acc = 0.f;
        for (size_t i = threadIdx.x; i < per / 4; i += 256) { float4 v = q[i]; acc += v.x + v.y + v.z + v.w; }
        if (acc == 123.456f) sink[0] = acc;
    } else {
        const int w = blockIdx.x - n_rd, n_wr = gridDim.x - n_rd;
        const size_t per = n_floats / n_wr;
        float4* q = reinterpret_cast<float4*>(hout + per * w);
        for (size_t i = threadIdx.x; i < per / 4; i += 256) q[i] = make_float4(1.f, 2.f, 3.f, 4.f);
    }
}

// The planned pipeline without arithmetic.  Grid = kT/2 workgroups (one per channel pair), group g =
// blockIdx / (grid / G) = a contiguous channel range.  ctr[0] = workgroups whose input has arrived (the gate),
// ctr[1 + g] = workgroups of group g whose results are parked, ctr[1 + G] = groups drained; done = pinned word.
__global__ __launch_bounds__(256) void pipe_kernel(const float* __restrict__ hin, float* __restrict__ hout, float* __restrict__ park,
                                                   unsigned* __restrict__ ctr, volatile unsigned* __restrict__ done, unsigned epoch,
                                                   int G, int depth, int gate) {
    const int tid = threadIdx.x;
    const int per_group = gridDim.x / G;
    const int g = blockIdx.x / per_group;
    __shared__ int last;
    if (gate && g >= depth) {
        if (tid == 0) {
            const unsigned want = epoch * gridDim.x + (unsigned)(g - depth + 1) * per_group;
            int tries = 0;
            while ((int)(__hip_atomic_load(&ctr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0 && ++tries < (1 << 22)) __builtin_amdgcn_s_sleep(2);
        }
        __syncthreads();
    }
    const int q = blockIdx.x;
    const float* xa = hin + (size_t)(2 * q) * kB;
    float a0 = xa[tid], a1 = xa[tid + 256], b0 = xa[kB + tid], b1 = xa[kB + tid + 256];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(&ctr[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // "results": sample-major, 8 bytes per sample at a 4 KiB stride, into device memory
    *reinterpret_cast<float2*>(park + (size_t)kT * tid + 2 * q) = make_float2(a0, b0);
    *reinterpret_cast<float2*>(park + (size_t)kT * (tid + 256) + 2 * q) = make_float2(a1, b1);
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(&ctr[1 + g], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = (old == epoch * per_group + per_group - 1);
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    // drain the group's slab: rows of per_group*2 channels
    const int row_f4 = per_group * 2 / 4;                    // float4 per row
    const int rows_per_iter = 256 / row_f4;
    const int r0 = tid / row_f4, c = tid % row_f4;
    const size_t col0 = (size_t)g * per_group * 2;
    for (int s = r0; s < kB; s += rows_per_iter * 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ss = s + u * rows_per_iter;
            v[u] = ss < kB ? *reinterpret_cast<const float4*>(park + (size_t)kT * ss + col0 + 4 * c) : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ss = s + u * rows_per_iter;
            if (ss < kB) *reinterpret_cast<float4*>(hout + (size_t)kT * ss + col0 + 4 * c) = v[u];
        }
    }
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(&ctr[1 + G], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == epoch * G + G - 1) { __threadfence_system(); *done = epoch + 1; }
    }
}

__global__ void flag_kernel(volatile unsigned* done, unsigned v) { *done = v; }
__global__ void empty_kernel() {}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static void wait_word(volatile unsigned* w, unsigned v) {      // bounded: a lost launch must not hang the box
    const double t0 = now_us();
    while (*w != v) {
        if (now_us() - t0 > 2e6) { fprintf(stderr, "link_duplex: completion word never arrived\n"); exit(2); }
    }
}
static double pct(std::vector<double> v, double p) {
    std::sort(v.begin(), v.end());
    double x = p * (v.size() - 1);
    size_t i = (size_t)x;
    return i + 1 < v.size() ? v[i] + (x - i) * (v[i + 1] - v[i]) : v[i];
}

template <class F>
static void wall(const char* name, F&& fn, hipStream_t s, const char* extra = "") {
    std::vector<double> t;
    for (int i = 0; i < 320; ++i) {
        double t0 = now_us();
        fn();
        double t1 = now_us();
        if (i >= 20) t.push_back(t1 - t0);
    }
    printf("{\"case\": \"%s\", \"p50_us\": %.2f, \"p95_us\": %.2f, \"min_us\": %.2f%s}\n", name, pct(t, 0.5), pct(t, 0.95), pct(t, 0.0), extra);
    fflush(stdout);
}

template <class F>
static double dev_us(F&& launch, hipStream_t s, int reps = 200) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) launch();
    CK(hipStreamSynchronize(s));
    // one launch at a time (a real-time caller has one buffer in flight): event pair per launch
    std::vector<double> t;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0, s));
        launch();
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms * 1e3);
    }
    return pct(t, 0.5);
}

int main() {
    hipStream_t s, s2;
    CK(hipStreamCreate(&s)); CK(hipStreamCreate(&s2));
    float *hin, *hout, *din, *dout, *park, *sink;
    unsigned *ctr; unsigned* done;
    CK(hipHostMalloc(&hin, kBytes)); CK(hipHostMalloc(&hout, kBytes));
    CK(hipHostMalloc(&done, 64));
    CK(hipMalloc(&din, kBytes)); CK(hipMalloc(&dout, kBytes)); CK(hipMalloc(&park, kBytes)); CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&ctr, 4 * 64)); CK(hipMemset(ctr, 0, 4 * 64));
    for (size_t i = 0; i < kBytes / 4; ++i) hin[i] = (float)(i % 977) * 0.001f;
    memset(hout, 0, kBytes); *done = 0;
    const size_t n = kBytes / 4;
    const double gbs = kBytes / 1e3;      // bytes / us -> GB/s when divided by us

    for (int wg : {16, 32, 64, 128, 256, 512, 1024}) {
        double r4 = dev_us([&] { rd_kernel<4><<<wg, 256, 0, s>>>(hin, sink, n); }, s);
        double r1 = dev_us([&] { rd_kernel<1><<<wg, 256, 0, s>>>(hin, sink, n); }, s);
        double w = dev_us([&] { wr_kernel<<<wg, 256, 0, s>>>(hout, n, 1.0f); }, s);
        double d = dev_us([&] { duplex_kernel<<<2 * wg, 256, 0, s>>>(hin, hout, sink, n, wg); }, s);
        printf("{\"case\": \"kernel_link\", \"workgroups\": %d, \"rd16_us\": %.2f, \"rd16_GBs\": %.1f, \"rd4_us\": %.2f, \"rd4_GBs\": %.1f, \"wr16_us\": %.2f, \"wr16_GBs\": %.1f, "
               "\"duplex_us\": %.2f, \"duplex_each_way_GBs\": %.1f}\n", wg, r4, gbs / r4, r1, gbs / r1, w, gbs / w, d, gbs / d);
        fflush(stdout);
    }
    {
        double s2u = dev_us([&] { wr_scatter_kernel<2><<<kT / 2, 256, 0, s>>>(hout, 1.0f); }, s);
        double s4u = dev_us([&] { wr_scatter_kernel<4><<<kT / 4, 256, 0, s>>>(hout, 1.0f); }, s);
        printf("{\"case\": \"wr_scatter\", \"piece8B_us\": %.2f, \"piece8B_GBs\": %.1f, \"piece16B_us\": %.2f, \"piece16B_GBs\": %.1f}\n", s2u, gbs / s2u, s4u, gbs / s4u);
    }
    // the pipeline without arithmetic
    unsigned epoch = 0;
    for (int G : {1, 2, 4, 8, 16, 32}) {
        for (int depth : {1, 2, 3}) {
            for (int gate : {0, 1}) {
                if (!gate && depth != 1) continue;
                if (G == 1 && (gate || depth != 1)) continue;
                CK(hipMemsetAsync(ctr, 0, 4 * 64, s)); epoch = 0;
                double t = dev_us([&] { pipe_kernel<<<kT / 2, 256, 0, s>>>(hin, hout, park, ctr, done, epoch, G, depth, gate); ++epoch; }, s);
                bool ok = true;
                for (int q = 0; q < kT / 2 && ok; q += 37)
                    for (int sm = 0; sm < kB; sm += 101) {
                        if (hout[(size_t)kT * sm + 2 * q] != hin[(size_t)(2 * q) * kB + sm] || hout[(size_t)kT * sm + 2 * q + 1] != hin[(size_t)(2 * q + 1) * kB + sm]) { ok = false; break; }
                    }
                printf("{\"case\": \"pipe\", \"groups\": %d, \"depth\": %d, \"gated\": %d, \"device_us\": %.2f, \"transposed_ok\": %s}\n", G, depth, gate, t, ok ? "true" : "false");
                fflush(stdout);
                memset(hout, 0, kBytes);
            }
        }
    }
    // the pipeline end to end on the host clock: launch -> pinned completion word (no stream synchronize)
    for (int G : {8, 16}) {
        CK(hipMemsetAsync(ctr, 0, 4 * 64, s)); CK(hipStreamSynchronize(s)); epoch = 0; *done = 0;
        char extra[64]; snprintf(extra, sizeof extra, ", \"groups\": %d", G);
        wall("pipe_launch_to_pinned_word", [&] {
            pipe_kernel<<<kT / 2, 256, 0, s>>>(hin, hout, park, ctr, done, epoch, G, 2, 1); ++epoch;
            wait_word(done, epoch);
        }, s, extra);
        CK(hipStreamSynchronize(s));
        CK(hipMemsetAsync(ctr, 0, 4 * 64, s)); CK(hipStreamSynchronize(s)); epoch = 0; *done = 0;
        wall("pipe_launch_to_stream_sync", [&] {
            pipe_kernel<<<kT / 2, 256, 0, s>>>(hin, hout, park, ctr, done, epoch, G, 2, 1); ++epoch;
            CK(hipStreamSynchronize(s));
        }, s, extra);
    }
    // copy engines
    for (size_t bytes : {kBytes, kBytes / 8}) {
        char extra[64]; snprintf(extra, sizeof extra, ", \"bytes\": %zu", bytes);
        wall("memcpy_h2d_sync", [&] { CK(hipMemcpyAsync(din, hin, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); }, s, extra);
        wall("memcpy_d2h_sync", [&] { CK(hipMemcpyAsync(hout, dout, bytes, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); }, s, extra);
        wall("memcpy_both_two_streams", [&] {
            CK(hipMemcpyAsync(din, hin, bytes, hipMemcpyHostToDevice, s));
            CK(hipMemcpyAsync(hout, dout, bytes, hipMemcpyDeviceToHost, s2));
            CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2)); }, s, extra);
        wall("memcpy_h2d_then_d2h_one_stream", [&] {
            CK(hipMemcpyAsync(din, hin, bytes, hipMemcpyHostToDevice, s));
            CK(hipMemcpyAsync(hout, dout, bytes, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s)); }, s, extra);
    }
    // launch -> completion
    wall("empty_kernel_stream_sync", [&] { empty_kernel<<<1, 64, 0, s>>>(); CK(hipStreamSynchronize(s)); }, s);
    unsigned v = 0; *done = 0;
    wall("flag_kernel_poll_pinned_word", [&] { ++v; flag_kernel<<<1, 64, 0, s>>>(done, v); wait_word(done, v); }, s);
    CK(hipStreamSynchronize(s));
    return 0;
}
