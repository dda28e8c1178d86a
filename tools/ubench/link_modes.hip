// link_modes — which MIX of movers gives one audio buffer each way (2 MiB up, 2 MiB down) the most overlap (tools only).
// link_duplex showed: a kernel reads pinned memory at 50 GB/s and writes it at 51 GB/s, but doing both at once it gets
// 27-29 GB/s each way; the copy engines run both directions at once in 53 us wall.  This times the remaining combinations
// on the host clock (issue -> everything synchronised), one JSON object per line:
//   mixed        kernel reads host || engine copies D2H;  engine copies H2D || kernel writes host
//   chunks       the 2 MiB copy cut into G back-to-back copies on one stream (per-copy cost of the engine)
//   pitched      a channel group's slab of the sample-major output (rows of `w` bytes at a 4 KiB pitch) by hipMemcpy2DAsync
//   waitvalue    a copy on stream 2 released by a word the KERNEL writes (hipStreamWaitValue32) against an event
//   staged       G x {kernel on a channel group: reads its rows from pinned memory, parks them sample-major in device memory;
//                     pitched D2H of the group's slab on a second stream behind an event}
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

__global__ __launch_bounds__(256) void rd_kernel(const float* __restrict__ h, float* __restrict__ sink, size_t n_floats) {
    const size_t per = n_floats / gridDim.x;
    const float4* q = reinterpret_cast<const float4*>(h + per * blockIdx.x);
    float acc = 0.f;
    for (size_t i = threadIdx.x; i < per / 4; i += 256) { float4 v = q[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void wr_kernel(float* __restrict__ h, size_t n_floats, float v) {
    const size_t per = n_floats / gridDim.x;
    float4* q = reinterpret_cast<float4*>(h + per * blockIdx.x);
    for (size_t i = threadIdx.x; i < per / 4; i += 256) q[i] = make_float4(v, v, v, v);
}
// one workgroup per channel pair of the group [pair0, pair0 + gridDim.x): rows from `in` (pinned or device), parked sample-major
__global__ __launch_bounds__(256) void park_kernel(const float* __restrict__ in, float* __restrict__ park, int pair0, unsigned* word, unsigned v) {
    const int tid = threadIdx.x, q = pair0 + blockIdx.x;
    const float* xa = in + (size_t)(2 * q) * kB;
    float a0 = xa[tid], a1 = xa[tid + 256], b0 = xa[kB + tid], b1 = xa[kB + tid + 256];
    *reinterpret_cast<float2*>(park + (size_t)kT * tid + 2 * q) = make_float2(a0, b0);
    *reinterpret_cast<float2*>(park + (size_t)kT * (tid + 256) + 2 * q) = make_float2(a1, b1);
    (void)word; (void)v;
}
__global__ void word_kernel(unsigned* word, unsigned v) { __threadfence_system(); *word = v; }

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double pct(std::vector<double> v, double p) {
    std::sort(v.begin(), v.end());
    double x = p * (v.size() - 1); size_t i = (size_t)x;
    return i + 1 < v.size() ? v[i] + (x - i) * (v[i + 1] - v[i]) : v[i];
}
template <class F> static void wall(const char* name, F&& fn, const char* extra = "") {
    std::vector<double> t;
    for (int i = 0; i < 220; ++i) { double t0 = now_us(); fn(); double t1 = now_us(); if (i >= 20) t.push_back(t1 - t0); }
    printf("{\"case\": \"%s\", \"p50_us\": %.2f, \"p95_us\": %.2f, \"min_us\": %.2f%s}\n", name, pct(t, 0.5), pct(t, 0.95), pct(t, 0.0), extra);
    fflush(stdout);
}

int main() {
    hipStream_t s1, s2, s3;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
    float *hin, *hout, *din, *dout, *park, *sink;
    CK(hipHostMalloc(&hin, kBytes)); CK(hipHostMalloc(&hout, kBytes));
    CK(hipMalloc(&din, kBytes)); CK(hipMalloc(&dout, kBytes)); CK(hipMalloc(&park, kBytes)); CK(hipMalloc(&sink, 64));
    for (size_t i = 0; i < kBytes / 4; ++i) hin[i] = (float)(i % 977) * 0.001f;
    memset(hout, 0, kBytes);
    const size_t n = kBytes / 4;
    hipEvent_t ev[64];
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    char extra[128];

    wall("kernel_rd_alone", [&] { rd_kernel<<<64, 256, 0, s1>>>(hin, sink, n); CK(hipStreamSynchronize(s1)); });
    wall("kernel_wr_alone", [&] { wr_kernel<<<64, 256, 0, s1>>>(hout, n, 1.f); CK(hipStreamSynchronize(s1)); });
    wall("engine_d2h_alone", [&] { CK(hipMemcpyAsync(hout, dout, kBytes, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2)); });
    wall("engine_h2d_alone", [&] { CK(hipMemcpyAsync(din, hin, kBytes, hipMemcpyHostToDevice, s2)); CK(hipStreamSynchronize(s2)); });
    wall("mixed_kernel_rd_engine_d2h", [&] {
        rd_kernel<<<64, 256, 0, s1>>>(hin, sink, n);
        CK(hipMemcpyAsync(hout, dout, kBytes, hipMemcpyDeviceToHost, s2));
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); });
    wall("mixed_engine_h2d_kernel_wr", [&] {
        CK(hipMemcpyAsync(din, hin, kBytes, hipMemcpyHostToDevice, s2));
        wr_kernel<<<64, 256, 0, s1>>>(hout, n, 1.f);
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); });
    wall("engines_both", [&] {
        CK(hipMemcpyAsync(din, hin, kBytes, hipMemcpyHostToDevice, s2));
        CK(hipMemcpyAsync(hout, dout, kBytes, hipMemcpyDeviceToHost, s3));
        CK(hipStreamSynchronize(s2)); CK(hipStreamSynchronize(s3)); });

    for (int G : {1, 2, 4, 8, 16}) {
        snprintf(extra, sizeof extra, ", \"chunks\": %d", G);
        const size_t cb = kBytes / G;
        wall("h2d_chunks_one_stream", [&] {
            for (int g = 0; g < G; ++g) CK(hipMemcpyAsync((char*)din + g * cb, (char*)hin + g * cb, cb, hipMemcpyHostToDevice, s2));
            CK(hipStreamSynchronize(s2)); }, extra);
        wall("d2h_chunks_one_stream", [&] {
            for (int g = 0; g < G; ++g) CK(hipMemcpyAsync((char*)hout + g * cb, (char*)dout + g * cb, cb, hipMemcpyDeviceToHost, s2));
            CK(hipStreamSynchronize(s2)); }, extra);
    }
    for (int wbytes : {256, 512, 1024, 2048, 4096}) {
        snprintf(extra, sizeof extra, ", \"row_bytes\": %d, \"bytes\": %d", wbytes, wbytes * kB);
        wall("d2h_pitched_slab", [&] {
            CK(hipMemcpy2DAsync(hout, kT * 4, dout, kT * 4, wbytes, kB, hipMemcpyDeviceToHost, s2));
            CK(hipStreamSynchronize(s2)); }, extra);
    }
    // staged pipeline, plain calls: kernels read pinned input themselves, slabs leave by the engine
    for (int G : {1, 2, 4, 8}) {
        const int pairs = kT / 2 / G;
        snprintf(extra, sizeof extra, ", \"groups\": %d", G);
        wall("staged_zero_copy_in_engine_out", [&] {
            for (int g = 0; g < G; ++g) {
                park_kernel<<<pairs, 256, 0, s1>>>(hin, park, g * pairs, nullptr, 0);
                CK(hipEventRecord(ev[g], s1));
                CK(hipStreamWaitEvent(s2, ev[g], 0));
                CK(hipMemcpy2DAsync((char*)hout + (size_t)g * pairs * 8, kT * 4, (char*)park + (size_t)g * pairs * 8, kT * 4, (size_t)pairs * 8, kB, hipMemcpyDeviceToHost, s2));
            }
            CK(hipStreamSynchronize(s2)); }, extra);
        bool ok = true;
        for (int q = 0; q < kT / 2 && ok; q += 37)
            for (int sm = 0; sm < kB; sm += 101)
                if (hout[(size_t)kT * sm + 2 * q] != hin[(size_t)(2 * q) * kB + sm]) { ok = false; break; }
        printf("{\"case\": \"staged_zero_copy_in_engine_out_ok\", \"groups\": %d, \"value\": %s}\n", G, ok ? "true" : "false");
        memset(hout, 0, kBytes);
        wall("staged_engine_in_engine_out", [&] {
            for (int g = 0; g < G; ++g) {
                const size_t rows = (size_t)pairs * 2 * kB * 4;
                CK(hipMemcpyAsync((char*)din + g * rows, (char*)hin + g * rows, rows, hipMemcpyHostToDevice, s3));
                CK(hipEventRecord(ev[16 + g], s3));
                CK(hipStreamWaitEvent(s1, ev[16 + g], 0));
                park_kernel<<<pairs, 256, 0, s1>>>(din, park, g * pairs, nullptr, 0);
                CK(hipEventRecord(ev[g], s1));
                CK(hipStreamWaitEvent(s2, ev[g], 0));
                CK(hipMemcpy2DAsync((char*)hout + (size_t)g * pairs * 8, kT * 4, (char*)park + (size_t)g * pairs * 8, kT * 4, (size_t)pairs * 8, kB, hipMemcpyDeviceToHost, s2));
            }
            CK(hipStreamSynchronize(s2)); }, extra);
    }
    // the same staged pipeline as ONE graph launch
    for (int G : {2, 4, 8}) {
        const int pairs = kT / 2 / G;
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
        for (int g = 0; g < G; ++g) {
            park_kernel<<<pairs, 256, 0, s1>>>(hin, park, g * pairs, nullptr, 0);
            CK(hipEventRecord(ev[g], s1));
            CK(hipStreamWaitEvent(s2, ev[g], 0));
            CK(hipMemcpy2DAsync((char*)hout + (size_t)g * pairs * 8, kT * 4, (char*)park + (size_t)g * pairs * 8, kT * 4, (size_t)pairs * 8, kB, hipMemcpyDeviceToHost, s2));
        }
        CK(hipEventRecord(ev[40], s2));
        CK(hipStreamWaitEvent(s1, ev[40], 0));
        CK(hipStreamEndCapture(s1, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        snprintf(extra, sizeof extra, ", \"groups\": %d", G);
        wall("graph_zero_copy_in_engine_out", [&] { CK(hipGraphLaunch(exec, s1)); CK(hipStreamSynchronize(s1)); }, extra);
        CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    }
    // a copy released by a kernel-written word against one released by an event
    {
        int can = 0;
        CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
        printf("{\"case\": \"can_use_stream_wait_value\", \"value\": %d}\n", can);
        unsigned* sig = nullptr;
        hipError_t e = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory);
        if (can && e == hipSuccess) {
            CK(hipMemset(sig, 0, 8));
            unsigned v = 0;
            bool ok = true;
            wall("kernel_word_releases_copy", [&] {
                ++v;
                if (hipStreamWaitValue32(s2, sig, v, hipStreamWaitValueGte, 0xffffffffu) != hipSuccess) { ok = false; return; }
                CK(hipMemcpyAsync(hout, dout, kBytes / 8, hipMemcpyDeviceToHost, s2));
                word_kernel<<<1, 64, 0, s1>>>(sig, v);
                CK(hipStreamSynchronize(s2)); CK(hipStreamSynchronize(s1)); }, ", \"copy_bytes\": 262144");
            printf("{\"case\": \"kernel_word_releases_copy_ok\", \"value\": %s}\n", ok ? "true" : "false");
        } else {
            printf("{\"case\": \"signal_memory\", \"error\": \"%s\"}\n", hipGetErrorString(e));
        }
        wall("event_releases_copy", [&] {
            word_kernel<<<1, 64, 0, s1>>>((unsigned*)sink, 1u);
            CK(hipEventRecord(ev[0], s1));
            CK(hipStreamWaitEvent(s2, ev[0], 0));
            CK(hipMemcpyAsync(hout, dout, kBytes / 8, hipMemcpyDeviceToHost, s2));
            CK(hipStreamSynchronize(s2)); }, ", \"copy_bytes\": 262144");
        wall("same_stream_kernel_then_copy", [&] {
            word_kernel<<<1, 64, 0, s1>>>((unsigned*)sink, 1u);
            CK(hipMemcpyAsync(hout, dout, kBytes / 8, hipMemcpyDeviceToHost, s1));
            CK(hipStreamSynchronize(s1)); }, ", \"copy_bytes\": 262144");
    }
    return 0;
}
