// two_chains — what do cross-stream event guards cost between two chains of dependent launches?
//
// Two independent 1024-channel streaming convolution plans (the C ABI of libgab_hip.so), one per
// HIP stream, one host thread each: the "two full-size chains on one device" case.  With --guard N
// every launch additionally records an event on its stream and first waits for the event the
// OTHER stream recorded N launches earlier — a dependency that is always satisfied in steady state
// (the pattern an alternate-buffer scheme would need as a correctness guard).
//
//   hipcc -O2 -I include tools/ubench/two_chains.cpp -o tools/ubench/bin/two_chains \
//         -L gpuaudiobench_amd -lgab_hip -Wl,-rpath,'$ORIGIN/../../../gpuaudiobench_amd'
//   tools/ubench/bin/two_chains [--buffers 3000] [--guard 0|1|2]
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "gab_c_api.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define GK(x) do { int r_ = (x); if (r_) { fprintf(stderr, "%s: %s\n", #x, gab_last_error()); exit(1); } } while (0)

int main(int argc, char** argv) {
    int buffers = 3000, guard = 0, chains = 2;
    unsigned evflags = hipEventDisableTiming;
    bool one_way = false;       // --one-way: only chain 0 waits (for chain 1's events); chain 1 only records
    bool value_ops = false;     // --value-ops: guards as hipStreamWriteValue32 / hipStreamWaitValue32 on signal memory
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--buffers")) buffers = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--guard")) guard = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--chains")) chains = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--release-to-device")) evflags |= hipEventReleaseToDevice;
        else if (!strcmp(argv[i], "--one-way")) one_way = true;
        else if (!strcmp(argv[i], "--value-ops")) value_ops = true;
    }
    const int T = 1024, B = 512, L = 4096, NIN = 8;
    std::vector<gab_conv_plan*> plan(chains);
    std::vector<hipStream_t> st(chains);
    std::vector<std::vector<float*>> in(chains);
    std::vector<float*> out(chains);
    std::vector<float> h((size_t)T * L);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 1e-3f;
    float* d_ir;
    CK(hipMalloc(&d_ir, h.size() * 4));
    CK(hipMemcpy(d_ir, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (int c = 0; c < chains; ++c) {
        GK(gab_conv_create(&plan[c], T, B, L));
        GK(gab_conv_set_ir(plan[c], d_ir, nullptr));
        CK(hipStreamCreateWithFlags(&st[c], hipStreamNonBlocking));
        for (int i = 0; i < NIN; ++i) {
            float* p;
            CK(hipMalloc(&p, (size_t)T * B * 4));
            CK(hipMemcpy(p, h.data() + (size_t)i * 1000, (size_t)T * B * 4, hipMemcpyHostToDevice));
            in[c].push_back(p);
        }
        CK(hipMalloc(&out[c], (size_t)T * B * 4));
    }
    const int total = buffers + 500;
    // events[c][i]: recorded on stream c after its launch i
    std::vector<std::vector<hipEvent_t>> ev(chains, std::vector<hipEvent_t>(guard ? total : 0));
    for (auto& v : ev) for (auto& e : v) CK(hipEventCreateWithFlags(&e, evflags));
    std::vector<std::atomic<int>> recorded(chains);
    for (auto& a : recorded) a = 0;
    std::vector<uint32_t*> flag(chains, nullptr);
    if (value_ops)
        for (int c = 0; c < chains; ++c) {
            CK(hipExtMallocWithFlags((void**)&flag[c], 8, hipMallocSignalMemory));
            CK(hipMemset(flag[c], 0, 8));
        }

    auto run = [&](int first, int count) {
        std::vector<std::thread> th;
        for (int c = 0; c < chains; ++c)
            th.emplace_back([&, c]() {
                CK(hipSetDevice(0));
                for (int i = first; i < first + count; ++i) {
                    if (guard && !(one_way && c != 0)) {
                        const int o = (c + 1) % chains, dep = i - guard;
                        if (dep >= 0 && value_ops) {
                            CK(hipStreamWaitValue32(st[c], flag[o], (uint32_t)(dep + 1), hipStreamWaitValueGte, 0xffffffffu));
                        } else if (dep >= 0) {
                            while (recorded[o].load(std::memory_order_acquire) <= dep) std::this_thread::yield();
                            CK(hipStreamWaitEvent(st[c], ev[o][dep], 0));
                        }
                    }
                    GK(gab_conv_process(plan[c], in[c][i % NIN], out[c], GAB_CONV_STREAMING, (gab_stream_t)st[c]));
                    if (guard && !(one_way && c == 0)) {
                        if (value_ops) {
                            CK(hipStreamWriteValue32(st[c], flag[c], (uint32_t)(i + 1), 0));
                        } else {
                            CK(hipEventRecord(ev[c][i], st[c]));
                            recorded[c].store(i + 1, std::memory_order_release);
                        }
                    }
                }
            });
        for (auto& t : th) t.join();
    };
    run(0, 500);
    CK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    run(500, buffers);
    auto t1 = std::chrono::steady_clock::now();
    CK(hipDeviceSynchronize());
    auto t2 = std::chrono::steady_clock::now();
    double us = std::chrono::duration<double, std::micro>(t2 - t0).count() / buffers;
    double host_us = std::chrono::duration<double, std::micro>(t1 - t0).count() / buffers;
    printf("{\"value_ops\": %d, \"release_to_device\": %d, \"one_way\": %d, \"chains\": %d, \"guard\": %d, \"us_per_round\": %.3f, \"us_per_1024ch_buffer\": %.3f, \"host_queue_us_per_round\": %.3f}\n",
           (int)value_ops, (int)((evflags & hipEventReleaseToDevice) != 0), (int)one_way, chains, guard, us, us / chains, host_us);
    return 0;
}
