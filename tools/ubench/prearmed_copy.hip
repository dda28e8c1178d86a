// How long an engine upload takes to START: hipMemcpyAsync called now, against the same copy enqueued AHEAD of time behind a
// hipStreamWaitValue32 that the host releases with one store (the runtime's and the queue's work done before the data exists).
// Host clock from "go" (the call / the store) to the copy's completion event; 2 MiB from pinned memory at ~56 GB/s are ~37 us of it.
//   tools/ubench/bin/prearmed_copy [bytes] [iterations]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s at line %d\"}\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double pct(std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; }
int main(int argc, char** argv) {
    const size_t bytes = argc > 1 ? (size_t)atol(argv[1]) : (2u << 20);
    const int iters = argc > 2 ? atoi(argv[2]) : 300;
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    void *src = nullptr, *dst = nullptr;
    CK(hipHostMalloc(&src, bytes, hipHostMallocDefault));
    CK(hipMalloc(&dst, bytes));
    hipStream_t cs;
    CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    std::vector<double> plain, armed;
    for (int i = 0; i < iters; ++i) {
        const double t0 = now_us();
        CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, cs));
        CK(hipEventRecord(ev, cs));
        while (hipEventQuery(ev) == hipErrorNotReady) {}
        plain.push_back(now_us() - t0);
    }
    printf("{\"what\": \"hipMemcpyAsync called at go\", \"bytes\": %zu, \"p50_us\": %.1f, \"p95_us\": %.1f, \"min_us\": %.1f}\n", bytes, pct(plain, 0.5), pct(plain, 0.95), pct(plain, 0.0));
    if (!can) { printf("{\"what\": \"hipStreamWaitValue32 not supported on this device\"}\n"); return 0; }
    // the wait word: signal memory (what the API asks for); is it the host's to write?
    unsigned* sig = nullptr;
    CK(hipExtMallocWithFlags(reinterpret_cast<void**>(&sig), 8, hipMallocSignalMemory));
    hipPointerAttribute_t at;
    CK(hipPointerGetAttributes(&at, sig));
    printf("{\"signal_memory\": {\"type\": %d, \"host_pointer\": %d, \"device_pointer\": %d}}\n", (int)at.type, at.hostPointer != nullptr, at.devicePointer != nullptr);
    hipStream_t ws;                                   // the word is written through a stream op when the host may not store to it
    CK(hipStreamCreateWithFlags(&ws, hipStreamNonBlocking));
    const bool host_writes = at.hostPointer != nullptr;
    volatile unsigned* hsig = reinterpret_cast<volatile unsigned*>(at.hostPointer);
    CK(hipStreamWriteValue32(ws, sig, 0, 0));
    CK(hipStreamSynchronize(ws));
    for (int i = 0; i < iters; ++i) {
        const unsigned k = (unsigned)i + 1;
        CK(hipStreamWaitValue32(cs, sig, k, hipStreamWaitValueGte, 0xffffffffu));
        CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, cs));
        CK(hipEventRecord(ev, cs));
        std::this_thread::sleep_for(std::chrono::microseconds(300));     // everything is queued and waiting
        const double t0 = now_us();
        if (host_writes) *hsig = k; else CK(hipStreamWriteValue32(ws, sig, k, 0));
        double t1 = 0;
        for (;;) {
            if (hipEventQuery(ev) != hipErrorNotReady) { t1 = now_us(); break; }
            if (now_us() - t0 > 2e6) { printf("{\"error\": \"the armed copy did not start within 2 s (iteration %d)\"}\n", i); if (host_writes) *hsig = 0x7fffffffu; else (void)hipStreamWriteValue32(ws, sig, 0x7fffffffu, 0); (void)hipDeviceSynchronize(); return 1; }
        }
        armed.push_back(t1 - t0);
    }
    printf("{\"what\": \"copy enqueued ahead behind hipStreamWaitValue32, released by %s\", \"bytes\": %zu, \"p50_us\": %.1f, \"p95_us\": %.1f, \"min_us\": %.1f}\n",
           host_writes ? "a host store" : "hipStreamWriteValue32 on another stream", bytes, pct(armed, 0.5), pct(armed, 0.95), pct(armed, 0.0));
    (void)hipDeviceSynchronize();
    return 0;
}
