// torn_word — does ONE hipMemcpyAsync from pinned memory write every destination word in one piece?
// A kernel watches a word of the destination while the engine copy lands and keeps the FIRST value it sees that is not the old one.
// If the runtime cuts the copy into engine packets of 4 MiB - 1 BYTES, the word at bytes 4 194 300 - 4 194 303 straddles the first
// boundary: its low three bytes arrive with one packet, its top byte with the next, and the watcher can see a mixture.
//   tools/ubench/bin/torn_word [bytes] [copies] [piece bytes: 0 = one copy] [source offset in bytes, a multiple of 4]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s line %d\"}\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void watch(const unsigned* dst, const unsigned* idx, int n, unsigned old, unsigned* first_seen, int limit) {
    const int i = blockIdx.x;
    if (i >= n || threadIdx.x) return;
    const unsigned* p = dst + idx[i];
    unsigned v = old;
    for (int t = 0; t < limit; ++t) {
        v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (v != old) break;
    }
    first_seen[i] = v;
}

int main(int argc, char** argv) {
    const size_t bytes = argc > 1 ? (size_t)atol(argv[1]) : (size_t(8) << 20);
    const int copies = argc > 2 ? atoi(argv[2]) : 2000;
    const size_t piece = argc > 3 ? (size_t)atol(argv[3]) : 0;
    const size_t src_off = argc > 4 ? (size_t)atol(argv[4]) : 0;
    const size_t words = bytes / 4;
    unsigned* src_base = nullptr;
    unsigned* dst = nullptr;
    CK(hipHostMalloc(reinterpret_cast<void**>(&src_base), bytes + 4096, hipHostMallocDefault));
    unsigned* const src = src_base + src_off / 4;
    CK(hipExtMallocWithFlags(reinterpret_cast<void**>(&dst), bytes, hipDeviceMallocFinegrained));
    const unsigned OLD = 0xffa5c3e1u;
    for (size_t i = 0; i < words; ++i) src[i] = 0x3e000000u + (unsigned)(i * 2654435761u >> 9);    // plain floats, top byte 0x3e/0x3f
    // watched words: the one straddling byte 4 194 303 (word 1 048 575), its neighbours, the next boundary (byte 8 388 606: word 2 097 151), controls
    std::vector<unsigned> idx = {1048574u, 1048575u, 1048576u, 2097150u, 2097151u, 2097152u, 1000u, 524288u, 1500000u};
    std::vector<unsigned> use;
    for (unsigned w : idx) if (w < words) use.push_back(w);
    const int n = (int)use.size();
    unsigned *d_idx, *d_seen;
    CK(hipMalloc(&d_idx, n * 4));
    CK(hipMalloc(&d_seen, n * 4));
    CK(hipMemcpy(d_idx, use.data(), n * 4, hipMemcpyHostToDevice));
    hipStream_t cs, ks;
    CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&ks, hipStreamNonBlocking));
    std::vector<int> torn(n, 0), late(n, 0);
    std::vector<unsigned> example(n, 0), seen(n);
    for (int c = 0; c < copies; ++c) {
        CK(hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(dst), (int)OLD, words));
        CK(hipDeviceSynchronize());
        watch<<<n, 64, 0, ks>>>(dst, d_idx, n, OLD, d_seen, 1 << 22);
        if (!piece) {
            CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, cs));
        } else {
            for (size_t off = 0; off < bytes; off += piece)
                CK(hipMemcpyAsync(reinterpret_cast<char*>(dst) + off, reinterpret_cast<const char*>(src) + off, bytes - off < piece ? bytes - off : piece, hipMemcpyHostToDevice, cs));
        }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(seen.data(), d_seen, n * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) {
            if (seen[i] == OLD) ++late[i];
            else if (seen[i] != src[use[i]]) { ++torn[i]; example[i] = seen[i]; }
        }
    }
    printf("{\"bytes\": %zu, \"copies\": %d, \"piece_bytes\": %zu, \"source_offset\": %zu, \"old\": \"%08x\", \"words\": [", bytes, copies, piece, src_off, OLD);
    for (int i = 0; i < n; ++i)
        printf("%s{\"word\": %u, \"first_byte\": %zu, \"torn\": %d, \"never_seen\": %d, \"new\": \"%08x\", \"example\": \"%08x\"}", i ? ", " : "", use[i], (size_t)use[i] * 4,
               torn[i], late[i], src[use[i]], example[i]);
    printf("]}\n");
    return 0;
}
