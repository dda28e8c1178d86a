// gab_common.hpp — shared host-side helpers for the HIP kernels and the C ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdio>
#include <stdexcept>
#include <string>

#include "gab_c_api.h"

namespace gab {

// Thread-local text of the last failure, surfaced through gab_last_error().
void set_last_error(const std::string& msg);
const char* last_error();
// GAB_OK, or GAB_ERR_UNSUPPORTED with the reason set (gab_runtime.cpp)
int refuse_unsupported_runtime_mode(const char* who);

// Mirrors the reference's CUDA_CHECK contract (cuda/bench_utils.cuh:248-254):
// a failing runtime call becomes a std::runtime_error carrying the call text.
inline void check_hip(hipError_t e, const char* what) {
    if (e != hipSuccess)
        throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

#define GAB_HIP_CHECK(call) ::gab::check_hip((call), #call)

// C-ABI wrapper: no exception may cross an extern "C" boundary.
template <class F>
inline int guarded(F&& body) noexcept {
    try {
        return body();
    } catch (const std::invalid_argument& e) {
        set_last_error(e.what());
        return GAB_ERR_INVALID_ARG;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return GAB_ERR_RUNTIME;
    } catch (...) {
        set_last_error("unknown exception");
        return GAB_ERR_RUNTIME;
    }
}

inline int bad_arg(const char* msg) {
    set_last_error(msg);
    return GAB_ERR_INVALID_ARG;
}

inline int launch_status(const char* kernel) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error(std::string(kernel) + " launch failed: " + hipGetErrorString(e));
        return static_cast<int>(e);
    }
    return GAB_OK;
}

inline hipStream_t as_stream(gab_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Launches that STAY on a device (gab_runtime.cpp): the keep-warm launch and the doorbell-fed engine.  They exclude each
// other — the engine's workgroup fills a compute unit's register files, so eight keep-warm waves keep eight of its
// workgroups out until they end (profiles/r05_paced_keep_warm.txt: first buffer 484 ms) — and the library makes every one
// of them, so it knows: whoever would start the second is refused at the call.
enum ResidentKind { kResidentKeepWarm = 0, kResidentEngine = 1 };
using ResidentProbe = bool (*)(const void* owner);       // is the owner's launch on the device right now?
void resident_add(const void* owner, int device, ResidentKind kind, ResidentProbe running);
void resident_remove(const void* owner);
// how many launches of `kind` are running on `device`, not counting `except` (may be null)
int resident_running(int device, ResidentKind kind, const void* except);

constexpr int kWave = 64;           // gfx950 wavefront
constexpr int kNumXCD = 8;          // MI355X: 8 XCDs, blocks are dealt round-robin over them

// Blocks b and b+8 share an XCD (observed dispatch order; speed only, never
// correctness).  Remap so that each XCD works on one contiguous range of
// logical work items and their shared output lines meet in one L2.
__host__ __device__ inline int xcd_contiguous(int bid, int nblocks) {
    int per = nblocks / kNumXCD;
    if (per * kNumXCD != nblocks) return bid;   // only bijective when divisible
    return (bid % kNumXCD) * per + bid / kNumXCD;
}

}  // namespace gab
