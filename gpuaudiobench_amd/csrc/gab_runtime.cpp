// gab_runtime.cpp — error text, version and device enumeration for the C ABI.
#include "gab_common.hpp"

#include <cstdlib>
#include <mutex>
#include <vector>

namespace gab {
namespace {
thread_local std::string g_last_error;
}
void set_last_error(const std::string& msg) { g_last_error = msg; }
const char* last_error() { return g_last_error.c_str(); }

// AMD_DIRECT_DISPATCH=0 (the runtime's separate dispatch thread) hung a process of this library in
// round 1 — which call it was blocked in was not recorded, so the cause is unknown and cannot be
// ruled out to be the library's own use of null-stream memsets beside non-blocking streams.  Until
// it is understood, plans refuse to be created in that mode instead of risking a hang.
int refuse_unsupported_runtime_mode(const char* who) {
    const char* v = getenv("AMD_DIRECT_DISPATCH");
    if (v && v[0] == '0' && v[1] == '\0') {
        set_last_error(std::string(who) + ": AMD_DIRECT_DISPATCH=0 is not supported (a process hang was observed "
                       "in that runtime mode); unset it");
        return GAB_ERR_UNSUPPORTED;
    }
    return GAB_OK;
}

namespace {
struct Resident { const void* owner; int device; ResidentKind kind; ResidentProbe running; };
std::mutex g_resident_mu;
std::vector<Resident>& residents() { static std::vector<Resident> v; return v; }
}
void resident_add(const void* owner, int device, ResidentKind kind, ResidentProbe running) {
    std::lock_guard<std::mutex> lock(g_resident_mu);
    residents().push_back({owner, device, kind, running});
}
void resident_remove(const void* owner) {
    std::lock_guard<std::mutex> lock(g_resident_mu);
    auto& v = residents();
    for (size_t i = 0; i < v.size();)
        if (v[i].owner == owner) v.erase(v.begin() + (long)i); else ++i;
}
int resident_running(int device, ResidentKind kind, const void* except) {
    std::lock_guard<std::mutex> lock(g_resident_mu);
    int n = 0;
    for (const Resident& r : residents())
        if (r.device == device && r.kind == kind && r.owner != except && r.running(r.owner)) ++n;
    return n;
}
}  // namespace gab

extern "C" {

int gab_version(void) { return 100; }   // 0.1.0

const char* gab_last_error(void) { return gab::last_error(); }

int gab_device_count(int* count) {
    if (!count) return gab::bad_arg("gab_device_count: null pointer");
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) {
        *count = 0;
        gab::set_last_error(std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
        return static_cast<int>(e);
    }
    return GAB_OK;
}

}  // extern "C"
