// gab_runtime.cpp — error text, version and device enumeration for the C ABI.
#include "gab_common.hpp"

namespace gab {
namespace {
thread_local std::string g_last_error;
}
void set_last_error(const std::string& msg) { g_last_error = msg; }
const char* last_error() { return g_last_error.c_str(); }
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
