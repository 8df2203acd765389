// Error reporting and version of the C ABI.
#include "sg_common.h"

#include <mutex>
#include <set>
#include <utility>

namespace sg {

char* error_slot() {
    static thread_local char text[512] = {0};
    return text;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_slot(), 512, fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what) {
    hipError_t err = hipGetLastError();
    if (err == hipSuccess) return SG_OK;
    return fail((int)err, "%s: launch failed: %s", what, hipGetErrorString(err));
}

hipError_t allow_large_lds(const void* kernel) {
    static std::mutex lock;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> guard(lock);
    if (done.count({kernel, dev})) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) done.insert({kernel, dev});
    return e;
}

}  // namespace sg

extern "C" int sg_version(void) { return 1000; }

extern "C" const char* sg_last_error(void) { return sg::error_slot(); }
