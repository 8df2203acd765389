// Error reporting and version of the C ABI.
#include "sg_common.h"

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

}  // namespace sg

extern "C" int sg_version(void) { return 1000; }

extern "C" const char* sg_last_error(void) { return sg::error_slot(); }
