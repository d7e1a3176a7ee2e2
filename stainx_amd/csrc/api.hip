// Library-wide entry points: version, thread-local error text, launch checking.
#include "common.hpp"

namespace sx {

char* err_buf() {
    static thread_local char buf[512] = "";
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SX_ERR_LAUNCH, "HIP error after %s: %s", what, hipGetErrorString(e));
    return SX_OK;
}

}  // namespace sx

extern "C" int sx_version(void) { return SX_ABI_VERSION; }
extern "C" const char* sx_last_error_string(void) { return sx::err_buf(); }
