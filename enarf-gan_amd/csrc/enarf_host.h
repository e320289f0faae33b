// enarf_host.h - host-side error plumbing of the C ABI (thread-local last-error string).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>

namespace enarf {
namespace host {

inline char *err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
inline const char *last_error() { return err_buf(); }

inline int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}
// launch errors are reported as positive hipError_t values
inline int check_launch(const char *who) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "%s: launch failed: %s", who, hipGetErrorString(e));
    return 0;
}

}  // namespace host
}  // namespace enarf
