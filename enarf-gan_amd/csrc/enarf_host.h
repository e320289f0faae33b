// enarf_host.h - host-side error plumbing of the C ABI (thread-local last-error string).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <mutex>

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

// Sticky status word of a device: 64 bytes of pinned host memory the device can write (allocated on first use, one per
// device, never freed). A kernel that has to give up ORs a bit into it (ENARF_STATUS_*); enarf_device_status reads it
// from the host without a synchronisation. Null when the allocation failed (the kernels then have nowhere to report).
inline unsigned int *status_word(bool device_side) {
    constexpr int kMaxDev = 64;
    static unsigned int *host_ptr[kMaxDev] = {nullptr};
    static unsigned int *dev_ptr[kMaxDev] = {nullptr};
    static bool tried[kMaxDev] = {false};
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!tried[dev]) {
        tried[dev] = true;
        void *h = nullptr, *d = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocDefault) == hipSuccess) {
            for (int i = 0; i < 16; ++i) reinterpret_cast<volatile unsigned int *>(h)[i] = 0u;
            if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
                host_ptr[dev] = reinterpret_cast<unsigned int *>(h);
                dev_ptr[dev] = reinterpret_cast<unsigned int *>(d);
            } else {
                (void)hipHostFree(h);
            }
        }
        (void)hipGetLastError();
    }
    return device_side ? dev_ptr[dev] : host_ptr[dev];
}

}  // namespace host
}  // namespace enarf
