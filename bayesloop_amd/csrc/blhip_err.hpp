// Error handling shared by the translation units of libblhip: a failed HIP call or an internal inconsistency throws blerr::Fail, the
// C-ABI entry points (blhip.hip: guarded()) turn it into a return code + blhip_last_error().
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <set>
#include <string>
#include <utility>

// (named namespace, inline: the library is several translation units -- build.py -- and an error thrown by one is caught by another)
namespace blerr {

struct Fail {
    std::string msg;
};

[[noreturn]] inline void fail(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Fail{buf};
}

}   // namespace blerr

#define HIPCHECK(expr)                                                                                        \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) blerr::fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

namespace blerr {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting: a process that drives several GPUs (HyperStudy.fit(nJobs = N):
// one context and one host thread per device) has to arm every kernel on every device it launches it on
inline void arm_kernel(const void *fn, int bytes = 160 * 1024) {
    static std::mutex mu;
    static std::set<std::pair<int, const void *>> armed;
    int dev = 0;
    HIPCHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (armed.count({dev, fn})) return;
    HIPCHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    armed.insert({dev, fn});
}

}   // namespace blerr

