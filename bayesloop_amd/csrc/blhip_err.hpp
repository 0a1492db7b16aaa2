// Error handling shared by the translation units of libblhip: a failed HIP call or an internal inconsistency throws blerr::Fail, the
// C-ABI entry points (blhip.hip: guarded()) turn it into a return code + blhip_last_error().
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <set>
#include <string>
#include <typeinfo>
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

// ---- the kernel registry: every __global__ instantiation the library can launch, and how often this process has launched it --------
// A launch site names its kernel as a template argument (BL_LAUNCH below); the static members of blreg::Site<K> put ONE entry per
// instantiation into a list when the library is loaded -- so the list IS what the binary holds (a kernel is instantiated by its launch
// site and by nothing else), launched or not.  blhip_kernel_census (include/blhip.h) reports it; the -m gpu suite fails if an
// instantiation ships that no test has compared with the oracle (tests/test_kernel_census.py).
// (hidden: 1 400 weak Site<> symbols in the dynamic symbol table were 1.5 MB of the library; the list is shared inside the library all the same)
namespace blreg __attribute__((visibility("hidden"))) {

struct Entry {
    const void *fn;
    const char *mangled;                              // typeid(Site<K>).name(): the mangled "blreg::Site<&blc::chain_kernel<4, 1, ...>(blc::ChainParams)>"
    std::atomic<unsigned long long> launches;
    Entry *next;
};

inline std::atomic<Entry *> &head() {
    static std::atomic<Entry *> h{nullptr};
    return h;
}

inline Entry *add(Entry *e) {
    Entry *old = head().load(std::memory_order_relaxed);
    do e->next = old; while (!head().compare_exchange_weak(old, e, std::memory_order_release, std::memory_order_relaxed));
    return e;
}

template <auto K>
struct Site {
    static Entry entry;
    static Entry *const registered;                   // (its dynamic initialiser runs when the library is loaded)
};
template <auto K> Entry Site<K>::entry{reinterpret_cast<const void *>(K), typeid(Site<K>).name(), {0}, nullptr};
template <auto K> Entry *const Site<K>::registered = add(&Site<K>::entry);

template <auto K>
inline void hit() {
    (void)Site<K>::registered;                        // (odr-use: instantiates the registration)
    Site<K>::entry.launches.fetch_add(1, std::memory_order_relaxed);
}

}   // namespace blreg

// hipLaunchKernelGGL through the registry.  K: the kernel, in parentheses if its template arguments contain commas.
#define BL_LAUNCH(K, grid, block, lds, stream, ...)                            \
    do {                                                                       \
        blreg::hit<&K>();                                                      \
        hipLaunchKernelGGL(K, grid, block, lds, stream, __VA_ARGS__);          \
    } while (0)
