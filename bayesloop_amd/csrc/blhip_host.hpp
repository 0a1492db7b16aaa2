// Host-side infrastructure of libblhip: error handling, device / page-locked buffers, the context object.
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <limits>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/blhip.h"

#include "blhip_err.hpp"

namespace {

using blerr::Fail;
using blerr::fail;
using blerr::arm_kernel;

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    void ensure(size_t bytes) {
        if (bytes <= cap) return;
        release();
        HIPCHECK(hipMalloc(&p, bytes));
        cap = bytes;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// a DevBuf that is not a member of the context: released when the scope is left, also by an exception (HIPCHECK throws)
struct ScopedDevBuf : DevBuf {
    ScopedDevBuf() = default;
    ScopedDevBuf(const ScopedDevBuf &) = delete;
    ScopedDevBuf &operator=(const ScopedDevBuf &) = delete;
    ~ScopedDevBuf() { release(); }
};

// page-locked host staging: a hipMemcpyAsync to / from pageable memory is staged by the runtime behind blocking waits whose
// wake-up is quantised (10-ms steps seen on a 6 KB read-back: 20-27 ms per call instead of 0.5 ms)
struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    void ensure(size_t bytes) {
        if (bytes <= cap) return;
        release();
        HIPCHECK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
        cap = bytes;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

thread_local std::string g_create_error;

}  // namespace

struct blhip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[8] = {};
    // radius buckets of a batch are independent pipelines: one stream per bucket key, joined with events
    static constexpr int NBS = 18;       // (>= the keys of bucket_step)
    hipStream_t bstream[NBS] = {};
    hipEvent_t bev[NBS] = {};
    hipEvent_t fork_ev = nullptr, sync_ev = nullptr;
    std::string err;
    std::string name;
    std::map<std::string, double> opt;
    // reusable device buffers
    DevBuf state, post, psumF, psumB, redF, redB, meta, tables, likbuf, small, accum_own, stats;
    // kept posterior of the last fit
    bool post_valid = false;
    bool post_scaled = true;     // false: the kept rows still carry their raw sums; postinv holds 1 / sum per (chain, step)
    DevBuf postinv;
    int64_t post_chains = 0, post_T = 0, post_G = 0;
    int64_t post_row0 = 0, post_row1 = 0;        // rows of the kept sequence that still carry their raw sums
    int post_n0 = 1, post_n1 = 1, acc_n0 = 1, acc_n1 = 1;
    // accumulator
    bool acc_active = false, acc_final = false, acc_first = true;
    double *acc = nullptr;
    int64_t acc_T = 0, acc_G = 0, acc_folded = 0;
    double acc_logref = -std::numeric_limits<double>::infinity();
    blhip_timing timing = {};
    // carried states of streaming fits (BLHIP_CARRY / BLHIP_RESUME): slot -> (chains, G) normalised distributions
    struct Carry { DevBuf buf; int64_t chains = 0, G = 0; bool valid = false; std::vector<double> maxv; };
    std::map<int, Carry> carry;
    DevBuf mix, unit, databuf;
    PinBuf pinA;                 // weights of the average-posterior folds
    PinBuf pinF, pinB, pinS;     // host staging of the reduced sums (forward, backward) and of small read-backs
    int64_t mix_G = 0;
    // multi-GPU exchange (blhip_comm.hpp): RCCL communicator of this context's device, staging buffers
    void *comm = nullptr;        // ncclComm_t
    int comm_world = 1, comm_rank = 0;
    double comm_reduce_ms = 0.0;   // HIP-event time of the last blhip_comm_reduce_accum
    DevBuf commbuf;
    PinBuf pinC;
    // time-resident path (blhip_resident.hpp): halo strips / flags, and whether every tile was co-resident so far
    DevBuf resx;
    bool resident_ok = true;
    // a resident launch that gave up (its blocks were not all co-resident: another process held CUs) parks the resident paths; they
    // are tried again after `resident_retry_after` further fits, and the wait doubles with every give-up in a row (8, 16, ... 1024)
    int resident_retry_after = 8, resident_fits_since = 0;
    long long resident_giveups = 0;
    int resident_last_reason = 0;                  // BLHIP_FALLBACK_* of the last resident batch that fell back
    int num_cus = 0;
    // average posterior folded on a second stream while the next batch's forward pass runs (do_fit): second sequence buffer,
    // private copies of the per-batch weights, the stream and its events
    DevBuf post2, accw;
    DevBuf postpad;              // padded strip-major sequence of a chain-resident batch whose posteriors are handed out (de-padded into post)
    DevBuf hsrc;                 // the axis-1 pre-pass's output of one step (blhip_hwide.hpp)
    DevBuf p1d, p1w;             // hand-off buffers / weight table of the persistent 1-D kernel (blhip_persist1d.hpp)
    DevBuf lik1d;                // (T, n) likelihood table the chains of a 1-D batch share (blhip_chain1d.hpp)
    DevBuf accpart;              // partial accumulators of the fused fold (one per launch slot of the chain-resident kernel)
    // The partial accumulators are CARRIED from batch to batch of one blhip_fit call (round 6): the batches' weights share the reference of
    // the first one, the slots go into the average posterior once, after the call's last batch (fold_parts_kernel read 8 + 2 sequences of
    // T x G cells per batch: 4 x 4 ms of BASELINE C5's 226).  A batch that fails its checks after its backward pass has written the slots
    // poisons them: the call repeats the batches since `first_batch` on the launch-per-step kernels (do_fit).
    struct PartState {
        bool live = false;                  // the slots hold contributions that are not in the accumulator yet
        double ref = -std::numeric_limits<double>::infinity();       // ... weighted relative to this log weight
        double maxlw = -std::numeric_limits<double>::infinity();     // largest log weight among them
        int slots_init = 0;                 // slots written so far (a batch that uses more zeroes the new ones)
        int nfold = 0;                      // chains they hold
        int64_t first_batch = 0;            // first batch of the call with contributions in them
        int n0 = 0, n1 = 0, T = 0, n0p = 0, ax1 = 0; long long Gk = 0;      // layout (fold_parts_kernel's arguments)
    } part;
    DevBuf anchbuf;              // the anchors of the likelihood recurrence of a chain-resident batch, tabulated once per batch (blc::anchor_table_kernel)
    DevBuf axlik;                // ... its likelihood table of the even time steps (transposed layout): ceil(T / 2) x n0p^2 doubles
    DevBuf xch;                  // exchange buffers of the both-axes chain-resident kernel (blhip_chainax.hpp): [slot][2 step parities][Gk]
    hipStream_t cstream = nullptr;      // copy stream: the forward pass's sums travel to the host beside the backward pass (do_fit: late_copy)
    hipEvent_t cev[2] = {nullptr, nullptr};
    hipStream_t astream = nullptr;
    hipEvent_t aev_done[2] = {nullptr, nullptr};
    // the co-residency probe (blr::residency_probe_kernel): when it last ran (steady-clock seconds, < 0: never / run it again), what it saw
    DevBuf probebuf;
    double probe_last = -1.0;
    bool xcd_order_ok = true;
    bool probe_announced = false;
    // the prior the tables buffer holds (blhip_problem.prior_token): uploaded again only when the caller's token, the grid or the place changes
    unsigned long long prior_token = 0;
    const double *prior_dev = nullptr;
    long long prior_G = 0;

    double option(const char *k, double dflt) const {
        auto it = opt.find(k);
        return it == opt.end() ? dflt : it->second;
    }
};

namespace {

// Wait for a stream by polling an event: hipStreamSynchronize blocks on an interrupt whose wake-up is quantised (~10 ms
// steps measured on long waits: 15-25 ms of wall time per fit on top of a 365 ms device timeline).
struct Trace {
    bool on; std::chrono::steady_clock::time_point t0;
    explicit Trace(bool o) : on(o), t0(std::chrono::steady_clock::now()) {}
    void mark(const char *what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[blhip trace] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

void sync_stream(blhip_ctx *ctx, hipStream_t st) {
    HIPCHECK(hipEventRecord(ctx->sync_ev, st));
    for (;;) {
        const hipError_t e = hipEventQuery(ctx->sync_ev);
        if (e == hipSuccess) return;
        if (e != hipErrorNotReady) HIPCHECK(e);
        std::this_thread::yield();
    }
}

template <class T> T *carve(char *&cur, size_t count) {
    T *p = reinterpret_cast<T *>(cur);
    cur += ((count * sizeof(T) + 255) / 256) * 256;
    return p;
}
size_t carve_size(size_t bytes) { return ((bytes + 255) / 256) * 256; }

template <class F> int guarded(blhip_ctx *ctx, F &&f) {
    if (!ctx) return -1;
    try {
        f();
        return 0;
    } catch (const Fail &e) {
        ctx->err = e.msg;
    } catch (const std::exception &e) {
        ctx->err = e.what();
    } catch (...) {
        ctx->err = "unknown error";
    }
    return -1;
}

}  // namespace
