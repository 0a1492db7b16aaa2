// Multi-GPU exchange of a sharded hyper-study: RCCL over xGMI, bound DIRECTLY (no PyTorch, no MPI).
//
// What it replaces in the reference: HyperStudy.fit(nJobs > 1) fans the hyper-grid out with pool.map(self._parallelFit, ...)
// (bayesloop/core.py:1317-1326) and merges the sub-studies on the host: list concatenation of the per-point evidences and
// np.logaddexp of the average posteriors (core.py:1335-1340).  Here every GPU of the node runs its share of the chains without
// any communication; afterwards
//   * ONE ncclAllGather of the packed per-chain rows [logEvidence | localEvidence (T) | abort step] plus one trailer row per
//     rank (reference exponent and per-step sums of its accumulator), staged by the caller as plain doubles      -> blhip_comm_allgather
//   * only when posteriors were requested: a local rescale to the common reference exponent (its maximum is in the
//     gathered trailers, so no second collective is needed) and ONE ncclReduce(sum) of the (T, G) accumulator to the
//     root, in place in HBM                                                                                   -> blhip_comm_reduce_accum
// librccl.so is dlopen'ed on first use: single-GPU users never load it, and the library does not link against it.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "blhip_host.hpp"

namespace {

struct RcclApi {
    void *handle = nullptr;
    std::string path;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclReduceScatter) ReduceScatter = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;

    template <class F> void sym(F &f, const char *name) {
        f = reinterpret_cast<F>(dlsym(handle, name));
        if (!f) fail("%s does not export %s", path.c_str(), name);
    }

    void load() {
        if (handle) return;
        std::vector<std::string> cands;
        if (const char *e = std::getenv("BLHIP_RCCL_LIBRARY")) cands.push_back(e);
        cands.push_back("librccl.so.1");
        if (const char *r = std::getenv("ROCM_PATH")) cands.push_back(std::string(r) + "/lib/librccl.so.1");
        cands.push_back("/opt/rocm/lib/librccl.so.1");
        cands.push_back("librccl.so");
        std::string errs;
        for (const auto &c : cands) {
            handle = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (handle) { path = c; break; }
            const char *e = dlerror();
            errs += "\n  " + c + ": " + (e ? e : "?");
        }
        if (!handle) fail("cannot load RCCL (needed for multi-GPU hyper-studies):%s", errs.c_str());
        sym(GetUniqueId, "ncclGetUniqueId");
        sym(CommInitRank, "ncclCommInitRank");
        sym(CommDestroy, "ncclCommDestroy");
        sym(AllGather, "ncclAllGather");
        sym(AllReduce, "ncclAllReduce");
        sym(Reduce, "ncclReduce");
        sym(ReduceScatter, "ncclReduceScatter");
        sym(Send, "ncclSend");
        sym(Recv, "ncclRecv");
        sym(GroupStart, "ncclGroupStart");
        sym(GroupEnd, "ncclGroupEnd");
        sym(GetErrorString, "ncclGetErrorString");
        sym(GetVersion, "ncclGetVersion");
    }
};

RcclApi &rccl() {
    static RcclApi api;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    api.load();
    return api;
}

#define RCCLCHECK(expr)                                                                                          \
    do {                                                                                                         \
        ncclResult_t r_ = (expr);                                                                                \
        if (r_ != ncclSuccess) fail("%s failed: %s (%s:%d)", #expr, rccl().GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

ncclComm_t comm_of(blhip_ctx *ctx) {
    if (!ctx->comm) fail("no communicator on this context (blhip_comm_init)");
    return reinterpret_cast<ncclComm_t>(ctx->comm);
}

}  // namespace

extern "C" {

int blhip_comm_unique_id(void *id_out) {
    try {
        if (!id_out) fail("id_out is NULL");
        static_assert(sizeof(ncclUniqueId) == BLHIP_UNIQUE_ID_BYTES, "unique id size");
        ncclUniqueId id;
        RCCLCHECK(rccl().GetUniqueId(&id));
        std::memcpy(id_out, &id, sizeof id);
        return 0;
    } catch (const Fail &e) {
        g_create_error = e.msg;
    } catch (...) {
        g_create_error = "unknown error in blhip_comm_unique_id";
    }
    return -1;
}

int blhip_comm_init(blhip_ctx *ctx, const void *unique_id, int world, int rank) {
    return guarded(ctx, [&] {
        if (!unique_id) fail("unique_id is NULL");
        if (world < 1 || rank < 0 || rank >= world) fail("blhip_comm_init: rank %d of %d", rank, world);
        if (ctx->comm) fail("this context already has a communicator (blhip_comm_destroy first)");
        HIPCHECK(hipSetDevice(ctx->device));
        ncclUniqueId id;
        std::memcpy(&id, unique_id, sizeof id);
        ncclComm_t c = nullptr;
        RCCLCHECK(rccl().CommInitRank(&c, world, id, rank));
        ctx->comm = c;
        ctx->comm_world = world;
        ctx->comm_rank = rank;
    });
}

int blhip_comm_info(blhip_ctx *ctx, int *world, int *rank, int *rccl_version) {
    return guarded(ctx, [&] {
        if (world) *world = ctx->comm ? ctx->comm_world : 1;
        if (rank) *rank = ctx->comm ? ctx->comm_rank : 0;
        if (rccl_version) {
            *rccl_version = 0;
            if (ctx->comm) RCCLCHECK(rccl().GetVersion(rccl_version));
        }
    });
}

int blhip_comm_allgather(blhip_ctx *ctx, const double *host_in, int64_t count, double *host_out) {
    return guarded(ctx, [&] {
        if (!host_in || !host_out || count < 1) fail("blhip_comm_allgather: bad arguments");
        ncclComm_t c = comm_of(ctx);
        HIPCHECK(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        const size_t W = (size_t)ctx->comm_world, nb = (size_t)count * 8;
        ctx->commbuf.ensure((W + 1) * nb);
        ctx->pinC.ensure((W + 1) * nb);
        double *d_send = ctx->commbuf.as<double>(), *d_recv = d_send + count;
        double *h_send = ctx->pinC.as<double>(), *h_recv = h_send + count;
        std::memcpy(h_send, host_in, nb);
        HIPCHECK(hipMemcpyAsync(d_send, h_send, nb, hipMemcpyHostToDevice, st));
        RCCLCHECK(rccl().AllGather(d_send, d_recv, (size_t)count, ncclDouble, c, st));
        HIPCHECK(hipMemcpyAsync(h_recv, d_recv, W * nb, hipMemcpyDeviceToHost, st));
        sync_stream(ctx, st);
        std::memcpy(host_out, h_recv, W * nb);
    });
}

int blhip_comm_allreduce(blhip_ctx *ctx, double *host_inout, int64_t count, int op) {
    return guarded(ctx, [&] {
        if (!host_inout || count < 1) fail("blhip_comm_allreduce: bad arguments");
        if (op < 0 || op > 2) fail("blhip_comm_allreduce: op must be BLHIP_SUM, BLHIP_MAX or BLHIP_MIN");
        ncclComm_t c = comm_of(ctx);
        HIPCHECK(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        const size_t nb = (size_t)count * 8;
        ctx->commbuf.ensure(nb);
        ctx->pinC.ensure(nb);
        double *d = ctx->commbuf.as<double>(), *h = ctx->pinC.as<double>();
        std::memcpy(h, host_inout, nb);
        HIPCHECK(hipMemcpyAsync(d, h, nb, hipMemcpyHostToDevice, st));
        const ncclRedOp_t rop = op == BLHIP_SUM ? ncclSum : (op == BLHIP_MAX ? ncclMax : ncclMin);
        RCCLCHECK(rccl().AllReduce(d, d, (size_t)count, ncclDouble, rop, c, st));
        HIPCHECK(hipMemcpyAsync(h, d, nb, hipMemcpyDeviceToHost, st));
        sync_stream(ctx, st);
        std::memcpy(host_inout, h, nb);
    });
}

int blhip_comm_reduce_accum(blhip_ctx *ctx, int root) {
    return guarded(ctx, [&] {
        ncclComm_t c = comm_of(ctx);
        if (!ctx->acc_active) fail("no active accumulator");
        if (ctx->acc_final) fail("blhip_comm_reduce_accum: the accumulator is already finalised");
        if (root < 0 || root >= ctx->comm_world) fail("blhip_comm_reduce_accum: root %d of %d ranks", root, ctx->comm_world);
        HIPCHECK(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        const size_t n = (size_t)ctx->acc_T * (size_t)ctx->acc_G;
        const int W = ctx->comm_world;
        // option comm_reduce_mode: 0 = ONE ncclReduce to the root (a ring: every link carries the whole buffer once);
        // 1 = reduce-scatter over time slices + the slices sent to the root (xGMI is point-to-point: the root then takes in
        // (W - 1) / W of ONE buffer over W - 1 links instead of W - 1 whole buffers); needs T divisible by the world size
        const int mode = (int)ctx->option("comm_reduce_mode", 0.0);
        HIPCHECK(hipEventRecord(ctx->ev[4], st));
        if (mode == 1 && W > 1 && ctx->acc_T % W == 0) {
            const size_t cnt = n / (size_t)W;
            // in place: rank r's slice of the sums lands where its own slice lives
            RCCLCHECK(rccl().ReduceScatter(ctx->acc, ctx->acc + (size_t)ctx->comm_rank * cnt, cnt, ncclDouble, ncclSum, c, st));
            RCCLCHECK(rccl().GroupStart());
            if (ctx->comm_rank == root) {
                for (int r = 0; r < W; ++r)
                    if (r != root) RCCLCHECK(rccl().Recv(ctx->acc + (size_t)r * cnt, cnt, ncclDouble, r, c, st));
            } else {
                RCCLCHECK(rccl().Send(ctx->acc + (size_t)ctx->comm_rank * cnt, cnt, ncclDouble, root, c, st));
            }
            RCCLCHECK(rccl().GroupEnd());
        } else {
            // in place: on the root the sum replaces its own share; the other ranks' buffers keep their (spent) shares
            RCCLCHECK(rccl().Reduce(ctx->acc, ctx->acc, n, ncclDouble, ncclSum, root, c, st));
        }
        HIPCHECK(hipEventRecord(ctx->ev[5], st));
        sync_stream(ctx, st);
        float ms = 0;
        HIPCHECK(hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5]));
        ctx->comm_reduce_ms = ms;
    });
}

int blhip_comm_timing(blhip_ctx *ctx, double *reduce_ms) {
    if (!ctx) return -1;
    if (reduce_ms) *reduce_ms = ctx->comm_reduce_ms;
    return 0;
}

// ---- several GPUs driven by ONE process (HyperStudy.fit(nJobs = N): one context per device, one host thread per context): the
//      accumulators are merged with peer copies over xGMI, no RCCL.  The caller (bayesloop_amd/dist.py: LocalGroup) brings every
//      accumulator to the common reference exponent first, synchronises every context and orders the calls with host barriers. -------
namespace {
// How a slice of another context's accumulator gets here (reported in blhip_timing.peer_copy_path of the DESTINATION context):
//   BLHIP_PEER_SAME_DEVICE  both contexts on one GPU (the 1-GPU test configuration): a device-to-device copy
//   BLHIP_PEER_DIRECT       hipMemcpyPeerAsync over xGMI, peer access enabled
//   BLHIP_PEER_HOST_STAGED  no peer access between the two devices (hipDeviceCanAccessPeer == 0), a peer copy that FAILED, or option
//                           peer_copy_mode = 1 (tests: the branch runs on one GPU too): an explicit copy through page-locked host
//                           memory, 64 MiB at a time -- slow, never wrong; one line on stderr the first time
std::atomic<bool> g_peer_staging_announced{false};
std::atomic<bool> g_peer_copy_broken{false};           // a hipMemcpyPeerAsync failed: the process stays on the staged path
bool peer_direct(blhip_ctx *dst, int self_dev, int peer_dev) {
    if (self_dev == peer_dev) return true;
    if (dst->option("peer_copy_mode", 0.0) == 1.0 || g_peer_copy_broken.load()) return false;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, self_dev, peer_dev) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (!can) return false;
    const hipError_t e = hipDeviceEnablePeerAccess(peer_dev, 0);
    (void)hipGetLastError();
    return e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
}
void staged_copy(blhip_ctx *dst, double *to, blhip_ctx *src, const double *from, size_t bytes, hipStream_t s, const char *why) {
    if (!g_peer_staging_announced.exchange(true) && dst->option("quiet", 0.0) == 0.0)
        std::fprintf(stderr, "[blhip] accumulator merge between devices %d and %d goes through host memory (%s): slower than xGMI, same result\n",
                     src->device, dst->device, why);
    constexpr size_t CHUNK = 64u << 20;
    void *host = nullptr;
    HIPCHECK(hipHostMalloc(&host, std::min(bytes, CHUNK), hipHostMallocDefault));
    try {
        for (size_t done = 0; done < bytes; done += CHUNK) {
            const size_t n = std::min(CHUNK, bytes - done);
            HIPCHECK(hipSetDevice(src->device));
            HIPCHECK(hipMemcpy(host, reinterpret_cast<const char *>(from) + done, n, hipMemcpyDeviceToHost));
            HIPCHECK(hipSetDevice(dst->device));
            HIPCHECK(hipMemcpyAsync(reinterpret_cast<char *>(to) + done, host, n, hipMemcpyHostToDevice, s));
            HIPCHECK(hipStreamSynchronize(s));          // (the staging block is reused)
        }
    } catch (...) { (void)hipSetDevice(dst->device); (void)hipHostFree(host); throw; }
    HIPCHECK(hipSetDevice(dst->device));
    HIPCHECK(hipHostFree(host));
}
void check_peers(blhip_ctx *dst, blhip_ctx *const *srcs, int n) {
    if (!dst->acc_active || dst->acc_final) fail("peer merge: the destination has no open accumulator");
    if (n < 0 || n > blhip_ctx::NBS) fail("peer merge: %d sources (at most %d)", n, blhip_ctx::NBS);
    for (int i = 0; i < n; ++i) {
        if (!srcs || !srcs[i] || srcs[i] == dst) fail("peer merge: bad source context %d", i);
        if (!srcs[i]->acc_active || srcs[i]->acc_T != dst->acc_T || srcs[i]->acc_G != dst->acc_G) fail("peer merge: accumulator shapes differ");
        if (srcs[i]->acc_logref != dst->acc_logref) fail("peer merge: the accumulators are not at a common reference exponent (blhip_accum_rescale)");
    }
}
static __global__ void peer_add_kernel(double *__restrict__ acc, const double *__restrict__ stage, int n, long long count) {
    // acc[i] += stage[0][i] + stage[1][i] + ... in list order (the same sum on every run)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
        double v = acc[i];
        for (int k = 0; k < n; ++k) v += __builtin_nontemporal_load(stage + (long long)k * count + i);
        acc[i] = v;
    }
}
void copy_rows(blhip_ctx *dst, double *to, blhip_ctx *src, long long off, size_t bytes, hipStream_t s) {
    const bool forced = dst->option("peer_copy_mode", 0.0) == 1.0;
    if (src->device == dst->device && !forced) {
        HIPCHECK(hipMemcpyAsync(to, src->acc + off, bytes, hipMemcpyDeviceToDevice, s));
        dst->timing.peer_copy_path = std::max(dst->timing.peer_copy_path, (int32_t)BLHIP_PEER_SAME_DEVICE);
        return;
    }
    if (!forced && peer_direct(dst, dst->device, src->device)) {
        const hipError_t e = hipMemcpyPeerAsync(to, dst->device, src->acc + off, src->device, bytes, s);
        if (e == hipSuccess) { dst->timing.peer_copy_path = std::max(dst->timing.peer_copy_path, (int32_t)BLHIP_PEER_DIRECT); return; }
        (void)hipGetLastError();
        g_peer_copy_broken.store(true);
        staged_copy(dst, to, src, src->acc + off, bytes, s, hipGetErrorString(e));
    } else {
        staged_copy(dst, to, src, src->acc + off, bytes, s, forced ? "option peer_copy_mode = 1" : "no peer access between the devices");
    }
    dst->timing.peer_copy_path = (int32_t)BLHIP_PEER_HOST_STAGED;
}
}  // namespace

int blhip_accum_peer_reduce(blhip_ctx *dst, blhip_ctx *const *srcs, int n_srcs, int64_t row0, int64_t row1) {
    return guarded(dst, [&] {
        check_peers(dst, srcs, n_srcs);
        if (row0 < 0 || row1 > dst->acc_T || row0 > row1) fail("peer reduce: rows [%lld, %lld) of %lld", (long long)row0, (long long)row1, (long long)dst->acc_T);
        if (n_srcs == 0 || row0 == row1) return;
        HIPCHECK(hipSetDevice(dst->device));
        hipStream_t st = dst->stream;
        const long long cnt = (long long)(row1 - row0) * dst->acc_G, off = (long long)row0 * dst->acc_G;
        dst->commbuf.ensure((size_t)n_srcs * (size_t)cnt * 8);
        double *stage = dst->commbuf.as<double>();
        HIPCHECK(hipEventRecord(dst->fork_ev, st));
        for (int i = 0; i < n_srcs; ++i) {               // one stream per source: the copies run over different xGMI links at once
            hipStream_t s = dst->bstream[i];
            HIPCHECK(hipStreamWaitEvent(s, dst->fork_ev, 0));
            copy_rows(dst, stage + (long long)i * cnt, srcs[i], off, (size_t)cnt * 8, s);
            HIPCHECK(hipEventRecord(dst->bev[i], s));
            HIPCHECK(hipStreamWaitEvent(st, dst->bev[i], 0));
        }
        BL_LAUNCH(peer_add_kernel, dim3(2048), dim3(256), 0, st, dst->acc + off, stage, n_srcs, cnt);
        HIPCHECK(hipGetLastError());
        sync_stream(dst, st);
    });
}

int blhip_accum_peer_gather(blhip_ctx *dst, blhip_ctx *const *srcs, int n_srcs, const int64_t *row0, const int64_t *row1) {
    return guarded(dst, [&] {
        check_peers(dst, srcs, n_srcs);
        if (n_srcs == 0) return;
        if (!row0 || !row1) fail("peer gather: row bounds are NULL");
        HIPCHECK(hipSetDevice(dst->device));
        hipStream_t st = dst->stream;
        HIPCHECK(hipEventRecord(dst->fork_ev, st));
        for (int i = 0; i < n_srcs; ++i) {
            if (row0[i] < 0 || row1[i] > dst->acc_T || row0[i] > row1[i]) fail("peer gather: rows [%lld, %lld) of %lld", (long long)row0[i], (long long)row1[i], (long long)dst->acc_T);
            if (row0[i] == row1[i]) continue;
            hipStream_t s = dst->bstream[i];
            HIPCHECK(hipStreamWaitEvent(s, dst->fork_ev, 0));
            const long long off = (long long)row0[i] * dst->acc_G;
            copy_rows(dst, dst->acc + off, srcs[i], off, (size_t)(row1[i] - row0[i]) * (size_t)dst->acc_G * 8, s);
            HIPCHECK(hipEventRecord(dst->bev[i], s));
            HIPCHECK(hipStreamWaitEvent(st, dst->bev[i], 0));
        }
        sync_stream(dst, st);
    });
}

int blhip_comm_destroy(blhip_ctx *ctx) {
    return guarded(ctx, [&] {
        if (!ctx->comm) return;
        HIPCHECK(hipSetDevice(ctx->device));
        sync_stream(ctx, ctx->stream);
        ncclComm_t c = reinterpret_cast<ncclComm_t>(ctx->comm);
        ctx->comm = nullptr;
        ctx->comm_world = 1;
        ctx->comm_rank = 0;
        RCCLCHECK(rccl().CommDestroy(c));
    });
}

}  // extern "C"
