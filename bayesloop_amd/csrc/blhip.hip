// libblhip: C-ABI (include/blhip.h) + host orchestration of the forward-backward recursion on one MI355X.
//
// The host side owns what the reference's Study.fit loop owns (bayesloop/core.py:330-486): the order of steps, the
// evidence bookkeeping and the per-step transition program; the arithmetic on the grid runs in the HIP kernels of
// blhip_kernels.hpp / blhip_fast.hpp.
#include <hip/hip_runtime.h>

#include <cxxabi.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <exception>
#include <map>
#include <memory>
#include <unordered_map>
#include <mutex>
#include <set>
#include <string>
#include <system_error>
#include <chrono>
#include <thread>
#include <vector>

#include "../../include/blhip.h"
#include "blhip_kernels.hpp"
#include "blhip_fast.hpp"
#include "blhip_mfma.hpp"
#include "blhip_hwide.hpp"
#include "blhip_fused1d.hpp"
#include "blhip_chain1d.hpp"
#include "blhip_persist1d.hpp"
#include "blhip_resident.hpp"
#include "blhip_chainres.hpp"
#include "blhip_chain_launch.hpp"
#include "blhip_nd.hpp"

using namespace blk;

#include "blhip_host.hpp"
#include "blhip_comm.hpp"

namespace {

#include "blhip_program.hpp"   // TapTable, Geometry, validate, build_records, build_program
#include "blhip_launch.hpp"    // Tile, launch_* (the launch tables), plan_resident
#include "blhip_batch.hpp"     // plan_batches, upload_tables / _metadata, chains_per_batch, plan_geometry, plan_chainres, folds, results
#include "blhip_book.hpp"      // resident_unlag / chain_unlag, forward_ / backward_bookkeeping, account
#include "blhip_fit_nd.hpp"       // do_fit_nd: grids with 3 and 4 parameters
#include "blhip_fit_paths.hpp"    // BatchEnv, ResidentRun, ChainRun: the resident paths of a batch

void do_fit(blhip_ctx *ctx, const blhip_problem *p_in, int64_t n_chains, const double *op_values,
            const double *log_w, uint32_t flags, blhip_result *res) {
    Trace tr(ctx->option("trace", 0.0) != 0.0);
    validate(p_in, n_chains, op_values);
    if (p_in->ndim > 2) { ctx->prior_token = 0; do_fit_nd(ctx, p_in, n_chains, op_values, log_w, flags, res); return; }      // (that path lays the tables buffer out its own way)
    // closed-form models without an in-kernel likelihood: their (T, G) table is built on the device, the step kernels
    // then see a tabulated likelihood
    blhip_problem p_local = *p_in;
    const int table_model = (p_in->obs_model >= BLHIP_OM_BERNOULLI && p_in->obs_model <= BLHIP_OM_SCALED_AR1) ? p_in->obs_model : 0;
    if (table_model) p_local.obs_model = BLHIP_OM_TABLE;
    const blhip_problem *p = &p_local;
    HIPCHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const FitFlags ff = decode_flags(ctx, p, flags, log_w);
    const bool evidence_only = ff.evidence_only, forward_only = ff.forward_only, full = ff.full, keep = ff.keep,
               accumulate = ff.accumulate, resume = ff.resume, carry = ff.carry;
    const int64_t T = p->T;

    Geometry g{};
    if (p->ndim == 1) { g.n0 = 1; g.n1 = (int)p->n[0]; g.axis_map[0] = 1; g.axis_map[1] = 1; }
    else { g.n0 = (int)p->n[0]; g.n1 = (int)p->n[1]; g.axis_map[0] = 0; g.axis_map[1] = 1; }
    g.G = (long long)g.n0 * g.n1;
    const long long G = g.G;
    if (accumulate && (ctx->acc_T != T || ctx->acc_G != G)) fail("accumulator shape mismatch");
    double dV = 1.0;
    for (int k = 0; k < p->ndim; ++k) dV *= p->lattice[k];

    ctx->post_valid = false;
    ctx->part = blhip_ctx::PartState{};            // (carried partial accumulators live inside one call)
    ctx->timing = blhip_timing{};
    ctx->resident_last_reason = 0;
    // a context whose resident launch once gave up tries the resident paths again after a while (one hiccup -- another process
    // holding CUs -- must not cost 1.4 - 2 x for the life of the process); the wait doubles with every give-up in a row
    if (!ctx->resident_ok && ++ctx->resident_fits_since >= ctx->resident_retry_after) {
        ctx->resident_ok = true;
        ctx->resident_fits_since = 0;
        ctx->probe_last = -1.0;                      // (re-armed: ask the chip first)
    }
    probe_residency(ctx);
    ctx->timing.xcd_order = ctx->xcd_order_ok ? 1 : 0;

    // ---- shared tables -> HBM ---------------------------------------------------------------------------------------------------
    const DeviceTables DT = upload_tables(ctx, p, g, ff, table_model);
    double *const d_m0 = DT.m0, *const d_m1 = DT.m1, *const d_colA = DT.colA, *const d_colB = DT.colB, *const d_rec = DT.rec;
    double *const d_prior = DT.prior, *const d_reset = DT.reset, *const d_uniform = DT.uniform, *const d_indep = DT.indep, *const d_lik = DT.lik;
    const int rec_len = DT.rec_len, d = DT.d;
    tr.mark("tables + H2D");

    // ---- memory plan ------------------------------------------------------------------------------------------------------------
    // The average posterior of a hyper-study is folded batch by batch (core.py:1358-1366): a pass over the batch's whole sequence at
    // the memory roof (C4: 2 x 24 ms of a 300 ms fit).  Option accum_overlap = 1 (default 0): two sequence buffers, the fold of batch b
    // runs on a second stream beside the forward pass of batch b + 1 (~4 batches instead of as few as fit).  Measured on C4: no
    // gain -- queued behind the pass's launches the fold is not started before they have finished (rocprofv3 time line), queued
    // ahead of them it slows the pass by about its own duration (13 ms fold: forward 25 -> 35 ms per batch; 1.15e11 either way).
    const bool overlap_acc = accumulate && full && ctx->option("accum_overlap", 0.0) != 0.0 && n_chains >= 64;
    if (!overlap_acc) ctx->post2.release();         // (a fit without the second stream gets the memory back)
    int64_t Bmax = chains_per_batch(ctx, p, g, ff, n_chains, overlap_acc ? 2 : 1);
    if (overlap_acc) Bmax = std::min<int64_t>(Bmax, std::max<int64_t>(32, ((n_chains + 3) / 4 + 31) / 32 * 32));
    // ---- batches: at most Bmax chains each, cut where the axis-0 radius bucket changes.  A bucket cut by a batch boundary becomes
    //      two under-filled launches per step (round 1: the 107-chain radius-24 bucket of the C4 study ran as 36 + 71 chains at
    //      4.1 TB/s); hyper-grids are usually monotone in the random-walk width, so contiguous cuts suffice.
    // (cuts on radius-bucket boundaries serve the launch-per-step kernels; grids the chain-resident kernel takes keep whole launches)
    const bool chain_shape = p->ndim == 2 && (p->obs_model == BLHIP_OM_GAUSSIAN || (p->obs_model == BLHIP_OM_TABLE && g.n0 <= 512)) && chain_rows_ok(g.n0) && g.n1 <= 16 * blc::MAX_STRIPS &&
                             ctx->option("chain_resident", 1.0) != 0.0 && ctx->resident_ok;
    std::vector<int64_t> batch_start = plan_batches(p, n_chains, op_values, Bmax, !overlap_acc && !chain_shape);
    if (overlap_acc) {                     // (equal batches, multiples of 32 chains: whole launches of the chain-resident kernel)
        batch_start.clear();
        for (int64_t c = 0; c < n_chains; c += Bmax) batch_start.push_back(c);
        batch_start.push_back(n_chains);
    }
    // (the cut: 40; grids the chain-resident kernels take: 80)
    const bool chain_wide = chain_shape && p->obs_model == BLHIP_OM_GAUSSIAN && ctx->option("chain_wide", 1.0) != 0.0;
    if (!overlap_acc && !ff.keep && !ff.resume && !ff.carry && ctx->option("wide_v", 1.0) != 0.0)
        split_wide_axis0(p, n_chains, op_values, batch_start, chain_wide ? CHAIN_R0_MAX : FAST_R0_MAX);
    // Batches of thousands of chains (1-D studies: the published break-point study runs 6 x 3900): the per-(step, chain) program of batch
    // bi + 1 is built by a second host thread while batch bi runs -- all but the FIRST one's, 3.9 ms of a 42-ms fit with the GPU idle.  A
    // short first batch (1024 chains: 0.7 ms of program) starts the pipeline earlier.
    if (!overlap_acc && batch_start.size() >= 2 && batch_start[1] - batch_start[0] >= 2048 && !ff.keep && !ff.resume && !ff.carry &&
        ctx->option("short_first_batch", 1.0) != 0.0)
        batch_start.insert(batch_start.begin() + 1, batch_start[0] + 1024);
    const int64_t nbatch = (int64_t)batch_start.size() - 1;
    ctx->timing.batches = nbatch;

    tr.mark("memory budget");
    hipEvent_t *ev = ctx->ev;
    HIPCHECK(hipEventRecord(ev[6], st));

    double *redF = nullptr, *redB = nullptr;   // reduced sums on the host (page-locked staging of the context)
    std::vector<hipEvent_t> fold_ev;           // overlapped folds: start / end events of every batch (timing)
    FoldJob fold_job;                          // the fold of the previous batch, launched once this batch's forward pass is queued
    auto launch_pending_fold = [&]() {
        if (!fold_job.pending) return;
        hipEvent_t e0, e1;
        HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
        fold_ev.push_back(e0); fold_ev.push_back(e1);
        launch_fold(ctx, T, G, fold_job, ctx->astream, e0, e1);
        HIPCHECK(hipEventRecord(ctx->aev_done[fold_job.parity], ctx->astream));
        fold_job.pending = false;
    };
    if (overlap_acc && !ctx->astream) {
        HIPCHECK(hipStreamCreateWithFlags(&ctx->astream, hipStreamNonBlocking));
        for (auto &e : ctx->aev_done) HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    // The per-(step, chain) programs of batch bi + 1 are built by a second host thread while batch bi runs -- its passes on the GPU, its
    // bookkeeping on this thread (the reference's published break-point study: 23 batches of 1017 chains x 41 steps, 23 - 30 ms of
    // build_program of a 90 - 130-ms fit; the builder touches only its own BatchProgram and the caller's read-only inputs, no HIP call).
    // A failure of the early build is raised where the build used to be: at the top of that batch.
    struct BatchProgram { TapTable taps; ChainProgram prog; bool ready = false; std::exception_ptr err; };
    std::unique_ptr<BatchProgram> bprog[2];
    auto build_batch = [&](int64_t bj) {
        std::unique_ptr<BatchProgram> &N = bprog[bj & 1];
        N.reset(new BatchProgram());
        try { build_program(p, g, batch_start[bj], batch_start[bj + 1] - batch_start[bj], op_values, N->taps, N->prog, resume); }
        catch (...) { N->err = std::current_exception(); }
        N->ready = true;
    };
    std::thread builder;
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } builder_guard{builder};      // (also when a batch throws)
    const bool build_ahead = nbatch > 1;
    int64_t fallback_until = -1;                   // batches up to this one repeat on the launch-per-step kernels (poisoned carried slots, below)
    for (int64_t bi = 0; bi < nbatch; ++bi) {
        const int64_t c0 = batch_start[bi], B = batch_start[bi + 1] - c0;
        tr.mark("batch setup");
        if (builder.joinable()) builder.join();
        if (!bprog[bi & 1] || !bprog[bi & 1]->ready) build_batch(bi);
        BatchProgram &BP = *bprog[bi & 1];
        BP.ready = false;                        // (consumed: the slot is rebuilt for batch bi + 2)
        if (BP.err) std::rethrow_exception(BP.err);
        TapTable &taps = BP.taps;
        ChainProgram &prog = BP.prog;
        if (build_ahead && bi + 1 < nbatch) {
            try { builder = std::thread(build_batch, bi + 1); }
            catch (const std::system_error &) {}        // (no thread to be had: the batch is built at its own top, as before)
        }
        tr.mark("build_program");
        const GeometryPlan gp = plan_geometry(ctx, p, g, prog, B, d, resume, carry, &taps);
        const bool fast = gp.fast, fused1d = gp.fused1d, use_mfma = gp.use_mfma;
        const int64_t fusedK = gp.fusedK;
        const int f1_TJ = gp.f1_TJ;
        Tile tile = gp.tile;
        const int fastS = gp.fastS, fast_nseg = gp.fast_nseg, fast_fnblk = gp.fast_fnblk, mS = gp.mS, m_nseg = gp.m_nseg, m_tiles_j = gp.m_tiles_j,
                  m_nblk = gp.m_nblk;
        auto f1_lds = [&](int64_t K) { return (size_t)(4 * (f1_TJ + 2 * K * prog.LW1) + K * f1_TJ + K * (prog.LW1 + 1) + K * rec_len + 4 * 8 + 2 + K + 8 + K * 3 * f1_TJ) * sizeof(double); };
        ctx->timing.fwd_kernel_variant = ctx->timing.bwd_kernel_variant = fast ? 1 : (fused1d ? 4 : 0);
        ctx->timing.cells_per_launch = std::max<int64_t>(ctx->timing.cells_per_launch, B * G);

        // --- device metadata ---
        DeviceMeta M;
        upload_metadata(ctx, p, prog, taps, B, full, fast, tile.nblk, M, gp.wideH, gp.wideV, gp.hSplit, gp.hFusedMax);
        unsigned char *const d_kindF = M.kindF, *const d_kindB = M.kindB, *const d_cmodeF = M.cmodeF, *const d_cmodeB = M.cmodeB;
        double *const d_limitF = M.limitF, *const d_limitB = M.limitB;
        int *const d_tapF0 = M.tapF0, *const d_tapF1 = M.tapF1, *const d_tapB0 = M.tapB0, *const d_tapB1 = M.tapB1;
        int *const d_orderF = M.orderF, *const d_orderB = M.orderB;
        double *const d_taps = M.taps;
        int *const d_off = M.off, *const d_lw = M.lw, *const d_lw2 = M.lw2;
        double *const d_invN = M.invN, *const d_w = M.w, *const d_dump = M.dump;
        const std::vector<int> &orderF = M.h_orderF, &orderB = M.h_orderB;
        const std::vector<std::vector<FastRange>> &rangesF = M.rangesF, &rangesB = M.rangesB;

        tr.mark("buckets + metadata H2D");
        // --- state ---
        size_t psz = (size_t)T * B * NRED * tile.nblk;
        ctx->psumF.ensure(psz * 8);
        ctx->redF.ensure((size_t)T * B * NRED * 8);
        double *d_psF = ctx->psumF.as<double>();
        double *d_post = nullptr, *d_pp[2] = {nullptr, nullptr};
        ctx->state.ensure((size_t)2 * B * G * 8);
        d_pp[0] = ctx->state.as<double>();
        d_pp[1] = d_pp[0] + (size_t)B * G;
        if (!evidence_only) {
            DevBuf &pb = (overlap_acc && (bi & 1)) ? ctx->post2 : ctx->post;
            pb.ensure((size_t)B * T * G * 8);
            d_post = pb.as<double>();
            // the fold of batch bi - 2 read this buffer on the second stream: it has to be through before this batch writes it
            if (overlap_acc && bi >= 2) HIPCHECK(hipStreamWaitEvent(st, ctx->aev_done[bi & 1], 0));
        }

        // BLHIP_RESUME: step 0 reads each chain's carried (normalised) state; its "previous partial sums" add up to 1
        const double *d_carry_src = nullptr, *d_unit = nullptr;
        if (resume) {
            d_carry_src = ctx->carry[p->carry_slot].buf.as<double>();
            std::vector<double> unit((size_t)B * NRED * tile.nblk, 0.0);
            const std::vector<double> &mv = ctx->carry[p->carry_slot].maxv;
            for (int64_t b = 0; b < B; ++b)
                for (int k = 0; k < NRED; ++k)       // slot 6 = maximum of the carried state (NotEqual inverts around it)
                    unit[((size_t)b * NRED + k) * tile.nblk] = (k == 6 && (int64_t)mv.size() == B) ? mv[b] : 1.0;
            ctx->unit.ensure(unit.size() * 8);
            HIPCHECK(hipMemcpyAsync(ctx->unit.p, unit.data(), unit.size() * 8, hipMemcpyHostToDevice, st));
            sync_stream(ctx, st);
            d_unit = ctx->unit.as<double>();
        }
        StepParams P{};
        P.n0 = g.n0; P.n1 = g.n1; P.TI = tile.TI; P.TJ = tile.TJ; P.LW0 = tile.LW0; P.LW1 = tile.LW1;
        P.tiles_i = tile.tiles_i; P.tiles_j = tile.tiles_j; P.nblk = tile.nblk; P.ndim = p->ndim; P.d = d;
        P.rec_len = rec_len; P.shared[SRC_PREV] = nullptr; P.shared[SRC_PRIOR] = d_prior; P.shared[SRC_RESET] = d_reset;
        P.shared[SRC_UNIFORM] = d_uniform; P.shared[SRC_INDEP] = d_indep; P.taps = d_taps; P.tap_off = d_off; P.tap_lw = d_lw; P.tap_lw2 = d_lw2;
        P.m0 = d_m0; P.m1 = d_m1; P.colA = d_colA; P.colB = d_colB; P.chains = (int)B;
        P.prev_nblk = tile.nblk;

        blf::FastParams FP{};
        if (fast) {
            FP.n0 = g.n0; FP.n1 = g.n1; FP.TJ = tile.TJ; FP.S = fastS; FP.nseg = fast_nseg;
            FP.tiles_j = tile.tiles_j; FP.nblk = tile.nblk; FP.fnblk = fast_fnblk;
            FP.dump = d_dump;
            FP.mlean = (g.n0 % mS == 0 && (double)G * 8.0 < 4.0e9 && g.n1 < (1 << 20) && g.n0 < (1 << 24)) ? 1 : 0;
            FP.mS = mS; FP.mnseg = m_nseg; FP.mtiles_j = m_tiles_j; FP.mnblk = m_nblk;
            FP.ndim = p->ndim; FP.d = d; FP.means = forward_only ? 1 : 0;
            FP.shared[SRC_PREV] = nullptr; FP.shared[SRC_PRIOR] = d_prior; FP.shared[SRC_RESET] = d_reset;
            FP.shared[SRC_UNIFORM] = d_uniform; FP.shared[SRC_INDEP] = d_indep; P.shared[SRC_INDEP] = d_indep; FP.taps = d_taps; FP.tap_off = d_off; FP.tap_lw = d_lw;
            FP.m0 = d_m0; FP.m1 = d_m1; FP.colA = d_colA; FP.colB = d_colB; FP.prev_nblk = tile.nblk;
            // likelihood recurrence along rows needs an equally spaced row axis (to rounding)
            const double *m0h = p->marginal[0];
            const double step = (m0h[g.n0 - 1] - m0h[0]) / (double)(g.n0 - 1);
            double dev = 0.0, mx = 0.0;
            for (int i = 0; i < g.n0; ++i) {
                dev = std::max(dev, std::fabs(m0h[i] - (m0h[0] + i * step)));
                mx = std::max(mx, std::fabs(m0h[i]));
            }
            FP.step0 = step;
            FP.use_rec = (p->obs_model == BLHIP_OM_GAUSSIAN && dev <= 8.0 * 2.3e-16 * mx && ctx->option("recurrence", 1.0) != 0.0) ? 1 : 0;
        }
        // ---- the resident paths (single chain: blhip_resident.hpp; batches of chains: blhip_chainres.hpp); the launch-per-step kernels
        //      below are their fall-back ---------------------------------------------------------------------------------------------------
        BatchEnv E{};
        E.ctx = ctx; E.p = p; E.st = st; E.g = g; E.G = G; E.T = T; E.B = B; E.c0 = c0; E.d = d; E.rec_len = rec_len; E.ff = ff;
        E.DT = &DT; E.M = &M; E.prog = &prog; E.taps = &taps; E.step0 = FP.step0; E.d_post = d_post; E.log_w = log_w;
        E.chain_means = res && res->posterior_mean;
        E.overlap_acc = overlap_acc;
        E.tr = &tr;
        E.fold_ev = &fold_ev; E.bi = bi; E.allow_chainres = bi > fallback_until;
        ResidentRun RR;
        RR.setup(E, n_chains, fast, FP.use_rec != 0, psz);
        ChainRun CR;
        if (!RR.on) CR.setup(E, fast, FP.use_rec != 0, psz);
        if (RR.on || CR.on) { ctx->psumF.ensure(psz * 8); d_psF = ctx->psumF.as<double>(); }
        if (CR.on && CR.depad) {                 // a padded batch that hands its posteriors out: scratch sequence for the kernels, d_post stays the result
            ctx->postpad.ensure((size_t)B * T * CR.Gk * 8);
            E.d_post = ctx->postpad.as<double>();
        } else if (CR.on && (CR.cp.pad || CR.ax1) && d_post) {      // the private sequence of a padded chain-resident batch lives on the padded geometry
            DevBuf &pb = (overlap_acc && (bi & 1)) ? ctx->post2 : ctx->post;
            pb.ensure((size_t)B * T * CR.Gk * 8);
            d_post = pb.as<double>();
            E.d_post = d_post;
        }
        const bool resident = RR.on, chainres = CR.on;
        bool resident_failed = false;

        // bucket streams: fork = every bucket stream waits for the main stream; join = the main stream waits for all of them
        // (only worth it when a step really has several launches: a launch on a secondary stream costs ~9 us more
        //  than a back-to-back launch on the main stream -- measured on single-chain fits)
        size_t max_ranges = 0;
        for (const auto &r : rangesF) max_ranges = std::max(max_ranges, r.size());
        for (const auto &r : rangesB) max_ranges = std::max(max_ranges, r.size());
        const bool multistream = fast && max_ranges >= 2;
        auto fork_streams = [&]() {
            if (!multistream) return;
            HIPCHECK(hipEventRecord(ctx->fork_ev, st));
            for (auto &bs : ctx->bstream) HIPCHECK(hipStreamWaitEvent(bs, ctx->fork_ev, 0));
        };
        auto join_streams = [&]() {
            if (!multistream) return;
            for (int k = 0; k < blhip_ctx::NBS; ++k) {
                HIPCHECK(hipEventRecord(ctx->bev[k], ctx->bstream[k]));
                HIPCHECK(hipStreamWaitEvent(st, ctx->bev[k], 0));
            }
        };
        // a chain only depends on its own previous step: streams need a barrier only where bucket membership changes
        auto same_membership = [&](const std::vector<int> &order, const std::vector<std::vector<FastRange>> &ranges,
                                   int64_t ta, int64_t tb) {
            if (ranges[ta].size() != ranges[tb].size()) return false;
            for (size_t k = 0; k < ranges[ta].size(); ++k)
                if (ranges[ta][k].key != ranges[tb][k].key || ranges[ta][k].count != ranges[tb][k].count) return false;
            return std::equal(order.begin() + ta * B, order.begin() + (ta + 1) * B, order.begin() + tb * B);
        };
        std::vector<char> any_hF, any_hB, any_vF, any_vB;
        double *d_hsrc = nullptr, *d_vsrc = nullptr;
        if (gp.wideH || gp.wideV) {
            any_hF.assign(T, 0); any_hB.assign(T, 0); any_vF.assign(T, 0); any_vB.assign(T, 0);
            for (int64_t t = 0; t < T; ++t)
                for (int64_t b = 0; b < B; ++b) {
                    if (gp.wideH && prog.tapF1[t * B + b] >= 0) any_hF[t] = 1;
                    if (gp.wideH && full && prog.tapB1[t * B + b] >= 0) any_hB[t] = 1;
                    if (gp.wideV && prog.tapF0[t * B + b] >= 0) any_vF[t] = 1;
                    if (gp.wideV && full && prog.tapB0[t * B + b] >= 0) any_vB[t] = 1;
                }
            ctx->hsrc.ensure((size_t)B * G * 8 * ((gp.wideH && gp.wideV) ? 2 : 1));
            d_hsrc = ctx->hsrc.as<double>();
            d_vsrc = (gp.wideH && gp.wideV) ? d_hsrc + (size_t)B * G : d_hsrc;
        }
        constexpr int mfma_min_r0 = 8;
        // both-axes launches: the matrix-pipe kernel wins while the launch is latency-bound (few cells per CU); with the chip
        // full the vector kernel's 2 x 17 FMAs per cell beat 2 x 32 band products (measured: 1024^2 12.7 vs 14.9 us,
        // 2048^2 30.0 vs 27.7 us, 4096^2 97 vs 77 us per forward step)
        const bool mfma_h = ctx->option("mfma_h", 1.0) != 0.0;
        const double mfma_h_max_cells = ctx->option("mfma_h_max_cells", 2.5e6);
        long long n_mfma[2] = {0, 0}, n_fast[2] = {0, 0};
        constexpr bool uniform_launch = true;
        auto run_step = [&](int mode, int64_t t, const double *srcp, long long src_stride, double *dstp, long long dst_stride,
                            double *postp, long long post_stride, const double *ps_prev, int prev_slot, double *ps_out,
                            bool means) {
            if (fast) {
                blf::FastParams Q = FP;
                Q.src = srcp; Q.src_stride = src_stride; Q.dst = dstp; Q.dst_stride = dst_stride;
                Q.post = postp; Q.post_stride = post_stride;
                Q.srckind = (mode == MODE_FWD ? d_kindF : d_kindB) + t * B;
                Q.tap0 = (mode == MODE_FWD ? d_tapF0 : d_tapB0) + t * B;
                Q.tap1 = (mode == MODE_FWD ? d_tapF1 : d_tapB1) + t * B;
                Q.psum_prev = ps_prev; Q.prev_slot = prev_slot; Q.psum_out = ps_out;
                Q.rec = d_rec + t * rec_len; Q.lik = d_lik ? d_lik + (size_t)t * G : nullptr;
                const int *ord = (mode == MODE_FWD ? d_orderF : d_orderB) + t * B;
                const bool prepass_step = gp.wideH && (mode == MODE_FWD ? any_hF : any_hB)[t];
                const bool prepass_v = gp.wideV && (mode == MODE_FWD ? any_vF : any_vB)[t];
                for (const FastRange &r : (mode == MODE_FWD ? rangesF[t] : rangesB[t])) {
                    Q.chain_ids = ord + r.start;
                    Q.u_valid = 0;
                    Q.hsrc = nullptr;
                    const bool prepass = prepass_step && r.pre;
                    if (r.count == 1 && uniform_launch) {      // single-chain launch: hand the chain's metadata over by value
                        const int64_t tb = t * B;
                        const int cb = (mode == MODE_FWD ? orderF : orderB)[tb + r.start];
                        const int k0 = (mode == MODE_FWD ? prog.tapF0 : prog.tapB0)[tb + cb];
                        const int k1 = (mode == MODE_FWD ? prog.tapF1 : prog.tapB1)[tb + cb];
                        Q.u_valid = 1; Q.u_chain = cb; Q.u_kind = (mode == MODE_FWD ? prog.kindF : prog.kindB)[tb + cb];
                        Q.u_t0 = k0; Q.u_lw0 = k0 >= 0 ? taps.lw[k0] : 0; Q.u_off0 = k0 >= 0 ? taps.off[k0] : 0;
                        Q.u_t1 = k1; Q.u_lw1 = k1 >= 0 ? taps.lw[k1] : 0; Q.u_off1 = k1 >= 0 ? taps.off[k1] : 0;
                    }
                    hipStream_t ls = multistream ? ctx->bstream[r.key] : st;
                    if (prepass) {
                        // row filter of the bucket's chains -> hsrc, on the bucket's stream: the launch below consumes it instead of its
                        // sources; the buckets' streams overlap one bucket's (HBM-bound) pre-pass with another's (matrix-pipe-bound) step
                        blh::HParams HP{};
                        HP.n0 = g.n0; HP.n1 = g.n1; HP.tiles_j = (g.n1 + blh::CB - 1) / blh::CB; HP.lwmax = prog.LW1; HP.pitch = blh::pitch_for(prog.LW1);
                        HP.src = srcp; HP.src_stride = src_stride;
                        for (int k = 0; k < 5; ++k) HP.shared[k] = FP.shared[k];
                        HP.chain_ids = Q.chain_ids;
                        HP.srckind = Q.srckind; HP.tap1 = Q.tap1; HP.taps = d_taps; HP.tap_off = d_off; HP.tap_lw = d_lw; HP.dst = d_hsrc;
                        launch_hwide(ls, HP, r.count);
                        Q.hsrc = d_hsrc;
                        account(ctx, mode != MODE_FWD, (double)r.count * G * 16.0, (double)r.count * G * 2.0 * (2.0 * prog.LW1 + 8.0));
                    }
                    if (prepass_v) {
                        // column filter of the bucket's chains (after the row filter, if there is one) -> the launch below runs without a stencil
                        blh::HParams VP{};
                        VP.n0 = g.n0; VP.n1 = g.n1; VP.tiles_j = (g.n1 + blh::CV - 1) / blh::CV; VP.lwmax = prog.LW0;
                        VP.src = srcp; VP.src_stride = src_stride; VP.presrc = Q.hsrc;
                        for (int k = 0; k < 5; ++k) VP.shared[k] = FP.shared[k];
                        VP.chain_ids = Q.chain_ids;
                        VP.srckind = Q.srckind; VP.tap1 = Q.tap0; VP.taps = d_taps; VP.tap_off = d_off; VP.tap_lw = d_lw; VP.dst = d_vsrc;
                        launch_vwide(ls, VP, r.count);
                        Q.hsrc = d_vsrc;
                        account(ctx, mode != MODE_FWD, (double)r.count * G * 16.0, (double)r.count * G * 2.0 * (2.0 * prog.LW0 + 8.0));
                    }
                    const bool on_pipe = use_mfma && (r.H ? (mfma_h && (double)r.count * g.n0 * g.n1 <= mfma_h_max_cells) : r.R0 >= mfma_min_r0);
                    if (on_pipe) { launch_mfma(ls, p->obs_model, mode, Q, r.R0, r.H, r.count); ++n_mfma[mode == MODE_FWD ? 0 : 1]; }
                    else { launch_fast(ls, p->obs_model, mode, Q, r.R0, r.H, r.count); ++n_fast[mode == MODE_FWD ? 0 : 1]; }
                    // streaming kernels: state in, state out (+ stored alpha in, posterior out backward; + a tabulated likelihood)
                    const bool bw = mode != MODE_FWD;
                    const double fl = (on_pipe ? (r.R0 > 0 ? band_stencil_flop(r.R0) : 0.0) + (r.H ? band_stencil_flop(8) : 0.0)
                                               : valu_stencil_flop(r.R0) + (r.H ? valu_stencil_flop(8) : 0.0)) + (bw ? EPI_BWD_FLOP : EPI_FWD_FLOP);
                    account(ctx, bw, (double)r.count * G * ((bw ? 32.0 : 16.0) + (d_lik ? 8.0 : 0.0)), (double)r.count * G * fl);
                }
            } else {
                StepParams Q = P;
                Q.src = srcp; Q.src_stride = src_stride; Q.dst = dstp; Q.dst_stride = dst_stride;
                Q.post = postp; Q.post_stride = post_stride;
                Q.srckind = (mode == MODE_FWD ? d_kindF : d_kindB) + t * B;
                Q.tap0 = (mode == MODE_FWD ? d_tapF0 : d_tapB0) + t * B;
                Q.tap1 = (mode == MODE_FWD ? d_tapF1 : d_tapB1) + t * B;
                Q.cmode = prog.has_clamp ? (mode == MODE_FWD ? d_cmodeF : d_cmodeB) + t * B : nullptr;
                Q.limit = prog.has_clamp ? (mode == MODE_FWD ? d_limitF : d_limitB) + t * B : nullptr;
                Q.psum_prev = ps_prev; Q.prev_slot = prev_slot; Q.psum_out = ps_out;
                Q.rec = d_rec + t * rec_len; Q.lik = d_lik ? d_lik + (size_t)t * G : nullptr;
                launch_step(st, p->obs_model, Q, tile, (int)B, mode, means);
                const bool bw = mode != MODE_FWD;
                account(ctx, bw, (double)B * G * ((bw ? 32.0 : 16.0) + (d_lik ? 8.0 : 0.0)),
                        (double)B * G * (valu_stencil_flop(prog.LW0) + valu_stencil_flop(prog.LW1) + (bw ? EPI_BWD_FLOP : EPI_FWD_FLOP)));
            }
        };

        float ms = 0;
        BatchOutcome O;
        std::vector<double> &invN = O.invN;
        auto passes = [&](const int64_t K) -> bool {
        bl1f::F1Params F1{};
        if (fused1d) {
            F1.n = g.n1; F1.TJ = f1_TJ; F1.nblk = tile.nblk; F1.LW = prog.LW1; F1.T = (int)T; F1.B = (int)B; F1.d = d; F1.rec_len = rec_len;
            F1.shared[SRC_PREV] = nullptr; F1.shared[SRC_PRIOR] = d_prior; F1.shared[SRC_RESET] = d_reset;
            F1.shared[SRC_UNIFORM] = d_uniform; F1.shared[SRC_INDEP] = d_indep;
            F1.taps = d_taps; F1.tap_off = d_off; F1.tap_lw = d_lw; F1.m1 = d_m1; F1.colA = d_colA; F1.rec = d_rec; F1.lik = d_lik;
            F1.cmode = nullptr; F1.tap_lw2 = d_lw2;
        }
        tr.mark("path setup");
        // --- forward pass (core.py:372-411) ---
        // the previous batch's fold goes to the second stream FIRST: queued behind this pass's launches it was not started before
        // they had all finished (measured: rocprofv3 time line), queued ahead of them it shares the chip with them
        launch_pending_fold();
        HIPCHECK(hipEventRecord(ev[0], st));
        // 1-D grids whose blocks all fit on the chip at once: ONE persistent launch per pass (blhip_persist1d.hpp), else a launch per K steps
        // (a pass of a single superstep -- OnlineStudy.step, T <= K -- has no launch boundary to save)
        const bool c1d_now = gp.chain1d;
        const bool p1d_now = fused1d && !c1d_now && !resident_failed && ctx->resident_ok && ctx->option("persist1d", 1.0) != 0.0 && T > K &&
                             (long long)tile.nblk * B <= std::min(ctx->num_cus, 256);
        // (the chains of a batch see the same likelihood: tabulated once per batch by the in-kernel function itself, shared by both passes)
        const bool c1d_table = c1d_now && B >= 4 && !d_lik && (p->obs_model == BLHIP_OM_POISSON || p->obs_model == BLHIP_OM_GAUSSIAN_MEAN);
        double *d_lik1 = nullptr;
        if (c1d_table) {
            ctx->lik1d.ensure((size_t)T * G * 8);
            d_lik1 = ctx->lik1d.as<double>();
            bl1f::F1Params Q = F1;
            build_lik1d_table(st, p->obs_model, Q, d_lik1);
        }
        auto launch_c1d = [&](bool bwd, double *psum) {
            bl1f::F1Params Q = F1;
            if (d_lik1) Q.lik = d_lik1;
            Q.K = 1; Q.dir = bwd ? -1 : 1; Q.t_first = bwd ? (int)(T - 1) : 0; Q.psum = psum; Q.prev_slot = bwd ? 2 : 0;
            Q.srckind = bwd ? d_kindB : d_kindF; Q.tap = bwd ? d_tapB1 : d_tapF1;
            if (gp.shift1d) Q.cmode = bwd ? d_cmodeB : d_cmodeF;
            if (gp.clamp1d) { Q.limit = bwd ? d_limitB : d_limitF; Q.no_shift = prog.has_shift ? 0 : 1; }
            Q.store = (bwd || !evidence_only) ? 1 : 0; Q.means = bwd ? 1 : (forward_only ? 1 : 0);
            Q.post = (bwd || !evidence_only) ? d_post : nullptr; Q.post_stride = (long long)T * G;
            Q.src = nullptr; Q.src_stride = 0; Q.dst = nullptr; Q.dst_stride = 0;
            launch_chain1d(st, d_lik1 ? BLHIP_OM_TABLE : p->obs_model, Q, bwd, 2);
        };
        bl1p::P1Params P1{};
        unsigned *d_abort1 = nullptr;
        size_t p1d_bytes = 0;
        if (p1d_now) {
            const size_t xb = carve_size((size_t)2 * B * g.n1 * 16), gb = carve_size((size_t)2 * B * tile.nblk * 16);
            p1d_bytes = xb + gb + carve_size(64);
            ctx->p1d.ensure(p1d_bytes);
            char *pc = ctx->p1d.as<char>();
            P1.xch = carve<unsigned long long>(pc, (size_t)2 * B * g.n1 * 2);
            P1.gran = carve<unsigned long long>(pc, (size_t)2 * B * tile.nblk * 2);
            d_abort1 = carve<unsigned>(pc, 16);
            P1.abort_word = d_abort1;
            P1.timeout_ticks = (unsigned long long)(resident_timeout_s(ctx, T) * 1e8);      // wall_clock64: 100 MHz
            P1.n = F1.n; P1.TJ = F1.TJ; P1.nblk = F1.nblk; P1.LW = F1.LW; P1.K = (int)K; P1.T = F1.T; P1.B = F1.B; P1.d = F1.d; P1.rec_len = F1.rec_len;
            for (int k = 0; k < 5; ++k) P1.shared[k] = F1.shared[k];
            P1.taps = F1.taps; P1.tap_off = F1.tap_off; P1.tap_lw = F1.tap_lw; P1.m1 = F1.m1; P1.colA = F1.colA; P1.rec = F1.rec; P1.lik = F1.lik;
        }
        auto launch_p1d = [&](bool bwd, double *psum) {
            HIPCHECK(hipMemsetAsync(ctx->p1d.p, 0, p1d_bytes, st));       // tags 0, abort word 0
            bl1p::P1Params Q = P1;
            Q.dir = bwd ? -1 : 1; Q.psum = psum; Q.prev_slot = bwd ? 2 : 0;
            Q.srckind = bwd ? d_kindB : d_kindF; Q.tap = bwd ? d_tapB1 : d_tapF1;
            {   // the pass's weights spelled out per (step, chain): the kernel stages a superstep's weights with ONE load per element
                const long long TB = (long long)T * B;
                const size_t wb = carve_size((size_t)TB * (P1.LW + 1) * 8), lb = carve_size((size_t)TB * 4);
                if (wb + lb <= ((size_t)256 << 20)) {
                    ctx->p1w.ensure(wb + lb);
                    char *wc = ctx->p1w.as<char>();
                    double *wtab = carve<double>(wc, (size_t)TB * (P1.LW + 1));
                    int *lwtab = carve<int>(wc, (size_t)TB);
                    const long long ne = TB * (P1.LW + 1);
                    BL_LAUNCH(bl1p::build_wtab_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, Q.tap, d_lw, d_off, d_taps, TB, P1.LW, wtab, lwtab);
                    HIPCHECK(hipGetLastError());
                    Q.wtab = wtab; Q.lwtab = lwtab;
                }
            }
            Q.store = (bwd || !evidence_only) ? 1 : 0; Q.means = bwd ? 1 : (forward_only ? 1 : 0);
            Q.post = (bwd || !evidence_only) ? d_post : nullptr; Q.post_stride = (long long)T * G;
            Q.src0 = (!bwd && resume) ? d_carry_src : nullptr; Q.src0_stride = G;
            Q.dst = (bwd || evidence_only) ? d_pp[((T - 1) / K) & 1] : nullptr; Q.dst_stride = G;
            launch_persist1d(st, p->obs_model, Q, bwd, f1_lds(K));
        };
        if (c1d_now) {
            launch_c1d(false, d_psF);
        } else if (p1d_now) {
            launch_p1d(false, d_psF);
        } else if (fused1d) {
            for (int64_t t = 0; t < T; t += K) {
                bl1f::F1Params Q = F1;
                Q.K = (int)std::min<int64_t>(K, T - t); Q.dir = 1; Q.t_first = (int)t;
                Q.srckind = d_kindF; Q.tap = d_tapF1; Q.psum = d_psF; Q.prev_slot = 0;
                Q.psum_prev = t > 0 ? d_psF + (size_t)(t - 1) * B * NRED * tile.nblk : d_unit;
                Q.store = evidence_only ? 0 : 1; Q.means = forward_only ? 1 : 0;
                const int64_t tl = t + Q.K - 1;              // last step of this launch
                if (evidence_only) {
                    Q.post = nullptr; Q.src = d_pp[(t / K + 1) & 1]; Q.src_stride = G; Q.dst = d_pp[(t / K) & 1]; Q.dst_stride = G;
                } else {
                    Q.post = d_post; Q.post_stride = (long long)T * G; Q.dst = nullptr; Q.dst_stride = 0;
                    Q.src = d_post + (t > 0 ? (size_t)(t - 1) * G : 0); Q.src_stride = (long long)T * G;
                }
                (void)tl;
                if (t == 0 && resume) { Q.src = d_carry_src; Q.src_stride = G; }
                launch_fused1d(st, p->obs_model, Q, false, f1_lds(Q.K));
            }
        }
        if (fused1d)        // 1-D paths: a row in, a row out per step (the K-steps-per-launch kernel re-reads only halos)
            account(ctx, false, (double)B * G * T * (16.0 + (d_lik ? 8.0 : 0.0)), (double)B * G * T * (valu_stencil_flop(prog.LW1) + EPI_FWD_FLOP));
        const bool res_now = resident && !resident_failed;
        const bool cres_now = chainres && !resident_failed;
        const int nblk_now = res_now ? RR.nblk : (cres_now ? CR.cp.strips : tile.nblk);      // partial-sum slots per (step, sum) of this pass
        if (res_now) RR.launch(E, false, d_psF);
        if (cres_now) CR.pass(E, false, d_psF);
        fork_streams();
        for (int64_t t = 0; t < T && !fused1d && !res_now && !cres_now; ++t) {
            if (multistream && t > 0 && !same_membership(orderF, rangesF, t - 1, t)) { join_streams(); fork_streams(); }
            const double *srcp; double *dstp; long long sstr, dstr;
            if (evidence_only) {
                srcp = d_pp[(t + 1) & 1]; sstr = G; dstp = d_pp[t & 1]; dstr = G;
            } else {
                srcp = d_post + (t > 0 ? (size_t)(t - 1) * G : 0); sstr = (long long)T * G;
                dstp = d_post + (size_t)t * G; dstr = (long long)T * G;
            }
            if (t == 0 && resume) { srcp = d_carry_src; sstr = G; }
            run_step(MODE_FWD, t, srcp, sstr, dstp, dstr, nullptr, 0,
                     t > 0 ? d_psF + (size_t)(t - 1) * B * NRED * tile.nblk : (resume ? d_unit : d_psF), 0,
                     d_psF + (size_t)t * B * NRED * tile.nblk, forward_only);
        }
        join_streams();
        HIPCHECK(hipEventRecord(ev[1], st));
        // (one slot per sum -- the kernels that give a chain ONE block: nothing to add up, the partials ARE the sums.  The published
        //  break-point study: 2 x 6 launches of T * B * NRED = 1.3 M one-value blocks, 0.48 ms each, 5.7 of a 44-ms fit)
        if (nblk_now > 1)
            BL_LAUNCH(reduce_partials_kernel, dim3((unsigned)(T * B * NRED)), dim3(NTHREADS), 0, st, d_psF,
                               ctx->redF.as<double>(), nblk_now, NRED);
        ctx->pinF.ensure((size_t)T * B * NRED * 8);
        redF = ctx->pinF.as<double>();
        // (the launch-per-step backward pass of a batch of chains needs nothing of the forward pass's host bookkeeping: it is done
        //  while the GPU runs that pass -- 11 ms of the published break-point study's fit, 23 batches of 1017 chains)
        const bool late_fb = full && (!fused1d || c1d_now) && !res_now && !cres_now && !p1d_now && B >= 64;
        // ... and nobody waits for its sums either: they travel to the host on the copy stream BESIDE the backward pass (10 MB per batch
        // of the break-point study, 2.4 of its 42 ms over PCIe with the chip idle)
        const bool late_copy = late_fb && ctx->cstream && ctx->option("late_sums", 1.0) != 0.0;
        if (late_copy) {
            HIPCHECK(hipEventRecord(ctx->cev[0], st));
            HIPCHECK(hipStreamWaitEvent(ctx->cstream, ctx->cev[0], 0));
            HIPCHECK(hipMemcpyAsync(redF, nblk_now > 1 ? ctx->redF.p : (const void *)d_psF, (size_t)T * B * NRED * 8, hipMemcpyDeviceToHost, ctx->cstream));
            HIPCHECK(hipEventRecord(ctx->cev[1], ctx->cstream));
            tr.mark("forward pass queued (its sums follow beside the backward pass)");
        } else {
            HIPCHECK(hipMemcpyAsync(redF, nblk_now > 1 ? ctx->redF.p : (const void *)d_psF, (size_t)T * B * NRED * 8, hipMemcpyDeviceToHost, st));
            tr.mark("forward pass queued");
            sync_stream(ctx, st);
            tr.mark("forward pass done + sums D2H");
            ms = 0;
            HIPCHECK(hipEventElapsedTime(&ms, ev[0], ev[1]));
            ctx->timing.forward_ms += ms;
        }
        ctx->timing.forward_launches += T;
        if (n_mfma[0] > 0 && n_mfma[0] >= n_fast[0]) ctx->timing.fwd_kernel_variant = 3;
        if (res_now) {
            ctx->timing.fwd_kernel_variant = 5;
            if (!RR.forward_ok(E, redF)) { resident_failed = true; return false; }
        }
        if (cres_now) {
            ctx->timing.fwd_kernel_variant = 6;
            if (!CR.forward_ok(E, redF)) { resident_failed = true; return false; }
            tr.mark("  forward_ok");
        }
        if (c1d_now) ctx->timing.fwd_kernel_variant = 9;
        if (p1d_now) {
            ctx->timing.fwd_kernel_variant = 8;
            if (resident_gave_up(ctx, st, d_abort1)) { resident_failed = true; return false; }
        }

        bool raw_ok = late_fb ? true : forward_bookkeeping(p, prog, redF, B, dV, fused1d, K, evidence_only, forward_only, O);
        tr.mark("forward checks + bookkeeping");

        // --- backward pass (core.py:424-470) ---
        invN.assign((size_t)B * T, 0.0);
        if (full) {
            ctx->psumB.ensure(psz * 8);
            ctx->redB.ensure((size_t)T * B * NRED * 8);
            double *d_psB = ctx->psumB.as<double>();
            if (fused1d && !raw_ok && K > 1) return false;
            HIPCHECK(hipEventRecord(ev[2], st));
            if (c1d_now) {
                launch_c1d(true, d_psB);
            } else if (p1d_now) {
                launch_p1d(true, d_psB);
            } else if (fused1d) {
                for (int64_t t = T - 1; t >= 0; t -= K) {
                    bl1f::F1Params Q = F1;
                    Q.K = (int)std::min<int64_t>(K, t + 1); Q.dir = -1; Q.t_first = (int)t;
                    Q.srckind = d_kindB; Q.tap = d_tapB1; Q.psum = d_psB; Q.prev_slot = 2;
                    Q.psum_prev = t < T - 1 ? d_psB + (size_t)(t + 1) * B * NRED * tile.nblk : nullptr;
                    Q.store = 1; Q.means = 1; Q.post = d_post; Q.post_stride = (long long)T * G;
                    const int64_t li = (T - 1 - t) / K;
                    Q.src = d_pp[(li + 1) & 1]; Q.src_stride = G; Q.dst = d_pp[li & 1]; Q.dst_stride = G;
                    launch_fused1d(st, p->obs_model, Q, true, f1_lds(Q.K));
                }
            }
            if (fused1d)
                account(ctx, true, (double)B * G * T * (32.0 + (d_lik ? 8.0 : 0.0)), (double)B * G * T * (valu_stencil_flop(prog.LW1) + EPI_BWD_FLOP));
            if (res_now) RR.launch(E, true, d_psB);
            if (cres_now && CR.fused) { CR.prepare_fold(E, O); tr.mark("  prepare_fold"); }
            if (cres_now) CR.pass(E, true, d_psB);
            fork_streams();
            for (int64_t t = T - 1; t >= 0 && !fused1d && !res_now && !cres_now; --t) {
                if (multistream && t < T - 1 && !same_membership(orderB, rangesB, t + 1, t)) { join_streams(); fork_streams(); }
                // reads c_{t+1} and the stored alpha_t, writes c_t and posterior_t
                run_step(MODE_BWD, t, d_pp[(t + 1) & 1], G, d_pp[t & 1], G, d_post + (size_t)t * G, (long long)T * G,
                         t < T - 1 ? d_psB + (size_t)(t + 1) * B * NRED * tile.nblk : d_psB, 2,
                         d_psB + (size_t)t * B * NRED * tile.nblk, true);
            }
            join_streams();
            HIPCHECK(hipEventRecord(ev[3], st));
            if (nblk_now > 1)
                BL_LAUNCH(reduce_partials_kernel, dim3((unsigned)(T * B * NRED)), dim3(NTHREADS), 0, st, d_psB,
                                   ctx->redB.as<double>(), nblk_now, NRED);
            ctx->pinB.ensure((size_t)T * B * NRED * 8);
            redB = ctx->pinB.as<double>();
            HIPCHECK(hipMemcpyAsync(redB, nblk_now > 1 ? ctx->redB.p : (const void *)d_psB, (size_t)T * B * NRED * 8, hipMemcpyDeviceToHost, st));
            tr.mark("backward pass queued");
            if (late_fb) {
                if (late_copy) HIPCHECK(hipEventSynchronize(ctx->cev[1]));
                raw_ok = forward_bookkeeping(p, prog, redF, B, dV, fused1d, K, evidence_only, forward_only, O);
                tr.mark("forward bookkeeping (behind the backward pass)");
            }
            sync_stream(ctx, st);
            tr.mark("backward pass done + sums D2H");
            if (late_copy) {
                HIPCHECK(hipEventElapsedTime(&ms, ev[0], ev[1]));
                ctx->timing.forward_ms += ms;
            }
            HIPCHECK(hipEventElapsedTime(&ms, ev[2], ev[3]));
            ctx->timing.backward_ms += ms;
            if (n_mfma[1] > 0 && n_mfma[1] >= n_fast[1]) ctx->timing.bwd_kernel_variant = 3;
            ctx->timing.backward_launches += T;
            if (res_now) {
                ctx->timing.bwd_kernel_variant = 5;
                if (!RR.backward_ok(E, redB)) { resident_failed = true; return false; }
            }
            if (c1d_now) ctx->timing.bwd_kernel_variant = 9;
            if (p1d_now) {
                ctx->timing.bwd_kernel_variant = 8;
                if (resident_gave_up(ctx, st, d_abort1)) { resident_failed = true; return false; }
            }
            if (cres_now) {
                ctx->timing.bwd_kernel_variant = 6;
                if (!CR.backward_ok(E, redB)) { resident_failed = true; return false; }
                tr.mark("  backward_ok");
                if (CR.fused && !CR.fold(E, redB, fold_ev)) { resident_failed = true; return false; }
                tr.mark("  prediction check + fold queued");
            }
            // (a batch the backward kernel folded: nobody reads its row normalisers; per-chain means only where the caller asked for them --
            //  hyper- / change-point studies take theirs from the average posterior)
            raw_ok = backward_bookkeeping(p, prog, redF, redB, B, dV, fused1d, res_now ? 0 : -1, O, !(cres_now && CR.fold_done), E.chain_means) && raw_ok;
            tr.mark("backward checks + fused fold + bookkeeping");
        } else if (forward_only) {
            for (int64_t b = 0; b < B; ++b)
                for (int64_t t = 0; t < T; ++t) {
                    const double n0 = res_now ? RR.rowsumF[t] : (cres_now ? CR.rowsumC[b][t] : redF[((size_t)t * B + b) * NRED]);
                    invN[(size_t)b * T + t] = (n0 != 0.0 && std::isfinite(n0)) ? 1.0 / n0 : 0.0;     // (a signed kernel can leave a negative raw sum)
                    if (res_now && t <= T - 1 - RR.RQ.lag) invN[(size_t)b * T + t] = 1.0;                // (already normalised by the resident kernel)
                }
        }

        return raw_ok || K == 1;
        };
        int64_t usedK = fusedK;
        if (!passes(fusedK)) {
            if (resident_failed) {                   // the launch-per-step kernels take over (timing of the failed attempt is dropped)
                ctx->timing.resident_fallbacks += 1;
                ctx->timing.resident_fallback_reason = ctx->resident_last_reason ? ctx->resident_last_reason : BLHIP_FALLBACK_RANGE;
                if (CR.on && CR.touched_parts && !CR.part_fresh0) {
                    // the failed backward pass has added to partial accumulators that carry earlier batches of this call: none of them is in
                    // the average posterior yet -- those batches and this one are repeated on the launch-per-step kernels
                    const int64_t from = ctx->part.live ? ctx->part.first_batch : bi;
                    ctx->part = blhip_ctx::PartState{};
                    ctx->timing.forward_ms = ctx->timing.backward_ms = 0.0;
                    ctx->timing.fwd_hbm_bytes = ctx->timing.bwd_hbm_bytes = ctx->timing.fwd_flops = ctx->timing.bwd_flops = 0.0;
                    ctx->timing.forward_launches = ctx->timing.backward_launches = 0;
                    fallback_until = bi;
                    if (builder.joinable()) builder.join();
                    bprog[0].reset(); bprog[1].reset();
                    sync_stream(ctx, st);
                    bi = from - 1;
                    continue;
                }
                ctx->timing.forward_ms = ctx->timing.backward_ms = 0.0;
                ctx->timing.fwd_hbm_bytes = ctx->timing.bwd_hbm_bytes = ctx->timing.fwd_flops = ctx->timing.bwd_flops = 0.0;
                ctx->timing.forward_launches = ctx->timing.backward_launches = 0;
                ctx->timing.fwd_kernel_variant = ctx->timing.bwd_kernel_variant = fast ? 1 : (fused1d ? 4 : 0);
                n_mfma[0] = n_mfma[1] = n_fast[0] = n_fast[1] = 0;
                if (!passes(fusedK)) fail("internal: the launch-per-step pass failed after the resident pass gave up");
            } else { usedK = 1; passes(1); }
        } else if (resident || chainres || ctx->timing.fwd_kernel_variant == 8) {
            ctx->resident_retry_after = 8;           // a resident pass went through: the next give-up starts from the short wait again
            ctx->resident_giveups = 0;               // (the wait doubles with give-ups IN A ROW only)
        }

        if (CR.on && CR.depad && !resident_failed && !evidence_only) {
            BL_LAUNCH(depad_kernel, dim3((unsigned)((G + NTHREADS - 1) / NTHREADS), (unsigned)(B * T)), dim3(NTHREADS), 0, st, d_post,
                               ctx->postpad.as<double>(), g.n0, g.n1, CR.cp.n0p, CR.Gk, (int)T, CR.ax1 ? 1 : 0);
            HIPCHECK(hipGetLastError());
        }

        // --- carried states / average posterior / kept posterior / results ---
        if (carry) {
            const double *fin; long long fstr;
            if (!evidence_only) { fin = d_post + (size_t)(T - 1) * G; fstr = (long long)T * G; }
            else { fin = fused1d ? d_pp[((T - 1) / usedK) & 1] : d_pp[(T - 1) & 1]; fstr = G; }
            store_carry(ctx, p, B, G, redF, fin, fstr, d_w, prog.has_clamp);
        }
        if (accumulate && CR.fold_done) {
            // (the backward kernel folded this batch)
        } else if (accumulate && overlap_acc) {
            // private copies of the weights, host (page-locked: the copies must not block the host) and device: the batch metadata
            // buffers are rewritten by the next batch while the fold is pending / running
            const size_t wsz = carve_size((size_t)Bmax * 8) + carve_size((size_t)T * Bmax * 8);
            ctx->accw.ensure(2 * wsz);
            ctx->pinA.ensure(2 * wsz);
            char *wc = ctx->accw.as<char>() + (bi & 1) * wsz, *hc = ctx->pinA.as<char>() + (bi & 1) * wsz;
            fold_job.d_w = carve<double>(wc, (size_t)Bmax); fold_job.d_invN = carve<double>(wc, (size_t)T * Bmax);
            fold_job.h_w = carve<double>(hc, (size_t)Bmax); fold_job.h_invN = carve<double>(hc, (size_t)T * Bmax);
            fold_job.d_post = d_post; fold_job.parity = (int)(bi & 1);
            fold_job.sm_n0 = (CR.post_private && !resident_failed && !CR.cp.pad && !CR.ax1) ? g.n0 : 0;
            if (CR.on && (CR.cp.pad || CR.ax1) && !CR.depad && !resident_failed) { fold_job.pad_n0p = CR.cp.n0p; fold_job.pad_n0 = g.n0; fold_job.pad_n1 = g.n1; fold_job.pad_step = CR.Gk; fold_job.pad_ax = CR.ax1 ? 1 : 0; }
            fold_job.pending = prepare_fold(ctx, T, B, O, log_w + c0, fold_job);
            // launched behind the NEXT batch's forward pass (see passes); the last batch has nothing to hide behind
            if (bi == nbatch - 1) launch_pending_fold();
        } else if (accumulate) {
            FoldJob lay;
            const bool padded = CR.on && (CR.cp.pad || CR.ax1) && !CR.depad && !resident_failed;
            if (padded) { lay.pad_n0p = CR.cp.n0p; lay.pad_n0 = g.n0; lay.pad_n1 = g.n1; lay.pad_step = CR.Gk; lay.pad_ax = CR.ax1 ? 1 : 0; }
            // (many small batches: nobody waits for a batch's fold; the last one is waited for with the stream below)
            fold_accumulate(ctx, T, G, B, O, log_w + c0, d_post, d_w, d_invN, (CR.post_private && !resident_failed && !padded) ? g.n0 : 0, padded ? &lay : nullptr,
                            (nbatch >= 4 && !keep && !carry) ? &fold_ev : nullptr);
        }
        if (keep) {
            int64_t row0 = 0, row1 = T;              // rows the resident kernel normalised in place (invN = 1 there) need no pass
            if (resident && !resident_failed) {
                if (full) row1 = 0;                  // (every posterior row was stored normalised)
                else row0 = std::max<int64_t>(0, T - RR.RQ.lag);
            }
            keep_posterior(ctx, g, T, B, O, row0, row1);
        }
        tr.mark("fold / keep / carry");
        write_results(res, p, c0, B, O, !evidence_only);
    }
    flush_partials(ctx, st, &fold_ev);         // the carried partial accumulators of the chain-resident fold -> the average posterior
    if (overlap_acc || !fold_ev.empty()) {     // the last fold(s) before anybody reads the accumulator; their time from their events
        sync_stream(ctx, overlap_acc ? ctx->astream : st);
        for (size_t k = 0; k + 1 < fold_ev.size(); k += 2) {
            float fms = 0;
            HIPCHECK(hipEventElapsedTime(&fms, fold_ev[k], fold_ev[k + 1]));
            ctx->timing.accumulate_ms += fms;
        }
        for (hipEvent_t e : fold_ev) (void)hipEventDestroy(e);
    }
    tr.mark("batches done");
    HIPCHECK(hipEventRecord(ev[7], st));
    HIPCHECK(hipEventSynchronize(ev[7]));
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, ev[6], ev[7]));
    tr.mark("final sync");
    ctx->timing.total_ms = ms;
}

}  // namespace

// Page-locked result arrays.  hipHostMalloc of 16 GiB takes 2.2 s (one thread allocates, zeroes and pins four million pages) -- more
// than the copy it is meant to speed up.  Here the block is an anonymous mapping pinned in PIN_CHUNK pieces (hipHostRegister faults
// the pages in and locks them: 1.06 s for 16 GiB; more threads are slower); the read-backs below copy piece by piece, so every
// DMA lands inside one registered range.  The Python side pins in the background (bayesloop_amd/engine.py: _PinnedPool).
namespace {
constexpr size_t PIN_CHUNK = (size_t)256 << 20;
struct PinnedBlock { size_t bytes; std::vector<char> ok; };
std::mutex g_pin_mu;
std::map<void *, PinnedBlock> g_pinned;

// pieces [a, b) of a host destination such that no piece crosses a chunk boundary of a block of blhip_host_alloc (any other
// destination: one piece)
template <class F> void for_pinned_pieces(void *host, size_t bytes, F &&f) {
    char *base = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        auto it = g_pinned.upper_bound(host);
        if (it != g_pinned.begin()) {
            --it;
            if ((char *)host >= (char *)it->first && (char *)host + bytes <= (char *)it->first + it->second.bytes) base = (char *)it->first;
        }
    }
    if (!base) { f((char *)host, bytes); return; }
    char *p = (char *)host, *end = p + bytes;
    while (p < end) {
        const size_t into = (size_t)(p - base) % PIN_CHUNK;
        const size_t n = std::min<size_t>(PIN_CHUNK - into, (size_t)(end - p));
        f(p, n);
        p += n;
    }
}
}  // namespace

extern "C" {

int blhip_abi_version(void) { return BLHIP_ABI_VERSION; }

int64_t blhip_kernel_census(char *buf, int64_t cap) {
    try {
        std::vector<std::pair<std::string, unsigned long long>> rows;
        for (blreg::Entry *e = blreg::head().load(std::memory_order_acquire); e; e = e->next) {
            // typeid name of blreg::Site<&kernel>: demangled "blreg::Site<&blc::chain_kernel<4, 1, false, false, false, false>(blc::ChainParams)>"
            int status = 0;
            char *dm = abi::__cxa_demangle(e->mangled, nullptr, nullptr, &status);
            std::string name = (status == 0 && dm) ? dm : (e->mangled ? e->mangled : "?");
            std::free(dm);
            // -> "blreg::Site<&(void blc::chain_kernel<10, 1, false, false, false, false>(blc::ChainParams))>": keep the kernel with its template arguments
            const std::string pre = "blreg::Site<&(", post = ")>";
            if (name.size() > pre.size() + post.size() && name.compare(0, pre.size(), pre) == 0 && name.compare(name.size() - post.size(), post.size(), post) == 0) {
                name = name.substr(pre.size(), name.size() - pre.size() - post.size());
                if (name.compare(0, 5, "void ") == 0) name.erase(0, 5);
                if (!name.empty() && name.back() == ')') {            // the parameter list: the last top-level parenthesis group
                    int depth = 0;
                    for (size_t k = name.size(); k-- > 0;) {
                        if (name[k] == ')') ++depth;
                        else if (name[k] == '(' && --depth == 0) { name.erase(k); break; }
                    }
                }
            }
            else if (name.compare(0, 13, "blreg::Site<&") == 0 && name.back() == '>') name = name.substr(13, name.size() - 14);      // (a kernel that is no template)
            rows.emplace_back(name, e->launches.load(std::memory_order_relaxed));
        }
        std::sort(rows.begin(), rows.end());
        std::string out;
        for (const auto &r : rows) out += std::to_string(r.second) + "\t" + r.first + "\n";
        if (buf && cap > 0) {
            const size_t n = std::min<size_t>((size_t)cap - 1, out.size());
            std::memcpy(buf, out.data(), n);
            buf[n] = 0;
        }
        return (int64_t)out.size();
    } catch (...) {
        return -1;
    }
}

int blhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

blhip_ctx *blhip_create(int device) {
    blhip_ctx *ctx = nullptr;
    try {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) fail("no HIP device visible");
        if (device < 0 || device >= n) fail("device %d out of range (%d visible)", device, n);
        HIPCHECK(hipSetDevice(device));
        ctx = new blhip_ctx();
        ctx->device = device;
        HIPCHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        for (auto &e : ctx->ev) HIPCHECK(hipEventCreate(&e));
        for (auto &bs : ctx->bstream) HIPCHECK(hipStreamCreateWithFlags(&bs, hipStreamNonBlocking));
        for (auto &e : ctx->bev) HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIPCHECK(hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming));
        HIPCHECK(hipEventCreateWithFlags(&ctx->sync_ev, hipEventDisableTiming));
        HIPCHECK(hipStreamCreateWithFlags(&ctx->cstream, hipStreamNonBlocking));
        for (auto &e : ctx->cev) HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        hipDeviceProp_t prop;
        HIPCHECK(hipGetDeviceProperties(&prop, device));
        ctx->num_cus = prop.multiProcessorCount;
        std::string marketing = prop.name;
        if (marketing.find_first_not_of(' ') == std::string::npos) marketing = "AMD Instinct (name not reported by the driver)";
        ctx->name = marketing + " (" + prop.gcnArchName + ")";
        return ctx;
    } catch (const Fail &e) {
        g_create_error = e.msg;
    } catch (...) {
        g_create_error = "unknown error in blhip_create";
    }
    delete ctx;
    return nullptr;
}

void blhip_destroy(blhip_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)blhip_comm_destroy(ctx);
    ctx->commbuf.release(); ctx->pinC.release(); ctx->resx.release(); ctx->post2.release(); ctx->postpad.release(); ctx->accw.release(); ctx->accpart.release(); ctx->xch.release(); ctx->axlik.release(); ctx->anchbuf.release(); ctx->probebuf.release(); ctx->p1d.release(); ctx->p1w.release(); ctx->lik1d.release();
    if (ctx->astream) { (void)hipStreamSynchronize(ctx->astream); (void)hipStreamDestroy(ctx->astream); }
    for (auto &e : ctx->aev_done) if (e) (void)hipEventDestroy(e);
    for (DevBuf *b : {&ctx->state, &ctx->post, &ctx->psumF, &ctx->psumB, &ctx->redF, &ctx->redB, &ctx->meta,
                      &ctx->tables, &ctx->likbuf, &ctx->small, &ctx->accum_own, &ctx->stats, &ctx->mix, &ctx->unit, &ctx->databuf, &ctx->postinv})
        b->release();
    for (auto &kv : ctx->carry) kv.second.buf.release();
    ctx->pinF.release(); ctx->pinB.release(); ctx->pinS.release(); ctx->pinA.release();
    for (auto &e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &bs : ctx->bstream)
        if (bs) { (void)hipStreamSynchronize(bs); (void)hipStreamDestroy(bs); }
    for (auto &e : ctx->bev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->fork_ev) (void)hipEventDestroy(ctx->fork_ev);
    if (ctx->sync_ev) (void)hipEventDestroy(ctx->sync_ev);
    if (ctx->cstream) { (void)hipStreamSynchronize(ctx->cstream); (void)hipStreamDestroy(ctx->cstream); }
    for (auto &e : ctx->cev) if (e) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *blhip_last_error(blhip_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int blhip_device_name(blhip_ctx *ctx, char *buf, int buflen) {
    if (!ctx || !buf || buflen <= 0) return -1;
    std::snprintf(buf, (size_t)buflen, "%s", ctx->name.c_str());
    return 0;
}

int blhip_host_unlag(int scheme, double *sums, int64_t T, int lag, const unsigned char *kinds, double *scales_out) {
    if (!sums || T < 1 || lag < 1) return -1;
    try {
        std::vector<double> red((size_t)T * NRED, 0.0), rowsum, scales;
        for (int64_t t = 0; t < T; ++t) red[(size_t)t * NRED] = sums[t];
        const bool ok = scheme == 0 ? resident_unlag(red.data(), T, lag, rowsum) : chain_unlag(red.data(), T, lag, rowsum, 1, 0, &scales, kinds);
        if (!ok) return 1;
        for (int64_t t = 0; t < T; ++t) sums[t] = red[(size_t)t * NRED];
        if (scales_out)
            for (int64_t t = 0; t < T; ++t) scales_out[t] = scheme == 0 ? (t >= lag ? 1.0 / rowsum[t - lag] : 1.0) : scales[t];
        return 0;
    } catch (...) {
        return -1;
    }
}

int blhip_set_option(blhip_ctx *ctx, const char *key, double value) {
    if (!ctx || !key) return -1;
    // "resident_ok": the context's memory of a resident launch that gave up (its blocks were not all co-resident) -- settable so that
    // a caller (the tests of the fall-back) can re-arm the resident paths
    if (std::strcmp(key, "resident_ok") == 0) { ctx->resident_ok = value != 0.0; ctx->resident_fits_since = 0; if (value != 0.0) { ctx->resident_retry_after = 8; ctx->resident_giveups = 0; } return 0; }
    if (std::strcmp(key, "resident_retry_after") == 0) { ctx->resident_retry_after = std::max(1, (int)value); return 0; }
    // (a key the library never reads is an error, not a silent no-op: an A/B run over a removed option measured nothing -- ADVICE r05)
    static const char *const known[] = {
        "accum_overlap", "carry_partials", "chain1d", "chain1d_clamp", "chain1d_shift", "chain_ax1", "chain_depad", "chain_prof", "chain_resident", "chain_resident_lag",
        "chain_table", "chain_wide", "comm_reduce_mode", "fast", "fast_S", "fold2", "fold2_cp", "fold_force_fail_batch", "fuse1d", "fuse_accumulate", "late_sums", "max_batch",
        "mem_budget_bytes", "mfma", "mfma_S", "mfma_h", "mfma_h_max_cells", "peer_copy_mode", "persist1d", "quiet", "recurrence", "resident",
        "resident_force_abort", "resident_lag", "resident_probe", "resident_probe_force_busy", "resident_probe_interval_s", "resident_probe_timeout_s",
        "resident_table", "resident_timeout_s", "share_prefix", "short_first_batch", "skip_prefix", "trace", "wide_h",
        "wide_h_fused_max", "wide_h_split", "wide_v"};
    bool ok = false;
    for (const char *k : known) ok = ok || std::strcmp(key, k) == 0;
    if (!ok) {
        ctx->err = std::string("blhip_set_option: unknown option '") + key + "' (README.md lists the options the library reads)";
        return -1;
    }
    ctx->opt[key] = value;
    return 0;
}

int blhip_synchronize(blhip_ctx *ctx) {
    return guarded(ctx, [&] {
        HIPCHECK(hipSetDevice(ctx->device));
        sync_stream(ctx, ctx->stream);
    });
}

int blhip_fit(blhip_ctx *ctx, const blhip_problem *problem, int64_t n_chains, const double *op_values,
              const double *log_chain_weight, uint32_t flags, blhip_result *result) {
    return guarded(ctx, [&] { do_fit(ctx, problem, n_chains, op_values, log_chain_weight, flags, result); });
}

int blhip_bandwidth_probe(blhip_ctx *ctx, int64_t bytes, int iterations, double *gb_per_s) {
    return guarded(ctx, [&] {
        if (!gb_per_s || bytes < (1 << 20) || iterations < 1) fail("blhip_bandwidth_probe: bad arguments");
        HIPCHECK(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        const long long n2 = bytes / 16;
        ScopedDevBuf a, b;          // (released on every exit path, also when a HIPCHECK below throws)
        a.ensure((size_t)n2 * 16); b.ensure((size_t)n2 * 16);
        BL_LAUNCH(fill_kernel, dim3(2048), dim3(256), 0, st, a.as<double>(), n2 * 2, 1.0);
        const unsigned gx = (unsigned)((n2 + 4 * NTHREADS - 1) / (4 * NTHREADS));
        double best = 0.0;
        for (int variant = 0; variant < 2; ++variant) {
            auto go = [&](const double2 *src, double2 *dst) {
                if (variant) BL_LAUNCH(copy16_kernel<true>, dim3(gx), dim3(NTHREADS), 0, st, src, dst, n2);
                else BL_LAUNCH(copy16_kernel<false>, dim3(gx), dim3(NTHREADS), 0, st, src, dst, n2);
            };
            go(a.as<double2>(), b.as<double2>());                                                   // warm-up
            HIPCHECK(hipEventRecord(ctx->ev[4], st));
            for (int k = 0; k < iterations; ++k) go((k & 1) ? b.as<double2>() : a.as<double2>(), (k & 1) ? a.as<double2>() : b.as<double2>());
            HIPCHECK(hipEventRecord(ctx->ev[5], st));
            HIPCHECK(hipGetLastError());
            sync_stream(ctx, st);
            float ms = 0;
            HIPCHECK(hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5]));
            best = std::max(best, 2.0 * (double)n2 * 16.0 * iterations / ((double)ms * 1e-3) / 1e9);
        }
        for (int variant = 0; variant < 2; ++variant) {
            // a store-only stream (what the no-stencil forward chain kernel does) can run above the copy rate: the calibrated peak is
            // the best of the streams measured
            auto go = [&](double2 *dst, double v) {
                if (variant) BL_LAUNCH(fill16_kernel<true>, dim3(gx), dim3(NTHREADS), 0, st, dst, n2, v);
                else BL_LAUNCH(fill16_kernel<false>, dim3(gx), dim3(NTHREADS), 0, st, dst, n2, v);
            };
            go(b.as<double2>(), 2.0);
            HIPCHECK(hipEventRecord(ctx->ev[4], st));
            for (int k = 0; k < iterations; ++k) go((k & 1) ? a.as<double2>() : b.as<double2>(), 3.0);
            HIPCHECK(hipEventRecord(ctx->ev[5], st));
            HIPCHECK(hipGetLastError());
            sync_stream(ctx, st);
            float ms = 0;
            HIPCHECK(hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5]));
            best = std::max(best, (double)n2 * 16.0 * iterations / ((double)ms * 1e-3) / 1e9);
        }
        *gb_per_s = best;
    });
}

int blhip_last_timing(blhip_ctx *ctx, blhip_timing *out) {
    if (!ctx || !out) return -1;
    *out = ctx->timing;
    out->resident_armed = ctx->resident_ok ? 1 : 0;
    return 0;
}

void *blhip_host_alloc(size_t bytes) {
    if (bytes == 0) return nullptr;
    const size_t len = (bytes + 4095) / 4096 * 4096;
    void *p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) return nullptr;
    const size_t nchunk = (len + PIN_CHUNK - 1) / PIN_CHUNK;
    PinnedBlock blk{len, std::vector<char>(nchunk, 0)};
    int nthreads = 1;      // (measured, 16 GiB: 1 thread 1.06 s, 4: 1.8 s, 16: 2.1 s -- the page-table lock serialises them; hipHostMalloc: 2.2 s)
    if (const char *e = std::getenv("BLHIP_PIN_THREADS")) nthreads = std::max(1, std::atoi(e));
    nthreads = (int)std::min<size_t>((size_t)nthreads, nchunk);
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::atomic<size_t> next{0};
    auto work = [&]() {
        (void)hipSetDevice(dev);
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= nchunk) break;
            char *c = (char *)p + k * PIN_CHUNK;
            const size_t n = std::min(PIN_CHUNK, len - k * PIN_CHUNK);
            if (hipHostRegister(c, n, hipHostRegisterDefault) == hipSuccess) blk.ok[k] = 1;
            else (void)hipGetLastError();
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    bool all = true;
    for (char o : blk.ok) all = all && o;
    if (!all) {
        for (size_t k = 0; k < nchunk; ++k) if (blk.ok[k]) (void)hipHostUnregister((char *)p + k * PIN_CHUNK);
        munmap(p, len);
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_pin_mu);
    g_pinned[p] = std::move(blk);
    return p;
}

void blhip_host_free(void *p) {
    if (!p) return;
    PinnedBlock blk;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        auto it = g_pinned.find(p);
        if (it == g_pinned.end()) return;
        blk = std::move(it->second);
        g_pinned.erase(it);
    }
    for (size_t k = 0; k < blk.ok.size(); ++k) (void)hipHostUnregister((char *)p + k * PIN_CHUNK);
    munmap(p, blk.bytes);
}

int blhip_posterior_read(blhip_ctx *ctx, int64_t chain, int64_t t0, int64_t t1, double *host_out) {
    return guarded(ctx, [&] {
        if (!ctx->post_valid) fail("no posterior kept (run blhip_fit with BLHIP_KEEP_POSTERIOR)");
        ensure_post_scaled(ctx);
        if (chain < 0 || chain >= ctx->post_chains || t0 < 0 || t1 > ctx->post_T || t0 > t1 || !host_out)
            fail("blhip_posterior_read: bad range");
        HIPCHECK(hipSetDevice(ctx->device));
        const char *src = reinterpret_cast<const char *>(ctx->post.as<double>() + ((size_t)chain * ctx->post_T + t0) * ctx->post_G);
        for_pinned_pieces(host_out, (size_t)(t1 - t0) * ctx->post_G * 8, [&](char *dst, size_t n) {
            HIPCHECK(hipMemcpyAsync(dst, src + (dst - (char *)host_out), n, hipMemcpyDeviceToHost, ctx->stream));
        });
        sync_stream(ctx, ctx->stream);
    });
}

int blhip_posterior_devptr(blhip_ctx *ctx, void **devptr, int64_t *chain_stride, int64_t *step_stride) {
    return guarded(ctx, [&] {
        if (!ctx->post_valid) fail("no posterior kept");
        ensure_post_scaled(ctx);
        sync_stream(ctx, ctx->stream);
        if (devptr) *devptr = ctx->post.p;
        if (chain_stride) *chain_stride = ctx->post_T * ctx->post_G;
        if (step_stride) *step_stride = ctx->post_G;
    });
}

int blhip_posterior_release(blhip_ctx *ctx) {
    return guarded(ctx, [&] {
        HIPCHECK(hipSetDevice(ctx->device));
        ctx->post_valid = false;
        ctx->post.release();
        ctx->post2.release();
    });
}

namespace {
struct SeqView { const double *p; int64_t T; int n0, n1; };
SeqView sequence_view(blhip_ctx *ctx, int source, int64_t chain) {
    if (source == 0) {
        if (!ctx->post_valid) fail("no posterior kept (run blhip_fit with BLHIP_KEEP_POSTERIOR)");
        if (chain < 0 || chain >= ctx->post_chains) fail("chain out of range");
        ensure_post_scaled(ctx);
        return SeqView{ctx->post.as<double>() + (size_t)chain * ctx->post_T * ctx->post_G, ctx->post_T, ctx->post_n0, ctx->post_n1};
    }
    if (source == 1) {
        if (!ctx->acc_active || !ctx->acc_final) fail("accumulator not finalised");
        return SeqView{ctx->acc, ctx->acc_T, ctx->acc_n0, ctx->acc_n1};
    }
    fail("bad source %d", source);
}
}  // namespace

int blhip_posterior_marginal(blhip_ctx *ctx, int source, int64_t chain, int keep_axis, double *host_out) {
    return guarded(ctx, [&] {
        if (!host_out) fail("host_out is NULL");
        HIPCHECK(hipSetDevice(ctx->device));
        const SeqView v = sequence_view(ctx, source, chain);
        const bool one_d = v.n0 == 1;
        if (keep_axis < 0 || keep_axis > (one_d ? 0 : 1)) fail("keep_axis out of range");
        const int nk = one_d ? v.n1 : (keep_axis == 0 ? v.n0 : v.n1);
        ctx->stats.ensure((size_t)v.T * nk * 8);
        double *d_out = ctx->stats.as<double>();
        hipStream_t st = ctx->stream;
        if (one_d) {
            HIPCHECK(hipMemcpyAsync(d_out, v.p, (size_t)v.T * nk * 8, hipMemcpyDeviceToDevice, st));
        } else if (keep_axis == 0) {
            BL_LAUNCH(marginal_rows_kernel, dim3(v.n0, (unsigned)v.T), dim3(NTHREADS), 0, st, v.p, d_out, v.n0, v.n1);
        } else {
            BL_LAUNCH(marginal_cols_kernel, dim3((v.n1 + NTHREADS - 1) / NTHREADS, (unsigned)v.T), dim3(NTHREADS), 0, st,
                               v.p, d_out, v.n0, v.n1);
        }
        HIPCHECK(hipMemcpyAsync(host_out, d_out, (size_t)v.T * nk * 8, hipMemcpyDeviceToHost, st));
        sync_stream(ctx, st);
    });
}

int blhip_posterior_time_average(blhip_ctx *ctx, int source, int64_t chain, double *host_out) {
    return guarded(ctx, [&] {
        if (!host_out) fail("host_out is NULL");
        HIPCHECK(hipSetDevice(ctx->device));
        const SeqView v = sequence_view(ctx, source, chain);
        const long long G = (long long)v.n0 * v.n1;
        ctx->stats.ensure((size_t)G * 8);
        double *d_out = ctx->stats.as<double>();
        hipStream_t st = ctx->stream;
        BL_LAUNCH(time_average_kernel, dim3((unsigned)std::min<long long>((G + NTHREADS - 1) / NTHREADS, 4096)),
                           dim3(NTHREADS), 0, st, v.p, d_out, G, (int)v.T);
        HIPCHECK(hipMemcpyAsync(host_out, d_out, (size_t)G * 8, hipMemcpyDeviceToHost, st));
        sync_stream(ctx, st);
    });
}

int blhip_carry_mix(blhip_ctx *ctx, int slot, int64_t n_chains, const double *weights, int accumulate) {
    return guarded(ctx, [&] {
        if (!weights) fail("weights is NULL");
        HIPCHECK(hipSetDevice(ctx->device));
        auto it = ctx->carry.find(slot);
        if (it == ctx->carry.end() || !it->second.valid) fail("blhip_carry_mix: carry slot %d holds no state", slot);
        blhip_ctx::Carry &cs = it->second;
        if (cs.chains != n_chains) fail("blhip_carry_mix: slot %d holds %lld chains, got %lld weights", slot, (long long)cs.chains, (long long)n_chains);
        if (accumulate && ctx->mix_G != cs.G) fail("blhip_carry_mix: accumulating %lld cells into a mix of %lld", (long long)cs.G, (long long)ctx->mix_G);
        hipStream_t st = ctx->stream;
        ctx->mix.ensure((size_t)cs.G * 8);
        ctx->mix_G = cs.G;
        ctx->small.ensure((size_t)n_chains * 8);
        HIPCHECK(hipMemcpyAsync(ctx->small.p, weights, (size_t)n_chains * 8, hipMemcpyHostToDevice, st));
        const unsigned gx = (unsigned)std::min<long long>((cs.G + NTHREADS - 1) / NTHREADS, 8192);
        BL_LAUNCH(carry_mix_kernel, dim3(gx), dim3(NTHREADS), 0, st, ctx->mix.as<double>(), cs.buf.as<double>(),
                           (long long)cs.G, (int)n_chains, ctx->small.as<double>(), accumulate ? 1 : 0);
        HIPCHECK(hipGetLastError());
        sync_stream(ctx, st);
    });
}

int blhip_carry_read(blhip_ctx *ctx, int slot, int64_t chain, double *host_out) {
    return guarded(ctx, [&] {
        if (!host_out) fail("host_out is NULL");
        HIPCHECK(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        if (chain < 0) {
            if (ctx->mix_G <= 0) fail("blhip_carry_read: no mix computed");
            HIPCHECK(hipMemcpyAsync(host_out, ctx->mix.p, (size_t)ctx->mix_G * 8, hipMemcpyDeviceToHost, st));
        } else {
            auto it = ctx->carry.find(slot);
            if (it == ctx->carry.end() || !it->second.valid) fail("blhip_carry_read: carry slot %d holds no state", slot);
            if (chain >= it->second.chains) fail("blhip_carry_read: chain %lld of %lld", (long long)chain, (long long)it->second.chains);
            HIPCHECK(hipMemcpyAsync(host_out, it->second.buf.as<double>() + (size_t)chain * it->second.G, (size_t)it->second.G * 8,
                                    hipMemcpyDeviceToHost, st));
        }
        sync_stream(ctx, st);
    });
}

int blhip_carry_write(blhip_ctx *ctx, int slot, int64_t n_chains, int64_t G, const double *host_in) {
    return guarded(ctx, [&] {
        if (!host_in || n_chains < 1 || G < 1 || slot < 0) fail("blhip_carry_write: bad arguments");
        HIPCHECK(hipSetDevice(ctx->device));
        blhip_ctx::Carry &cs = ctx->carry[slot];
        cs.buf.ensure((size_t)n_chains * G * 8);
        HIPCHECK(hipMemcpyAsync(cs.buf.p, host_in, (size_t)n_chains * G * 8, hipMemcpyHostToDevice, ctx->stream));
        sync_stream(ctx, ctx->stream);
        cs.chains = n_chains; cs.G = G; cs.valid = true;
        cs.maxv.assign(n_chains, 0.0);           // (NotEqual inverts around the maximum of the carried state)
        for (int64_t b = 0; b < n_chains; ++b) {
            double m = -INFINITY;
            for (int64_t c = 0; c < G; ++c) m = std::max(m, host_in[b * G + c]);
            cs.maxv[b] = m;
        }
    });
}

int blhip_carry_release(blhip_ctx *ctx, int slot) {
    return guarded(ctx, [&] {
        HIPCHECK(hipSetDevice(ctx->device));
        if (slot < 0) {
            for (auto &kv : ctx->carry) kv.second.buf.release();
            ctx->carry.clear(); ctx->mix.release(); ctx->mix_G = 0;
        } else {
            auto it = ctx->carry.find(slot);
            if (it != ctx->carry.end()) { it->second.buf.release(); ctx->carry.erase(it); }
        }
    });
}

int blhip_accum_begin(blhip_ctx *ctx, int64_t T, int64_t G, void *external_devptr) {
    return guarded(ctx, [&] {
        if (T < 1 || G < 1) fail("blhip_accum_begin: bad shape");
        HIPCHECK(hipSetDevice(ctx->device));
        if (external_devptr) {
            ctx->acc = reinterpret_cast<double *>(external_devptr);
        } else {
            ctx->accum_own.ensure((size_t)T * G * 8);
            ctx->acc = ctx->accum_own.as<double>();
        }
        ctx->acc_T = T; ctx->acc_G = G; ctx->acc_folded = 0;
        ctx->acc_logref = -std::numeric_limits<double>::infinity();
        ctx->acc_active = true; ctx->acc_final = false; ctx->acc_first = true;
    });
}

int blhip_accum_state(blhip_ctx *ctx, double *log_ref, void **devptr, int64_t *n_folded) {
    return guarded(ctx, [&] {
        if (!ctx->acc_active) fail("no active accumulator");
        if (log_ref) *log_ref = ctx->acc_logref;
        if (devptr) *devptr = ctx->acc;
        if (n_folded) *n_folded = ctx->acc_folded;
    });
}

int blhip_accum_rescale(blhip_ctx *ctx, double new_log_ref) {
    return guarded(ctx, [&] {
        if (!ctx->acc_active) fail("no active accumulator");
        HIPCHECK(hipSetDevice(ctx->device));
        const long long n = (long long)ctx->acc_T * ctx->acc_G;
        if (ctx->acc_first) {
            // nothing folded on this rank: contributes zeros
            BL_LAUNCH(fill_kernel, dim3(1024), dim3(256), 0, ctx->stream, ctx->acc, n, 0.0);
            ctx->acc_first = false;
        } else {
            if (new_log_ref < ctx->acc_logref) fail("blhip_accum_rescale: new reference below the current one");
            const double r = std::exp(ctx->acc_logref - new_log_ref);
            if (r != 1.0)
                BL_LAUNCH(scale_all_kernel, dim3(2048), dim3(NTHREADS), 0, ctx->stream, ctx->acc, n, r);
        }
        ctx->acc_logref = new_log_ref;
        sync_stream(ctx, ctx->stream);
    });
}

int blhip_accum_fold_host(blhip_ctx *ctx, const double *posterior, double log_weight) {
    return guarded(ctx, [&] {
        if (!ctx->acc_active) fail("blhip_accum_fold_host without blhip_accum_begin");
        if (ctx->acc_final) fail("blhip_accum_fold_host: the accumulator is finalised");
        if (!posterior) fail("posterior is NULL");
        if (!std::isfinite(log_weight)) return;                 // np.isfinite(logEvidence) guard, a zero hyper-prior (core.py:1358, :1364)
        HIPCHECK(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        const int64_t T = ctx->acc_T;
        const long long G = ctx->acc_G;
        ctx->post_valid = false;                                  // (the sequence buffer is the staging area)
        ctx->post.ensure((size_t)T * G * 8);
        for_pinned_pieces(const_cast<double *>(posterior), (size_t)T * G * 8, [&](char *src, size_t n) {
            HIPCHECK(hipMemcpyAsync(ctx->post.as<char>() + (src - (const char *)posterior), src, n, hipMemcpyHostToDevice, st));
        });
        BatchOutcome O;
        O.logE.assign(1, log_weight);
        O.abort_step.assign(1, -1);
        O.abort_phase.assign(1, 0);
        O.invN.assign((size_t)T, 1.0);                             // (rows arrive normalised)
        const double zero = 0.0;
        ctx->small.ensure(carve_size(8) + carve_size((size_t)T * 8));
        char *cur = ctx->small.as<char>();
        double *d_w = carve<double>(cur, 1), *d_invN = carve<double>(cur, (size_t)T);
        fold_accumulate(ctx, T, G, 1, O, &zero, ctx->post.as<double>(), d_w, d_invN);
    });
}

namespace {
// per-step sums [sum A, sum A grid_0, sum A grid_1] of the accumulator -> page-locked host memory (T, 3) (+ T spare doubles
// behind them); d_inv: T doubles of device scratch
struct RowStats { double *red, *d_inv; };      // red: (T, RS_W) page-locked, + T spare doubles behind it
constexpr int RS_W = 1 + bln::MAXD;
RowStats accum_row_stats(blhip_ctx *ctx, const blhip_problem *p) {
    if (!ctx->acc_active) fail("no active accumulator");
    if (!p) fail("problem is NULL");
    HIPCHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t T = ctx->acc_T, G = ctx->acc_G;
    bln::NdGrid ng{};
    ng.ndim = p->ndim;
    long long Gp = 1;
    for (int k = p->ndim - 1; k >= 0; --k) { ng.n[k] = (int)p->n[k]; ng.stride[k] = Gp; Gp *= p->n[k]; }
    ng.G = Gp;
    if (Gp != G) fail("accumulator / grid mismatch");
    const unsigned gx = (unsigned)std::min<long long>((G + NTHREADS - 1) / NTHREADS, 256);
    size_t bytes = carve_size((size_t)T * RS_W * gx * 8) + carve_size((size_t)T * RS_W * 8) + carve_size((size_t)T * 8);
    for (int k = 0; k < p->ndim; ++k) bytes += carve_size(8 * (size_t)p->n[k]);
    ctx->stats.ensure(bytes);
    char *cur = ctx->stats.as<char>();
    double *d_part = carve<double>(cur, (size_t)T * RS_W * gx);
    double *d_red = carve<double>(cur, (size_t)T * RS_W);
    double *d_inv = carve<double>(cur, (size_t)T);
    for (int k = 0; k < p->ndim; ++k) {
        double *dm = carve<double>(cur, (size_t)p->n[k]);
        HIPCHECK(hipMemcpyAsync(dm, p->marginal[k], 8 * (size_t)p->n[k], hipMemcpyHostToDevice, st));
        ng.m[k] = dm;
    }
    BL_LAUNCH(bln::row_stats_kernel, dim3(gx, (unsigned)T), dim3(NTHREADS), 0, st, ctx->acc, ng, d_part);
    BL_LAUNCH(reduce_partials_kernel, dim3((unsigned)(T * RS_W)), dim3(NTHREADS), 0, st, d_part, d_red, (int)gx, 0);
    ctx->pinS.ensure((size_t)T * (RS_W + 1) * 8);
    double *red = ctx->pinS.as<double>();
    HIPCHECK(hipMemcpyAsync(red, d_red, (size_t)T * RS_W * 8, hipMemcpyDeviceToHost, st));
    sync_stream(ctx, st);
    ctx->acc_n0 = p->ndim == 1 ? 1 : (int)p->n[0];
    ctx->acc_n1 = (int)(G / ctx->acc_n0);
    return RowStats{red, d_inv};
}
}  // namespace

int blhip_accum_row_stats(blhip_ctx *ctx, const blhip_problem *p, double *host_out) {
    return guarded(ctx, [&] {
        if (!host_out) fail("host_out is NULL");
        if (!ctx->acc_active) fail("no active accumulator");
        if (!p) fail("problem is NULL");
        const int64_t T = ctx->acc_T;
        const int w = 1 + p->ndim;
        if (ctx->acc_first) { std::fill(host_out, host_out + T * w, 0.0); return; }
        const double *red = accum_row_stats(ctx, p).red;
        for (int64_t t = 0; t < T; ++t)
            for (int k = 0; k < w; ++k) host_out[t * w + k] = red[t * RS_W + k];
    });
}

int blhip_accum_finalize(blhip_ctx *ctx, const blhip_problem *p, double *posterior_mean) {
    return guarded(ctx, [&] {
        if (!ctx->acc_active) fail("no active accumulator");
        if (ctx->acc_first) fail("blhip_accum_finalize: nothing was accumulated");
        const RowStats rs = accum_row_stats(ctx, p);
        double *red = rs.red, *d_inv = rs.d_inv;
        hipStream_t st = ctx->stream;
        const int64_t T = ctx->acc_T, G = ctx->acc_G;
        double *inv = red + (size_t)T * RS_W;
        for (int64_t t = 0; t < T; ++t) {
            inv[t] = 1.0 / red[t * RS_W];                                        // core.py:1379-1382
            if (posterior_mean)
                for (int k = 0; k < p->ndim; ++k) posterior_mean[k * T + t] = red[t * RS_W + 1 + k] / red[t * RS_W];   // :1416-1419
        }
        HIPCHECK(hipMemcpyAsync(d_inv, inv, T * 8, hipMemcpyHostToDevice, st));
        const unsigned gs = (unsigned)std::min<long long>((G + NTHREADS - 1) / NTHREADS, 4096);
        BL_LAUNCH(scale_rows_kernel, dim3(gs, (unsigned)T), dim3(NTHREADS), 0, st, ctx->acc, (long long)G, d_inv);
        sync_stream(ctx, st);
        ctx->acc_final = true;
    });
}

int blhip_accum_read(blhip_ctx *ctx, int64_t t0, int64_t t1, double *host_out) {
    return guarded(ctx, [&] {
        if (!ctx->acc_active) fail("no active accumulator");
        if (t0 < 0 || t1 > ctx->acc_T || t0 > t1 || !host_out) fail("blhip_accum_read: bad range");
        HIPCHECK(hipSetDevice(ctx->device));
        const char *src = reinterpret_cast<const char *>(ctx->acc + (size_t)t0 * ctx->acc_G);
        for_pinned_pieces(host_out, (size_t)(t1 - t0) * ctx->acc_G * 8, [&](char *dst, size_t n) {
            HIPCHECK(hipMemcpyAsync(dst, src + (dst - (char *)host_out), n, hipMemcpyDeviceToHost, ctx->stream));
        });
        sync_stream(ctx, ctx->stream);
    });
}

int blhip_accum_end(blhip_ctx *ctx) {
    return guarded(ctx, [&] {
        ctx->acc_active = false;
        ctx->acc = nullptr;
    });
}

}  // extern "C"
