// The resident paths of a batch (kernels: blhip_resident.hpp, blhip_chainres.hpp): eligibility, buffers, launches and the host-side
// checks after each pass, one struct per path.  Included by blhip.hip INSIDE its anonymous namespace, after the helpers it uses
// (DeviceTables, DeviceMeta, ChainProgram, TapTable, plan_ / launch_ functions, resident_unlag / chain_unlag, BatchOutcome).
#pragma once

// ---- what the resident paths of a batch need to know about it -------------------------------------------------------------------------
struct BatchEnv {
    blhip_ctx *ctx; const blhip_problem *p; hipStream_t st;
    Geometry g; long long G; int64_t T, B, c0; int d, rec_len;
    FitFlags ff;
    const DeviceTables *DT; const DeviceMeta *M; const ChainProgram *prog; const TapTable *taps;
    double step0;                 // lattice step of the row axis (likelihood recurrence)
    double *d_post;               // the batch's sequence buffer (null: evidence-only)
    const double *log_w;          // log weights of ALL chains of the call (accumulate)
    bool chain_means;             // the caller asked for per-chain posterior means
    bool overlap_acc;             // folds of earlier batches may still run on the second stream (option accum_overlap)
    std::vector<hipEvent_t> *fold_ev = nullptr;      // start / end events of fold kernels nobody waits for (do_fit reads them after its last batch)
    int64_t bi = 0;               // index of the batch in the call
    bool allow_chainres = true;   // false: a repeated batch (its first attempt poisoned the carried partial accumulators)
    Trace *tr = nullptr;          // option trace: host phases on stderr
    void mark(const char *what) const { if (tr) tr->mark(what); }
};

// The resident backward kernels normalise every posterior by a PREDICTED sum (self-adjoint stencil identity); the host accepts the
// batch only if the prediction reproduces the reduced sums to this relative tolerance, three decades inside the parity bar of the
// posteriors (1e-9) -- the prediction is a running product of two rounded factors per step, so its error grows with the number of
// steps: a few ulp per step at worst (observed ~1e-14 at T = 2000); else: the launch-per-step kernels
inline double pred_rtol(int64_t T) { return 1e-12 + 2e-15 * (double)T; }

// Bound of every in-kernel wait of a resident launch.  A block that is not co-resident never arrives, so the bound IS the cost of the
// failure a user on a shared GPU meets first: it scales with the pass (20 x the predicted pass time at ~15 us per step, at least 250 ms)
// instead of a flat 2 s.  (50 ms was tried: four test processes sharing ONE GPU -- pytest -n 4 -- starve each other's blocks for longer than
// that and the paths gave up spuriously.)  Option resident_timeout_s > 0 overrides.
inline double resident_timeout_s(blhip_ctx *ctx, int64_t T) {
    const double opt = ctx->option("resident_timeout_s", 0.0);
    return opt > 0.0 ? opt : std::max(0.25, 20.0 * 15e-6 * (double)T);
}

// did a block of a resident launch time out waiting for a peer (not every block co-resident)?  -> the context stops using the paths
bool resident_gave_up(blhip_ctx *ctx, hipStream_t st, const unsigned *d_abort) {
    ctx->pinS.ensure(64);
    unsigned *h = reinterpret_cast<unsigned *>(ctx->pinS.as<char>());
    HIPCHECK(hipMemcpyAsync(h, d_abort, 4, hipMemcpyDeviceToHost, st));
    sync_stream(ctx, st);
    // (option resident_force_abort: the tests of the fall-back pretend that a block gave up)
    if (*h != 0u || ctx->option("resident_force_abort", 0.0) != 0.0) {
        ctx->resident_last_reason = *h != 0u ? BLHIP_FALLBACK_GAVE_UP : BLHIP_FALLBACK_FORCED;
        if (ctx->resident_ok && ctx->option("quiet", 0.0) == 0.0)
            std::fprintf(stderr, "[blhip] a resident launch gave up waiting for a peer block (its blocks were not all co-resident: a shared or "
                                 "partitioned GPU?); this batch is repeated and the context continues with the launch-per-step kernels\n");
        ctx->resident_ok = false;
        ctx->resident_fits_since = 0;
        ctx->resident_giveups += 1;
        if (ctx->resident_giveups > 1) ctx->resident_retry_after = std::min(1024, 2 * ctx->resident_retry_after);
        return true;
    }
    return false;
}

// Is the chip ours?  A grid of one block per CU with the resident kernels' footprint; every block counts itself in and waits (bounded)
// for the others.  -> false: not all of them were on the chip together -- another process holds CUs, or the device is partitioned: a
// resident launch would sit out its time-out (>= 0.25 s) before the fit fell back.  ~15 us when the chip is free.
bool chip_is_ours(blhip_ctx *ctx) {
    hipStream_t st = ctx->stream;
    ctx->pinS.ensure((8 + 256) * sizeof(unsigned));
    unsigned *h = reinterpret_cast<unsigned *>(ctx->pinS.as<char>());
    DevBuf &d = ctx->probebuf;
    const size_t pbytes = (8 + 256) * sizeof(unsigned);
    d.ensure(pbytes);
    HIPCHECK(hipMemsetAsync(d.p, 0, pbytes, st));
    const unsigned nb = (unsigned)std::min(ctx->num_cus, 256);
    const size_t lds = 150 * 1024;
    arm_kernel(reinterpret_cast<const void *>(&blr::residency_probe_kernel<0>));
    const unsigned long long ticks = (unsigned long long)(std::max(1e-4, ctx->option("resident_probe_timeout_s", 0.002)) * 1e8);
    BL_LAUNCH(blr::residency_probe_kernel<0>, dim3(nb), dim3(512), lds, st, d.as<unsigned>(), nb, ticks);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(h, d.p, pbytes, hipMemcpyDeviceToHost, st));
    sync_stream(ctx, st);
    // the dispatch order the both-axes kernels count on: blocks b, b + 8, b + 16, ... on ONE XCD, eight different ones for b = 0 .. 7
    bool order = nb >= 8;
    for (unsigned b = 0; b < nb && order; ++b) order = h[8 + b] != 0u && h[8 + b] == h[8 + (b & 7u)];
    for (unsigned a = 0; a < 8 && order; ++a)
        for (unsigned c = a + 1; c < 8 && order; ++c) order = h[8 + a] != h[8 + c];
    ctx->xcd_order_ok = order;
    return h[0] == nb && h[1] == 0u && ctx->option("resident_probe_force_busy", 0.0) == 0.0;
}

// Before a fit that may take a resident path: on the context's first fit, when the paths are re-armed after a give-up, and at most once
// per resident_probe_interval_s (1 s) otherwise.  A busy chip parks the resident paths exactly as a give-up does (retry after 8, 16, ..
// fits) -- at the price of the probe's bound (2 ms), not of a launch's.
void probe_residency(blhip_ctx *ctx) {
    if (!ctx->resident_ok || ctx->option("resident_probe", 1.0) == 0.0) return;
    const double now = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    if (ctx->probe_last >= 0.0 && now - ctx->probe_last < ctx->option("resident_probe_interval_s", 1.0)) return;
    ctx->probe_last = now;
    const bool ours = chip_is_ours(ctx);
    ctx->timing.resident_probe = ours ? 1 : 2;
    if (ours) return;
    if (!ctx->probe_announced && ctx->option("quiet", 0.0) == 0.0)
        std::fprintf(stderr, "[blhip] the GPU is not exclusively ours (a one-block-per-CU probe did not get all its blocks onto the chip within "
                             "%.1f ms: another process, or a partitioned device); fits run on the launch-per-step kernels until a later probe finds it free\n",
                     1e3 * std::max(1e-4, ctx->option("resident_probe_timeout_s", 0.002)));
    ctx->probe_announced = true;
    ctx->resident_ok = false;
    ctx->resident_fits_since = 0;
    ctx->resident_giveups += 1;
    ctx->resident_last_reason = BLHIP_FALLBACK_BUSY;
    ctx->timing.resident_fallback_reason = BLHIP_FALLBACK_BUSY;
    if (ctx->resident_giveups > 1) ctx->resident_retry_after = std::min(1024, 2 * ctx->resident_retry_after);
}

// ---- the time-resident path of a single-chain batch: one launch per pass instead of one per step (blhip_resident.hpp) -----------------
struct ResidentRun {
    bool on = false;
    ResidentPlan rp;
    blr::ResParams RQ{};
    int nblk = 0;                                  // partial-sum slots per (step, sum): one per tile
    std::vector<double> rowsumF;                   // forward pass: the actual sums of the stored rows
    std::vector<double> sfwd;                      // the forward pass's scales s_k (backward: predicted posterior sums)
    double *d_sfwd = nullptr;
    unsigned *d_abort = nullptr;
    size_t flag_bytes = 0;

    // eligibility (one chain, Gaussian model with the likelihood recurrence, the same radius <= 8 kernels at every step) + buffers
    void setup(const BatchEnv &E, int64_t n_chains, bool fast, bool use_rec, size_t &psz) {
        blhip_ctx *ctx = E.ctx;
        const ChainProgram &prog = *E.prog;
        const TapTable &taps = *E.taps;
        const int64_t T = E.T;
        const bool full = E.ff.full;
        double w0[blr::R + 1] = {1.0}, w1[blr::R + 1] = {1.0};
        // (a caller-supplied backward message: the predicted posterior sums assume the uniform one)
        // (the likelihood: the Gaussian recurrence, or -- every other model -- the (T, G) table: blr::Res TAB, tiles of up to 64 x 64)
        const bool gauss = E.p->obs_model == BLHIP_OM_GAUSSIAN && use_rec && E.d <= blr::DMAX;
        const bool tab = E.p->obs_model == BLHIP_OM_TABLE && E.DT->lik != nullptr && ctx->option("resident_table", 1.0) != 0.0;
        if (fast && n_chains == 1 && (gauss || tab) && !E.ff.resume && !E.ff.carry && !E.p->backward_init &&
            prog.LW0 <= blr::R && prog.LW1 <= blr::R && ctx->option("resident", 1.0) != 0.0 && ctx->resident_ok &&
            plan_resident(E.g.n0, E.g.n1, std::min(ctx->num_cus, 256), rp)) {
            on = prog.kindF[0] == SRC_PRIOR && (!full || prog.kindB[T - 1] == SRC_UNIFORM);
            // (the padded 128 x 128 BACKWARD kernel spills 231 registers: 2000 x 1100, backward step 40 - 45 us against 26.8 us with one
            //  launch per step -- full fits of such grids keep the launch-per-step kernels, evidence-only / forward-only fits do not)
            if (rp.pad && rp.TR == 128 && full) on = false;
            if (tab && rp.TR == 128) on = false;
            const int k0 = T > 1 ? prog.tapF0[1] : -1, k1 = T > 1 ? prog.tapF1[1] : -1;
            for (int64_t t = 1; t < T && on; ++t)
                on = prog.kindF[t] == SRC_PREV && prog.tapF0[t] == k0 && prog.tapF1[t] == k1;
            for (int64_t t = 0; t < T - 1 && on && full; ++t)
                on = prog.kindB[t] == SRC_PREV && prog.tapB0[t] == k0 && prog.tapB1[t] == k1;
            if (on) {
                for (int k = 1; k <= blr::R; ++k) w0[k] = w1[k] = 0.0;
                if (k0 >= 0) for (int k = 0; k <= taps.lw[k0]; ++k) w0[k] = taps.w[taps.off[k0] + k];
                if (k1 >= 0) for (int k = 0; k <= taps.lw[k1]; ++k) w1[k] = taps.w[taps.off[k1] + k];
            }
        }
        if (!on) return;
        const size_t nt = (size_t)rp.ntiles;
        // the halo strips hold TAGGED elements (the value with a one-bit tag in its sign bit): the tags are what the consumers poll
        const size_t n_cols = 2 * nt * 2 * blr::R * rp.TR, n_rows = 2 * nt * 2 * blr::R * rp.TC;
        const size_t b_cols = carve_size(n_cols * 8), b_rows = carve_size(n_rows * 8);
        const size_t b_w = carve_size(2 * (blr::R + 1) * 8) + carve_size((size_t)T * 8);
        flag_bytes = b_cols + b_rows + carve_size(blr::NSLOT * nt * 4 * 8) + carve_size(64);
        ctx->resx.ensure(b_w + flag_bytes);
        char *rc = ctx->resx.as<char>();
        double *d_w = carve<double>(rc, 2 * (blr::R + 1));
        d_sfwd = carve<double>(rc, (size_t)T);
        RQ.cols = carve<double>(rc, n_cols);              // (everything polled is contiguous from here on: one memset per launch)
        RQ.rows = carve<double>(rc, n_rows);
        RQ.cols_bytes = (unsigned)(n_cols * 8); RQ.rows_bytes = (unsigned)(n_rows * 8);
        RQ.gran = carve<unsigned long long>(rc, blr::NSLOT * nt * 4);
        d_abort = carve<unsigned>(rc, 16);
        RQ.abort_word = d_abort;
        double hw[2 * (blr::R + 1)];
        for (int k = 0; k <= blr::R; ++k) { hw[k] = w0[k]; hw[blr::R + 1 + k] = w1[k]; }
        HIPCHECK(hipMemcpyAsync(d_w, hw, sizeof hw, hipMemcpyHostToDevice, E.st));
        sync_stream(ctx, E.st);
        RQ.w0 = d_w; RQ.w1 = d_w + blr::R + 1;
        RQ.n0 = E.g.n0; RQ.n1 = E.g.n1; RQ.tr = rp.tr; RQ.tc = rp.tc; RQ.ntiles = rp.ntiles; RQ.T = (int)T; RQ.d = tab ? 0 : E.d; RQ.rec_len = E.rec_len;
        RQ.lik = tab ? E.DT->lik : nullptr;
        RQ.lag = std::max(1, std::min(blr::MAXLAG, (int)ctx->option("resident_lag", 2.0)));
        RQ.m0 = E.DT->m0; RQ.m1 = E.DT->m1; RQ.colA = E.DT->colA; RQ.colB = E.DT->colB; RQ.rec = E.DT->rec; RQ.step0 = E.step0;
        RQ.timeout_ticks = (unsigned long long)(resident_timeout_s(ctx, T) * 1e8);      // wall_clock64: 100 MHz
        // partial-sum slots per step: one per tile (the launch-per-step kernels are the fall-back and keep theirs)
        nblk = rp.ntiles;
        psz = std::max(psz, (size_t)T * NRED * nblk);
    }

    void launch(const BatchEnv &E, bool bwd, double *psum) {
        blhip_ctx *ctx = E.ctx;
        hipStream_t st = E.st;
        const int64_t T = E.T;
        blr::ResParams Q = RQ;
        HIPCHECK(hipMemsetAsync(RQ.cols, 0, flag_bytes, st));          // strip tags, granules, abort word: zero before EVERY launch
        HIPCHECK(hipMemsetAsync(psum, 0, (size_t)T * NRED * nblk * 8, st));
        Q.psum = psum;
        // rows of the kept sequence are normalised inside the kernel, `lag` steps behind (what the host then still scales: the
        // last / first `lag` rows)
        if (bwd) {
            // the posteriors are stored normalised: their sums follow from the forward scales and the last forward row sum
            // (blhip_resident.hpp: predicted_sum)
            sfwd.assign(T, 1.0);
            for (int64_t t = RQ.lag; t < T; ++t) sfwd[t] = 1.0 / rowsumF[t - RQ.lag];
            HIPCHECK(hipMemcpyAsync(d_sfwd, sfwd.data(), (size_t)T * 8, hipMemcpyHostToDevice, st));
            Q.sfwd = d_sfwd; Q.n_first = rowsumF[T - 1] * (1.0 / (double)E.G);
            Q.src0 = E.DT->uniform; Q.post = E.d_post; Q.store = 1; Q.means = 1; Q.normalise = 0;
        } else {
            Q.src0 = E.DT->prior; Q.post = E.ff.evidence_only ? nullptr : E.d_post; Q.store = E.ff.evidence_only ? 0 : 1;
            Q.means = E.ff.forward_only ? 1 : 0; Q.normalise = E.ff.forward_only ? 1 : 0;
        }
#ifdef BLR_PROF
        ctx->small.ensure(4 * 16 * 16 * 8);
        HIPCHECK(hipMemsetAsync(ctx->small.p, 0, 4 * 16 * 16 * 8, st));
        Q.prof = ctx->small.as<unsigned long long>();
#endif
        launch_resident(st, rp, Q, bwd);
        {   // HBM: the halo strips (2 R rows + 2 R columns of every tile, 8-byte tagged elements written and read once per step) + what the fit keeps
            const double halo = 2.0 * 8.0 * 2.0 * blr::R * (rp.TR + rp.TC) / ((double)rp.TR * rp.TC);
            const double kept = bwd ? 16.0 : (Q.store ? 8.0 : 0.0) + (Q.normalise ? 16.0 : 0.0);
            account(ctx, bwd, (double)E.G * T * (halo + kept), (double)E.G * T * (2.0 * valu_stencil_flop(blr::R) + (bwd ? EPI_BWD_FLOP : EPI_FWD_FLOP)));
        }
#ifdef BLR_PROF
        {   // development build: where a step of one interior tile spends its time (shader-clock cycles between stamps)
            unsigned long long hh[4 * 16 * 16];
            HIPCHECK(hipMemcpyAsync(hh, ctx->small.p, sizeof hh, hipMemcpyDeviceToHost, st));
            sync_stream(ctx, st);
            static const char *names[15] = {"start", "h_pre", "gath_issue", "B1", "h_walk", "book", "gather", "B2", "pubR", "v_pre", "B3", "v_walk", "sums", "B4", "pubC"};
            static const int tids[4] = {0, 128, 256, -64};
            for (int wv = 0; wv < 4; ++wv) {
                const unsigned long long *h = hh + wv * 256;
                double acc[15] = {0}; int n = 0;
                for (int q = 0; q < 16; ++q) {          // (stamps a kernel flavour does not set stay zero: the interval goes to the next one that is set)
                    if (!h[q * 16 + 14] || !h[q * 16]) continue;
                    ++n;
                    unsigned long long last = h[q * 16];
                    for (int i = 1; i < 15; ++i) if (h[q * 16 + i]) { acc[i] += (double)(h[q * 16 + i] - last); last = h[q * 16 + i]; }
                }
                std::fprintf(stderr, "[blr prof %s thread %d] %d steps:", bwd ? "bwd" : "fwd", tids[wv], n);
                double tot = 0; for (int i = 1; i < 15; ++i) { std::fprintf(stderr, " %s %.0f", names[i], n ? acc[i] / n : 0.0); tot += n ? acc[i] / n : 0.0; }
                std::fprintf(stderr, " | total %.0f\n", tot);
            }
        }
#endif
    }

    // after the forward pass: every tile made it, and the sums allow the lag to be undone
    bool forward_ok(const BatchEnv &E, double *redF) {
        if (resident_gave_up(E.ctx, E.st, d_abort)) return false;
        if (!resident_unlag(redF, E.T, RQ.lag, rowsumF)) { E.ctx->resident_last_reason = BLHIP_FALLBACK_RANGE; return false; }
        return true;
    }

    // after the backward pass: every tile made it, the lagged scale stayed in range, and the PREDICTED sums the kernel normalised
    // the stored posteriors by reproduce the reduced ones
    bool backward_ok(const BatchEnv &E, const double *redB) {
        const int64_t T = E.T;
        if (resident_gave_up(E.ctx, E.st, d_abort)) return false;
        for (int64_t t = 0; t < T; ++t) {
            const double Ct = redB[(size_t)t * NRED + 2], Nt = redB[(size_t)t * NRED];
            if (!(Ct > 1e-150 && Ct < 1e150) || !(Nt > 1e-250)) { E.ctx->resident_last_reason = BLHIP_FALLBACK_RANGE; return false; }
        }
        double npred = rowsumF[T - 1] * (1.0 / (double)E.G);
        for (int64_t t = T - 1; t >= 0; --t) {
            const int64_t k = T - 1 - t;
            if (k > 0) npred = (k >= RQ.lag ? 1.0 / redB[(size_t)(t + RQ.lag) * NRED + 2] : 1.0) * npred / sfwd[t + 1];
            const double Nt = redB[(size_t)t * NRED];
            if (!(std::fabs(npred - Nt) <= pred_rtol(T) * Nt)) { E.ctx->resident_last_reason = BLHIP_FALLBACK_PREDICTION; return false; }
        }
        return true;
    }
};

// ---- the chain-resident path of a batch: rounds of chains stay in LDS for a whole pass (blhip_chainres.hpp) ---------------------------
// The carried partial accumulators (blhip_ctx::PartState) -> the average posterior.  Nobody waits for the kernel here.
void flush_partials(blhip_ctx *ctx, hipStream_t st, std::vector<hipEvent_t> *later_ev) {
    blhip_ctx::PartState &ps = ctx->part;
    if (!ps.live) return;
    const double newref = std::max(ctx->acc_logref, ps.maxlw);
    const double r = ctx->acc_first ? 0.0 : std::exp(ctx->acc_logref - newref), rb = std::exp(ps.ref - newref);
    const long long G = (long long)ps.n0 * ps.n1;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (later_ev) {
        HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
        later_ev->push_back(e0); later_ev->push_back(e1);
        HIPCHECK(hipEventRecord(e0, st));
    }
    BL_LAUNCH(fold_parts_kernel, dim3((unsigned)(((G + 1) / 2 + NTHREADS - 1) / NTHREADS), (unsigned)ps.T), dim3(NTHREADS), 0, st, ctx->acc,
                       ctx->accpart.as<double>(), (long long)ps.T * ps.Gk, ps.slots_init, ps.n0, ps.n1, ps.T, r, rb,
                       ctx->acc_first ? 1 : 0, ps.n0p, ps.Gk, ps.ax1);
    HIPCHECK(hipGetLastError());
    if (later_ev) HIPCHECK(hipEventRecord(e1, st));
    ctx->timing.accumulate_launches += 1;
    ctx->acc_logref = newref; ctx->acc_first = false; ctx->acc_folded += ps.nfold;
    ps = blhip_ctx::PartState{};
}

struct ChainRun {
    bool on = false;
    ChainResPlan cp;
    blc::ChainParams CQ{};
    int *d_order = nullptr;
    unsigned char *d_ckF = nullptr, *d_ckB = nullptr;      // per (step, chain): what the chain kernels consume (ChainResPlan::ckF / ckB)
    size_t gran_bytes = 0;
    unsigned *d_abort = nullptr;
    std::vector<std::vector<double>> rowsumC;      // forward pass: the actual sums of the stored rows, per chain
    std::vector<std::vector<double>> sfwdC;        // forward pass: the scales it used, per chain
    // the batch's sequence buffer is read only by this fit's own backward pass / fold: free to use the kernel's strip-major layout
    bool post_private = false;
    bool tab = false;                              // the likelihood comes out of a table (blc::chain_kernel TAB)
    bool ax1 = false;                              // walks on both parameters: blc::chainax_kernel (the distribution is transposed between the filters)
    size_t xch_bytes = 0;
    int ax_mode = -1;                              // exchange mode (-1: per launch, see pass())
    // a padded grid whose sequence is NOT private (an ordinary fit that keeps its posteriors): the kernels work on a scratch sequence on
    // the padded geometry (blhip_ctx::postpad), depad_kernel writes the grid's rows into the sequence everybody else reads
    bool depad = false;
    // fused fold: the backward kernel adds the weighted, normalised posteriors to per-slot partial accumulators instead of storing
    // them (the separate fold re-read the whole sequence at the memory roof while the backward pass of the wide bands left
    // bandwidth unused: same bytes, one pass)
    bool fused = false;
    bool fold_done = false;                        // this batch's posteriors are in the accumulator already (or in the carried partial accumulators)
    bool part_fresh0 = true;                       // the first launch of the backward pass starts the slots (false: they carry earlier batches)
    bool touched_parts = false;                    // the backward pass of this batch has written the carried slots
    // fused fold with TWO chains per block (blc::chain_fold2_kernel): the backward pass runs rounds of 2 x cpr chains of its own
    bool fold2 = false;
    std::vector<int> round_start_b, round_nk_b;
    bool share_prefix = false;                     // change-point batches: states before a chain's first restart are stored once (see setup)
    bool skip_prefix = false;                      // ... and computed once: a chain's forward pass begins at its first restart
    int prov = 0;                                  // the chain that stores / computes everything
    long long shared_steps = 0;                    // chain-steps not stored
    int *d_tshare = nullptr;
    std::vector<int> h_tshare;
    const double *redF_keep = nullptr;             // the forward pass's reduced sums (slot 1: the restart sums of change-point batches)
    int slots_used = 0;                            // partial accumulators the backward launches write (one per block column)
    long long Gk = 0;                              // cells per distribution on the geometry the kernels work on (padded: >= G)
    double *d_fold_sfwd = nullptr, *d_fold_w = nullptr, *d_fold_inf = nullptr, *d_zeros = nullptr;
    std::vector<double> fold_lw;                   // log weight of every chain of the batch (-inf: none)
    double fold_ref = -INFINITY;                    // the reference the batch's weights are relative to (the carried slots' one where they are continued)
    double fold_max = -INFINITY;                    // the batch's largest log weight

    void setup(const BatchEnv &E, bool fast, bool use_rec, size_t &psz) {
        blhip_ctx *ctx = E.ctx;
        const ChainProgram &prog = *E.prog;
        const int64_t T = E.T, B = E.B;
        const long long G = E.G;
        // the likelihood: the Gaussian recurrence, or a table (every other model on a 2-D grid -- built on the device for the closed-form
        // ones, evaluated by the caller otherwise: blc::chain_kernel TAB, geometries of <= 512 rows -- padded grids too --, radius <= 40, one chain per block)
        const bool gauss = E.p->obs_model == BLHIP_OM_GAUSSIAN && use_rec && E.d <= blc::DMAX;
        tab = E.p->obs_model == BLHIP_OM_TABLE && E.DT->lik != nullptr && ctx->option("chain_table", 1.0) != 0.0;
        // walks on the second parameter too: the transposing kernels (blc::chainax_kernel; Gaussian recurrence, exact square geometries)
        const bool may_ax1 = (gauss || tab) && prog.LW1 > 0 && ctx->option("chain_ax1", 1.0) != 0.0;
        if (fast && (gauss || tab) && !E.ff.resume && !E.ff.carry && !E.p->backward_init &&
            !prog.has_clamp && (prog.LW1 == 0 || may_ax1) && (double)G * 8.0 < 4.0e9 && ctx->option("chain_resident", 1.0) != 0.0 && ctx->resident_ok &&
            E.allow_chainres) {
            cp.r0_max = (!tab && ctx->option("chain_wide", 1.0) != 0.0) ? CHAIN_R0_MAX : FAST_R0_MAX;
            cp.allow_ax1 = may_ax1;
            on = plan_chainres(E.g, prog, *E.taps, B, T, E.ff.full, std::min(ctx->num_cus, 256), cp);
            E.mark("  plan_chainres");
            if (on && tab && cp.ntw > 4) on = false;
            if (on && tab && cp.pad && cp.ntw >= 3 && !cp.ax1) on = false;      // (padded 384 / 512-row geometries with a likelihood table: no kernel -- never selected by a test or workload, pruned in round 5)
            if (on && prog.LW1 > 0 && !cp.ax1) on = false;
        }
        if (!on) return;
        ax1 = cp.ax1;
        Gk = (long long)cp.n0p * cp.n1p;
        // (granule slots for 2 x cpr chains: the two-chain fold kernel runs rounds of that size)
        gran_bytes = carve_size((size_t)blc::NSLOT * 2 * cp.cpr * cp.strips * 2 * 8);
        ctx->resx.ensure(carve_size((size_t)B * 4) * 4 + gran_bytes + carve_size(64) + 2 * carve_size((size_t)T * B));
        char *rc = ctx->resx.as<char>();
        d_order = carve<int>(rc, (size_t)B);
        int *d_tapid = carve<int>(rc, (size_t)B);
        int *d_tapid1 = carve<int>(rc, (size_t)B);
        d_tshare = carve<int>(rc, (size_t)B);
        CQ.gran = carve<unsigned long long>(rc, (size_t)blc::NSLOT * 2 * cp.cpr * cp.strips * 2);
        d_abort = carve<unsigned>(rc, 16);
        CQ.abort_word = d_abort;
        d_ckF = carve<unsigned char>(rc, (size_t)T * B);
        d_ckB = carve<unsigned char>(rc, (size_t)T * B);
        if (cp.has_reset) {
            HIPCHECK(hipMemcpyAsync(d_ckF, cp.ckF.data(), (size_t)T * B, hipMemcpyHostToDevice, E.st));
            HIPCHECK(hipMemcpyAsync(d_ckB, cp.ckB.data(), (size_t)T * B, hipMemcpyHostToDevice, E.st));
        }
        HIPCHECK(hipMemcpyAsync(d_order, cp.order.data(), (size_t)B * 4, hipMemcpyHostToDevice, E.st));
        HIPCHECK(hipMemcpyAsync(d_tapid, cp.tap_id.data(), (size_t)B * 4, hipMemcpyHostToDevice, E.st));
        HIPCHECK(hipMemcpyAsync(d_tapid1, cp.tap_id1.data(), (size_t)B * 4, hipMemcpyHostToDevice, E.st));
        sync_stream(ctx, E.st);
        E.mark("  chain metadata H2D");
        if (ax1) {               // the exchange buffers of a launch's chain slots: [slot][2 step parities][Gk] tagged elements
            xch_bytes = (size_t)cp.cpr * 2 * (size_t)Gk * 8;
            {   // the exchange buffers and the likelihood table of the even steps are not part of chains_per_batch's budget: a long series
                // on a large geometry declines the path here instead of failing in the allocator
                const bool want_tab = tab || B <= 2;
                const double need = (double)xch_bytes + (want_tab ? (double)((T + 1) / 2) * (double)Gk * 8.0 : 0.0);
                size_t free_b = 0, total_b = 0;
                (void)hipMemGetInfo(&free_b, &total_b);
                if (need > 0.8 * ((double)free_b + (double)ctx->xch.cap + (double)ctx->axlik.cap)) { on = false; ax1 = false; Gk = 0; return; }
            }
            ctx->xch.ensure(xch_bytes);
            CQ.xch = ctx->xch.as<double>(); CQ.xch_chain = 2 * Gk; CQ.tap_id1 = d_tapid1;
            ax_mode = -1;             // (development: 0 write-through stores on any XCD, 1 one XCD per chain, 2 plain stores, 3 both = the default)
            // the likelihood of the even time steps (epilogue in the transposed layout) as a table: an exp per cell there otherwise.  Measured
            // (profiles/r05_notes.md): single fits are latency-bound and gain (200 x 200, radius 20: 7.9 -> 7.5 us per step); launches of
            // eight chains sit on eight XCDs, every one of which fetches the table through the fabric the exchange already saturates
            // (c4_both_axes 553 -> 578 ms): one or two chains only
            const double tab_bytes = (double)((T + 1) / 2) * (double)Gk * 8.0;
            constexpr double tab_opt = 1.0;
            if (tab) {          // a tabulated model: every step reads its likelihood -- the even ones out of a transposed copy
                if (tab_bytes >= 8.0e9) { on = false; ax1 = false; Gk = 0; return; }
                ctx->axlik.ensure((size_t)tab_bytes);
                blcl::chainax_lik_transpose(E.st, E.DT->lik, ctx->axlik.as<double>(), cp.n0p, E.g.n0, E.g.n1, (int)T);
                HIPCHECK(hipGetLastError());
                CQ.lik = ctx->axlik.as<double>(); CQ.lik_nat = E.DT->lik;
            } else if (tab_opt != 0.0 && (B <= 2 || tab_opt == 2.0) && tab_bytes < 8.0e9) {
                ctx->axlik.ensure((size_t)tab_bytes);
                blcl::chainax_lik_table(E.st, cp.n0p, E.g.n0, E.g.n1, (int)T, E.d, E.rec_len, E.DT->m0, E.DT->colA, E.DT->colB, E.DT->rec, ctx->axlik.as<double>());
                HIPCHECK(hipGetLastError());
                CQ.lik = ctx->axlik.as<double>();
            }
        }
        CQ.n0 = cp.n0p; CQ.n1 = cp.n1p; CQ.n0t = E.g.n0; CQ.n1t = E.g.n1; CQ.strips = cp.strips; CQ.T = (int)T; CQ.d = tab ? 0 : E.d; CQ.rec_len = E.rec_len;
        CQ.lag = std::max(2, std::min(blc::MAXLAG, (int)ctx->option("chain_resident_lag", 4.0)));      // (lag 1 would need the sum of the step in flight)
        CQ.B = (int)B; CQ.nblk = cp.strips; CQ.tap_id = d_tapid; CQ.taps = E.M->taps; CQ.tap_off = E.M->off; CQ.tap_lw = E.M->lw;
        CQ.post_stride = (long long)T * Gk;
        CQ.m0 = E.DT->m0; CQ.m1 = E.DT->m1; CQ.colA = E.DT->colA; CQ.colB = E.DT->colB; CQ.rec = E.DT->rec; if (tab && !ax1) CQ.lik = E.DT->lik; CQ.step0 = E.step0;      // (both-axes kernels: CQ.lik is their table of the even steps, set above)
        CQ.timeout_ticks = (unsigned long long)(resident_timeout_s(ctx, T) * 1e8);
        psz = std::max(psz, (size_t)T * B * NRED * cp.strips);
        if (!tab && !ax1 && (prog.LW0 > 0 || E.ff.full)) {          // (no-stencil batches: only their folding backward pass reads it)
            // the anchors of the likelihood recurrence: the same for every chain of the batch (same data, same grid) -- tabulated once, 32 bytes
            // per lane and step (C4: 134 MB), read by the kernels a step ahead (ChainParams::anch)
            const size_t ab = (size_t)T * blc::NW * cp.strips * 64 * 4 * sizeof(double);
            size_t free_b = 0, total_b = 0;
            (void)hipMemGetInfo(&free_b, &total_b);
            if ((double)ab > 0.5 * ((double)free_b + (double)ctx->anchbuf.cap)) { on = false; return; }
            ctx->anchbuf.ensure(ab);
            blc::AnchorParams AP{};
            AP.T = (int)T; AP.d = E.d; AP.rec_len = E.rec_len; AP.strips = cp.strips; AP.rows_per_wave = cp.ntw * blc::TM; AP.n0t = E.g.n0; AP.n1t = E.g.n1;
            AP.m0 = E.DT->m0; AP.colA = E.DT->colA; AP.colB = E.DT->colB; AP.rec = E.DT->rec; AP.out = ctx->anchbuf.as<double>();
            BL_LAUNCH(blc::anchor_table_kernel, dim3((unsigned)cp.strips, (unsigned)T), dim3(blc::NT), 0, E.st, AP);
            HIPCHECK(hipGetLastError());
            CQ.anch = ctx->anchbuf.as<double>();
            E.mark("  anchor table queued");
        }
        // (the separate fold reads pairs of cells: an even number of columns; the fused fold's partials take any grid -- fold_parts_kernel)
        post_private = E.ff.accumulate && E.ff.full && !E.ff.keep && !E.ff.carry && ((G & 1) == 0 || cp.pad || ax1) && ((uintptr_t)ctx->acc & 15) == 0;
        // (not beside overlapped folds: a FoldJob of the previous batch may still read ctx->accw and read-modify-write ctx->acc on the
        //  second stream while this batch re-carves accw and its fused fold writes ctx->acc on the main stream)
        // (a restart inside a FILTERING chain would need sum(alpha T(reset)) for the predicted sums: such batches store and fold separately)
        fused = post_private && !cp.mixed && !E.overlap_acc && ctx->option("fuse_accumulate", 1.0) != 0.0;
        if (fused && cp.has_reset) {
            // change-point batches: the predicted sums survive a restart only through the two-chain fold kernel's restart rule, and
            // only if the two passes restart at the same places (backward step t restarts <=> forward step t + 1 does: unit-spaced
            // time stamps, transitionModels.py:316-317)
            // (the both-axes kernels' fold has no restart rule: such batches store and fold separately -- ADVICE r05)
            bool aligned = !tab && !ax1 && !E.chain_means && fold2_shape(cp.ntw) && ctx->option("fold2", 1.0) != 0.0 && ctx->option("fold2_cp", 1.0) != 0.0;
            for (int64_t b = 0; b < B && aligned; ++b)
                for (int64_t t = 0; t + 1 < T && aligned; ++t)
                    aligned = (prog.kindB[(size_t)t * B + b] != SRC_PREV) == (prog.kindF[(size_t)(t + 1) * B + b] != SRC_PREV);
            fused = aligned;
        }
        slots_used = (int)std::min<int64_t>(cp.cpr, B);
        if (fused && !tab && !ax1 && !E.chain_means && fold2_shape(cp.ntw) && ctx->option("fold2", 1.0) != 0.0) {      // (the two-chain kernel has no table / both-axes flavour)
            fold2 = true;
            const int per = 2 * cp.cpr;
            for (int64_t s0 = 0; s0 < B; s0 += per) {
                const int64_t s1 = std::min<int64_t>(B, s0 + per);
                const int tp = cp.tap_id[cp.order[s1 - 1]];                   // (sorted by radius: the last chain of the round is its widest)
                const int r0 = std::max(4, ((tp >= 0 ? E.taps->lw[tp] : 0) + 3) / 4 * 4);
                round_start_b.push_back((int)s0);
                round_nk_b.push_back(prog.LW0 == 0 ? 4 : (blc::TM + 2 * r0) / 4);          // (4: the no-stencil variant)
            }
            round_start_b.push_back((int)B);
            slots_used = (int)std::min<int64_t>(cp.cpr, (B + 1) / 2);
        }
        // a grid smaller than the geometry: only fits whose sequences are private to the fit (strip-major, padded) -- evidence-only fits
        // and full fits of hyper- / change-point studies (folded in the backward kernel, or stored and folded by accumulate_pad_kernel);
        // everything else keeps the launch-per-step kernels
        // (the both-axes kernels keep their sequences in two alternating strip-major layouts: never the API's, also on an exact geometry)
        if ((cp.pad || ax1) && !(E.ff.evidence_only || post_private)) {
            // ... or, at the price of a second sequence in memory, any fit: the posteriors are copied out of the padded layout afterwards
            // (<= 512 rows: every flavour has a padded storing kernel; 1024 rows: forward passes only)
            size_t free_b = 0, total_b = 0;
            (void)hipMemGetInfo(&free_b, &total_b);
            const double need = (double)B * (double)T * (double)Gk * 8.0;
            depad = ctx->option("chain_depad", 1.0) != 0.0 && !E.overlap_acc && (cp.ntw <= 4 || !E.ff.full) &&
                    need < 0.8 * ((double)free_b + (double)ctx->postpad.cap);
            if (!depad) { on = false; fused = false; fold2 = false; return; }
            fused = false; fold2 = false;
        }
        // (<= 512 rows: what the two-chain kernel does not fold -- per-chain means asked for through the C-ABI, option fold2 = 0 -- is stored and
        //  folded separately: the Gaussian one-chain kernel has no folding flavour there (round 6).  The tabulated one has, on exact
        //  geometries; 1024 rows: the one-chain kernel is the only folding one, exact and padded; the both-axes kernels fold on padded grids too)
        if (fused && !fold2 && cp.ntw <= 4 && !ax1 && (cp.pad || !tab)) fused = false;
        if (cp.pad && cp.ntw > 4 && E.ff.full && !fused) { on = false; fold2 = false; return; }
        // Change-point batches without a stencil whose backward pass folds: the chains are identical up to their first restart.  The chain
        // with the LATEST first restart stores all its states; every other chain stores only from its own first restart on, and the
        // folding backward pass reads the earlier ones from that chain (same values bit for bit: same kernel, same strips, same lagged
        // scales).  Halves the forward pass's stores, and the backward pass's reads of those states hit the cache (the chains of a launch
        // read the same rows at about the same time).  (Only when nothing overwrites the stored states: the fused fold.)
        // The same chains need not COMPUTE that prefix either (a restart consumes the reset distribution, not the chain's past): with
        // skip_prefix a chain's forward pass begins at its first restart, the sums of its earlier steps are copied from the providing chain
        // (forward_ok).  Also for evidence-only fits, which store nothing.  C5: 250 chains x 1000 steps -> 125 k chain-steps of 250 k.
        E.mark("  fold plan");
        share_prefix = false; skip_prefix = false;
        const bool may_share = fused && ctx->option("share_prefix", 1.0) != 0.0;
        const bool may_skip = (fused || E.ff.evidence_only) && ctx->option("skip_prefix", 1.0) != 0.0;
        if ((may_share || may_skip) && cp.has_reset && prog.LW0 == 0 && !cp.mixed && !ax1 && B >= 2) {      // (blc::chainax_kernel reads neither tshare nor skip_prefix)
            std::vector<int> tfirst((size_t)B, (int)T);
            bool plain = true;
            for (int64_t b = 0; b < B && plain; ++b) {
                plain = prog.kindF[b] == SRC_PRIOR && cp.tap_id[b] < 0 && cp.tap_id1[b] < 0;
                for (int64_t t = 1; t < T; ++t)
                    if (prog.kindF[(size_t)t * B + b] != SRC_PREV) { tfirst[b] = (int)t; break; }
            }
            if (plain) {
                prov = 0;
                for (int64_t b = 1; b < B; ++b) if (tfirst[b] > tfirst[prov]) prov = (int)b;
                std::vector<int> tsh((size_t)B, 0);
                long long saved = 0;
                for (int64_t b = 0; b < B; ++b)
                    if (b != prov) { tsh[b] = std::min(tfirst[b], tfirst[prov]); saved += tsh[b]; }
                if (saved > 0) {
                    HIPCHECK(hipMemcpyAsync(d_tshare, tsh.data(), (size_t)B * 4, hipMemcpyHostToDevice, E.st));
                    sync_stream(ctx, E.st);
                    share_prefix = may_share || may_skip; skip_prefix = may_skip; shared_steps = saved; h_tshare = tsh;
                    CQ.tshare = d_tshare; CQ.bprov = prov; CQ.skip_prefix = skip_prefix ? 1 : 0;
                }
            }
        }
        E.mark("  prefix plan");
        if (fused) {
            // (carried slots of earlier batches: a larger buffer or another layout takes them into the accumulator first)
            if (ctx->part.live && ((size_t)slots_used * T * Gk * 8 > ctx->accpart.cap || ctx->part.T != (int)T || ctx->part.Gk != Gk || ctx->part.n0 != E.g.n0 ||
                                   ctx->part.n1 != E.g.n1 || ctx->part.n0p != cp.n0p || ctx->part.ax1 != (ax1 ? 1 : 0)))
                flush_partials(ctx, E.st, E.fold_ev);
            ctx->accpart.ensure((size_t)slots_used * T * Gk * 8);
            ctx->accw.ensure(carve_size((size_t)T * B * 8) + 2 * carve_size((size_t)B * 8) + carve_size(8192));
            char *wc = ctx->accw.as<char>();
            d_fold_sfwd = carve<double>(wc, (size_t)T * B);
            d_fold_w = carve<double>(wc, (size_t)B);
            d_fold_inf = carve<double>(wc, (size_t)B);
            d_zeros = carve<double>(wc, 1024);      // (8 KB: a tile's four cells are read at base + 0 / 512 / 1024 / 1536 bytes, base < 4 KB)
        }
    }

    // one pass: the launches of the rounds follow each other on the stream
    void pass(const BatchEnv &E, bool bwd, double *psum) {
        blhip_ctx *ctx = E.ctx;
        hipStream_t st = E.st;
        const int64_t T = E.T, B = E.B;
        HIPCHECK(hipMemsetAsync(psum, 0, (size_t)T * B * NRED * cp.strips * 8, st));
        HIPCHECK(hipMemsetAsync(d_abort, 0, 64, st));
        const bool two = bwd && fused && fold2;
        const std::vector<int> &rstart = two ? round_start_b : cp.round_start, &rnk = two ? round_nk_b : cp.round_nk;
        for (size_t r = 0; r + 1 < rstart.size(); ++r) {
            blc::ChainParams Q = CQ;
            HIPCHECK(hipMemsetAsync(CQ.gran, 0, gran_bytes, st));       // tags restart with every launch
            Q.chain_ids = d_order + rstart[r];
            Q.nslots = rstart[r + 1] - rstart[r];
            Q.psum = psum;
            Q.src0 = bwd ? E.DT->uniform : E.DT->prior;
            Q.kinds = cp.has_reset ? (bwd ? d_ckB : d_ckF) : nullptr;
            Q.reset = E.DT->reset;
            Q.post = E.d_post;
            Q.means = bwd ? (E.chain_means ? 1 : 0) : (E.ff.forward_only ? 1 : 0);
            Q.strip_major = (post_private || cp.pad || ax1) ? 1 : 0;            // (the stored sequence is private to the fit then)
            const bool fold_now = bwd && fused;
            if (fold_now) {
                Q.sfwd = d_fold_sfwd; Q.wchain = d_fold_w; Q.infirst = d_fold_inf;
                Q.part = ctx->accpart.as<double>(); Q.part_stride = (long long)T * Gk;
                Q.zeros = d_zeros; Q.part_fresh = (r == 0 && part_fresh0) ? 1 : 0;          // (every slot is first used by the first launch: no memset)
                touched_parts = true;
            }
#ifdef BLC_PROF
            ctx->small.ensure(2 * 16 * 16 * 8);
            HIPCHECK(hipMemsetAsync(ctx->small.p, 0, 2 * 16 * 16 * 8, st));
            Q.prof = ctx->small.as<unsigned long long>();
#endif
            if (ax1) {
                HIPCHECK(hipMemsetAsync(CQ.xch, 0, xch_bytes, st));           // (tags restart with every launch)
                // the blocks of a chain on ONE XCD (block b runs on XCD b % 8 -- observed, for speed only) and plain publishing stores, which
                // keep the lines in that XCD's L2 (write-through stores drop them: MI355X_MICROARCH.md); chain_ax1_mode = 0: write-through
                // stores and the dispatch order of the other chain kernels -- nothing then depends on the placement, not even the speed
                // (the placement is checked by the co-residency probe: where block b was NOT seen on XCD b % 8 -- another partition mode, a CU
                //  mask -- the plain stores' lines would never reach a consumer on another XCD: write-through stores in dispatch order then)
                Q.xch_mode = ax_mode >= 0 ? ax_mode : (ctx->xcd_order_ok ? 3 : 0);
                const bool prof = ctx->option("chain_prof", 0.0) != 0.0;
                if (prof) {
                    ctx->small.ensure(2 * 16 * 16 * 8);
                    HIPCHECK(hipMemsetAsync(ctx->small.p, 0, 2 * 16 * 16 * 8, st));
                    Q.prof = ctx->small.as<unsigned long long>();
                }
                launch_chainax(st, Q, rnk[r], cp.ntw, bwd, fold_now ? false : (bwd || !E.ff.evidence_only), cp.pad);
                if (prof) {          // where a step of strip 0 spends its time (shader-clock cycles between stamps; waves 0 and 2)
                    unsigned long long hh[2 * 16 * 16];
                    HIPCHECK(hipMemcpyAsync(hh, ctx->small.p, sizeof hh, hipMemcpyDeviceToHost, st));
                    sync_stream(ctx, st);
                    static const char *names[6] = {"start", "filter 1 + publish", "gather", "barrier", "filter 2 + epilogue", "sums + barrier"};
                    for (int wvi = 0; wvi < 2; ++wvi) {
                        const unsigned long long *h = hh + wvi * 256;
                        double acc[6] = {0}; int n = 0;
                        for (int q = 0; q < 16; ++q) { if (!h[q * 16 + 5] || !h[q * 16]) continue; ++n; for (int i = 1; i < 6; ++i) acc[i] += (double)(h[q * 16 + i] - h[q * 16 + i - 1]); }
                        std::fprintf(stderr, "[blc ax prof %s NK %d wave %d] %d steps:", bwd ? "bwd" : "fwd", rnk[r], wvi ? 2 : 0, n);
                        double tot = 0; for (int i = 1; i < 6; ++i) { std::fprintf(stderr, " %s %.0f", names[i], n ? acc[i] / n : 0.0); tot += n ? acc[i] / n : 0.0; }
                        std::fprintf(stderr, " | total %.0f\n", tot);
                    }
                }
            }
            else if (two) launch_fold2(st, Q, rnk[r], cp.ntw, cp.pad);
            else launch_chain(st, Q, rnk[r], cp.ntw, bwd, fold_now ? false : (bwd || !E.ff.evidence_only), cp.pad);
            {   // HBM: only what the fit keeps -- forward the stored state (8 B; nothing for evidence-only fits), backward the stored
                // state in + the posterior out (16 B) or + the read-modify-write of the partial accumulator (24 B; shared by the two
                // chains of a block of the two-chain fold kernel: 8 + 16 / 2 = 16 B)
                double cells = (double)Q.nslots * Gk * T;
                if (!bwd && skip_prefix)               // (chain-steps the forward pass does not run)
                    for (int q = rstart[r]; q < rstart[r + 1]; ++q) cells -= (double)h_tshare[cp.order[q]] * Gk;
                double bytes = bwd ? (fold_now ? (two ? 8.0 + 16.0 * ((Q.nslots + 1) / 2) / (double)Q.nslots : 24.0) : 16.0) : (E.ff.evidence_only ? 0.0 : 8.0);
                // (both-axes kernels: the transposing exchange of a step leaves the XCD's L2 as well -- 8 written + 8 read;
                //  with the blocks of a chain on one XCD the reads mostly hit that L2: the PMC counters say what really moves)
                if (ax1) bytes += 16.0;
                const int r0 = (4 * rnk[r] - blc::TM) / 2;
                double shared = 0.0;                                  // stored states not written (forward) / read once per launch instead of once per chain (backward)
                if (share_prefix && (bwd ? fold_now : !E.ff.evidence_only)) {
                    long long sum = 0, mx = 0;
                    for (int q = rstart[r]; q < rstart[r + 1]; ++q) { const long long v = h_tshare[cp.order[q]]; sum += v; mx = std::max(mx, v); }
                    shared = (double)(bwd ? sum - mx : sum) * Gk * 8.0;
                }
                // (tabulated likelihood: the table of the pass is read once per launch -- its chains read the same rows at about the same time)
                const double table = tab ? (double)T * Gk * 8.0 : 0.0;
                account(ctx, bwd, cells * bytes - shared + table, cells * ((rnk[r] > 4 ? band_stencil_flop(r0) * (ax1 ? 2.0 : 1.0) : 0.0) + (bwd ? EPI_BWD_FLOP : EPI_FWD_FLOP)));
            }
#ifdef BLC_PROF
            {   // development build: where a step of strip 0 spends its time (shader-clock cycles between stamps; waves 0 and 2)
                unsigned long long hh[2 * 16 * 16];
                HIPCHECK(hipMemcpyAsync(hh, ctx->small.p, sizeof hh, hipMemcpyDeviceToHost, st));
                sync_stream(ctx, st);
                static const char *names[8] = {"start", "ring", "chain0", "scale+anchor", "epi0", "tiles1..", "sums", "barrier"};
                for (int wvi = 0; wvi < 2; ++wvi) {
                    const unsigned long long *h = hh + wvi * 256;
                    double acc[8] = {0}; int n = 0;
                    for (int q = 0; q < 16; ++q) { if (!h[q * 16 + 7] || !h[q * 16]) continue; ++n; for (int i = 1; i < 8; ++i) acc[i] += (double)(h[q * 16 + i] - h[q * 16 + i - 1]); }
                    std::fprintf(stderr, "[blc prof %s NK %d wave %d] %d steps:", bwd ? "bwd" : "fwd", rnk[r], wvi ? 2 : 0, n);
                    double tot = 0; for (int i = 1; i < 8; ++i) { std::fprintf(stderr, " %s %.0f", names[i], n ? acc[i] / n : 0.0); tot += n ? acc[i] / n : 0.0; }
                    std::fprintf(stderr, " | total %.0f\n", tot);
                }
            }
#endif
        }
    }

    // after the forward pass: every strip made it, and the sums of every chain allow the scales to be undone
    bool forward_ok(const BatchEnv &E, double *redF) {
        if (resident_gave_up(E.ctx, E.st, d_abort)) return false;
        redF_keep = redF;
        if (skip_prefix)               // the steps a chain did not compute: the providing chain's sums (raw: before anybody's scales are undone)
            for (int64_t b = 0; b < E.B; ++b)
                for (int64_t t = 0; t < h_tshare[b]; ++t)
                    std::memcpy(&redF[((size_t)t * E.B + b) * NRED], &redF[((size_t)t * E.B + prov) * NRED], NRED * sizeof(double));
        if (chain_unlag_batch(redF, E.T, CQ.lag, E.B, rowsumC, sfwdC, cp.has_reset ? E.prog->kindF.data() : nullptr,
                              skip_prefix ? h_tshare.data() : nullptr) >= 0) { E.ctx->resident_last_reason = BLHIP_FALLBACK_RANGE; return false; }
        return true;
    }

    // fused fold, before the backward pass: weights relative to the batch's own reference (core.py:1358-1366: chains without a finite
    // evidence do not count), the forward scales and the sum of the last step's posterior of every chain -> device; partials zeroed
    void prepare_fold(const BatchEnv &E, const BatchOutcome &O) {
        blhip_ctx *ctx = E.ctx;
        const int64_t T = E.T, B = E.B;
        fold_lw.assign(B, -INFINITY);
        fold_ref = -INFINITY;
        for (int64_t b = 0; b < B; ++b) {
            if (O.abort_step[b] >= 0 || !std::isfinite(O.logE[b]) || !std::isfinite(E.log_w[E.c0 + b])) continue;
            fold_lw[b] = O.logE[b] + E.log_w[E.c0 + b];
            fold_ref = std::max(fold_ref, fold_lw[b]);
        }
        // the slots carry earlier batches of this call: this batch's weights take THEIR reference (a batch whose evidence towers 500
        // e-folds above it would overflow them: the carried slots go into the accumulator first and the batch starts afresh)
        fold_max = fold_ref;
        part_fresh0 = true;
        if (ctx->part.live && ctx->option("carry_partials", 1.0) != 0.0) {
            if (!(fold_ref - ctx->part.ref <= 500.0)) flush_partials(ctx, E.st, E.fold_ev);
            else { fold_ref = ctx->part.ref; part_fresh0 = false; }
        } else if (ctx->part.live) flush_partials(ctx, E.st, E.fold_ev);
        ctx->pinA.ensure(((size_t)T * B + 2 * (size_t)B) * 8);
        double *h = ctx->pinA.as<double>(), *hw = h + (size_t)T * B, *hi = hw + B;
        for (int64_t b = 0; b < B; ++b) {
            std::memcpy(h + (size_t)b * T, sfwdC[b].data(), (size_t)T * 8);
            // a restart of the backward pass at step t: the kernel divides by sum(alpha_t reset), which the forward pass left in slot 1
            // of step t -- handed over where the forward scale of step t + 1 would be (the identity does not use it there)
            if (cp.has_reset)
                for (int64_t t = 0; t + 1 < T; ++t)
                    if (E.prog->kindB[(size_t)t * B + b] != SRC_PREV) h[(size_t)b * T + t + 1] = redF_keep[((size_t)t * B + b) * NRED + 1];
            hw[b] = std::isfinite(fold_lw[b]) ? std::exp(fold_lw[b] - fold_ref) : 0.0;
            hi[b] = 1.0 / (rowsumC[b][T - 1] * (1.0 / (double)E.G));
        }
        HIPCHECK(hipMemcpyAsync(d_fold_sfwd, h, (size_t)T * B * 8, hipMemcpyHostToDevice, E.st));
        HIPCHECK(hipMemcpyAsync(d_fold_w, hw, (size_t)B * 8, hipMemcpyHostToDevice, E.st));
        HIPCHECK(hipMemcpyAsync(d_fold_inf, hi, (size_t)B * 8, hipMemcpyHostToDevice, E.st));
        // the slots need no memset: the first launch of the pass reads zeros instead of them (part_fresh) -- except slots it does not
        // use (a first launch with fewer chains than slots), which later launches may
        const int first_n = fold2 ? (round_start_b[1] - round_start_b[0] + 1) / 2 : cp.round_start[1] - cp.round_start[0];
        // (carried slots: only those no earlier batch has written)
        const int have = part_fresh0 ? first_n : ctx->part.slots_init;
        if (have < slots_used)
            HIPCHECK(hipMemsetAsync(ctx->accpart.as<double>() + (size_t)have * T * Gk, 0, (size_t)(slots_used - have) * T * Gk * 8, E.st));
        HIPCHECK(hipMemsetAsync(d_zeros, 0, 8192, E.st));
    }

    // after the backward pass: every strip made it and the lagged scale of the backward state stayed in range
    bool backward_ok(const BatchEnv &E, const double *redB) {
        if (resident_gave_up(E.ctx, E.st, d_abort)) return false;
        const size_t n = (size_t)E.T * E.B;             // (every record of the pass, in memory order)
        for (size_t q = 0; q < n; ++q) {
            const double *r = &redB[q * NRED];
            if (!(r[2] > 1e-150 && r[2] < 1e150) || !(r[0] > 1e-250)) { E.ctx->resident_last_reason = BLHIP_FALLBACK_RANGE; return false; }
        }
        return true;
    }

    // fused fold, after the backward pass: the kernel normalised every posterior by its PREDICTED sum -- the prediction must
    // reproduce the reduced sums (false: the caller repeats the batch with the launch-per-step kernels; the partials are dropped);
    // then the partial accumulators go into the average posterior (running reference exponent as in prepare_fold)
    // (later_ev: the fold kernel's start / end events -- nobody waits for it here: the host's bookkeeping of this batch and the next batch's
    //  setup run beside it, do_fit reads the events after its last batch)
    bool fold(const BatchEnv &E, const double *redB, std::vector<hipEvent_t> &later_ev) {
        blhip_ctx *ctx = E.ctx;
        // (tests of the repeat after poisoned carried slots: pretend that this batch's prediction check failed)
        if ((int64_t)ctx->option("fold_force_fail_batch", -1.0) == E.bi) { ctx->resident_last_reason = BLHIP_FALLBACK_PREDICTION; return false; }
        hipStream_t st = E.st;
        const int64_t T = E.T, B = E.B;
        const long long G = E.G;
        {
            // the backward scales, in processing order k = T - 1 - t (the kernel's rule, from the sums C of its new states), and the predicted
            // sums they lead to -- step by step over all chains (the records of a step are consecutive; chain by chain every read was a
            // cache line of its own: 0.4 ms per batch of 256 x 256 beside an idle GPU); per chain the operations and their order are unchanged
            const int lag = CQ.lag;
            std::vector<double> csum((size_t)T * B), sb((size_t)T * B, 1.0), npred((size_t)B);      // [k][b]
            const double rtol = pred_rtol(T);
            for (int64_t k = 0; k < T; ++k) {
                const int64_t t = T - 1 - k;
                const double *rt = redB + (size_t)t * B * NRED;
                double *ck = &csum[(size_t)k * B], *sk = &sb[(size_t)k * B];
                for (int64_t b = 0; b < B; ++b) {
                    ck[b] = rt[b * NRED + 2];
                    if (k >= lag) sk[b] = (k - lag - 1 >= 0 ? csum[(size_t)(k - lag - 1) * B + b] : 1.0) * sb[(size_t)(k - lag) * B + b] / csum[(size_t)(k - lag) * B + b];
                    const bool restart = cp.has_reset && E.prog->kindB[(size_t)t * B + b] != SRC_PREV && k > 0;
                    if (k == 0) npred[b] = rowsumC[b][T - 1] * (1.0 / (double)G);
                    else if (restart) npred[b] = sk[b] * redF_keep[((size_t)t * B + b) * NRED + 1];
                    else npred[b] = sk[b] * npred[b] / sfwdC[b][t + 1];
                    const double Nt = rt[b * NRED];
                    if (!(std::fabs(npred[b] - Nt) <= rtol * Nt)) { E.ctx->resident_last_reason = BLHIP_FALLBACK_PREDICTION; return false; }
                }
            }
        }
        if (std::isfinite(fold_ref)) {
            // the slots stay where they are: the next batch of the call adds to them, do_fit folds them after the last one (flush_partials)
            blhip_ctx::PartState &ps = ctx->part;
            int nfold = 0;
            for (int64_t b = 0; b < B; ++b) nfold += std::isfinite(fold_lw[b]) ? 1 : 0;
            if (!ps.live) { ps = blhip_ctx::PartState{}; ps.live = true; ps.ref = fold_ref; ps.first_batch = E.bi; }
            ps.maxlw = std::max(ps.maxlw, fold_max);
            ps.slots_init = std::max(ps.slots_init, slots_used);
            ps.nfold += nfold;
            ps.n0 = E.g.n0; ps.n1 = E.g.n1; ps.T = (int)T; ps.n0p = cp.n0p; ps.ax1 = ax1 ? 1 : 0; ps.Gk = Gk;
            (void)later_ev; (void)G; (void)st;
        }
        fold_done = true;
        return true;
    }
};
