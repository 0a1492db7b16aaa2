// A batch of chains on the device: batch plan and memory plan, shared tables and per-(step, chain) metadata (uploads), which kernel family runs the
// batch (plan_geometry, plan_chainres), the average-posterior folds, kept posteriors, carried states, results.
// Part of libblhip's host side: included by blhip.hip INSIDE its anonymous namespace (one translation unit; the split is by subject, not by linkage).
#pragma once

// normalise the kept posterior rows (core.py:389 / :441) (eagerly at the end of the fit, or on first access with option lazy_normalise)
void ensure_post_scaled(blhip_ctx *ctx) {
    if (!ctx->post_valid || ctx->post_scaled) return;
    HIPCHECK(hipSetDevice(ctx->device));
    const long long G = ctx->post_G;
    const unsigned gx = (unsigned)std::min<long long>((G + NTHREADS - 1) / NTHREADS, 4096);
    // (rows [post_row0, post_row1) only: the time-resident kernel has normalised the others itself)
    const int64_t r0 = ctx->post_row0, nrows = ctx->post_row1 - ctx->post_row0;
    for (int64_t b = 0; b < ctx->post_chains && nrows > 0; ++b)
        BL_LAUNCH(scale_rows_kernel, dim3(gx, (unsigned)nrows), dim3(NTHREADS), 0, ctx->stream,
                           ctx->post.as<double>() + ((size_t)b * ctx->post_T + r0) * G, G, ctx->postinv.as<double>() + b * ctx->post_T + r0);
    HIPCHECK(hipGetLastError());
    ctx->post_scaled = true;
}

// axis-0 radius bucket of a chain as bucket_step() will see it (0: no filter, k: radius in (8 (k - 1), 8 k]); -1 if it cannot be told
// from the op values alone
int chain_bucket(const blhip_problem *p, const double *val) {
    int r0 = 0;
    for (int k = 0; k < p->n_ops; ++k) {
        const blhip_op &op = p->ops[k];
        if (op.kind == BLHIP_OP_GRW) {
            if (p->ndim == 2 && op.axis == 0) {
                const double ns = val[k] / p->lattice[0];
                if (!(ns >= 0.0) || ns > 1e6) return -1;
                r0 = std::max(r0, (int)(4.0 * ns + 0.5));
            }
        } else if (op.kind != BLHIP_OP_STATIC) {
            return -1;                               // (other models: their launches are not bucketed by radius)
        }
    }
    return r0 == 0 ? 0 : (r0 + 7) / 8;
}

// -> start index of every batch (+ n_chains at the end): equal shares of at most Bmax chains, each cut moved to the nearest change of
// radius bucket within the slack the memory budget leaves
std::vector<int64_t> plan_batches(const blhip_problem *p, int64_t n_chains, const double *op_values, int64_t Bmax, bool align) {
    // batches of a multiple of 8 chains: whole launches of the chain-resident kernel (8 chains per launch on 512-column grids)
    if (Bmax >= 16) Bmax -= Bmax % 8;
    const int64_t nbatch = (n_chains + Bmax - 1) / Bmax;
    int64_t even = (n_chains + nbatch - 1) / nbatch;
    if (nbatch > 1 && even >= 16) even = std::min(Bmax, (even + 7) / 8 * 8);
    std::vector<int64_t> start;
    for (int64_t c = 0; c < n_chains; c += even) start.push_back(c);
    start.push_back(n_chains);
    if (!align || (int64_t)start.size() != nbatch + 1 || nbatch < 2 || !op_values || p->n_ops == 0) return start;
    std::vector<int> bucket(n_chains);
    for (int64_t c = 0; c < n_chains; ++c) {
        bucket[c] = chain_bucket(p, op_values + c * p->n_ops);
        if (bucket[c] < 0) return start;
    }
    for (int64_t b = 1; b < nbatch; ++b) {
        // candidates: bucket changes between the previous cut and the next one; the batches on both sides must stay <= Bmax
        int64_t best = -1;
        for (int64_t c = start[b - 1] + 1; c < start[b + 1]; ++c) {
            if (bucket[c] == bucket[c - 1]) continue;
            if (c - start[b - 1] > Bmax || start[b + 1] - c > Bmax) continue;
            if (best < 0 || std::llabs(c - start[b]) < std::llabs(best - start[b])) best = c;
        }
        if (best >= 0) start[b] = best;
    }
    return start;
}

// One more cut where the axis-0 radius of a hyper-grid crosses the largest band of the matrix-pipe / chain-resident kernels (40): the
// chains below it keep those kernels, the chains above it take the column pre-pass (blh::vwide_kernel) -- without the cut ONE wide chain
// would route its whole batch through the pre-pass.  Only for grids sorted that way (every chain before the cut <= 40 < every chain after).
void split_wide_axis0(const blhip_problem *p, int64_t n_chains, const double *op_values, std::vector<int64_t> &start, int r_max) {
    if (p->ndim != 2 || !op_values || p->n_ops == 0 || n_chains < 2) return;
    auto radius0 = [&](int64_t c) {
        int r0 = 0;
        for (int k = 0; k < p->n_ops; ++k) {
            const blhip_op &op = p->ops[k];
            if (op.kind == BLHIP_OP_GRW && op.axis == 0) {
                const double ns = op_values[c * p->n_ops + k] / p->lattice[0];
                if (!(ns >= 0.0) || ns > 1e6) return -1;
                r0 = std::max(r0, (int)(4.0 * ns + 0.5));
            }
        }
        return r0;
    };
    int64_t cut = -1;
    for (int64_t c = 0; c < n_chains; ++c) {
        const int r = radius0(c);
        if (r < 0) return;
        if (r > r_max) { if (cut < 0) cut = c; }
        else if (cut >= 0) return;                   // a narrow chain after a wide one: not sorted by radius
    }
    if (cut <= 0) return;
    for (int64_t v : start) if (v == cut) return;
    start.insert(std::upper_bound(start.begin(), start.end(), cut), cut);
}

// ---- the phases of a fit: flags, shared tables (upload), memory plan; then per batch: program, geometry, metadata, forward pass +
//      evidence bookkeeping, backward pass + bookkeeping, carried states / average posterior / kept posterior, results ------------
struct FitFlags {
    bool evidence_only, forward_only, full, keep, accumulate, resume, carry;
};

FitFlags decode_flags(blhip_ctx *ctx, const blhip_problem *p, uint32_t flags, const double *log_w) {
    FitFlags f{};
    f.evidence_only = flags & BLHIP_EVIDENCE_ONLY;
    f.forward_only = (flags & BLHIP_FORWARD_ONLY) && !f.evidence_only;
    f.full = !f.evidence_only && !f.forward_only;
    f.keep = (flags & BLHIP_KEEP_POSTERIOR) && !f.evidence_only;
    f.accumulate = (flags & BLHIP_ACCUMULATE) && !f.evidence_only;
    f.resume = flags & BLHIP_RESUME;
    f.carry = flags & BLHIP_CARRY;
    if (f.resume || f.carry) {
        if (f.full) fail("BLHIP_RESUME / BLHIP_CARRY need a forward-only or evidence-only fit");
        if (p->carry_slot < 0) fail("carry_slot must be >= 0");
    }
    if (f.accumulate && !ctx->acc_active) fail("BLHIP_ACCUMULATE without blhip_accum_begin");
    if (f.accumulate && !log_w) fail("BLHIP_ACCUMULATE needs log_chain_weight");
    return f;
}

// what every chain of the call shares, resident in HBM for the duration of the call
struct DeviceTables {
    double *m0, *m1, *colA, *colB, *rec, *prior, *reset, *uniform, *indep, *lik;
    int rec_len, d;
};

// upload: marginal grids, per-column likelihood constants, per-step data records, prior(s); the (T, G) likelihood table of the
// closed-form table models is built on the device (table_model != 0), a caller-evaluated one (BLHIP_OM_TABLE) is copied
DeviceTables upload_tables(blhip_ctx *ctx, const blhip_problem *p, const Geometry &g, const FitFlags &ff, int table_model) {
    hipStream_t st = ctx->stream;
    const int64_t T = p->T;
    const long long G = g.G;
    DeviceTables D{};
    std::vector<double> rec;
    build_records(p, rec, D.rec_len, D.d);
    std::vector<double> colA(g.n1, 0.0), colB(g.n1, 0.0);
    const double *mcol = p->ndim == 1 ? p->marginal[0] : p->marginal[1];
    if (p->obs_model == BLHIP_OM_GAUSSIAN)
        for (int j = 0; j < g.n1; ++j) {
            const double s = mcol[j];
            colA[j] = 1.0 / (2.0 * s * s);
            colB[j] = 0.5 * std::log(2.0 * M_PI * s * s);
        }
    if (p->obs_model == BLHIP_OM_POISSON)
        for (int j = 0; j < g.n1; ++j) colA[j] = std::exp(-mcol[j]);

    size_t tb = 0;
    tb += carve_size(sizeof(double) * std::max(1, g.n0)) + 3 * carve_size(sizeof(double) * g.n1);
    tb += carve_size(sizeof(double) * rec.size()) + 4 * carve_size(sizeof(double) * G);
    ctx->tables.ensure(tb);
    char *cur = ctx->tables.as<char>();
    D.m0 = carve<double>(cur, std::max(1, g.n0));
    D.m1 = carve<double>(cur, g.n1);
    D.colA = carve<double>(cur, g.n1);
    D.colB = carve<double>(cur, g.n1);
    D.rec = carve<double>(cur, rec.size());
    D.prior = carve<double>(cur, G);
    D.reset = carve<double>(cur, G);
    D.uniform = carve<double>(cur, G);
    D.indep = carve<double>(cur, G);
    if (p->ndim == 2) HIPCHECK(hipMemcpyAsync(D.m0, p->marginal[0], sizeof(double) * g.n0, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(D.m1, mcol, sizeof(double) * g.n1, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(D.colA, colA.data(), sizeof(double) * g.n1, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(D.colB, colB.data(), sizeof(double) * g.n1, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(D.rec, rec.data(), sizeof(double) * rec.size(), hipMemcpyHostToDevice, st));
    // (the prior of a study that is fitted again: the caller's token says the content is the one this place already holds)
    const bool prior_resident = p->prior_token != 0 && p->prior_token == ctx->prior_token && D.prior == ctx->prior_dev && G == ctx->prior_G;
    if (!prior_resident) HIPCHECK(hipMemcpyAsync(D.prior, p->prior, sizeof(double) * G, hipMemcpyHostToDevice, st));
    ctx->prior_token = p->prior_token; ctx->prior_dev = D.prior; ctx->prior_G = G;
    if (p->reset_prior) HIPCHECK(hipMemcpyAsync(D.reset, p->reset_prior, sizeof(double) * G, hipMemcpyHostToDevice, st));
    if (p->indep_prior) HIPCHECK(hipMemcpyAsync(D.indep, p->indep_prior, sizeof(double) * G, hipMemcpyHostToDevice, st));
    if (ff.full) {
        // beta_T = 1/G   core.py:424-425 -- or the caller's backward message (blhip_problem.backward_init)
        if (p->backward_init) HIPCHECK(hipMemcpyAsync(D.uniform, p->backward_init, sizeof(double) * G, hipMemcpyHostToDevice, st));
        else BL_LAUNCH(fill_kernel, dim3(256), dim3(256), 0, st, D.uniform, G, 1.0 / (double)G);
    }
    D.lik = nullptr;
    if (p->obs_model == BLHIP_OM_TABLE) {
        ctx->likbuf.ensure(sizeof(double) * T * G);
        D.lik = ctx->likbuf.as<double>();
        if (table_model) {
            const size_t nd = (size_t)T * p->seg_len * p->data_dim;
            ctx->databuf.ensure(nd * 8);
            HIPCHECK(hipMemcpyAsync(ctx->databuf.p, p->data, nd * 8, hipMemcpyHostToDevice, st));
            const unsigned gx = (unsigned)std::min<long long>((G + NTHREADS - 1) / NTHREADS, 2048);
            BL_LAUNCH(lik_table_kernel, dim3(gx, (unsigned)T), dim3(NTHREADS), 0, st, table_model, D.lik, (long long)G, g.n1,
                               p->ndim, D.m0, D.m1, ctx->databuf.as<double>(), p->seg_len, p->data_dim);
            HIPCHECK(hipGetLastError());
        } else {
            HIPCHECK(hipMemcpyAsync(D.lik, p->lik, sizeof(double) * T * G, hipMemcpyHostToDevice, st));
        }
    }
    sync_stream(ctx, st);   // the host vectors above go out of use
    return D;
}

// memory plan: how many chains fit one batch (state ping-pong + the stored sequence + partial sums per chain within the budget)
constexpr int CHAIN_MIN_ROWS = 32;            // smallest grid (rows) the chain-resident kernels take (on the 128-row geometry)
constexpr int CHAIN_R0_MAX = 80;               // widest band of the chain-resident kernels (ring of 44 entries); option chain_wide = 0: FAST_R0_MAX
constexpr int CHAIN_TALL_ROWS = 1024;         // ... and the one geometry beyond 512 rows: grids of 513 .. 1024 rows (option chain_tall = 0: off)
inline bool chain_rows_ok(int n0) { return n0 >= CHAIN_MIN_ROWS && n0 <= CHAIN_TALL_ROWS; }
inline bool chain_tall(int n0) { return n0 > 512 && n0 <= CHAIN_TALL_ROWS; }

int64_t chains_per_batch(blhip_ctx *ctx, const blhip_problem *p, const Geometry &g, const FitFlags &ff, int64_t n_chains, int post_buffers) {
    const int64_t T = p->T;
    const long long G = g.G;
    size_t free_b = 0, total_b = 0;
    HIPCHECK(hipMemGetInfo(&free_b, &total_b));
    // (every reusable buffer of the context counts as available, so that the plan -- and with it the buffer sizes -- is the same from
    //  fit to fit: a plan that changed between two fits of one study re-allocated the 100-GB sequence buffer, 5 s)
    double budget = std::min((double)free_b + (double)ctx->state.cap + (double)ctx->post.cap + (double)ctx->post2.cap + (double)ctx->accpart.cap,
                             ctx->option("mem_budget_bytes", 0.70 * (double)total_b)) * 0.9;
    // the chain-resident kernels lay their sequences out on a padded geometry (rows 128 / 256 / 512, columns a multiple of 16)
    double Gk = (double)G;
    if (p->ndim == 2 && chain_rows_ok(g.n0))
        Gk = (double)((g.n0 + 127) / 128 * 128) * (double)((g.n1 + blc::WCOL - 1) / blc::WCOL * blc::WCOL);
    {   // walks on both parameters: the transposing kernels lay their sequences out on a SQUARE geometry (blhip_chainax.hpp)
        bool w0 = false, w1 = false;
        for (int k = 0; k < p->n_ops; ++k)
            if (p->ops[k].kind == BLHIP_OP_GRW) { if (g.axis_map[p->ops[k].axis] == 0) w0 = true; else w1 = true; }
        if (p->ndim == 2 && w0 && w1 && std::max(g.n0, g.n1) <= 512) {
            const double n = std::max(g.n0, g.n1) <= 128 ? 128.0 : (std::max(g.n0, g.n1) <= 256 ? 256.0 : 512.0);
            Gk = std::max(Gk, n * n);
        }
    }
    // the partial accumulators of the fused fold (ChainRun::setup: one (T, G) slot per block column of a launch) come out of the same memory
    // -- only where the chain-resident path can be taken at all (else they are never allocated: a narrow grid with a long series
    //    gave up its whole budget to 128 slots it never used and ran one chain per batch), and never more than half of the budget
    if (ff.accumulate && ff.full && p->ndim == 2 && p->obs_model == BLHIP_OM_GAUSSIAN && chain_rows_ok(g.n0) &&
        g.n1 >= 1 && g.n1 <= 16 * blc::MAX_STRIPS && ctx->option("chain_resident", 1.0) != 0.0 && ctx->resident_ok) {
        const double slots = std::max(1, std::min(ctx->num_cus, 256) / ((g.n1 + blc::WCOL - 1) / blc::WCOL));
        budget -= std::min(0.5 * budget, std::min<double>(slots, (double)n_chains) * (double)T * Gk * 8.0);
    }
    const double per_chain = (ff.evidence_only ? 2.0 : (double)post_buffers * (double)T + 2.0) * Gk * 8.0 +
                             (double)T * NRED * 8.0 * 2 * 64.0 /*partials, rough*/;
    int64_t Bmax = (int64_t)std::max(1.0, std::floor(budget / per_chain));
    // (1-D grids: a chain is a few KB and a batch of the chain-resident 1-D kernel costs a fixed ~1 ms of launches, syncs and read-backs --
    //  the reference's published break-point study, 23 400 chains: 61 ms with batches of 1024, 52 with 4096, 54 with 8192)
    Bmax = std::min<int64_t>(Bmax, (int64_t)ctx->option("max_batch", p->ndim == 1 ? 4096 : 1024));
    Bmax = std::min<int64_t>(Bmax, 65535);
    if (ff.keep && n_chains > Bmax) fail("BLHIP_KEEP_POSTERIOR: %lld chains do not fit in device memory at once", (long long)n_chains);
    if ((ff.resume || ff.carry) && n_chains > Bmax) fail("carried states: %lld chains do not fit in one batch", (long long)n_chains);
    if (ff.resume) {
        auto it = ctx->carry.find(p->carry_slot);
        if (it == ctx->carry.end() || !it->second.valid) fail("BLHIP_RESUME: carry slot %d holds no state", p->carry_slot);
        if (it->second.chains != n_chains || it->second.G != G)
            fail("BLHIP_RESUME: carry slot %d holds %lld chains x %lld cells, the call has %lld x %lld", p->carry_slot,
                 (long long)it->second.chains, (long long)it->second.G, (long long)n_chains, (long long)G);
    }
    return Bmax;
}

// which kernel family runs a batch and with what block geometry (segment lengths from a small cost model: long segments read every
// element once + 2 R0 halo rows per segment, short ones give enough blocks to fill 256 CUs when there are few chains)
struct GeometryPlan {
    bool shift1d = false;        // the chain-resident 1-D kernel's flavours with spline shifts (Deterministic steps) / clamps (RegimeSwitch, NotEqual)
    bool clamp1d = false;        // ... the one with clamps (CL = 2)
    bool fast = false, fused1d = false, use_mfma = false;
    bool chain1d = false;         // 1-D batches: one block per chain runs the whole pass (blhip_chain1d.hpp); bookkeeping of a K = 1 fused pass
    bool wideH = false;           // axis-1 walks wider than the fused kernels' halo: row filter as a pre-pass per step (blhip_hwide.hpp)
    bool wideV = false;           // axis-0 walks wider than the matrix-pipe kernels' largest band: column filter as a pre-pass, no stencil left
    bool hSplit = false;          // wideH: chains of the batch whose axis-1 filter is absent (or narrow: hFusedMax > 0) keep their fused kernels
    int hFusedMax = 0;
    int64_t fusedK = 1;
    int f1_TJ = 128;
    Tile tile{};
    int fastS = 0, fast_nseg = 1, fast_fnblk = 1, mS = 0, m_nseg = 1, m_tiles_j = 1, m_nblk = 1;
};

GeometryPlan plan_geometry(blhip_ctx *ctx, const blhip_problem *p, const Geometry &g, const ChainProgram &prog, int64_t B, int d,
                       bool resume, bool carry, const TapTable *taps = nullptr) {
    GeometryPlan gp;
    const int64_t T = p->T;
    // fast path (blhip_fast.hpp) when the whole batch qualifies, otherwise the generic LDS-tile kernel
    const bool wide_h_ok = prog.LW1 <= blh::HW_MAX && prog.LW1 < g.n1 && ctx->option("wide_h", 1.0) != 0.0;
    const bool wide_v = prog.LW0 > FAST_R0_MAX && prog.LW0 <= blh::VW_MAX && prog.LW0 < g.n0 && wide_h_ok && ctx->option("wide_v", 1.0) != 0.0;
    gp.fast = p->ndim == 2 && (p->obs_model == BLHIP_OM_GAUSSIAN || p->obs_model == BLHIP_OM_TABLE) &&
                      ctx->option("fast", 1.0) != 0.0 && !prog.has_clamp && (prog.LW0 <= FAST_R0_MAX || wide_v) &&
                      (prog.LW1 <= blf::R1MAX || wide_h_ok) &&
                      g.n0 >= (wide_v ? 0 : ((prog.LW0 + 7) / 8) * 8) + 2 * blf::CH && g.n1 >= 2 * blf::R1MAX && d <= blf::DMAX;
    gp.wideV = gp.fast && wide_v;
    gp.wideH = gp.fast && (prog.LW1 > blf::R1MAX || (gp.wideV && prog.LW1 > 0));       // (with a column pre-pass every filter runs as a pre-pass)
    // Chains of such a batch WITHOUT an axis-1 filter (a hyper-grid that includes the width 0) skip the pre-pass: it would be a copy.
    // wide_h_fused_max = 8: chains with a narrow filter keep the fused kernels' own axis-1 part as well -- 15 % fewer bytes on
    // extra.c4_both_axes but no faster (the fused both-axes kernels at radii up to 40 are bound by the fp64 pipe: 5.44e10 -> 5.47e10), so
    // the default sends every filter through the pre-pass.
    bool any_narrow = false;
    gp.hFusedMax = std::min(blf::R1MAX, std::max(0, (int)ctx->option("wide_h_fused_max", 0.0)));
    if (gp.wideH && !gp.wideV && taps && ctx->option("wide_h_split", 1.0) != 0.0) {
        bool any_none = false;
        for (size_t e = 0; e < prog.tapF1.size() && !(any_narrow && any_none); ++e) {
            const int k = prog.tapF1[e];
            if (k < 0) any_none = true; else if (gp.hFusedMax > 0 && taps->lw[k] <= gp.hFusedMax) any_narrow = true;
        }
        gp.hSplit = any_narrow || any_none;
    }
    // (programs whose only clamp mode is a Deterministic model's spline shift: the chain-resident kernel has a flavour for them
    //  -- bl1c::chain1d_kernel SHIFT --, the K-steps-per-launch and persistent kernels have not: chain1d or the generic kernel)
    // (round 6: ... and the one with the clamps of RegimeSwitch / NotEqual -- bl1c::chain1d_kernel CL = 2; the dense zero-boundary kernels of
    //  the AlphaStable walk keep the generic kernel)
    const bool shift1d = p->ndim == 1 && prog.has_clamp && !prog.dense_clamp && ctx->option("chain1d_shift", 1.0) != 0.0 &&
                         (!prog.other_clamp || ctx->option("chain1d_clamp", 1.0) != 0.0);
    if (p->ndim == 1 && !gp.fast && (!prog.has_clamp || shift1d) && ctx->option("fuse1d", 8.0) >= 1.0 &&
        (p->obs_model == BLHIP_OM_POISSON || p->obs_model == BLHIP_OM_GAUSSIAN_MEAN || p->obs_model == BLHIP_OM_TABLE)) {
        gp.f1_TJ = 128;
        gp.fusedK = std::max<int64_t>(1, std::min<int64_t>((int64_t)ctx->option("fuse1d", 8.0), T));
        // keep the redundantly recomputed halo (K * LW cells per side) within ~4x the owned cells and the window in LDS
        while (gp.fusedK > 1 && (gp.fusedK * prog.LW1 > 2 * gp.f1_TJ || (size_t)(gp.f1_TJ + 2 * gp.fusedK * prog.LW1) * 32 > 96 * 1024)) --gp.fusedK;
        gp.fused1d = (size_t)(gp.f1_TJ + 2 * gp.fusedK * prog.LW1) * 32 + (size_t)gp.fusedK * gp.f1_TJ * 32 + 4096 <= 150 * 1024;
        // Batches of chains: one block per chain for the whole pass when that is cheaper per step than the alternatives.  Per step
        // (shader cycles; measured with tools/probe.py chain1d, profiles/r04_notes.md): the chain's row is filtered out of ONE CU's LDS --
        // n (2 lw + 1) 16-byte operand pairs at 128 B per clock, ~1 k cycles of barrier / sums / likelihood -- and ceil(B / CUs) blocks
        // share a CU one after the other; the K-steps-per-launch path costs a launch (~14 k cycles) every K steps, the persistent
        // one (all blocks of all chains on the chip at once) ~4 k (K > 1) / ~7 k (K = 1: a hand-off per step) per step.
        const double c1d_mode = ctx->option("chain1d", 1.0);
        if (shift1d) gp.fused1d = true;              // (decided below: without the chain-resident kernel the batch keeps the generic one)
        if (gp.fused1d && c1d_mode != 0.0 && !resume && !carry && prog.LW1 < g.n1 && g.n1 <= bl1c::NMAX &&
            bl1c::lds_doubles(g.n1, prog.LW1, shift1d) * 8 <= 150 * 1024) {
            // microseconds per time step of the whole batch, fitted to tools/probe.py chain1d (profiles/r04_notes.md): a block's step =
            // 1.5 us + 1.0 ns per cell (likelihood from the shared table; 2.2 ns with Poisson's pow() in the kernel) + 44 ps per cell and
            // tap (the stencil's operand pairs come out of ONE CU's LDS at ~9 per clock), blocks beyond the chip's capacity queue up;
            // persistent K-step kernel (all blocks of all chains on the chip at once) 1.7 us + 40 ns per cell of radius; a launch per K
            // steps 1.5 us + (29 + 0.63 radius) ps per cell of the batch
            const int cus = std::min(ctx->num_cus, 256);
            const double n = g.n1, lw = prog.LW1;
            // (rows longer than a block: two cells per thread share their operand pairs, 21 ps per cell and tap -- launch_chain1d)
            const double tap_us = g.n1 > bl1c::NT && true ? 2.1e-5 : 4.4e-5;
            const double est_c1d = (double)((B + cus - 1) / cus) * (1.5 + n * (B >= 4 ? 0.0010 : 0.0022) + n * (2.0 * lw + 1.0) * tap_us);
            const int nblk_f = (g.n1 + gp.f1_TJ - 1) / gp.f1_TJ;
            const bool p1d_ok = ctx->option("persist1d", 1.0) != 0.0 && (long long)nblk_f * B <= cus && T > gp.fusedK;
            const double est_other = p1d_ok ? 1.7 + 0.04 * lw : 1.5 + (double)B * n * (29.0 + 0.63 * lw) * 1e-6;
            // (shift1d: the alternative is a launch per step.  A single chain takes the kernel too when the model says so -- rows of a few
            //  hundred cells with a narrow stencil: 200 cells, radius 27: 2.4 against 2.8 us per step)
            // (clamps without a Deterministic model -- the reference's regime-switch tutorial is ONE such chain: a step of a few hundred cells
            //  costs the block ~2 us against a launch of the generic kernel)
            gp.chain1d = c1d_mode == 2.0 || (shift1d && (B >= 2 || !prog.has_shift)) || (!shift1d && est_c1d < est_other);
            if (gp.chain1d) { gp.fusedK = 1; gp.f1_TJ = g.n1; }
        }
        if (shift1d && !gp.chain1d) gp.fused1d = false;
        gp.shift1d = shift1d && gp.chain1d;
        gp.clamp1d = gp.shift1d && prog.other_clamp;
    }
    if (gp.fast) {
        gp.tile.TI = blf::CH; gp.tile.LW0 = gp.wideV ? 0 : prog.LW0; gp.tile.LW1 = (prog.LW1 > 0 && (!gp.wideH || (gp.hSplit && any_narrow))) ? blf::R1MAX : 0;
        gp.tile.TJ = blf::BW - 2 * gp.tile.LW1;
        gp.tile.tiles_j = (g.n1 + gp.tile.TJ - 1) / gp.tile.TJ;
        // rows per block segment: long segments read every element once (+ 2*R0 halo rows per segment), short ones
        // give enough blocks to fill 256 CUs when there are few chains.  Model: cost = waves * blocks_per_CU * rows.
        const long long colblocks = (long long)gp.tile.tiles_j * B;
        const int R0 = (prog.LW0 == 0 || gp.wideV) ? 0 : ((prog.LW0 + 7) / 8) * 8;
        double best = 1e300;
        const int forceS = (int)ctx->option("fast_S", 0);
        for (int k = 1; k <= 4; k *= 2) {
            const double pen = k == 1 ? 1.6 : (k == 2 ? 1.15 : 1.0);
            for (int ns = 1; ns <= std::max(1, g.n0 / 16); ++ns) {
                int S = ((g.n0 + ns - 1) / ns + blf::CH - 1) / blf::CH * blf::CH;
                const int real = (g.n0 + S - 1) / S;
                const long long blocks = colblocks * real;
                const long long waves = (blocks + 256LL * k - 1) / (256LL * k);
                const double cost = pen * (double)waves * k * (S + 2.0 * R0 + 4.0);
                if (cost < best - 1e-9) { best = cost; gp.fastS = S; gp.fast_nseg = real; }
            }
        }
        if (forceS > 0) { gp.fastS = (forceS + blf::CH - 1) / blf::CH * blf::CH; gp.fast_nseg = (g.n0 + gp.fastS - 1) / gp.fastS; }
        gp.tile.tiles_i = gp.fast_nseg;
        gp.fast_fnblk = gp.tile.tiles_j * gp.fast_nseg;
        // geometry of the matrix-pipe kernel (64-column strips, segments of mS rows, mS a multiple of 16)
        {
            gp.m_tiles_j = (g.n1 + blm::BCOL - 1) / blm::BCOL;
            const long long mcol = (long long)gp.m_tiles_j * B;
            double mbest = 1e300;
            for (int k = 1; k <= 4; ++k) {                     // resident blocks per CU
                const double pen = k == 1 ? 1.5 : (k == 2 ? 1.15 : 1.0);
                for (int ns = 1; ns <= std::max(1, g.n0 / 32); ++ns) {
                    int S = ((g.n0 + ns - 1) / ns + blm::SEG_Q - 1) / blm::SEG_Q * blm::SEG_Q;
                    if (S > blm::MS_MAX) continue;
                    const int real = (g.n0 + S - 1) / S;
                    const long long blocks = mcol * real;
                    const long long waves = (blocks + 256LL * k - 1) / (256LL * k);
                    const double cost = pen * (double)waves * k * (S + 1.0 * R0 + 24.0);
                    if (cost < mbest - 1e-9) { mbest = cost; gp.mS = S; gp.m_nseg = real; }
                }
            }
            const int forceM = (int)ctx->option("mfma_S", 0);
            if (forceM > 0) { gp.mS = std::min(blm::MS_MAX, (forceM + blm::SEG_Q - 1) / blm::SEG_Q * blm::SEG_Q); gp.m_nseg = (g.n0 + gp.mS - 1) / gp.mS; }
            if (gp.mS == 0) { gp.mS = blm::MS_MAX; gp.m_nseg = (g.n0 + gp.mS - 1) / gp.mS; }
            gp.m_nblk = gp.m_tiles_j * gp.m_nseg;
        }
        gp.use_mfma = ctx->option("mfma", 1.0) != 0.0;
        gp.tile.nblk = gp.use_mfma ? std::max(gp.fast_fnblk, gp.m_nblk) : gp.fast_fnblk;
        gp.tile.lds_bytes = 0;
    } else if (gp.fused1d) {
        gp.tile.TI = 1; gp.tile.TJ = gp.f1_TJ; gp.tile.LW0 = 0; gp.tile.LW1 = prog.LW1; gp.tile.tiles_i = 1;
        gp.tile.tiles_j = (g.n1 + gp.f1_TJ - 1) / gp.f1_TJ; gp.tile.nblk = gp.tile.tiles_j; gp.tile.lds_bytes = 0;
    } else {
        gp.tile = choose_tile(ctx, g, prog.LW0, prog.LW1, prog.whole_row);
        if (prog.whole_row && gp.tile.tiles_j != 1) fail("internal: a two-stage spline shift needs the whole row in one tile");
    }
    return gp;
}

// per-(step, chain) metadata of a batch in HBM: source kinds, tap-set ids, clamp modes, the per-step launch order of the radius buckets,
// the tap table; plus scratch the finalisation kernels use
struct DeviceMeta {
    unsigned char *kindF, *kindB, *cmodeF, *cmodeB;
    double *limitF, *limitB;
    int *tapF0, *tapF1, *tapB0, *tapB1, *orderF, *orderB;
    double *taps;
    int *off, *lw, *lw2;
    double *invN, *w, *dump;
    // host copies the launch loop reads
    std::vector<int> h_orderF, h_orderB;
    std::vector<std::vector<FastRange>> rangesF, rangesB;
};

void upload_metadata(blhip_ctx *ctx, const blhip_problem *p, const ChainProgram &prog, TapTable &taps, int64_t B, bool full, bool fast, int nblk,
                     DeviceMeta &M, bool wideH = false, bool wideV = false, bool h_split = false, int h_fused_max = 0) {
    hipStream_t st = ctx->stream;
    const int64_t T = p->T;
    const size_t nT = (size_t)T * B;
    taps.w.resize(taps.w.size() + 64, 0.0);      // zero padding: the fast kernels read up to R0 weights per tap set
    size_t mb = 4 * carve_size(nT) + 6 * carve_size(nT * sizeof(int)) + carve_size(taps.w.size() * 8 + 8) +
                3 * carve_size(taps.off.size() * 4 + 4) + 4 * carve_size(sizeof(double) * nT) + carve_size(8 * B) + carve_size(8 * 4 * NTHREADS);
    ctx->meta.ensure(mb);
    char *cur = ctx->meta.as<char>();
    M.kindF = carve<unsigned char>(cur, nT); M.kindB = carve<unsigned char>(cur, nT);
    M.cmodeF = carve<unsigned char>(cur, nT); M.cmodeB = carve<unsigned char>(cur, nT);
    M.limitF = carve<double>(cur, nT); M.limitB = carve<double>(cur, nT);
    M.tapF0 = carve<int>(cur, nT); M.tapF1 = carve<int>(cur, nT);
    M.tapB0 = carve<int>(cur, nT); M.tapB1 = carve<int>(cur, nT);
    M.orderF = carve<int>(cur, nT); M.orderB = carve<int>(cur, nT);
    M.taps = carve<double>(cur, taps.w.size() + 1);
    M.off = carve<int>(cur, taps.off.size() + 1); M.lw = carve<int>(cur, taps.off.size() + 1);
    M.lw2 = carve<int>(cur, taps.off.size() + 1);
    M.invN = carve<double>(cur, nT);
    (void)carve<double>(cur, nT);
    M.w = carve<double>(cur, B);
    M.dump = carve<double>(cur, 4 * NTHREADS);   // (the halo wave of an H block spreads its dummy stores over 8 x 64 slots)
    HIPCHECK(hipMemcpyAsync(M.kindF, prog.kindF.data(), nT, hipMemcpyHostToDevice, st));
    // (what a batch does not use is not copied: on 1-D grids every walk acts on the internal column axis -- the row-axis tap ids are all -1, a
    //  memset --, and the clamp levels exist only for RegimeSwitch / NotEqual: the published break-point study, 23 400 chains x 41 steps with
    //  Deterministic shifts only, uploaded 6.7 MB of metadata per batch from pageable memory, 0.9 ms each, 2.7 MB of it zeros)
    const bool one_d = p->ndim == 1;
    if (one_d) HIPCHECK(hipMemsetAsync(M.tapF0, 0xFF, nT * 4, st));
    else HIPCHECK(hipMemcpyAsync(M.tapF0, prog.tapF0.data(), nT * 4, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(M.tapF1, prog.tapF1.data(), nT * 4, hipMemcpyHostToDevice, st));
    if (prog.has_clamp) {
        HIPCHECK(hipMemcpyAsync(M.cmodeF, prog.cmodeF.data(), nT, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(M.cmodeB, prog.cmodeB.data(), nT, hipMemcpyHostToDevice, st));
        if (prog.other_clamp) {
            HIPCHECK(hipMemcpyAsync(M.limitF, prog.limitF.data(), nT * 8, hipMemcpyHostToDevice, st));
            HIPCHECK(hipMemcpyAsync(M.limitB, prog.limitB.data(), nT * 8, hipMemcpyHostToDevice, st));
        } else {
            HIPCHECK(hipMemsetAsync(M.limitF, 0, nT * 8, st));
            HIPCHECK(hipMemsetAsync(M.limitB, 0, nT * 8, st));
        }
    }
    if (full) {
        HIPCHECK(hipMemcpyAsync(M.kindB, prog.kindB.data(), nT, hipMemcpyHostToDevice, st));
        if (one_d) HIPCHECK(hipMemsetAsync(M.tapB0, 0xFF, nT * 4, st));
        else HIPCHECK(hipMemcpyAsync(M.tapB0, prog.tapB0.data(), nT * 4, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(M.tapB1, prog.tapB1.data(), nT * 4, hipMemcpyHostToDevice, st));
    }
    if (fast) {
        // a launch with fewer than ~128 blocks leaves most of the 256 CUs idle: a radius bucket with fewer blocks joins the next one
        const long long min_blocks = 128;
        const int min_chains = (int)std::min<long long>(B, (min_blocks + (long long)nblk - 1) / nblk);
        // (wideH: the axis-1 filters wider than the fused kernels' 8 columns run in the pre-pass, and the fused kernels of those chains are
        //  launched without an axis-1 part; chains of the same step with a narrow filter or none keep their fused kernels -- unless the
        //  step also has the axis-0 pre-pass, or with wide_h_split = 0: then every chain of the step goes through the pre-pass)
        const std::vector<int> no_h((wideH || wideV) ? (size_t)B : 0, -1);
        auto all_pre = [&](std::vector<FastRange> &rs) { if (wideH && !h_split) for (auto &r : rs) r.pre = true; };
        M.h_orderF.resize(nT); M.rangesF.resize(T);
        for (int64_t t = 0; t < T; ++t) {
            bucket_step(wideV ? no_h.data() : &prog.tapF0[t * B], (wideH && !h_split) ? no_h.data() : &prog.tapF1[t * B], taps.lw, (int)B, &M.h_orderF[t * B], M.rangesF[t], min_chains,
                        h_split ? h_fused_max : -1);
            all_pre(M.rangesF[t]);
        }
        HIPCHECK(hipMemcpyAsync(M.orderF, M.h_orderF.data(), nT * 4, hipMemcpyHostToDevice, st));
        if (full) {
            M.h_orderB.resize(nT); M.rangesB.resize(T);
            for (int64_t t = 0; t < T; ++t) {
                bucket_step(wideV ? no_h.data() : &prog.tapB0[t * B], (wideH && !h_split) ? no_h.data() : &prog.tapB1[t * B], taps.lw, (int)B, &M.h_orderB[t * B], M.rangesB[t], min_chains,
                            h_split ? h_fused_max : -1);
                all_pre(M.rangesB[t]);
            }
            HIPCHECK(hipMemcpyAsync(M.orderB, M.h_orderB.data(), nT * 4, hipMemcpyHostToDevice, st));
        }
    }
    if (!taps.w.empty()) {
        HIPCHECK(hipMemcpyAsync(M.taps, taps.w.data(), taps.w.size() * 8, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(M.off, taps.off.data(), taps.off.size() * 4, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(M.lw, taps.lw.data(), taps.lw.size() * 4, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(M.lw2, taps.lw2.data(), taps.lw2.size() * 4, hipMemcpyHostToDevice, st));
    }
}

// host-side results of one batch of chains
struct BatchOutcome {
    std::vector<double> logE, local, means, invN;    // (B,), (B, T), (B, ndim, T), (B, T): 1 / row sum of the stored sequence
    std::vector<int64_t> abort_step;
    std::vector<int32_t> abort_phase;
};

// BLHIP_CARRY: keep every chain's filtered distribution of the last step, normalised (core.py:2173)
void store_carry(blhip_ctx *ctx, const blhip_problem *p, int64_t B, long long G, const double *redF, const double *fin, long long fstr,
                 double *d_w, bool has_clamp) {
    hipStream_t st = ctx->stream;
    const int64_t T = p->T;
    blhip_ctx::Carry &cs = ctx->carry[p->carry_slot];
    cs.buf.ensure((size_t)B * G * 8);
    std::vector<double> inv(B);
    for (int64_t b = 0; b < B; ++b) inv[b] = 1.0 / redF[((size_t)(T - 1) * B + b) * NRED];
    HIPCHECK(hipMemcpyAsync(d_w, inv.data(), B * 8, hipMemcpyHostToDevice, st));
    const unsigned gx = (unsigned)std::min<long long>((G + NTHREADS - 1) / NTHREADS, 4096);
    BL_LAUNCH(carry_store_kernel, dim3(gx, (unsigned)B), dim3(NTHREADS), 0, st, cs.buf.as<double>(), fin, fstr, G, d_w);
    sync_stream(ctx, st);
    cs.chains = B; cs.G = G; cs.valid = true;
    cs.maxv.clear();
    if (has_clamp)                               // clamp batches run the generic kernel, which reports the state maximum
        for (int64_t b = 0; b < B; ++b) cs.maxv.push_back(redF[((size_t)(T - 1) * B + b) * NRED + 6] * inv[b]);
}

// fold the batch into the average posterior (core.py:1358-1366): linear accumulator with a running reference exponent.
// Two halves, so that the kernel can be launched later than the bookkeeping is done (overlapped folds, see do_fit): prepare_fold
// turns the batch's evidences into weights (staged in h_w / h_invN, which must stay valid until the launch has consumed them) and
// advances the accumulator's reference; launch_fold copies them to the device and runs the pass on `st`.
struct FoldJob {
    bool pending = false;
    const double *d_post = nullptr;
    int64_t B = 0;
    double *h_w = nullptr, *h_invN = nullptr;        // host staging (B), (T * B)
    double *d_w = nullptr, *d_invN = nullptr;        // device copies
    double r = 0.0;                                  // factor of what the accumulator already holds
    int first = 0;
    int parity = 0;
    int sm_n0 = 0;                                   // > 0: d_post is in the chain-resident kernel's strip-major layout (rows per strip)
    int pad_n0p = 0, pad_n0 = 0, pad_n1 = 0;         // > 0: ... on a padded geometry (rows per strip n0p; the grid's true sizes)
    int pad_ax = 0;                                  //      ... in the alternating layouts of the both-axes kernels (blk::ax_layout_b)
    long long pad_step = 0;                          //      doubles per time step of a chain's sequence there
};

bool prepare_fold(blhip_ctx *ctx, int64_t T, int64_t B, const BatchOutcome &out, const double *log_w_batch, FoldJob &job) {
    double newref = ctx->acc_logref;
    std::vector<double> lw(B, -INFINITY);
    std::vector<char> valid(B, 0);
    for (int64_t b = 0; b < B; ++b) {
        // np.isfinite(logEvidence) guard (core.py:1358); a zero hyper-prior contributes log(0) = -inf, i.e. nothing
        valid[b] = out.abort_step[b] < 0 && std::isfinite(out.logE[b]) && std::isfinite(log_w_batch[b]);
        if (!valid[b]) continue;
        lw[b] = out.logE[b] + log_w_batch[b];
        if (lw[b] > newref) newref = lw[b];
    }
    if (!std::isfinite(newref)) return false;
    int nfold = 0;
    for (int64_t b = 0; b < B; ++b) {
        job.h_w[b] = valid[b] ? std::exp(lw[b] - newref) : 0.0;
        nfold += valid[b] ? 1 : 0;
    }
    std::memcpy(job.h_invN, out.invN.data(), (size_t)T * B * 8);
    job.r = ctx->acc_first ? 0.0 : std::exp(ctx->acc_logref - newref);
    job.first = ctx->acc_first ? 1 : 0;
    job.B = B;
    ctx->timing.accumulate_launches += 1;
    ctx->acc_logref = newref;
    ctx->acc_first = false;
    ctx->acc_folded += nfold;
    return true;
}

void launch_fold(blhip_ctx *ctx, int64_t T, long long G, const FoldJob &job, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    const int64_t B = job.B;
    HIPCHECK(hipMemcpyAsync(job.d_w, job.h_w, B * 8, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(job.d_invN, job.h_invN, (size_t)T * B * 8, hipMemcpyHostToDevice, st));
    HIPCHECK(hipEventRecord(ev0, st));
    if (job.pad_n0p > 0) {
        BL_LAUNCH(accumulate_pad_kernel, dim3((unsigned)((G + NTHREADS - 1) / NTHREADS), (unsigned)T), dim3(NTHREADS), 0, st, ctx->acc,
                           job.d_post, (long long)T * job.pad_step, (int)B, job.pad_n0, job.pad_n1, (int)T, job.d_w, job.d_invN, job.r, job.first,
                           job.pad_n0p, job.pad_step, job.pad_ax);
    } else if (job.sm_n0 == 0 && B >= 16 && ((G / 2 + NTHREADS - 1) / NTHREADS) * T < 1024) {        // small grids: too few blocks with a thread per cell
        BL_LAUNCH(accumulate_small_kernel, dim3((unsigned)((G + 63) / 64), (unsigned)T), dim3(NTHREADS), 0, st, ctx->acc, job.d_post,
                           (long long)T * G, (int)B, G, (int)T, job.d_w, job.d_invN, job.r, job.first);
    } else if ((G & 1) == 0 && ((uintptr_t)ctx->acc & 15) == 0) {
        const unsigned gx2 = (unsigned)((G / 2 + NTHREADS - 1) / NTHREADS);
        BL_LAUNCH(accumulate2_kernel, dim3(gx2, (unsigned)T), dim3(NTHREADS), 0, st, ctx->acc, job.d_post,
                           (long long)T * G, (int)B, G, (int)T, job.d_w, job.d_invN, job.r, job.first, job.sm_n0);
    } else {
        if (job.sm_n0 > 0) fail("internal: strip-major sequences need an even number of cells and a 16-byte aligned accumulator");
        const unsigned gx = (unsigned)std::min<long long>((G + NTHREADS - 1) / NTHREADS, 4096);
        BL_LAUNCH(accumulate_kernel, dim3(gx, (unsigned)T), dim3(NTHREADS), 0, st, ctx->acc, job.d_post,
                           (long long)T * G, (int)B, G, (int)T, job.d_w, job.d_invN, job.r, job.first);
    }
    HIPCHECK(hipEventRecord(ev1, st));
}

// the whole fold on the main stream, waited for
// (later_ev: do not wait -- the fold stays in front of whatever the stream runs next, e.g. the next batch's metadata uploads and forward
//  pass; its two timing events are appended for the caller to read after the stream has drained.  The page-locked staging of the weights is
//  reused by the next batch's fold only after that batch's passes have been waited for, on the same stream.)
void fold_accumulate(blhip_ctx *ctx, int64_t T, long long G, int64_t B, const BatchOutcome &out, const double *log_w_batch, const double *d_post,
                     double *d_w, double *d_invN, int sm_n0 = 0, const FoldJob *layout = nullptr, std::vector<hipEvent_t> *later_ev = nullptr) {
    ctx->pinA.ensure(((size_t)B + (size_t)T * B) * 8);
    FoldJob job;
    if (layout) job = *layout;
    job.sm_n0 = sm_n0;
    job.h_w = ctx->pinA.as<double>(); job.h_invN = job.h_w + B;
    job.d_w = d_w; job.d_invN = d_invN; job.d_post = d_post;
    if (!prepare_fold(ctx, T, B, out, log_w_batch, job)) return;
    if (later_ev) {
        hipEvent_t e0, e1;
        HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
        later_ev->push_back(e0); later_ev->push_back(e1);
        launch_fold(ctx, T, G, job, ctx->stream, e0, e1);
        return;
    }
    launch_fold(ctx, T, G, job, ctx->stream, ctx->ev[4], ctx->ev[5]);
    sync_stream(ctx, ctx->stream);
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5]));
    ctx->timing.accumulate_ms += ms;
}

// the batch's sequence stays on the device as the kept posterior; rows [row0, row1) still carry their raw sums (core.py:389 / :441)
void keep_posterior(blhip_ctx *ctx, const Geometry &g, int64_t T, int64_t B, const BatchOutcome &out, int64_t row0, int64_t row1) {
    hipStream_t st = ctx->stream;
    ctx->postinv.ensure((size_t)T * B * 8);
    HIPCHECK(hipMemcpyAsync(ctx->postinv.p, out.invN.data(), (size_t)T * B * 8, hipMemcpyHostToDevice, st));
    sync_stream(ctx, st);
    ctx->post_valid = true; ctx->post_scaled = false; ctx->post_chains = B; ctx->post_T = T; ctx->post_G = g.G;
    ctx->post_row0 = row0; ctx->post_row1 = row1;
    ctx->post_n0 = g.n0; ctx->post_n1 = g.n1;
    // normalised now, as part of the fit (core.py:441 is inside Study.fit)
    ensure_post_scaled(ctx); sync_stream(ctx, st);
}

void write_results(blhip_result *res, const blhip_problem *p, int64_t c0, int64_t B, const BatchOutcome &out, bool with_means) {
    if (!res) return;
    const int64_t T = p->T;
    for (int64_t b = 0; b < B; ++b) {
        if (res->log_evidence) res->log_evidence[c0 + b] = out.logE[b];
        if (res->abort_step) res->abort_step[c0 + b] = out.abort_step[b];
        if (res->abort_phase) res->abort_phase[c0 + b] = out.abort_phase[b];
        if (res->local_evidence)
            std::memcpy(res->local_evidence + (size_t)(c0 + b) * T, &out.local[(size_t)b * T], T * 8);
        if (res->posterior_mean && with_means)
            std::memcpy(res->posterior_mean + (size_t)(c0 + b) * p->ndim * T, &out.means[(size_t)b * p->ndim * T], (size_t)p->ndim * T * 8);
    }
}

// The chain-resident path (blhip_chainres.hpp): which chains of the batch run together, in which order, with which band width.
struct ChainResPlan {
    int ntw = 0, strips = 0, cpr = 0;            // product tiles per wave, strips per chain, chains per launch
    int n0p = 0, n1p = 0;                        // the geometry the kernels work on: rows 128 / 256 / 512, columns a multiple of 16
    bool pad = false;                            // the grid is smaller than that (padded cells hold zeros; sequences private to the fit only)
    int r0_max = 40;                             // widest axis-0 radius the launch may carry (set by the caller: 80 for 1024 rows)
    bool has_reset = false;                      // change points: some steps consume the reset distribution
    bool mixed = false;                          // ... in chains that also filter (random walk + change point in one model)
    std::vector<unsigned char> ckF, ckB;         // [T][B] what a step of the chain kernels consumes: SRC_PREV / SRC_RESET (| 0x80: unfiltered)
    std::vector<int> order, tap_id;              // chains sorted by stencil radius; the chain's axis-0 kernel (-1: none)
    bool allow_ax1 = false;                      // (set by the caller) walks on the second parameter too may be planned: blc::chainax_kernel
    bool ax1 = false;                            // ... and some chain has one: the batch runs the transposing kernels (square exact geometry)
    std::vector<int> tap_id1;                    // the chain's axis-1 kernel (-1: none)
    std::vector<int> round_start, round_nk;      // launches: order[round_start[r] .. round_start[r + 1]), band blocks NK
};

// every chain: prior, then the SAME axis-0 kernel at every step, nothing on axis 1 (forward; mirrored backward)
bool plan_chainres(const Geometry &g, const ChainProgram &prog, const TapTable &taps, int64_t B, int64_t T, bool full, int cus, ChainResPlan &cp) {
    // any grid of 32 .. 512 rows: the kernels work on the next geometry of 128 / 256 / 384 / 512 rows x a multiple of 16 columns;
    // 513 .. 1024 rows: 1024 rows x a multiple of 16 columns (one copy of the strip in LDS: blc::chain_kernel TALL; change-point batches keep their state in registers there too)
    if (!chain_rows_ok(g.n0)) return false;
    cp.n0p = (g.n0 + 127) / 128 * 128;
    cp.n1p = (g.n1 + blc::WCOL - 1) / blc::WCOL * blc::WCOL;
    if (chain_tall(g.n0)) cp.n0p = CHAIN_TALL_ROWS;
    cp.pad = cp.n0p != g.n0 || cp.n1p != g.n1;
    cp.strips = cp.n1p / blc::WCOL;
    cp.ntw = cp.n0p / (blc::NW * blc::TM);
    if (cp.strips > blc::MAX_STRIPS || cp.strips > cus) return false;
    cp.cpr = cus / cp.strips;
    cp.tap_id.assign(B, -1);
    cp.tap_id1.assign(B, -1);
    cp.ax1 = false;
    // walks on both parameters (blhip_chainax.hpp): an exact square geometry (the blocks of a chain change between column strips and row strips)
    // -- the next square geometry of 128 / 256 / 512 rows = columns that holds the grid (PAD kernels where it is larger)
    const int ax_n = std::max(g.n0, g.n1) <= 128 ? 128 : (std::max(g.n0, g.n1) <= 256 ? 256 : 512);
    const bool ax1_geom = cp.allow_ax1 && std::max(g.n0, g.n1) <= 512 && g.n0 >= 32 && g.n1 >= 32 && ax_n / blc::WCOL <= cus;
    std::vector<int> lw(B, 0);
    cp.ckF.assign((size_t)T * B, (unsigned char)SRC_PREV);
    cp.ckB.assign((size_t)T * B, (unsigned char)SRC_PREV);
    for (int64_t b = 0; b < B; ++b) {
        if (prog.kindF[b] != SRC_PRIOR || prog.tapF0[b] >= 0 || prog.tapF1[b] >= 0) return false;
        // the chain's band: the kernel of the first step that filters (every filtering step must use the same one)
        int k0 = -1, k1 = -1;
        for (int64_t t = 1; t < T && k0 < 0; ++t) k0 = prog.tapF0[(size_t)t * B + b];
        for (int64_t t = 0; t + 1 < T && k0 < 0 && full; ++t) k0 = prog.tapB0[(size_t)t * B + b];
        if (ax1_geom) {
            for (int64_t t = 1; t < T && k1 < 0; ++t) k1 = prog.tapF1[(size_t)t * B + b];
            for (int64_t t = 0; t + 1 < T && k1 < 0 && full; ++t) k1 = prog.tapB1[(size_t)t * B + b];
        }
        // a step either continues from the previous state through the chain's band, or RESTARTS from the reset distribution (a
        // change point, transitionModels.py:300-312) -- through the band (the change point comes before the random walk in the
        // combined model's list) or unfiltered (it comes after: the walk's output is discarded)
        auto classify = [&](unsigned char kind, int t0, int t1, unsigned char &out) {
            // (k1 = -1 unless the both-axes kernels may be planned; a restart passes through BOTH of the chain's bands or through none)
            if (kind == SRC_PREV && t0 == k0 && t1 == k1) { out = (unsigned char)SRC_PREV; return true; }
            if (kind == SRC_RESET && ((t0 == k0 && t1 == k1) || (t0 < 0 && t1 < 0))) {
                const bool filters = k0 >= 0 || k1 >= 0;
                out = (unsigned char)(SRC_RESET | ((filters && t0 < 0 && t1 < 0) ? 0x80 : 0));          // bit 7: no filter at this step
                cp.has_reset = true;
                if (filters) cp.mixed = true;
                return true;
            }
            return false;
        };
        for (int64_t t = 1; t < T; ++t) {
            const size_t k = (size_t)t * B + b;
            if (!classify(prog.kindF[k], prog.tapF0[k], prog.tapF1[k], cp.ckF[k])) return false;
        }
        if (full) {
            const size_t kl = (size_t)(T - 1) * B + b;
            if (prog.kindB[kl] != SRC_UNIFORM || prog.tapB0[kl] >= 0 || prog.tapB1[kl] >= 0) return false;
            for (int64_t t = 0; t < T - 1; ++t) {
                const size_t k = (size_t)t * B + b;
                if (!classify(prog.kindB[k], prog.tapB0[k], prog.tapB1[k], cp.ckB[k])) return false;
            }
        }
        cp.tap_id[b] = k0;
        cp.tap_id1[b] = k1;
        lw[b] = k0 >= 0 ? taps.lw[k0] : 0;
        if (lw[b] > cp.r0_max || lw[b] >= g.n0) return false;          // (single-period reflection)
        if (k1 >= 0) {
            cp.ax1 = true;
            if (taps.lw[k1] >= g.n1) return false;
            lw[b] = std::max(lw[b], taps.lw[k1]);                       // (one ring length for both filters: the wider walk's)
        }
    }
    if (cp.ax1) {
        // the transposing kernels: bands of radius <= 40 on either axis (ring lengths 8 .. 24 in steps of 4; the band's rounded radius inside
        // the grid: single-period reflection); the square geometry replaces the strip geometry planned above
        for (int64_t b = 0; b < B; ++b) if (lw[b] > FAST_R0_MAX || (std::max(8, (lw[b] + 7) / 8 * 8)) >= std::min(g.n0, g.n1)) return false;
        cp.n0p = cp.n1p = ax_n;
        cp.strips = ax_n / blc::WCOL;
        cp.ntw = ax_n / (blc::NW * blc::TM);
        cp.pad = ax_n != g.n0 || ax_n != g.n1;
        cp.cpr = cus / cp.strips;
    }
    cp.order.resize(B);
    for (int64_t b = 0; b < B; ++b) cp.order[b] = (int)b;
    std::stable_sort(cp.order.begin(), cp.order.end(), [&](int a, int c) { return lw[a] < lw[c]; });
    cp.round_start.clear(); cp.round_nk.clear();
    for (int64_t s0 = 0; s0 < B; s0 += cp.cpr) {
        const int64_t s1 = std::min<int64_t>(B, s0 + cp.cpr);
        int r0 = std::max(4, (lw[cp.order[s1 - 1]] + 3) / 4 * 4);             // (a product costs 64 cycles: bands as narrow as the widest chain of the launch allows)
        if (cp.ax1) r0 = std::max(8, (r0 + 7) / 8 * 8);                        // (ring lengths 8, 12, .. 24)
        cp.round_start.push_back((int)s0);
        cp.round_nk.push_back((prog.LW0 == 0 && !cp.ax1) ? 4 : (blc::TM + 2 * r0) / 4);          // (4: the no-stencil kernel)
    }
    cp.round_start.push_back((int)B);
    return true;
}
