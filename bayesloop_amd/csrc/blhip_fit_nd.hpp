// Host side of the plain N-D path (kernels: blhip_nd.hpp).  Included by blhip.hip INSIDE its anonymous namespace, after the helpers
// it uses (FitFlags, TapTable, BatchOutcome, forward_ / backward_bookkeeping, fold_accumulate, keep_posterior, write_results).
#pragma once

// ---- grids with 3 and 4 parameters: the plain formulation (blhip_nd.hpp) ---------------------------------------------------------------
void do_fit_nd(blhip_ctx *ctx, const blhip_problem *p, int64_t n_chains, const double *op_values, const double *log_w, uint32_t flags,
               blhip_result *res) {
    HIPCHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const FitFlags ff = decode_flags(ctx, p, flags, log_w);
    // (streaming fits -- OnlineStudy.step, core.py:2062-2226, which has no limit on the grid's dimensions: step 0 consumes the chains'
    //  carried states through the transition evaluated at resume_time, the last step's filtered distributions are kept normalised)
    const int64_t T = p->T;
    bln::NdGrid ng{};
    ng.ndim = p->ndim;
    long long G = 1;
    for (int k = p->ndim - 1; k >= 0; --k) { ng.n[k] = (int)p->n[k]; ng.stride[k] = G; G *= p->n[k]; }
    ng.G = G;
    if (ff.accumulate && (ctx->acc_T != T || ctx->acc_G != G)) fail("accumulator shape mismatch");
    double dV = 1.0;
    for (int k = 0; k < p->ndim; ++k) dV *= p->lattice[k];
    ctx->post_valid = false;
    ctx->timing = blhip_timing{};
    hipEvent_t *ev = ctx->ev;
    HIPCHECK(hipEventRecord(ev[6], st));

    // ---- shared tables: marginals, prior(s), uniform, the likelihood table -------------------------------------------------------
    size_t msum = 0;
    for (int k = 0; k < p->ndim; ++k) msum += carve_size(8 * (size_t)p->n[k]);
    ctx->tables.ensure(msum + 3 * carve_size(8 * (size_t)G));
    char *cur = ctx->tables.as<char>();
    for (int k = 0; k < p->ndim; ++k) {
        double *dm = carve<double>(cur, (size_t)p->n[k]);
        HIPCHECK(hipMemcpyAsync(dm, p->marginal[k], 8 * (size_t)p->n[k], hipMemcpyHostToDevice, st));
        ng.m[k] = dm;
    }
    double *d_prior = carve<double>(cur, (size_t)G), *d_reset = carve<double>(cur, (size_t)G), *d_uniform = carve<double>(cur, (size_t)G);
    HIPCHECK(hipMemcpyAsync(d_prior, p->prior, 8 * (size_t)G, hipMemcpyHostToDevice, st));
    if (p->reset_prior) HIPCHECK(hipMemcpyAsync(d_reset, p->reset_prior, 8 * (size_t)G, hipMemcpyHostToDevice, st));
    if (p->backward_init) HIPCHECK(hipMemcpyAsync(d_uniform, p->backward_init, sizeof(double) * G, hipMemcpyHostToDevice, st));
    else BL_LAUNCH(fill_kernel, dim3(256), dim3(256), 0, st, d_uniform, G, 1.0 / (double)G);           // beta_T = 1/G, core.py:424-425
    ctx->likbuf.ensure(8 * (size_t)T * G);
    double *d_lik = ctx->likbuf.as<double>();
    HIPCHECK(hipMemcpyAsync(d_lik, p->lik, 8 * (size_t)T * G, hipMemcpyHostToDevice, st));
    sync_stream(ctx, st);

    // ---- batches ---------------------------------------------------------------------------------------------------------------------
    size_t free_b = 0, total_b = 0;
    HIPCHECK(hipMemGetInfo(&free_b, &total_b));
    const double budget = std::min((double)free_b + (double)ctx->state.cap + (double)ctx->post.cap, 0.70 * (double)total_b) * 0.9;
    const double per_chain = ((ff.evidence_only ? 0.0 : (double)T) + 3.0) * (double)G * 8.0 + (double)T * NRED * 8.0 * 2 * 256.0;
    int64_t Bmax = (int64_t)std::max(1.0, std::floor(budget / per_chain));
    Bmax = std::min<int64_t>(std::min<int64_t>(Bmax, (int64_t)ctx->option("max_batch", 1024)), 65535);
    if (ff.keep && n_chains > Bmax) fail("BLHIP_KEEP_POSTERIOR: %lld chains do not fit in device memory at once", (long long)n_chains);
    if ((ff.resume || ff.carry) && n_chains > Bmax) fail("carried states: %lld chains do not fit in one batch", (long long)n_chains);
    const double *d_carry_src = nullptr;
    if (ff.resume) {
        auto it = ctx->carry.find(p->carry_slot);
        if (it == ctx->carry.end() || !it->second.valid) fail("BLHIP_RESUME: carry slot %d holds no state", p->carry_slot);
        if (it->second.chains != n_chains || it->second.G != G)
            fail("BLHIP_RESUME: carry slot %d holds %lld chains x %lld cells, the call has %lld x %lld", p->carry_slot,
                 (long long)it->second.chains, (long long)it->second.G, (long long)n_chains, (long long)G);
        d_carry_src = it->second.buf.as<double>();
    }
    const int nblk = (int)std::min<long long>((G + NTHREADS - 1) / NTHREADS, 256);
    std::vector<int> grw_ops;                       // the random walks of the program, in list order (transitionModels.py:645-649)
    bool time_dependent = false;
    for (int k = 0; k < p->n_ops; ++k) {
        if (p->ops[k].kind == BLHIP_OP_GRW) grw_ops.push_back(k);
        if (p->ops[k].kind == BLHIP_OP_CHANGEPOINT) time_dependent = true;
    }
    const int npass = (int)grw_ops.size();
    ChainProgram no_clamp;                           // (the bookkeeping helpers only ask it for clamp modes)

    for (int64_t c0 = 0; c0 < n_chains; c0 += Bmax) {
        const int64_t B = std::min<int64_t>(Bmax, n_chains - c0);
        ctx->timing.batches += 1;
        ctx->timing.cells_per_launch = std::max<int64_t>(ctx->timing.cells_per_launch, B * G);
        ctx->timing.fwd_kernel_variant = ctx->timing.bwd_kernel_variant = 7;
        // ---- the program of every chain: source kind and the kernel of every pass, per step and direction ------------------------------
        TapTable taps;
        const size_t nT = (size_t)T * B;
        std::vector<unsigned char> kindF(nT, SRC_PREV), kindB(nT, SRC_PREV);
        std::vector<int> tapF((size_t)std::max(1, npass) * nT, -1), tapB((size_t)std::max(1, npass) * nT, -1);     // [pass][t][b]
        for (int64_t b = 0; b < B; ++b) {
            const double *val = op_values ? op_values + (c0 + b) * p->n_ops : nullptr;
            std::vector<int> op_tap(p->n_ops, -1);
            for (int k = 0; k < p->n_ops; ++k)
                if (p->ops[k].kind == BLHIP_OP_GRW) {
                    const double ns = val[k] / p->lattice[p->ops[k].axis];                    // transitionModels.py:108
                    if (std::isnan(ns)) fail("chain %lld: GRW sigma is NaN", (long long)(c0 + b));
                    op_tap[k] = ns > 0.0 ? taps.get(p->ops[k].axis, ns) : -1;                // :110-113
                }
            // the transition into a step, evaluated at time stamp tau (list order; a change point restarts from the reset
            // distribution and drops what the models before it did, transitionModels.py:300-312)
            auto run = [&](bool have_tau, double tau, unsigned char &kind, int *tp, size_t stride) {
                kind = SRC_PREV;
                for (int q = 0; q < npass; ++q) tp[q * stride] = -1;
                for (int k = 0; k < p->n_ops; ++k) {
                    const blhip_op &op = p->ops[k];
                    if (op.kind == BLHIP_OP_GRW) {
                        for (int q = 0; q < npass; ++q) if (grw_ops[q] == k) tp[q * stride] = op_tap[k];
                    } else if (op.kind == BLHIP_OP_CHANGEPOINT && have_tau && tau == val[k]) {
                        kind = SRC_RESET;
                        for (int q = 0; q < npass; ++q) tp[q * stride] = -1;
                    }
                }
            };
            for (int64_t t = 0; t < T; ++t) {
                const size_t k = (size_t)t * B + b;
                if (t == 0 && ff.resume) run(true, p->resume_time, kindF[k], &tapF[k], nT);       // continues a carried state (core.py:2164-2165)
                else if (t == 0) kindF[k] = SRC_PRIOR;                                  // core.py:363
                else run(time_dependent, time_dependent ? p->timestamps[t - 1] : 0.0, kindF[k], &tapF[k], nT);          // core.py:411
                if (t == T - 1) kindB[k] = SRC_UNIFORM;
                else run(time_dependent, time_dependent ? p->timestamps[t + 1] - 1.0 : 0.0, kindB[k], &tapB[k], nT);    // core.py:467, transitionModels.py:316-317
            }
        }
        taps.w.resize(taps.w.size() + 8, 0.0);
        // ---- device buffers ----------------------------------------------------------------------------------------------------------
        ctx->state.ensure((size_t)3 * B * G * 8);
        double *d_state = ctx->state.as<double>(), *d_tmp[2] = {d_state + (size_t)B * G, d_state + (size_t)2 * B * G};
        double *d_post = nullptr;
        if (!ff.evidence_only) { ctx->post.ensure((size_t)B * T * G * 8); d_post = ctx->post.as<double>(); }
        const size_t psz = (size_t)T * B * NRED * nblk;
        ctx->psumF.ensure(psz * 8); ctx->redF.ensure(nT * NRED * 8);
        if (ff.full) { ctx->psumB.ensure(psz * 8); ctx->redB.ensure(nT * NRED * 8); }
        const size_t ntap = taps.off.size() + 1;
        size_t mb = 2 * carve_size(nT) + 2 * carve_size(tapF.size() * 4) + carve_size(taps.w.size() * 8) + 2 * carve_size(ntap * 4) +
                    2 * carve_size(nT * 8) + 3 * carve_size((size_t)B * 8) + carve_size((size_t)B * 8) + carve_size(nT * 8);
        ctx->meta.ensure(mb);
        char *mc = ctx->meta.as<char>();
        unsigned char *d_kindF = carve<unsigned char>(mc, nT), *d_kindB = carve<unsigned char>(mc, nT);
        int *d_tapF = carve<int>(mc, tapF.size()), *d_tapB = carve<int>(mc, tapB.size());
        double *d_taps = carve<double>(mc, taps.w.size());
        int *d_off = carve<int>(mc, ntap), *d_lw = carve<int>(mc, ntap);
        const double **d_src0F = carve<const double *>(mc, nT), **d_src0B = carve<const double *>(mc, nT);
        const double **d_ptr_state = carve<const double *>(mc, (size_t)B), **d_ptr_tmp0 = carve<const double *>(mc, (size_t)B),
                     **d_ptr_tmp1 = carve<const double *>(mc, (size_t)B);
        double *d_w = carve<double>(mc, (size_t)B), *d_invN = carve<double>(mc, nT);
        // where a step's input lives: the chain's state, or a shared distribution at a restart
        std::vector<const double *> src0F(nT), src0B(nT), pst(B), pt0(B), pt1(B);
        for (int64_t b = 0; b < B; ++b) {
            pst[b] = d_state + (size_t)b * G; pt0[b] = d_tmp[0] + (size_t)b * G; pt1[b] = d_tmp[1] + (size_t)b * G;
            for (int64_t t = 0; t < T; ++t) {
                const size_t k = (size_t)t * B + b;
                src0F[k] = kindF[k] == SRC_PREV ? ((t == 0 && ff.resume) ? d_carry_src + (size_t)b * G : pst[b]) : (kindF[k] == SRC_PRIOR ? d_prior : d_reset);
                src0B[k] = kindB[k] == SRC_PREV ? pst[b] : (kindB[k] == SRC_UNIFORM ? d_uniform : d_reset);
            }
        }
        HIPCHECK(hipMemcpyAsync(d_kindF, kindF.data(), nT, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(d_kindB, kindB.data(), nT, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(d_tapF, tapF.data(), tapF.size() * 4, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(d_tapB, tapB.data(), tapB.size() * 4, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(d_taps, taps.w.data(), taps.w.size() * 8, hipMemcpyHostToDevice, st));
        if (!taps.off.empty()) {
            HIPCHECK(hipMemcpyAsync(d_off, taps.off.data(), taps.off.size() * 4, hipMemcpyHostToDevice, st));
            HIPCHECK(hipMemcpyAsync(d_lw, taps.lw.data(), taps.lw.size() * 4, hipMemcpyHostToDevice, st));
        }
        HIPCHECK(hipMemcpyAsync(d_src0F, src0F.data(), nT * 8, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(d_src0B, src0B.data(), nT * 8, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(d_ptr_state, pst.data(), (size_t)B * 8, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(d_ptr_tmp0, pt0.data(), (size_t)B * 8, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(d_ptr_tmp1, pt1.data(), (size_t)B * 8, hipMemcpyHostToDevice, st));
        sync_stream(ctx, st);

        // one time step: the passes of the transition, then the fused elementwise kernel
        auto step = [&](bool bwd, int64_t t, const double *ps_prev, double *ps_out) {
            const double *const *in = (bwd ? d_src0B : d_src0F) + (size_t)t * B;
            const int *tp = (bwd ? d_tapB : d_tapF) + (size_t)t * B;
            const std::vector<int> &htp = bwd ? tapB : tapF;
            int flip = 0;
            for (int q = 0; q < npass; ++q) {
                bool any = false;
                for (int64_t b = 0; b < B && !any; ++b) any = htp[(size_t)q * nT + (size_t)t * B + b] >= 0;
                if (!any) continue;
                const int ax = p->ops[grw_ops[q]].axis;
                BL_LAUNCH(bln::filter_axis_kernel, dim3((unsigned)nblk, (unsigned)B), dim3(NTHREADS), 0, st, d_tmp[flip], in, G, ng.n[ax],
                                   ng.stride[ax], tp + (size_t)q * nT, d_taps, d_off, d_lw);
                in = flip ? d_ptr_tmp1 : d_ptr_tmp0;
                flip ^= 1;
            }
            bln::NdStep Q{};
            Q.g = ng; Q.B = (int)B; Q.T = (int)T; Q.nblk = nblk; Q.srcs = in; Q.kind = (bwd ? d_kindB : d_kindF) + (size_t)t * B;
            Q.psum_prev = ps_prev; Q.prev_slot = bwd ? 2 : 0; Q.psum_out = ps_out; Q.lik = d_lik + (size_t)t * G; Q.state = d_state;
            Q.post = d_post ? d_post + (size_t)t * G : nullptr; Q.post_stride = (long long)T * G;
            if (bwd) BL_LAUNCH(bln::step_kernel<true>, dim3((unsigned)nblk, (unsigned)B), dim3(NTHREADS), 0, st, Q);
            else BL_LAUNCH(bln::step_kernel<false>, dim3((unsigned)nblk, (unsigned)B), dim3(NTHREADS), 0, st, Q);
        };
        double *d_psF = ctx->psumF.as<double>();
        const size_t per_step = (size_t)B * NRED * nblk;
        float ms = 0;
        // BLHIP_RESUME: the carried states are normalised -- the "partial sums of the step before" add up to 1
        const double *d_unit = nullptr;
        if (ff.resume) {
            std::vector<double> unit(per_step, 0.0);
            for (int64_t b = 0; b < B; ++b) unit[(size_t)b * NRED * nblk] = 1.0;
            ctx->unit.ensure(unit.size() * 8);
            HIPCHECK(hipMemcpyAsync(ctx->unit.p, unit.data(), unit.size() * 8, hipMemcpyHostToDevice, st));
            sync_stream(ctx, st);
            d_unit = ctx->unit.as<double>();
        }
        // ---- forward pass (core.py:372-411) ---------------------------------------------------------------------------------------------
        HIPCHECK(hipEventRecord(ev[0], st));
        for (int64_t t = 0; t < T; ++t) step(false, t, t > 0 ? d_psF + (size_t)(t - 1) * per_step : (d_unit ? d_unit : d_psF), d_psF + (size_t)t * per_step);
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipEventRecord(ev[1], st));
        BL_LAUNCH(reduce_partials_kernel, dim3((unsigned)(nT * NRED)), dim3(NTHREADS), 0, st, d_psF, ctx->redF.as<double>(), nblk, 0);      // (0: every slot is a sum -- slot 6 is the 4th parameter's mean here)
        ctx->pinF.ensure(nT * NRED * 8);
        double *redF = ctx->pinF.as<double>();
        HIPCHECK(hipMemcpyAsync(redF, ctx->redF.p, nT * NRED * 8, hipMemcpyDeviceToHost, st));
        sync_stream(ctx, st);
        HIPCHECK(hipEventElapsedTime(&ms, ev[0], ev[1]));
        ctx->timing.forward_ms += ms; ctx->timing.forward_launches += T;
        BatchOutcome O;
        forward_bookkeeping(p, no_clamp, redF, B, dV, false, 1, ff.evidence_only, ff.forward_only, O);
        O.invN.assign(nT, 0.0);
        // ---- backward pass (core.py:424-470) --------------------------------------------------------------------------------------------
        if (ff.full) {
            double *d_psB = ctx->psumB.as<double>();
            HIPCHECK(hipEventRecord(ev[2], st));
            for (int64_t t = T - 1; t >= 0; --t) step(true, t, t < T - 1 ? d_psB + (size_t)(t + 1) * per_step : d_psB, d_psB + (size_t)t * per_step);
            HIPCHECK(hipGetLastError());
            HIPCHECK(hipEventRecord(ev[3], st));
            BL_LAUNCH(reduce_partials_kernel, dim3((unsigned)(nT * NRED)), dim3(NTHREADS), 0, st, d_psB, ctx->redB.as<double>(), nblk, 0);
            ctx->pinB.ensure(nT * NRED * 8);
            double *redB = ctx->pinB.as<double>();
            HIPCHECK(hipMemcpyAsync(redB, ctx->redB.p, nT * NRED * 8, hipMemcpyDeviceToHost, st));
            sync_stream(ctx, st);
            HIPCHECK(hipEventElapsedTime(&ms, ev[2], ev[3]));
            ctx->timing.backward_ms += ms; ctx->timing.backward_launches += T;
            backward_bookkeeping(p, no_clamp, redF, redB, B, dV, false, -1, O);
        } else if (ff.forward_only) {
            for (int64_t b = 0; b < B; ++b)
                for (int64_t t = 0; t < T; ++t) {
                    const double n0 = redF[((size_t)t * B + b) * NRED];
                    O.invN[(size_t)b * T + t] = (n0 != 0.0 && std::isfinite(n0)) ? 1.0 / n0 : 0.0;
                }
        }
        // BLHIP_CARRY: every chain's filtered distribution of the last step, normalised (core.py:2173)
        if (ff.carry) store_carry(ctx, p, B, G, redF, d_post ? d_post + (size_t)(T - 1) * G : d_state, d_post ? (long long)T * G : G, d_w, false);
        if (ff.accumulate) fold_accumulate(ctx, T, G, B, O, log_w + c0, d_post, d_w, d_invN);
        if (ff.keep) {
            Geometry g2{};
            g2.n0 = (int)p->n[0]; g2.n1 = (int)(G / p->n[0]); g2.G = G;
            keep_posterior(ctx, g2, T, B, O, 0, T);
        }
        write_results(res, p, c0, B, O, !ff.evidence_only);
    }
    HIPCHECK(hipEventRecord(ev[7], st));
    HIPCHECK(hipEventSynchronize(ev[7]));
    float tot = 0;
    HIPCHECK(hipEventElapsedTime(&tot, ev[6], ev[7]));
    ctx->timing.total_ms = tot;
}
