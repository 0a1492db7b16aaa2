// Host-side bookkeeping of a pass in the reference's order (core.py:385-404, :417, :441-464, :480-483): undoing the resident kernels' lagged scales,
// evidence / abort / local evidence / means from the reduced sums; what the launches move and compute by construction.
// Part of libblhip's host side: included by blhip.hip INSIDE its anonymous namespace (one translation unit; the split is by subject, not by linkage).
#pragma once

// Undo the lagged scale of the time-resident kernel (blhip_resident.hpp): its step k divided by the sum of step k - lag, so its row
// sums are S_k; the reference's normaliser is norm_k = S_k / (S_{k-1} s_k), s_k = k >= lag ? 1 / S_{k-lag} : 1.  The sums of every step
// are rewritten to what the launch-per-step kernels (lag 1) would have reported; rowsum keeps S_k, the normaliser of the stored row.
// false: a sum near the bottom / top of the fp64 range (a run of extreme outliers times the lag) -> the caller falls back to the
// launch-per-step kernels, whose magnitudes are the reference's.
bool resident_unlag(double *redF, int64_t T, int lag, std::vector<double> &rowsum, int64_t B = 1, int64_t b = 0) {
    rowsum.assign(T, 0.0);
    for (int64_t t = 0; t < T; ++t) rowsum[t] = redF[((size_t)t * B + b) * NRED];
    for (int64_t t = 0; t < T; ++t) {
        const double St = rowsum[t];
        if (!(St > 1e-150 && St < 1e150)) return false;
        const double sk = t >= lag ? 1.0 / rowsum[t - lag] : 1.0;
        const double norm = t == 0 ? St : St / (rowsum[t - 1] * sk);
        double *r = &redF[((size_t)t * B + b) * NRED];
        r[0] = norm; r[3] *= norm / St; r[4] *= norm / St;
    }
    return true;
}

// The same for the chain-resident kernel (blhip_chainres.hpp), whose step k divides by the normaliser of step k - lag:
// s_k = S_(k-lag-1) s_(k-lag) / S_(k-lag)  (1 while k < lag; S_(-1) = 1).
// kinds (may be null): a step whose source kind is not SRC_PREV consumed a distribution of known mass instead of the previous
// state: its normaliser is S_k / s_k.
bool chain_unlag(double *redF, int64_t T, int lag, std::vector<double> &rowsum, int64_t B, int64_t b, std::vector<double> *scales = nullptr,
                 const unsigned char *kinds = nullptr, int64_t t0 = 0) {
    // t0 > 0 (blc::ChainParams::skip_prefix): the chain's own pass began at step t0 -- the rows before it are another chain's (same
    // values, that chain's scale history), the scale history of the rows from t0 on starts there (s = 1 for lag steps, S_(t0 - 1) := 1)
    rowsum.assign(T, 0.0);
    for (int64_t t = 0; t < T; ++t) rowsum[t] = redF[((size_t)t * B + b) * NRED];
    std::vector<double> s_local;
    std::vector<double> &s = scales ? *scales : s_local;
    s.assign(T, 1.0);
    for (int64_t t = 0; t < T; ++t) {
        const double St = rowsum[t];
        if (!(St > 1e-150 && St < 1e150)) return false;
        const int64_t base = t >= t0 ? t0 : 0;                                 // first step of the scale history this step belongs to
        if (t - base >= lag) s[t] = (t - lag - 1 >= base ? rowsum[t - lag - 1] : 1.0) * s[t - lag] / rowsum[t - lag];
        const bool fresh = t == 0 || (kinds && kinds[(size_t)t * B + b] != SRC_PREV);
        const double norm = fresh ? St / s[t] : St / (rowsum[t - 1] * s[t]);
        double *r = &redF[((size_t)t * B + b) * NRED];
        r[0] = norm; r[3] *= norm / St; r[4] *= norm / St;
    }
    return true;
}

// chain_unlag for every chain of a batch.  Chain by chain every record read is a cache line of its own (B x NRED doubles between the steps
// of a chain: 0.62 ms per batch of 256 chains x 256 steps, beside a GPU that waits for the backward pass's weights); here the records are
// read and rewritten step by step over all chains, the recurrence in between runs on the gathered rows.  Same operations per chain, same
// order: bit-identical to chain_unlag.  -> index of the first chain that failed, or -1.
int64_t chain_unlag_batch(double *redF, int64_t T, int lag, int64_t B, std::vector<std::vector<double>> &rowsum, std::vector<std::vector<double>> &scales,
                          const unsigned char *kinds, const int *t0s) {
    rowsum.assign(B, std::vector<double>());
    scales.assign(B, std::vector<double>());
    for (int64_t b = 0; b < B; ++b) { rowsum[b].resize(T); scales[b].assign(T, 1.0); }
    for (int64_t t = 0; t < T; ++t) {
        const double *rt = redF + (size_t)t * B * NRED;
        for (int64_t b = 0; b < B; ++b) rowsum[b][t] = rt[b * NRED];
    }
    std::vector<double> norms((size_t)B * T);
    for (int64_t b = 0; b < B; ++b) {
        const std::vector<double> &rs = rowsum[b];
        std::vector<double> &s = scales[b];
        const int64_t t0 = t0s ? t0s[b] : 0;
        for (int64_t t = 0; t < T; ++t) {
            const double St = rs[t];
            if (!(St > 1e-150 && St < 1e150)) return b;
            const int64_t base = t >= t0 ? t0 : 0;
            if (t - base >= lag) s[t] = (t - lag - 1 >= base ? rs[t - lag - 1] : 1.0) * s[t - lag] / rs[t - lag];
            const bool fresh = t == 0 || (kinds && kinds[(size_t)t * B + b] != SRC_PREV);
            norms[(size_t)b * T + t] = fresh ? St / s[t] : St / (rs[t - 1] * s[t]);
        }
    }
    for (int64_t t = 0; t < T; ++t) {
        double *rt = redF + (size_t)t * B * NRED;
        for (int64_t b = 0; b < B; ++b) {
            double *r = rt + b * NRED;
            const double norm = norms[(size_t)b * T + t], St = rowsum[b][t];
            r[0] = norm; r[3] *= norm / St; r[4] *= norm / St;
        }
    }
    return -1;
}

// evidence bookkeeping of the forward pass on the host, in the reference's order (core.py:385-404, 417); K > 1: raw sums of the
// K-steps-per-launch 1-D kernels.  -> false if such a raw sum came near the bottom of the fp64 range (the caller repeats with K = 1)
bool forward_bookkeeping(const blhip_problem *p, const ChainProgram &prog, const double *redF, int64_t B, double dV, bool fused1d, int64_t K,
                         bool evidence_only, bool forward_only, BatchOutcome &O) {
    const int64_t T = p->T;
    O.logE.assign(B, 0.0);
    O.abort_step.assign(B, -1);
    O.abort_phase.assign(B, 0);
    O.local.assign((size_t)B * T, 0.0);
    bool raw_ok = true;
    // (step by step over all chains: the sums of a step are B consecutive records -- chain by chain every read was a cache line of
    //  its own, 8 ms of the published break-point study's 23 batches of 1017 chains x 41 steps.  Per chain the order of the
    //  operations is the one of the reference's loop)
    std::vector<double> &le = O.logE;
    for (int64_t t = 0; t < T; ++t) {
        const double *rt = redF + (size_t)t * B * NRED;
        for (int64_t b = 0; b < B; ++b) {
            if (O.abort_step[b] >= 0) continue;
            double norm = rt[b * NRED + 0];
            if (fused1d) {
                // raw sums of the K-step launches (blhip_fused1d.hpp): inner steps carry the scale of their predecessor
                if (!(norm > 1e-200)) raw_ok = false;
                if (t % K != 0 && prog.kindF[(size_t)t * B + b] == SRC_PREV) norm /= redF[((size_t)(t - 1) * B + b) * NRED];
            }
            // RegimeSwitch renormalises the clamped prior (transitionModels.py:410): alpha = (u / sum u) L
            if (prog.has_clamp && prog.cmodeF[(size_t)t * B + b]) norm /= rt[b * NRED + 1];
            if (!(norm > 0.0)) { O.abort_step[b] = t; O.abort_phase[b] = 0; le[b] = -INFINITY; continue; }
            le[b] += std::log(norm);
            O.local[(size_t)b * T + t] = norm * dV;
        }
    }
    const double ldv = std::log(dV);
    for (int64_t b = 0; b < B; ++b)
        if (O.abort_step[b] < 0) le[b] += ldv;
    O.means.clear();
    if (!evidence_only) O.means.assign((size_t)B * p->ndim * T, 0.0);
    if (forward_only) {
        for (int64_t b = 0; b < B; ++b)
            for (int64_t t = 0; t < T; ++t) {
                const double *r = &redF[((size_t)t * B + b) * NRED];
                for (int k = 0; k < p->ndim; ++k) O.means[((size_t)b * p->ndim + k) * T + t] = r[3 + k] / r[0];
            }
    }
    return raw_ok;
}

// bookkeeping of the backward pass (core.py:441-464, 480-483): abort test, local evidence, row normalisers, posterior means.
// rows_done_from >= 0: rows t >= rows_done_from were normalised by the resident kernel itself (their invN is 1).
// want_invN / want_means = false: nobody reads the row normalisers (the backward kernel folded the batch's posteriors itself) / the
// per-chain means (the caller did not ask for them) -- three of the four divisions and of the scattered writes per (chain, step).
bool backward_bookkeeping(const blhip_problem *p, const ChainProgram &prog, const double *redF, const double *redB, int64_t B, double dV,
                          bool fused1d, int64_t rows_done_from, BatchOutcome &O, bool want_invN = true, bool want_means = true) {
    const int64_t T = p->T;
    bool raw_ok = true;
    // (step by step over all chains: see forward_bookkeeping.  Tiles of 32 chains -- rows of `local` / `invN` / `means` that stay in the first-level
    //  cache -- were measured and dropped: the records come out of page-locked memory the copy engine has just written, and 2-KB pieces 65 KB
    //  apart lose the prefetcher: the published break-point study, 23 batches of 1017 chains x 41 steps, 44 -> 49 ms per fit)
    for (int64_t t = T - 1; t >= 0; --t) {
        for (int64_t b = 0; b < B; ++b) {
            if (O.abort_step[b] >= 0) continue;
            const double *r = &redB[((size_t)t * B + b) * NRED];
            if (fused1d && !(r[0] > 1e-200)) raw_ok = false;
            // The reference tests sum(alpha_norm * beta_norm) > 0 (core.py:441).  r[0] is the same sum up to the lazily
            // dropped normalisers, which are positive -- except with signed kernels (Deterministic's cubic-spline
            // shift, AlphaStable's FFT kernel): there sum(alpha) = redF[t][0] and sum(beta) = r[5] may be negative and
            // the reference divides by them, so the sign test has to include them.
            double refnorm = r[0];
            if (prog.has_clamp) refnorm = r[0] / (redF[((size_t)t * B + b) * NRED] * (prog.cmodeB[(size_t)t * B + b] ? r[5] : 1.0));
            if (!(refnorm > 0.0)) { O.abort_step[b] = t; O.abort_phase[b] = 1; O.logE[b] = -INFINITY; continue; }
            O.local[(size_t)b * T + t] = 1.0 / ((r[1] / r[0]) * dV);                      // core.py:463-464
            if (want_invN) O.invN[(size_t)b * T + t] = (rows_done_from >= 0 && t >= rows_done_from) ? 1.0 : 1.0 / r[0];
            if (want_means) for (int k = 0; k < p->ndim; ++k) O.means[((size_t)b * p->ndim + k) * T + t] = r[3 + k] / r[0];
        }
    }
    return raw_ok;
}

// ---- what the launches move and compute BY CONSTRUCTION (blhip_timing: fwd / bwd _hbm_bytes, _flops) ------------------------------------
// fp64 flop per cell of the fused epilogues (FMA = 2; ldexp, compare and select count 1): forward  a = v L (1), sum (1), the two
// recurrence products (2), ldexp (1) + the exponentials of the anchors spread over their rows (2 x ~42 flop per 16 rows: 5);
// backward: beta, p, c (3), p / L by the reciprocal recurrence (2), three sums (3), the fold's product, max, add (3), four recurrence
// products (4), ldexp (1) + four exponentials per 16 rows (10)
constexpr double EPI_FWD_FLOP = 10.0, EPI_BWD_FLOP = 26.0;
// a radius-r stencil pass per cell: on the vector ALU (SciPy's pair order) r adds + 1 product + r FMAs; as a banded product on the
// matrix pipe (16 output rows per tile) 16 + 2 r products, zeros of the band included
inline double valu_stencil_flop(int r) { return r > 0 ? 3.0 * r + 1.0 : 0.0; }
inline double band_stencil_flop(int r) { return 2.0 * (16.0 + 2.0 * r); }
inline void account(blhip_ctx *ctx, bool bwd, double bytes, double flops) {
    (bwd ? ctx->timing.bwd_hbm_bytes : ctx->timing.fwd_hbm_bytes) += bytes;
    (bwd ? ctx->timing.bwd_flops : ctx->timing.fwd_flops) += flops;
}
