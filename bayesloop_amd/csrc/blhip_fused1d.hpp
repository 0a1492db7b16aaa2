// Step kernels for 1-D grids (configs C1 / C2), K TIME STEPS PER LAUNCH.
//
// A 1-D distribution is a few KB; with one launch per time step a fit is pure launch latency (~6 us per step measured on
// MI355X for the 4096-cell grid of C2, of which the arithmetic is well under 1 us).  The recursion cannot skip the
// step-to-step dependency, but it does not need a GLOBAL synchronisation per step either:
//
//  * lazy normalisation makes a step LINEAR in its input: a'_t = T(a'_{t-1}) L_t without the division by sum(a_{t-1}) is
//    the true state times a scalar, and everything the host needs is a ratio of sums (norm_t = N'_t / N'_{t-1},
//    sum(p/L) / sum(p), means) or is normalised later by its own sum (stored posteriors);
//  * the stencil has a finite radius: a block that owns TJ cells of step t+K-1 only needs TJ + 2 K lw cells of step t-1.
//
// So a block loads its owned cells plus a halo of K*LW cells per side ONCE, advances K steps entirely in LDS (the valid
// range shrinks by LW per step; the halo is recomputed redundantly by the neighbouring blocks), stores the owned cells
// of every step (posterior rows, per-step partial sums) and the final state.  Beyond the grid edges the halo is the
// half-sample mirror image (SciPy 'reflect'), which the step operator maps to the mirror image of its output bit for
// bit (symmetric taps, commutative pair sums), so no special casing at the edges.  A step whose source is not the
// previous state (change-point restart, independent observations) reloads the window from the shared array.
// Unnormalised values decay by norm_t per step: the host watches the raw sums and re-runs a pass with K = 1 if a sum
// gets anywhere near the bottom of the fp64 range (never seen on the configs of BASELINE.json).
#pragma once
#include "blhip_kernels.hpp"

namespace bl1f {

using blk::NRED;
using blk::SRC_PREV;
constexpr int NT = 512;           // 8 waves: at most ~1 cell per thread per step, two waves per SIMD to hide fp64 latency

struct F1Params {
    int n, TJ, nblk, LW, K, dir, t_first, T, B, d, rec_len, store, means;
    const double *shared[5];
    const double *src; long long src_stride;       // state left by the previous launch (per chain), nullptr on the first
    double *dst; long long dst_stride;             // state after the last step of this launch
    double *post; long long post_stride;           // (B, T, n) stored rows or nullptr (evidence only)
    const unsigned char *srckind;                  // [T][B]
    const int *tap;                                // [T][B] tap-set id, -1 = identity
    const double *taps; const int *tap_off; const int *tap_lw;
    const double *psum_prev; int prev_slot;        // partial sums of the step before t_first (nullptr: scale 1)
    double *psum;                                  // [T][B][NRED][nblk]
    const double *m1, *colA, *rec, *lik;
    // chain-resident kernel with spline shifts (blhip_chain1d.hpp, SHIFT): clamp mode of every step (6 = Deterministic's shift) and the
    // layout marks of the tap sets (-1: all 2 lw + 1 weights, -2: two-stage form)
    const unsigned char *cmode; const int *tap_lw2;
    const double *limit;                           // ... CL = 2 (programs with RegimeSwitch / NotEqual clamps): [T][B] the clamp level of every step
    int no_shift;                                  // ... and none of its steps is a Deterministic shift: long rows may take two cells per thread (M = 2)
};

// One step over the cells of a block's window that are still exact after it (lo .. hi - 1): stencil out of LDS, likelihood, new state
// -> nxt; the owned cells store their row and park the terms of the step's sums in pt[3][TJ].  Shared with the persistent kernel
// (blhip_persist1d.hpp): the same instructions in the same order, so the two paths agree bit for bit.
//   wl: the step's weights [LW + 1], lw its radius; als_s: the stored forward row of the step's owned cells (backward only)
template <int OM, bool BWD>
__device__ __forceinline__ void advance_cells(const blk::StepParams &Q, int n, int j0, int halo, int tw, int TJ, int lo, int hi, const double *cur,
                                              double *nxt, const double *g1s, const double *cAs, const double *wl, int lw,
                                              const double *als_s, double *row, double *pt, bool store, int tid) {
    for (int e = lo + tid; e < hi; e += NT) {
        // four interleaved accumulators: a single fp64 FMA chain of 2 lw + 1 links costs ~32 cycles per link
        double o0 = cur[e] * wl[0], o1 = 0.0, o2 = 0.0, o3 = 0.0;
        int k = lw;
        for (; k >= 4; k -= 4) {
            o0 = fma(cur[e - k] + cur[e + k], wl[k], o0);
            o1 = fma(cur[e - k + 1] + cur[e + k - 1], wl[k - 1], o1);
            o2 = fma(cur[e - k + 2] + cur[e + k - 2], wl[k - 2], o2);
            o3 = fma(cur[e - k + 3] + cur[e + k - 3], wl[k - 3], o3);
        }
        for (; k >= 1; --k) o0 = fma(cur[e - k] + cur[e + k], wl[k], o0);
        const double o = (o0 + o1) + (o2 + o3);
        const double g1 = g1s[e];
        const double cA = (OM == blk::OM_POISSON) ? cAs[e] : 0.0;
        const int j = (OM == blk::OM_TABLE) ? blk::reflect(j0 - halo + e, n) : 0;
        const double L = blk::likelihood<OM>(Q, 0, j, cA, 0.0, g1);
        const int oc = e - halo;                               // owned cell index of this block (0 <= oc < tw)
        const bool owned = oc >= 0 && oc < tw;
        if (!BWD) {
            const double a = o * L;
            nxt[e] = a;
            if (owned) {
                if (store) row[j0 + oc] = a;
                pt[oc] = a;
            }
        } else {
            const double cn = o * L;
            nxt[e] = cn;
            if (owned) {
                const double p = als_s[oc] * o;
                row[j0 + oc] = p;
                pt[oc] = p;
                pt[TJ + oc] = p / L;                           // 0/0 -> NaN as numpy (core.py:463)
                pt[2 * TJ + oc] = cn;
            }
        }
    }
}

template <int OM, bool BWD>
__global__ __launch_bounds__(NT) void fused1d_kernel(const F1Params P) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int n = P.n, halo = P.K * P.LW, W = P.TJ + 2 * halo;
    double *cur = lds, *nxt = lds + W, *g1s = lds + 2 * W, *cAs = lds + 3 * W, *als = lds + 4 * W;
    double *wls = als + (BWD ? P.K * P.TJ : 0);                   // [K][LW + 1] stencil weights of the K steps
    double *recs = wls + P.K * (P.LW + 1);                        // [K][rec_len] data records of the K steps
    double *red = recs + P.K * P.rec_len;
    int *meta = (int *)(red + 4 * (NT / 64) + 2);                 // [K] source kind, [K] radius
    double *part = red + 4 * (NT / 64) + 2 + P.K;                 // [K][3][TJ] per-cell terms of the K steps' sums
    const int b = blockIdx.y, blkid = blockIdx.x, tid = threadIdx.x;
    const int j0 = blkid * P.TJ, tw = min(P.TJ, n - j0);
    double *post = P.post ? P.post + (long long)b * P.post_stride : nullptr;

    blk::StepParams Q{};
    Q.d = P.d; Q.n1 = n; Q.m0 = nullptr; Q.m1 = P.m1;

    // ---- window of the input state: cells j0 - halo .. j0 + TJ + halo (mirrored beyond the grid), lazily normalised;
    //      everything a step needs per cell comes from LDS afterwards: a dependent global load inside the step loop would
    //      put an HBM round trip (~2 us) on the critical path of every step
    {
        const int kind = P.srckind[(long long)P.t_first * P.B + b];
        double scale = 1.0;
        if (kind == SRC_PREV && P.psum_prev) {
            const double *pp = P.psum_prev + ((long long)b * NRED + P.prev_slot) * P.nblk;
            double v[1] = {0.0};
            for (int k = tid; k < P.nblk; k += NT) v[0] += pp[k];
            blk::block_sums<1, NT / 64>(v, red);
            scale = 1.0 / v[0];
        }
        const double *src = kind == SRC_PREV ? P.src + (long long)b * P.src_stride : P.shared[kind];
        for (int e = tid; e < W; e += NT) {
            const int j = blk::reflect(j0 - halo + e, n);
            cur[e] = src[j] * scale;
            g1s[e] = P.m1[j];
            if (OM == blk::OM_POISSON) cAs[e] = P.colA[j];
        }
        // per-step metadata, weights and data records of all K steps (the step loop must not chase pointers through HBM)
        for (int e = tid; e < P.K * (P.LW + 1); e += NT) {
            const int s = e / (P.LW + 1), k = e - s * (P.LW + 1);
            const long long tb = (long long)(P.t_first + P.dir * s) * P.B + b;
            const int tp = P.tap[tb];
            const int lw = tp >= 0 ? P.tap_lw[tp] : 0;
            wls[e] = k <= lw ? (lw > 0 ? P.taps[P.tap_off[tp] + k] : 1.0) : 0.0;
            if (k == 0) { meta[s] = P.srckind[tb]; meta[P.K + s] = lw; }
        }
        for (int e = tid; e < P.K * P.rec_len; e += NT) {
            const int s = e / P.rec_len, k = e - s * P.rec_len;
            recs[e] = P.rec[(long long)(P.t_first + P.dir * s) * P.rec_len + k];
        }
        if (BWD) {                                                 // stored forward rows of the K steps (owned cells)
            for (int e = tid; e < P.K * P.TJ; e += NT) {
                const int s = e / P.TJ, c = e - s * P.TJ;
                als[e] = c < tw ? post[(long long)(P.t_first + P.dir * s) * n + j0 + c] : 0.0;
            }
        }
    }

    for (int s = 0; s < P.K; ++s) {
        const int t = P.t_first + P.dir * s;
        __syncthreads();                                           // previous step's writes of nxt / reads of cur are done
        const int kind = meta[s], lw = meta[P.K + s];
        const double *wl = wls + s * (P.LW + 1);
        if (s > 0 && kind != SRC_PREV) {                           // restart from a shared distribution
            const double *src = P.shared[kind];
            for (int e = tid; e < W; e += NT) cur[e] = src[blk::reflect(j0 - halo + e, n)];
            __syncthreads();
        }
        Q.rec = recs + s * P.rec_len;
        Q.lik = P.lik ? P.lik + (long long)t * n : nullptr;
        double *row = post ? post + (long long)t * n : nullptr;

        // The step's sums (normaliser, p / L, next state, mean) are not on the critical path of the recursion: the owned
        // cells park their terms in LDS and all K steps are reduced after the loop.  ONE barrier per step (top of the loop).
        double *pt = part + (size_t)s * 3 * P.TJ;
        const int lo = (s + 1) * P.LW, hi = W - (s + 1) * P.LW;    // cells that are still exact after this step
        advance_cells<OM, BWD>(Q, n, j0, halo, tw, P.TJ, lo, hi, cur, nxt, g1s, cAs, wl, lw, als + s * P.TJ, row, pt, P.store != 0, tid);
        double *tmp = cur; cur = nxt; nxt = tmp;
    }
    __syncthreads();
    // ---- the sums of the K steps: wave w takes the (step, slot) pairs w, w + 8, ...; fixed order -> deterministic ----------
    for (int pr = tid >> 6; pr < 4 * P.K; pr += NT / 64) {
        const int s = pr >> 2, k = pr & 3, lane = tid & 63;
        if (!(k == 0 || BWD || (k == 3 && P.means))) continue;
        const double *pt = part + (size_t)s * 3 * P.TJ + (k == 3 ? 0 : k) * P.TJ;
        double acc = 0.0;
        if (k == 3) { for (int c = lane; c < tw; c += 64) acc = fma(pt[c], g1s[halo + c], acc); }
        else { for (int c = lane; c < tw; c += 64) acc += pt[c]; }
        acc = blk::wave_sum(acc);
        if (lane == 0) {
            const long long tb = (long long)(P.t_first + P.dir * s) * P.B + b;
            P.psum[(tb * NRED + k) * P.nblk + blkid] = acc;
        }
    }
    if (P.dst) {
        double *d = P.dst + (long long)b * P.dst_stride + j0;
        for (int c = tid; c < tw; c += NT) d[c] = cur[halo + c];
    }
}

}  // namespace bl1f
