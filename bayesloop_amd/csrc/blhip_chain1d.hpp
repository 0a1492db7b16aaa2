// CHAIN-RESIDENT pass for batches of chains on 1-D grids: ONE block per chain runs all T time steps of the forward (or the backward)
// pass with the chain's state in the LDS of its CU.
//
// Why: the fits the reference's tutorials are made of are 1-D -- a Poisson rate on 200 .. 1000 grid points, tens to thousands of chains
// of a hyper- or change-point study (HyperStudy.fit core.py:1349-1366, ChangepointStudy.fit :1765-1852, each chain = Study.fit
// :372-411 / :434-470) -- and a 1-D distribution is a few KB.  The K-steps-per-launch kernels (blhip_fused1d.hpp) cut a row into blocks of
// 128 cells that recompute K lw halo cells per side; with the wide walks such studies use (coal mining: sigma / delta = 6.7 on 200
// points, 33 on 1000: radius 133) K drops to 1 and a pass is T launches of ~6 us each; the persistent variant (blhip_persist1d.hpp) needs
// all blocks of all chains on the chip at once (<= 256 blocks).  A chain's whole row (<= 4096 cells here) fits one CU's LDS with any
// halo, so: block = chain, the time loop runs inside the kernel, ONE barrier per step, no hand-off between blocks at all (the
// normaliser is a block sum), any number of chains (blocks beyond the chip's capacity simply queue), any radius < n.
//
// Same arithmetic and the same conventions as the launch-per-step kernels with K = 1 (bl1f::advance_cells: centre first, pairs from
// the outside in, four interleaved accumulators; lazy normaliser = 1 / sum of the previous state; partial-sum slots 0 N, 1 sum p / L,
// 2 sum c, 3 mean; one slot per chain and sum: nblk = 1), so the host's bookkeeping is that of a K = 1 pass.
#pragma once
#include "blhip_fused1d.hpp"

namespace bl1c {

using blk::NRED;
using blk::SRC_PREV;
constexpr int NT = 512;            // 8 waves
constexpr int NW = NT / 64;
constexpr int NMAX = 4096;         // cells per row: at most 8 per thread (the stored forward row of the next step waits in registers)
constexpr int CPT = NMAX / NT;

constexpr int NS = 6;              // sums of a step per wave: N, sum p / L, sum c, mean, mass of the shifted / clamped distribution, maximum of the new state (clamps)
// doubles of LDS: two state buffers with halo, grid values, exp(-lambda) (Poisson), weights, the waves' partial sums (two parities);
// shift: + the other half of an asymmetric tap set and the spline coefficients of the 12-padded row (two-stage shifts)
inline size_t lds_doubles(int n, int LW, bool shift = false) {
    return (size_t)2 * (n + 1 + 2 * (LW + 1)) + 2 * (size_t)n + (LW + 3) + 2 * NW * NS + 8 + (shift ? (size_t)LW + 40 + n + 24 + blk::SPLINE_CONST_DOUBLES : 0);
}

// The likelihood of a 1-D batch is the same for every chain (same data, same grid: only the transition differs).  Poisson's pow() per
// cell and step was most of a short-radius step (n = 4000, radius 8: 11.6 us per step of which ~9 are the likelihood); evaluated ONCE
// per fit into a (T, n) table -- by the very function the step kernels call, so the values are bit for bit the in-kernel ones -- the
// chains read it like a tabulated model's (8 bytes per cell and step from L2: T n 8 bytes, e.g. 880 KB for 110 steps x 1000 cells).
template <int OM>
__global__ __launch_bounds__(256) void lik1d_table_kernel(const bl1f::F1Params P, double *__restrict__ out) {
    const int t = blockIdx.y;
    blk::StepParams Q{};
    Q.d = P.d; Q.n1 = P.n; Q.m0 = nullptr; Q.m1 = P.m1;
    Q.rec = P.rec + (long long)t * P.rec_len;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < P.n; j += gridDim.x * 256)
        out[(long long)t * P.n + j] = blk::likelihood<OM>(Q, 0, j, OM == blk::OM_POISSON ? P.colA[j] : 0.0, 0.0, P.m1[j]);
}

// M = 2: TWO ADJACENT cells per thread on a plane-interleaved state (even positions in one plane, odd ones in the other: the lanes of a
// wave still read consecutive 8-byte slots, conflict-free).  At large radii the M = 1 loop is bound by the LDS pipe -- every output reads
// both operands and the weight of every tap: 3 reads per output and tap.  Two adjacent outputs share their operands (the right operand of
// output 0 at tap k is the one of output 1 at tap k - 1, likewise on the left) and their weights: per PAIR of taps 4 operand reads + 2
// weight reads serve 4 output-taps, 1.5 reads per output and tap.  Used for rows of more than NT cells (below, one cell per thread keeps
// all waves busy).
// SHIFT: the batch's programs contain Deterministic steps (clamp mode 6; transitionModels.py:571-602): scipy.ndimage.shift(order = 3,
// mode = 'nearest') of the row, as in the launch-per-step kernel (blk::step_kernel, modes asym1 / big1) -- an asymmetric stencil of
// 2 lw + 1 weights over the extension SciPy works on (12 edge samples, then half-sample reflection: blk::extend_index rule 2) for
// |d| <= 12 cells, the two-stage form beyond (prefilter of the padded row, cubic B-spline at the shifted coordinates with the coefficient
// index clamped).  The reference renormalises the shifted distribution: its mass goes to the host as one more sum per step (slot 1
// forward, slot 5 backward, as the launch-per-step kernel's).  One cell per thread and pass over the row (M = 1).
// CL = 2: ... and programs with the clamps of RegimeSwitch (transitionModels.py:394-415: clamp from below at 10^pMin dV, renormalise -- on the
// step's source when the model stands in front of a walk, after the stencil when it follows one) and NotEqual (:450-474: invert around the
// maximum, renormalise, clamp, renormalise), with the launch-per-step kernel's arithmetic and sum slots (blk::step_kernel, clamp modes
// 1 / 2 / 3): the clamp acts on the NORMALISED distribution, so the backward pass of such a batch scales by 1 / sum(beta) of the producing
// step (slot 5) and carries kappa = sum(beta) / sum(c) into its products; the mass of the clamped distribution leaves as slot 1 (forward) /
// slot 5 (backward), the maximum of the new state (NotEqual inverts around it) as slot 6.  Block = chain: every one of these sums is a
// block sum of the previous step -- no hand-off, which is what the 2-D resident kernels would need for it.  (CL = 1 keeps the arithmetic
// of the shift-only flavour bit for bit: the published break-point study's borderline chains, DESIGN 6 COAL_NOISE_CHAINS.)
template <int OM, bool BWD, int M = 1, int CL = 0>
__global__ __launch_bounds__(NT) void chain1d_kernel(const bl1f::F1Params P) {
    constexpr bool SHIFT = CL != 0, CLAMP = CL == 2;
    static_assert(!(CL == 1 && M != 1), "spline shifts: one cell per thread");      // (CL = 2 with M = 2: programs with clamps but WITHOUT Deterministic steps -- the host's choice)
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int n = P.n;
    const int LW = M == 2 ? (P.LW + 1) & ~1 : P.LW;                  // (M = 2: an even halo, so that cell 0 sits in plane 0)
    const int W = M == 2 ? ((n + 1) & ~1) + 2 * LW : n + 2 * LW, PS = W / 2;
    double *cur = lds, *nxt = lds + W, *g1s = lds + 2 * W, *cAs = g1s + n, *wl = cAs + n;
    double *red = wl + (LW + 2) + (SHIFT ? LW + 40 : 0);     // [2 parities][NW][NS] wave sums of a step
    double *vt = red + 2 * NW * NS + 8;                      // (SHIFT) [n + 24] spline coefficients of the padded row
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double *post = P.post ? P.post + (long long)b * P.post_stride : nullptr;
    blk::StepParams Q{};
    Q.d = P.d; Q.n1 = n; Q.m0 = nullptr; Q.m1 = P.m1;

    for (int j = tid; j < n; j += NT) {
        g1s[j] = P.m1[j];
        if (OM == blk::OM_POISSON) cAs[j] = P.colA[j];
    }
    // position e of the extended row (e = LW + cell index) -> slot of the state buffer
    auto slot = [&](int e) { return M == 2 ? (e & 1) * PS + (e >> 1) : e; };
    // a row -> the state buffer incl. its mirror image beyond both ends (half-sample reflection, radius < n: one period)
    auto put = [&](double *buf, int j, double v) {
        buf[slot(LW + j)] = v;
        if (j < LW) buf[slot(LW - 1 - j)] = v;
        if (j >= n - LW) buf[slot(LW + n + (n - 1 - j))] = v;
    };
    int tap_now = -2, lw = 0;
    if (SHIFT && wv == 0) blk::spline_consts(vt + n + 24, n + 24, lane);      // (read by wave 0 only: no barrier)
    const int t0 = P.t_first;
    // backward: the stored forward row of the step that runs next waits in registers (requested a step ahead)
    double al[CPT];
    auto cell_of = [&](int q) { return M == 2 ? 2 * (tid + (q >> 1) * NT) + (q & 1) : tid + q * NT; };      // the q-th cell of this thread
    if (BWD) {
#pragma unroll
        for (int q = 0; q < CPT; ++q) { const int j = cell_of(q); al[q] = j < n ? post[(long long)t0 * n + j] : 0.0; }
    }
    __syncthreads();

    for (int s = 0; s < P.T; ++s) {
        const int t = t0 + P.dir * s;
        const long long tb = (long long)t * P.B + b;
        const int kind = P.srckind[tb];
        const int tp = P.tap[tb];
        double *rk = red + (s & 1) * (NW * NS), *rp = red + ((s + 1) & 1) * (NW * NS);
        const int l2 = (SHIFT && M == 1 && tp >= 0 && P.cmode[tb] == 6) ? P.tap_lw2[tp] : 0;
        const bool asym = SHIFT && l2 == -1, big = SHIFT && l2 == -2, shift = asym || big;
        // ---- the step's weights (block-uniform: re-staged only when the chain's tap set changes) ---------------------------------------
        const bool staged = tp != tap_now || kind != SRC_PREV || s == 0;
        if (tp != tap_now) {
            lw = tp >= 0 ? P.tap_lw[tp] : 0;
            if (shift) {                             // asymmetric: w[k + lw], k = -lw .. lw; two-stage: [d, g(0) .. g(34)]
                const int nw = asym ? 2 * lw + 1 : 36;
                for (int k = tid; k < nw; k += NT) wl[k] = P.taps[P.tap_off[tp] + k];
            } else {
                for (int k = tid; k <= LW + 1; k += NT) wl[k] = k <= lw ? (lw > 0 ? P.taps[P.tap_off[tp] + k] : 1.0) : 0.0;
            }
            tap_now = tp;
        }
        // ---- the source: the previous state (in `cur` since the last barrier), or a shared distribution (prior, restart, uniform) -------
        double scale = 1.0, kappa = 1.0;
        const int cm = CLAMP ? (int)P.cmode[tb] : 0;                 // 1 / 2: RegimeSwitch before / after the stencil, 3: NotEqual (0, 6: none here)
        const double lim = (CLAMP && cm >= 1 && cm <= 3) ? P.limit[tb] : 0.0;
        double ne_max = 0.0, ne_inv = 0.0;
        if (kind != SRC_PREV || s == 0) {
            const double *src = (kind == SRC_PREV) ? P.src + (long long)b * P.src_stride : P.shared[kind];
            for (int j = tid; j < n; j += NT) put(cur, j, src[j]);
        } else {
            // lazy normaliser: every thread adds the waves' sums of the previous step in the same order
            double sN = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) sN += rp[w * NS + (BWD ? 2 : 0)];
            scale = 1.0 / sN;
            if (CLAMP && cm == 3) {                                   // (max - x) / (n max - sum x), transitionModels.py:465-466
                double mx = -1.0;
#pragma unroll
                for (int w = 0; w < NW; ++w) mx = fmax(mx, rp[w * NS + 5]);
                ne_max = mx;
                ne_inv = 1.0 / ((double)n * mx - sN);
            }
            if (CLAMP && BWD) {                                       // (blk::step_kernel: "RegimeSwitch clamps F(beta_norm * L)")
                double sb = 0.0;
#pragma unroll
                for (int w = 0; w < NW; ++w) sb += rp[w * NS + 4];
                kappa = sb * scale;
                scale = 1.0 / sb;
            }
        }
        // (block-uniform: a barrier only where something was staged -- all reads of the buffer this step overwrites ended before the
        //  previous step's last barrier)
        if (staged) __syncthreads();
        if (CLAMP && (cm == 1 || cm == 3)) {
            // a clamp on the step's SOURCE: the whole buffer, mirror image included (the mirror of the clamped row is the clamped mirror)
            for (int e = tid; e < W; e += NT) {
                double v = cur[e];
                v = cm == 1 ? v * scale : (ne_max - v) * ne_inv;
                cur[e] = v < lim ? lim : v;
            }
            scale = 1.0;
            if (cm == 3) kappa = 1.0;                                 // (NotEqual is scale-invariant in its input)
            __syncthreads();
        }
        if (shift) {
            // the halo of a shifted row follows SciPy's extension, not the mirror image put() leaves there (sources: cells of the row)
            for (int h = tid; h < 2 * lw; h += NT) {
                const int i = h < lw ? h - lw : n + (h - lw);
                cur[LW + i] = cur[LW + blk::extend_index(i, n, 2)];
            }
            __syncthreads();
            if (big) {                               // spline coefficients of the padded row (position q = grid coordinate q - 12, the 12
                for (int q = tid; q < n + 24; q += NT)   // cells beyond either end repeat the edge sample) by SciPy's recursion
                    vt[q] = blk::SPLINE_GAIN * cur[LW + min(max(q - 12, 0), n - 1)];
                __syncthreads();
                if (wv == 0) blk::spline_prefilter_wave(vt, n + 24, lane, blk::spline_k_load(vt + n + 24, lane));
                __syncthreads();
            }
        }
        Q.rec = P.rec + (long long)t * P.rec_len;
        Q.lik = P.lik ? P.lik + (long long)t * n : nullptr;
        double *row = post ? post + (long long)t * n : nullptr;
        double aN = 0.0, aS = 0.0, aC = 0.0, aM = 0.0, aU = 0.0, aX = 0.0;
        // the epilogue of one cell: o = the transition's output (scaled)
        auto finish = [&](int j, int q, double o) {
            const double g1 = g1s[j];
            const double L = blk::likelihood<OM>(Q, 0, j, OM == blk::OM_POISSON ? cAs[j] : 0.0, 0.0, g1);
            if (CLAMP) {
                if (cm == 2) o = o < lim ? lim : o;      // RegimeSwitch after the stencil
                // mass of the clamped distribution (the reference renormalises by it, transitionModels.py:410); steps with a shift add theirs below
                if (cm != 6) aU += (cm == 1 || cm == 3) ? cur[slot(LW + j)] : o;
                o *= kappa;                              // backward: beta_used
            }
            if (!BWD) {
                const double a = o * L;
                put(nxt, j, a);
                if (P.store) row[j] = a;
                aN += a;
                if (CLAMP) aX = fmax(aX, a);
                if (P.means) aM = fma(a, g1, aM);
            } else {
                const double cn = o * L, p = al[q] * o;
                put(nxt, j, cn);
                row[j] = p;
                aN += p; aS += p / L; aC += cn;          // 0/0 -> NaN as numpy (core.py:463)
                if (CLAMP) aX = fmax(aX, cn);
                aM = fma(p, g1, aM);
            }
        };
        if (M == 2) {
            const double *X0 = cur, *X1 = cur + PS;
            const int A = (lw + 1) >> 1;                 // pairs of taps (the weight beyond the radius is zero)
#pragma unroll
            for (int g = 0; g < CPT / 2; ++g) {
                const int j0 = 2 * (tid + g * NT);
                if (j0 < n) {
                    const int s0 = (LW + j0) >> 1;
                    // four accumulators: even / odd taps of both outputs (independent chains)
                    double e0 = X0[s0] * wl[0], e1 = X1[s0] * wl[0], d0 = 0.0, d1 = 0.0;
                    double Rc = X1[s0 + A], Lc = X0[s0 - A];                 // x[e + 1 + 2A], x[e - 2A]
                    for (int a = A; a >= 1; --a) {
                        const double R0 = X0[s0 + a], L1 = X1[s0 - a], w2 = wl[2 * a], w1 = wl[2 * a - 1];
                        e0 = fma(Lc + R0, w2, e0);                           // output 0, tap 2a:     x[e - 2a] + x[e + 2a]
                        e1 = fma(L1 + Rc, w2, e1);                           // output 1, tap 2a:     x[e + 1 - 2a] + x[e + 1 + 2a]
                        const double Rn = X1[s0 + a - 1], Ln = X0[s0 - a + 1];
                        d0 = fma(L1 + Rn, w1, d0);                           // output 0, tap 2a - 1: x[e - 2a + 1] + x[e + 2a - 1]
                        d1 = fma(Ln + R0, w1, d1);                           // output 1, tap 2a - 1: x[e - 2a + 2] + x[e + 2a]
                        Rc = Rn; Lc = Ln;
                    }
                    finish(j0, 2 * g, (e0 + d0) * scale);
                    if (j0 + 1 < n) finish(j0 + 1, 2 * g + 1, (e1 + d1) * scale);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                const int j = tid + q * NT;
                if (j < n) {
                    const int e = LW + j;
                    if (SHIFT && shift) {
                        double o = 0.0;
                        if (big) {                   // cubic B-spline at the shifted coordinate of the padded array, index clamped
                            const int N = n + 24;
                            const double pp = (double)j - wl[0] + 12.0, fl = floor(pp);
                            const int k0 = (int)fmax(fmin(fl, 1.0e9), -1.0e9);
                            for (int dk = -1; dk <= 2; ++dk) {
                                const double a = fabs(pp - (fl + (double)dk));
                                const double b3 = a < 1.0 ? 2.0 / 3.0 - a * a + a * a * a * 0.5 : (a < 2.0 ? (2.0 - a) * (2.0 - a) * (2.0 - a) / 6.0 : 0.0);
                                const long long kk = (long long)k0 + dk;
                                o = fma(b3, vt[kk < 0 ? 0 : (kk > N - 1 ? N - 1 : (int)kk)], o);
                            }
                        } else {                     // out[i] = sum_m w[m + lw] in[i + m]
                            for (int k = -lw; k <= lw; ++k) o = fma(wl[k + lw], cur[e + k], o);
                        }
                        const double u = o * scale;
                        aU += u;
                        finish(j, q, u);
                        continue;
                    }
                    double o0 = cur[e] * wl[0], o1 = 0.0, o2 = 0.0, o3 = 0.0;
                    int k = lw;
                    for (; k >= 4; k -= 4) {
                        o0 = fma(cur[e - k] + cur[e + k], wl[k], o0);
                        o1 = fma(cur[e - k + 1] + cur[e + k - 1], wl[k - 1], o1);
                        o2 = fma(cur[e - k + 2] + cur[e + k - 2], wl[k - 2], o2);
                        o3 = fma(cur[e - k + 3] + cur[e + k - 3], wl[k - 3], o3);
                    }
                    for (; k >= 1; --k) o0 = fma(cur[e - k] + cur[e + k], wl[k], o0);
                    finish(j, q, ((o0 + o1) + (o2 + o3)) * scale);
                }
            }
        }
        if (BWD && s + 1 < P.T) {                     // the stored row of the next step: a whole step to arrive
#pragma unroll
            for (int q = 0; q < CPT; ++q) { const int j = cell_of(q); al[q] = j < n ? post[(long long)(t - 1) * n + j] : 0.0; }
        }
        // ---- the step's sums: waves -> LDS (this step's parity); the totals go to the host in a fixed order --------------------------
        aN = blk::wave_sum(aN);
        if (BWD) { aS = blk::wave_sum(aS); aC = blk::wave_sum(aC); }
        if (BWD || P.means) aM = blk::wave_sum(aM);
        if (SHIFT) aU = blk::wave_sum(aU);
        if (CLAMP) {
            aU *= kappa;                                  // (backward: sum of beta_used, what the next step normalises by)
            for (int off = 32; off >= 1; off >>= 1) aX = fmax(aX, __shfl_xor(aX, off));
        }
        if (lane == 0) { rk[wv * NS + 0] = aN; rk[wv * NS + 1] = aS; rk[wv * NS + 2] = aC; rk[wv * NS + 3] = aM; if (SHIFT) rk[wv * NS + 4] = aU; if (CLAMP) rk[wv * NS + 5] = aX; }
        __syncthreads();                              // `nxt` and the sums are complete
        if (tid < 4 && (tid == 0 || BWD || (tid == 3 && P.means))) {
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += rk[w * NS + tid];
            P.psum[(tb * NRED + tid) * P.nblk] = tot;
        }
        if (SHIFT && tid == 4) {                      // the shifted distribution's mass (steps without a shift: unused by the host)
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += rk[w * NS + 4];
            P.psum[(tb * NRED + (BWD ? 5 : 1)) * P.nblk] = tot;
        }
        if (CLAMP && tid == 5) {                      // the maximum of the new state (NotEqual inverts around it; carried states)
            double mx = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) mx = fmax(mx, rk[w * NS + 5]);
            P.psum[(tb * NRED + 6) * P.nblk] = mx;
        }
        double *tmp = cur; cur = nxt; nxt = tmp;
    }
    if (P.dst) {
        double *d = P.dst + (long long)b * P.dst_stride;
        for (int j = tid; j < n; j += NT) d[j] = cur[slot(LW + j)];
    }
}

}  // namespace bl1c
