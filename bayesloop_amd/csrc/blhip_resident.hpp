// TIME-RESIDENT step kernel for single-chain 2-D fits (gfx950): ONE launch runs all T time steps of the forward (or the
// backward) pass; the state never leaves the chip.
//
// Why: a plain Study.fit on a 1024^2 .. 2048^2 grid is one chain of T dependent steps over a state of 8 .. 32 MiB.  With one
// launch per step (blf:: / blm:: kernels) a step pays a kernel boundary (~2 us), a block prologue (~3.5 us: dependent scalar
// loads, bands, the normaliser reduction) and streams the state through L2 / Infinity Cache twice -- 12 .. 25 us per step for
// 3 .. 10 us of work (profiles/r01_notes.md).  The whole state fits the LDS of the chip (256 CUs x 160 KiB = 40 MiB), so here
//   * the grid is cut into tr x tc tiles (<= one per CU), block = tile, and the tile lives in LDS across ALL time steps;
//   * a step is two in-place 1-D passes over the tile: axis 1 (a thread owns a ROW segment and walks along columns), then
//     axis 0 (a thread owns a COLUMN segment and walks along rows) with the fused epilogue (lazy normaliser, Gaussian
//     likelihood recurrence along the rows, new state, sums, optional posterior store).  A pass keeps a sliding window of
//     2 R + 8 values in registers and writes its outputs over its own inputs; the 8 + 8 halo values that belong to a
//     neighbouring segment of the same tile are read into registers before anyone writes (one barrier);
//   * tiles exchange only halos, through HBM-side strips: the raw edge COLUMNS of the new state (consumed by the horizontal
//     neighbours' axis-1 pass of the next step) and the axis-1-filtered edge ROWS (consumed by the vertical neighbours'
//     axis-0 pass of the same step).  Point-to-point, no grid-wide barrier: write-through (sc1) payload stores, every
//     storing wave drains, one lane publishes a monotonic epoch flag; the consumer polls that one word, then reads the strip
//     with sc1 loads (cdna_hip_programming.md Guideline 16, form R1).  Edge segments walk TOWARDS the tile edge, so the
//     neighbour's strip is needed only for the last chunk of a pass and the hand-off latency hides under the pass;
//   * the lazy normaliser needs a GLOBAL sum.  A step is linear in its input, so the scale may lag: step k divides by the sum
//     of step k - LAG (LAG = 2 by default), which every tile published LAG steps earlier as two tagged 8-byte granules; the
//     host undoes the lag when it forms the per-step normalisers (blhip.hip: resident_unlag).  No tile ever waits for a sum.
// The axis-1 pass runs BEFORE the axis-0 pass (the reference filters axis 0 first, transitionModels.py:645-649): separable
// reflect-boundary filters commute exactly in real arithmetic; in floating point the results differ by rounding (~1e-16).
//
// Scope: Gaussian observation model with the likelihood recurrence (equally spaced row axis), one chain, every transition a
// GaussianRandomWalk of radius <= 8 per axis (or none), no change-points, grid divisible into the tile shapes below.
// Everything else keeps the launch-per-step kernels.  All tiles must be co-resident (grid <= number of CUs, one block per
// CU by LDS size); every spin is bounded and a time-out makes the host fall back to the launch-per-step path.
//
// Algorithmic HBM traffic per cell and step: forward evidence-only 0 B (halos only: 4 x 8 x tile edge), with posterior
// storage 8 B (write), backward 16 B (read alpha, write posterior).  bench.py still prices the forward step at the 16 B
// of the streaming formulation, i.e. `achieved` may exceed what HBM could deliver.
//
// This header also compiles on the HOST (-DBLR_EMULATE, g++): tools/emu/resident_emu.cpp runs the per-thread phase
// functions below sequentially to check the index / halo / publish logic against a direct evaluation (development aid).
#pragma once
#include "blhip_expmn.hpp"

#ifdef BLR_EMULATE
#include <cassert>
#include <cstdint>
#include <cstring>
#define BLR_INL inline
#else
#include <hip/hip_runtime.h>
#define BLR_INL __device__ __forceinline__
#endif

namespace blr {

constexpr int R = 8;             // stencil radius bucket of both axes (weights zero-padded)
constexpr int CHK = 8;           // outputs per chunk of a pass (== R: the window is 3 chunks)
constexpr int NSLOT = 8;         // ring of granule slots for the lagged sums (>= 2 * max lag)
constexpr int MAXLAG = 4;
constexpr int DMAX = 4;          // data dimensions kept in registers
constexpr int NRED = 7;          // == blk::NRED: partial-sum slots per step
constexpr int ANCHOR = 16;       // rows between exact re-anchorings of the likelihood recurrence

struct ResParams {
    int n0, n1, tr, tc, ntiles;
    int T, d, rec_len, lag;
    int store;                   // forward: write every step's state to post (full / forward-only fits)
    int means;                   // forward: also sum a * grid values (forward-only fits)
    const double *src0;          // what the first step consumes instead of a transition: prior (forward) / uniform (backward)
    double *post;                // [T][n0 * n1]: forward: stored states (out); backward: stored states (in) -> posteriors (out)
    const double *w0, *w1;       // R + 1 half-kernel weights per axis, zero-padded; {1, 0, ...} = no filter
    const double *m0, *m1, *colA, *colB, *rec;
    double step0;
    double *psum;                // [T][NRED][ntiles]
    double *cols;                // [2][ntiles][2][R][TR]   raw edge columns of the new state (parity = step & 1)
    double *rows;                // [2][ntiles][2][R][TC]   axis-1-filtered edge rows
    unsigned *flagC, *flagR;     // [ntiles] epochs
    unsigned long long *gran;    // [NSLOT][ntiles][2] {tag << 32 | half of the double}
    unsigned *abort_word;
    unsigned long long timeout_ticks;   // wall_clock64 ticks (100 MHz)
};

// ---- memory primitives: agent-scope (sc1) accesses on the device, plain ones in the emulation -------------------------------
#ifdef BLR_EMULATE
BLR_INL void st_sc1(double *p, double v) { *p = v; }
BLR_INL double ld_sc1(const double *p) { return *p; }
BLR_INL void st_u64(unsigned long long *p, unsigned long long v) { *p = v; }
BLR_INL unsigned long long ld_u64(const unsigned long long *p) { return *p; }
BLR_INL void st_flag(unsigned *p, unsigned v) { *p = v; }
BLR_INL unsigned ld_flag(const unsigned *p) { return *p; }
BLR_INL void drain() {}
BLR_INL unsigned long long now_ticks() { return 0; }
BLR_INL void nap() {}
BLR_INL double ldexp_(double m, int n) { return std::ldexp(m, n); }
BLR_INL double nan_() { return std::nan(""); }
BLR_INL double ldu(const double *p, long long i) { return p[i]; }
#else
typedef unsigned long long __attribute__((address_space(1))) gu64;
typedef unsigned __attribute__((address_space(1))) gu32;
BLR_INL void st_sc1(double *p, double v) {
    __hip_atomic_store((gu64 *)(unsigned long long)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
BLR_INL double ld_sc1(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load((gu64 *)(unsigned long long)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
BLR_INL void st_u64(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store((gu64 *)(unsigned long long)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
BLR_INL unsigned long long ld_u64(const unsigned long long *p) {
    return __hip_atomic_load((gu64 *)(unsigned long long)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
BLR_INL void st_flag(unsigned *p, unsigned v) { __hip_atomic_store((gu32 *)(unsigned long long)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
BLR_INL unsigned ld_flag(const unsigned *p) { return __hip_atomic_load((gu32 *)(unsigned long long)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
BLR_INL void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
BLR_INL unsigned long long now_ticks() { return wall_clock64(); }
BLR_INL void nap() { __builtin_amdgcn_s_sleep(2); }
BLR_INL double ldexp_(double m, int n) { return ldexp(m, n); }
BLR_INL double nan_() { return __builtin_nan(""); }
// wave-uniform read-only values (stencil weights, the step's data record) through the scalar cache: SGPRs, not VGPRs
BLR_INL double ldu(const double *p, long long i) { return ((const double __attribute__((address_space(4))) *)(unsigned long long)p)[i]; }
#endif

// bounded wait for an epoch flag; false = timed out / another block gave up (the caller marks the block dead)
BLR_INL bool wait_ge(const unsigned *flag, unsigned epoch, const ResParams &P) {
#ifdef BLR_EMULATE
    assert(*flag >= epoch && "hand-off protocol: consumed before published");
    (void)P;
    return true;
#else
    if (ld_flag(flag) >= epoch) return true;
    const unsigned long long t0 = now_ticks();
    for (unsigned spins = 1;; ++spins) {
        nap();
        if (ld_flag(flag) >= epoch) return true;
        if ((spins & 255u) == 0u) {
            if (ld_flag(P.abort_word) != 0u) return false;
            if (now_ticks() - t0 > P.timeout_ticks) { st_flag(P.abort_word, 1u); return false; }
        }
    }
#endif
}

// block id -> tile: spatially adjacent tiles on the same XCD where the dispatcher deals blocks round-robin (b % 8); a pure
// performance hint (same-XCD hand-offs are ~1.7x faster), any placement is correct
BLR_INL int tile_of_block(int b, int ntiles) {
    return (ntiles % 8 == 0) ? (b % 8) * (ntiles / 8) + b / 8 : b;
}

// One thread's in-place pass over its SEG-element line segment.  Positions p = 0 .. SEG-1 in WALKING order live at
// x0[p * STRIDE] (compile-time stride, negative = towards lower addresses: every access is base + immediate offset);
// nearv[k] = x(k - 8), the far halo x(SEG + k) comes from far_fetch(f) (registers read before the barrier, or the neighbour
// tile's strip).  pre8(p0) may start loads the epilogue of the chunk needs; emit8(p0, v) receives the 8 filtered values of
// positions p0 .. p0+7 and may overwrite x(p0 .. p0+7): the window already holds what later chunks need.
template <int SEG, int STRIDE, class FarFn, class Pre, class Emit>
BLR_INL void walk(const double *x0, const double (&nearv)[R], const double (&wk)[R + 1], FarFn &&far_fetch, Pre &&pre8, Emit &&emit8) {
    static_assert(SEG % CHK == 0 && SEG >= CHK && R == CHK, "segment = whole chunks; window = 3 chunks");
    double w[3 * CHK];
#pragma unroll
    for (int k = 0; k < CHK; ++k) w[k] = nearv[k];
#pragma unroll
    for (int k = 0; k < CHK; ++k) w[CHK + k] = x0[k * STRIDE];
    if (SEG > CHK) {
#pragma unroll
        for (int k = 0; k < CHK; ++k) w[2 * CHK + k] = x0[(CHK + k) * STRIDE];
    } else {
        double f[R];
        far_fetch(f);
#pragma unroll
        for (int k = 0; k < CHK; ++k) w[2 * CHK + k] = f[k];
    }
#pragma unroll
    for (int p0 = 0; p0 < SEG; p0 += CHK) {
        const bool more = p0 + CHK < SEG;
        double nx[CHK];
        if (more) {                                  // what enters the window for the next chunk: x(p0+16 .. p0+23)
            if (p0 + 2 * CHK < SEG) {
#pragma unroll
                for (int k = 0; k < CHK; ++k) nx[k] = x0[(p0 + 2 * CHK + k) * STRIDE];
            } else {
                far_fetch(nx);
            }
        }
        pre8(p0);                                    // (loads the epilogue of THIS chunk needs: they fly under the arithmetic below)
        double v[CHK];
#pragma unroll
        for (int j = 0; j < CHK; ++j) v[j] = w[CHK + j] * wk[0];
#pragma unroll
        for (int k = R; k >= 1; --k) {               // outermost pair inwards, the 8 outputs interleaved (independent chains)
            double t[CHK];
#pragma unroll
            for (int j = 0; j < CHK; ++j) t[j] = w[CHK + j - k] + w[CHK + j + k];
#pragma unroll
            for (int j = 0; j < CHK; ++j) v[j] = fma(t[j], wk[k], v[j]);
        }
        emit8(p0, v);
        if (more) {
#pragma unroll
            for (int k = 0; k < 2 * CHK; ++k) w[k] = w[k + CHK];
#pragma unroll
            for (int k = 0; k < CHK; ++k) w[2 * CHK + k] = nx[k];
        }
    }
}

// keep the optimiser from hoisting a thread's (time-invariant) address arithmetic out of the time loop: hoisted, the
// addresses of every row / column a thread touches stay live across the whole step and spill (598 spilled VGPRs measured)
BLR_INL int launder(int x) {
#ifndef BLR_EMULATE
    asm volatile("" : "+v"(x));
#endif
    return x;
}

template <int TR_, int TC_, int SEG_, bool BWD_>
struct Res {
    static constexpr int TR = TR_, TC = TC_, SEG = SEG_;
    static constexpr bool BWD = BWD_;
    static constexpr int P = TC + 1;                 // LDS pitch in doubles: odd => the row-strided accesses of the axis-1 pass
                                                     // and the contiguous ones of the axis-0 pass are both conflict-free
    static constexpr int NSH = TC / SEG, NSV = TR / SEG;
    static constexpr int NT = TR * NSH;
    static_assert(TR * NSH == TC * NSV, "both passes use every thread");
    static_assert(NSH >= 2 && NSV >= 2, "edge segments need an in-tile neighbour segment on their near side");
    static_assert(TR % SEG == 0 && TC % SEG == 0 && TR >= 2 * R && TC >= 2 * R, "tile shape");
    static constexpr int LDS_TILE = TR * P;          // doubles
    static constexpr int LDS_M0 = LDS_TILE;          // TR row coordinates of the tile
    static constexpr int LDS_MISC = LDS_M0 + TR;     // [0] scale of this step  [1] dead flag  [2] arrival counter  [8..] reduction scratch
    static constexpr int LDS_DOUBLES = LDS_MISC + 8 + 5 * (NT / 64 + 1) + 8;

    struct Rec { double mE, mR, mq, iE, iR, iq; int nE, nR, nq; };

    struct Thread {
        // geometry (time-invariant)
        int tid, tile, ti, tj, i0, j0, tr, tc;
        int hr, hs, hfar, hnb, hside;                // axis-1 pass: row, segment, far-halo kind (0 LDS neighbour segment, 1 mirror,
        int vc, vs, vfar, vnb, vside;                //   2 neighbour tile's strip), which tile / side;  axis-0 pass: column, ...
        double *lds;
        double g1, cA, cB;                           // column constants of the axis-0 pass / epilogue
        // registers that live across a barrier
        double nearv[R], farv[R];
        double xd[DMAX];                             // this step's data record (wave-uniform)
        double al8[BWD ? CHK : 1];                   // backward: the stored forward state of the chunk being processed
        double sums[5];
        bool dead;

        BLR_INL void init(const ResParams &Q, int block, int tid_, double *lds_) {
            tid = tid_; lds = lds_; dead = false;
            tr = Q.tr; tc = Q.tc;
            tile = tile_of_block(block, Q.ntiles);
            ti = tile / Q.tc; tj = tile - ti * Q.tc;
            i0 = ti * TR; j0 = tj * TC;
            hr = tid % TR; hs = tid / TR;
            hfar = 0; hnb = tile; hside = 0;
            if (hs == 0) { if (tj > 0) { hfar = 2; hnb = tile - 1; hside = 1; } else hfar = 1; }
            else if (hs == NSH - 1) { if (tj < Q.tc - 1) { hfar = 2; hnb = tile + 1; hside = 0; } else hfar = 1; }
            vc = tid % TC; vs = tid / TC;
            vfar = 0; vnb = tile; vside = 0;
            if (vs == 0) { if (ti > 0) { vfar = 2; vnb = tile - Q.tc; vside = 1; } else vfar = 1; }
            else if (vs == NSV - 1) { if (ti < Q.tr - 1) { vfar = 2; vnb = tile + Q.tc; vside = 0; } else vfar = 1; }
            const int gj = j0 + vc;
            g1 = Q.m1[gj]; cA = Q.colA[gj]; cB = Q.colB[gj];
#pragma unroll
            for (int k = 0; k < DMAX; ++k) xd[k] = nan_();
        }

        // time index of the k-th executed step
        BLR_INL static int time_of(const ResParams &Q, int k) { return BWD ? Q.T - 1 - k : k; }
        // segment s walks towards lower indices when it is the first one (its far side is then the tile's low edge)
        BLR_INL static constexpr int first_pos(int s, int dir) { return dir < 0 ? SEG - 1 : s * SEG; }

        // ---- start of a step: this step's data record ---------------------------------------------------------------------------
        BLR_INL void begin_step(const ResParams &Q, int k) {
            const int t = time_of(Q, k);
#pragma unroll
            for (int q = 0; q < DMAX; ++q) xd[q] = q < Q.d ? ldu(Q.rec, (long long)t * Q.rec_len + q) : nan_();
#pragma unroll
            for (int q = 0; q < 5; ++q) sums[q] = 0.0;
        }

        // ---- axis-1 pass ---------------------------------------------------------------------------------------------------------
        template <int DIR>
        BLR_INL void h_preread_d() {
            const double *x0 = lds + launder(hr) * P + first_pos(hs, DIR);
#pragma unroll
            for (int k = 0; k < R; ++k) nearv[k] = x0[DIR * (k - R)];
            if (hfar == 0) {
#pragma unroll
                for (int k = 0; k < R; ++k) farv[k] = x0[DIR * (SEG + k)];
            } else if (hfar == 1) {                  // grid edge: half-sample mirror = the segment's own last values
#pragma unroll
                for (int k = 0; k < R; ++k) farv[k] = x0[DIR * (SEG - 1 - k)];
            }
        }
        BLR_INL void h_preread() { if (hs == 0) h_preread_d<-1>(); else h_preread_d<1>(); }

        template <int DIR>
        BLR_INL void h_walk_d(const ResParams &Q, int k) {
            double *x0 = lds + launder(hr) * P + first_pos(hs, DIR);
            auto far_fetch = [&](double (&f)[R]) {
                if (hfar == 2) {
                    if (!wait_ge(Q.flagC + hnb, (unsigned)k, Q)) dead = true;
                    const double *s = Q.cols + (((long long)((k - 1) & 1) * Q.ntiles + hnb) * 2 + hside) * R * TR + hr;
#pragma unroll
                    for (int q = 0; q < R; ++q) f[q] = ld_sc1(s + (long long)(hside == 1 ? R - 1 - q : q) * TR);
                } else {
#pragma unroll
                    for (int q = 0; q < R; ++q) f[q] = farv[q];
                }
            };
            auto emit8 = [&](int p0, const double (&v)[CHK]) {
#pragma unroll
                for (int j = 0; j < CHK; ++j) x0[DIR * (p0 + j)] = v[j];
            };
            double wk[R + 1];
#pragma unroll
            for (int q = 0; q <= R; ++q) wk[q] = ldu(Q.w1, q);
            walk<SEG, DIR>(x0, nearv, wk, far_fetch, [](int) {}, emit8);
        }
        BLR_INL void h_walk(const ResParams &Q, int k) { if (hs == 0) h_walk_d<-1>(Q, k); else h_walk_d<1>(Q, k); }

        // after the axis-1 pass (barrier): the tile's filtered edge rows -> strips of step k, coalesced, every thread takes part
        BLR_INL void publish_rows(const ResParams &Q, int k) {
            double *base = Q.rows + ((long long)(k & 1) * Q.ntiles + tile) * 2 * R * TC;
            for (int idx = tid; idx < 2 * R * TC; idx += NT) {
                const int side = idx / (R * TC), rem = idx - side * (R * TC), rr = rem / TC, col = rem - rr * TC;
                if (side == 0 ? ti > 0 : ti < tr - 1)          // (my top rows are the up neighbour's lower halo)
                    st_sc1(base + idx, lds[(side ? TR - R + rr : rr) * P + col]);
            }
        }
        // after the axis-0 pass + epilogue (barrier): the new state's edge columns -> strips of step k  [side][cc][row]
        BLR_INL void publish_cols(const ResParams &Q, int k) {
            double *base = Q.cols + ((long long)(k & 1) * Q.ntiles + tile) * 2 * R * TR;
            for (int idx = tid; idx < 2 * R * TR; idx += NT) {
                const int side = idx / (R * TR), rem = idx - side * (R * TR), cc = rem / TR, row = rem - cc * TR;
                if (side == 0 ? tj > 0 : tj < tc - 1)
                    st_sc1(base + idx, lds[row * P + (side ? TC - R + cc : cc)]);
            }
        }

        // ---- axis-0 pass + epilogue ----------------------------------------------------------------------------------------------
        template <int DIR>
        BLR_INL void v_preread_d() {
            const double *x0 = lds + first_pos(vs, DIR) * P + launder(vc);
#pragma unroll
            for (int k = 0; k < R; ++k) nearv[k] = x0[DIR * (k - R) * P];
            if (vfar == 0) {
#pragma unroll
                for (int k = 0; k < R; ++k) farv[k] = x0[DIR * (SEG + k) * P];
            } else if (vfar == 1) {
#pragma unroll
                for (int k = 0; k < R; ++k) farv[k] = x0[DIR * (SEG - 1 - k) * P];
            }
        }
        BLR_INL void v_preread() { if (vs == 0) v_preread_d<-1>(); else v_preread_d<1>(); }

        // backward: the stored forward state alpha_t of positions p0 .. p0+7 (read before the posterior overwrites it in place)
        template <int DIR>
        BLR_INL void load_alpha8(const double *pt0, long long n1, int p0) {
            if (BWD) {
#pragma unroll
                for (int j = 0; j < CHK; ++j) al8[j] = pt0[(long long)(DIR * (p0 + j)) * n1];
            }
        }

        // the epilogue of positions p0 .. p0+7 of this thread's column segment: v = transition output (unscaled).
        // x0 / m0p / pt0 point at position 0 of the segment in the LDS tile / the tile's row coordinates / the global row
        template <int DIR>
        BLR_INL void epilogue8(const ResParams &Q, double *x0, const double *m0p, double *pt0, int p0, const double (&v)[CHK],
                               double scale, Rec &rc) {
            if (p0 % ANCHOR == 0) {
                // arg(r) = sum_q [-(x_q - mu_r)^2 cA - cB]  (observationModels.py:566-567; product over dimensions :49-50), along the
                // walking direction: arg(1) - arg(0) = cA (mu_1 - mu_0) sum_q (2 x_q - mu_0 - mu_1); second difference = -2 cA dn step^2
                const double mu0 = m0p[DIR * p0], mu1 = m0p[DIR * (p0 + 1)];
                double a0 = 0.0, s1 = 0.0, dn = 0.0;
#pragma unroll
                for (int q = 0; q < DMAX; ++q) {
                    const double x = xd[q];
                    if (x == x) {
                        const double dq = x - mu0;
                        a0 = fma(-(dq * dq), cA, a0) - cB;
                        s1 += (x - mu0) + (x - mu1);
                        dn += 1.0;
                    }
                }
                const double d1 = cA * (mu1 - mu0) * s1;
                const double d2 = -2.0 * cA * dn * Q.step0 * Q.step0;
                blmath::exp_mn(a0, rc.mE, rc.nE);
                blmath::exp_mn(d1, rc.mR, rc.nR);
                blmath::exp_mn(d2, rc.mq, rc.nq);
                if (BWD) {
                    int tmp;
                    blmath::exp_mn(-a0, rc.iE, tmp);
                    blmath::exp_mn(-d1, rc.iR, tmp);
                    blmath::exp_mn(-d2, rc.iq, tmp);
                }
            }
#pragma unroll
            for (int j = 0; j < CHK; ++j) {
                const int p = p0 + j;
                const double Lv = ldexp_(rc.mE, rc.nE);
                double keep;                                         // what becomes the tile's new state
                if (!BWD) {
                    const double a = v[j] * scale * Lv;
                    keep = a;
                    if (Q.store) pt0[(long long)(DIR * p) * Q.n1] = a;
                    sums[0] += a;
                    if (Q.means) { sums[3] = fma(a, m0p[DIR * p], sums[3]); sums[4] = fma(a, g1, sums[4]); }
                } else {
                    const double beta = v[j] * scale;
                    const double pp = al8[j] * beta;
                    const double cn = beta * Lv;
                    // p / L: reciprocal recurrence (no division, no intermediate overflow); 0/0 -> NaN (core.py:463)
                    const double pl = Lv == 0.0 ? nan_() : ldexp_(pp * rc.iE, -rc.nE);
                    keep = cn;
                    pt0[(long long)(DIR * p) * Q.n1] = pp;
                    sums[0] += pp; sums[1] += pl; sums[2] += cn;
                    sums[3] = fma(pp, m0p[DIR * p], sums[3]); sums[4] = fma(pp, g1, sums[4]);
                }
                x0[DIR * p * P] = keep;
                rc.mE *= rc.mR; rc.nE += rc.nR;
                rc.mR *= rc.mq; rc.nR += rc.nq;
                if (BWD) { rc.iE *= rc.iR; rc.iR *= rc.iq; }
            }
        }

        template <int DIR>
        BLR_INL void v_walk_d(const ResParams &Q, int k) {
            const double scale = lds[LDS_MISC];
            Rec rc{1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0, 0, 0};
            const int r0 = first_pos(vs, DIR), c = launder(vc);
            double *x0 = lds + r0 * P + c;
            const double *m0p = lds + LDS_M0 + r0;
            double *pt0 = Q.post ? Q.post + (long long)time_of(Q, k) * Q.n0 * Q.n1 + (long long)(i0 + r0) * Q.n1 + (j0 + c) : nullptr;
            auto far_fetch = [&](double (&f)[R]) {
                if (vfar == 2) {
                    if (!wait_ge(Q.flagR + vnb, (unsigned)k, Q)) dead = true;
                    const double *s = Q.rows + (((long long)(k & 1) * Q.ntiles + vnb) * 2 + vside) * R * TC + vc;
#pragma unroll
                    for (int q = 0; q < R; ++q) f[q] = ld_sc1(s + (long long)(vside == 1 ? R - 1 - q : q) * TC);
                } else {
#pragma unroll
                    for (int q = 0; q < R; ++q) f[q] = farv[q];
                }
            };
            auto pre8 = [&](int p0) { load_alpha8<DIR>(pt0, Q.n1, p0); };
            auto emit8 = [&](int p0, const double (&v)[CHK]) { epilogue8<DIR>(Q, x0, m0p, pt0, p0, v, scale, rc); };
            double wk[R + 1];
#pragma unroll
            for (int q = 0; q <= R; ++q) wk[q] = ldu(Q.w0, q);
            walk<SEG, DIR * P>(x0, nearv, wk, far_fetch, pre8, emit8);
        }
        BLR_INL void v_walk(const ResParams &Q, int k) { if (vs == 0) v_walk_d<-1>(Q, k); else v_walk_d<1>(Q, k); }

        // the first executed step has no transition: its input is src0 (prior / uniform), scale 1
        template <int DIR>
        BLR_INL void first_step_d(const ResParams &Q) {
            Rec rc{1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0, 0, 0};
            const int r0 = first_pos(vs, DIR), c = launder(vc);
            double *x0 = lds + r0 * P + c;
            const double *m0p = lds + LDS_M0 + r0;
            const long long g0 = (long long)(i0 + r0) * Q.n1 + (j0 + c);
            double *pt0 = Q.post ? Q.post + (long long)time_of(Q, 0) * Q.n0 * Q.n1 + g0 : nullptr;
            const double *s = Q.src0 + g0;
#pragma unroll 1
            for (int p0 = 0; p0 < SEG; p0 += CHK) {
                double v[CHK];
#pragma unroll
                for (int j = 0; j < CHK; ++j) v[j] = s[(long long)(DIR * (p0 + j)) * Q.n1];
                load_alpha8<DIR>(pt0, Q.n1, p0);
                epilogue8<DIR>(Q, x0, m0p, pt0, p0, v, 1.0, rc);
            }
        }
        BLR_INL void first_step(const ResParams &Q) { if (vs == 0) first_step_d<-1>(Q); else first_step_d<1>(Q); }
    };

    // ---- the lagged global sum: publish / gather (one granule pair per tile and step) ----------------------------------------
    BLR_INL static void publish_sum(const ResParams &Q, int tile, int k, double value) {
        unsigned long long bits;
#ifdef BLR_EMULATE
        std::memcpy(&bits, &value, 8);
#else
        bits = (unsigned long long)__double_as_longlong(value);
#endif
        const unsigned long long tag = (unsigned long long)(unsigned)(k + 1) << 32;
        unsigned long long *g = Q.gran + ((long long)(k % NSLOT) * Q.ntiles + tile) * 2;
        st_u64(g, tag | (bits & 0xffffffffull));
        st_u64(g + 1, tag | (bits >> 32));
    }

    // sum of the tiles' partials of step ks, tiles lane, lane + nl, ... in this order (nl lanes cooperate); false = timed out
    BLR_INL static bool gather_partial(const ResParams &Q, int ks, int lane, int nl, double &out) {
        const unsigned long long want = (unsigned long long)(unsigned)(ks + 1);
        double acc = 0.0;
        bool ok = true;
        for (int idx = lane; idx < Q.ntiles; idx += nl) {
            const unsigned long long *g = Q.gran + ((long long)(ks % NSLOT) * Q.ntiles + idx) * 2;
            unsigned long long a = ld_u64(g), b = ld_u64(g + 1);
#ifdef BLR_EMULATE
            assert((a >> 32) == want && (b >> 32) == want && "lagged sum consumed before published");
#else
            if ((a >> 32) != want || (b >> 32) != want) {
                const unsigned long long t0 = now_ticks();
                for (unsigned spins = 1; ok; ++spins) {
                    nap();
                    a = ld_u64(g); b = ld_u64(g + 1);
                    if ((a >> 32) == want && (b >> 32) == want) break;
                    if ((spins & 255u) == 0u) {
                        if (ld_flag(Q.abort_word) != 0u) ok = false;
                        else if (now_ticks() - t0 > Q.timeout_ticks) { st_flag(Q.abort_word, 1u); ok = false; }
                    }
                }
            }
#endif
            const unsigned long long bits = (a & 0xffffffffull) | (b << 32);
            double v;
#ifdef BLR_EMULATE
            std::memcpy(&v, &bits, 8);
#else
            v = __longlong_as_double((long long)bits);
#endif
            acc += v;
        }
        out = acc;
        return ok;
    }
};

}  // namespace blr

#ifndef BLR_EMULATE
#include "blhip_kernels.hpp"

namespace blr {

// every wave has drained its strip stores -> the LAST wave to arrive publishes the epoch flag (no block barrier, no wave waits
// for another one's store acknowledgements)
template <int NW>
__device__ __forceinline__ void arrive_and_flag(double *misc, unsigned *flag, unsigned epoch) {
    drain();
    if ((threadIdx.x & 63) == 0) {
        unsigned *cnt = reinterpret_cast<unsigned *>(misc + 2);
        const unsigned old = atomicAdd(cnt, 1u);      // LDS atomic
        if (old == NW - 1) { *cnt = 0u; st_flag(flag, epoch); }
    }
}

template <int TR, int TC, int SEG, bool BWD>
__global__ __launch_bounds__(TR *TC / SEG) void resident_kernel(const ResParams Q) {
    using K = Res<TR, TC, SEG, BWD>;
    constexpr int NT = K::NT, NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *misc = lds + K::LDS_MISC;
    double *red = misc + 8;
    const int tid = threadIdx.x;
    typename K::Thread th;
    th.init(Q, blockIdx.x, tid, lds);
    for (int e = tid; e < TR; e += NT) lds[K::LDS_M0 + e] = Q.m0[th.i0 + e];
    if (tid == 0) { misc[0] = 1.0; misc[1] = 0.0; misc[2] = 0.0; }
    __syncthreads();

    for (int k = 0; k < Q.T; ++k) {
        const int t = K::Thread::time_of(Q, k);
        th.begin_step(Q, k);
        if (k == 0) {
            th.first_step(Q);
        } else {
            th.h_preread();
            __syncthreads();
            th.h_walk(Q, k);
            __syncthreads();
            th.publish_rows(Q, k);
            th.v_preread();
            if (tid < 64) {                           // wave 0: the scale of this step from the sums of step k - lag
                double s = 1.0;
                if (k >= Q.lag) {
                    double part;
                    if (!K::gather_partial(Q, k - Q.lag, tid, 64, part)) th.dead = true;
                    s = 1.0 / blk::wave_sum(part);
                }
                if (tid == 0) misc[0] = s;
            }
            arrive_and_flag<NW>(misc, Q.flagR + th.tile, (unsigned)k);
            __syncthreads();
            th.v_walk(Q, k);
        }
        // ---- sums of the step: partials for the host, the scale sum for the other tiles ----------------------------------------
        if (th.dead) misc[1] = 1.0;
        double *out = Q.psum + (long long)t * NRED * Q.ntiles + th.tile;
        if (BWD) {
            double v[5] = {th.sums[0], th.sums[1], th.sums[2], th.sums[3], th.sums[4]};
            blk::block_sums<5, NW>(v, red);
            if (tid == 0) {
#pragma unroll
                for (int q = 0; q < 5; ++q) out[(long long)q * Q.ntiles] = v[q];
                K::publish_sum(Q, th.tile, k, v[2]);
            }
        } else if (Q.means) {
            double v[3] = {th.sums[0], th.sums[3], th.sums[4]};
            blk::block_sums<3, NW>(v, red);
            if (tid == 0) {
                out[0] = v[0]; out[3LL * Q.ntiles] = v[1]; out[4LL * Q.ntiles] = v[2];
                K::publish_sum(Q, th.tile, k, v[0]);
            }
        } else {
            double v[1] = {th.sums[0]};
            blk::block_sums<1, NW>(v, red);
            if (tid == 0) { out[0] = v[0]; K::publish_sum(Q, th.tile, k, v[0]); }
        }
        // (block_sums ends with a barrier: the tile's new state is complete in LDS)
        th.publish_cols(Q, k);
        arrive_and_flag<NW>(misc, Q.flagC + th.tile, (unsigned)(k + 1));
        if (misc[1] != 0.0) return;                   // a wait timed out somewhere in this block: uniform exit (host falls back)
    }
}

}  // namespace blr
#endif
