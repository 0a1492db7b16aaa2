// Step kernels for 2-D grids on the fp64 MATRIX pipe (gfx950): the Gaussian-random-walk stencils as banded Toeplitz
// products with v_mfma_f64_16x16x4_f64.
//
// Why: with a stencil radius of 8..40 rows a cell costs 17..81 fp64 FMAs.  On the vector ALU one wave issues a dependent-
// free fp64 instruction every ~9 cycles and the 2*R0+8-row register window of blf::fast_step_kernel leaves 1-2 waves per
// SIMD, so those launches run at ~30 % of the fp64 peak and far below the HBM roof (profiles/r01_notes.md).  The same
// stencil is a banded Toeplitz product  OUT(16 x 16) = W(16 x K) * X(K x 16),  K = 16 + 2*R0, and
// v_mfma_f64_16x16x4_f64 sustains ~75 TFLOP/s = the whole fp64 pipe from ONE wave per SIMD with a single dependent
// accumulator chain (tools/ubench/mfma_f64_rate.hip).  The band wastes 16 of the K products per output; the issue
// efficiency more than pays for it.  On gfx950 the f64 matrix op runs on the same fp64 lanes as the vector ALU and blocks
// ALL other VALU issue of its SIMD while it executes (same microbenchmark): kernel time ~ MFMA cycles + every other VALU
// instruction, so the loop body is kept lean.  (This is fp64 compute-bound work; nothing is "reshaped into a GEMM" to
// dodge HBM: a launch still reads and writes every state element once.)
//
// Structure (one wave = one 16-column strip, streaming down a segment of rows in tiles of 16):
//  * axis 0.  B operand = the state itself: lane (g = lane>>4, c = lane&15) holds X[row0 + 4*kb + g][col c] for the
//    NK = K/4 k-blocks of the current window: a register ring that advances by 4 k-blocks (16 rows) per tile; the 4 new
//    values per lane are loaded BLM_PF tiles ahead (a load instruction = 4 rows x 128 contiguous bytes).
//    A operand = the band of the weight matrix: W[m][k] = w(|k - R0 - m|), identical for every tile, wave and block of a
//    chain: built once per block in LDS (NK x 64 doubles), conflict-free ds_read_b64.
//  * axis 1 (H kernels: blocks of 5 waves).  Waves 0-3 own 16 columns each, wave 4 the 8 + 8 halo columns left and right
//    of the block (reflected at the grid edge through the column index).  Every wave runs the axis-0 product for its
//    columns and writes the 16 x 16 result to an LDS tile (16 x 80, double-buffered, one barrier per tile); waves 0-3
//    then compute  OUT(16 x 16) = V(16 x 32) * W1(32 x 16)  with V read from LDS as the A operand (row stride 82
//    doubles: conflict-free) and the axis-1 band as a per-lane constant B operand (8 registers).
//  * D (lane holds rows g, g+4, g+8, g+12 of column c) feeds the same fused epilogue as blf::fast_step_kernel: lazy
//    normaliser, likelihood, posterior / next state stores, deterministic per-block partial sums.  The Gaussian
//    likelihood recurrence runs along the lane's own rows, i.e. with stride 4.
//
// Algorithmic HBM traffic per cell and step: forward 16 B, backward 32 B (as the other kernels).
#pragma once
#include "blhip_fast.hpp"

namespace blm {

using blf::FastParams;
using blf::exp_mn;
using blf::reflect1;
using blf::sld;
using blf::sldi;
using blf::DMAX;
using blk::NRED;
using blk::SRC_PREV;

typedef double d4 __attribute__((ext_vector_type(4)));

#ifndef BLM_PF
#define BLM_PF 2             // tiles between the load of a row block and its use
#endif
constexpr int TM = 16;            // rows per tile (MFMA M)
constexpr int WCOL = 16;          // columns per wave (MFMA N)
constexpr int BCOL = 4 * WCOL;    // output columns per block
constexpr int RSTEPS = 16;        // recurrence steps (of 4 rows) between exact re-anchorings
constexpr int SEG_Q = BLM_PF * TM; // segment lengths are multiples of this (the tile loop is unrolled BLM_PF times)
constexpr int MS_MAX = 2048;      // longest row segment of a block (the row coordinates are staged in LDS)
constexpr int R1 = 8;             // axis-1 radius bucket of the H kernels (= blf::R1MAX)
constexpr int NK1 = (TM + 2 * R1) / 4;   // k-blocks of the axis-1 product
constexpr int RS = BCOL + 2 * R1 + 2;    // LDS row stride of the row-filtered tile (82 doubles)
constexpr int NT_V = 256, NT_H = 320;    // threads per block without / with the axis-1 pass

template <int NW>
__device__ __forceinline__ double block_sum_w(double v, double *red) {
    v = blk::wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    double s = red[0];
#pragma unroll
    for (int k = 1; k < NW; ++k) s += red[k];
    return s;
}

// LEAN (axis-0-only launches whose segments are whole groups of BLM_PF tiles inside the grid, state < 4 GB per chain): no
// row masks, a lane past the last column recomputes and re-stores column n1 - 1 bit for bit (except the in-place posterior
// of the backward step, which it sends to a dump slot), 32-bit byte
// offsets from wave-uniform base pointers.  On gfx950 every VALU instruction adds to the f64 MFMA time (see above), and the
// generic addressing / masking was ~30 % of the non-MFMA instructions of a tile.
__device__ __forceinline__ double ld32(const double *base, unsigned byteoff) { return *(const double *)((const char *)base + byteoff); }
__device__ __forceinline__ void st32(double *base, unsigned byteoff, double v) { *(double *)((char *)base + byteoff) = v; }
// the same with a non-temporal hint: what a launch writes or reads ONCE (new state, stored alpha, posterior)
__device__ __forceinline__ double ld32nt(const double *base, unsigned byteoff) { return __builtin_nontemporal_load((const double *)((const char *)base + byteoff)); }
__device__ __forceinline__ void st32nt(double *base, unsigned byteoff, double v) { __builtin_nontemporal_store(v, (double *)((char *)base + byteoff)); }

template <int OM, int MODE, int NK, bool REC, bool H, bool LEAN>
__global__ __launch_bounds__(H ? NT_H : NT_V) void mfma_step_kernel(const FastParams P) {
    static_assert(!(H && LEAN), "the lean addressing is for the axis-0-only kernels");
    constexpr bool BWD = MODE == blk::MODE_BWD;
    constexpr bool GAUSS = OM == blk::OM_GAUSSIAN;
    constexpr int R0 = (4 * NK - TM) / 2;
    constexpr int NT = H ? NT_H : NT_V, NW = NT / 64;
    static_assert(NK >= 4 && (NK > 4 || H), "NK = 4 (no axis-0 filter) only makes sense with the axis-1 pass");
    __shared__ double As[NK * 64];
    __shared__ double m0s[MS_MAX + 2 * TM];
    __shared__ double red[5 * NW + 1];
    __shared__ double Vt[H ? 2 * TM * RS : 1];

    const blf::ChainMeta cmeta = blf::chain_meta(P, NK > 4, H);
    const int b = cmeta.b;
    const int blkid = blockIdx.x;
    const int tj = blkid / P.mnseg, seg = blkid - tj * P.mnseg;
    const int i_lo = seg * P.mS, i_hi = min(P.n0, i_lo + P.mS);
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool mainw = !H || wv < 4;                              // wave-uniform: wave 4 of an H block = halo columns

    const int kind = cmeta.kind, lw0 = cmeta.lw0;
    // (hsrc: the chain's source after the axis-1 pre-pass, blhip_hwide.hpp -- whatever its kind, which still selects the scale)
    const double *src = P.hsrc ? P.hsrc + (long long)b * P.n0 * P.n1 : (kind == SRC_PREV ? P.src + (long long)b * P.src_stride : P.shared[kind]);

    // ---- this lane's column ---------------------------------------------------------------------------------------------
    const int jc = mainw ? tj * BCOL + wv * WCOL + c : (c < R1 ? tj * BCOL - R1 + c : tj * BCOL + BCOL + (c - R1));
    const int gj = H ? reflect1(jc, P.n1) : min(jc, P.n1 - 1);
    const bool owner = mainw && jc < P.n1;
    const double *col = src + gj;

    // ---- prologue: first window of the strip (B ring), while those loads fly: bands + row coordinates -> LDS, normaliser
    double Bv[NK];
#pragma unroll
    for (int kb = 0; kb < NK; ++kb) Bv[kb] = col[(long long)reflect1(i_lo - R0 + 4 * kb + g, P.n0) * P.n1];

    if (NK > 4) {
        const long long o0 = cmeta.o0;
        for (int e = tid; e < NK * 64; e += NT) {
            const int kb = e >> 6, l = e & 63;
            const int a = abs(4 * kb + (l >> 4) - R0 - (l & 15));
            As[e] = a == 0 ? (lw0 > 0 ? P.taps[o0] : 1.0) : (a <= lw0 ? P.taps[o0 + a] : 0.0);
        }
    }
    for (int e = tid; e < P.mS + 2 * TM; e += NT) m0s[e] = P.m0[min(i_lo + e, P.n0 - 1)];
    double w1b[H ? NK1 : 1];                                      // axis-1 band, B operand: W1[k = 4 kb + g][n = c]
    if (H) {
        const int lw1 = cmeta.lw1;
        const long long o1 = cmeta.o1;
#pragma unroll
        for (int kb = 0; kb < NK1; ++kb) {
            const int a = abs(4 * kb + g - R1 - c);
            w1b[kb] = a == 0 ? (lw1 > 0 ? P.taps[o1] : 1.0) : (a <= lw1 ? P.taps[o1 + a] : 0.0);
        }
    }
    double scale = 1.0;
    if (kind == SRC_PREV) {
        const double *pp = P.psum_prev + ((long long)b * NRED + P.prev_slot) * P.prev_nblk;
        double v = 0.0;
        for (int k = tid; k < P.prev_nblk; k += NT) v += pp[k];
        scale = 1.0 / block_sum_w<NW>(v, red);
    }
    __syncthreads();

    double xd[DMAX];
#pragma unroll
    for (int k = 0; k < DMAX; ++k) xd[k] = (GAUSS && k < P.d) ? sld(P.rec, k) : __builtin_nan("");
    const double g1 = P.m1[gj];
    double cA = 0.0, cB = 0.0;
    if (GAUSS) { cA = P.colA[gj]; cB = P.colB[gj]; }
    asm volatile("" : "+v"(cA), "+v"(cB) : "v"(g1), "v"(scale), "v"(Bv[NK - 1]));

    double mE = 1.0, mR = 1.0, mq = 1.0, iE = 1.0, iR = 1.0, iq = 1.0;
    int nE = 0, nR = 0, nq = 0;
    double sN = 0.0, sS = 0.0, sC = 0.0, sM0 = 0.0, sM1 = 0.0;
    // Lanes that own no output column (columns past the grid edge, the halo wave) address a dump slot with row stride 0:
    // their loads and stores need no mask inside the loop, their sums are discarded at the end.
    double *const dump = P.dump + (tid & 255);
    const bool addressed = owner || LEAN;          // LEAN: dead column lanes duplicate column n1 - 1 (identical values)
    const long long rs = addressed ? P.n1 : 0;
    double *const dcol = addressed ? P.dst + (long long)b * P.dst_stride + gj : dump;
    double *const pcol = (BWD && addressed) ? P.post + (long long)b * P.post_stride + gj : dump;
    const double *const lcol = (!GAUSS && addressed) ? P.lik + gj : dump;
    // LEAN: wave-uniform bases + 32-bit byte offsets (row * n1 * 8 + column * 8 via a 24-bit multiply-add)
    double *const dbase = P.dst + (long long)b * P.dst_stride;
    double *const pbase = BWD ? P.post + (long long)b * P.post_stride : nullptr;
    const unsigned n1x8 = (unsigned)P.n1 * 8u, gj8 = (unsigned)gj * 8u;
    // rows entering the window in the steady state are >= 0: only the upper mirror and the clamp of the tail remain
    auto refl_hi = [&](int r) { return max(min(r, 2 * P.n0 - 1 - r), 0); };
    typedef const double __attribute__((address_space(3))) *lds_cp;
    lds_cp Al = (lds_cp)As + lane;                 // (the compiler hoists the NK band values of a lane into registers; re-reading
                                                   //  them from LDS per tile was measured and was not faster)

    static_assert(BLM_PF % 2 == 0, "prefetch depth must be even (two alpha / likelihood slots, two LDS tiles)");
    // Memory pipeline.  Everything inside the tile loop is straight-line and UNCONDITIONAL (clamped / reflected addresses
    // for the loads, a dump slot for the stores of dead lanes): with control flow around a memory instruction the compiler
    // cannot count what is outstanding and falls back to s_waitcnt vmcnt(0), i.e. drains the stores of the tile and the
    // prefetch before every ring advance.  Register slots are static (the loop is unrolled BLM_PF times): a rotating
    // copy would read -- i.e. wait for -- registers whose loads are still in flight.
    //   nxt[u] : the 4 k-blocks (16 rows) that enter the window when the ring advances from tile u to tile u+1 (mod BLM_PF),
    //            re-filled right after they are consumed, i.e. BLM_PF tiles before their next use
    //   al/lk  : stored forward posterior / tabulated likelihood of a tile, loaded one tile ahead
    double nxt[BLM_PF][4];
#pragma unroll
    for (int u = 0; u < BLM_PF; ++u) {
#pragma unroll
        for (int q = 0; q < 4; ++q) nxt[u][q] = col[(long long)reflect1(i_lo + (u + 1) * TM + R0 + 4 * q + g, P.n0) * P.n1];
    }
    double al[2][4], lk[2][4];
    if (BWD) {
#pragma unroll
        for (int r = 0; r < 4; ++r) al[0][r] = pcol[(long long)min(i_lo + g + 4 * r, P.n0 - 1) * rs];
    }
    if (!GAUSS) {
#pragma unroll
        for (int r = 0; r < 4; ++r) lk[0][r] = lcol[(long long)min(i_lo + g + 4 * r, P.n0 - 1) * rs];
    }

    for (int i0 = i_lo; i0 < i_hi; i0 += BLM_PF * TM) {
#pragma unroll
        for (int u = 0; u < BLM_PF; ++u) {
            const int i = i0 + u * TM;             // (a tile past the end of the segment is all dead rows: mS % (BLM_PF * TM) == 0
                                                   //  keeps that to the ragged end of the grid)
            const int li = i - i_lo + g;           // this lane's first row of the tile, relative to the segment
            if (BWD) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    al[(u + 1) & 1][r] = LEAN ? ld32nt(pbase, __umul24(min(i + TM + g + 4 * r, P.n0 - 1), n1x8) + gj8)
                                              : pcol[(long long)min(i + TM + g + 4 * r, P.n0 - 1) * rs];
            }
            if (!GAUSS) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    lk[(u + 1) & 1][r] = LEAN ? ld32(P.lik, __umul24(min(i + TM + g + 4 * r, P.n0 - 1), n1x8) + gj8)
                                              : lcol[(long long)min(i + TM + g + 4 * r, P.n0 - 1) * rs];
            }

            // ---- axis-0 stencil: NK chained MFMAs (k ascending); NK == 4: no filter, the ring IS the D layout -------------
            d4 acc = {0.0, 0.0, 0.0, 0.0};
            if (NK > 4) {
#pragma unroll
                for (int kb = 0; kb < NK; ++kb) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Al[kb * 64], Bv[kb], acc, 0, 0, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = Bv[r];
            }

            // ---- axis-1 stencil: row-filtered tile -> LDS -> second banded product (waves 0-3) ----------------------------
            if (H) {
                double *vt = Vt + (u & 1) * (TM * RS);
                const int lc = mainw ? R1 + wv * WCOL + c : (c < R1 ? c : BCOL + c);
#pragma unroll
                for (int r = 0; r < 4; ++r) vt[(g + 4 * r) * RS + lc] = acc[r];
                __syncthreads();
                if (mainw) {
                    lds_cp va = (lds_cp)vt + c * RS + wv * WCOL + g;
                    d4 acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int kb = 0; kb < NK1; ++kb) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(va[4 * kb], w1b[kb], acc2, 0, 0, 0);
                    acc = acc2;
                }
            }

            double st1[4], st2[4];                 // what the 4 cells store: forward a | backward posterior, next state
#pragma unroll
            for (int r = 0; r < 4; ++r) { st1[r] = acc[r]; st2[r] = acc[r]; }   // (halo wave: anything, goes to the dump slot)
            if (mainw) {
                // ---- re-anchor the stride-4 likelihood recurrence of this lane's rows ------------------------------------
                if (GAUSS && REC && ((i - i_lo) % (4 * RSTEPS)) == 0) {
                    // arg(r) = sum_k [-(x_k - mu_r)^2 cA - cB]  (observationModels.py:566-567; product over dimensions :49-50)
                    // arg(r+4) - arg(r) = cA (mu_{r+4} - mu_r) sum_k (2 x_k - mu_r - mu_{r+4});  2nd difference = -2 cA dn (4 step)^2
                    const double mu0 = m0s[li], mu4 = m0s[li + 4];
                    double a0 = 0.0, s1 = 0.0, dn = 0.0;
#pragma unroll
                    for (int k = 0; k < DMAX; ++k) {
                        const double x = xd[k];
                        if (x == x) {
                            const double q = x - mu0;
                            a0 = fma(-(q * q), cA, a0) - cB;
                            s1 += (x - mu0) + (x - mu4);
                            dn += 1.0;
                        }
                    }
                    const double d1 = cA * (mu4 - mu0) * s1;
                    const double d2 = -32.0 * cA * dn * P.step0 * P.step0;
                    exp_mn(a0, mE, nE);
                    exp_mn(d1, mR, nR);
                    exp_mn(d2, mq, nq);
                    if (BWD) {
                        int t;
                        exp_mn(-a0, iE, t);
                        exp_mn(-d1, iR, t);
                        exp_mn(-d2, iq, t);
                    }
                }

                // ---- epilogue arithmetic: the lane's 4 cells (rows i + g + 4 r) ------------------------------------------
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gi = i + g + 4 * r;
                    const double mu = m0s[li + 4 * r];
                    double Lv;
                    if (GAUSS && REC) {
                        Lv = ldexp(mE, nE);
                    } else if (GAUSS) {
                        Lv = 1.0;
#pragma unroll
                        for (int k = 0; k < DMAX; ++k) {
                            const double xx = xd[k];
                            if (xx == xx) { const double dq = xx - mu; Lv *= exp(-(dq * dq) * cA - cB); }
                        }
                    } else {
                        Lv = lk[u & 1][r];
                    }
                    const bool live = LEAN || gi < i_hi;
                    if (!BWD) {
                        const double a = acc[r] * scale * Lv;
                        st1[r] = a;
                        const double am = live ? a : 0.0;
                        sN += am;
                        if (P.means) { sM0 = fma(am, mu, sM0); sM1 = fma(am, g1, sM1); }
                    } else {
                        const double beta = acc[r] * scale;
                        const double p = al[u & 1][r] * beta;
                        const double cn = beta * Lv;
                        // p / L: reciprocal recurrence (no division, no intermediate overflow); 0/0 -> NaN (core.py:463)
                        const double pl = (GAUSS && REC) ? (Lv == 0.0 ? __builtin_nan("") : ldexp(p * iE, -nE)) : p / Lv;
                        st1[r] = p;
                        st2[r] = cn;
                        const double pm = live ? p : 0.0;
                        sN += pm;
                        sS += live ? pl : 0.0;
                        sC += live ? cn : 0.0;
                        sM0 = fma(pm, mu, sM0);
                        sM1 = fma(pm, g1, sM1);
                    }
                    if (GAUSS && REC) {
                        mE *= mR; nE += nR;
                        mR *= mq; nR += nq;
                        if (BWD) { iE *= iR; iR *= iq; }
                    }
                }
            }

            // ---- stores (every lane: dead rows / lanes hit the dump slot) ---------------------------------------------------
            // The halo wave of an H block shares its SIMD with wave 0, whose f64 MFMAs block every other VALU issue there: each
            // VALU instruction the halo wave does not need shortens the wait of the four main waves at the next barrier.  It
            // issues the same NUMBER of stores as a main wave (the compiler's vmcnt bookkeeping stays exact where the paths
            // join), to fixed dump addresses (the dump area holds 4 x NTHREADS doubles), with no address arithmetic.
            if (H && !mainw) {
#pragma unroll
                for (int r = 0; r < (BWD ? 8 : 4); ++r) dump[64 * r] = acc[r & 3];      // (distinct addresses: nothing to eliminate)
            } else
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = i + g + 4 * r;
                if (LEAN) {
                    const unsigned off = __umul24(gi, n1x8) + gj8;
                    if (!BWD) {
                        st32nt(dbase, off, st1[r]);
                    } else {
                        // the posterior overwrites the stored alpha IN PLACE: a lane past the last column must not touch
                        // column n1 - 1 (another wave may own it and be at a different tile); the state store is idempotent
                        __builtin_nontemporal_store(st1[r], owner ? (double *)((char *)pbase + off) : dump);
                        st32nt(dbase, off, st2[r]);
                    }
                } else {
                    const bool live = gi < i_hi;
                    const long long off = (long long)gi * rs;
                    if (!BWD) {
                        *(live ? dcol + off : dump) = st1[r];
                    } else {
                        *(live ? pcol + off : dump) = st1[r];
                        *(live ? dcol + off : dump) = st2[r];
                    }
                }
            }

            // ---- advance the ring by one tile, re-fill the slot ------------------------------------------------------------
#pragma unroll
            for (int kb = 0; kb < NK - 4; ++kb) Bv[kb] = Bv[kb + 4];
#pragma unroll
            for (int q = 0; q < 4; ++q) Bv[NK - 4 + q] = nxt[u][q];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                nxt[u][q] = LEAN ? ld32(src, __umul24(refl_hi(i + (BLM_PF + 1) * TM + R0 + 4 * q + g), n1x8) + gj8)
                                 : col[(long long)reflect1(i + (BLM_PF + 1) * TM + R0 + 4 * q + g, P.n0) * P.n1];
        }
    }

    if (!owner) { sN = 0.0; sS = 0.0; sC = 0.0; sM0 = 0.0; sM1 = 0.0; }
    double *out = P.psum_out + (long long)b * NRED * P.nblk + blkid;
    const int left = P.nblk - blkid;
    const bool z = tid == 0;
    if (BWD) {
        double v[5] = {sN, sS, sC, sM0, sM1};
        blk::block_sums<5, NW>(v, red);
        if (z) {
#pragma unroll
            for (int k = 0; k < 5; ++k) blf::put_partial(out + k * P.nblk, v[k], P.mnblk, left);
        }
    } else if (P.means) {
        double v[3] = {sN, sM0, sM1};
        blk::block_sums<3, NW>(v, red);
        if (z) { blf::put_partial(out, v[0], P.mnblk, left); blf::put_partial(out + 3 * P.nblk, v[1], P.mnblk, left); blf::put_partial(out + 4 * P.nblk, v[2], P.mnblk, left); }
    } else {
        double v[1] = {sN};
        blk::block_sums<1, NW>(v, red);
        if (z) blf::put_partial(out, v[0], P.mnblk, left);
    }
}

}  // namespace blm
