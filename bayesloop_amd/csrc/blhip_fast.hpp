// Fast fused step kernels for 2-D grids (gfx950): the STREAMING COLUMN kernel.
// Same math as blk::step_kernel (blhip_kernels.hpp), organised for the MI355X memory system and fp64 VALU budget:
//
//  * a block owns a strip of 256 grid columns (240 useful + 2 x 8 halo lanes when the launch has an axis-1 filter) and
//    a SEGMENT of S rows; a thread owns ONE COLUMN and walks down the segment in chunks of CH = 8 rows.
//  * every load is one fully coalesced 512-B wave access (consecutive lanes = consecutive doubles of a grid row).
//    The thread keeps a sliding window of 2*R0 + CH rows of its column in registers (R0 = compile-time radius bucket of
//    the axis-0 stencil), so each state element is read ONCE per step (plus 2*R0 rows per segment), and the loads of
//    chunk k+1 are issued before the arithmetic of chunk k starts: HBM latency hides under the fp64 work of the same
//    wave instead of relying on other waves that would be in the same phase anyway.
//  * axis-0 (row) stencil: out of the register window, weights in SGPRs, SciPy's symmetric correlate1d order.
//  * axis-1 (column) stencil: the row-filtered chunk goes through one LDS tile (CH x 257 doubles, conflict-free
//    ds_read_b64); halo columns are owned by the halo lanes of the same block (reflect boundary via the column index).
//  * Gaussian likelihood WITHOUT an exp per cell: along a column the exponent is a quadratic in the row index, so
//    L(row) follows a second-order multiplicative recurrence, carried as mantissa * 2^exponent (cannot overflow, and
//    cannot lose a value to underflow on the way towards the likelihood peak); re-anchored every 16 rows with exact
//    exponentials (error << 1e-12 relative).  p / L of the backward pass uses the reciprocal recurrence: no division.
//  * lazy normalisation and deterministic per-block partial sums as in blk::step_kernel.
//
// Algorithmic HBM traffic per cell and step: forward 16 B (read state, write state), backward 32 B.
#pragma once
#include "blhip_kernels.hpp"
#include "blhip_expmn.hpp"

namespace blf {

using blk::NRED;
using blk::NTHREADS;
using blk::SRC_PREV;

constexpr int CH = 8;           // rows per chunk
constexpr int BW = NTHREADS;    // columns per block including the axis-1 halo lanes
constexpr int R1MAX = 8;        // largest axis-1 radius of the fast path
constexpr int ANCHOR = 16;      // rows between exact re-anchorings of the likelihood recurrence

struct FastParams {
    int n0, n1;
    int TJ;                      // useful columns per block: BW - 2*R1MAX if the batch has an axis-1 filter, else BW
    int S;                       // rows per segment (multiple of CH)
    int nseg, tiles_j, fnblk;    // fnblk = nseg * tiles_j blocks per chain
    int nblk;                    // partial-sum slots per chain and reduction = max(fnblk, mnblk): both kernel families of a
                                 // batch write the same slot layout (blocks zero the slots their family does not use)
    int mS, mnseg, mtiles_j, mnblk;   // geometry of blm::mfma_step_kernel (blhip_mfma.hpp): 64-column strips, mS rows
    int mlean;                        // every segment is mS full rows (mS % 32 == 0) and offsets fit 32 bits: LEAN kernels
    int ndim, d, means, use_rec;
    double step0;                // lattice step of the row axis (likelihood recurrence)
    const double *src;  long long src_stride;
    const double *hsrc;          // != nullptr: [chains][n0 * n1] sources already filtered along axis 1 (wide random walks: blhip_hwide.hpp)
    double       *dst;  long long dst_stride;
    double       *post; long long post_stride;
    const double *shared[5];
    const int *chain_ids;        // [gridDim.y] -> chain index in the batch
    const unsigned char *srckind;
    const int *tap0, *tap1;
    const double *taps; const int *tap_off; const int *tap_lw;
    const double *psum_prev; int prev_slot; int prev_nblk;
    double *psum_out;
    const double *m0, *m1, *colA, *colB, *rec, *lik;
    double *dump;                // NTHREADS doubles nobody reads: where dead lanes store (keeps the stores branch-free)
    // single-chain launches (every plain Study.fit): the host already knows what the block would fetch through a chain of
    // dependent scalar loads (chain id -> source kind / tap ids -> radius / offset, ~1 us per block): u_valid = 1
    int u_valid, u_chain, u_kind, u_t0, u_t1, u_lw0, u_lw1;
    long long u_off0, u_off1;
};

// per-chain launch metadata: from the kernel arguments (single-chain launch) or from the device tables
struct ChainMeta { int b, kind, t0, lw0, t1, lw1; long long o0, o1; };
__device__ __forceinline__ int sldi_(const int *p, long long i) { return ((const int __attribute__((address_space(4))) *)(unsigned long long)p)[i]; }
__device__ __forceinline__ ChainMeta chain_meta(const FastParams &P, bool want0, bool want1) {
    ChainMeta m;
    if (P.u_valid) {
        m.b = P.u_chain; m.kind = P.u_kind; m.t0 = P.u_t0; m.lw0 = want0 ? P.u_lw0 : 0; m.o0 = P.u_off0;
        m.t1 = P.u_t1; m.lw1 = want1 ? P.u_lw1 : 0; m.o1 = P.u_off1;
    } else {
        m.b = sldi_(P.chain_ids, blockIdx.y);
        m.kind = ((const unsigned char __attribute__((address_space(4))) *)(unsigned long long)P.srckind)[m.b];
        m.t0 = sldi_(P.tap0, m.b);
        m.lw0 = (want0 && m.t0 >= 0) ? sldi_(P.tap_lw, m.t0) : 0;
        m.o0 = (want0 && m.t0 >= 0) ? sldi_(P.tap_off, m.t0) : 0;
        m.t1 = want1 ? sldi_(P.tap1, m.b) : -1;
        m.lw1 = (want1 && m.t1 >= 0) ? sldi_(P.tap_lw, m.t1) : 0;
        m.o1 = (want1 && m.t1 >= 0) ? sldi_(P.tap_off, m.t1) : 0;
    }
    return m;
}

// block `blkid` of a kernel family with my_nblk blocks per chain publishes its partial sum in slot blkid and zeroes the
// slots blkid + k * my_nblk the family does not own (left = slots from blkid to the end)
__device__ __forceinline__ void put_partial(double *out, double v, int my_nblk, int left) {
    out[0] = v;
    for (int j = my_nblk; j < left; j += my_nblk) out[j] = 0.0;
}

// Pin a wave-uniform double into SGPRs (the compiler cannot prove that the tap table is not aliased by the stores,
// so without this the stencil weights occupy 2 VGPRs each).
__device__ __forceinline__ double uniform(double x) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
    return __hiloint2double(hi, lo);
}

using blmath::exp_mn;

// wave-uniform read-only tables go through the scalar cache (s_load): no VGPRs, no vmcnt traffic
typedef const double __attribute__((address_space(4))) *cdptr;
typedef const int __attribute__((address_space(4))) *ciptr;
__device__ __forceinline__ double sld(const double *p, long long i) { return ((cdptr)(unsigned long long)p)[i]; }
__device__ __forceinline__ int sldi(const int *p, long long i) { return ((ciptr)(unsigned long long)p)[i]; }

// single-period half-sample reflection, branch-free (the host guarantees |offset| < n for every use in the fast path)
__device__ __forceinline__ int reflect1(int i, int n) {
    i = i < 0 ? -1 - i : i;
    i = i >= n ? 2 * n - 1 - i : i;
    return min(max(i, 0), n - 1);
}

constexpr int DMAX = 4;         // data dimensions kept in registers

template <int OM, int MODE, int R0, bool H, bool REC>
__global__ __launch_bounds__(NTHREADS, 1) void fast_step_kernel(const FastParams P) {
    constexpr bool BWD = MODE == blk::MODE_BWD;
    constexpr bool GAUSS = OM == blk::OM_GAUSSIAN;
    constexpr int WIN = 2 * R0 + CH;
    // two LDS tiles, alternating per chunk: ONE barrier per chunk (a lane can only reach the next write of a tile after
    // the barrier of the chunk in between, i.e. after every lane has finished reading that tile)
    constexpr int VTSZ = CH * (BW + 1);
    __shared__ __attribute__((aligned(16))) double vtbuf[H ? 2 * VTSZ : 1];
    __shared__ double red[5 * (NTHREADS / 64) + 1];

    const ChainMeta cmeta = chain_meta(P, R0 > 0, H);
    const int b = cmeta.b;
    const int blkid = blockIdx.x;
    const int tj = blkid / P.nseg, seg = blkid - tj * P.nseg;
    const int i_lo = seg * P.S, i_hi = min(P.n0, i_lo + P.S);
    const int j0 = tj * P.TJ;
    const int tid = threadIdx.x;

    const int kind = cmeta.kind, lw0 = cmeta.lw0;
    // (hsrc: the chain's source after the axis-1 pre-pass, blhip_hwide.hpp -- whatever its kind, which still selects the scale)
    const double *src = P.hsrc ? P.hsrc + (long long)b * P.n0 * P.n1 : (kind == SRC_PREV ? P.src + (long long)b * P.src_stride : P.shared[kind]);

    // ---- this thread's column ------------------------------------------------------------------------------------
    const int jc = j0 - (H ? R1MAX : 0) + tid;                 // grid column (may lie in the halo / outside)
    const int gj = reflect1(jc, P.n1);
    const bool owner = H ? (tid >= R1MAX && tid < R1MAX + P.TJ && jc < P.n1) : (tid < P.TJ && jc < P.n1);
    const double *col = src + gj;

    // ---- prologue: first window of the column, then the lazy normaliser while those loads are in flight -----------
    double w[WIN];
#pragma unroll
    for (int k = 0; k < WIN; ++k) w[k] = col[(long long)reflect1(i_lo - R0 + k, P.n0) * P.n1];

    double scale = 1.0;
    if (kind == SRC_PREV) {
        const double s = blk::sum_partials(P.psum_prev + ((long long)b * NRED + P.prev_slot) * P.prev_nblk, P.prev_nblk, red);
        scale = 1.0 / s;
    }

    // stencil weights (SGPRs), zero beyond this chain's radius
    double wk[R0 + 1];
    if (R0 > 0) {
        const long long o0 = cmeta.o0;
#pragma unroll
        for (int k = 0; k <= R0; ++k) wk[k] = (lw0 > 0 && k <= lw0) ? sld(P.taps, o0 + k) : (k == 0 ? 1.0 : 0.0);
    }
    double w1[R1MAX + 1];
    if (H) {
        const int t1 = cmeta.t1, lw1 = cmeta.lw1;
        const long long o1 = cmeta.o1;
#pragma unroll
        for (int k = 0; k <= R1MAX; ++k) w1[k] = (t1 >= 0 && k <= lw1) ? sld(P.taps, o1 + k) : (k == 0 ? 1.0 : 0.0);
    }
    // data of this step (NaN = missing, observationModels.py:53-54)
    double xd[DMAX];
#pragma unroll
    for (int k = 0; k < DMAX; ++k) xd[k] = (GAUSS && k < P.d) ? sld(P.rec, k) : __builtin_nan("");

    const double g1 = P.m1[gj];
    double cA = 0.0, cB = 0.0;
    if (GAUSS) { cA = P.colA[gj]; cB = P.colB[gj]; }
    // land the per-column table values (and the window) BEFORE the loop: a first use inside the loop would make the
    // compiler wait with vmcnt(0) there, i.e. drain the prefetch of the next chunk every iteration
    asm volatile("" : "+v"(cA), "+v"(cB) : "v"(g1), "v"(scale), "v"(w[WIN - 1]));

    // likelihood recurrence state (REC): L = mE 2^nE, ratio mR 2^nR, curvature mq 2^nq; reciprocals for the backward pass
    double mE = 1.0, mR = 1.0, mq = 1.0, iE = 1.0, iR = 1.0, iq = 1.0;
    int nE = 0, nR = 0, nq = 0;
    double sN = 0.0, sS = 0.0, sC = 0.0, sM0 = 0.0, sM1 = 0.0;
    double *dcol = P.dst + (long long)b * P.dst_stride + gj;
    double *pcol = BWD ? P.post + (long long)b * P.post_stride + gj : nullptr;
    const double *lcol = (!GAUSS) ? P.lik + gj : nullptr;

    // ---- software pipeline: while chunk c is computed, the CH rows that enter the window for chunk c+1 are in flight ----
    for (int i = i_lo; i < i_hi; i += CH) {
        // loads in consumption order: this chunk's stored alpha / tabulated likelihood first, then the prefetch, so that
        // waiting for the former (vmcnt is in-order) leaves the prefetch in flight during the arithmetic
        double al[CH], lk[CH];
        if (BWD) {
#pragma unroll
            for (int r = 0; r < CH; ++r) al[r] = pcol[(long long)min(i + r, P.n0 - 1) * P.n1];
        }
        if (!GAUSS) {
#pragma unroll
            for (int r = 0; r < CH; ++r) lk[r] = lcol[(long long)min(i + r, P.n0 - 1) * P.n1];
        }
        double nx[CH];
        const bool more = i + CH < i_hi;
        if (more) {
#pragma unroll
            for (int k = 0; k < CH; ++k) nx[k] = col[(long long)reflect1(i + CH + R0 + k, P.n0) * P.n1];
        }

        // ---- axis-0 stencil out of the register window: the CH rows are independent accumulator chains, interleaved
        //      tap by tap so that consecutive fp64 instructions never depend on each other (per-row order = SciPy's) ------
        double v[CH];
        if (R0 > 0) {
#pragma unroll
            for (int r = 0; r < CH; ++r) v[r] = w[r + R0] * wk[0];
#pragma unroll
            for (int k = R0; k >= 1; --k) {
                double t[CH];
#pragma unroll
                for (int r = 0; r < CH; ++r) t[r] = w[r + R0 - k] + w[r + R0 + k];
#pragma unroll
                for (int r = 0; r < CH; ++r) v[r] = fma(t[r], wk[k], v[r]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < CH; ++r) v[r] = w[r + R0];
        }

        // ---- axis-1 stencil through LDS ---------------------------------------------------------------------------------
        double *vt = vtbuf + (H ? (((i - i_lo) / CH) & 1) * VTSZ : 0);
        if (H) {
#pragma unroll
            for (int r = 0; r < CH; ++r) vt[r * (BW + 1) + tid] = v[r];
            __syncthreads();
        }

        // ---- re-anchor the likelihood recurrence with exact exponentials every ANCHOR rows ------------------------------
        if (GAUSS && REC && ((i - i_lo) % ANCHOR) == 0) {
            // arg(r) = sum_k [-(x_k - mu_r)^2 cA - cB]  (observationModels.py:566-567; product over dimensions :49-50)
            // arg(1) - arg(0) = cA (mu_1 - mu_0) sum_k (2 x_k - mu_0 - mu_1);  second difference = -2 cA dn step^2
            const double mu0 = sld(P.m0, min(i, P.n0 - 1)), mu1 = sld(P.m0, min(i + 1, P.n0 - 1));
            double a0 = 0.0, s1 = 0.0, dn = 0.0;
#pragma unroll
            for (int k = 0; k < DMAX; ++k) {
                const double x = xd[k];
                if (x == x) {
                    const double q = x - mu0;
                    a0 = fma(-(q * q), cA, a0) - cB;
                    s1 += (x - mu0) + (x - mu1);
                    dn += 1.0;
                }
            }
            const double d1 = cA * (mu1 - mu0) * s1;
            const double d2 = -2.0 * cA * dn * P.step0 * P.step0;
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

        // ---- epilogue (straight-line: no loads besides LDS); rows in pairs so that two independent chains interleave ----
#pragma unroll
        for (int r2 = 0; r2 < CH; r2 += 2) {
            double o[2];
            if (H) {
                // volatile: keeps the reads as ds_read_b64 (256 B/clk); the compiler otherwise fuses neighbours into
                // ds_read2_b64, which gfx950 services at half that rate (MI355X_MICROARCH.md, LDS table)
                typedef const volatile double __attribute__((address_space(3))) *lds_cvp;
                lds_cvp cen = (lds_cvp)(const double __attribute__((address_space(3))) *)vt + r2 * (BW + 1) +
                              min(max(tid, R1MAX), BW - 1 - R1MAX) - R1MAX;
                double c0[2 * R1MAX + 1], c1[2 * R1MAX + 1];
#pragma unroll
                for (int k = 0; k <= 2 * R1MAX; ++k) { c0[k] = cen[k]; c1[k] = cen[(BW + 1) + k]; }   // all LDS reads in flight
                o[0] = c0[R1MAX] * w1[0];
                o[1] = c1[R1MAX] * w1[0];
#pragma unroll
                for (int k = R1MAX; k >= 1; --k) {
                    const double t0 = c0[R1MAX - k] + c0[R1MAX + k];
                    const double t1 = c1[R1MAX - k] + c1[R1MAX + k];
                    o[0] = fma(t0, w1[k], o[0]);
                    o[1] = fma(t1, w1[k], o[1]);
                }
            } else {
                o[0] = v[r2]; o[1] = v[r2 + 1];
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = r2 + q;
                const int gi = i + r;
                double Lv;
                if (GAUSS && REC) {
                    Lv = ldexp(mE, nE);
                } else if (GAUSS) {
                    Lv = 1.0;
                    const double mu = sld(P.m0, min(gi, P.n0 - 1));
#pragma unroll
                    for (int k = 0; k < DMAX; ++k) {
                        const double xx = xd[k];
                        if (xx == xx) { const double dq = xx - mu; Lv *= exp(-(dq * dq) * cA - cB); }
                    }
                } else {
                    Lv = lk[r];
                }
                const bool live = owner && gi < i_hi;
                const long long off = (long long)gi * P.n1;
                if (!BWD) {
                    const double a = o[q] * scale * Lv;
                    if (live) {
                        dcol[off] = a;
                        sN += a;
                        if (P.means) { sM0 = fma(a, sld(P.m0, min(gi, P.n0 - 1)), sM0); sM1 = fma(a, g1, sM1); }
                    }
                } else {
                    const double beta = o[q] * scale;
                    const double p = al[r] * beta;
                    const double cn = beta * Lv;
                    // p / L: reciprocal recurrence (no division, no intermediate overflow); 0/0 -> NaN (core.py:463)
                    const double pl = (GAUSS && REC) ? (Lv == 0.0 ? __builtin_nan("") : ldexp(p * iE, -nE)) : p / Lv;
                    if (live) {
                        pcol[off] = p;
                        dcol[off] = cn;
                        sN += p;
                        sS += pl;
                        sC += cn;
                        sM0 = fma(p, sld(P.m0, min(gi, P.n0 - 1)), sM0);
                        sM1 = fma(p, g1, sM1);
                    }
                }
                if (GAUSS && REC) {
                    mE *= mR; nE += nR;
                    mR *= mq; nR += nq;
                    if (BWD) { iE *= iR; iR *= iq; }
                }
            }
        }

        // ---- slide the window ---------------------------------------------------------------------------------------------
        if (more) {
#pragma unroll
            for (int k = 0; k < 2 * R0; ++k) w[k] = w[k + CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) w[2 * R0 + k] = nx[k];
        }
    }

    double *out = P.psum_out + (long long)b * NRED * P.nblk + blkid;
    const int left = P.nblk - blkid;
    if (BWD) {
        double v[5] = {sN, sS, sC, sM0, sM1};
        blk::block_sums<5, NTHREADS / 64>(v, red);
        if (tid == 0) {
#pragma unroll
            for (int k = 0; k < 5; ++k) put_partial(out + k * P.nblk, v[k], P.fnblk, left);
        }
    } else if (P.means) {
        double v[3] = {sN, sM0, sM1};
        blk::block_sums<3, NTHREADS / 64>(v, red);
        if (tid == 0) { put_partial(out, v[0], P.fnblk, left); put_partial(out + 3 * P.nblk, v[1], P.fnblk, left); put_partial(out + 4 * P.nblk, v[2], P.fnblk, left); }
    } else {
        double v[1] = {sN};
        blk::block_sums<1, NTHREADS / 64>(v, red);
        if (tid == 0) put_partial(out, v[0], P.fnblk, left);
    }
}

}  // namespace blf
