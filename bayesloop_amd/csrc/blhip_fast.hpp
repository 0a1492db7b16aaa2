// Fast fused step kernels for 2-D grids (gfx950).  Same math as blk::step_kernel (blhip_kernels.hpp), organised for
// the MI355X memory system and its fp64 VALU budget:
//
//  * a thread OWNS ONE GRID COLUMN of a (TI x 256) tile.  It loads its column strip (TI + 2*R0 rows) straight from
//    HBM/L2 into registers -- consecutive lanes read consecutive doubles, so every load instruction is one fully
//    coalesced 512-B wave access and all TI + 2*R0 loads of a thread are in flight together -- and runs the axis-0
//    (row) stencil out of registers with a compile-time radius bucket R0 (weights in SGPRs).
//  * the axis-1 (column) stencil needs neighbouring columns: the row-filtered values go through one LDS tile
//    (TI x 256 doubles = 32 KiB, conflict-free ds_read_b64), halo columns are owned by halo threads of the same block.
//  * the Gaussian likelihood is NOT evaluated with one exp per cell: along a column the exponent is a quadratic in the
//    row index, so L(row) follows a second-order multiplicative recurrence.  It is carried as mantissa * 2^exponent
//    (three fp64 multiplies + one v_ldexp per cell), which can neither overflow nor lose a value to underflow on the
//    way towards the likelihood peak; three exps per column per tile re-anchor it (error << 1e-12 relative).
//  * lazy normalisation, per-block partial sums and the XCD-aware tile order are as in DESIGN.md.
//
// Algorithmic HBM traffic per cell and step: forward 16 B (read state, write state), backward 32 B.
#pragma once
#include "blhip_kernels.hpp"

namespace blf {

using blk::NRED;
using blk::NTHREADS;
using blk::SRC_PREV;

constexpr int TI = 16;          // rows per tile
constexpr int BW = NTHREADS;    // columns per tile including the axis-1 halo
constexpr int R1MAX = 8;        // largest axis-1 radius of the fast path

struct FastParams {
    int n0, n1;
    int TJ;                      // useful columns per tile = BW - 2*LW1 (same for every launch of a batch)
    int LW1;                     // axis-1 halo lanes on each side of a tile: R1MAX if the batch has an axis-1 filter, else 0
    int tiles_i, tiles_j, nblk;
    int swizzle;                 // 1: XCD-aware tile order (nblk % 8 == 0)
    int ndim, d, means, use_rec;
    double step0;                // lattice step of the row axis (likelihood recurrence)
    const double *src;  long long src_stride;
    double       *dst;  long long dst_stride;
    double       *post; long long post_stride;
    const double *shared[4];
    const int *chain_ids;        // [gridDim.y] -> chain index in the batch
    const unsigned char *srckind;
    const int *tap0, *tap1;
    const double *taps; const int *tap_off; const int *tap_lw;
    const double *psum_prev; int prev_slot; int prev_nblk;
    double *psum_out;
    const double *m0, *m1, *colA, *colB, *rec, *lik;
};

// exp(a) = m * 2^n with m in [0.70, 1.42]; never overflows / underflows.  |error| < 2e-16 relative.
__device__ __forceinline__ void exp_mn(double a, double &m, int &n) {
    a = fmin(fmax(a, -1.4e9), 1.4e9);
    const double kn = rint(a * 1.44269504088896340736);
    double r = fma(-kn, 6.93147180369123816490e-01, a);
    r = fma(-kn, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;            // 1/13!
    p = fma(p, r, 2.0876756987868100e-09);        // 1/12!
    p = fma(p, r, 2.5052108385441720e-08);        // 1/11!
    p = fma(p, r, 2.7557319223985893e-07);        // 1/10!
    p = fma(p, r, 2.7557319223985888e-06);        // 1/9!
    p = fma(p, r, 2.4801587301587302e-05);        // 1/8!
    p = fma(p, r, 1.9841269841269841e-04);        // 1/7!
    p = fma(p, r, 1.3888888888888889e-03);        // 1/6!
    p = fma(p, r, 8.3333333333333332e-03);        // 1/5!
    p = fma(p, r, 4.1666666666666664e-02);        // 1/4!
    p = fma(p, r, 1.6666666666666666e-01);        // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    m = p;
    n = (int)kn;
}

// Likelihood along one column as a recurrence over rows: L(i0 + r) = mE * 2^nE, advanced by step().
struct GaussRec {
    double mE, mR, mq;       // value, ratio to the next row, ratio of ratios (mantissas)
    int nE, nR, nq;          // their binary exponents
    double iE, iR, iq;       // mantissas of the reciprocals (backward pass: p / L without a division)

    template <bool INV>
    __device__ __forceinline__ void init(const FastParams &P, int i0, double cA, double cB) {
        // arg(r) = sum_k [ -(x_k - mu_r)^2 cA - cB ]   (observationModels.py:566-567, product over data dimensions :49-50)
        // first difference  arg(1)-arg(0) = cA (mu_1 - mu_0) sum_k (2 x_k - mu_0 - mu_1)
        // second difference                = -2 cA dn step^2      (regular grid)
        const int i1 = min(i0 + 1, P.n0 - 1);
        const double mu0 = P.m0[i0], mu1 = P.m0[i1];
        double a0 = 0.0, s1 = 0.0, dn = 0.0;
        for (int k = 0; k < P.d; ++k) {
            const double x = P.rec[k];
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
        exp_mn(d1, mR, nR);                        // ratio L(1)/L(0)
        exp_mn(d2, mq, nq);
        if (INV) {
            int t;
            exp_mn(-a0, iE, t);
            exp_mn(-d1, iR, t);
            exp_mn(-d2, iq, t);
        }
    }
    __device__ __forceinline__ double value() const { return ldexp(mE, nE); }
    // p / L with L = mE 2^nE, without forming 1/L (which may overflow while p/L does not)
    __device__ __forceinline__ double divide(double p, double L) const {
        return L == 0.0 ? __builtin_nan("") : ldexp(p * iE, -nE);
    }
    template <bool INV>
    __device__ __forceinline__ void step() {
        mE *= mR; nE += nR;
        mR *= mq; nR += nq;
        if (INV) { iE *= iR; iR *= iq; }
    }
};

template <int OM, int MODE, int R0, bool H>
__global__ __launch_bounds__(NTHREADS) void fast_step_kernel(const FastParams P) {
    constexpr bool BWD = MODE == blk::MODE_BWD;
    constexpr int NROW = TI + 2 * R0;
    __shared__ __attribute__((aligned(16))) double vt[H ? TI * (BW + 1) : 1];
    __shared__ double red[NTHREADS / 64 + 1];

    const int b = P.chain_ids[blockIdx.y];
    int tile = blockIdx.x;
    if (P.swizzle) {                              // XCD-aware order: block x runs on XCD x % 8; give each XCD a
        const int per = P.nblk >> 3;              // contiguous range of tiles so that row-halo re-reads hit its L2
        tile = (tile & 7) * per + (tile >> 3);
    }
    const int tj = tile / P.tiles_i, ti = tile - tj * P.tiles_i;     // consecutive tiles are vertical neighbours
    const int i0 = ti * TI, j0 = tj * P.TJ;
    const int tid = threadIdx.x;

    const int kind = P.srckind[b];
    const int t0 = P.tap0[b];
    const int lw0 = (R0 > 0 && t0 >= 0) ? P.tap_lw[t0] : 0;
    const double *src = kind == SRC_PREV ? P.src + (long long)b * P.src_stride : P.shared[kind];

    // ---- this thread's column ------------------------------------------------------------------------------------
    const int jc = j0 - (H ? R1MAX : 0) + tid;                 // grid column (may lie in the halo / outside)
    const int gj = blk::reflect(jc, P.n1);
    const bool owner = H ? (tid >= R1MAX && tid < R1MAX + P.TJ && jc < P.n1) : (tid < P.TJ && jc < P.n1);

    // ---- column strip -> registers (coalesced: lanes = consecutive columns) -----------------------------------------
    double x[NROW];
#pragma unroll
    for (int k = 0; k < NROW; ++k) {
        const int gi = blk::reflect(i0 - R0 + k, P.n0);
        x[k] = src[(long long)gi * P.n1 + gj];
    }

    // ---- lazy normaliser of the producing step ------------------------------------------------------------------------
    double scale = 1.0;
    if (kind == SRC_PREV) {
        const double s = blk::sum_partials(P.psum_prev + ((long long)b * NRED + P.prev_slot) * P.prev_nblk, P.prev_nblk, red);
        scale = 1.0 / s;
    }

    // ---- axis-0 stencil out of registers (SciPy's symmetric correlate1d order, zero weights beyond lw0) ------------
    double v[TI];
    if (R0 > 0 && lw0 > 0) {
        const double *w = P.taps + P.tap_off[t0];
        double wk[R0 + 1];
#pragma unroll
        for (int k = 0; k <= R0; ++k) wk[k] = k <= lw0 ? w[k] : 0.0;
#pragma unroll
        for (int r = 0; r < TI; ++r) {
            double acc = x[r + R0] * wk[0];
#pragma unroll
            for (int k = R0; k >= 1; --k) acc = fma(x[r + R0 - k] + x[r + R0 + k], wk[k], acc);
            v[r] = acc;
        }
    } else {
#pragma unroll
        for (int r = 0; r < TI; ++r) v[r] = x[r + R0];
    }

    // ---- axis-1 stencil through LDS -------------------------------------------------------------------------------------
    double w1[R1MAX + 1];
    int lw1 = 0;
    if (H) {
        const int t1 = P.tap1[b];
        lw1 = t1 >= 0 ? P.tap_lw[t1] : 0;
        const double *w = P.taps + (t1 >= 0 ? P.tap_off[t1] : 0);
#pragma unroll
        for (int k = 0; k <= R1MAX; ++k) w1[k] = (t1 >= 0 && k <= lw1) ? w[k] : (k == 0 ? 1.0 : 0.0);
#pragma unroll
        for (int r = 0; r < TI; ++r) vt[r * (BW + 1) + tid] = v[r];
        __syncthreads();
    }

    // ---- epilogue: likelihood, products, partial sums -----------------------------------------------------------------
    double sN = 0.0, sS = 0.0, sC = 0.0, sM0 = 0.0, sM1 = 0.0;
    if (owner) {
        const double g1 = P.m1[gj];
        double cA = 0.0, cB = 0.0;
        GaussRec L;
        if (OM == blk::OM_GAUSSIAN) {
            cA = P.colA[gj]; cB = P.colB[gj];
            if (P.use_rec) L.template init<BWD>(P, i0, cA, cB);
        }
#pragma unroll
        for (int r = 0; r < TI; ++r) {
            const int gi = i0 + r;
            double o;
            if (H) {
                const double *cen = vt + r * (BW + 1) + tid;
                o = cen[0] * w1[0];
#pragma unroll
                for (int k = R1MAX; k >= 1; --k) o = fma(cen[-k] + cen[k], w1[k], o);   // halo lanes exist: tid in [R1MAX, R1MAX+TJ)
            } else {
                o = v[r];
            }
            double Lv;
            if (OM == blk::OM_GAUSSIAN && P.use_rec) {
                Lv = L.value();
            } else if (OM == blk::OM_GAUSSIAN) {
                Lv = 1.0;
                const double mu = P.m0[min(gi, P.n0 - 1)];
                for (int k = 0; k < P.d; ++k) {
                    const double xx = P.rec[k];
                    if (xx == xx) { const double q = xx - mu; Lv *= exp(-(q * q) * cA - cB); }
                }
            } else {
                Lv = gi < P.n0 ? P.lik[(long long)gi * P.n1 + gj] : 1.0;
            }
            if (gi < P.n0) {
                const long long cell = (long long)gi * P.n1 + gj;
                if (!BWD) {
                    const double a = o * scale * Lv;
                    P.dst[(long long)b * P.dst_stride + cell] = a;
                    sN += a;
                    if (P.means) { sM0 = fma(a, P.m0[gi], sM0); sM1 = fma(a, g1, sM1); }
                } else {
                    const double beta = o * scale;
                    double *pp = P.post + (long long)b * P.post_stride + cell;
                    const double p = (*pp) * beta;
                    *pp = p;
                    const double cn = beta * Lv;
                    P.dst[(long long)b * P.dst_stride + cell] = cn;
                    sN += p;
                    sS += (OM == blk::OM_GAUSSIAN && P.use_rec) ? L.divide(p, Lv) : p / Lv;     // 0/0 -> NaN (core.py:463)
                    sC += cn;
                    sM0 = fma(p, P.m0[gi], sM0);
                    sM1 = fma(p, g1, sM1);
                }
            }
            if (OM == blk::OM_GAUSSIAN && P.use_rec) L.template step<BWD>();
        }
    }

    double *out = P.psum_out + (long long)b * NRED * P.nblk + tile;
    const double r0 = blk::block_sum(sN, red);
    if (tid == 0) out[0] = r0;
    if (BWD) {
        const double r1 = blk::block_sum(sS, red);
        const double r2 = blk::block_sum(sC, red);
        if (tid == 0) { out[1 * P.nblk] = r1; out[2 * P.nblk] = r2; }
    }
    if (BWD || P.means) {
        const double r3 = blk::block_sum(sM0, red);
        const double r4 = blk::block_sum(sM1, red);
        if (tid == 0) { out[3 * P.nblk] = r3; out[4 * P.nblk] = r4; }
    }
}

}  // namespace blf
