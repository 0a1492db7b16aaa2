// Persistent kernels for 1-D grids (configs C1 / C2: Poisson, GaussianMean, tabulated likelihoods on <= 8192 cells).
// EXPERIMENTAL, OFF BY DEFAULT (blhip_set_option("persist1d", 1)): measured on MI355X for config C2 (4096 cells, 41
// taps) it runs 12.9-14.5 us per step against 5.9 us for one launch of blk::step_kernel per step -- a single CU is
// fp64-bound at ~2.5 us for such a step (4096 x 41 x 2 fp64 ops / 64 lanes per clock), and the per-step dependent loads
// and barriers of one workgroup cost more than a kernel boundary.  Kept because it is parity-tested and is the right
// shape for hyper-studies over thousands of tiny 1-D chains.
//
// A 1-D distribution is a few KB: one launch per time step would be pure launch latency (~6 us per step measured).
// Here ONE WORKGROUP OWNS ONE CHAIN FOR THE WHOLE TIME LOOP: the state lives in LDS (double-buffered, with a mirrored
// halo so that the reflect boundary costs nothing in the tap loop), the step-to-step dependency is a workgroup barrier,
// and the only global traffic per step is the stored posterior row.  Chains of a hyper-study run concurrently, one per
// workgroup (256 CUs => hundreds of chains in flight).  Same math and summation order per cell as blk::step_kernel.
#pragma once
#include "blhip_kernels.hpp"

namespace bl1 {

using blk::NRED;
constexpr int NT = 1024;          // 16 waves per workgroup
constexpr int NW = NT / 64;
constexpr int MAXC = 8;           // cells per thread (grids up to 8192 cells)

struct P1Params {
    int n, T, B, LW, d, rec_len, store, means;
    const double *shared[5];      // [SRC_PRIOR], [SRC_RESET], [SRC_UNIFORM]
    double *post; long long post_stride;            // (B, T, n) or nullptr (evidence only)
    const unsigned char *srckind;                   // [T][B]
    const int *tap;                                 // [T][B] tap-set id of the (single) axis, -1 = identity
    const double *taps; const int *tap_off; const int *tap_lw;
    double *red_out;                                // [T][B][NRED] block-reduced sums (final: one block per chain)
    const double *m1, *colA, *rec, *lik;
};

__device__ __forceinline__ double block_sum(double v, double *red) {
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

// (re)load the LDS state from a shared source array, with mirrored halo
__device__ __forceinline__ void load_source(double *cur, const double *src, int n, int LW) {
    for (int j = threadIdx.x; j < n; j += NT) {
        const double v = src[j];
        cur[LW + j] = v;
        if (j < LW) cur[LW - 1 - j] = v;
        if (j >= n - LW) cur[LW + n + (n - 1 - j)] = v;
    }
    __syncthreads();
}

template <int OM, bool BWD>
__global__ __launch_bounds__(NT) void persist1d_kernel(const P1Params P) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int n = P.n, LW = P.LW, pitch = n + 2 * LW;
    double *cur = lds, *nxt = lds + pitch, *red = lds + 2 * pitch, *wl = red + 32;
    const int b = blockIdx.x;
    double *post = P.post ? P.post + (long long)b * P.post_stride : nullptr;
    double scale = 1.0;

    blk::StepParams Q{};
    Q.d = P.d; Q.n1 = n; Q.m0 = nullptr; Q.m1 = P.m1;

    for (int step = 0; step < P.T; ++step) {
        const int t = BWD ? P.T - 1 - step : step;
        const int kind = P.srckind[(long long)t * P.B + b];
        const int tp = P.tap[(long long)t * P.B + b];
        const int lw = tp >= 0 ? P.tap_lw[tp] : 0;
        const double *w = P.taps + (tp >= 0 ? P.tap_off[tp] : 0);
        if (kind != blk::SRC_PREV) {
            __syncthreads();
            load_source(cur, P.shared[kind], n, LW);
            scale = 1.0;
        }
        Q.rec = P.rec + (long long)t * P.rec_len;
        Q.lik = P.lik ? P.lik + (long long)t * n : nullptr;
        double *row = post ? post + (long long)t * n : nullptr;

        // stencil weights of this step -> LDS (broadcast reads), then taps outermost / cells innermost: the MAXC cells of
        // a thread are independent accumulator chains (per-cell order = SciPy's symmetric correlate1d)
        __syncthreads();
        for (int k = threadIdx.x; k <= lw; k += NT) wl[k] = lw > 0 ? w[k] : 1.0;
        __syncthreads();
        double o[MAXC];
#pragma unroll
        for (int q = 0; q < MAXC; ++q) {
            const int j = min(threadIdx.x + q * NT, n - 1);
            o[q] = cur[LW + j] * wl[0];
        }
#pragma unroll 2
        for (int k = lw; k >= 1; --k) {
            const double wk = wl[k];
#pragma unroll
            for (int q = 0; q < MAXC; ++q) {
                const int j = min(threadIdx.x + q * NT, n - 1);
                o[q] = fma(cur[LW + j - k] + cur[LW + j + k], wk, o[q]);
            }
        }

        double sN = 0.0, sS = 0.0, sC = 0.0, sM = 0.0;
#pragma unroll
        for (int q = 0; q < MAXC; ++q) {
            const int j = threadIdx.x + q * NT;
            if (j < n) {
                const double g1 = P.m1[j];
                const double cA = (OM == blk::OM_POISSON) ? P.colA[j] : 0.0;
                const double L = blk::likelihood<OM>(Q, 0, j, cA, 0.0, g1);
                double ns;
                if (!BWD) {
                    const double a = o[q] * scale * L;
                    ns = a;
                    if (P.store) row[j] = a;
                    sN += a;
                    if (P.means) sM += a * g1;
                } else {
                    const double beta = o[q] * scale;
                    const double p = row[j] * beta;
                    row[j] = p;
                    ns = beta * L;
                    sN += p;
                    sS += p / L;                                 // 0/0 -> NaN as numpy (core.py:463)
                    sC += ns;
                    sM += p * g1;
                }
                nxt[LW + j] = ns;
                if (j < LW) nxt[LW - 1 - j] = ns;
                if (j >= n - LW) nxt[LW + n + (n - 1 - j)] = ns;
            }
        }
        double *out = P.red_out + ((long long)t * P.B + b) * NRED;
        const double rN = block_sum(sN, red);
        double norm = rN;
        if (BWD) {
            const double rS = block_sum(sS, red);
            const double rC = block_sum(sC, red);
            norm = rC;
            if (threadIdx.x == 0) { out[1] = rS; out[2] = rC; }
        }
        if (BWD || P.means) {
            const double rM = block_sum(sM, red);
            if (threadIdx.x == 0) out[3] = rM;
        }
        if (threadIdx.x == 0) out[0] = rN;
        scale = 1.0 / norm;
        double *tmp = cur; cur = nxt; nxt = tmp;             // block_sum's barriers order the LDS writes before the next reads
    }
}

}  // namespace bl1
