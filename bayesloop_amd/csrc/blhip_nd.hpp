// Grids with 3 and 4 parameters (reference: core.py:130-176 builds the meshgrid over ANY number of observation-model parameters;
// transitionModels.py:107-111 filters any axis).  Only the reference's plug-in models (SciPy / SymPy / NumPy, likelihood tables
// evaluated by the caller) have more than two parameters, so this path is the plain formulation, not a tuned one: per time step
// one separable-filter pass per random walk (the array viewed as (outer, n, inner) around the filtered axis) and ONE fused
// elementwise kernel (lazy normaliser, likelihood, sums).  Algorithmic HBM traffic per cell and step: 16 B per filter pass + 24 B
// forward (filtered state, likelihood, new state; + 8 with posterior storage) / 40 B backward.
#pragma once
#include "blhip_kernels.hpp"

namespace bln {

using blk::NRED;
using blk::NTHREADS;
using blk::SRC_PREV;

constexpr int MAXD = 4;

struct NdGrid {
    int ndim;
    int n[MAXD];
    long long stride[MAXD];      // cells between neighbours along axis k (C order: the last parameter is contiguous)
    long long G;
    const double *m[MAXD];       // marginal grids (device)
};

// multi-period half-sample reflection (scipy.ndimage mode='reflect': ... b a | a b c ...), any offset
__device__ __forceinline__ int reflect_any(long long i, int n) {
    const long long p = 2ll * n;
    i %= p;
    if (i < 0) i += p;
    return (int)(i < n ? i : p - 1 - i);
}

// One axis of the separable transition (scipy.ndimage.gaussian_filter1d -> correlate1d, symmetric weights, pairs added from the
// OUTERMOST one inward: SURVEY 8 a-5).  blockIdx.y = chain of the batch; a chain without a kernel on this axis copies.
//   srcs[b]: where chain b's input lives (its state, or a shared distribution at a restart); dst: [B][G]
static __global__ __launch_bounds__(NTHREADS) void filter_axis_kernel(double *dst, const double *const *srcs, long long G, int n, long long inner,
                                                                const int *tap_id, const double *taps, const int *tap_off, const int *tap_lw) {
    const int b = blockIdx.y;
    const double *src = srcs[b];
    double *out = dst + (long long)b * G;
    const int id = tap_id[b];
    const int lw = id >= 0 ? tap_lw[id] : 0;
    const double *w = taps + (id >= 0 ? tap_off[id] : 0);
    for (long long e = (long long)blockIdx.x * NTHREADS + threadIdx.x; e < G; e += (long long)gridDim.x * NTHREADS) {
        if (lw == 0) { out[e] = src[e]; continue; }
        const int i = (int)((e / inner) % n);
        const long long base = e - (long long)i * inner;
        double acc = src[e] * w[0];
        for (int j = lw; j >= 1; --j)
            acc += (src[base + (long long)reflect_any((long long)i - j, n) * inner] + src[base + (long long)reflect_any((long long)i + j, n) * inner]) * w[j];
        out[e] = acc;
    }
}

struct NdStep {
    NdGrid g;
    int B, T, nblk;
    const double *const *srcs;   // [B] the step's (filtered) input per chain
    const unsigned char *kind;   // [B] source kind of the step: SRC_PREV -> the lazy normaliser applies
    const double *psum_prev; int prev_slot;    // [B][NRED][nblk] partial sums of the producing step
    double *psum_out;            // [B][NRED][nblk]
    const double *lik;           // (G) likelihood of this step
    double *state;               // [B][G] new state (forward: a_t; backward: c_t = beta_t L_t)
    double *post; long long post_stride;       // forward: stored a_t (may be null); backward: stored a_t in -> posterior out
};

// slots of the partial sums: 0 N, 1 S = sum p / L (backward), 2 C (backward), 3 + k = sum p * grid_k
template <bool BWD>
__global__ __launch_bounds__(NTHREADS) void step_kernel(const NdStep P) {
    __shared__ double red[(NTHREADS / 64) * 7 + 1];
    const int b = blockIdx.y;
    const long long G = P.g.G;
    double scale = 1.0;
    if (P.kind[b] == SRC_PREV) {
        const double *pp = P.psum_prev + ((long long)b * NRED + P.prev_slot) * P.nblk;
        double v = 0.0;
        for (int k = threadIdx.x; k < P.nblk; k += NTHREADS) v += pp[k];
        scale = 1.0 / blk::block_sum(v, red);
    }
    const double *src = P.srcs[b];
    double *st = P.state + (long long)b * G;
    double *po = P.post ? P.post + (long long)b * P.post_stride : nullptr;
    double s[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (long long e = (long long)blockIdx.x * NTHREADS + threadIdx.x; e < G; e += (long long)gridDim.x * NTHREADS) {
        const double L = P.lik[e];
        const double x = src[e] * scale;
        double p;
        if (!BWD) {
            p = x * L;                                           // core.py:382
            st[e] = p;
            if (po) po[e] = p;
        } else {
            p = po[e] * x;                                       // core.py:436
            const double cn = x * L;                             // core.py:467 (the transition is applied by the next step's passes)
            po[e] = p;
            st[e] = cn;
            s[1] += p / L;                                       // core.py:463 (0 / 0 -> NaN, as there)
            s[2] += cn;
        }
        s[0] += p;
#pragma unroll
        for (int k = 0; k < MAXD; ++k)
            if (k < P.g.ndim) s[3 + k] += p * P.g.m[k][(e / P.g.stride[k]) % P.g.n[k]];
    }
    blk::block_sums<7, NTHREADS / 64>(s, red);
    if (threadIdx.x == 0) {
        double *out = P.psum_out + (long long)b * NRED * P.nblk + blockIdx.x;
#pragma unroll
        for (int k = 0; k < 7; ++k) out[(long long)k * P.nblk] = s[k];
    }
}

// per-step sums of an (T, G) array times the grid values: out[t][0] = sum A, out[t][1 + k] = sum A grid_k  (partials per block)
static __global__ __launch_bounds__(NTHREADS) void row_stats_kernel(const double *A, const NdGrid g, double *partial) {
    __shared__ double red[(NTHREADS / 64) * 5 + 1];
    const long long t = blockIdx.y;
    double s[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (long long c = (long long)blockIdx.x * NTHREADS + threadIdx.x; c < g.G; c += (long long)gridDim.x * NTHREADS) {
        const double v = A[t * g.G + c];
        s[0] += v;
#pragma unroll
        for (int k = 0; k < MAXD; ++k)
            if (k < g.ndim) s[1 + k] += v * g.m[k][(c / g.stride[k]) % g.n[k]];
    }
    blk::block_sums<5, NTHREADS / 64>(s, red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) partial[((t * 5) + k) * gridDim.x + blockIdx.x] = s[k];
    }
}

}  // namespace bln
