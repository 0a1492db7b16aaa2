// AXIS-1 PRE-PASS for random walks on the SECOND parameter that are wider than the fused step kernels' 8-column halo (gfx950).
//
// The launch-per-step kernels (blhip_fast.hpp, blhip_mfma.hpp) apply the axis-1 (row-wise) part of a separable Gaussian transition
// inside the block, from a halo of R1MAX = 8 columns per side.  A hyper-study whose random walk on the second parameter is wider
// than 8 grid steps -- the usual case as soon as that width is a hyper-parameter (reference: transitionModels.py:96-118,
// CombinedTransitionModel :632-662, tests/test_hyperstudy.py) -- used to drop the WHOLE batch to the generic LDS-tile kernel
// (blk::step_kernel), ~50 x slower.  Here the row filter runs as its own launch per step and the fused kernels then consume the
// filtered state with their own axis-1 part switched off:
//     hsrc[b] = H_b(source of chain b)           this kernel: reflect boundary (half-sample symmetric, scipy mode 'reflect'),
//                                                runtime radius <= HW_MAX, one launch per radius bucket of the step
//     state'  = epilogue(V_b(hsrc[b]))           fast / matrix-pipe step kernel with FastParams::hsrc set
// (axis 1 before axis 0; the reference filters axis 0 first, transitionModels.py:645-649: separable reflect-boundary filters commute
// exactly in real arithmetic, in floating point the results differ by rounding, ~1e-16, as in the fused kernels).
//
// Kernel: block = RB rows x CB columns of one chain.  The rows (+ radius + slack columns per side, reflected) are staged in LDS by
// coalesced loads; a thread then computes OC = 8 CONSECUTIVE outputs of two rows from a sliding set of inputs:
//     for every input x(m) of the 8 + 2 lw a row needs:  out[c] += W[m - c] x(m),  c = 0 .. 7
// i.e. one LDS read per 8 FMAs instead of two per FMA (the weights W are block-uniform: LDS broadcast reads, 15 per 128 FMAs).
// Lanes run along ROWS (odd LDS pitch: conflict-free), so results go back to the LDS tile and leave by coalesced stores.
// Cost per cell: 8 B read + 8 B written and 2 lw + 8 + (up to 7) FMAs.  Measured (profiles/r03_notes.md): 16.8 B of HBM traffic per cell
// (a batch's states are hundreds of MB: nothing of the previous kernel's output is still in a cache) at 4.2 TB/s -- the kernel runs at the
// memory roof; the host issues it per radius bucket on the bucket's stream, so that it overlaps another bucket's matrix-pipe-bound step.
#pragma once
#include <hip/hip_runtime.h>

#include "blhip_kernels.hpp"

namespace blh {

using blk::SRC_PREV;

constexpr int NT = 256;
constexpr int RB = 16;          // rows per block
constexpr int CB = 256;         // columns per block
constexpr int OC = 8;           // consecutive outputs per thread task
constexpr int HW_MAX = 256;     // largest radius (LDS: 16 rows x (256 + 2 x 256 + 15) doubles = 100 KB)

struct HParams {
    int n0, n1, tiles_j;
    int pitch;                   // LDS row pitch in doubles: odd, >= CB + 2 lwmax + 15
    int lwmax;                   // largest radius among the chains of the launch (sizes the LDS tile)
    const double *src; long long src_stride;
    const double *shared[5];
    const double *presrc;        // != nullptr: [chains][n0 * n1] the sources after an earlier pre-pass (axis 1 before axis 0), whatever their kind
    const int *chain_ids;        // [gridDim.y] -> chain of the batch (a bucket of the step: its launch follows on the same stream), or nullptr
    const unsigned char *srckind;
    const int *tap1;             // [chains] the chain's filter of THIS pass in the tap table (axis-1 taps for hwide_kernel, axis-0 taps for vwide_kernel), -1 = none
    const double *taps; const int *tap_off; const int *tap_lw;
    double *dst;                 // [chains][n0 * n1]
};

inline int pitch_for(int lwmax) { return (CB + 2 * lwmax + 15) | 1; }
inline size_t lds_bytes(int lwmax) { return ((size_t)RB * pitch_for(lwmax) + 2 * (size_t)lwmax + 32) * sizeof(double); }

__device__ __forceinline__ int reflect1(int i, int n) {      // single-period half-sample reflection (host: radius < n), clamped beyond
    i = i < 0 ? -1 - i : i;
    i = i >= n ? 2 * n - 1 - i : i;
    return min(max(i, 0), n - 1);
}

static __global__ __launch_bounds__(NT) void hwide_kernel(const HParams P) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    const int b = P.chain_ids ? P.chain_ids[blockIdx.y] : (int)blockIdx.y;
    const int ti = blockIdx.x / P.tiles_j, tj = blockIdx.x - ti * P.tiles_j;
    const int i0 = ti * RB, j0 = tj * CB;
    const int rows = min(RB, P.n0 - i0), cols = min(CB, P.n1 - j0);
    const int kind = P.srckind[b];
    const int t1 = P.tap1[b];
    const double *src = (P.presrc ? P.presrc + (long long)b * P.n0 * P.n1 : (kind == SRC_PREV ? P.src + (long long)b * P.src_stride : P.shared[kind])) + (long long)i0 * P.n1;
    double *dst = P.dst + (long long)b * P.n0 * P.n1 + (long long)i0 * P.n1;

    if (t1 < 0) {                                    // no axis-1 filter for this chain at this step: the fused kernel still reads hsrc
        for (int e = tid; e < rows * CB; e += NT) {
            const int r = e / CB, c = e - r * CB;
            if (c < cols) dst[(long long)r * P.n1 + j0 + c] = src[(long long)r * P.n1 + j0 + c];
        }
        return;
    }
    const int lw = P.tap_lw[t1];
    const double *w = P.taps + P.tap_off[t1];
    const int pitch = P.pitch;
    const int groups = (2 * lw + OC + 7) / 8;        // input offsets m'' = 0 .. 8 groups - 1 relative to (first output - lw)
    const int width = CB - OC + 8 * groups + OC;     // staged columns per row: outputs CB, inputs up to (CB - 8) + 8 groups + 7
    double *wt = lds + RB * pitch;                   // W[d], d = -lw-7 .. : wt[d + lw + 7] = w[|d|] (|d| <= lw) or 0
    for (int e = tid; e < 8 * groups + 16; e += NT) {
        const int d = e - lw - 7, a = d < 0 ? -d : d;
        wt[e] = a <= lw ? w[a] : 0.0;
    }
    {   // stage the rows: a flat element index per thread (NT < width: at most one row wrap per stride), loads in batches of 8
        const int total = rows * width;
        int r = 0, lc = tid;                          // element tid of the tile (width >= CB + 8 > NT)
        for (int e0 = tid; e0 < total; e0 += 8 * NT) {
            double v[8];
            int at[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                at[k] = r * pitch + lc;
                v[k] = (e0 + k * NT < total) ? src[(long long)r * P.n1 + reflect1(j0 - lw + lc, P.n1)] : 0.0;
                lc += NT;
                if (lc >= width) { lc -= width; ++r; }
                if (r >= rows) r = rows - 1;          // (past the end: a valid address, the value is not stored)
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (e0 + k * NT < total) lds[at[k]] = v[k];
        }
    }
    __syncthreads();

    // a thread: OC consecutive outputs of TWO rows (row pair rp, rp + RB / 2) of one column block -- the block-uniform weights it reads
    // from LDS serve both rows; the next group's operands are requested before this group's 128 FMAs
    static_assert(NT == (RB / 2) * (CB / OC), "one task per thread");
    const int rp = tid % (RB / 2), cb = tid / (RB / 2);
    double out[2][OC];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int c = 0; c < OC; ++c) out[h][c] = 0.0;
    }
    if (cb * OC < cols) {
        const double *x0 = lds + rp * pitch + cb * OC, *x1 = x0 + (RB / 2) * pitch;
        double xa[8], xb[8], wv[15];
#pragma unroll
        for (int r = 0; r < 8; ++r) { xa[r] = x0[r]; xb[r] = x1[r]; }
#pragma unroll
        for (int k = 0; k < 15; ++k) wv[k] = wt[k];
        for (int g = 0; g < groups; ++g) {
            double na[8], nb[8], nw[15];
            const int gn = g + 1 < groups ? g + 1 : g;             // (last group: re-reads itself, unused)
#pragma unroll
            for (int r = 0; r < 8; ++r) { na[r] = x0[8 * gn + r]; nb[r] = x1[8 * gn + r]; }
#pragma unroll
            for (int k = 0; k < 15; ++k) nw[k] = wt[8 * gn + k];    // W[m' - c] for m'' = 8 g + r: index 8 g + r - c + 7
#pragma unroll
            for (int r = 0; r < 8; ++r) {
#pragma unroll
                for (int c = 0; c < OC; ++c) {
                    out[0][c] = fma(wv[r - c + 7], xa[r], out[0][c]);
                    out[1][c] = fma(wv[r - c + 7], xb[r], out[1][c]);
                }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) { xa[r] = na[r]; xb[r] = nb[r]; }
#pragma unroll
            for (int k = 0; k < 15; ++k) wv[k] = nw[k];
        }
    }
    __syncthreads();                                  // every input has been read: the results overwrite the tile
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int c = 0; c < OC; ++c) lds[(rp + h * (RB / 2)) * pitch + cb * OC + c] = out[h][c];
    }
    __syncthreads();
    for (int e = tid; e < rows * CB; e += NT) {
        const int r = e / CB, c = e - r * CB;
        if (c < cols) dst[(long long)r * P.n1 + j0 + c] = lds[r * pitch + c];
    }
}

// ---- the same for the FIRST parameter: random walks wider than the matrix-pipe kernels' largest band (radius 40) ---------------------
// A hyper-study over a random-walk width on a fine grid (1024 rows and the widths of C4: radius 77) used to fall back to the generic
// kernel as a whole.  vwide_kernel filters the COLUMNS of a tile (RV rows x CV columns + radius rows above and below, reflected at the
// grid's first / last row) with the same sliding scheme -- a thread computes 8 consecutive ROWS of one column, twice -- and the streaming
// kernel then runs with no stencil at all.  Lanes run along columns (coalesced loads and stores, conflict-free LDS).  Rows above and
// below a tile are staged again by its neighbours: (RV + 2 lw) / RV reads per cell (2.2 x at radius 77), mostly from L2 -- the tiles of
// a column block are consecutive block indices.
constexpr int RV = 128;         // rows per block
constexpr int CV = 32;          // columns per block
constexpr int PV = CV + 1;      // LDS pitch (odd)
constexpr int VW_MAX = 128;     // largest radius (LDS: (128 + 256 + 15) x 33 doubles = 105 KB)

inline size_t vlds_bytes(int lwmax) { return ((size_t)(RV + 2 * lwmax + 15) * PV + 2 * (size_t)lwmax + 32) * sizeof(double); }

static __global__ __launch_bounds__(NT) void vwide_kernel(const HParams P) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    const int b = P.chain_ids ? P.chain_ids[blockIdx.y] : (int)blockIdx.y;
    const int tiles_i = (P.n0 + RV - 1) / RV;
    const int tj = blockIdx.x / tiles_i, ti = blockIdx.x - tj * tiles_i;      // (row tiles of a column block: consecutive blocks)
    const int i0 = ti * RV, j0 = tj * CV;
    const int rows = min(RV, P.n0 - i0), cols = min(CV, P.n1 - j0);
    const int kind = P.srckind[b];
    const int t0 = P.tap1[b];
    const double *src = P.presrc ? P.presrc + (long long)b * P.n0 * P.n1 : (kind == SRC_PREV ? P.src + (long long)b * P.src_stride : P.shared[kind]);
    double *dst = P.dst + (long long)b * P.n0 * P.n1;

    if (t0 < 0) {                                    // no axis-0 filter for this chain at this step: the streaming kernel still reads dst
        for (int e = tid; e < rows * CV; e += NT) {
            const int r = e / CV, c = e - r * CV;
            if (c < cols) dst[(long long)(i0 + r) * P.n1 + j0 + c] = src[(long long)(i0 + r) * P.n1 + j0 + c];
        }
        return;
    }
    const int lw = P.tap_lw[t0];
    const double *w = P.taps + P.tap_off[t0];
    const int groups = (2 * lw + OC + 7) / 8;
    const int height = RV - OC + 8 * groups;         // staged rows: i0 - lw .. i0 - lw + height - 1
    double *wt = lds + (RV + 2 * P.lwmax + 15) * PV;
    for (int e = tid; e < 8 * groups + 16; e += NT) {
        const int d = e - lw - 7, a = d < 0 ? -d : d;
        wt[e] = a <= lw ? w[a] : 0.0;
    }
    {
        const int total = height * CV;
        for (int e0 = tid; e0 < total; e0 += 8 * NT) {
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = min(e0 + k * NT, total - 1), lr = e / CV, lc = e - lr * CV;
                v[k] = src[(long long)reflect1(i0 - lw + lr, P.n0) * P.n1 + min(j0 + lc, P.n1 - 1)];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = e0 + k * NT, lr = e / CV, lc = e - lr * CV;
                if (e < total) lds[lr * PV + lc] = v[k];
            }
        }
    }
    __syncthreads();

    // a thread: OC consecutive ROWS of one column, for two row groups (rg, rg + RV / 16)
    static_assert(2 * NT == (RV / OC) * CV, "two tasks per thread");
    const int col = tid % CV, rg = tid / CV;
    double out[2][OC];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int c = 0; c < OC; ++c) out[h][c] = 0.0;
    }
    if (col < cols) {
        const double *x0 = lds + (rg * OC) * PV + col, *x1 = x0 + (RV / 2) * PV;
        double xa[8], xb[8], wv[15];
#pragma unroll
        for (int r = 0; r < 8; ++r) { xa[r] = x0[r * PV]; xb[r] = x1[r * PV]; }
#pragma unroll
        for (int k = 0; k < 15; ++k) wv[k] = wt[k];
        for (int g = 0; g < groups; ++g) {
            double na[8], nb[8], nw[15];
            const int gn = g + 1 < groups ? g + 1 : g;
#pragma unroll
            for (int r = 0; r < 8; ++r) { na[r] = x0[(8 * gn + r) * PV]; nb[r] = x1[(8 * gn + r) * PV]; }
#pragma unroll
            for (int k = 0; k < 15; ++k) nw[k] = wt[8 * gn + k];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
#pragma unroll
                for (int c = 0; c < OC; ++c) {
                    out[0][c] = fma(wv[r - c + 7], xa[r], out[0][c]);
                    out[1][c] = fma(wv[r - c + 7], xb[r], out[1][c]);
                }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) { xa[r] = na[r]; xb[r] = nb[r]; }
#pragma unroll
            for (int k = 0; k < 15; ++k) wv[k] = nw[k];
        }
    }
    // a thread's results leave directly: lanes are consecutive columns (a wave stores two rows of 32 contiguous doubles per instruction)
    if (col < cols) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int c = 0; c < OC; ++c) {
                const int r = rg * OC + h * (RV / 2) + c;
                if (r < rows) dst[(long long)(i0 + r) * P.n1 + j0 + col] = out[h][c];
            }
        }
    }
}

}  // namespace blh
