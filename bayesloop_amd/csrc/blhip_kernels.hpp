// Device code of libblhip: fused forward / backward step kernels of the grid-based forward-backward recursion
// (reference: bayesloop/core.py:372-411 forward, :434-470 backward) for gfx950 / MI355X.  fp64 throughout.
//
// Data layout in HBM: every distribution is a dense C-order array (n0, n1) of doubles, n1 (the LAST observation-model
// parameter) fastest; 1-D grids are (1, n).  A batch of chains is an outer dimension with an explicit stride.
//
// Formulation (see DESIGN.md "lazy normalisation"): the state written by step t is the UNNORMALISED product
//     a_t = T(a_{t-1}) * (1 / sum a_{t-1}) * L_t
// so one step reads the previous state once (plus stencil halo, served by L2) and writes the new state once
// (16 B / cell); the normaliser of step t is produced by step t as per-block partial sums and consumed lazily by
// step t+1.  The transition is linear, so filtering before normalising equals the reference's order to rounding.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace blk {

constexpr int NTHREADS = 256;
constexpr int NRED = 7;          // partial slots per (step, chain): 0 N  1 S(p/L) [fwd: U]  2 C  3 M0  4 M1  5 B (clamp only)
                                 //                                  6 MAX of the new state (clamp batches only: NotEqual)

enum Mode { MODE_FWD = 0, MODE_BWD = 1, MODE_FILTER = 2 };
enum SrcKind { SRC_PREV = 0, SRC_PRIOR = 1, SRC_RESET = 2, SRC_UNIFORM = 3, SRC_INDEP = 4 };

struct StepParams {
    // geometry (internal axes: 0 = rows (slow), 1 = cols (fast))
    int n0, n1;
    int TI, TJ;                  // output tile
    int LW0, LW1;                // launch-wide maximal filter radii (LDS geometry)
    int tiles_i, tiles_j, nblk;
    int ndim;                    // 1 or 2 observation-model parameters
    int d;                       // data dimensions per step
    int rec_len;                 // doubles per step record
    // state
    const double *src;  long long src_stride;     // SRC_PREV source, per-chain stride (doubles)
    double       *dst;  long long dst_stride;     // new state (FWD: a_t ; BWD: c_i ; FILTER: filtered)
    double       *post; long long post_stride;    // BWD: stored alpha_i (in) -> posterior_i (out), per-chain stride
    const double *shared[5];                      // [SRC_PRIOR], [SRC_RESET], [SRC_UNIFORM], [SRC_INDEP] shared sources (G)
    // per-chain metadata of this step (pre-offset to the step)
    const unsigned char *srckind;                 // [B]
    const int *tap0, *tap1;                       // [B] tap-set ids per internal axis, -1 = identity
    const unsigned char *cmode; const double *limit;   // [B] clamp mode: 0 none, 1 RegimeSwitch on the source, 2 RegimeSwitch after the stencil,
                                                       //     3 NotEqual (invert + clamp the source), 4 dense 2-D kernel tap0 with zero boundary, renormalised
                                                       //     (BivariateRandomWalk), 5 separable taps with zero boundary, renormalised (AlphaStableRandomWalk);
                                                       //     6 cubic-spline shift of one axis (asymmetric taps, spline boundary rule), renormalised (Deterministic);
                                                       //     nullptr if the batch has none
    // tap table
    const double *taps; const int *tap_off; const int *tap_lw;
    const int *tap_lw2;          // dense 2-D kernels (clamp mode 4): axis-1 radius; weights at taps[off + a * (2 lw2 + 1) + b]
    // lazy normalisation
    const double *psum_prev; int prev_slot; int prev_nblk;   // partials of the producing step [B][NRED][prev_nblk]
    double *psum_out;                                         // [B][NRED][nblk]
    // likelihood
    const double *m0, *m1;       // marginal grids along internal axes (m0 unused for 1-D)
    const double *colA, *colB;   // per-column tables (model specific)
    const double *rec;           // step record (rec_len doubles)
    const double *lik;           // OM_TABLE: likelihood of this step (G)
    int chains;                  // B
};

__device__ __forceinline__ int reflect(int i, int n) {
    // half-sample symmetric extension with period 2n (SciPy NI_EXTEND_REFLECT), any offset
    if ((unsigned)i < (unsigned)n) return i;
    const int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i >= n ? p - 1 - i : i;
}

// Boundary rules of the tile loader per axis: 0 reflect (GaussianRandomWalk), 1 zero (Bivariate / AlphaStable), 2 the
// extension scipy.ndimage.shift(order=3, mode='nearest') works on: 12 edge samples, then half-sample reflection of that
// padded array (Deterministic; oracle/bl_oracle.py: spline_shift_nearest).  -1 = outside with zero fill.
__device__ __forceinline__ int extend_index(int i, int n, int rule) {
    if ((unsigned)i < (unsigned)n) return i;
    if (rule == 1) return -1;
    if (rule == 2) return min(max(reflect(i + 12, n + 24) - 12, 0), n - 1);
    return reflect(i, n);
}

// The both-axes chain-resident kernels (blhip_chainax.hpp) keep what belongs to time t -- stored state, posterior, partial accumulator -- in
// one of two strip-major layouts on their SQUARE geometry of n0p rows = columns, alternating with t:
//   layout A (odd t):  [strip = column / 16][row][column % 16]          (chain_kernel's)
//   layout B (even t): [strip = row / 16][column][row % 16]             (the transposed one)
__host__ __device__ __forceinline__ bool ax_layout_b(int t) { return (t & 1) == 0; }
// doubles from the beginning of a time step's slice to cell (row, col); ax = 0: always layout A (every other chain-resident kernel)
__device__ __forceinline__ long long strip_major_index(int row, int col, int n0p, int t, int ax) {
    return (ax && ax_layout_b(t)) ? ((long long)(row >> 4) * n0p + col) * 16 + (row & 15) : ((long long)(col >> 4) * n0p + row) * 16 + (col & 15);
}

// Wave-wide sum on the DPP cross-lane path (no LDS round trips): 4 row_shr steps inside each row of 16 lanes, then
// row_bcast:15 / row_bcast:31 fold the four rows; the total lands in lane 63 and is broadcast through an SGPR.
// (__shfl_down on a double is 2 ds_bpermute_b32 per step: ~1200 cycles per sum vs ~100 here -- per block that was 1-2 us
// of serial tail for the 5 sums of a step kernel.)  Fixed summation tree => deterministic.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
    // (every row enabled: bound_ctrl gives the lanes without a source their 0 -- no `old` operand to materialise, two v_mov less per step
    //  of the reduction: 24 of the two-chain fold kernel's ~580 vector instructions per chain-step)
    constexpr bool BC = ROW_MASK == 0xf;
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, BC);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, BC);
    return v + __hiloint2double(hi, lo);             // lanes without a source (or outside ROW_MASK) add 0.0
}

__device__ __forceinline__ double wave_sum(double v) {
    v = dpp_add<0x111, 0xf>(v);                      // row_shr:1
    v = dpp_add<0x112, 0xf>(v);                      // row_shr:2
    v = dpp_add<0x114, 0xf>(v);                      // row_shr:4
    v = dpp_add<0x118, 0xf>(v);                      // row_shr:8   -> lane 15 of every row = row total
    v = dpp_add<0x142, 0xa>(v);                      // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);                      // row_bcast:31 into rows 2 and 3 -> lane 63 = wave total
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// ---- SciPy's cubic-spline prefilter as the recursion it is (transitionModels.py:581, :600 -> scipy.ndimage.shift(order = 3,
// mode = 'nearest') -> ni_splines.c: apply_filter with _init_causal_reflect / _init_anticausal_reflect; restated bit for bit in
// oracle/spline_iir.c) ------------------------------------------------------------------------------------------------------------------
//     c *= gain;  c[0] <- c0 + z / (1 - z^2N) sum_i z^i (c[i] + z^N c[N-1-i]);  c[i] += z c[i-1];  c[N-1] *= z / (z - 1);  c[i] = z (c[i+1] - c[i])
// Rounds 1 - 4 applied its impulse response truncated at 34 cells (below fp64 resolution wherever the row has mass).  But the
// recursion's tail A z^k keeps decaying -- alternating in sign -- down to the denormals, and that tail is ALL that is left of a
// distribution shifted off the grid, which the reference then renormalises (:603): its sign decides whether a chain of the reference's
// published break-point study stops (14 of 23 400 do; the truncated response stopped 3).
// One wave runs the two first-order recursions over the row v[0 .. N) in LDS as segmented scans: a lane owns a chunk of C (odd: the
// lanes' accesses fall into different banks) consecutive elements, runs the recursion over it with a zero carry-in, the carries are
// resolved by a 6-step DPP scan of the affine maps (carry -> z^len carry + end value) across the lanes, and the recursion runs again
// from the true carry-in: every element is produced by SciPy's own operation  c[i] + z c[i-1]  /  z (c[i+1] - c[i])  from a carry that
// is exact to rounding.  Four sweeps of C dependent multiply-adds + two scans.  Call with ALL lanes of ONE wave; v holds gain x row.
constexpr double SPLINE_POLE = -0.2679491924311227;            // SciPy's literal (sqrt(3.) - 2. in double arithmetic is two ulp away)
constexpr double SPLINE_GAIN = (1.0 - SPLINE_POLE) * (1.0 - 1.0 / SPLINE_POLE);
__device__ __forceinline__ double pow_pole(int k) {           // z^k, k >= 0
    const double m = pow(-SPLINE_POLE, (double)k);
    return (k & 1) ? -m : m;
}
// the value of this lane's DPP source lane; `ident` where there is none (row_shr beyond the row's first lanes, rows outside ROW_MASK)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_src(double v, double ident) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(ident), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(ident), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
// inclusive scan, low lanes first, of the affine maps  carry -> A carry + B  (composition: this lane's map after the lower lanes'):
// four row_shr steps inside the rows of 16 lanes, row_bcast:15 into rows 1 / 3, row_bcast:31 into rows 2 / 3 -- no LDS round trips
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void affine_step(double &A, double &B) {
    const double Ap = dpp_src<CTRL, ROW_MASK>(A, 1.0), Bp = dpp_src<CTRL, ROW_MASK>(B, 0.0);
    B = fma(A, Bp, B);
    A *= Ap;
}
__device__ __forceinline__ void affine_scan(double &A, double &B) {
    affine_step<0x111, 0xf>(A, B);
    affine_step<0x112, 0xf>(A, B);
    affine_step<0x114, 0xf>(A, B);
    affine_step<0x118, 0xf>(A, B);
    affine_step<0x142, 0xa>(A, B);
    affine_step<0x143, 0xc>(A, B);
}
// (what depends on the row length and the lane only -- three pow() calls -- is computed once per kernel into 4 x 64 doubles of LDS)
// causal pass: lane l owns chunk l; anti-causal pass: lane l owns chunk 63 - l (so that both scans run low lanes first)
constexpr int SPLINE_CONST_DOUBLES = 4 * 64;
__device__ __forceinline__ int spline_chunk(int N) { return ((N + 63) >> 6) | 1; }       // odd: the lanes' accesses fall into different banks
__device__ __forceinline__ void spline_consts(double *K, int N, int lane) {
    const int C = spline_chunk(N);
    const int b0F = min(lane * C, N), b1F = min(b0F + C, N), b0B = min((63 - lane) * C, N), b1B = min(b0B + C, N);
    K[lane] = pow_pole(N); K[64 + lane] = pow_pole(b1F - b0F); K[128 + lane] = pow_pole(b1B - b0B); K[192 + lane] = pow_pole(b0F);   // (z^len = 1 for the empty chunks behind the row)
}
struct SplineK { double z_n, zlenF, zlenB, zb0; };       // z^N, z^(length of the lane's causal / anti-causal chunk), z^(first index of the causal chunk)
__device__ __forceinline__ SplineK spline_k_compute(int N, int lane) {
    const int C = spline_chunk(N);
    const int b0F = min(lane * C, N), b1F = min(b0F + C, N), b0B = min((63 - lane) * C, N), b1B = min(b0B + C, N);
    return SplineK{pow_pole(N), pow_pole(b1F - b0F), pow_pole(b1B - b0B), pow_pole(b0F)};
}
__device__ __forceinline__ SplineK spline_k_load(const double *Kc, int lane) { return SplineK{Kc[lane], Kc[64 + lane], Kc[128 + lane], Kc[192 + lane]}; }
__device__ __forceinline__ void spline_prefilter_wave(double *v, int N, int lane, const SplineK Kd) {
    const double z = SPLINE_POLE;
    if (N < 2) return;
    struct { double z_n, zlenF, zlenB, zb0; int b0F, b1F, b0B, b1B; } K;
    {
        const int C = spline_chunk(N);
        K.b0F = min(lane * C, N); K.b1F = min(K.b0F + C, N); K.b0B = min((63 - lane) * C, N); K.b1B = min(K.b0B + C, N);
        K.z_n = Kd.z_n; K.zlenF = Kd.zlenF; K.zlenB = Kd.zlenB; K.zb0 = Kd.zb0;
    }
    const double z_n = K.z_n;
    // ---- causal initialisation (the whole row enters c[0]) + the chunk's causal recursion with a zero carry-in, one sweep --------------
    double A = K.zlenF, B = 0.0;
    {
        double zi = K.zb0, part = 0.0;
        for (int i = K.b0F; i < K.b1F; ++i) {
            const double x = v[i];
            part += zi * (x + z_n * v[N - 1 - i]);
            zi *= z;
            B = x + z * B;
        }
        const double S = wave_sum(part);
        if (lane == 0) {                              // c[0] <- c0 + z / (1 - z^2N) S: the chunk's end value moves by z^(len - 1) times the change
            const double delta = S * (z / (1.0 - z_n * z_n));
            B = fma(K.zlenF / z, delta, B);
            v[0] = delta + v[0];
        }
    }
    // ---- causal pass: y[i] = c[i] + z y[i-1] from the true carry-in ---------------------------------------------------------------------
    affine_scan(A, B);
    double t = dpp_src<0x138, 0xf>(B, 0.0);           // wave_shr:1 -- the true value in front of the chunk (lane 0: none)
    for (int i = K.b0F; i < K.b1F; ++i) { t = v[i] + z * t; v[i] = t; }
    __builtin_amdgcn_wave_barrier();
    // ---- anti-causal pass: u[N-1] = y[N-1] z / (z - 1);  u[i] = z (u[i+1] - y[i]) --------------------------------------------------------
    const bool top = K.b0B < N && K.b1B == N;         // the lane whose chunk holds the last element: the recursion starts there
    const double uN1 = v[N - 1] * (z / (z - 1.0));
    const int hi = top ? K.b1B - 2 : K.b1B - 1;
    A = top ? 0.0 : K.zlenB; B = top ? uN1 : 0.0;
    for (int i = hi; i >= K.b0B; --i) B = z * (B - v[i]);
    affine_scan(A, B);
    t = dpp_src<0x138, 0xf>(B, 0.0);                   // the true value behind the chunk
    if (top) { t = uN1; v[N - 1] = uN1; }
    for (int i = hi; i >= K.b0B; --i) { t = z * (t - v[i]); v[i] = t; }
}


// Sum over the block; result valid in every thread.  `red` = NTHREADS/64 doubles of LDS scratch (+1).
__device__ __forceinline__ double block_sum(double v, double *red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    double s = red[0];
#pragma unroll
    for (int k = 1; k < NTHREADS / 64; ++k) s += red[k];
    return s;
}

// NV sums over a block of NW waves with two barriers in total; results valid in every thread; red = NV * NW doubles.
template <int NV, int NW>
__device__ __forceinline__ void block_sums(double (&v)[NV], double *red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_sum(v[k]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) red[w * NV + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double s = red[k];
#pragma unroll
        for (int q = 1; q < NW; ++q) s += red[q * NV + k];
        v[k] = s;
    }
}

// Block-wide maximum (clamp batches only, not on the hot path); result valid in every thread.
__device__ __forceinline__ double block_max(double v, double *red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    double s = red[0];
#pragma unroll
    for (int k = 1; k < NTHREADS / 64; ++k) s = fmax(s, red[k]);
    return s;
}

// Deterministic sum of n partials (fixed order for a fixed block size).
__device__ __forceinline__ double sum_partials(const double *p, int n, double *red) {
    double v = 0.0;
    for (int k = threadIdx.x; k < n; k += NTHREADS) v += p[k];
    return block_sum(v, red);
}

enum { OM_POISSON = 1, OM_GAUSSIAN = 2, OM_GAUSSIAN_MEAN = 3, OM_BERNOULLI = 4, OM_LAPLACE = 5, OM_WHITE_NOISE = 6, OM_AR1 = 7,
       OM_SCALED_AR1 = 8, OM_TABLE = 100 };

// Likelihood of the step's data segment at grid cell (i, j).  NaN data => factor 1 (observationModels.py:53-54).
template <int OM>
__device__ __forceinline__ double likelihood(const StepParams &P, int i, int j, double cA, double cB, double g1) {
    if constexpr (OM == OM_GAUSSIAN) {
        // exp(-(x-mu)^2 / (2 s^2) - 0.5 log(2 pi s^2))  observationModels.py:566-567; cA = 1/(2 s^2), cB = 0.5 log(2 pi s^2)
        const double mu = P.m0[i];
        double L = 1.0;
        for (int k = 0; k < P.d; ++k) {
            const double x = P.rec[k];
            if (x == x) {
                const double q = x - mu;
                L *= exp(-(q * q) * cA - cB);
            }
        }
        return L;
    } else if constexpr (OM == OM_GAUSSIAN_MEAN) {
        // observationModels.py:705-706; rec = [x, 1/(2 s^2), 0.5 log(2 pi s^2)]
        const double x = P.rec[0];
        if (x != x) return 1.0;
        const double q = x - g1;
        return exp(-(q * q) * P.rec[1] - P.rec[2]);
    } else if constexpr (OM == OM_POISSON) {
        // lambda^k exp(-lambda) / k!   observationModels.py:502; cA = exp(-lambda); rec = [k, k!] per dimension
        double L = 1.0;
        for (int k = 0; k < P.d; ++k) {
            const double cnt = P.rec[2 * k];
            if (cnt == cnt) L *= pow(g1, cnt) * cA / P.rec[2 * k + 1];
        }
        return L;
    } else {
        return P.lik[(long long)i * P.n1 + j];
    }
}

// One fused step for a batch of chains.  grid = (nblk, B), block = 256, dynamic LDS:
//   in_tile [(TI + 2 LW0)][pitch]  +  v_tile [TI][pitch]  + scratch,  pitch = TJ + 2 LW1
template <int OM, int MODE, bool MEANS>
__global__ __launch_bounds__(NTHREADS) void step_kernel(const StepParams P) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int b = blockIdx.y;
    const int blk = blockIdx.x;
    const int ti = blk / P.tiles_j, tj = blk - ti * P.tiles_j;
    const int i0 = ti * P.TI, j0 = tj * P.TJ;
    const int th = min(P.TI, P.n0 - i0), tw = min(P.TJ, P.n1 - j0);
    const int pitch = P.TJ + 2 * P.LW1;
    double *in_tile = lds;
    double *v_tile = lds + (size_t)(P.TI + 2 * P.LW0) * pitch;
    double *red = v_tile + (size_t)P.TI * pitch;

    const int kind = P.srckind[b];
    const int cm = P.cmode ? P.cmode[b] : 0;
    const double lim = P.cmode ? P.limit[b] : 0.0;
    const bool dense = cm == 4;                    // BivariateRandomWalk: tap0 is a dense (2 lw0 + 1) x (2 lw1 + 1) kernel
    // asymmetric tap sets (tap_lw2 == -1: full 2 lw + 1 weights, Deterministic spline shift) and the boundary rule per axis
    const bool asym0 = cm == 6 && P.tap0[b] >= 0 && P.tap_lw2[P.tap0[b]] < 0;
    const bool asym1 = cm == 6 && P.tap1[b] >= 0 && P.tap_lw2[P.tap1[b]] == -1;
    // tap_lw2 == -2: the two-stage form of the spline shift for |d| > 12 cells (1-D grids, the whole row in this block; blhip.hip:
    // TapTable::get_bigshift): prefilter the 12-sample-padded row, then evaluate the B-spline at the shifted coordinates with the
    // coefficient index clamped (scipy.ndimage.shift, mode='nearest'; oracle/bl_oracle.py: spline_shift_nearest)
    const bool big1 = cm == 6 && P.tap1[b] >= 0 && P.tap_lw2[P.tap1[b]] == -2;
    const int rule0 = (cm == 4 || cm == 5) ? 1 : (asym0 ? 2 : 0), rule1 = (cm == 4 || cm == 5) ? 1 : ((asym1 || big1) ? 2 : 0);
    const int t0 = P.tap0[b], t1 = dense ? -1 : P.tap1[b];
    const int lw0 = t0 >= 0 ? P.tap_lw[t0] : 0;
    const int lw1 = dense ? P.tap_lw2[t0] : (t1 >= 0 ? P.tap_lw[t1] : 0);
    const double *w0 = t0 >= 0 ? P.taps + P.tap_off[t0] : nullptr;   // w[0] = centre, w[k] = offset +-k (dense: row-major kernel)
    const double *w1 = t1 >= 0 ? P.taps + P.tap_off[t1] : nullptr;
    const double *src = kind == SRC_PREV ? P.src + (long long)b * P.src_stride : P.shared[kind];

    // thread mapping: XW columns x YN row groups (a single-row tile, i.e. a 1-D grid, uses all 256 threads on columns)
    const int XW = P.TI == 1 ? NTHREADS : 64, YN = NTHREADS / XW;
    const int x = threadIdx.x & (XW - 1), y = threadIdx.x / XW;

    // lazy normaliser of the producing step (every block sums the same partials in the same order)
    double scale = 1.0, kappa = 1.0;
    double ne_max = 0.0, ne_inv = 0.0;             // NotEqual: (max - x) / (G max - sum x) of the producing step's state
    if (MODE != MODE_FILTER && kind == SRC_PREV) {
        const double s = sum_partials(P.psum_prev + ((long long)b * NRED + P.prev_slot) * P.prev_nblk, P.prev_nblk, red);
        scale = 1.0 / s;
        if (cm == 3) {
            const double *pm = P.psum_prev + ((long long)b * NRED + 6) * P.prev_nblk;
            double m = -1.0;
            for (int k = threadIdx.x; k < P.prev_nblk; k += NTHREADS) m = fmax(m, pm[k]);
            ne_max = block_max(m, red);
            ne_inv = 1.0 / ((double)P.n0 * (double)P.n1 * ne_max - s);     // transitionModels.py:465-466 (0/0 -> NaN as numpy)
        }
        if (MODE == MODE_BWD && P.cmode) {
            // RegimeSwitch clamps F(beta_norm * L) (transitionModels.py:405-407): that needs 1 / sum(beta) of the producing
            // step (slot 5); products keep the well-scaled normaliser 1 / sum(c) (slot 2):  beta_used = u * kappa
            const double sb = sum_partials(P.psum_prev + ((long long)b * NRED + 5) * P.prev_nblk, P.prev_nblk, red);
            kappa = sb * scale;
            scale = 1.0 / sb;
        }
    }

    // ---- phase 1: source tile + halo -> LDS (reflect boundary resolved here; a source clamp is applied here too) ------
    for (int r = P.LW0 - lw0 + y; r < P.LW0 + th + lw0; r += YN) {
        const int gi = extend_index(i0 - P.LW0 + r, P.n0, rule0);
        const double *row = src + (long long)max(gi, 0) * P.n1;
        double *dstrow = in_tile + (size_t)r * pitch;
        for (int c = P.LW1 - lw1 + x; c < P.LW1 + tw + lw1; c += XW) {
            const int gj = extend_index(j0 - P.LW1 + c, P.n1, rule1);
            double v = (gi < 0 || gj < 0) ? 0.0 : row[gj];             // (zero fill: convolve2d / fftconvolve padding)
            if (cm == 1) { v *= scale; v = v < lim ? lim : v; }
            if (cm == 3) { v = (ne_max - v) * ne_inv; v = v < lim ? lim : v; }                 // transitionModels.py:465-467
            dstrow[c] = v;
        }
    }
    if (cm == 1 || cm == 3) scale = 1.0;
    if (cm == 3) kappa = 1.0;                      // NotEqual is scale-invariant in its input: beta_used = u
    __syncthreads();

    // ---- phase 2: filter along axis 0 (rows), SciPy's symmetric correlate1d order -------------------------------
    const double *hsrc = in_tile + (size_t)P.LW0 * pitch;      // rows [0, th) of the un-filtered tile
    if (lw0 > 0 && !dense) {
        for (int r = y; r < th; r += YN) {
            const double *cen = in_tile + (size_t)(r + P.LW0) * pitch;
            for (int c = P.LW1 - lw1 + x; c < P.LW1 + tw + lw1; c += XW) {
                double acc;
                if (asym0) {                                                           // out[i] = sum_m w[m + lw] in[i + m]
                    acc = 0.0;
                    for (int k = -lw0; k <= lw0; ++k) acc = fma(w0[k + lw0], cen[c + (long long)k * pitch], acc);
                } else {
                    acc = cen[c] * w0[0];
                    for (int k = lw0; k >= 1; --k)
                        acc += (cen[c - (long long)k * pitch] + cen[c + (long long)k * pitch]) * w0[k];
                }
                v_tile[(size_t)r * pitch + c] = acc;
            }
        }
        hsrc = v_tile;
        __syncthreads();
    }

    if (big1) {
        // spline coefficients of the padded row: position q = 0 .. n + 23 of the padded array = grid coordinate q - 12 (12 edge samples on
        // both sides), by SciPy's recursion (spline_prefilter_wave)
        const int N = P.n1 + 24;
        for (int q = threadIdx.x; q < N; q += NTHREADS)
            v_tile[q] = SPLINE_GAIN * in_tile[(size_t)P.LW0 * pitch + P.LW1 + min(max(q - 12, 0), P.n1 - 1)];
        __syncthreads();
        if (threadIdx.x < 64) spline_prefilter_wave(v_tile, N, threadIdx.x, spline_k_compute(N, threadIdx.x));
        __syncthreads();
    }

    // ---- phase 3: filter along axis 1 (cols) + epilogue ---------------------------------------------------------
    double sN = 0.0, sS = 0.0, sC = 0.0, sM0 = 0.0, sM1 = 0.0, sU = 0.0, sMax = 0.0;
    for (int c = x; c < tw; c += XW) {
        const int gj = j0 + c;
        double cA = 0.0, cB = 0.0, g1 = 0.0;
        if (MODE != MODE_FILTER) {
            if (OM == OM_GAUSSIAN) { cA = P.colA[gj]; cB = P.colB[gj]; }
            if (OM == OM_POISSON) cA = P.colA[gj];
            g1 = P.m1[gj];
        }
        for (int r = y; r < th; r += YN) {
            const double *cen = hsrc + (size_t)r * pitch + P.LW1 + c;
            double o;
            if (dense) {                                   // transitionModels.py:889: convolve2d(posterior, kernel, 'same')
                o = 0.0;                                   // (the kernel is point-symmetric: convolution = correlation)
                const int kw = 2 * lw1 + 1;
                for (int a = -lw0; a <= lw0; ++a) {
                    const double *line = cen + (long long)a * pitch;
                    const double *wl = w0 + (a + lw0) * kw + lw1;
                    for (int q = -lw1; q <= lw1; ++q) o = fma(wl[q], line[q], o);
                }
            } else if (big1) {
                const int N = P.n1 + 24;
                const double pp = (double)(j0 + c) - w1[0] + 12.0;           // sampled coordinate in the padded array
                const double fl = floor(pp);
                const int k0 = (int)fmax(fmin(fl, 1.0e9), -1.0e9);
                o = 0.0;
                for (int dk = -1; dk <= 2; ++dk) {
                    const double a = fabs(pp - (fl + (double)dk));
                    const double b3 = a < 1.0 ? 2.0 / 3.0 - a * a + a * a * a * 0.5 : (a < 2.0 ? (2.0 - a) * (2.0 - a) * (2.0 - a) / 6.0 : 0.0);
                    const long long kk = (long long)k0 + dk;
                    o = fma(b3, v_tile[kk < 0 ? 0 : (kk > N - 1 ? N - 1 : (int)kk)], o);
                }
            } else if (asym1) {
                o = 0.0;
                for (int k = -lw1; k <= lw1; ++k) o = fma(w1[k + lw1], cen[k], o);
            } else if (lw1 > 0) {
                o = cen[0] * w1[0];
                for (int k = lw1; k >= 1; --k) o += (cen[-k] + cen[k]) * w1[k];
            } else {
                o = cen[0];
            }
            const int gi = i0 + r;
            const long long cell = (long long)gi * P.n1 + gj;
            if (MODE == MODE_FILTER) {
                P.dst[(long long)b * P.dst_stride + cell] = o;
            } else {
                const double L = likelihood<OM>(P, gi, gj, cA, cB, g1);
                double u = o * scale;                                    // the (normalised) prior of this step
                if (cm == 2) u = u < lim ? lim : u;                      // RegimeSwitch after the stencil
                // mass of the clamped distribution (the reference renormalises by it, transitionModels.py:410)
                sU += (cm == 1 || cm == 3) ? in_tile[(size_t)(r + P.LW0) * pitch + P.LW1 + c] : u;
                if (MODE == MODE_FWD) {
                    const double a = u * L;
                    P.dst[(long long)b * P.dst_stride + cell] = a;
                    sN += a;
                    sMax = fmax(sMax, a);
                    if (MEANS) {
                        if (P.ndim == 2) { sM0 += a * P.m0[gi]; sM1 += a * g1; } else { sM0 += a * g1; }
                    }
                } else {
                    const double beta = u * kappa;
                    double *pp = P.post + (long long)b * P.post_stride + cell;
                    const double p = (*pp) * beta;
                    *pp = p;
                    const double cn = beta * L;
                    P.dst[(long long)b * P.dst_stride + cell] = cn;
                    sMax = fmax(sMax, cn);
                    sN += p;
                    sS += p / L;               // 0/0 -> NaN exactly as numpy does (core.py:463)
                    sC += cn;
                    if (P.ndim == 2) { sM0 += p * P.m0[gi]; sM1 += p * g1; } else { sM0 += p * g1; }
                }
            }
        }
    }
    if (MODE == MODE_FILTER) return;

    double *out = P.psum_out + (long long)b * NRED * P.nblk + blk;
    // all sums of the block with two barriers (red holds 6 * NTHREADS/64 doubles)
    double v[6] = {sN, sS, sC, sM0, sM1, MODE == MODE_BWD ? sU * kappa : sU};
    block_sums<6, NTHREADS / 64>(v, red);
    if (threadIdx.x == 0) {
        out[0] = v[0];
        if (MODE == MODE_BWD) { out[1 * P.nblk] = v[1]; out[2 * P.nblk] = v[2]; }
        if (MODE == MODE_BWD || MEANS) { out[3 * P.nblk] = v[3]; out[4 * P.nblk] = v[4]; }
        if (P.cmode) out[(MODE == MODE_BWD ? 5 : 1) * P.nblk] = v[5];   // clamp bookkeeping: fwd U = sum u, bwd B = sum beta_used
    }
    if (P.cmode) {                                 // NotEqual needs the maximum of the state it inverts
        const double mx = block_max(sMax, red);
        if (threadIdx.x == 0) out[6 * P.nblk] = mx;
    }
}

// out[k] = sum of nblk partials, k over (step, chain, slot); same summation order as sum_partials in the step kernel.
static __global__ __launch_bounds__(NTHREADS) void reduce_partials_kernel(const double *psum, double *out, int nblk, int period) {
    __shared__ double red[NTHREADS / 64 + 1];
    const long long k = blockIdx.x;
    if (period > 0 && k % period == 6) {           // step-kernel partials (period = NRED): slot 6 holds block MAXIMA
        double m = -1.0;
        for (int q = threadIdx.x; q < nblk; q += NTHREADS) m = fmax(m, psum[k * nblk + q]);
        m = block_max(m, red);
        if (threadIdx.x == 0) out[k] = m;
        return;
    }
    const double s = sum_partials(psum + k * nblk, nblk, red);
    if (threadIdx.x == 0) out[k] = s;
}

// Likelihood table of the closed-form models without an in-kernel path: lik[t][cell] = processedPdf(grid, segment_t)
// (observationModels.py:35-56: product over the data dimensions; a dimension whose segment holds a NaN counts as 1).  grid = (blockIdx.x chunks, T).
static __global__ __launch_bounds__(NTHREADS) void lik_table_kernel(int om, double *lik, long long G, int n1, int ndim, const double *m0,
                                                             const double *m1, const double *data, int seg, int d) {
    const long long t = blockIdx.y;
    const double *x = data + t * seg * d;              // (seg, d); a NaN makes its data DIMENSION a factor of 1 (:49-54)
    for (long long c = (long long)blockIdx.x * NTHREADS + threadIdx.x; c < G; c += (long long)gridDim.x * NTHREADS) {
        const double g0 = ndim == 2 ? m0[c / n1] : m1[c];
        const double g1 = ndim == 2 ? m1[c % n1] : 0.0;
        double L = 1.0;
        for (int k = 0; k < d; ++k) {
            const double x0 = x[k], x1 = seg > 1 ? x[d + k] : 0.0;
            if (x0 != x0 || x1 != x1) continue;
            double f;
            if (om == OM_BERNOULLI) {                     // observationModels.py:428-430 (values outside [0, 1] -> 0)
                const double p = (g0 > 1.0 || g0 < 0.0) ? 0.0 : g0;
                f = x0 != 0.0 ? p : 1.0 - p;
            } else if (om == OM_LAPLACE) {                // :635
                f = exp(-fabs(x0 - g0) / g1) / (2.0 * g1);
            } else if (om == OM_WHITE_NOISE) {            // :767
                f = exp(-(x0 * x0) / (2.0 * g0 * g0) - 0.5 * log(2.0 * M_PI * g0 * g0));
            } else if (om == OM_AR1) {                    // :830-831
                const double r = x1 - g0 * x0;
                f = exp(-(r * r) / (2.0 * g1 * g1) - 0.5 * log(2.0 * M_PI * g1 * g1));
            } else {                                      // OM_SCALED_AR1 :893-896
                const double sc = g1 * sqrt(1.0 - g0 * g0);
                const double r = x1 - g0 * x0;
                f = exp(-(r * r) / (2.0 * sc * sc) - 0.5 * log(2.0 * M_PI * sc * sc));
            }
            L *= f;
        }
        lik[t * G + c] = L;
    }
}

// dst[b][c] = src[b * stride + c] * inv[b]   (BLHIP_CARRY: the filtered distribution of the last step, normalised)
static __global__ __launch_bounds__(NTHREADS) void carry_store_kernel(double *dst, const double *src, long long stride, long long G,
                                                               const double *inv) {
    const long long b = blockIdx.y;
    const double s = inv[b];
    for (long long c = (long long)blockIdx.x * NTHREADS + threadIdx.x; c < G; c += (long long)gridDim.x * NTHREADS)
        dst[b * G + c] = src[b * stride + c] * s;
}

// mix[c] = (accumulate ? mix[c] : 0) + sum_b w[b] * states[b][c]   (OnlineStudy: core.py:2196-2212)
static __global__ __launch_bounds__(NTHREADS) void carry_mix_kernel(double *mix, const double *states, long long G, int B, const double *w,
                                                             int accumulate) {
    for (long long c = (long long)blockIdx.x * NTHREADS + threadIdx.x; c < G; c += (long long)gridDim.x * NTHREADS) {
        double s = accumulate ? mix[c] : 0.0;
        for (int b = 0; b < B; ++b) s = fma(w[b], states[(long long)b * G + c], s);
        mix[c] = s;
    }
}

// rows[t][cell] *= inv[t]   (normalisation of stored posteriors, core.py:389 / :441 applied lazily)
static __global__ __launch_bounds__(NTHREADS) void scale_rows_kernel(double *rows, long long G, const double *inv) {
    const long long t = blockIdx.y;
    const double s = inv[t];
    if (s == 1.0) return;                            // (rows the time-resident kernel has normalised already)
    double *row = rows + t * G;
    for (long long c = (long long)blockIdx.x * NTHREADS + threadIdx.x; c < G; c += (long long)gridDim.x * NTHREADS)
        row[c] *= s;
}

// A[t][cell] = A[t][cell] * r + sum_b w[b] * max(post[b][t][cell] * invN[b][t], 1e-300)   (core.py:1362-1366, linear space)
static __global__ __launch_bounds__(NTHREADS) void accumulate_kernel(double *A, const double *post, long long chain_stride,
                                                              int B, long long G, int T, const double *w,
                                                              const double *invN, double r, int first) {
    const long long t = blockIdx.y;
    for (long long c = (long long)blockIdx.x * NTHREADS + threadIdx.x; c < G; c += (long long)gridDim.x * NTHREADS) {
        double acc = first ? 0.0 : A[t * G + c] * r;
        for (int b = 0; b < B; ++b) {
            const double wb = w[b];
            if (wb > 0.0) {
                double p = post[(long long)b * chain_stride + t * G + c] * invN[(long long)b * T + t];
                p = p < 1e-300 ? 1e-300 : p;
                acc += wb * p;
            }
        }
        A[t * G + c] = acc;
    }
}

// The same fold for SMALL grids (1-D studies: a thousand cells, tens of steps, thousands of chains): with a thread per cell the launch is
// a few dozen blocks that each walk all B chains one after the other (the published break-point study: 82 blocks, 0.39 ms per batch of
// 1017 chains = 0.85 TB/s).  Here a block is 64 cells x 4 groups of chains (chains g, g + 4, ...: a wave reads 512 contiguous bytes of
// a chain's row), the groups' sums are added in a fixed order.
static __global__ __launch_bounds__(NTHREADS) void accumulate_small_kernel(double *A, const double *post, long long chain_stride, int B, long long G, int T,
                                                                    const double *w, const double *invN, double r, int first) {
    __shared__ double part[NTHREADS / 64][64];
    const long long t = blockIdx.y;
    const int cl = threadIdx.x & 63, gq = threadIdx.x >> 6;
    const long long c = (long long)blockIdx.x * 64 + cl;
    double acc = 0.0;
    if (c < G) {
        for (int b = gq; b < B; b += NTHREADS / 64) {
            const double wb = w[b];
            if (wb > 0.0) {
                double p = post[(long long)b * chain_stride + t * G + c] * invN[(long long)b * T + t];
                p = p < 1e-300 ? 1e-300 : p;
                acc += wb * p;
            }
        }
    }
    part[gq][cl] = acc;
    __syncthreads();
    if (gq == 0 && c < G) {
        double sum = 0.0;
#pragma unroll
        for (int q = 0; q < NTHREADS / 64; ++q) sum += part[q][cl];
        A[t * G + c] = (first ? 0.0 : A[t * G + c] * r) + sum;
    }
}

// The partial accumulators of a batch whose backward kernel folded the posteriors itself (blhip_chainres.hpp): one per launch slot,
// already weighted and normalised, in the kernel's strip-major layout [t][strip][row][16]:
//     A[t][row][col] = r A + rb sum_slots part[slot][t][col / 16][row][col % 16]          (two cells per lane)
// (n0p: rows per strip of the partials -- the kernel's padded row count when the grid's is not 128 / 256 / 512; pstep: doubles per time
//  step of a partial accumulator on that padded geometry)
static __global__ __launch_bounds__(NTHREADS) void fold_parts_kernel(double *A, const double *part, long long part_stride, int nslots, int n0, int n1,
                                                               int T, double r, double rb, int first, int n0p, long long pstep, int ax) {
    const long long G = (long long)n0 * n1;
    const int t = blockIdx.y;
    if (ax && ax_layout_b(t)) {            // the transposed layout of the both-axes kernels: one cell per lane
        for (int h = 0; h < 2; ++h) {
            const long long c = ((long long)blockIdx.x * NTHREADS + threadIdx.x) * 2 + h;
            if (c >= G) return;
            const int row = (int)(c / n1), col = (int)(c - (long long)row * n1);
            double *ap = A + (long long)t * G + c;
            const double acc = first ? 0.0 : *ap * r;
            const double *src = part + (long long)t * pstep + strip_major_index(row, col, n0p, t, 1);
            double sum = 0.0;
            for (int k = 0; k < nslots; ++k) sum += src[(long long)k * part_stride];
            *ap = fma(rb, sum, acc);
        }
        return;
    }
    if (n1 & 1) {                          // an odd number of columns: pairs of cells would straddle rows -- one cell per lane
        for (int h = 0; h < 2; ++h) {
            const long long c = ((long long)blockIdx.x * NTHREADS + threadIdx.x) * 2 + h;
            if (c >= G) return;
            const int row = (int)(c / n1), col = (int)(c - (long long)row * n1);
            double *ap = A + (long long)t * G + c;
            double acc = first ? 0.0 : *ap * r;
            const double *src = part + (long long)t * pstep + ((long long)(col >> 4) * n0p + row) * 16 + (col & 15);
            double sum = 0.0;
            for (int k = 0; k < nslots; ++k) sum += src[(long long)k * part_stride];
            *ap = fma(rb, sum, acc);
        }
        return;
    }
    const long long c = ((long long)blockIdx.x * NTHREADS + threadIdx.x) * 2;
    if (c >= G) return;
    const int row = (int)(c / n1), col = (int)(c - (long long)row * n1);
    double2 *ap = reinterpret_cast<double2 *>(A + (long long)t * G + c);
    double2 acc = first ? make_double2(0.0, 0.0) : *ap;
    if (!first) { acc.x *= r; acc.y *= r; }
    const double *src = part + (long long)t * pstep + ((long long)(col >> 4) * n0p + row) * 16 + (col & 15);
    double2 sum = make_double2(0.0, 0.0);
    for (int k = 0; k < nslots; ++k) {
        const double2 v = *reinterpret_cast<const double2 *>(src + (long long)k * part_stride);
        sum.x += v.x; sum.y += v.y;
    }
    acc.x = fma(rb, sum.x, acc.x);
    acc.y = fma(rb, sum.y, acc.y);
    *ap = acc;
}

// The separate fold of a batch whose posteriors sit in the chain-resident kernels' strip-major layout on a PADDED geometry
// ([t][column / 16][row of n0p][16], pstep doubles per time step, chain_stride per chain): one cell per lane (any number of columns).
static __global__ __launch_bounds__(NTHREADS) void accumulate_pad_kernel(double *A, const double *post, long long chain_stride, int B, int n0, int n1,
                                                                  int T, const double *w, const double *invN, double r, int first,
                                                                  int n0p, long long pstep, int ax) {
    const long long G = (long long)n0 * n1;
    const long long t = blockIdx.y;
    const long long c = (long long)blockIdx.x * NTHREADS + threadIdx.x;
    if (c >= G) return;
    const int row = (int)(c / n1), col = (int)(c - (long long)row * n1);
    double acc = first ? 0.0 : A[t * G + c] * r;
    const double *pp = post + t * pstep + strip_major_index(row, col, n0p, (int)t, ax);
    for (int b = 0; b < B; ++b) {
        const double wb = w[b];
        if (wb > 0.0) {
            double p0 = __builtin_nontemporal_load(pp + (long long)b * chain_stride) * invN[(long long)b * T + t];
            p0 = p0 < 1e-300 ? 1e-300 : p0;
            acc = fma(wb, p0, acc);
        }
    }
    A[t * G + c] = acc;
}

// Same, two cells per lane (16-B accesses) and four chains in flight per iteration; needs an even number of cells.
// sm_n0 > 0: the sequences are in the chain-resident kernel's strip-major layout [t][column / 16][row][16] (n0 = sm_n0 rows; the
// accumulator keeps the API's [t][row][column]).
static __global__ __launch_bounds__(NTHREADS) void accumulate2_kernel(double *A, const double *post, long long chain_stride,
                                                               int B, long long G, int T, const double *w,
                                                               const double *invN, double r, int first, int sm_n0) {
    const long long t = blockIdx.y;
    const long long c = ((long long)blockIdx.x * NTHREADS + threadIdx.x) * 2;
    if (c >= G) return;
    double2 *ap = reinterpret_cast<double2 *>(A + t * G + c);
    double2 acc = first ? make_double2(0.0, 0.0) : *ap;
    if (!first) { acc.x *= r; acc.y *= r; }
    long long cs = c;
    if (sm_n0 > 0) {
        const int n1 = (int)(G / sm_n0), row = (int)(c / n1), col = (int)(c - (long long)row * n1);
        cs = ((long long)(col >> 4) * sm_n0 + row) * 16 + (col & 15);
    }
    const double *pp = post + t * G + cs;
    int b = 0;
    for (; b + 4 <= B; b += 4) {
        double2 v[4];
        double wb[4], nb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            wb[k] = w[b + k];
            nb[k] = invN[(long long)(b + k) * T + t];
            if (wb[k] > 0.0) {                                   // (read once: non-temporal)
                typedef double dv2 __attribute__((ext_vector_type(2)));
                const dv2 q = __builtin_nontemporal_load(reinterpret_cast<const dv2 *>(pp + (long long)(b + k) * chain_stride));
                v[k] = make_double2(q.x, q.y);
            } else {
                v[k] = make_double2(0.0, 0.0);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (wb[k] > 0.0) {
                double p0 = v[k].x * nb[k], p1 = v[k].y * nb[k];
                p0 = p0 < 1e-300 ? 1e-300 : p0;
                p1 = p1 < 1e-300 ? 1e-300 : p1;
                acc.x = fma(wb[k], p0, acc.x);
                acc.y = fma(wb[k], p1, acc.y);
            }
        }
    }
    for (; b < B; ++b) {
        const double wb = w[b];
        if (wb > 0.0) {
            const double nb = invN[(long long)b * T + t];
            const double2 v = *reinterpret_cast<const double2 *>(pp + (long long)b * chain_stride);
            double p0 = v.x * nb, p1 = v.y * nb;
            p0 = p0 < 1e-300 ? 1e-300 : p0;
            p1 = p1 < 1e-300 ? 1e-300 : p1;
            acc.x = fma(wb, p0, acc.x);
            acc.y = fma(wb, p1, acc.y);
        }
    }
    *ap = acc;
}

static __global__ __launch_bounds__(NTHREADS) void scale_all_kernel(double *A, long long n, double r) {
    for (long long c = (long long)blockIdx.x * NTHREADS + threadIdx.x; c < n; c += (long long)gridDim.x * NTHREADS)
        A[c] *= r;
}

// out[t][i] = sum_j p[t][i][j]   (one block per (row i, t); coalesced along j)
static __global__ __launch_bounds__(NTHREADS) void marginal_rows_kernel(const double *p, double *out, int n0, int n1) {
    __shared__ double red[NTHREADS / 64 + 1];
    const long long t = blockIdx.y, i = blockIdx.x;
    const double *row = p + (t * n0 + i) * n1;
    double s = 0.0;
    for (int j = threadIdx.x; j < n1; j += NTHREADS) s += row[j];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[t * n0 + i] = s;
}

// out[t][j] = sum_i p[t][i][j]   (a thread owns a column; coalesced along j)
static __global__ __launch_bounds__(NTHREADS) void marginal_cols_kernel(const double *p, double *out, int n0, int n1) {
    const long long t = blockIdx.y;
    const int j = blockIdx.x * NTHREADS + threadIdx.x;
    if (j >= n1) return;
    const double *col = p + t * (long long)n0 * n1 + j;
    double s = 0.0;
    for (int i = 0; i < n0; ++i) s += col[(long long)i * n1];
    out[t * n1 + j] = s;
}

// A sequence the chain-resident kernels left in their strip-major layout on a PADDED geometry ([t][column / 16][row of n0p][16], pstep
// doubles per time step) -> the row-major sequence of the grid itself ([t][n0][n1]) that every consumer outside the fit reads.
// blockIdx.y = chain * T + t; lanes run along the destination (coalesced stores, 128-byte runs of the source).
static __global__ __launch_bounds__(NTHREADS) void depad_kernel(double *dst, const double *src, int n0, int n1, int n0p, long long pstep, int T, int ax) {
    const long long G = (long long)n0 * n1;
    const long long c = (long long)blockIdx.x * NTHREADS + threadIdx.x;
    if (c >= G) return;
    const int row = (int)(c / n1), col = (int)(c - (long long)row * n1);
    dst[(long long)blockIdx.y * G + c] = __builtin_nontemporal_load(src + (long long)blockIdx.y * pstep + strip_major_index(row, col, n0p, (int)(blockIdx.y % (unsigned)T), ax));
}

// out[c] = (1/T) sum_t p[t][c]
static __global__ __launch_bounds__(NTHREADS) void time_average_kernel(const double *p, double *out, long long G, int T) {
    for (long long c = (long long)blockIdx.x * NTHREADS + threadIdx.x; c < G; c += (long long)gridDim.x * NTHREADS) {
        double s = 0.0;
        for (int t = 0; t < T; ++t) s += p[(long long)t * G + c];
        out[c] = s / (double)T;
    }
}

// store-only stream, 16 B per lane and access (the other calibration stream: the no-stencil forward chain kernel only writes)
template <bool NT>
__global__ __launch_bounds__(NTHREADS) void fill16_kernel(double2 *__restrict__ dst, long long n2, double v) {
    const long long base = (long long)blockIdx.x * (4 * NTHREADS) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long c = base + k * NTHREADS;
        if (c < n2) {
            if (NT) { __builtin_nontemporal_store(v, &dst[c].x); __builtin_nontemporal_store(v, &dst[c].y); }
            else dst[c] = double2{v, v};
        }
    }
}

// streaming copy, 16 B per lane and access, 4 independent accesses per thread in flight (the calibration of what this part
// reaches on a pure read + write stream); NT: non-temporal loads / stores
template <bool NT>
__global__ __launch_bounds__(NTHREADS) void copy16_kernel(const double2 *__restrict__ src, double2 *__restrict__ dst, long long n2) {
    const long long base = (long long)blockIdx.x * (4 * NTHREADS) + threadIdx.x;
    double2 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long c = base + k * NTHREADS;
        if (c < n2) {
            if (NT) { v[k].x = __builtin_nontemporal_load(&src[c].x); v[k].y = __builtin_nontemporal_load(&src[c].y); }
            else v[k] = src[c];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long c = base + k * NTHREADS;
        if (c < n2) {
            if (NT) { __builtin_nontemporal_store(v[k].x, &dst[c].x); __builtin_nontemporal_store(v[k].y, &dst[c].y); }
            else dst[c] = v[k];
        }
    }
}

static __global__ void fill_kernel(double *p, long long n, double v) {
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (long long)gridDim.x * blockDim.x)
        p[c] = v;
}

}  // namespace blk
