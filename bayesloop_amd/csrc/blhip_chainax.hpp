// Chain-resident pass for batches of chains whose transition filters BOTH parameters: a hyper-study over the widths of two Gaussian random
// walks (HyperStudy.fit, core.py:1349-1366, with CombinedTransitionModel(GRW on parameter 1, GRW on parameter 2), transitionModels.py:
// 632-662 -> :107-111 per axis) -- the shape of the reference's own two-hyper-parameter tests (tests/test_hyperstudy.py:61-103) -- and any
// single fit with such walks wider than the time-resident kernel's 8 grid steps.
//
// Why: rounds 1 - 4 ran these studies with a launch per step and a row PRE-PASS per step (blhip_hwide.hpp + blhip_mfma.hpp): the state
// streams through HBM three times per step, 78 B per cell-step against 24 of the single-axis headline, at 4.4 TB/s.  Here the state
// stays on the chip for the whole pass, as in blc::chain_kernel (blhip_chainres.hpp), whose scheme this kernel extends:
//
//  * layout A = chain_kernel's: block = one strip of 16 grid columns x ALL rows of one chain, state in LDS [row][16], the axis-0 stencil as
//    banded Toeplitz products on v_mfma_f64_4x4x4_4b;
//  * the axis-1 stencil runs along the rows of the grid, i.e. ACROSS the strips.  Instead of exchanging halos (the walks of a
//    hyper-study are as wide as two strips) the blocks of a chain TRANSPOSE the distribution between the two filters: on a square
//    geometry (rows = columns = 16 x strips) block j, which owns columns 16 j .. 16 j + 15 in layout A, owns ROWS 16 j .. 16 j + 15 in
//    layout B, stored [column][16 rows] -- the same LDS shape, so the axis-1 filter is the SAME banded product with the other band;
//  * ONE exchange per step: the two filters commute (separable; to rounding in floating point), so a step applies first the filter of the
//    layout its state is in, transposes, and applies the other filter FUSED WITH THE EPILOGUE (scale, likelihood, sums, store / fold)
//    in the other layout -- where the next step then starts.  The layouts alternate with the time index: everything a step keeps for time
//    t (stored state, posterior, partial accumulator) is in layout B for even t and in layout A for odd t, in both passes, so the
//    backward pass finds the forward pass's states in the layout its own epilogue of that time works in.  (The two-exchange form --
//    axis 0, transpose, axis 1, transpose back, epilogue in layout A -- was built first and measured: the exchanges, not the products,
//    are the step; profiles/r05_notes.md.)  The epilogue of layout A runs the likelihood recurrence along the rows (chain_kernel's); in
//    layout B a lane walks along the SECOND parameter, where the Gaussian has no such recurrence: one exp per cell there (mantissa /
//    exponent form, p / L from the reciprocal mantissa);
//  * the exchange goes through a cache-resident buffer (2 MiB per chain and step parity) and THE DATA IS THE FLAG (blhip_resident.hpp):
//    every element is the 8-byte value with a one-bit tag in its sign bit (everything handed over is >= +0), written as 16-byte pairs
//    (a tile's registers hold rows g, g + 4, ...: one v_permlane16_swap per register pair gives a lane two CONSECUTIVE rows), read by
//    16-byte sc1 loads: the gather is a plain copy of 512 contiguous bytes per wave access into LDS; a consumer whose elements do not
//    carry the step's tag yet asks again (bounded; abort word as in the other resident kernels);
//  * HBM sees what chain_kernel's passes see: 8 B per cell forward (stored state), 16 / 24 B backward (store / fold).
// Sequences are always strip-major on the square geometry (private to the fit, or de-layouted afterwards: blk::depad_kernel).
#pragma once
#include "blhip_chainres.hpp"

namespace blc {

template <int NK, int NTW>
constexpr size_t lds_doubles_ax() { return (size_t)2 * NW * NTW * TM * WCOL + 2 * NK * 16 + 4 * NW * NTW * TM + 2 * NW * 4 * 5 + 2 * NSLOT + 8; }

using blk::ax_layout_b;       // layout of everything kept for time t (and of the epilogue that produces it): B for even t, A for odd t

// 16-byte tagged accesses (two elements each): an 8-byte write-through store costs 2.7 x a 16-byte one per byte on this part
// (MI355X_MICROARCH.md: "scalar sc1 stores are one fabric write each"), and a consumer should keep >= 8 wide loads in flight
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct Tq2 { blr::Tq a, b; };
__device__ __forceinline__ void st_tq2(blr::Rsrc r, unsigned off, double v0, double v1, unsigned bit) {
    const unsigned long long b0 = (unsigned long long)__double_as_longlong(v0), b1 = (unsigned long long)__double_as_longlong(v1);
    const u32x4 q = {(unsigned)b0, (unsigned)(b0 >> 32) | (bit << 31), (unsigned)b1, (unsigned)(b1 >> 32) | (bit << 31)};
    __builtin_amdgcn_raw_buffer_store_b128(q, r, (int)off, 0, 16);
}
__device__ __forceinline__ void st_tq2_plain(blr::Rsrc r, unsigned off, double v0, double v1, unsigned bit) {
    const unsigned long long b0 = (unsigned long long)__double_as_longlong(v0), b1 = (unsigned long long)__double_as_longlong(v1);
    const u32x4 q = {(unsigned)b0, (unsigned)(b0 >> 32) | (bit << 31), (unsigned)b1, (unsigned)(b1 >> 32) | (bit << 31)};
    __builtin_amdgcn_raw_buffer_store_b128(q, r, (int)off, 0, 0);
}
__device__ __forceinline__ Tq2 ld_tq2(blr::Rsrc r, unsigned off) {
    const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 16);
    return Tq2{(unsigned long long)q.x | ((unsigned long long)q.y << 32), (unsigned long long)q.z | ((unsigned long long)q.w << 32)};
}
// v_permlane16_swap on a pair of doubles: rows 1 / 3 of `x` change places with rows 0 / 2 of `y` (rows of 16 lanes; measured:
// tools/ubench/permlane16.hip).  With x = the tile's values of register r, y = those of register r + 1 (lane (g, c): row g + 4 r of
// column c) a lane of an EVEN row g then holds rows g, g + 1 of register r in (x, y), a lane of an ODD row rows g - 1, g of register
// r + 1: two consecutive rows each -- one 16-byte element pair.  Its own inverse.
__device__ __forceinline__ void swap_rows(double &x, double &y) {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    x = __hiloint2double((int)hi[0], (int)lo[0]);
    y = __hiloint2double((int)hi[1], (int)lo[1]);
}

// the tag of the elements step k publishes: a buffer (parity k & 1) is rewritten every second step, zeroed before the launch
__device__ __forceinline__ unsigned ax_tag(int k) { return 1u - (((unsigned)k >> 1) & 1u); }

// The likelihood of the EVEN time steps (the ones whose epilogue works in layout B, where a lane walks along the second parameter and the
// Gaussian has no recurrence: one exp per cell was 16 x 35 fp64 instructions per lane and step -- a quarter of such a step), tabulated once
// per batch in that layout: out[t / 2][strip = row / 16][column][row % 16], zero outside the grid.  Same formula as the kernel's own
// -(sum_d (x_d - mu)^2) cA(column) - n cB(column) (observationModels.py:566-567, product over the data dimensions :49-50).
struct AxLikParams { int n0p, n0t, n1t, T, d, rec_len; const double *m0, *colA, *colB, *rec; double *out; };
static __global__ __launch_bounds__(256) void ax_lik_table_kernel(const AxLikParams P) {
    const int te = blockIdx.y, t = 2 * te;
    const long long G = (long long)P.n0p * P.n0p;
    double *o = P.out + (long long)te * G;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < G; e += (long long)gridDim.x * 256) {
        const int rr = (int)(e & 15), col = (int)((e >> 4) % P.n0p), row = (int)(e / ((long long)P.n0p * 16)) * 16 + rr;
        double v = 0.0;
        if (row < P.n0t && col < P.n1t) {
            const double mu = P.m0[row];
            double s2 = 0.0, dn = 0.0;
            for (int q = 0; q < P.d; ++q) {
                const double x = P.rec[(long long)t * P.rec_len + q];
                if (x == x) { const double dq = x - mu; s2 = fma(dq, dq, s2); dn += 1.0; }
            }
            v = exp(fma(-s2, P.colA[col], -dn * P.colB[col]));
        }
        o[e] = v;
    }
}

// ... and for every other observation model (Laplace, AR1, ScaledAR1, a caller's pdf: the (T, G) table the other kernels read, row-major on the
// grid's true sizes): its even time steps copied into the transposed layout
static __global__ __launch_bounds__(256) void ax_lik_transpose_kernel(const double *lik, double *out, int n0p, int n0t, int n1t) {
    const int te = blockIdx.y, t = 2 * te;
    const long long G = (long long)n0p * n0p;
    const double *src = lik + (long long)t * n0t * n1t;
    double *o = out + (long long)te * G;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < G; e += (long long)gridDim.x * 256) {
        const int rr = (int)(e & 15), col = (int)((e >> 4) % n0p), row = (int)(e / ((long long)n0p * 16)) * 16 + rr;
        o[e] = (row < n0t && col < n1t) ? src[(long long)row * n1t + col] : 0.0;
    }
}

// PAD: the grid is smaller than the square geometry (any n0, n1 <= 512: the geometry is the next of 128 / 256 / 512 that holds both).  As in
// chain_kernel: the stencils reflect at the grid's TRUE last row / column, cells outside the grid are kept at zero and out of every sum,
// read-only inputs are read with bounds.  Lines beyond the grid pick up mirrored values in the first filter (and keep them through the
// second: the filters act along one axis each): the epilogue drops them.
template <int NK, int NTW, bool BWD, bool STORE, bool PAD = false>
__global__ __launch_bounds__(NT, 1) void chainax_kernel(const ChainParams Parg) {
    const ChainParams P = own_chain_params(Parg);      // (every field a scalar of its own: the step loop reloaded whole argument tuples for single fields)
    static_assert(NTW == 1 || NTW == 2 || NTW == 4, "square geometries of 128 / 256 / 512 rows and columns");
    constexpr int R0 = (4 * NK - TM) / 2;
    static_assert(NK >= 6 && R0 % 4 == 0, "band = 16 + 2 R0 columns, R0 a multiple of 4");
    constexpr int N0 = NW * NTW * TM;             // rows = columns of the geometry
    constexpr int XSZ = N0 * WCOL;                // doubles of a strip
    constexpr int AST = 16;                       // compact band tables (band_products)
    constexpr bool FOLD = BWD && !STORE;
    const int n0t = PAD ? P.n0t : N0, n1t = PAD ? P.n1t : N0;          // the grid's true sizes
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *const X0 = lds;                       // [N0][16]  the state at the beginning of a step (layout A or B)
    double *const X1 = lds + XSZ;                 // [N0][16]  after the first filter, transposed into the other layout
    double *const As0 = X1 + XSZ;                 // [NK][16]  band of axis 0
    double *const As1 = As0 + NK * AST;           // [NK][16]  band of axis 1
    double *const m0s = As1 + NK * AST;           // [N0]      coordinates of the first parameter (rows)
    double *const m1s = m0s + N0;                 // [N0]      ... of the second (columns)
    double *const cAs = m1s + N0;                 // [N0]      the columns' likelihood constants (layout B: a lane walks along the columns)
    double *const cBs = cAs + N0;                 // [N0]
    double *const red = cBs + N0;                 // [2][NW * 4][5]
    double *const scal = red + 2 * NW * 4 * 5;    // [NSLOT]
    double *const iscal = scal + NSLOT;           // [NSLOT]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // xch_mode bit 0: the blocks of a chain share an XCD (block b runs on XCD b % 8 -- observed dispatch order, used for speed only):
    // the launch has 8 x strips x ceil(chains / 8) blocks, XCD x = b % 8 works on chains x, x + 8, ...; blocks without a chain leave
    const unsigned xl = blockIdx.x >> 3;
    const int cs = (P.xch_mode & 1) ? (int)((blockIdx.x & 7u) + 8u * (xl / (unsigned)P.strips)) : (int)(blockIdx.x / (unsigned)P.strips);
    const int tj = (P.xch_mode & 1) ? (int)(xl % (unsigned)P.strips) : (int)(blockIdx.x - cs * P.strips);
    if (cs >= P.nslots) return;
    const unsigned bid = (unsigned)(cs * P.strips + tj);
    const int b = sldi(P.chain_ids, cs);
    const int tap0 = sldi(P.tap_id, b), tap1 = sldi(P.tap_id1, b);
    const int lw0 = tap0 >= 0 ? sldi(P.tap_lw, tap0) : 0, lw1 = tap1 >= 0 ? sldi(P.tap_lw, tap1) : 0;
    const long long o0 = tap0 >= 0 ? sldi(P.tap_off, tap0) : 0, o1 = tap1 >= 0 ? sldi(P.tap_off, tap1) : 0;
    const int gj = tj * WCOL + (lane & 15);       // layout A: the lane's grid column; layout B: its grid ROW
    const long long G = (long long)N0 * N0;

    // a distribution (prior, uniform, reset: row-major on the grid's true sizes) -> X0 in layout A / B
    auto load_dist = [&](const double *src, bool lay_b) {
        for (int e = tid; e < XSZ; e += NT) {
            const int line = e >> 4, own = tj * WCOL + (e & 15);
            const int row = lay_b ? own : line, col = lay_b ? line : own;
            X0[e] = (!PAD || (row < n0t && col < n1t)) ? src[(long long)row * n1t + col] : 0.0;
        }
    };
    // ---- prologue (the first step consumes its source unfiltered: identity bands, replaced after step 0) -------------------------------
    const int t_first = BWD ? P.T - 1 : 0;
    for (int e = tid; e < 2 * NK * AST; e += NT) As0[e] = band_distance16(e % (NK * AST), R0) == 0 ? 1.0 : 0.0;
    for (int e = tid; e < N0; e += NT) {
        m0s[e] = P.m0[PAD ? min(e, n0t - 1) : e];
        m1s[e] = P.m1[PAD ? min(e, n1t - 1) : e];
        cAs[e] = P.colA[PAD ? min(e, n1t - 1) : e];
        cBs[e] = P.colB[PAD ? min(e, n1t - 1) : e];
    }
    if (tid < 2 * NSLOT) scal[tid] = 1.0;
    load_dist(P.src0, !ax_layout_b(t_first));     // (a step STARTS in the layout its epilogue does not work in)
    // layout A: the lane's column constants; layout B: its row's coordinate
    const bool okA = !PAD || gj < n1t, okB = !PAD || gj < n0t;
    const double g1 = P.m1[PAD ? min(gj, n1t - 1) : gj];
    const double cA = P.colA[PAD ? min(gj, n1t - 1) : gj], cB = P.colB[PAD ? min(gj, n1t - 1) : gj];
    const double mub = P.m0[PAD ? min(gj, n0t - 1) : gj];
    double *const pchain = P.post + (long long)b * P.post_stride;
    const int row0 = wv * (NTW * TM);
    auto fresh_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
    // the lane's cells in its strip of a stored sequence / partial accumulator (strip-major, either layout: [strip][line][16])
    auto cell_off = [&](int l, int it, int r) { return (unsigned)tj * (unsigned)(XSZ * 8) + (unsigned)(row0 + it * TM + (l >> 4) + 4 * r) * 128u + (unsigned)(l & 15) * 8u; };
    // the exchange buffers of this chain slot: [parity][strip][N0][16] tagged elements, one descriptor
    const blr::Rsrc xr = blr::strip_rsrc(P.xch + (long long)cs * P.xch_chain, (unsigned)(2u * (unsigned)(G * 8)));
    auto xbuf = [&](int k) { return (unsigned)(k & 1) * (unsigned)(G * 8); };
    // where a product tile's elements go: the consumer strip is the tile's 16 lines, inside it [this strip's own index][line in 16].  After
    // swap_rows a lane holds two consecutive lines (g & ~1, + 1) of register 2 rp + (g & 1), rp = 0 / 1: one 16-byte store each
    auto pub_off = [&](int l, int it, int rp) {
        const int g = l >> 4;
        return (unsigned)(wv * NTW + it) * (unsigned)(XSZ * 8) + (unsigned)tj * 2048u + (unsigned)(l & 15) * 128u +
               (unsigned)((g & 2) + 4 * (2 * rp + (g & 1))) * 8u;
    };

    double xd[DMAX], xn[DMAX];
#pragma unroll
    for (int q = 0; q < DMAX; ++q) xd[q] = q < P.d ? P.rec[(long long)t_first * P.rec_len + q] : __builtin_nan("");
    // backward: the stored alpha / fold: the accumulator cells of the lane's cells at this step -- requested behind the gather (whose tagged
    // loads must not queue behind HBM loads: loads return in order), a whole filter phase ahead of the epilogue that consumes them
    double al[BWD ? NTW : 1][4];
    double pa[FOLD ? NTW : 1][4];
    double *const pslot = FOLD ? P.part + (long long)cs * P.part_stride : nullptr;
    const double wch = FOLD ? P.wchain[b] : 0.0;
    double inpred = FOLD ? P.infirst[b] : 0.0;
    double sfn = 1.0;
    if (wv == SCALE_WAVE || wv == 5 || wv == 0 || wv == NW - 1) __builtin_amdgcn_s_setprio(2);
    bool dead = false;
    // change points (transitionModels.py:300-312): a step may consume the reset distribution instead of the previous state -- through
    // both bands (the change point stands in front of the walks in the combined model's list) or unfiltered (behind them: bit 7)
    int kind_n = blk::SRC_PREV;
    typedef const double __attribute__((address_space(3))) *lds_cp;
    unsigned long long gq0 = 0ull, gq1 = 0ull;
    double Sprev = 1.0;
    double mq = 1.0, iq = 1.0, dn_prev = -1.0;
    int nq = 0;
    __syncthreads();

    // a bounded wait: `again` re-requests what is missing
    // (the wave waits as one -- blr::wave_all: scalar branches; lane by lane the loop is a stack of lane masks in scalar registers)
    auto wait_for = [&](auto &&all_there, auto &&again) {
        if (dead || blr::wave_all(all_there())) return;
        const unsigned long long t0 = blr::now_ticks();
        for (unsigned spins = 1;; ++spins) {
            blr::nap();
            again();
            if (blr::wave_all(all_there())) return;
            if ((spins & 255u) == 0u) {
                if (blr::uni((int)blr::ld_flag(P.abort_word)) != 0) { dead = true; return; }
                if (blr::now_ticks() - t0 > P.timeout_ticks) { blr::st_flag(P.abort_word, 1u); dead = true; return; }
            }
        }
    };
    // the wave's product ring over the strip in `S`: fill / advance by one tile (reflection at the true last line `nlim`)
    // (exact geometries whose waves own >= R0 lines: only the first / last wave reflect, at compile-time entries -- ring_fill_*_wave, blhip_chainres.hpp)
    constexpr bool FASTEDGE = BLC_FASTEDGE && !PAD && NTW * TM >= R0;
    auto ring_fill = [&](const double *S, double (&Bv)[NK], bool edge, int nlim) {
        const int l = fresh_lane(), g = l >> 4, c = l & 15;
        if (FASTEDGE && wv == 0) ring_fill_first_wave(Bv, S, g, c, 0, NK, -R0);
        else if (FASTEDGE && wv == NW - 1 && R0 + 12 >= NTW * TM) ring_fill_last_wave<N0, NTW * TM>(Bv, S, g, c, 0, NK, -R0);
        else if (!FASTEDGE && edge) {
#pragma unroll
            for (int kb = 0; kb < NK; ++kb) Bv[kb] = S[reflect1(row0 - R0 + 4 * kb + g, nlim) * WCOL + c];
        } else {
            const double *s0 = S + (row0 - R0 + g) * WCOL + c;
#pragma unroll
            for (int kb = 0; kb < NK; ++kb) Bv[kb] = s0[kb * 4 * WCOL];
        }
    };
    auto ring_next = [&](const double *S, double (&Bv)[NK], bool edge, int nlim, int i) {
        const int l = fresh_lane(), g = l >> 4, c = l & 15;
#pragma unroll
        for (int kb = 0; kb < NK - 4; ++kb) Bv[kb] = Bv[kb + 4];
        if (FASTEDGE && wv == NW - 1 && (i - row0) + TM + R0 + 12 >= NTW * TM) ring_fill_last_wave<N0, NTW * TM>(Bv, S, g, c, NK - 4, 4, (i - row0) + TM + R0);
        else if (!FASTEDGE && edge) {
#pragma unroll
            for (int q = 0; q < 4; ++q) Bv[NK - 4 + q] = S[reflect1(i + TM + R0 + 4 * q + g, nlim) * WCOL + c];
        } else {
            const double *s1 = S + (i + TM + R0 + g) * WCOL + c;
#pragma unroll
            for (int q = 0; q < 4; ++q) Bv[NK - 4 + q] = s1[q * 4 * WCOL];
        }
    };
    auto band_ptr = [&](const double *Ab, int l) {
        const unsigned aoff = (unsigned)(((l >> 4) << 2) | (l & 3)) * 8u;
        return (lds_cp)((const char __attribute__((address_space(3))) *)(lds_cp)Ab + aoff);
    };

    // (option chain_prof: shader-clock stamps of block 0, waves 0 and 2, steps 8 .. 23 -- where a step spends its time)
    const bool prof_me = P.prof != nullptr && bid == 0 && lane == 0 && (wv == 0 || wv == 2);
#define BLX_STAMP(i) do { if (prof_me && k >= 8 && k < 24) P.prof[(wv ? 256 : 0) + (k - 8) * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
    for (int k = 0; k < P.T; ++k) {
        const int t = BWD ? P.T - 1 - k : k;
        const int tn = (k + 1 < P.T) ? (BWD ? t - 1 : t + 1) : t;
        const unsigned bit = ax_tag(k);
        const bool lay_b = ax_layout_b(t);                 // the layout of this step's epilogue; the step starts in the other one
        BLX_STAMP(0);
        // ---- the scale wave's requests (chain_kernel: the sums the scale of step k + 1 is made of were requested a step ago) ------------
        const bool scale_wave = wv == SCALE_WAVE;
        const int jn = k + 1;
        const bool need = scale_wave && jn >= P.lag && jn < P.T;
        const unsigned long long *gp = P.gran + ((((long long)((jn - P.lag) & (NSLOT - 1)) * P.nslots + cs) * P.strips + lane) << 1);
        const bool mine = need && lane < P.strips;
        const unsigned long long hq0 = gq0, hq1 = gq1;
        if (scale_wave && jn + 1 >= P.lag && jn + 1 < P.T && lane < P.strips) {
            const unsigned long long *gn = P.gran + ((((long long)((jn + 1 - P.lag) & (NSLOT - 1)) * P.nslots + cs) * P.strips + lane) << 1);
            gq0 = blr::ld_u64(gn); gq1 = blr::ld_u64(gn + 1);
        }
        const int kind = kind_n & 0x7f;
        const bool nofilter = (kind_n & 0x80) != 0;
        if (P.kinds) kind_n = P.kinds[(long long)tn * P.B + b];
        if (kind != blk::SRC_PREV && k > 0) {          // (rare: once per chain and change point)
            load_dist(P.reset, !lay_b);
            if (nofilter) for (int e = tid; e < 2 * NK * AST; e += NT) As0[e] = band_distance16(e % (NK * AST), R0) == 0 ? 1.0 : 0.0;
            __syncthreads();
        }
        const double sf_now = sfn;
        if (FOLD) sfn = P.sfwd[(long long)b * P.T + min(tn + 1, P.T - 1)];
#pragma unroll
        for (int q = 0; q < DMAX; ++q) xn[q] = q < P.d ? P.rec[(long long)tn * P.rec_len + q] : __builtin_nan("");

        // ---- the first filter, along the lines of the layout the state is in -> the strips of the other layout ----------------------------
        // (state in layout A: axis 0 along the rows; in layout B: axis 1 along the columns)
        {
            const double *Ab = lay_b ? As0 : As1;
            const int nlim = lay_b ? n0t : n1t;
            const bool edge = row0 < R0 || row0 + NTW * TM + R0 > nlim;
            double Bv[NK];
            ring_fill(X0, Bv, edge, nlim);
            // (the band's entries once per filter for all the wave's tiles: band_products_w, blhip_chainres.hpp)
            constexpr bool HOIST1 = BLC_HOIST_A && NTW >= 2;
            double Aw[HOIST1 ? NK - 3 : 1];
            if constexpr (HOIST1) {
                const lds_cp Al = band_ptr(Ab, fresh_lane());
#pragma unroll
                for (int q = 0; q < NK - 3; ++q) Aw[q] = Al[q * AST];
            }
#pragma unroll
            for (int it = 0; it < NTW; ++it) {
                const int l = fresh_lane();
                d4 acc;
                if constexpr (HOIST1) acc = band_products_w<NK, 0, NK>(Aw, Bv); else acc = band_products<NK, 0, NK, AST>(band_ptr(Ab, l), Bv);
                double a0 = acc[0], a1 = acc[1], a2 = acc[2], a3 = acc[3];
                swap_rows(a0, a1);
                swap_rows(a2, a3);
                if (P.xch_mode & 2) {
                    st_tq2_plain(xr, xbuf(k) + pub_off(l, it, 0), a0, a1, bit);
                    st_tq2_plain(xr, xbuf(k) + pub_off(l, it, 1), a2, a3, bit);
                } else {
                    st_tq2(xr, xbuf(k) + pub_off(l, it, 0), a0, a1, bit);
                    st_tq2(xr, xbuf(k) + pub_off(l, it, 1), a2, a3, bit);
                }
                if (it + 1 < NTW) ring_next(X0, Bv, edge, nlim, row0 + it * TM);
            }
        }
        BLX_STAMP(1);
        // ---- this block's strip of the other layout -> X1 (a plain copy: XSZ elements, 512 contiguous bytes per wave access) ---------------
        {
            constexpr int NG = XSZ / NT / 2;      // 16-byte pairs per thread: 2 (128 rows) .. 8 (512 rows), all in flight at once
            Tq2 fq[NG];
            const unsigned base = xbuf(k) + (unsigned)tj * (unsigned)(XSZ * 8) + (unsigned)tid * 16u;
            auto issue = [&]() {
#pragma unroll
                for (int j = 0; j < NG; ++j) fq[j] = ld_tq2(xr, base + (unsigned)j * (unsigned)(NT * 16));
            };
            // (all tag bits at once first -- AND / OR of the high words; element by element, branch-free, only when that fails: a NaN
            //  counts as arrived whatever its sign, blr::tq_ok)
            auto there = [&]() {
                unsigned hand = 0xffffffffu, hor = 0u;
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    const unsigned ha = (unsigned)(fq[j].a >> 32), hb = (unsigned)(fq[j].b >> 32);
                    hand &= ha & hb; hor |= ha | hb;
                }
                if (blr::wave_all(bit ? (hand >> 31) != 0u : (hor >> 31) == 0u)) return true;
                unsigned ok = 1u;
#pragma unroll
                for (int j = 0; j < NG; ++j) ok &= blr::tq_ok_bits(fq[j].a, bit) & blr::tq_ok_bits(fq[j].b, bit);
                return ok != 0u;
            };
            issue();
            wait_for(there, issue);
            BLX_STAMP(2);
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                X1[2 * (tid + j * NT)] = blr::tq_value(fq[j].a);
                X1[2 * (tid + j * NT) + 1] = blr::tq_value(fq[j].b);
            }
        }
        double *const pstep = pchain + (long long)t * G;
        double *const pslot_t = FOLD ? pslot + (long long)t * G : nullptr;
        if (BWD) {
            const int l = fresh_lane();
#pragma unroll
            for (int it = 0; it < NTW; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) al[BWD ? it : 0][r] = ldnt(pstep, cell_off(l, it, r));
        }
        if (FOLD) {
            const int l = fresh_lane();
            const double *abase = P.part_fresh ? P.zeros : pslot_t;
#pragma unroll
            for (int it = 0; it < NTW; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned aoffs = cell_off(l, it, r);
                    pa[FOLD ? it : 0][r] = ldnt(abase, P.part_fresh ? (aoffs & 4088u) : aoffs);
                }
        }
        // (the likelihood of the lane's cells out of a table: layout-B steps where there is one -- the Gaussian's exponentials otherwise --,
        //  every step of the other observation models, whose layout-A steps read the row-major table the other kernels read)
        double lk[NTW][4];
        const bool tabled = lay_b ? P.lik != nullptr : P.lik_nat != nullptr;
        if (tabled && lay_b) {
            const int l = fresh_lane();
            const double *lrow = P.lik + (long long)(t >> 1) * G;
#pragma unroll
            for (int it = 0; it < NTW; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) lk[it][r] = blm::ld32(lrow, cell_off(l, it, r));
        } else if (tabled) {
            const int l = fresh_lane();
            const double *lrow = P.lik_nat + (long long)t * n0t * n1t;
#pragma unroll
            for (int it = 0; it < NTW; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = row0 + it * TM + (l >> 4) + 4 * r;
                    const unsigned off = __umul24(PAD ? min(row, n0t - 1) : row, (unsigned)n1t * 8u) + (unsigned)(PAD ? min(gj, n1t - 1) : gj) * 8u;
                    lk[it][r] = blm::ld32(lrow, off);
                }
        }
        __syncthreads();
        BLX_STAMP(3);

        // ---- the scale of this step; the scale wave prepares the next one (see chain_kernel) ------------------------------------------------
        const double scale = scal[k & (NSLOT - 1)];
        if (FOLD && k > 0) inpred *= sf_now * iscal[k & (NSLOT - 1)];
        const double wq = FOLD ? wch * inpred : 0.0, wfloor = FOLD ? wch * 1e-300 : 0.0;
        if (scale_wave) {
            double sj = 1.0;
            if (need) {
                const unsigned long long want = (unsigned long long)(unsigned)(jn - P.lag + 1);
                unsigned long long q0 = hq0, q1 = hq1;
                bool ok = !mine || ((q0 >> 32) == want && (q1 >> 32) == want);
                if (!dead && !__all(ok)) {
                    const unsigned long long t0 = blr::now_ticks();
                    for (unsigned spins = 1; !__all(ok); ++spins) {
                        if (!ok) { q0 = blr::ld_u64(gp); q1 = blr::ld_u64(gp + 1); ok = (q0 >> 32) == want && (q1 >> 32) == want; }
                        blr::nap();
                        if ((spins & 255u) == 0u) {
                            if (blr::ld_flag(P.abort_word) != 0u) { dead = true; break; }
                            if (blr::now_ticks() - t0 > P.timeout_ticks) { blr::st_flag(P.abort_word, 1u); dead = true; break; }
                        }
                    }
                }
                const double v = mine ? __longlong_as_double((long long)((q0 & 0xffffffffull) | (q1 << 32))) : 0.0;
                const double Sg = blk::wave_sum(v);
                sj = dead ? 1.0 : Sprev * scal[(jn - P.lag) & (NSLOT - 1)] / Sg;
                Sprev = Sg;
            }
            if (lane == 0) { scal[jn & (NSLOT - 1)] = sj; if (FOLD) iscal[jn & (NSLOT - 1)] = 1.0 / sj; }
        }

        // ---- the second filter, fused with the epilogue in this step's layout -----------------------------------------------------------------
        double sN = 0.0, sS = 0.0, sC = 0.0, sM0 = 0.0, sM1 = 0.0;
        {
            const double *Ab = lay_b ? As1 : As0;
            const int nlim = lay_b ? n1t : n0t;
            const bool edge = row0 < R0 || row0 + NTW * TM + R0 > nlim;
            double Bv[NK];
            ring_fill(X1, Bv, edge, nlim);
            // layout A: anchors of the stride-4 likelihood recurrence along the lane's rows; layout B: the row's squared distances
            double mE = 1.0, mR = 1.0, iE = 1.0, iR = 1.0;
            int nE = 0, nR = 0;
            double s2 = 0.0, dnb = 0.0;
            if (lay_b) {
#pragma unroll
                for (int q = 0; q < DMAX; ++q) {
                    const double x = xd[q];
                    if (x == x) { const double dq = x - mub; s2 = fma(dq, dq, s2); dnb += 1.0; }
                }
            }
            // (second filter: where the epilogue leaves 2 (NK - 3) registers -- the forward kernels)
            constexpr bool HOIST2 = BLC_HOIST_A && NTW >= 2 && !BWD;
            double Aw2[HOIST2 ? NK - 3 : 1];
            if constexpr (HOIST2) {
                const lds_cp Al = band_ptr(Ab, fresh_lane());
#pragma unroll
                for (int q = 0; q < NK - 3; ++q) Aw2[q] = Al[q * AST];
            }
#pragma unroll
            for (int it = 0; it < NTW; ++it) {
                const int i = row0 + it * TM;
                const int l = fresh_lane(), g = l >> 4, c = l & 15;
                d4 acc;
                if constexpr (HOIST2) acc = band_products_w<NK, 0, NK>(Aw2, Bv); else acc = band_products<NK, 0, NK, AST>(band_ptr(Ab, l), Bv);
                if (it == 0 && !lay_b && !tabled) {
                    // arg(r) = sum_k [-(x_k - mu_r)^2 cA - cB]  (observationModels.py:566-567; product over dimensions :49-50; blhip_mfma.hpp)
                    const double mu0 = m0s[i + g], mu4 = m0s[i + g + 4];
                    double a0 = 0.0, s1 = 0.0, dn = 0.0;
#pragma unroll
                    for (int q = 0; q < DMAX; ++q) {
                        const double x = xd[q];
                        if (x == x) {
                            const double dq = x - mu0;
                            a0 = fma(-(dq * dq), cA, a0) - cB;
                            s1 += (x - mu0) + (x - mu4);
                            dn += 1.0;
                        }
                    }
                    const double d1 = cA * (mu4 - mu0) * s1;
                    const double d2 = -32.0 * cA * dn * P.step0 * P.step0;
                    exp_mn(a0, mE, nE);
                    exp_mn(d1, mR, nR);
                    if (dn != dn_prev) {
                        int tmp;
                        exp_mn(d2, mq, nq);
                        if (BWD) exp_mn(-d2, iq, tmp);
                        dn_prev = dn;
                    }
                    if (BWD) { iE = blmath::inv_m(mE); iR = blmath::inv_m(mR); }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int li = i + g + 4 * r;                       // layout A: the cell's row; layout B: its column
                    if (lay_b && !tabled) {                              // one exponential per cell: -(sum (x - mu)^2) cA(column) - n cB(column)
                        exp_mn(fma(-s2, cAs[li], -dnb * cBs[li]), mE, nE);
                        if (BWD) iE = blmath::inv_m(mE);
                    }
                    const bool in = !PAD || (lay_b ? (okB && li < n1t) : (okA && li < n0t));          // (cells outside the grid stay zero)
                    const double Lv = tabled ? lk[it][r] : ldexp(mE, nE);
                    const unsigned off = cell_off(l, it, r);
                    if (!BWD) {
                        const double a = in ? acc[r] * scale * Lv : 0.0;
                        X0[li * WCOL + c] = a;
                        if (STORE) stnt(pstep, off, a);
                        sN += a;
                        acc[r] = a;
                    } else {
                        const double beta = in ? acc[r] * scale : 0.0;
                        const double p = al[BWD ? it : 0][r] * beta;
                        const double cn = beta * Lv;
                        // p / L; 0 / 0 -> NaN (core.py:463).  (The recurrence's quotient is computed for every cell and SELECTED: written as one
                        //  conditional expression with the table's division in it, each of the lane's 16 cells got a branch of its own -- 16 lane
                        //  masks per chain-step in the epilogue)
                        double pl;
                        if (tabled) pl = in ? p / Lv : 0.0;
                        else {
                            const double quot = ldexp(p * iE, -nE);
                            pl = nan_if(Lv == 0.0, quot);
                            if (PAD) pl = in ? pl : 0.0;
                        }
                        X0[li * WCOL + c] = cn;
                        if (!FOLD) stnt(pstep, off, p);
                        else stnt(pslot_t, off, pa[FOLD ? it : 0][r] + fmax(p * wq, wfloor));
                        sN += p;
                        sS += pl;
                        sC += cn;
                        acc[r] = p;
                    }
                    if (!lay_b && !tabled) {
                        mE *= mR; nE += nR;
                        mR *= mq; nR += nq;
                        if (BWD) { iE *= iR; iR *= iq; }
                    }
                    if ((BWD || STORE) && P.means) {
                        sM0 = fma(acc[r], lay_b ? mub : m0s[li], sM0);
                        sM1 = fma(acc[r], lay_b ? m1s[li] : g1, sM1);
                    }
                }
                if (it + 1 < NTW) ring_next(X1, Bv, edge, nlim, i);
            }
        }
        BLX_STAMP(4);
        // ---- sums: waves -> LDS; after the barrier wave 5 adds them up, writes the strip's partial sums, publishes the granule ------------
        double v[5] = {sN, sS, BWD ? sC : sM0, BWD ? sM0 : sM1, sM1};
        constexpr int NV = BWD ? 5 : 4;
        const int nv = BWD ? (P.means ? 5 : 3) : (P.means ? 4 : 1);
        double *rk = red + (k & 1) * (NW * 4 * 5);
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            if (q < nv) {
                double x = v[q];
                x = blk::dpp_add<0x111, 0xf>(x);
                x = blk::dpp_add<0x112, 0xf>(x);
                x = blk::dpp_add<0x114, 0xf>(x);
                x = blk::dpp_add<0x118, 0xf>(x);
                if ((lane & 15) == 15) rk[(wv * 4 + (lane >> 4)) * 5 + q] = x;
            }
        }
        __syncthreads();
        BLX_STAMP(5);
        if (k == 0 || (nofilter && kind != blk::SRC_PREV)) {            // the chain's bands replace the identities of the first step / of an unfiltered restart
            for (int e = tid; e < NK * AST; e += NT) {
                const int a = band_distance16(e, R0);
                As0[e] = a == 0 ? (lw0 > 0 ? P.taps[o0] : 1.0) : (a <= lw0 ? P.taps[o0 + a] : 0.0);
                As1[e] = a == 0 ? (lw1 > 0 ? P.taps[o1] : 1.0) : (a <= lw1 ? P.taps[o1 + a] : 0.0);
            }
            __syncthreads();
        }
        if (wv == 5 && lane < nv) {
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < NW * 4; ++w) tot += rk[w * 5 + lane];
            const int slot = BWD ? lane : (lane < 2 ? lane : lane + 1);
            if (lane == 0 || (BWD ? (P.means || lane < 3) : (lane != 1 && P.means != 0)))
                P.psum[(((long long)t * P.B + b) * NRED + slot) * P.nblk + tj] = tot;
            if (lane == (BWD ? 2 : 0)) {
                const unsigned long long bits = (unsigned long long)__double_as_longlong(tot);
                const unsigned long long tag = (unsigned long long)(unsigned)(k + 1) << 32;
                unsigned long long *gw = P.gran + ((((long long)(k & (NSLOT - 1)) * P.nslots + cs) * P.strips + tj) << 1);
                blr::st_u64(gw, tag | (bits & 0xffffffffull));
                blr::st_u64(gw + 1, tag | (bits >> 32));
            }
        }
#pragma unroll
        for (int q = 0; q < DMAX; ++q) xd[q] = xn[q];
    }
#undef BLX_STAMP
}

}  // namespace blc
