// Time-resident step kernel for BATCHES of chains whose transition filters axis 0 only: the hyper-study over the width of one
// Gaussian random walk (HyperStudy.fit, core.py:1349-1366, with transitionModels.py:107-111 on the first parameter) -- BASELINE C4.
//
// Why: with a launch per step (blhip_mfma.hpp) every chain streams its state through HBM twice per step: 16 B per cell forward, 32 B
// backward, and the backward launches already run at 0.97 of what a streaming copy reaches on the part.  A chain's state is 2 MiB
// (512 x 512); the LDS of the chip holds 40 MiB.  So a ROUND of chains stays resident:
//
//  * a chain is cut into column strips of 16 grid columns x ALL n0 rows (the stencil runs along the rows, so strips never exchange
//    halos); block = strip = 8 waves; n1 / 16 strips per chain, floor(CUs / strips) chains per launch (C4: 32 strips, 8 chains);
//    the launch runs all T steps of its chains; the state lives in LDS (two buffers of n0 x 16 doubles, ping-pong: one barrier per
//    step); the launches of a pass follow each other on one stream, chains sorted by stencil radius;
//  * per step a wave owns n0 / 8 rows: banded Toeplitz products on the fp64 matrix pipe exactly as in blm::mfma_step_kernel (B
//    operand = a register ring over the state, here filled from LDS: 512 contiguous bytes per read, conflict-free; A operand = the
//    weight band, in LDS), the same fused epilogue (lazy normaliser, Gaussian likelihood recurrence with stride 4, sums);
//  * HBM sees only what the fit has to keep: forward 8 B per cell (the stored filtered distribution; nothing at all for
//    evidence-only fits), backward 16 B (stored alpha in, posterior out).  The stored alpha of the NEXT step is requested one whole
//    step ahead, so the loads have a full step (~5 us) to arrive;
//  * the lazy normaliser is a per-chain sum over its strips.  A step is linear in its input, so its scale may be ANY positive
//    number the host can reconstruct; it only has to keep the state in range.  Every strip publishes its partial sum of step k as
//    a data-tagged granule pair (no flag, no fence); step k divides by the NORMALISER of step k - lag, n_j = S_j / (S_(j-1) s_j)
//    (S = sums over the strips, s = the scales used): s_k = S_(k-lag-1) s_(k-lag) / S_(k-lag).  The state then carries the product
//    of the last `lag` normalisers -- bounded, whatever the lag.  (Dividing by the lagged SUM, as the single-chain kernel does with
//    lag 2, is a feedback loop x_k = x_(k-1) - x_(k-lag) + nu in the log domain: marginally stable for lag 2, exponentially unstable
//    for lag >= 3 -- measured: the C4 study overflowed before step 256.)  One wave per strip turns the sums into scales (a wave
//    reduction + a division per step; done by every wave it was ~8 % of the vector instructions): during step k it prepares the
//    scale of step k + 1 from granules it requested when step k - 1 began; with lag = 4 they were published a whole step before
//    that, so nobody waits for a sum in the steady state.  The strips sum in one fixed order: bit-identical scales across the
//    strips of a chain.  The host undoes the scales (chain_unlag).
//
// Bounded spins + abort word as in blhip_resident.hpp: if the blocks of a launch are not co-resident the fit falls back to the
// launch-per-step kernels.
#pragma once
#include "blhip_mfma.hpp"
#include "blhip_resident.hpp"

namespace blc {

using blf::exp_mn;
using blf::reflect1;
using blf::sldi;
using blf::DMAX;
using blk::NRED;
using blm::d4;

constexpr int NT = 512, NW = 8;   // threads / waves per block
constexpr int TM = 16;            // rows per product tile (MFMA M)
constexpr int WCOL = 16;          // columns per strip (MFMA N)
constexpr int NSLOT = 8;          // ring of granule slots (>= 2 * max lag)
constexpr int MAXLAG = 4;
constexpr int MAX_STRIPS = 64;    // strips per chain: one granule per lane
constexpr int SCALE_WAVE = 6;     // the wave that turns the strips' sums into the next step's scale (shares its SIMD with no edge wave)

struct ChainParams {
    int n0, n1, strips;          // n0 == NW * NTW * TM, strips = n1 / 16: the geometry the kernels work on
    int n0t, n1t;                // the grid's TRUE sizes (PAD kernels: n0t <= n0, n1t <= n1; everything the fit itself owns -- states,
                                 // stored sequences, partial accumulators -- is laid out on the padded geometry, padded cells hold zeros)
    int T, d, rec_len, lag;
    int means;                   // also sum the stored distribution times the grid values (forward-only fits; backward: per-chain means wanted)
    int strip_major;             // layout of post / part: 0 = [t][row][column] (the API's), 1 = [t][strip][row][16] (private to a fit whose
                                 // backward pass folds: every wave access is 512 contiguous bytes, a strip's step 64 KB)
    int B;                       // chains of the batch (partial-sum layout)
    int nslots;                  // chains of this launch
    int nblk;                    // partial-sum slots per (step, chain, sum)
    const int *chain_ids;        // [nslots] -> chain of the batch
    const int *tap_id;           // [B] the chain's axis-0 kernel in the tap table, -1 = none
    const double *taps; const int *tap_off; const int *tap_lw;
    const double *src0;          // what the first step consumes instead of a transition: prior (forward) / uniform (backward), (G)
    const unsigned char *kinds;  // NK = 4 (no stencil at all: change-point studies): [T][B] source kind of every step -- blk::SRC_PREV or
    const double *reset;         //   blk::SRC_RESET = the step consumes `reset` (G) instead of the previous state (transitionModels.py:300-312)
    double *post; long long post_stride;       // [chain][T][G]: stored states (forward out, backward in) -> posteriors (backward out)
    const double *m0, *m1, *colA, *colB, *rec;
    double step0;
    double *psum;                // [T][B][NRED][nblk]
    unsigned long long *gran;    // [NSLOT][nslots][strips][2] {tag << 32 | half of a double}
    // backward pass of a hyper-study, fused fold (template STORE = false): instead of storing the posteriors, every chain adds
    // w_chain * max(posterior / its sum, 1e-300) (core.py:1362-1366) to the partial accumulator of its launch slot
    const double *sfwd;          // [B][T] the forward pass's scales: the sums of the posteriors follow from scalars (see below)
    const double *wchain;        // [B] weights (0: the chain does not contribute)
    const double *infirst;       // [B] 1 / sum of the posterior of the last time step
    double *part; long long part_stride;       // [slot][T][G]
    const double *zeros;         // first launch of a batch (part_fresh): the slots hold nothing yet -- their cells are read from this 4 KB of
    int part_fresh;              //   zeros instead (same instruction count: no memset of the slots, no HBM read of them in this launch)
    // Change-point batches whose backward pass folds (the stored states are private and never overwritten): every chain repeats the
    // steps before its FIRST restart bit for bit.  tshare[b] > 0: chain b does not store its states of steps t < tshare[b]; the backward
    // pass reads them from chain `bprov` of the batch (the chain with the latest first restart, which stores everything).
    const int *tshare;           // [B] or nullptr
    int bprov;
    // ... and need not be COMPUTED twice either: a restart consumes the reset distribution, not the chain's past.  skip_prefix != 0: the
    // forward pass of chain b starts at step tshare[b] (its first restart; lagged scales as for a chain that begins there); the host
    // copies the sums of the earlier steps from chain bprov, which also sums its state times the reset distribution at EVERY step (the
    // restart sum of whichever chain restarts next).  Forward passes of no-stencil batches only.
    int skip_prefix;
    unsigned *abort_word;
    unsigned long long timeout_ticks;
    unsigned long long *prof;    // development builds (-DBLC_PROF): [2 waves][16 steps][16 stamps] shader-clock stamps of block 0
    // (last: the fields above keep their offsets -- the two-chain kernel's register allocation is sensitive to how the argument block loads)
    const double *lik;           // TAB kernels: the likelihood of every step, [T][n0 * n1] row-major (table models: built on the device or by the caller)
    // blc::chainax_kernel (blhip_chainax.hpp: walks on BOTH parameters, the distribution is transposed between the two filters)
    const int *tap_id1;          // [B] the chain's axis-1 kernel in the tap table, -1 = none
    double *xch;                 // [nslots][2 step parities][n0 * n1] tagged elements: the exchange buffers (zeroed before every launch)
    long long xch_chain;         // doubles between the buffers of two chain slots (2 n0 n1)
    const double *lik_nat;       // blc::chainax_kernel: tabulated likelihood (every model but the Gaussian), [T][n0t * n1t] row-major, for the steps whose
                                 //   epilogue works in layout A; `lik` then holds the even steps in the transposed layout
    int xch_mode;                // experiments (option chain_ax1_mode): bit 0 = the blocks of a chain share an XCD (block b runs on XCD b % 8), bit 1 = plain publishing stores
    // (round 6) the anchors of the likelihood recurrence, tabulated once per fit (anchor_table_kernel below): [T][NW waves][strips][64 lanes] x
    // {mantissa E, mantissa R, exponents (nE, nR) as two ints, pad} = 32 bytes per lane and step.  The Gaussian kernels of this file read
    // them a step ahead instead of evaluating two exponentials per lane and step -- every chain of a hyper-study sees the SAME likelihood, so
    // the 512 chains of BASELINE C4 repeated each anchor 512 times (5.6 % of its forward pass, 2.8 % of the folding backward pass)
    const double *anch;
};

// ---- the anchors of the stride-4 likelihood recurrence (blhip_mfma.hpp) of a lane's rows at one time step --------------------------------------
// arg(r) = sum_k [-(x_k - mu_r)^2 cA - cB]  (observationModels.py:566-567; product over dimensions :49-50): a0 = arg at the lane's first row,
// d1 = arg(+4 rows) - arg, dn = valid data dimensions (the second difference is -32 cA dn step0^2, constant in time while dn is)
// (the data record as four scalars: an array handed over by reference was kept in scratch memory by the no-stencil kernels -- 72 bytes per
//  lane, C5's forward pass 46.6 -> 61.2 ms)
static_assert(DMAX == 4, "anchor_terms takes the record's four slots as scalars");
__device__ __forceinline__ void anchor_terms(double x0, double x1, double x2, double x3, double mu0, double mu4, double cA, double cB, double &a0, double &d1,
                                             double &dn) {
    double s1 = 0.0, a = 0.0, n = 0.0;
    auto slot = [&](double x) {
        if (x == x) {
            const double dq = x - mu0;
            a = fma(-(dq * dq), cA, a) - cB;
            s1 += (x - mu0) + (x - mu4);
            n += 1.0;
        }
    };
    slot(x0); slot(x1); slot(x2); slot(x3);
    a0 = a; dn = n;
    d1 = cA * (mu4 - mu0) * s1;
}

struct AnchorParams {
    int T, d, rec_len, strips, rows_per_wave, n0t, n1t;       // rows_per_wave = NTW * 16; n0t, n1t: the grid's true sizes (padded geometries clamp)
    const double *m0, *colA, *colB, *rec;
    double *out;                                              // [T][NW][strips][64][4]
};
__device__ __forceinline__ long long anchor_index(int t, int wv, int strips, int tj, int lane) { return ((((long long)t * NW + wv) * strips + tj) * 64 + lane) * 4; }

// one block per (strip, time step): thread (wave wv, lane) computes what lane `lane` of wave `wv` of that strip's blocks needs at that step --
// with the very operations the kernels used to run, so the tabulated anchors are bit for bit the in-kernel ones
static __global__ __launch_bounds__(NT) void anchor_table_kernel(const AnchorParams A) {
    const int tj = blockIdx.x, t = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int g = lane >> 4, gj = tj * WCOL + (lane & 15), gjc = min(gj, A.n1t - 1);
    const int i = wv * A.rows_per_wave;
    const double mu0 = A.m0[min(i + g, A.n0t - 1)], mu4 = A.m0[min(i + g + 4, A.n0t - 1)];
    const double cA = A.colA[gjc], cB = A.colB[gjc];
    double xd[DMAX];
#pragma unroll
    for (int q = 0; q < DMAX; ++q) xd[q] = q < A.d ? A.rec[(long long)t * A.rec_len + q] : __builtin_nan("");
    double a0, d1, dn, mE, mR;
    int nE, nR;
    anchor_terms(xd[0], xd[1], xd[2], xd[3], mu0, mu4, cA, cB, a0, d1, dn);
    exp_mn(a0, mE, nE);
    exp_mn(d1, mR, nR);
    double *o = A.out + anchor_index(t, wv, A.strips, tj, lane);
    o[0] = mE; o[1] = mR;
    o[2] = __longlong_as_double((long long)(((unsigned long long)(unsigned)nR << 32) | (unsigned long long)(unsigned)nE));
    o[3] = dn;
}
// a lane's entry, requested a step ahead: two 16-byte loads
struct AnchorEntry { double mE, mR, n, dn; };
__device__ __forceinline__ AnchorEntry anchor_load(const double *anch, int t, int wv, int strips, int tj, int lane) {
    const double2 *p = reinterpret_cast<const double2 *>(anch + anchor_index(t, wv, strips, tj, lane));
    const double2 a = p[0], b = p[1];
    return AnchorEntry{a.x, a.y, b.x, b.y};
}
__device__ __forceinline__ void anchor_unpack(const AnchorEntry &e, double &mE, int &nE, double &mR, int &nR) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(e.n);
    mE = e.mE; mR = e.mR; nE = (int)(unsigned)(u & 0xffffffffull); nR = (int)(unsigned)(u >> 32);
}

// The kernel arguments arrive as 16-register tuples (s_load_dwordx16) and the register allocator spills and restores a tuple as ONE
// unit: with ~100 live scalars in the step loop of the two-chain fold kernel a single field of ChainParams cost a 16-lane v_readlane
// burst per use (ISA, NK = 14: 304 v_readlane of 1473 vector instructions per chain-step, 16-lane bursts repeated 7 times).  Every
// field that loop uses is therefore copied into a scalar of its own first -- the empty asm makes the copy a separate value the
// allocator can place, spill or keep on its own (137 v_readlane, 1403 vector instructions).  (The single-chain kernels have few
// spills -- 35 v_readlane per step -- and do not gain: measured on the ISA, not applied there.)
// (round 6: with an EMPTY asm the copy was coalesced back into the tuple's sub-register wherever the allocator liked -- the fold kernel's
//  step loop reloaded a whole 16-register tuple six times for `gran`, `psum` and `step0`; a real s_mov defines a register of its own)
#ifndef BLC_OWN_MOV
#define BLC_OWN_MOV 1
#endif
#if BLC_OWN_MOV
__device__ __forceinline__ int own_sgpr(int v) { int r; asm volatile("s_mov_b32 %0, %1" : "=s"(r) : "s"(v)); return r; }
__device__ __forceinline__ unsigned long long own_sgpr(unsigned long long v) { unsigned long long r; asm volatile("s_mov_b64 %0, %1" : "=s"(r) : "s"(v)); return r; }
__device__ __forceinline__ long long own_sgpr(long long v) { return (long long)own_sgpr((unsigned long long)v); }
__device__ __forceinline__ double own_sgpr(double v) { return __longlong_as_double((long long)own_sgpr((unsigned long long)__double_as_longlong(v))); }
template <class T>
__device__ __forceinline__ T *own_sgpr(T *p) { return (T *)(T __attribute__((address_space(1))) *)own_sgpr((unsigned long long)p); }
#else
using blr::own_sgpr;
#endif
struct LoopParams {
    int T, d, rec_len, lag, B, nslots, strips, nblk, part_fresh, bprov;
    const double *rec, *sfwd, *zeros, *reset, *anch;
    const unsigned char *kinds;
    double *post, *psum;
    long long post_stride;
    unsigned long long *gran;
    unsigned *abort_word;
    unsigned long long timeout_ticks;
    double step0;
};
__device__ __forceinline__ LoopParams loop_params(const ChainParams &P) {
    LoopParams Q;
    Q.T = own_sgpr(P.T); Q.d = own_sgpr(P.d); Q.rec_len = own_sgpr(P.rec_len); Q.lag = own_sgpr(P.lag); Q.B = own_sgpr(P.B);
    Q.nslots = own_sgpr(P.nslots); Q.strips = own_sgpr(P.strips); Q.nblk = own_sgpr(P.nblk); Q.part_fresh = own_sgpr(P.part_fresh);
    Q.bprov = own_sgpr(P.bprov);
    Q.rec = own_sgpr(P.rec); Q.sfwd = own_sgpr(P.sfwd); Q.zeros = own_sgpr(P.zeros); Q.reset = own_sgpr(P.reset); Q.anch = own_sgpr(P.anch);
    Q.kinds = own_sgpr(P.kinds); Q.post = own_sgpr(P.post); Q.psum = own_sgpr(P.psum); Q.post_stride = own_sgpr(P.post_stride);
    Q.gran = own_sgpr(P.gran); Q.abort_word = own_sgpr(P.abort_word); Q.timeout_ticks = own_sgpr(P.timeout_ticks);
    Q.step0 = own_sgpr(P.step0);
    return Q;
}

// Every field of the argument block as a scalar value of its own (see own_sgpr): the both-axes kernels' step loops reloaded one 16-register
// tuple of the block 15 - 17 times per step (272 of 508 v_readlane in chainax_kernel<12, 4, backward, fold>) for single fields of it.
static_assert(sizeof(ChainParams) == 352, "own_chain_params copies ChainParams field by field: a new field needs its line there (and this size)");
__device__ __forceinline__ ChainParams own_chain_params(const ChainParams &P) {
    ChainParams Q;
    Q.n0 = own_sgpr(P.n0);
    Q.n1 = own_sgpr(P.n1);
    Q.strips = own_sgpr(P.strips);
    Q.n0t = own_sgpr(P.n0t);
    Q.n1t = own_sgpr(P.n1t);
    Q.T = own_sgpr(P.T);
    Q.d = own_sgpr(P.d);
    Q.rec_len = own_sgpr(P.rec_len);
    Q.lag = own_sgpr(P.lag);
    Q.means = own_sgpr(P.means);
    Q.strip_major = own_sgpr(P.strip_major);
    Q.B = own_sgpr(P.B);
    Q.nslots = own_sgpr(P.nslots);
    Q.nblk = own_sgpr(P.nblk);
    Q.chain_ids = own_sgpr(P.chain_ids);
    Q.tap_id = own_sgpr(P.tap_id);
    Q.taps = own_sgpr(P.taps);
    Q.tap_off = own_sgpr(P.tap_off);
    Q.tap_lw = own_sgpr(P.tap_lw);
    Q.src0 = own_sgpr(P.src0);
    Q.kinds = own_sgpr(P.kinds);
    Q.reset = own_sgpr(P.reset);
    Q.post = own_sgpr(P.post);
    Q.post_stride = own_sgpr(P.post_stride);
    Q.m0 = own_sgpr(P.m0);
    Q.m1 = own_sgpr(P.m1);
    Q.colA = own_sgpr(P.colA);
    Q.colB = own_sgpr(P.colB);
    Q.rec = own_sgpr(P.rec);
    Q.step0 = own_sgpr(P.step0);
    Q.psum = own_sgpr(P.psum);
    Q.gran = own_sgpr(P.gran);
    Q.sfwd = own_sgpr(P.sfwd);
    Q.wchain = own_sgpr(P.wchain);
    Q.infirst = own_sgpr(P.infirst);
    Q.part = own_sgpr(P.part);
    Q.part_stride = own_sgpr(P.part_stride);
    Q.zeros = own_sgpr(P.zeros);
    Q.part_fresh = own_sgpr(P.part_fresh);
    Q.tshare = own_sgpr(P.tshare);
    Q.bprov = own_sgpr(P.bprov);
    Q.skip_prefix = own_sgpr(P.skip_prefix);
    Q.abort_word = own_sgpr(P.abort_word);
    Q.timeout_ticks = own_sgpr(P.timeout_ticks);
    Q.prof = own_sgpr(P.prof);
    Q.lik = own_sgpr(P.lik);
    Q.tap_id1 = own_sgpr(P.tap_id1);
    Q.xch = own_sgpr(P.xch);
    Q.xch_chain = own_sgpr(P.xch_chain);
    Q.lik_nat = own_sgpr(P.lik_nat);
    Q.xch_mode = own_sgpr(P.xch_mode);
    Q.anch = own_sgpr(P.anch);
    return Q;
}

// streaming accesses (every byte of the sequence / the partial accumulators is touched once per launch): non-temporal hints --
// measured: C4 backward + fold 306 -> 293 us, C5 backward 63.9 -> 59.9 us per logical step
__device__ __forceinline__ double ldnt(const double *base, unsigned byteoff) { return __builtin_nontemporal_load((const double *)((const char *)base + byteoff)); }
__device__ __forceinline__ void stnt(double *base, unsigned byteoff, double v) { __builtin_nontemporal_store(v, (double *)((char *)base + byteoff)); }

// x, or NaN where `nan` holds: ONE select on the high word (a NaN is a NaN whatever its low word holds; the 64-bit select is two v_cndmask)
__device__ __forceinline__ double nan_if(bool nan, double x) {
    return __hiloint2double(nan ? 0x7ff80000 : __double2hiint(x), __double2loint(x));
}

// The banded product of a 16-row tile.  As ONE v_mfma_f64_16x16x4 chain the band is 16 + 2 R0 columns wide: every output row
// multiplies 16 structural zeros (NK products of 64 cycles).  v_mfma_f64_4x4x4_4b computes four INDEPENDENT 4 x 4 x 4 products per
// instruction (16 cycles, measured: tools/ubench/mfma_f64_4x4.hip -- 70 TFLOP/s with 8 chains in flight, 60 with 4): the blocks are
// the strip's four column groups, the four accumulators of a tile its four row groups, and row group rg takes its inputs from ring
// entries rg + s, s = 0 .. R0 / 2 -- a band of 4 + 2 R0 columns: (NK - 3) x 4 instructions of 16 cycles instead of NK of 64, i.e. 192
// cycles less per tile whatever the radius.  Same operand layouts: B = the ring (lane (g, c) = X[row + g][c]), D lane (g, c) = rows
// g + 4 rg of column c; A for shift s: lane (k = l >> 4, i = l & 3) = w(|4 s - R0 + k - i|).
#ifndef BLC_BAND4
#define BLC_BAND4 1
#endif
#ifndef BLC_FASTEDGE
#define BLC_FASTEDGE 1
#endif
#ifndef BLC_HOIST_A
#define BLC_HOIST_A 1
#endif
#ifndef BLC_OWNREG
#define BLC_OWNREG 1
#endif
#ifndef BLC_CIRC_RING
#define BLC_CIRC_RING 1
#endif
#ifndef BLC_HOIST_LOOP
#define BLC_HOIST_LOOP 1
#endif
#ifndef BLC_FOLD_HOIST_MAX
#define BLC_FOLD_HOIST_MAX 20
#endif
typedef const double __attribute__((address_space(3))) *band_cp;
// (Bv: a ring of NR >= OFF + NK entries, the tile's window begins at entry OFF -- compile-time, so the entries stay registers)
// (AST: doubles between the tables of two shifts.  64 = one entry per lane; 16 = one entry per (k, i) pair, read by the four lanes
//  that share it -- the four column groups multiply the same band: a quarter of the LDS, same-address reads are broadcasts)
template <int NK, int OFF = 0, int NR = NK, int AST = 64>
__device__ __forceinline__ d4 band_products(band_cp Al, const double (&Bv)[NR]) {
    static_assert(OFF + NK <= NR, "the tile's window lies inside the ring");
#if BLC_BAND4
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
    for (int s = 0; s < NK - 3; ++s) {
        const double A = Al[s * AST];
        a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(A, Bv[OFF + s], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(A, Bv[OFF + s + 1], a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_4x4x4f64(A, Bv[OFF + s + 2], a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_4x4x4f64(A, Bv[OFF + s + 3], a3, 0, 0, 0);
    }
    return d4{a0, a1, a2, a3};
#else
    static_assert(AST == 64, "the 16x16x4 form: one entry per lane");
    d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kb = 0; kb < NK; ++kb) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Al[kb * 64], Bv[OFF + kb], acc, 0, 0, 0);
    return acc;
#endif
}
// ... with the band's entries in registers (read once per step for all the wave's tiles: the A operand was 180 of the 350 KB a forward
// step of a 512-row strip moved through the LDS -- 44 reads of 512 bytes per wave, the same 11 values for each of its four tiles)
template <int NK, int OFF = 0, int NR = NK>
__device__ __forceinline__ d4 band_products_w(const double (&Aw)[NK > 3 ? NK - 3 : 1], const double (&Bv)[NR]) {
    static_assert(OFF + NK <= NR, "the tile's window lies inside the ring");
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
    for (int s = 0; s < NK - 3; ++s) {
        a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(Aw[s], Bv[OFF + s], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(Aw[s], Bv[OFF + s + 1], a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_4x4x4f64(Aw[s], Bv[OFF + s + 2], a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_4x4x4f64(Aw[s], Bv[OFF + s + 3], a3, 0, 0, 0);
    }
    return d4{a0, a1, a2, a3};
}
// ... over a CIRCULAR ring: entry e of the strip's ring lives in register Bv[e % NR].  A tile's window begins 4 entries behind the previous
// one's, so the four entries a tile no longer needs are exactly the slots of the four new ones: no ring shift (the sliding ring copied
// NK - 4 entries = 2 (NK - 4) v_mov per tile, 60 of the two-chain fold kernel's ~580 vector instructions per chain-step at NK 14).  NR = NK
// where 4 divides NK, NK + 2 otherwise (the slots of a window's last entries must not be those of the next window's first ones while
// the window is in use).  `first` = the window's first entry; a constant once the tile loop is unrolled.
template <int NK> constexpr int ring_mod() { return NK % 4 == 0 ? NK : NK + 2; }
template <int NK, int NR>
__device__ __forceinline__ d4 band_products_wc(const double (&Aw)[NK > 3 ? NK - 3 : 1], const double (&Bv)[NR], int first) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
    for (int s = 0; s < NK - 3; ++s) {
        a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(Aw[s], Bv[(first + s) % NR], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(Aw[s], Bv[(first + s + 1) % NR], a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_4x4x4f64(Aw[s], Bv[(first + s + 2) % NR], a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_4x4x4f64(Aw[s], Bv[(first + s + 3) % NR], a3, 0, 0, 0);
    }
    return d4{a0, a1, a2, a3};
}
template <int NK, int NR, int AST>
__device__ __forceinline__ d4 band_products_c(band_cp Al, const double (&Bv)[NR], int first) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
    for (int s = 0; s < NK - 3; ++s) {
        const double A = Al[s * AST];
        a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(A, Bv[(first + s) % NR], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(A, Bv[(first + s + 1) % NR], a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_4x4x4f64(A, Bv[(first + s + 2) % NR], a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_4x4x4f64(A, Bv[(first + s + 3) % NR], a3, 0, 0, 0);
    }
    return d4{a0, a1, a2, a3};
}
// entry e of a band table [NK][64]: the distance |input row - output row| its lane multiplies
__device__ __forceinline__ int band_distance(int e, int R0) {
#if BLC_BAND4
    return abs(4 * (e >> 6) + ((e & 63) >> 4) - R0 - (e & 3));
#else
    return abs(4 * (e >> 6) + ((e & 63) >> 4) - R0 - (e & 15));
#endif
}
// ... of a compact table [NK][16] (entry 4 k + i of a shift)
__device__ __forceinline__ int band_distance16(int e, int R0) { return abs(4 * (e >> 4) + ((e & 15) >> 2) - R0 - (e & 3)); }


// ---- the product ring of the waves at the grid's edges (exact geometries) ---------------------------------------------------------------------
// SciPy's 'reflect' extension (row -1 - q = row q, row n0 + q = row n0 - 1 - q) costs ~8 integer instructions per ring entry (reflect1), 26
// entries per step in the forward kernels of a 512-row strip: the two edge waves of a block ran a third more vector instructions than the
// six interior ones, and the step's barrier waits for them.  On an exact geometry (no PAD) whose waves own at least R0 rows only the first
// and the last wave reach beyond the grid, and WHICH of their entries do is known at compile time (entries begin at multiples of 4 rows, R0
// is one): the mirrored entries are read at immediate offsets from ONE mirrored base address, the others from the usual one -- an edge
// wave's ring then costs what an interior wave's does.
// Entries [k0, k0 + cnt) of the ring Bv; entry k0 + q begins rel0 + 4 q rows from the wave's first row (lane (g, c) holds row + g, column c).
// (k0, cnt, rel0 are constants at every call site once the tile loop is unrolled: the selections below fold away)
template <int NR>
__device__ __forceinline__ void ring_fill_first_wave(double (&Bv)[NR], const double *S, int g, int c, int k0, int cnt, int rel0) {      // (row0 = 0)
    const double *sm = S + (3 - g) * WCOL + c;          // mirrored at the first row: row -1 - (rel + g); rel = -4 -> row 3 - g
    const double *s0 = S + g * WCOL + c;
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const int rel = rel0 + 4 * q;
        if (q < cnt) Bv[k0 + q] = rel < 0 ? sm[(-rel - 4) * WCOL] : s0[rel * WCOL];
    }
}
// (kmod: slot of entry k0 + q = (k0 + q) % kmod -- the circular rings of the backward kernels; 0: the slot is the entry)
template <int N0, int ROWS_W, int NR>
__device__ __forceinline__ void ring_fill_last_wave(double (&Bv)[NR], const double *S, int g, int c, int k0, int cnt, int rel0, int kmod = 0) {       // (row0 = N0 - ROWS_W)
    constexpr int row0 = N0 - ROWS_W;
    const int relmax = rel0 + 4 * (cnt - 1);
    const double *sm = S + (2 * N0 - 1 - row0 - relmax - g) * WCOL + c;      // mirrored at the last row: row 2 N0 - 1 - (row0 + rel + g)
    const double *s0 = S + (row0 + rel0 + g) * WCOL + c;                      // (row0 + rel0 >= 0: the last wave's ring begins inside the grid)
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const int rel = rel0 + 4 * q;
        if (q < cnt) Bv[kmod ? (k0 + q) % kmod : k0 + q] = rel >= ROWS_W ? sm[(relmax - rel) * WCOL] : s0[(rel - rel0) * WCOL];
    }
}

template <int NK, int NTW>
constexpr size_t lds_doubles() { return (size_t)(NTW > 4 ? 1 : 2) * NW * NTW * TM * WCOL + NK * (NTW > 4 ? 16 : 64) + NW * NTW * TM + 2 * NW * 4 * 5 + 2 * NSLOT + 8; }

#ifdef BLC_PROF
#define BLC_STAMP(i) do { if (prof_me && k >= 8 && k < 24) P.prof[(wv ? 256 : 0) + (k - 8) * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define BLC_STAMP(i) do { } while (0)
#endif

// STORE: the forward pass keeps every step's state (full / forward-only fits); backward: true = the posteriors are stored (in place of
// the forward states), false = they are folded into the average posterior right away.  The fold needs every posterior NORMALISED, and
// the sum of a step's posterior is complete only after all strips have finished the step -- but it follows from scalars: the
// stencil is self-adjoint, so  N_t = sum alpha_t beta_t = s'_t N_(t+1) / s_(t+1)  (s' = this pass's scales, s = the forward pass's,
// N_(T-1) = sum alpha_(T-1) / G).  The host checks the prediction against the reduced sums afterwards (1e-9) and repeats the batch
// with the launch-per-step kernels if it ever differs (the partial accumulators of a batch are folded only after that check).
// PAD: the grid is smaller than the geometry (rows not 128 / 256 / 512, columns not a multiple of 16).  The kernel works on the padded
// geometry; what differs: the stencil reflects at the grid's TRUE last row, cells outside the grid are kept at zero (and out of the
// sums), the read-only inputs (source distribution, coordinates, column constants) are read with bounds.  Only for sequences private
// to the fit (strip-major layout on the padded geometry): forward passes and storing backward passes here, folding backward passes in
// chain_fold2_kernel (<= 512 rows) / here (1024 rows).
template <int NK, int NTW, bool BWD, bool STORE, bool PAD = false, bool TAB = false>
__global__ __launch_bounds__(NT, 1) void chain_kernel(const ChainParams P) {
    // TAB: the likelihood comes out of a table (every observation model but the Gaussian on a 2-D grid: Laplace, AR1, a caller's own pdf)
    // instead of the recurrence -- one more 8-byte read per cell and step, shared by the chains of a launch (they read the same rows
    // at about the same time); geometries of <= 512 rows (PAD: the table lives on the grid's true sizes, cells outside read as zero)
    static_assert(!TAB || NTW <= 4, "tabulated likelihood: geometries of <= 512 rows");
    constexpr int R0 = (4 * NK - TM) / 2;
    constexpr int N0 = NW * NTW * TM;
    constexpr int XSZ = N0 * WCOL;
    static_assert(!(PAD && BWD && !STORE) || NTW > 4, "padded grids of <= 512 rows: the folding backward pass is chain_fold2_kernel's");
    static_assert(!(PAD && BWD && STORE && NTW > 4), "padded grids of 1024 rows: posteriors are folded, never stored");
    const int n0t = PAD ? P.n0t : N0, n1t = PAD ? P.n1t : P.n1;      // the grid's true sizes
    static_assert(NK == 4 || (NK >= 6 && R0 % 4 == 0), "band = 16 + 2 R0 columns, R0 a multiple of 4; NK = 4: no stencil");
    constexpr bool FILTER = NK > 4;
    // TALL (NTW = 8: 1024 rows): LDS holds ONE copy of the strip (128 KB), not two.  A step's new state waits in registers until every
    // wave has read its rings (the step's barrier), is written then, and a second barrier releases the next step -- the scheme of
    // chain_fold2_kernel.  (The folding backward pass is the single-chain one -- also on padded grids, 513 .. 1023 rows: the partial
    // accumulators live on the padded geometry and fold_parts_kernel reads the grid's cells out of them.)
    constexpr bool TALL = NTW > 4;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *const X = lds;                     // [2][N0][16]   (TALL: [1][N0][16])
    constexpr int AST = (TALL && BLC_BAND4) ? 16 : 64;     // (TALL: the compact band table -- the strip leaves 32 KB, the widest band has 44 shifts)
    double *const As = X + (TALL ? 1 : 2) * XSZ;   // [NK][AST]  A operand: W[m][k] = w(|k - R0 - m|)
    double *const m0s = As + NK * AST;         // [N0]       row coordinates
    double *const red = m0s + N0;              // [2][NW * 4][5] sums of the waves' rows of 16 lanes, double-buffered by step parity
    double *const scal = red + 2 * NW * 4 * 5;     // [NSLOT] the scales s_j of the steps around the current one (written by the scale wave)
    double *const iscal = scal + NSLOT;            // [NSLOT] 1 / s_j

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cs = blockIdx.x / P.strips, tj = blockIdx.x - cs * P.strips;
    const int b = sldi(P.chain_ids, cs);
    const int tap = sldi(P.tap_id, b);
    const int lw0 = tap >= 0 ? sldi(P.tap_lw, tap) : 0;
    const long long o0 = tap >= 0 ? sldi(P.tap_off, tap) : 0;
    const int gj = tj * WCOL + (lane & 15);
    const long long G = (long long)P.n0 * P.n1;
    // steps t < tsh: shared with chain P.bprov (see ChainParams).  Only batches WITHOUT a stencil share prefixes (blhip_fit_paths.hpp): in the
    // filtering kernels the test is compile-time (it was a scalar branch around every store of the epilogue)
    const int tsh = (!FILTER && P.tshare) ? sldi(P.tshare, b) : 0;

    // ---- prologue: band, row coordinates, first source -> LDS ---------------------------------------------------------------------
    // the first step consumes its source unfiltered: it runs with the identity band (exact), the chain's band replaces it afterwards
    // (one code path for every step: no per-step branches around the ring and the products)
    if (FILTER) for (int e = tid; e < NK * AST; e += NT) As[e] = (AST == 16 ? band_distance16(e, R0) : band_distance(e, R0)) == 0 ? 1.0 : 0.0;
    for (int e = tid; e < N0; e += NT) m0s[e] = P.m0[PAD ? min(e, n0t - 1) : e];
    if (tid < 2 * NSLOT) scal[tid] = 1.0;
    if (FILTER) for (int e = tid; e < XSZ; e += NT) {
        const int row = e >> 4, col = tj * WCOL + (e & 15);
        X[e] = (!PAD || (row < n0t && col < n1t)) ? P.src0[(long long)row * n1t + col] : 0.0;
    }
    const bool colok = !PAD || gj < n1t;
    const int gjc = PAD ? min(gj, n1t - 1) : gj;
    const double g1 = P.m1[gjc];
    const double cA = TAB ? 0.0 : P.colA[gjc], cB = TAB ? 0.0 : P.colB[gjc];
    double *const pchain = P.post + (long long)b * P.post_stride;
    const unsigned rowx8 = P.strip_major ? (unsigned)WCOL * 8u : (unsigned)P.n1 * 8u;                        // bytes between rows
    const unsigned strip0 = P.strip_major ? (unsigned)tj * (unsigned)(P.n0 * WCOL * 8) : (unsigned)tj * (unsigned)(WCOL * 8);   // the strip's first byte
    const int row0 = wv * (NTW * TM);
    // The lane's coordinates are re-derived from a laundered lane id wherever they are used: carried through the time loop, every
    // index expression of the step (ring rows, LDS offsets, byte offsets of the 4 NTW cells) is loop-invariant and the optimiser
    // parks it in a VGPR -- more than a hundred of them (the time-resident single-chain kernel spilled for the same reason).
    auto fresh_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
    auto cell_off = [&](int l, int it, int r) { return __umul24(row0 + it * TM + (l >> 4) + 4 * r, rowx8) + strip0 + (unsigned)(l & 15) * 8u; };

    const int kb = (!BWD && !FILTER && P.skip_prefix) ? tsh : 0;      // the first step this chain computes (see ChainParams::skip_prefix)
    if (kb >= P.T) return;                                              // (a chain without a restart that is not the provider: nothing of its own)
    const int t_first = BWD ? P.T - 1 : kb;
    double xd[DMAX], xn[DMAX];
#pragma unroll
    for (int q = 0; q < DMAX; ++q) xd[q] = q < P.d ? P.rec[(long long)t_first * P.rec_len + q] : __builtin_nan("");
    // (TALL: 1024 rows leave no room for a whole step of stored alpha beside the new state -- a ring of ALD tiles, re-filled ALD tiles ahead)
    constexpr int ALD = NTW > 4 ? 2 : NTW;        // (a divisor of NTW: a tile keeps its slot from step to step)
    double al[ALD][4];                     // backward: the stored alpha of the lane's cells; a slot is re-filled for the NEXT step right
    if (BWD) {                             // after it has been consumed, i.e. a whole step before its next use
#pragma unroll
        for (int it = 0; it < NTW; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) if (it < ALD) al[it][r] = blm::ld32(pchain + (long long)t_first * G, cell_off(lane, it, r));
    }
    // NK = 4 (no stencil): a step is elementwise, so the state of the lane's 4 NTW cells never leaves its registers -- and neither does
    // the reset distribution a change point restarts from (loaded once).  No LDS traffic, no memory instruction under control flow
    // (the compiler's wait-count bookkeeping drains every outstanding access where paths with different numbers of them join).
    double stt[FILTER ? 1 : NTW][4], rst[FILTER ? 1 : NTW][4];
    if (!FILTER) {
#pragma unroll
        for (int it = 0; it < NTW; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + it * TM + (lane >> 4) + 4 * r;
                const bool in = !PAD || (row < n0t && colok);
                const long long cell = in ? (long long)row * n1t + gj : 0;
                stt[it][r] = in ? P.src0[cell] : 0.0;
                rst[it][r] = (in && P.kinds) ? P.reset[cell] : 0.0;
            }
    }
    constexpr bool FOLD = BWD && !STORE;
    double *const pslot = FOLD ? P.part + (long long)cs * P.part_stride : nullptr;
    const double wch = FOLD ? P.wchain[b] : 0.0;
    double inpred = FOLD ? P.infirst[b] : 0.0;        // 1 / predicted sum of the step's posterior
    double sfn = 1.0;                                 // the forward scale the NEXT step's prediction needs
    int kind_n = blk::SRC_PREV;                      // NK = 4: source kind of the next step (the first step's source is in LDS already)
    if (kb > 0) kind_n = P.kinds[(long long)kb * P.B + b];             // (a restart: the step takes the reset distribution)
    double pa[4] = {0.0, 0.0, 0.0, 0.0};              // fold: the accumulator cells of the tile in flight, requested one tile ahead
    if (FOLD) {
#pragma unroll
        for (int r = 0; r < 4; ++r) pa[r] = P.part_fresh ? blm::ld32(P.zeros, cell_off(lane, 0, r) & 4088u) : blm::ld32(pslot + (long long)t_first * G, cell_off(lane, 0, r));
    }
    // the waves with extra duties (scale, block sums, reflection at the grid edges) arrive last at the step's barrier: they get issue
    // priority over the wave they share a SIMD with (measured: forward 185 -> 180 us per logical step of C4)
    if (wv == SCALE_WAVE || wv == 5 || wv == 0 || wv == NW - 1) __builtin_amdgcn_s_setprio(2);
    bool dead = false;
    typedef const double __attribute__((address_space(3))) *lds_cp;
    // the granule of the sum step k + 1 divides by is requested when step k begins (it was published a step before that, lag >= 3:
    // a load that crosses XCDs takes a few thousand cycles, more than the first product chain of a step would hide)
    unsigned long long gq0 = 0ull, gq1 = 0ull;
    double Sprev = 1.0;                            // the sum the previous step's scale was made of
    double mq = 1.0, iq = 1.0, dn_prev = -1.0;     // exp(second difference of the exponent): changes only with the number of valid data dimensions
    int nq = 0;
    // the anchors of the step in flight (ChainParams::anch), requested a whole step ahead
    // (the filtering kernels; the no-stencil ones -- change-point batches: a chain-step is a few hundred cycles, chains of a launch run at
    //  different time steps once prefixes are skipped -- evaluate their anchors themselves: C5's forward pass lost 30 % with the table)
    constexpr bool ANCT = !TAB && FILTER;
    AnchorEntry anc{1.0, 1.0, 0.0, 0.0};
    if constexpr (ANCT) anc = anchor_load(P.anch, t_first, wv, P.strips, tj, lane);
#ifdef BLC_PROF
    const bool prof_me = blockIdx.x == 0 && lane == 0 && (wv == 0 || wv == 2);
#endif
    __syncthreads();
    // OWNREG (forward kernels of exact geometries of <= 512 rows): a wave's OWN rows of the state stay in its registers from step to step --
    // the ring entries over them are exactly the epilogue's outputs of the step before, lane for lane (both are "row + g + 4 r, column c").
    // LDS then carries only what a neighbour reads: the R0 rows next to a wave's boundaries (and the mirrored rows of the grid's edges).
    // Per wave and step of a 512-row strip with R0 = 20: 10 ring reads instead of 26, 10 state writes instead of 16 (with the band's
    // entries in registers -- band_products_w -- 31 LDS accesses of 512 bytes instead of 86).
#ifdef BLC_NO_WHOLE_RING
    constexpr bool OWNREG = false;
#else
    constexpr bool OWNREG = BLC_OWNREG && BLC_FASTEDGE && FILTER && !BWD && NTW <= 4 && !PAD && NTW * TM >= R0;
#endif
#if BLC_HOIST_LOOP
    // the band's entries of this lane in registers (band_products_w): read when the band changes -- the identity of the first step, the
    // chain's band from the second on -- instead of 11 .. 21 LDS reads per step (c4 evidence-only 66.8 -> 64.5 ms)
    constexpr bool HOISTA = BLC_HOIST_A && BLC_BAND4 && FILTER && NTW >= 2 && NK <= (BWD ? 20 : 24);      // (the storing backward kernels of the longest rings have no 2 (NK - 3) registers to spare)
    double Aw[(HOISTA && NK > 3) ? NK - 3 : 1];
#endif
    double own[OWNREG ? NTW : 1][4];
    auto own_from_lds = [&](const double *Xs) {
        const int l = fresh_lane(), g = l >> 4, c = l & 15;
#pragma unroll
        for (int it = 0; it < NTW; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) own[OWNREG ? it : 0][r] = Xs[(row0 + it * TM + g + 4 * r) * WCOL + c];
    };
    if constexpr (OWNREG) own_from_lds(X);

    for (int k = kb; k < P.T; ++k) {
        const int t = BWD ? P.T - 1 - k : k;
        const int tn = (k + 1 < P.T) ? (BWD ? t - 1 : t + 1) : t;          // the step after this one (clamped: a harmless re-load)
        BLC_STAMP(0);
        // ---- the scale wave: the sums the scale of step k + 1 is made of were requested a step ago (gq0 / gq1); those for step k + 2
        //      are requested now -------------------------------------------------------------------------------------------------------
        const bool scale_wave = wv == SCALE_WAVE;
        const int jn = k + 1;                                               // the step whose scale this step prepares
        const bool need = scale_wave && jn >= kb + P.lag && jn < P.T;
        const unsigned long long *gp = P.gran + ((((long long)((jn - P.lag) & (NSLOT - 1)) * P.nslots + cs) * P.strips + lane) << 1);
        const bool mine = need && lane < P.strips;
        const unsigned long long hq0 = gq0, hq1 = gq1;
        if (scale_wave && jn + 1 >= kb + P.lag && jn + 1 < P.T && lane < P.strips) {
            const unsigned long long *gn = P.gran + ((((long long)((jn + 1 - P.lag) & (NSLOT - 1)) * P.nslots + cs) * P.strips + lane) << 1);
            gq0 = blr::ld_u64(gn); gq1 = blr::ld_u64(gn + 1);
        }
        // ---- what the NEXT step needs from HBM: its data record (its stored alpha: see the epilogue) --------------------------------
        const double *const pnext = (BWD && tn < tsh ? P.post + (long long)P.bprov * P.post_stride : pchain) + (long long)tn * G;
        const int kind = kind_n & 0x7f;                          // what this step consumes; bit 7: ... without the filter
        const bool nofilter = FILTER && (kind_n & 0x80) != 0;
        if (P.kinds) kind_n = P.kinds[(long long)tn * P.B + b];
        if (!FILTER && kind != blk::SRC_PREV) {               // a change point: the chain restarts from the reset distribution
#pragma unroll
            for (int it = 0; it < NTW; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) stt[it][r] = rst[it][r];
        }
        const double sf_now = sfn;
        if (FOLD) sfn = P.sfwd[(long long)b * P.T + min(tn + 1, P.T - 1)];
#pragma unroll
        for (int q = 0; q < DMAX; ++q) xn[q] = q < P.d ? P.rec[(long long)tn * P.rec_len + q] : __builtin_nan("");
        AnchorEntry anc_next{1.0, 1.0, 0.0, 0.0};
        if constexpr (ANCT) anc_next = anchor_load(P.anch, tn, wv, P.strips, tj, fresh_lane());

        // ---- ring over the source state -----------------------------------------------------------------------------------------------
        double *S = TALL ? X : X + (k & 1) * XSZ;
        double *D = TALL ? X : X + ((k + 1) & 1) * XSZ;
        double nst[(TALL && FILTER) ? NTW : 1][4];     // (TALL) the step's new state, written after the barrier
        if (FILTER && kind != blk::SRC_PREV && k > 0) {
            // a change point inside a filtering chain (random walk + change point in one model): this step's source buffer takes the
            // reset distribution (rare: once per chain and change point; the loads are consumed inside the branch)
            for (int e = tid; e < XSZ; e += NT) {
                const int row = e >> 4, col = tj * WCOL + (e & 15);
                S[e] = (!PAD || (row < n0t && col < n1t)) ? P.reset[(long long)row * n1t + col] : 0.0;
            }
            __syncthreads();
            if constexpr (OWNREG) own_from_lds(S);
        }
        // (interior waves: consecutive k-blocks are 512 bytes apart -- one address register, immediate offsets; only waves whose
        //  window reaches beyond the grid edge pay for the reflection)
        const bool edge = row0 < R0 || row0 + NTW * TM + R0 > n0t;          // (the reflection is at the grid's true last row)
        // (exact geometries whose waves own >= R0 rows: only the first / last wave reflect, at compile-time entries -- ring_fill_*_wave)
        constexpr bool FASTEDGE = BLC_FASTEDGE && FILTER && !PAD && NTW * TM >= R0;
        // WHOLE_RING (forward kernels of <= 512 rows): the ring entries of ALL the wave's tiles are read when the step begins (NK + 4 (NTW - 1)
        // registers instead of NK) and a tile's products take their window at a compile-time offset -- no ring shift, no reads and no
        // edge test between the tiles.  (The sliding ring cost 20 v_mov per tile where its edge / interior load paths joined: a fifth of the
        // forward kernel's vector instructions were moves.  The two-chain backward kernel fits it too and does not gain: profiles/r04_notes.md.)
#ifdef BLC_NO_WHOLE_RING
        constexpr bool WHOLE_RING = false;
#else
        constexpr bool WHOLE_RING = FILTER && !BWD && NTW <= 4;
#endif
        constexpr int NRING = WHOLE_RING ? NK + 4 * (NTW - 1) : NK;
        double Bv[NRING];
        if constexpr (OWNREG) {
            constexpr int H = R0 / 4;                // ring entries on either side of the wave's own rows
            const int l = fresh_lane(), g = l >> 4, c = l & 15;
            const double *s0 = S + (row0 - R0 + g) * WCOL + c;
            if (wv == 0) ring_fill_first_wave(Bv, S, g, c, 0, H, -R0);
            else {
#pragma unroll
                for (int kb = 0; kb < H; ++kb) Bv[kb] = s0[kb * 4 * WCOL];
            }
            if (wv == NW - 1) ring_fill_last_wave<N0, NTW * TM>(Bv, S, g, c, H + 4 * NTW, H, NTW * TM);
            else {
#pragma unroll
                for (int kb = H + 4 * NTW; kb < NRING; ++kb) Bv[kb] = s0[kb * 4 * WCOL];
            }
#pragma unroll
            for (int it = 0; it < NTW; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) Bv[H + 4 * it + r] = own[OWNREG ? it : 0][r];
        } else if (FILTER) {
            const int l = fresh_lane(), g = l >> 4, c = l & 15;
            if (FASTEDGE && wv == 0) ring_fill_first_wave(Bv, S, g, c, 0, NRING, -R0);
            else if (FASTEDGE && wv == NW - 1) ring_fill_last_wave<N0, NTW * TM>(Bv, S, g, c, 0, NRING, -R0);
            else if (!FASTEDGE && edge) {
#pragma unroll
                for (int kb = 0; kb < NRING; ++kb) Bv[kb] = S[reflect1(row0 - R0 + 4 * kb + g, n0t) * WCOL + c];
            } else {
                const double *s0 = S + (row0 - R0 + g) * WCOL + c;
#pragma unroll
                for (int kb = 0; kb < NRING; ++kb) Bv[kb] = s0[kb * 4 * WCOL];
            }
        }

        // (TAB) the likelihood of the lane's cells at this step: requested in front of the products, consumed by the epilogues
        double lk[TAB ? NTW : 1][4];
        if (TAB) {
            const double *const lrow = P.lik + (long long)t * n0t * n1t;
            const int l = fresh_lane();
#pragma unroll
            for (int it = 0; it < NTW; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = row0 + it * TM + (l >> 4) + 4 * r;
                    const unsigned off = __umul24(PAD ? min(row, n0t - 1) : row, (unsigned)n1t * 8u) + (unsigned)(PAD ? min(tj * WCOL + (l & 15), n1t - 1) : tj * WCOL + (l & 15)) * 8u;
                    const double v = blm::ld32(lrow, off);
                    lk[TAB ? it : 0][r] = (!PAD || (row < n0t && colok)) ? v : 0.0;
                }
        }

        BLC_STAMP(1);
        double scale = 1.0;
        double mE = 1.0, mR = 1.0, iE = 1.0, iR = 1.0;
        int nE = 0, nR = 0;
        double sN = 0.0, sS = 0.0, sC = 0.0, sM0 = 0.0, sM1 = 0.0;
        // forward pass of a change-point batch: the step BEFORE a restart also sums its new state times the reset distribution --
        // the sum of the posterior at the restart of the backward pass, which the fused fold normalises by (chain_fold2_kernel)
        const bool want_x = !BWD && !FILTER && P.kinds && (kind_n != blk::SRC_PREV || (P.skip_prefix && b == P.bprov));
        double *const pstep = pchain + (long long)t * G;
        double *const pslot_t = FOLD ? pslot + (long long)t * G : nullptr;
        double *const pslot_tn = FOLD ? pslot + (long long)tn * G : nullptr;
        double wq = 0.0, wfloor = 0.0;                 // w / N_t and w * 1e-300: w max(p / N, 1e-300) = max(p wq, wfloor)

        // the band's entries of this lane, once per step for all the wave's tiles (band_products_w)
#if !BLC_HOIST_LOOP
        constexpr bool HOISTA = BLC_HOIST_A && BLC_BAND4 && FILTER && NTW >= 2 && NK <= (BWD ? 20 : 24);      // (the storing backward kernels of the longest rings have no 2 (NK - 3) registers to spare)
#endif
#if BLC_HOIST_LOOP
        if (HOISTA && k <= kb + 1) {
#else
        double Aw[(HOISTA && NK > 3) ? NK - 3 : 1];
        if constexpr (HOISTA) {
#endif
            const int l = fresh_lane();
            const unsigned aoff = AST == 16 ? (unsigned)(((l >> 4) << 2) | (l & 3)) * 8u : (unsigned)l * 8u;
            lds_cp Al = (lds_cp)((const char __attribute__((address_space(3))) *)(lds_cp)As + aoff);
#pragma unroll
            for (int q = 0; q < NK - 3; ++q) Aw[q] = Al[q * AST];
        }
#pragma unroll
        for (int it = 0; it < NTW; ++it) {
            const int i = row0 + it * TM;
            const int l = fresh_lane(), g = l >> 4, c = l & 15;
            // ---- axis-0 stencil: NK chained matrix products (k ascending) -----------------------------------------------------------
            d4 acc = {0.0, 0.0, 0.0, 0.0};
            if (!FILTER) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = stt[it][r];          // no stencil: the product tile IS the state
            } else {
                // (the band stays in LDS: hoisted out of the time loop it costs 2 NK VGPRs)
                const unsigned aoff = AST == 16 ? (unsigned)(((l >> 4) << 2) | (l & 3)) * 8u : (unsigned)l * 8u;
                lds_cp Al = (lds_cp)((const char __attribute__((address_space(3))) *)(lds_cp)As + aoff);
                if (nofilter) {                              // (the change point comes after the walk in the model's list: the source unfiltered)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = OWNREG ? own[OWNREG ? it : 0][r] : S[(i + g + 4 * r) * WCOL + c];
                } else if constexpr (WHOLE_RING && HOISTA) {
                    acc = it == 0 ? band_products_w<NK, 0, NRING>(Aw, Bv) : (it == 1 ? band_products_w<NK, (NTW > 1 ? 4 : 0), NRING>(Aw, Bv) :
                          (it == 2 ? band_products_w<NK, (NTW > 2 ? 8 : 0), NRING>(Aw, Bv) : band_products_w<NK, (NTW > 3 ? 12 : 0), NRING>(Aw, Bv)));
                } else if constexpr (WHOLE_RING) {
                    // (`it` is a compile-time constant of the unrolled loop: one instantiation per tile)
                    acc = it == 0 ? band_products<NK, 0, NRING, AST>(Al, Bv) : (it == 1 ? band_products<NK, (NTW > 1 ? 4 : 0), NRING, AST>(Al, Bv) :
                          (it == 2 ? band_products<NK, (NTW > 2 ? 8 : 0), NRING, AST>(Al, Bv) : band_products<NK, (NTW > 3 ? 12 : 0), NRING, AST>(Al, Bv)));
                } else if constexpr (HOISTA) {
                    acc = band_products_w<NK, 0, NK>(Aw, Bv);
                } else {
                    acc = band_products<NK, 0, NK, AST>(Al, Bv);
                }
            }

            if (it == 0) {
                BLC_STAMP(2);
                // ---- the scale of this step (prepared during the previous one); the scale wave prepares the next one: it sums the
                //      strips' granules in one fixed order, so every strip of the chain arrives at bit-identical scales ----------------
                scale = scal[k & (NSLOT - 1)];
                if (FOLD && k > 0) inpred *= sf_now * iscal[k & (NSLOT - 1)];      // N_t = s'_t N_(t+1) / s_(t+1)
                if (FOLD) { wq = wch * inpred; wfloor = wch * 1e-300; }
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
                    if (lane == 0) { scal[jn & (NSLOT - 1)] = sj; if (BWD && !STORE) iscal[jn & (NSLOT - 1)] = 1.0 / sj; }
                }
                // ---- anchor of the stride-4 likelihood recurrence of this lane's rows (blhip_mfma.hpp) -------------------------------
                // arg(r) = sum_k [-(x_k - mu_r)^2 cA - cB]  (observationModels.py:566-567; product over dimensions :49-50)
                if constexpr (!TAB) {
                    double dn;
                    if constexpr (ANCT) {                // (tabulated once per fit: anchor_table_kernel -- the same operations, bit for bit)
                        anchor_unpack(anc, mE, nE, mR, nR);
                        dn = anc.dn;
                    } else {
                        double a0, d1;
                        anchor_terms(xd[0], xd[1], xd[2], xd[3], m0s[i + g], m0s[i + g + 4], cA, cB, a0, d1, dn);
                        exp_mn(a0, mE, nE);
                        exp_mn(d1, mR, nR);
                    }
                    if (dn != dn_prev) {                 // (wave-uniform: the records are)
                        const double d2 = -32.0 * cA * dn * P.step0 * P.step0;
                        int tmp;
                        exp_mn(d2, mq, nq);
                        if (BWD) exp_mn(-d2, iq, tmp);
                        dn_prev = dn;
                    }
                    if (BWD) {
                        iE = blmath::inv_m(mE); iR = blmath::inv_m(mR);      // (exp(-a0), exp(-d1): same exponents, reciprocal mantissas)
                    } else {
                        mE *= scale;                     // forward: the scale rides on the likelihood's mantissa (one product per cell less)
                    }
                }
                BLC_STAMP(3);
            }

            // ---- epilogue: the lane's 4 cells (rows i + g + 4 r) ---------------------------------------------------------------------
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int li = i + g + 4 * r;
                const double Lv = TAB ? (BWD ? lk[TAB ? it : 0][r] : lk[TAB ? it : 0][r] * scale) : ldexp(mE, nE);
                const unsigned off = cell_off(l, it, r);
                if (!BWD) {
                    const double a = (!PAD || (colok && li < n0t)) ? acc[r] * Lv : 0.0;          // (cells outside the grid stay zero)
                    if constexpr (OWNREG) {
                        own[OWNREG ? it : 0][r] = a;
                        // (rows nobody else reads stay out of the LDS: a neighbour's ring reaches R0 rows into this wave's)
                        if (it * TM + 4 * r < R0 || it * TM + 4 * r + 4 > NTW * TM - R0) D[li * WCOL + c] = a;
                    } else if (TALL && FILTER) nst[it][r] = a; else if (FILTER) D[li * WCOL + c] = a; else stt[it][r] = a;
                    if (STORE && t >= tsh) stnt(pstep, off, a);
                    sN += a;
                    if (!FILTER && want_x) sS = fma(a, rst[it][r], sS);
                    acc[r] = a;
                } else {
                    const bool in = !PAD || (colok && li < n0t);                                 // (cells outside the grid stay zero)
                    const double beta = in ? acc[r] * scale : 0.0;
                    const double p = al[it % ALD][r] * beta;
                    const double cn = beta * Lv;
                    // p / L: reciprocal recurrence (no division, no intermediate overflow); 0/0 -> NaN (core.py:463)
                    const double pl = !in ? 0.0 : (TAB ? p / Lv : nan_if(Lv == 0.0, ldexp(p * iE, -nE)));      // (0 / 0 -> NaN either way)
                    if (TALL && FILTER) nst[it][r] = cn; else if (FILTER) D[li * WCOL + c] = cn; else stt[it][r] = cn;
                    if (!FOLD) stnt(pstep, off, p);
                    else stnt(pslot_t, off, pa[r] + fmax(p * wq, wfloor));      // (nobody else touches the slot's cell during the launch)
                    sN += p;
                    sS += pl;
                    sC += cn;
                    acc[r] = p;
                }
                if constexpr (!TAB) {
                    mE *= mR; nE += nR;
                    mR *= mq; nR += nq;
                    if (BWD) { iE *= iR; iR *= iq; }
                }
            }

            if (BWD) {
                // requests, soonest use first (loads return in order): the accumulator cells of the NEXT tile, then the stored alpha
                // of this tile's cells for the next step
                if (FOLD) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // (one load instruction either way: base and offset are selected, not the instruction)
                        const double *abase = P.part_fresh ? P.zeros : (it + 1 < NTW ? pslot_t : pslot_tn);
                        const unsigned aoffs = cell_off(l, (it + 1) % NTW, r);
                        pa[r] = ldnt(abase, P.part_fresh ? (aoffs & 4088u) : aoffs);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)         // the slot's next use: this tile of the next step -- or, TALL, the tile ALD further on
                    al[it % ALD][r] = (it + ALD < NTW) ? ldnt(pstep, cell_off(l, it + ALD, r)) : ldnt(pnext, cell_off(l, it + ALD - NTW, r));
            }
            if ((BWD || STORE) && P.means) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { sM0 = fma(acc[r], m0s[i + g + 4 * r], sM0); sM1 = fma(acc[r], g1, sM1); }
            }

            if (it == 0) BLC_STAMP(4);
            if (it == NTW - 1) BLC_STAMP(5);
            // ---- advance the ring by one tile --------------------------------------------------------------------------------------
            if (FILTER && !WHOLE_RING && it + 1 < NTW) {
#pragma unroll
                for (int kb = 0; kb < NK - 4; ++kb) Bv[kb] = Bv[kb + 4];
                if (FASTEDGE && wv == NW - 1 && (it + 1) * TM + R0 + 12 >= NTW * TM) ring_fill_last_wave<N0, NTW * TM>(Bv, S, g, c, NK - 4, 4, (it + 1) * TM + R0);
                else if (!FASTEDGE && edge) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) Bv[NK - 4 + q] = S[reflect1(i + TM + R0 + 4 * q + g, n0t) * WCOL + c];
                } else {
                    const double *s1 = S + (i + TM + R0 + g) * WCOL + c;
#pragma unroll
                    for (int q = 0; q < 4; ++q) Bv[NK - 4 + q] = s1[q * 4 * WCOL];
                }
            }
        }

        // ---- sums: waves -> LDS (this step's parity); after the barrier wave 0 adds them up, writes the partial sums of the strip
        //      and publishes the one the scale of step k + lag is made of -----------------------------------------------------------------
        double v[5] = {sN, sS, BWD ? sC : sM0, BWD ? sM0 : sM1, sM1};
        constexpr int NV = BWD ? 5 : 4;                          // forward: N, X (the restart sum, no-stencil batches only), M0, M1
        // (X sits in front of the means: a change-point batch without means reduces two sums per step, not four -- the four sums were
        //  1.3 k of the 4.9 k cycles of a no-stencil chain-step)
        const int nv = BWD ? (P.means ? 5 : 3) : (P.means ? 4 : ((!FILTER && P.kinds) ? 2 : 1));
        // (a full wave reduction costs ~45 vector instructions per sum and wave; the waves reduce only within their rows of 16 lanes
        //  -- 12 instructions -- and one wave adds the 32 row sums of the block after the barrier, in a fixed order)
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
        BLC_STAMP(6);
        __syncthreads();
        BLC_STAMP(7);
        if (TALL && FILTER) {              // every ring of this step has been read: the new state takes the strip's place
            const int l = fresh_lane(), g = l >> 4, c = l & 15;
#pragma unroll
            for (int it = 0; it < NTW; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) D[(row0 + it * TM + g + 4 * r) * WCOL + c] = nst[it][r];
        }
        if (FILTER && k == 0) {            // the chain's band replaces the identity of the first step
            for (int e = tid; e < NK * AST; e += NT) {
                const int a = AST == 16 ? band_distance16(e, R0) : band_distance(e, R0);
                As[e] = a == 0 ? (lw0 > 0 ? P.taps[o0] : 1.0) : (a <= lw0 ? P.taps[o0 + a] : 0.0);
            }
        }
        if (FILTER && (TALL || k == 0)) __syncthreads();
        if (wv == 5 && lane < nv) {           // (a wave that shares its SIMD with no edge wave)
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < NW * 4; ++w) tot += rk[w * 5 + lane];
            const int slot = BWD ? lane : (lane < 2 ? lane : lane + 1);                          // forward: N, X, M0, M1 -> slots 0, 1, 3, 4
            if (lane == 0 || (BWD ? (P.means || lane < 3) : (lane == 1 ? (!FILTER && P.kinds != nullptr) : P.means != 0)))
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
        if constexpr (ANCT) anc = anc_next;
    }
}


// ---- backward pass of a hyper-study with the fused fold, TWO chains per block ---------------------------------------------------------
// The fold is a read-modify-write of the partial accumulator: 16 of the 24 bytes a chain's backward step moves per cell (the other 8:
// its stored alpha), and the pass runs at the memory roof for all but the widest bands.  Here a block owns one strip of TWO chains:
// both add their weighted posteriors to the accumulator cell while it is in registers -- 8 + 16 / 2 = 16 B per chain, cell and step.
//  * LDS has room for one 64 KB buffer per chain, not two: a chain's state (the lane's 4 NTW cells) lives in REGISTERS; LDS is the
//    exchange buffer the waves read their product rings from.  Per chain and step: the lanes write their cells to the chain's buffer,
//    ONE barrier, rings + products + epilogue (new cells -> registers).  The chains alternate, so the barrier of one chain's step
//    also fences the other chain's buffer between its readers and its next writers;
//  * the stored alpha is requested one CHAIN-step ahead into one register set (the slot a tile's epilogue has just consumed is
//    re-filled for the other chain); the accumulator cells of a step are requested during the step before, updated by chain 0,
//    updated and stored by chain 1;
//  * both chains see the same likelihood: the second one starts its recurrence from the first one's anchors (no exponentials).
// Everything else -- bands on the matrix pipe, lagged-normaliser scales from data-tagged granules, predicted posterior sums, bounded
// spins -- is chain_kernel<NK, NTW, true, false>'s.  Launched for rounds of 2 x (CUs / strips) chains when the batch folds.
template <int NK, int NTW>
constexpr size_t lds_doubles_fold2() { return (size_t)2 * NW * NTW * TM * WCOL + 2 * NK * (NK > 24 ? 16 : 64) + NW * NTW * TM + 2 * 2 * NW * 4 * 3 + 4 * NSLOT + 8; }      // (NK = 4: the band tables stay unused)

template <int NK, int NTW, bool PAD = false>
__global__ __launch_bounds__(NT, 1) void chain_fold2_kernel(const ChainParams P) {
    constexpr int R0 = (4 * NK - TM) / 2;
    constexpr int N0 = NW * NTW * TM;
    constexpr int XSZ = N0 * WCOL;
    const int n0t = PAD ? P.n0t : N0, n1t = PAD ? P.n1t : P.n1;      // the grid's true sizes (PAD: see chain_kernel)
    static_assert(NK == 4 || (NK >= 6 && R0 % 4 == 0), "band = 16 + 2 R0 columns, R0 a multiple of 4; NK = 4: no stencil");
    // NK = 4: no stencil at all (change-point studies: every chain is Static except for the steps that restart from the reset
    // distribution, transitionModels.py:300-312).  A step is elementwise: a lane reads and writes only its own cells of the chain's
    // buffer; at a restart of the backward pass the sum of the posterior is s' sum(alpha reset), which the forward pass has summed
    // (P.sfwd carries it in place of the forward scale the identity does not need there).
    constexpr bool FILTER = NK > 4;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *const X = lds;                          // [2 chains][N0][16]   exchange buffers
    // (bands beyond radius 40 -- NK > 24 -- keep the compact table of band_products: two tables of 44 x 64 doubles would not fit beside the buffers)
    constexpr int AST = (NK > 24 && BLC_BAND4) ? 16 : 64;
    double *const As = X + 2 * XSZ;                 // [2 chains][NK][AST]  A operands
    double *const m0s = As + 2 * NK * AST;          // [N0]
    double *const red = m0s + N0;                   // [2 chains][2 parities][NW * 4][3]
    double *const scal = red + 2 * 2 * NW * 4 * 3;  // [2 chains][NSLOT]
    double *const iscal = scal + 2 * NSLOT;         // [2 chains][NSLOT]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cs2 = blockIdx.x / P.strips, tj = blockIdx.x - cs2 * P.strips;
    const int nch = min(2, P.nslots - 2 * cs2);                   // chains of this block (the last pair of a launch may be single)
    int bch[2], tap[2], lw0[2], tshv[2];
    long long o0[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        bch[j] = sldi(P.chain_ids, 2 * cs2 + min(j, nch - 1));
        tshv[j] = P.tshare ? sldi(P.tshare, bch[j]) : 0;              // steps t < tshv: the stored states are chain P.bprov's (see ChainParams)
        tap[j] = sldi(P.tap_id, bch[j]);
        lw0[j] = tap[j] >= 0 ? sldi(P.tap_lw, tap[j]) : 0;
        o0[j] = tap[j] >= 0 ? sldi(P.tap_off, tap[j]) : 0;
    }
    const int gj = tj * WCOL + (lane & 15);
    const long long G = (long long)P.n0 * P.n1;

    // first step: the source (uniform) is consumed unfiltered -> identity bands; the chains' bands replace them after step 0
    if (FILTER) for (int e = tid; e < 2 * NK * AST; e += NT) As[e] = (AST == 16 ? band_distance16(e % (NK * AST), R0) : band_distance(e % (NK * AST), R0)) == 0 ? 1.0 : 0.0;
    for (int e = tid; e < N0; e += NT) m0s[e] = P.m0[PAD ? min(e, n0t - 1) : e];
    if (tid < 4 * NSLOT) scal[tid] = 1.0;
    for (int e = tid; e < 2 * XSZ; e += NT) {
        const int q = e % XSZ, row = q >> 4, col = tj * WCOL + (q & 15);
        X[e] = (!PAD || (row < n0t && col < n1t)) ? P.src0[(long long)row * n1t + col] : 0.0;
    }
    const bool colok = !PAD || gj < n1t;
    const int gjc = PAD ? min(gj, n1t - 1) : gj;
    const double cA = P.colA[gjc], cB = P.colB[gjc];
    const unsigned rowx8 = (unsigned)WCOL * 8u;                    // (strip-major sequences: the fold is private to the fit)
    const unsigned strip0 = (unsigned)tj * (unsigned)(P.n0 * WCOL * 8);
    const int row0 = wv * (NTW * TM);
    auto fresh_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
    auto cell_off = [&](int l, int it, int r) { return __umul24(row0 + it * TM + (l >> 4) + 4 * r, rowx8) + strip0 + (unsigned)(l & 15) * 8u; };

    const int t_first = P.T - 1;
    double *const pslot = P.part + (long long)cs2 * P.part_stride;
    double stt[NTW][4];                             // the state (c = beta L of the lane's cells) the LAST chain-step produced: it goes to
                                                    // that chain's exchange buffer right after the next barrier (nobody reads it then)
    double al[NTW][4];                              // stored alpha of the chain-step that runs next
    double pacc[NTW][4];                            // accumulator cells of the time step in flight
    {
        const double *p0 = P.post + (long long)bch[0] * P.post_stride + (long long)t_first * G;
#pragma unroll
        for (int it = 0; it < NTW; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                al[it][r] = blm::ld32(p0, cell_off(lane, it, r));
                pacc[it][r] = P.part_fresh ? blm::ld32(P.zeros, cell_off(lane, it, r) & 4088u) : blm::ld32(pslot + (long long)t_first * G, cell_off(lane, it, r));
                stt[it][r] = 0.0;
            }
    }
    double wch[2], inpred[2];
    double sf_next = 1.0;                           // the forward scale the NEXT chain-step's prediction needs (requested a chain-step ahead
                                                    // and not touched before: a value that is used right after its request makes the wave
                                                    // wait for everything it has in flight)
#pragma unroll
    for (int j = 0; j < 2; ++j) { wch[j] = j < nch ? P.wchain[bch[j]] : 0.0; inpred[j] = P.infirst[bch[j]]; }
    if (wv == SCALE_WAVE || wv == 5 || wv == 0 || wv == NW - 1) __builtin_amdgcn_s_setprio(2);
    bool dead = false;
    typedef const double __attribute__((address_space(3))) *lds_cp;
    unsigned long long gq0[2] = {0ull, 0ull}, gq1[2] = {0ull, 0ull};
    double Sprev[2] = {1.0, 1.0};
    double mq = 1.0, iq = 1.0, dn_prev = -1.0;
    int nq = 0;
    // the first chain's anchors of the step, reused by the second one
    double a_mE = 1.0, a_mR = 1.0, a_iE = 1.0, a_iR = 1.0;
    int a_nE = 0, a_nR = 0;
    // ... from the table of the fit (ChainParams::anch), requested a whole time step ahead
    AnchorEntry anc_next = anchor_load(P.anch, t_first, wv, P.strips, tj, lane);
    int pend_j = -1, pend_k = 0;                   // the chain-step whose row sums wave 5 still has to add up (after the next barrier)
    int kind_next = blk::SRC_PREV;                 // NK = 4: source kind of the chain-step after this one (the first steps consume src0)

#ifdef BLC_PROF
    const bool prof_me = blockIdx.x == 0 && lane == 0 && (wv == 0 || wv == 2);
#define BLC_STAMP2(i) do { if (prof_me && cstep >= 16 && cstep < 32) P.prof[(wv ? 256 : 0) + (cstep - 16) * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define BLC_STAMP2(i) do { } while (0)
#endif
    const LoopParams Q = loop_params(P);
    // wave 5: block totals of a finished chain-step -> partial sums of the strip + the granule the scale of step k + lag is made of
    auto totals = [&](int j, int k) {
        if (wv == 5 && lane < 3) {
            const double *rk = red + (j * 2 + (k & 1)) * (NW * 4 * 3);
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < NW * 4; ++w) tot += rk[w * 3 + lane];
            const int t = Q.T - 1 - k;
            Q.psum[(((long long)t * Q.B + bch[j]) * NRED + lane) * Q.nblk + tj] = tot;
            if (lane == 2) {
                const unsigned long long bits = (unsigned long long)__double_as_longlong(tot);
                const unsigned long long tag = (unsigned long long)(unsigned)(k + 1) << 32;
                unsigned long long *gw = Q.gran + ((((long long)(k & (NSLOT - 1)) * Q.nslots + 2 * cs2 + j) * Q.strips + tj) << 1);
                blr::st_u64(gw, tag | (bits & 0xffffffffull));
                blr::st_u64(gw + 1, tag | (bits >> 32));
            }
        }
    };
    __syncthreads();

    // ONE body for both chains of the block (j is block-uniform and changes every iteration): the two chain-steps of a time step are
    // not unrolled into one another -- unrolled, the scheduler interleaves their loads and the kernel spills (measured: 54 - 93 VGPRs)
    const bool scale_wave = wv == SCALE_WAVE;
    const int nsteps = Q.T * nch;
#pragma unroll 1
    for (int cstep = 0; cstep < nsteps; ++cstep) {
        const int k = nch == 2 ? cstep >> 1 : cstep;
        const int j = __builtin_amdgcn_readfirstlane(nch == 2 ? cstep & 1 : 0);
        const int t = Q.T - 1 - k;
        const int tn = (k + 1 < Q.T) ? t - 1 : t;
        const int jn = k + 1;
        const int bj = j ? bch[1] : bch[0];
        BLC_STAMP2(0);
        double *const pslot_t = pslot + (long long)t * G;
        double *const pslot_tn = pslot + (long long)tn * G;
        double *const Xj = X + j * XSZ;
        // scale wave: the granules of the sums the scale of step k + 2 of this chain is made of (requested a step ahead)
        const bool need = scale_wave && jn >= Q.lag && jn < Q.T;
        const unsigned long long *gp = Q.gran + ((((long long)((jn - Q.lag) & (NSLOT - 1)) * Q.nslots + 2 * cs2 + j) * Q.strips + lane) << 1);
        const bool mine = need && lane < Q.strips;
        const unsigned long long hq0 = j ? gq0[1] : gq0[0], hq1 = j ? gq1[1] : gq1[0];
        if (scale_wave && jn + 1 >= Q.lag && jn + 1 < Q.T && lane < Q.strips) {
            const unsigned long long *gn = Q.gran + ((((long long)((jn + 1 - Q.lag) & (NSLOT - 1)) * Q.nslots + 2 * cs2 + j) * Q.strips + lane) << 1);
            const unsigned long long n0_ = blr::ld_u64(gn), n1_ = blr::ld_u64(gn + 1);
            if (j) { gq0[1] = n0_; gq1[1] = n1_; } else { gq0[0] = n0_; gq1[0] = n1_; }
        }
        const int kind = kind_next;
        if (!FILTER && Q.kinds) {                   // (requested a chain-step ahead: a dependent scalar load in front of the cells costs a round trip)
            const bool lastc = j + 1 == nch;
            const int tq = lastc ? tn : t, kq = lastc ? k + 1 : k;          // (the first step of a chain consumes src0, which its buffer holds)
            kind_next = (kq == 0 || kq >= Q.T) ? blk::SRC_PREV : (int)Q.kinds[(long long)tq * Q.B + (lastc ? bch[0] : bch[1])];
        }
        const double sf_now = sf_next;
        {
            const bool lastc = j + 1 == nch;
            sf_next = Q.sfwd[(long long)(lastc ? bch[0] : bch[1]) * Q.T + min((lastc ? tn : t) + 1, Q.T - 1)];
        }
        __syncthreads();                                            // every wave's rows of this chain are in Xj
        BLC_STAMP2(1);
        if (pend_j >= 0) {
            // the previous chain-step's new state -> that chain's exchange buffer: all its readers have passed the barrier above, its
            // next readers wait at the next one  (NK = 4: a lane's cells are its own -- written in the epilogue)
            if (FILTER) {
                double *const Xp = X + pend_j * XSZ;
                const int l = fresh_lane(), g = l >> 4, c = l & 15;
#pragma unroll
                for (int it = 0; it < NTW; ++it)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Xp[(row0 + it * TM + g + 4 * r) * WCOL + c] = stt[it][r];
            }
            totals(pend_j, pend_k);
        }
        // a block with ONE chain (the last pair of a round with an odd number of chains) has just written the buffer it is about to read
        // its rings from: the waves' rows have to be there first.  (With two chains the write went to the OTHER chain's buffer.)
        if (FILTER && nch == 1 && pend_j >= 0) __syncthreads();
        pend_j = j; pend_k = k;
        if (!FILTER && kind != blk::SRC_PREV) {
            // a restart (once per chain and change point): the chain's buffer takes the reset distribution.  A lane's cells are its own
            // (no barrier); the values are consumed inside the branch -- nothing fetched here is pending where the paths join.  (Held in
            // registers instead, the 16 values were spilled and re-loaded from scratch before every tile: each re-load drained the
            // requests in flight -- 62 % of the wave time parked.)
            const int l = fresh_lane(), g = l >> 4, c = l & 15;
#pragma unroll
            for (int it = 0; it < NTW; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                {
                    const int row = row0 + it * TM + g + 4 * r;
                    const bool in = !PAD || (row < n0t && colok);
                    Xj[row * WCOL + c] = in ? Q.reset[(long long)row * n1t + tj * WCOL + c] : 0.0;
                }
        }

        const bool edge = row0 < R0 || row0 + NTW * TM + R0 > n0t;          // (the reflection is at the grid's true last row)
        constexpr bool FASTEDGE = BLC_FASTEDGE && FILTER && !PAD && NTW * TM >= R0;       // (see ring_fill_first_wave)
        constexpr int NRM = BLC_CIRC_RING ? ring_mod<NK>() : NK;      // (circular ring: band_products_wc)
        double Bv[NRM];
        if (FILTER) {
            const int l = fresh_lane(), g = l >> 4, c = l & 15;
            if (FASTEDGE && wv == 0) ring_fill_first_wave(Bv, Xj, g, c, 0, NK, -R0);
            else if (FASTEDGE && wv == NW - 1 && R0 + 12 >= NTW * TM) ring_fill_last_wave<N0, NTW * TM>(Bv, Xj, g, c, 0, NK, -R0);
            else if (!FASTEDGE && edge) {
#pragma unroll
                for (int kb = 0; kb < NK; ++kb) Bv[kb] = Xj[reflect1(row0 - R0 + 4 * kb + g, n0t) * WCOL + c];
            } else {
                const double *s0 = Xj + (row0 - R0 + g) * WCOL + c;
#pragma unroll
                for (int kb = 0; kb < NK; ++kb) Bv[kb] = s0[kb * 4 * WCOL];
            }
        }
        BLC_STAMP2(2);
        double scale = 1.0, wq = 0.0, wfloor = 0.0;
        double mE = 1.0, mR = 1.0, iE = 1.0, iR = 1.0;
        int nE = 0, nR = 0;
        double sN = 0.0, sS = 0.0, sC = 0.0;
        // the chain-step that runs after this one: chain 1 of this step, or chain 0 of the next (its stored alpha is requested now)
        const bool last_chain = j + 1 == nch;
        const int bnext = last_chain ? bch[0] : bch[1];
        const int tnext = last_chain ? tn : t;
        const double *const pnext = Q.post + (long long)(tnext < (last_chain ? tshv[0] : tshv[1]) ? Q.bprov : bnext) * Q.post_stride + (long long)tnext * G;

        // the band's entries of this lane and chain, once per chain-step for all the wave's tiles (band_products_w) -- where the registers allow
        constexpr bool HOISTA = BLC_HOIST_A && BLC_BAND4 && FILTER && NTW >= 2 && NK <= BLC_FOLD_HOIST_MAX;
        double Aw[(HOISTA && NK > 3) ? NK - 3 : 1];
        if constexpr (HOISTA) {
            const int l = fresh_lane();
            const unsigned aoff = (AST == 16 ? (unsigned)(((l >> 4) << 2) | (l & 3)) * 8u : (unsigned)l * 8u) + (unsigned)(j * NK * AST * 8);
            lds_cp Al = (lds_cp)((const char __attribute__((address_space(3))) *)(lds_cp)As + aoff);
#pragma unroll
            for (int q = 0; q < NK - 3; ++q) Aw[q] = Al[q * AST];
        }
#pragma unroll
        for (int it = 0; it < NTW; ++it) {
            const int i = row0 + it * TM;
            const int l = fresh_lane(), g = l >> 4, c = l & 15;
            d4 acc = {0.0, 0.0, 0.0, 0.0};
            if constexpr (FILTER && HOISTA) {
                acc = band_products_wc<NK, NRM>(Aw, Bv, BLC_CIRC_RING ? 4 * it : 0);
            } else if (FILTER) {
                const unsigned aoff = (AST == 16 ? (unsigned)(((l >> 4) << 2) | (l & 3)) * 8u : (unsigned)l * 8u) + (unsigned)(j * NK * AST * 8);
                lds_cp Al = (lds_cp)((const char __attribute__((address_space(3))) *)(lds_cp)As + aoff);
                acc = band_products_c<NK, NRM, AST>(Al, Bv, BLC_CIRC_RING ? 4 * it : 0);
            } else {
                // no stencil: the product tile IS the state (at a restart: the reset distribution, written above)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = Xj[(i + g + 4 * r) * WCOL + c];
            }
            if (it == 0) {
                BLC_STAMP2(3);
                scale = scal[j * NSLOT + (k & (NSLOT - 1))];
                double ip = j ? inpred[1] : inpred[0];
                // N_t = s'_t N_(t+1) / s_(t+1); at a restart N_t = s'_t sum(alpha_t reset) (sf_now carries that sum)
                if (k > 0) ip = (!FILTER && kind != blk::SRC_PREV) ? iscal[j * NSLOT + (k & (NSLOT - 1))] / sf_now : ip * sf_now * iscal[j * NSLOT + (k & (NSLOT - 1))];
                if (j) inpred[1] = ip; else inpred[0] = ip;
                const double wc = j ? wch[1] : wch[0];
                wq = wc * ip; wfloor = wc * 1e-300;
                if (scale_wave) {
                    double sj = 1.0;
                    if (need) {
                        const unsigned long long want = (unsigned long long)(unsigned)(jn - Q.lag + 1);
                        unsigned long long q0 = hq0, q1 = hq1;
                        bool ok = !mine || ((q0 >> 32) == want && (q1 >> 32) == want);
                        if (!dead && !__all(ok)) {
                            const unsigned long long t0 = blr::now_ticks();
                            for (unsigned spins = 1; !__all(ok); ++spins) {
                                if (!ok) { q0 = blr::ld_u64(gp); q1 = blr::ld_u64(gp + 1); ok = (q0 >> 32) == want && (q1 >> 32) == want; }
                                blr::nap();
                                if ((spins & 255u) == 0u) {
                                    if (blr::ld_flag(Q.abort_word) != 0u) { dead = true; break; }
                                    if (blr::now_ticks() - t0 > Q.timeout_ticks) { blr::st_flag(Q.abort_word, 1u); dead = true; break; }
                                }
                            }
                        }
                        const double v = mine ? __longlong_as_double((long long)((q0 & 0xffffffffull) | (q1 << 32))) : 0.0;
                        const double Sg = blk::wave_sum(v);
                        const double sp = j ? Sprev[1] : Sprev[0];
                        sj = dead ? 1.0 : sp * scal[j * NSLOT + ((jn - Q.lag) & (NSLOT - 1))] / Sg;
                        if (j) Sprev[1] = Sg; else Sprev[0] = Sg;
                    }
                    if (lane == 0) { scal[j * NSLOT + (jn & (NSLOT - 1))] = sj; iscal[j * NSLOT + (jn & (NSLOT - 1))] = 1.0 / sj; }
                }
                // anchors of the stride-4 likelihood recurrence: both chains see the same likelihood -- the first one computes them
                if (j == 0) {
                    anchor_unpack(anc_next, a_mE, a_nE, a_mR, a_nR);
                    const double dn = anc_next.dn;
                    if (dn != dn_prev) {
                        const double d2 = -32.0 * cA * dn * Q.step0 * Q.step0;
                        int tmp;
                        exp_mn(d2, mq, nq);
                        exp_mn(-d2, iq, tmp);
                        dn_prev = dn;
                    }
                    a_iE = blmath::inv_m(a_mE); a_iR = blmath::inv_m(a_mR);      // (exp(-a0), exp(-d1): same exponents, reciprocal mantissas)
                    anc_next = anchor_load(Q.anch, tn, wv, Q.strips, tj, fresh_lane());          // (the next time step's: consumed a whole time step from now)
                }
                mE = a_mE; nE = a_nE; mR = a_mR; nR = a_nR; iE = a_iE; iR = a_iR;
                BLC_STAMP2(4);
            }

            // ---- epilogue: the lane's 4 cells (rows i + g + 4 r) ---------------------------------------------------------------------
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double Lv = ldexp(mE, nE);
                const bool in = !PAD || (colok && i + g + 4 * r < n0t);          // (cells outside the grid: state, posterior and p / L stay zero)
                const double beta = in ? acc[r] * scale : 0.0;
                const double p = al[it][r] * beta;
                const double cn = beta * Lv;
                const double pl = !in ? 0.0 : nan_if(Lv == 0.0, ldexp(p * iE, -nE));
                if (FILTER) stt[it][r] = cn; else Xj[(i + g + 4 * r) * WCOL + c] = cn;
                pacc[it][r] += fmax(p * wq, wfloor);
                sN += p; sS += pl; sC += cn;
                mE *= mR; nE += nR;
                mR *= mq; nR += nq;
                iE *= iR; iR *= iq;
            }
            // the last chain of the time step stores the accumulator cells (nobody else touches the slot's cells during the launch) and
            // requests those of the next time step; every chain-step requests the stored alpha of the chain-step after it (the tile's
            // slot has just been consumed)
            // (ONE 64-bit address per array and tile, the four cells at immediate offsets -- rows 4 r are 4 x 128 bytes apart in the
            //  strip-major layout: a cell_off() per access was ~200 integer instructions per chain-step)
            const unsigned off_t = cell_off(l, it, 0);
            if (last_chain) {
                char *const ps = (char *)pslot_t + off_t;
#pragma unroll
                for (int r = 0; r < 4; ++r) __builtin_nontemporal_store(pacc[it][r], (double *)(ps + r * (4 * WCOL * 8)));
                const char *const pn = Q.part_fresh ? (const char *)Q.zeros + (off_t & 4088u) : (const char *)pslot_tn + off_t;
#pragma unroll
                for (int r = 0; r < 4; ++r) pacc[it][r] = __builtin_nontemporal_load((const double *)(pn + r * (4 * WCOL * 8)));
            }
            {
                const char *const pa_ = (const char *)pnext + off_t;
#pragma unroll
                for (int r = 0; r < 4; ++r) al[it][r] = __builtin_nontemporal_load((const double *)(pa_ + r * (4 * WCOL * 8)));
            }
            // (no products to pace the no-stencil variant: without a fence the scheduler interleaves the four tiles' cells and spills)
            if (!FILTER) __builtin_amdgcn_sched_barrier(0);
            if (it == 0) BLC_STAMP2(5);
            // ---- advance the ring by one tile --------------------------------------------------------------------------------------
            if (FILTER && it + 1 < NTW) {
                // (circular ring: the new entries 4 it + NK .. + 3 take the slots of the four this tile's window began with; else the ring slides)
                constexpr bool CIRC = BLC_CIRC_RING != 0;
                if (!CIRC) {
#pragma unroll
                    for (int kb = 0; kb < NK - 4; ++kb) Bv[kb] = Bv[kb + 4];
                }
                const int e0 = CIRC ? 4 * it + NK : NK - 4;
                if (FASTEDGE && wv == NW - 1 && (it + 1) * TM + R0 + 12 >= NTW * TM) ring_fill_last_wave<N0, NTW * TM>(Bv, Xj, g, c, e0, 4, (it + 1) * TM + R0, CIRC ? NRM : 0);
                else if (!FASTEDGE && edge) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) Bv[CIRC ? (e0 + q) % NRM : e0 + q] = Xj[reflect1(i + TM + R0 + 4 * q + g, n0t) * WCOL + c];
                } else {
                    const double *s1 = Xj + (i + TM + R0 + g) * WCOL + c;
#pragma unroll
                    for (int q = 0; q < 4; ++q) Bv[CIRC ? (e0 + q) % NRM : e0 + q] = s1[q * 4 * WCOL];
                }
            }
        }

        BLC_STAMP2(6);
        // ---- row sums of the chain-step -> LDS (wave 5 adds them up after the next barrier) -------------------------------------------
        {
            double v[3] = {sN, sS, sC};
            double *rk = red + (j * 2 + (k & 1)) * (NW * 4 * 3);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                double x = v[q];
                x = blk::dpp_add<0x111, 0xf>(x);
                x = blk::dpp_add<0x112, 0xf>(x);
                x = blk::dpp_add<0x114, 0xf>(x);
                x = blk::dpp_add<0x118, 0xf>(x);
                if ((lane & 15) == 15) rk[(wv * 4 + (lane >> 4)) * 3 + q] = x;
            }
        }
        BLC_STAMP2(7);
        if (FILTER && k == 0 && last_chain) {        // the chains' bands replace the identity of the first step
            __syncthreads();
            {
                double *const Xp = X + pend_j * XSZ;
                const int l = fresh_lane(), g = l >> 4, c = l & 15;
#pragma unroll
                for (int it = 0; it < NTW; ++it)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Xp[(row0 + it * TM + g + 4 * r) * WCOL + c] = stt[it][r];
                totals(pend_j, pend_k);
                pend_j = -1;
            }
            for (int e = tid; e < 2 * NK * AST; e += NT) {
                const int jj = e / (NK * AST), q = e - jj * (NK * AST);
                const int a = AST == 16 ? band_distance16(q, R0) : band_distance(q, R0);
                As[e] = a == 0 ? (lw0[jj] > 0 ? P.taps[o0[jj]] : 1.0) : (a <= lw0[jj] ? P.taps[o0[jj] + a] : 0.0);
            }
        }
    }
    __syncthreads();
    if (pend_j >= 0) totals(pend_j, pend_k);
}

}  // namespace blc
