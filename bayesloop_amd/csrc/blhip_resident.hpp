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
//     axis-0 pass of the same step).  Point-to-point, no grid-wide barrier, no flags, no drains: THE DATA IS THE FLAG
//     (cdna_hip_programming.md Guideline 16, form R2).  A strip element is the 8-byte value itself with a ONE-BIT tag in its sign
//     bit (everything handed over is >= +0; the bit flips with every reuse of a slot, see tag_bit), written by one write-through
//     (sc1) store; clearing the bit gives the value back exactly.  (Round 3 began with 16-byte elements {low word, tag, high word,
//     tag}: twice the halo traffic, which is as much as the payload at 64 x 64 tiles.)  The
//     consumer requests its 8 elements with sc1 loads when its pass BEGINS, uses them chunks later and re-polls only the
//     elements whose tags are not there yet (bounded).  Edge segments walk TOWARDS the tile edge, so the neighbour's strip
//     is needed only for the last chunk of a pass; the hand-off costs one memory round trip that flies under the pass
//     (the flag form of round 2 cost the producer a drain of its stores' acknowledgements and the consumer two dependent
//     round trips: flag, then payload);
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
// HBM traffic per cell and step: the halo strips (2 R rows + 2 R columns of every tile, 8-byte tagged elements written and
// read once: 4 B per cell at 128 x 128 tiles, 8 B at 64 x 64) + what the fit keeps: nothing (evidence-only), the stored state
// 8 B (forward), stored state in + posterior out 16 B (backward).  bench.py reports these real bytes per kernel and, beside
// them, the rate the streaming formulation's 16 / 32 B per cell would need.
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
constexpr int NSLOT = 8;         // ring of granule slots for the lagged sums (>= 2 * max lag)
constexpr int MAXLAG = 4;
constexpr int DMAX = 4;          // data dimensions kept in registers
constexpr int NRED = 7;          // == blk::NRED: partial-sum slots per step

struct ResParams {
    int n0, n1, tr, tc, ntiles;
    int T, d, rec_len, lag;
    int store;                   // forward: write every step's state to post (full / forward-only fits)
    int means;                   // forward: also sum a * grid values (forward-only fits)
    int normalise;               // FORWARD-ONLY fits: normalise the stored filtered distributions in the kernel, `lag` steps behind (the
                                 // sum of a row is known `lag` steps later; the last `lag` rows are left to the host)
    const double *sfwd;          // BACKWARD: the forward pass's scales s_k (T doubles) and the sum of the last step's posterior: the
    double n_first;              //   posteriors are stored normalised -- their sums follow from scalars (see predicted_sum)
    const double *src0;          // what the first step consumes instead of a transition: prior (forward) / uniform (backward)
    double *post;                // [T][n0 * n1]: forward: stored states (out); backward: stored states (in) -> posteriors (out)
    const double *w0, *w1;       // R + 1 half-kernel weights per axis, zero-padded; {1, 0, ...} = no filter
    const double *m0, *m1, *colA, *colB, *rec;
    double step0;
    double *psum;                // [T][NRED][ntiles]
    double *cols;                // [2][ntiles][2][R][TR] tagged elements (8 bytes: the value, the tag in its sign bit): raw edge columns of the new state (parity = step & 1)
    double *rows;                // [2][ntiles][2][R][TC] tagged elements: axis-1-filtered edge rows
    unsigned cols_bytes, rows_bytes;
    unsigned long long *gran;    // [NSLOT][ntiles][4] {tag << 32 | half of a double}: the scale sum, the row sum (backward)
    unsigned *abort_word;
    unsigned long long timeout_ticks;   // wall_clock64 ticks (100 MHz)
    unsigned long long *prof;           // development builds (-DBLR_PROF): [16 steps][16 stamps] shader-clock stamps of one tile
    // (last: the fields above keep their offsets)
    const double *lik;                  // TAB kernels: the likelihood of every step, [T][n0 * n1] (observation models other than the Gaussian)
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
BLR_INL double nan_if_(bool n, double x) { return n ? std::nan("") : x; }
BLR_INL double ldu(const double *p, long long i) { return p[i]; }
BLR_INL int uni(int x) { return x; }
BLR_INL double ld_stream(const double *p) { return *p; }
BLR_INL void st_stream(double *p, double v) { *p = v; }
typedef unsigned long long Tq;                   // a tagged strip element: the value's bits, the tag in the sign bit
typedef char *Rsrc;
BLR_INL Rsrc strip_rsrc(double *p, unsigned) { return reinterpret_cast<char *>(p); }
BLR_INL void st_tq(Rsrc r, unsigned off, double v, unsigned bit) {
    unsigned long long b; std::memcpy(&b, &v, 8);
    assert((b >> 63) == 0 && "strip values are non-negative");
    b |= (unsigned long long)bit << 63;
    std::memcpy(r + off, &b, 8);
}
BLR_INL Tq ld_tq(Rsrc r, unsigned off) { Tq q; std::memcpy(&q, r + off, 8); return q; }
BLR_INL double tq_value(Tq q) { q &= ~(1ull << 63); double v; std::memcpy(&v, &q, 8); return v; }
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
// x, or NaN where `n` holds: one select on the high word (a NaN is a NaN whatever its low word holds; the 64-bit select is two v_cndmask)
BLR_INL double nan_if_(bool n, double x) { return __hiloint2double(n ? 0x7ff80000 : __double2hiint(x), __double2loint(x)); }
// wave-uniform read-only values (stencil weights, the step's data record) through the scalar cache: SGPRs, not VGPRs
BLR_INL double ldu(const double *p, long long i) { return ((const double __attribute__((address_space(4))) *)(unsigned long long)p)[i]; }
BLR_INL int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }       // a block-uniform value: SGPR
// the stored sequence is written / read once per pass (C3: 16 GiB): non-temporal hints keep it from sweeping the caches
BLR_INL double ld_stream(const double *p) { return __builtin_nontemporal_load(p); }
BLR_INL void st_stream(double *p, double v) { __builtin_nontemporal_store(v, p); }
// tagged strip elements through a buffer descriptor: ONE 8-byte write-through store / ONE 8-byte sc1 load each (aux 16 = sc1)
typedef unsigned long long Tq;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t Rsrc;
BLR_INL Rsrc strip_rsrc(double *p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)bytes, 0x00020000); }
BLR_INL void st_tq(Rsrc r, unsigned off, double v, unsigned bit) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const u32x2 q = {(unsigned)b, (unsigned)(b >> 32) | (bit << 31)};
    __builtin_amdgcn_raw_buffer_store_b64(q, r, (int)off, 0, 16);
}
BLR_INL Tq ld_tq(Rsrc r, unsigned off) {
    const u32x2 q = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 16);
    return (unsigned long long)q.x | ((unsigned long long)q.y << 32);
}
BLR_INL double tq_value(Tq q) { return __longlong_as_double((long long)(q & ~(1ull << 63))); }
#endif
// The tag: ONE bit, in the sign bit of the value.  Everything the tiles hand over -- a state, or a state filtered with positive weights -- is
// >= +0, so the sign bit is free, and clearing it gives the value back exactly.  A strip slot (step parity) is rewritten every second
// step and a tile is never more than a step ahead of its neighbour, so the bit only has to tell a step from the one two steps before:
// it starts at 1 with a slot's first use (the strips are zeroed before every launch) and flips with every reuse.
BLR_INL unsigned tag_bit(unsigned epoch, bool rows) {               // epoch = producing step + 1 (rows are first produced at step 1)
    const unsigned ks = epoch - 1u;
    return 1u - ((((rows ? ks - 1u : ks)) >> 1) & 1u);
}
// (a NaN counts as arrived whatever its sign: a degenerate fit -- zero normaliser, NaN state -- must not spin until the time-out; NaN
//  states stay NaN, so a stale one is as good as a fresh one, and the host rejects the pass by its sums anyway)
// every active lane of the wave (a scalar condition: the waits below branch on it as a wave, not lane by lane -- a loop some lanes leave
// early is a stack of lane masks in scalar registers, which the allocator takes from the values that live across the step)
#ifdef BLR_EMULATE
BLR_INL bool wave_all(bool x) { return x; }
#else
BLR_INL bool wave_all(bool x) { return __all(x ? 1 : 0) != 0; }
#endif
// tq_ok without a branch: 1 / 0
BLR_INL unsigned tq_ok_bits(Tq q, unsigned bit) {
    const unsigned hi = (unsigned)(q >> 32), lo = (unsigned)q, mag = hi & 0x7fffffffu;
    const unsigned isnan_ = (unsigned)(mag > 0x7ff00000u) | ((unsigned)(mag == 0x7ff00000u) & (unsigned)(lo != 0u));
    return ((hi >> 31) ^ bit ^ 1u) | isnan_;
}
BLR_INL bool tq_ok(Tq q, unsigned bit) { return (unsigned)(q >> 63) == bit || (q & 0x7fffffffffffffffull) > 0x7ff0000000000000ull; }

// a neighbour's strip: R tagged elements at byte offsets off0 + q * dstep (q = 0 .. R-1 in the consumer's walking order)
BLR_INL void strip_issue(Rsrc rs, int off0, int dstep, Tq (&fq)[R]) {
#pragma unroll
    for (int q = 0; q < R; ++q) fq[q] = ld_tq(rs, (unsigned)(off0 + q * dstep));
}
// -> f; elements whose tags are not `tag` yet are requested again (bounded); false = timed out / another block gave up
template <bool FAST>
BLR_INL bool strip_finish(const ResParams &P, Rsrc rs, int off0, int dstep, unsigned tag, Tq (&fq)[R], double (&f)[R]) {
    bool alive = true;
    // FAST: the eight tag bits at once first (AND / OR of the high words, one compare) -- per element the test is five vector
    // instructions, 40 per strip; only a strip that fails it is looked at element by element (a NaN counts as arrived, see tq_ok).
    // Measured, same box (profiles/r04_notes.md): 2048^2 forward step 9.77 - 9.97 -> 9.48 - 9.53 us; the one-chunk tiles lose with it
    // (C3 5.19 / 6.70 -> 5.26 / 6.95 us, same call; the cause was not looked for): multi-chunk shapes only
    auto all_there = [&]() {
        if (FAST) {
            unsigned hand = 0xffffffffu, hor = 0u;
#pragma unroll
            for (int q = 0; q < R; ++q) { const unsigned hi = (unsigned)(fq[q] >> 32); hand &= hi; hor |= hi; }
            if (wave_all(tag ? (hand >> 31) != 0u : (hor >> 31) == 0u)) return true;      // (a scalar branch)
        }
        // (branch-free: a short-circuit chain of eight tests is eight nested lane masks, whose scalar registers the allocator takes from
        //  the values that live across the step -- and reloads those in the walks)
        unsigned ok = 1u;
#pragma unroll
        for (int q = 0; q < R; ++q) ok &= tq_ok_bits(fq[q], tag);
        return ok != 0u;
    };
#ifdef BLR_EMULATE
    assert(all_there() && "hand-off protocol: consumed before published");
    (void)P; (void)rs; (void)off0; (void)dstep;
#else
    if (!wave_all(all_there())) {                  // (the wave waits as one: every lane requests its strip again until all have theirs)
        const unsigned long long t0 = now_ticks();
        for (unsigned spins = 1;; ++spins) {
            nap();
            strip_issue(rs, off0, dstep, fq);
            if (wave_all(all_there())) break;
            if ((spins & 255u) == 0u) {
                if (uni((int)ld_flag(P.abort_word)) != 0) { alive = false; break; }
                if (now_ticks() - t0 > P.timeout_ticks) { st_flag(P.abort_word, 1u); alive = false; break; }
            }
        }
    }
#endif
#pragma unroll
    for (int q = 0; q < R; ++q) f[q] = tq_value(fq[q]);
    return alive;
}

// block id -> tile: spatially adjacent tiles on the same XCD where the dispatcher deals blocks round-robin (b % 8); a pure
// performance hint (same-XCD hand-offs are ~1.7x faster), any placement is correct
BLR_INL int tile_of_block(int b, int ntiles) {
    return (ntiles % 8 == 0) ? (b % 8) * (ntiles / 8) + b / 8 : b;
}

// One thread's in-place pass over its SEG-element line segment, C outputs per chunk.  Positions p = 0 .. SEG-1 in WALKING order
// live at x0[p * STRIDE] (compile-time stride, negative = towards lower addresses: every access is base + immediate offset);
// nearv[k] = x(k - R); the far halo x(SEG + k) = f[k] was read before the barrier, or it is a neighbour tile's strip: far_issue()
// requests it when the pass begins, far_fetch(f) completes it at the top of the first chunk whose NEXT window reaches past the segment.  pre(p0) may start loads
// the epilogue of the chunk needs; emit(p0, v) receives the C filtered values of positions p0 .. p0+C-1 and may overwrite
// x(p0 .. p0+C-1): the register window (2 R + C values) already holds what later chunks need.
//   A one-chunk segment (SEG == C) needs its far halo for every output.  When that halo is a neighbour's strip (`far_is_strip`),
// the chunk is first evaluated with zeros in the far slots -- everything the thread can do before the hand-off -- and the
// (R + 1) R / 2 products with the far values are added once they have arrived.
template <int SEG, int STRIDE, int C, class FarIssue, class FarFn, class Pre, class Emit>
BLR_INL void walk(const double *x0, const double (&nearv)[R], double (&f)[R], bool far_is_strip, const double (&wk)[R + 1], FarIssue &&far_issue,
                  FarFn &&far_fetch, Pre &&pre, Emit &&emit) {
    static_assert(SEG % C == 0 && SEG >= C && SEG >= R, "segment = whole chunks, at least one radius long");
    constexpr int W = 2 * R + C;
    constexpr bool ONE = SEG == C;                   // single chunk: deferred far halo
    double w[W];
    bool have_far = false;                           // f: the far halo; far_fetch(f) completes it when it is a neighbour tile's strip
    far_issue();                                     // (the strip's elements are requested now and looked at chunks later)
#pragma unroll
    for (int k = 0; k < R; ++k) w[k] = nearv[k];
    if (R + C > SEG && !ONE) { far_fetch(f); have_far = true; }
#pragma unroll
    for (int q = 0; q < R + C; ++q) w[R + q] = q < SEG ? x0[q * STRIDE] : ((ONE && far_is_strip) ? 0.0 : f[q < SEG ? 0 : q - SEG]);
#pragma unroll
    for (int p0 = 0; p0 < SEG; p0 += C) {
        const bool more = p0 + C < SEG;
        double nx[C];
        if (more) {                                  // what enters the window for the next chunk: x(p0+C+R .. p0+2C+R-1)
            if (p0 + 2 * C + R > SEG && !have_far) { far_fetch(f); have_far = true; }
#pragma unroll
            for (int k = 0; k < C; ++k) {
                const int pos = p0 + C + R + k;
                nx[k] = pos < SEG ? x0[pos * STRIDE] : f[pos < SEG ? 0 : pos - SEG];
            }
        }
        pre(p0);                                     // (loads the epilogue of THIS chunk needs: they fly under the arithmetic below)
        double v[C];
#pragma unroll
        for (int j = 0; j < C; ++j) v[j] = w[R + j] * wk[0];
#pragma unroll
        for (int k = R; k >= 1; --k) {               // outermost pair inwards, the C outputs interleaved (independent chains)
            double t[C];
#pragma unroll
            for (int j = 0; j < C; ++j) t[j] = w[R + j - k] + w[R + j + k];
#pragma unroll
            for (int j = 0; j < C; ++j) v[j] = fma(t[j], wk[k], v[j]);
        }
        if (ONE && far_is_strip) {                   // far element m = x(SEG + m) reaches output j through tap R + m - j (m <= j)
            far_fetch(f);
#pragma unroll
            for (int m = 0; m < R; ++m) {
#pragma unroll
                for (int j = m; j < C; ++j) v[j] = fma(f[m], wk[R + m - j], v[j]);
            }
        }
        emit(p0, v);
        if (more) {
#pragma unroll
            for (int k = 0; k < 2 * R; ++k) w[k] = w[k + C];
#pragma unroll
            for (int k = 0; k < C; ++k) w[2 * R + k] = nx[k];
        }
    }
}

// The same pass for the many-threads shapes (SEG = 16: 1024 threads per 128 x 128 tile = 4 waves per SIMD, 128 registers per thread):
// the thread's WHOLE window -- near halo, its own SEG inputs, far halo -- is read before the barrier; the pass itself is
// arithmetic and stores only (no loads of the next chunk's inputs, no window shifts).  A far halo that is a neighbour tile's
// strip is fetched at the first chunk that reaches beyond the segment.
template <int SEG, int C, class FarIssue, class FarFn, class Pre, class Emit>
BLR_INL void walk_full(const double (&nearv)[R], const double (&own)[SEG], double (&f)[R], bool far_is_strip, const double (&wk)[R + 1],
                       FarIssue &&far_issue, FarFn &&far_fetch, Pre &&pre, Emit &&emit) {
    static_assert(SEG % C == 0 && SEG >= R, "segment = whole chunks, at least one radius long");
    constexpr int W = 2 * R + SEG;
    constexpr int FIRST_FAR = ((SEG - R) / C) * C;   // first chunk with an output p, p + R >= SEG
    double w[W];
    far_issue();
#pragma unroll
    for (int k = 0; k < R; ++k) w[k] = nearv[k];
#pragma unroll
    for (int q = 0; q < SEG; ++q) w[R + q] = own[q];
#pragma unroll
    for (int k = 0; k < R; ++k) w[R + SEG + k] = f[k];
#pragma unroll
    for (int p0 = 0; p0 < SEG; p0 += C) {
        if (p0 == FIRST_FAR && far_is_strip) {
            far_fetch(f);
#pragma unroll
            for (int k = 0; k < R; ++k) w[R + SEG + k] = f[k];
        }
        pre(p0);
        double v[C];
#pragma unroll
        for (int j = 0; j < C; ++j) v[j] = w[R + p0 + j] * wk[0];
#pragma unroll
        for (int k = R; k >= 1; --k) {
            double t[C];
#pragma unroll
            for (int j = 0; j < C; ++j) t[j] = w[R + p0 + j - k] + w[R + p0 + j + k];
#pragma unroll
            for (int j = 0; j < C; ++j) v[j] = fma(t[j], wk[k], v[j]);
        }
        emit(p0, v);
    }
}

// A block-uniform pointer the optimiser cannot see through: what is loaded through it is loaded where the code says, every time -- not
// once before the time loop, to live in scalar registers for the whole kernel (the two passes' weights: 2 x 18 of ~100; with more
// invariants than registers the allocator parks them in vector-register lanes and fetches them back with v_readlane inside the walks)
template <class T>
BLR_INL const T *launder_uniform(const T *p) {
#ifndef BLR_EMULATE
    unsigned long long v = (unsigned long long)p;
    asm volatile("" : "+s"(v));
    return (const T *)v;
#else
    return p;
#endif
}
// ... and a block-uniform integer: conditions derived from it (is there a neighbour tile on this side, is this data dimension in use) are
// evaluated where they are needed by the scalar unit -- hoisted out of the time loop each of them is a 64-bit lane mask the allocator keeps
BLR_INL int launder_uniform(int x) {
#ifndef BLR_EMULATE
    asm volatile("" : "+s"(x));
#endif
    return x;
}
BLR_INL int min_(int a, int b) { return a < b ? a : b; }

// keep the optimiser from hoisting a thread's (time-invariant) address arithmetic out of the time loop: hoisted, the
// addresses of every row / column a thread touches stay live across the whole step and spill (598 spilled VGPRs measured)
BLR_INL int launder(int x) {
#ifndef BLR_EMULATE
    asm volatile("" : "+v"(x));
#endif
    return x;
}

template <int TR_, int TC_, int SEG_, int CHK_, bool BWD_, int MODE_ = 0, bool PAD_ = false, bool TAB_ = false>
struct Res {
    static constexpr int TR = TR_, TC = TC_, SEG = SEG_, CHK = CHK_;    // CHK: outputs per chunk of a pass
    static constexpr bool BWD = BWD_;
    // TAB: the likelihood of a cell comes out of the (T, G) table (every observation model but the Gaussian: Laplace, AR1, ScaledAR1, a
    // caller's pdf) instead of the recurrence along the column -- 8 more bytes read per cell and step; one-chunk shapes only
    static constexpr bool TAB = TAB_;
    static_assert(!TAB || SEG_ == CHK_, "tabulated likelihood: the one-chunk tile shapes");
    // PAD: the grid does not fill its last tile row / column (n0 < tr * TR or n1 < tc * TC; at least R cells of padding where there is
    // any).  The kernel works on the tiles' geometry: cells outside the grid are kept at ZERO and out of every sum and store, except the
    // R rows / columns next to the grid's true edge, which hold the MIRROR image of the cells inside (rewritten after every step:
    // mirror_edges) -- so the plain stencil of the two passes IS the half-sample-reflecting one for every cell of the grid, and the
    // strips the last tiles hand to their neighbours carry the right values.  Row coordinates continue the lattice beyond the grid (the
    // likelihood recurrence runs through padded rows).
    static constexpr bool PAD = PAD_;
    // MODE 1 (EVID): forward pass of an evidence-only fit -- nothing is stored, no means, no rows to normalise: the flags of ResParams
    // are compile-time constants (fewer live values: the many-threads shape has 128 registers per thread).  MODE 2 (FULLFWD): forward pass
    // of a FULL fit -- every state stored, no means, no rows to normalise (the backward pass makes the posteriors).  MODE 3 (FWDONLY): forward-only fit -- stored,
    // means, rows normalised `lag` steps behind.  MODE 0: the flags are read from ResParams (padded grids; the backward kernels have none).
    // Measured (profiles/r04_notes.md): with the flags at run time the epilogue carries a scalar branch per cell and flag and the kernel
    // spills twice the scalars -- C3's forward pass 5.65 -> 5.21 us per step, the storing 2048^2 forward pass 13.1 -> 11.6 us with MODE 2.
    static constexpr int MODE = MODE_;
    static constexpr bool EVID = MODE_ == 1, FULLFWD = MODE_ == 2, FWDONLY = MODE_ == 3;
    static_assert(!(MODE_ != 0 && BWD), "the forward flavours");
    BLR_INL static bool f_store(const ResParams &Q) { return EVID ? false : ((FULLFWD || FWDONLY) ? true : Q.store != 0); }
    BLR_INL static bool f_means(const ResParams &Q) { return FWDONLY ? true : (MODE != 0 ? false : Q.means != 0); }
    BLR_INL static bool f_norm(const ResParams &Q) { return FWDONLY ? true : (MODE != 0 ? false : Q.normalise != 0); }
    BLR_INL static double *f_post(const ResParams &Q) { return EVID ? nullptr : Q.post; }
    static constexpr int P = TC + 1;                 // LDS pitch in doubles: odd => the row-strided accesses of the axis-1 pass
                                                     // and the contiguous ones of the axis-0 pass are both conflict-free
    static constexpr int NSH = TC / SEG, NSV = TR / SEG;
    static constexpr int NT = TR * NSH;
    static_assert(TR * NSH == TC * NSV, "both passes use every thread");
    static_assert(NSH >= 2 && NSV >= 2, "edge segments need an in-tile neighbour segment on their near side");
    // rows between exact re-anchorings of the likelihood recurrence: once per segment, at most 32 rows apart.  The recurrence's
    // relative error grows like n^2 / 2 ulp in the worst case (n rows since the anchor): 32 rows -> 6e-14, four decades inside the bar
    static constexpr int ANCHOR = SEG < 32 ? SEG : 32;
    static_assert(TR % SEG == 0 && TC % SEG == 0 && TR >= 2 * R && TC >= 2 * R && ANCHOR % CHK == 0 && SEG % ANCHOR == 0, "tile shape");
    static constexpr int NW = NT / 64;
    static constexpr bool ONE = SEG == CHK;          // one-chunk segments
    static_assert(NT % 64 == 0, "whole waves");
    static constexpr bool FULLW = NT > 512;          // many-threads shapes: whole window in registers before the barrier (walk_full)
    // a neighbour's strip is requested when the pass begins, 2+ chunks before it is used (32 registers per thread in flight; the
    // multi-chunk backward shape has none to spare and requests it where it needs it)
    static constexpr bool EARLY = ONE || !BWD;
    // a step's data record is requested a step AHEAD (under the last chunk of the previous axis-0 pass, see v_walk_d) -- except by
    // the multi-chunk backward kernel, which requests it when the step begins: with the record's registers alive across the whole
    // step that kernel went from 80 to 254 spilled VGPRs and its step from 25.5 to 44.7 us (2048^2 full fit)
    static constexpr bool REC_AHEAD = !(BWD && !ONE);
    // The multi-chunk backward kernel runs its axis-0 pass in TWO phases: the stencil alone (filtered values back into the tile, like the
    // axis-1 pass), then the epilogue over the thread's SEG cells.  The stored forward state the epilogue multiplies by is requested for
    // the whole segment BEFORE the stencil phase and consumed after it: a full pass (~2 us) for the loads to arrive, and no load that
    // queues up behind the previous chunk's posterior stores (one counter orders a wave's vector loads and stores: with a load per
    // chunk every wait for the state also waited for the stores of the chunk before -- 27 us per step at 2048^2, twice the forward pass).
    // The window registers and the epilogue's registers are never live together (the fused form spilled 70 VGPRs).
    static constexpr bool DEFER = BWD && !ONE && !FULLW;
    // The lagged global sum is gathered by HALF of the block's waves -- the half that reaches the barrier after the axis-1 pass early
    // (multi-chunk shapes: the edge segments, dealt to the first waves, have issue priority over their SIMD partners; one-chunk
    // shapes: the edge waves wait for their neighbours there, the others are early) -- in that slack, not by everybody after it.
    static constexpr int GW = NW >= 2 ? NW / 2 : 1;
    static constexpr bool GATHER_FIRST = !ONE;
    static constexpr int MAX_TILES = 256;                // (one tile per CU)
    static constexpr int GPL = (MAX_TILES / GW + 63) / 64;   // tiles per lane when GW waves share the tiles' partial sums
    static constexpr int NG = 1;                         // sums every tile publishes per step: the scale sum (forward: sum a = the row sum
                                                         // of the stored state; backward: sum c)
    static constexpr int LDS_TILE = TR * P;          // doubles
    static constexpr int LDS_M0 = LDS_TILE;          // TR row coordinates of the tile
    static constexpr int LDS_COL = LDS_M0 + TR;      // the tile's column constants: [TC] grid value, [TC] cA, [TC] cB
    static constexpr int LDS_MISC = LDS_COL + 3 * TC;    // [1] dead flag  [2] arrival counter of the gathering waves  [7] 1 / lagged sum  [8 .. 8+GW) their shares
    static_assert(NG == 1, "one lagged sum");
    static constexpr int LDS_RED = LDS_MISC + 8 + 2 * NW;    // reduction scratch of the step's sums  (misc[8 + g NW + w]: wave w's share of sum g)
    static constexpr int LDS_DOUBLES = LDS_RED + 5 * (NW + 1) + 8;

    struct Rec { double mE, mR, mq, iE, iR, iq; int nE, nR, nq; };
    struct ColC { double g1, cA, cB; };              // a column's grid value and likelihood constants (from LDS, per pass)

    // where a thread works in a pass, derived from its id whenever needed (a few integer operations) instead of being carried
    // in registers through both passes: far = kind of the far halo (0 neighbour segment in LDS, 1 mirror at the grid edge,
    // 2 the strip of tile nb, side)
    struct Geo { int line, seg, far, nb, side; };
    // A wave's 64 lanes share their segment when a line has a multiple of 64 positions (t / TR resp. t / TC is constant over the wave).
    // Handed to the compiler as a SCALAR (readfirstlane) it turns the far-halo cases and the walking direction into scalar branches;
    // as a per-lane value they were lane masks the allocator kept -- and spilled -- as 64-bit pairs (most of the ~600 v_readlane of the
    // step loop restore such masks).  Same box, A/B (profiles/r04_notes.md): 2048^2 forward step 10.35 - 10.48 -> 9.74 - 9.80 us.  Only in
    // the multi-chunk forward kernels: the one-chunk tiles got 4 % SLOWER with it (C3 5.68 / 6.73 -> 5.93 / 6.94 us) and the multi-chunk
    // backward kernel measured slower with it, before and after its two-phase axis-0 pass (21.7 - 22.1 vs 20.8 - 21.3 us).
    static constexpr bool WAVE_UNIFORM_SEG_H = TR % 64 == 0 && SEG != CHK && !BWD, WAVE_UNIFORM_SEG_V = TC % 64 == 0 && SEG != CHK && !BWD;

    struct Thread {
        int tid, tile, ti, tj, i0, j0, tr, tc;       // (tile .. tc: block-uniform)
        int rlim, clim;                              // rows / columns of this tile inside the grid (PAD; else TR / TC)
        double *lds;
        // registers that live across a barrier
        double nearv[R], farv[R];
        double own[FULLW ? SEG : 1];                 // (FULLW) the segment's own inputs, read before the barrier
        double xd[DMAX];                             // this step's data record (wave-uniform)
        double al8[BWD ? CHK : 1];                   // backward: the stored forward state of the chunk being processed
        double alS[DEFER ? SEG : 1];                 // (DEFER) ... of the thread's whole column segment, in flight across the stencil phase
        unsigned long long gq[GPL][2 * NG];          // this wave's share of the lagged sums' granules, in flight since the step began
        double nz8[BWD ? 1 : CHK];                   // forward-only: the row being normalised (the chunk's cells, `lag` steps back)
        double npred;                                // backward: the sum of this step's posterior, predicted from scalars
        // exp(second difference of the likelihood's exponent along the rows): depends on the thread's column and on the NUMBER of valid
        // data dimensions of the step only -- recomputed when that number changes (almost never)
        double mq_c, iq_c, dn_prev;
        int nq_c;
        double sums[5];
        bool dead;

        BLR_INL void init(const ResParams &Q, int block, int tid_, double *lds_) {
            tid = tid_; lds = lds_; dead = false;
            mq_c = 1.0; iq_c = 1.0; dn_prev = -1.0; nq_c = 0;
            tr = uni(Q.tr); tc = uni(Q.tc);
            tile = uni(tile_of_block(block, Q.ntiles));
            ti = uni(tile / Q.tc); tj = uni(tile - ti * Q.tc);
            i0 = ti * TR; j0 = tj * TC;
            rlim = PAD ? uni(min_(TR, Q.n0 - i0)) : TR; clim = PAD ? uni(min_(TC, Q.n1 - j0)) : TC;
#pragma unroll
            for (int k = 0; k < DMAX; ++k) xd[k] = nan_();
#pragma unroll
            for (int q = 0; q < 5; ++q) sums[q] = 0.0;
        }
        // Segments are dealt to the waves edge segments first: 0, last, 1, 2, ...  The edge segments wait for a neighbour tile;
        // on the first waves they have issue priority (oldest first) over the interior segment that shares their SIMD
        // (waves w and w + 4 of a 512-thread block), which then fills their stall.
        BLR_INL static int seg_of(int s, int nseg) { return s == 0 ? 0 : (s == 1 ? nseg - 1 : s - 1); }
        BLR_INL Geo hgeo() const {                   // axis-1 pass: a row and a segment of its columns
            const unsigned t = (unsigned)launder(tid);
            Geo g{(int)(t % TR), WAVE_UNIFORM_SEG_H ? uni(seg_of((int)(t / TR), NSH)) : seg_of((int)(t / TR), NSH), 0, tile, 0};
            if (g.seg == 0) { if (tj > 0) { g.far = 2; g.nb = tile - 1; g.side = 1; } else g.far = 1; }
            else if (g.seg == NSH - 1) { if (tj < tc - 1) { g.far = 2; g.nb = tile + 1; g.side = 0; } else g.far = 1; }
            return g;
        }
        BLR_INL Geo vgeo() const {                   // axis-0 pass: a column and a segment of its rows
            const unsigned t = (unsigned)launder(tid);
            Geo g{(int)(t % TC), WAVE_UNIFORM_SEG_V ? uni(seg_of((int)(t / TC), NSV)) : seg_of((int)(t / TC), NSV), 0, tile, 0};
            if (g.seg == 0) { if (ti > 0) { g.far = 2; g.nb = tile - tc; g.side = 1; } else g.far = 1; }
            else if (g.seg == NSV - 1) { if (ti < tr - 1) { g.far = 2; g.nb = tile + tc; g.side = 0; } else g.far = 1; }
            return g;
        }

        // time index of the k-th executed step
        BLR_INL static int time_of(const ResParams &Q, int k) { return BWD ? Q.T - 1 - k : k; }
        // segment s walks towards lower indices when it is the first one (its far side is then the tile's low edge)
        BLR_INL static constexpr int first_pos(int s, int dir) { return dir < 0 ? SEG - 1 : s * SEG; }

        // ---- the data record of step k (requested a step ahead, see v_walk_d) ---------------------------------------------------
        BLR_INL void begin_step(const ResParams &Q, int k) {
            const int t = time_of(Q, k);
            const int d = ONE ? launder_uniform(Q.d) : Q.d;
#pragma unroll
            for (int q = 0; q < DMAX; ++q) xd[q] = q < d ? ldu(Q.rec, (long long)t * Q.rec_len + q) : nan_();
        }
        BLR_INL double *row_ptr(const ResParams &Q, int k, int r0, int c) const {
            return f_post(Q) ? f_post(Q) + (long long)time_of(Q, k) * Q.n0 * Q.n1 + (long long)(i0 + r0) * Q.n1 + (j0 + c) : nullptr;
        }
        // forward-only: the row written `lag` steps ago (time t - lag), whose sum arrives with this step's lagged sums
        BLR_INL double *lagged_row_ptr(const ResParams &Q, int k, double *pt0) const {
            return (!BWD && f_norm(Q) && k >= Q.lag && pt0) ? pt0 - (long long)Q.lag * Q.n0 * Q.n1 : nullptr;
        }
        // backward: N_t = sum_cells alpha_t beta_t WITHOUT a reduction.  beta_t = T(c_{t+1}) s'_t and the reflect-boundary Gaussian
        // stencil T is self-adjoint, so N_t = s'_t sum T(alpha_t) c_{t+1}; the forward pass made alpha_{t+1} = T(alpha_t) s_{t+1} L_{t+1}
        // and c_{t+1} = beta_{t+1} L_{t+1}, hence  N_t = s'_t N_{t+1} / s_{t+1}  -- scalars every tile has (s' = this step's lagged
        // scale, s = the forward scales, N_{T-1} = sum alpha_{T-1} / G).  The posterior is stored normalised right away; the host
        // compares the prediction with the reduced sums (rounding-level agreement, ~2e-16 per step) and falls back if it ever differs.
        BLR_INL void predicted_sum(const ResParams &Q, int k, double scale) {
            if (BWD) npred = k == 0 ? Q.n_first : scale * npred / ldu(Q.sfwd, time_of(Q, k) + 1);
        }
        // ---- the lagged global sums: this wave's share of the tiles' partial sums of step ks (tiles wv * tpw + lane + 64 j) --------
        // The loads are issued when the step begins and consumed before the axis-0 pass: their latency hides under the axis-1 pass.
        // index of this wave among the gathering waves, or -1
        BLR_INL int gather_wave() const {
            const int wv = uni(tid >> 6);
            return GATHER_FIRST ? (wv < GW ? wv : -1) : (wv >= NW - GW ? wv - (NW - GW) : -1);
        }
        BLR_INL void gather_issue(const ResParams &Q, int ks) {
            const int wv = gather_wave(), lane = tid & 63, tpw = (Q.ntiles + GW - 1) / GW;
#pragma unroll
            for (int j = 0; j < GPL; ++j) {
                const int loc = lane + 64 * j, idx = wv * tpw + loc;
#pragma unroll
                for (int q = 0; q < 2 * NG; ++q) gq[j][q] = 0ull;
                if (loc < tpw && idx < Q.ntiles) {
                    const unsigned long long *g = Q.gran + ((long long)(ks % NSLOT) * Q.ntiles + idx) * 4;
#pragma unroll
                    for (int q = 0; q < 2 * NG; ++q) gq[j][q] = ld_u64(g + q);
                }
            }
        }
        // -> this lane's part of the sums (fixed order); a granule that has not arrived yet is polled (bounded)
        BLR_INL void gather_finish(const ResParams &Q, int ks, double (&acc)[NG]) {
            const int wv = gather_wave(), lane = tid & 63, tpw = (Q.ntiles + GW - 1) / GW;
            const unsigned long long want = (unsigned long long)(unsigned)(ks + 1);
#pragma unroll
            for (int g2 = 0; g2 < NG; ++g2) acc[g2] = 0.0;
#pragma unroll
            for (int j = 0; j < GPL; ++j) {
                const int loc = lane + 64 * j, idx = wv * tpw + loc;
                const bool mine = loc < tpw && idx < Q.ntiles;
                auto tags_ok = [&]() {               // (lanes without a tile of their own: nothing to wait for)
                    unsigned ok = 1u;
#pragma unroll
                    for (int q = 0; q < 2 * NG; ++q) ok &= (unsigned)((gq[j][q] >> 32) == want);
                    return !mine || ok != 0u;
                };
#ifdef BLR_EMULATE
                assert(tags_ok() && "lagged sum consumed before published");
#else
                if (!wave_all(tags_ok())) {          // (the wave waits as one, see wave_all)
                    const unsigned long long *g = Q.gran + ((long long)(ks % NSLOT) * Q.ntiles + (mine ? idx : 0)) * 4;
                    const unsigned long long t0 = now_ticks();
                    for (unsigned spins = 1;; ++spins) {
                        if (mine) {
#pragma unroll
                            for (int q = 0; q < 2 * NG; ++q) gq[j][q] = ld_u64(g + q);
                        }
                        if (wave_all(tags_ok())) break;
                        nap();
                        if ((spins & 255u) == 0u) {
                            if (uni((int)ld_flag(Q.abort_word)) != 0) { dead = true; break; }
                            if (now_ticks() - t0 > Q.timeout_ticks) { st_flag(Q.abort_word, 1u); dead = true; break; }
                        }
                    }
                }
#endif
                if (mine) {
#pragma unroll
                    for (int g2 = 0; g2 < NG; ++g2) {
                        const unsigned long long bits = (gq[j][2 * g2] & 0xffffffffull) | (gq[j][2 * g2 + 1] << 32);
                        double v;
#ifdef BLR_EMULATE
                        std::memcpy(&v, &bits, 8);
#else
                        v = __longlong_as_double((long long)bits);
#endif
                        acc[g2] += v;
                    }
                }
            }
        }
        // 1 / (sum g of step k - lag) from the waves' shares (LDS, after the barrier), or 1 while k < lag.  g = 0: the scale of the
        // step; g = NG - 1: the normaliser of the row written `lag` steps ago (forward: the same number)
        BLR_INL double lagged_inverse(const ResParams &Q, int k, int) const {
            return k < Q.lag ? 1.0 : lds[LDS_MISC + 7];      // (left there by the last gathering wave, before the barrier after the axis-1 pass)
        }
        // the gathering waves' shares (misc[8 + w], fixed order) -> the inverse everybody reads
        BLR_INL void combine_shares() {
            double ssum = 0.0;
#pragma unroll
            for (int w = 0; w < GW; ++w) ssum += lds[LDS_MISC + 8 + w];
            lds[LDS_MISC + 7] = 1.0 / ssum;
        }

        // ---- axis-1 pass ---------------------------------------------------------------------------------------------------------
        template <int DIR>
        BLR_INL void h_preread_d(const Geo &hg) {
            const double *x0 = lds + hg.line * P + first_pos(hg.seg, DIR);
#pragma unroll
            for (int k = 0; k < R; ++k) nearv[k] = x0[DIR * (k - R)];
            if (FULLW) {
#pragma unroll
                for (int q = 0; q < SEG; ++q) own[q] = x0[DIR * q];
            }
            if (hg.far == 0) {
#pragma unroll
                for (int k = 0; k < R; ++k) farv[k] = x0[DIR * (SEG + k)];
            } else if (hg.far == 1) {                  // grid edge: half-sample mirror = the segment's own last values
#pragma unroll
                for (int k = 0; k < R; ++k) farv[k] = x0[DIR * (SEG - 1 - k)];
            }
        }
        BLR_INL void h_preread() { const Geo hg = hgeo(); if (hg.seg == 0) h_preread_d<-1>(hg); else h_preread_d<1>(hg); }

        template <int DIR>
        BLR_INL void h_walk_d(const ResParams &Q, int k, const Geo &hg) {
            double *x0 = lds + hg.line * P + first_pos(hg.seg, DIR);
            // the neighbour's raw edge columns of step k - 1 (tag k): element [(k-1) & 1][nb][side][cc][row], cc towards the far side
            const Rsrc rs = strip_rsrc(Q.cols, Q.cols_bytes);
            const int e0 = (((((k - 1) & 1) * Q.ntiles + hg.nb) * 2 + hg.side) * R + (hg.side == 1 ? R - 1 : 0)) * TR + hg.line;
            const int dstep = (hg.side == 1 ? -TR : TR) * 8;
            Tq fq[R];
            auto far_issue = [&]() { if (EARLY && hg.far == 2) strip_issue(rs, e0 * 8, dstep, fq); };
            auto far_fetch = [&](double (&f)[R]) {
                if (hg.far == 2) {
                    if (!EARLY) strip_issue(rs, e0 * 8, dstep, fq);
                    if (!strip_finish<!ONE>(Q, rs, e0 * 8, dstep, tag_bit((unsigned)k, false), fq, f)) dead = true;
                }
            };
            auto emit8 = [&](int p0, const double (&v)[CHK]) {
#pragma unroll
                for (int j = 0; j < CHK; ++j) x0[DIR * (p0 + j)] = v[j];
            };
            // (weights as scalars.  In vector registers -- read from LDS per pass, to free ~18 SGPRs of the ~100 spilled ones -- the
            //  128 x 128 kernels ran out of VGPRs instead: 4 / 54 / 271 spilled; measured by compiling, not kept)
            double wk[R + 1];
            const double *const w1p = launder_uniform(Q.w1);
#pragma unroll
            for (int q = 0; q <= R; ++q) wk[q] = ldu(w1p, q);
            if constexpr (FULLW) walk_full<SEG, CHK>(nearv, own, farv, hg.far == 2, wk, far_issue, far_fetch, [](int) {}, emit8);
            else walk<SEG, DIR, CHK>(x0, nearv, farv, hg.far == 2, wk, far_issue, far_fetch, [](int) {}, emit8);
        }
        BLR_INL void h_walk(const ResParams &Q, int k) {
            const Geo hg = hgeo();
            if (hg.seg == 0) h_walk_d<-1>(Q, k, hg); else h_walk_d<1>(Q, k, hg);
        }

        // after the axis-1 pass (barrier): the tile's filtered edge rows -> strips of step k  [side][rr][col]; every thread takes
        // part, a wave stores 512 contiguous bytes per instruction (scattered 8-byte write-through stores out of the walks'
        // registers were measured: 2 - 3 x slower walks)
        // Straight-line code: the element <-> thread map is compile-time per iteration (NT divides R * edge length or the other way
        // round), unsigned arithmetic, the only branch is block-uniform (does that neighbour exist).
        template <int EDGE, class Src>
        BLR_INL void publish_strip(Rsrc rs, int base, unsigned tag, bool have0, bool have1, Src &&src) {
            constexpr int PER = 2 * R * EDGE / NT, HALF = PER / 2;
            static_assert(PER * NT == 2 * R * EDGE && HALF * 2 == PER && HALF >= 1, "whole iterations per side");
            const unsigned t = (unsigned)launder(tid);
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                if (side == 0 ? have0 : have1) {
                    double v[HALF];
#pragma unroll
                    for (int it = 0; it < HALF; ++it) {
                        const unsigned rem = (unsigned)(it * NT) + t;
                        v[it] = src(side, rem / EDGE, rem % EDGE);
                    }
#pragma unroll
                    for (int it = 0; it < HALF; ++it)
                        st_tq(rs, ((unsigned)(base + side * R * EDGE + it * NT) + t) * 8u, v[it], tag);
                }
            }
        }
        BLR_INL void publish_rows(const ResParams &Q, int k) {          // [side][rr][col]
            publish_strip<TC>(strip_rsrc(Q.rows, Q.rows_bytes), ((k & 1) * Q.ntiles + tile) * 2 * R * TC, tag_bit((unsigned)(k + 1), true), ti > 0, ti < tr - 1,
                              [&](int side, unsigned rr, unsigned col) { return lds[((side ? TR - R : 0) + rr) * P + col]; });
        }
        // after the axis-0 pass + epilogue (barrier): the new state's edge columns -> strips of step k  [side][cc][row]
        BLR_INL void publish_cols(const ResParams &Q, int k) {
            publish_strip<TR>(strip_rsrc(Q.cols, Q.cols_bytes), ((k & 1) * Q.ntiles + tile) * 2 * R * TR, tag_bit((unsigned)(k + 1), false), tj > 0, tj < tc - 1,
                              [&](int side, unsigned cc, unsigned row) { return lds[row * P + (side ? TC - R : 0) + cc]; });
        }

        // ---- axis-0 pass + epilogue ----------------------------------------------------------------------------------------------
        template <int DIR>
        BLR_INL void v_preread_d(const Geo &vg) {
            const double *x0 = lds + first_pos(vg.seg, DIR) * P + vg.line;
#pragma unroll
            for (int k = 0; k < R; ++k) nearv[k] = x0[DIR * (k - R) * P];
            if (FULLW) {
#pragma unroll
                for (int q = 0; q < SEG; ++q) own[q] = x0[DIR * q * P];
            }
            if (vg.far == 0) {
#pragma unroll
                for (int k = 0; k < R; ++k) farv[k] = x0[DIR * (SEG + k) * P];
            } else if (vg.far == 1) {
#pragma unroll
                for (int k = 0; k < R; ++k) farv[k] = x0[DIR * (SEG - 1 - k) * P];
            }
        }
        BLR_INL void v_preread() { const Geo vg = vgeo(); if (vg.seg == 0) v_preread_d<-1>(vg); else v_preread_d<1>(vg); }

        // (DEFER) the stored forward state of the thread's column segment, requested before the axis-0 stencil phase
        template <int DIR>
        BLR_INL void alpha_issue_d(const ResParams &Q, int k, const Geo &vg) {
            const int r0 = first_pos(vg.seg, DIR), c = vg.line;
            const double *pt0 = row_ptr(Q, k, r0, c);
            const bool colok = c < clim;
#pragma unroll
            for (int p = 0; p < SEG; ++p)
                alS[p] = (!PAD || (colok && r0 + DIR * p < rlim)) ? ld_stream(pt0 + (long long)(DIR * p) * Q.n1) : 0.0;
        }
        BLR_INL void alpha_issue(const ResParams &Q, int k) {
            if (DEFER) { const Geo vg = vgeo(); if (vg.seg == 0) alpha_issue_d<-1>(Q, k, vg); else alpha_issue_d<1>(Q, k, vg); }
        }

        // backward: the stored forward state alpha_t of positions p0 .. p0+7 (read before the posterior overwrites it in place)
        template <int DIR>
        BLR_INL void load_alpha8(const double *pt0, const double *ptn0, long long n1, int p0, int r0 = 0, bool colok = true) {
            if (BWD) {
#pragma unroll
                for (int j = 0; j < CHK; ++j) al8[j] = (!PAD || (colok && r0 + DIR * (p0 + j) < rlim)) ? ld_stream(pt0 + (long long)(DIR * (p0 + j)) * n1) : 0.0;
            }
            if (!BWD && ptn0) {                      // the row `lag` steps back, to be normalised in this step
#pragma unroll
                for (int j = 0; j < CHK; ++j) nz8[j] = (!PAD || (colok && r0 + DIR * (p0 + j) < rlim)) ? ptn0[(long long)(DIR * (p0 + j)) * n1] : 0.0;
            }
        }

        // the epilogue of positions p0 .. p0+7 of this thread's column segment: v = transition output (unscaled).
        // x0 / m0p / pt0 point at position 0 of the segment in the LDS tile / the tile's row coordinates / the global row
        template <int DIR>
        BLR_INL void epilogue8(const ResParams &Q, double *x0, const double *m0p, double *pt0, double *ptn0, double invn, int p0,
                               const double (&v)[CHK], double scale, Rec &rc, const ColC &cc, int r0 = 0, bool colok = true,
                               const double *lt0 = nullptr) {
            const double g1 = cc.g1, cA = cc.cA, cB = cc.cB;
            double lk8[TAB ? CHK : 1];           // (TAB) the table's values of the chunk's cells; lt0 = position 0 of the segment in the step's table
            if constexpr (TAB) {
#pragma unroll
                for (int j = 0; j < CHK; ++j)
                    lk8[j] = (!PAD || (colok && r0 + DIR * (p0 + j) < rlim)) ? lt0[(long long)(DIR * (p0 + j)) * Q.n1] : 0.0;
            }
            if (!BWD && ptn0) {
#pragma unroll
                for (int j = 0; j < CHK; ++j)
                    if (!PAD || (colok && r0 + DIR * (p0 + j) < rlim)) ptn0[(long long)(DIR * (p0 + j)) * Q.n1] = nz8[j] * invn;
            }
            if (!TAB && p0 % ANCHOR == 0) {
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
                blmath::exp_mn(a0, rc.mE, rc.nE);
                blmath::exp_mn(d1, rc.mR, rc.nR);
                if (dn != dn_prev) {                 // (wave-uniform: the data record is)
                    const double d2 = -2.0 * cA * dn * Q.step0 * Q.step0;
                    int tmp;
                    blmath::exp_mn(d2, mq_c, nq_c);
                    if (BWD) blmath::exp_mn(-d2, iq_c, tmp);
                    dn_prev = dn;
                }
                rc.mq = mq_c; rc.nq = nq_c; rc.iq = iq_c;
                if (BWD) {
                    rc.iE = blmath::inv_m(rc.mE); rc.iR = blmath::inv_m(rc.mR);      // (exp(-a0), exp(-d1): same exponents, reciprocal mantissas)
                } else {
                    rc.mE *= scale;                  // forward: the step's scale rides on the likelihood's mantissa (one product per cell less)
                }
            }
#pragma unroll
            for (int j = 0; j < CHK; ++j) {
                const int p = p0 + j;
                const double Lv = TAB ? (BWD ? lk8[TAB ? j : 0] : lk8[TAB ? j : 0] * scale) : ldexp_(rc.mE, rc.nE);
                const bool in = !PAD || (colok && r0 + DIR * p < rlim);          // (cells outside the grid stay zero, out of sums and stores)
                double keep;                                         // what becomes the tile's new state
                if (!BWD) {
                    const double a = in ? v[j] * Lv : 0.0;
                    keep = a;
                    if (f_store(Q) && in) st_stream(pt0 + (long long)(DIR * p) * Q.n1, a);
                    sums[0] += a;
                    if (f_means(Q)) { sums[3] = fma(a, m0p[DIR * p], sums[3]); sums[4] = fma(a, g1, sums[4]); }
                } else {
                    const double beta = in ? v[j] * scale : 0.0;
                    const double pp = al8[j] * beta;
                    const double cn = beta * Lv;
                    // p / L: reciprocal recurrence (no division, no intermediate overflow); 0/0 -> NaN (core.py:463)
                    const double pl = !in ? 0.0 : (TAB ? pp / Lv : nan_if_(Lv == 0.0, ldexp_(pp * rc.iE, -rc.nE)));      // (0 / 0 -> NaN either way)
                    keep = cn;
                    if (in) st_stream(pt0 + (long long)(DIR * p) * Q.n1, pp * invn);   // (invn = 1 / predicted sum: stored normalised)
                    sums[0] += pp; sums[1] += pl; sums[2] += cn;
                    sums[3] = fma(pp, m0p[DIR * p], sums[3]); sums[4] = fma(pp, g1, sums[4]);
                }
                x0[DIR * p * P] = keep;
                if constexpr (!TAB) {
                    rc.mE *= rc.mR; rc.nE += rc.nR;
                    rc.mR *= rc.mq; rc.nR += rc.nq;
                    if (BWD) { rc.iE *= rc.iR; rc.iR *= rc.iq; }
                }
            }
        }
        // (TAB) position 0 of the thread's segment in the table of step k
        BLR_INL const double *lik_ptr(const ResParams &Q, int k, int r0, int c) const {
            return TAB ? Q.lik + (long long)time_of(Q, k) * Q.n0 * Q.n1 + (long long)(i0 + r0) * Q.n1 + (j0 + c) : nullptr;
        }

        template <int DIR>
        BLR_INL void v_walk_d(const ResParams &Q, int k, const Geo &vg) {
            const double scale = lagged_inverse(Q, k, 0);
            Rec rc{1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0, 0, 0};
            const int r0 = first_pos(vg.seg, DIR), c = vg.line;
            const ColC cc{lds[LDS_COL + c], lds[LDS_COL + TC + c], lds[LDS_COL + 2 * TC + c]};
#pragma unroll
            for (int q = 0; q < 5; ++q) sums[q] = 0.0;
            double *x0 = lds + r0 * P + c;
            const double *m0p = lds + LDS_M0 + r0;
            double *pt0 = row_ptr(Q, k, r0, c);
            double *ptn0 = lagged_row_ptr(Q, k, pt0);
            const double *lt0 = lik_ptr(Q, k, r0, c);
            predicted_sum(Q, k, scale);
            const double invn = BWD ? 1.0 / npred : (ptn0 ? lagged_inverse(Q, k, 0) : 1.0);
            // the neighbour's axis-1-filtered edge rows of THIS step (tag k + 1): element [k & 1][nb][side][rr][col]
            const Rsrc rs = strip_rsrc(Q.rows, Q.rows_bytes);
            const int e0 = ((((k & 1) * Q.ntiles + vg.nb) * 2 + vg.side) * R + (vg.side == 1 ? R - 1 : 0)) * TC + vg.line;
            const int dstep = (vg.side == 1 ? -TC : TC) * 8;
            Tq fq[R];
            auto far_issue = [&]() { if (EARLY && vg.far == 2) strip_issue(rs, e0 * 8, dstep, fq); };
            auto far_fetch = [&](double (&f)[R]) {
                if (vg.far == 2) {
                    if (!EARLY) strip_issue(rs, e0 * 8, dstep, fq);
                    if (!strip_finish<!ONE>(Q, rs, e0 * 8, dstep, tag_bit((unsigned)(k + 1), true), fq, f)) dead = true;
                }
            };
            // (requested right before the chunk's arithmetic.  Requesting the one-chunk shapes' 16 HBM-missing loads per thread earlier --
            //  when the step begins, or after the axis-1 pass -- was measured: their issue alone holds a wave for ~2.8 k cycles wherever
            //  it is placed, and the step got slower, 24.4 k / 26.8 k vs 22.6 k cycles backward)
            // The NEXT step's data record (scalar loads; first needed by that step's first anchor) is requested under the LAST chunk's
            // arithmetic of this pass -- no LDS read waits follow it there, so its latency is never waited for.  (Requested when a step
            // begins, every wave sat out a scalar-cache miss before its first LDS read: ~0.5 k cycles per step.)
            constexpr bool REC_IN_PASS = !ONE && !FULLW;
            static_assert(!REC_IN_PASS || (SEG - CHK) % ANCHOR != 0, "the last chunk has no anchor (its epilogue does not read the record)");
            auto pre8 = [&](int p0) {
                if (!DEFER) load_alpha8<DIR>(pt0, ptn0, Q.n1, p0, r0, c < clim);
                if (REC_AHEAD && REC_IN_PASS && p0 == SEG - CHK && k + 1 < Q.T) begin_step(Q, k + 1);
            };
            auto emit8 = [&](int p0, const double (&v)[CHK]) {
                if constexpr (DEFER) {               // stencil phase: the filtered values back into the tile
#pragma unroll
                    for (int j = 0; j < CHK; ++j) x0[DIR * (p0 + j) * P] = v[j];
                } else {
                    epilogue8<DIR>(Q, x0, m0p, pt0, ptn0, invn, p0, v, scale, rc, cc, r0, c < clim, lt0);
                }
            };
            double wk[R + 1];
            const double *const w0p = launder_uniform(Q.w0);
#pragma unroll
            for (int q = 0; q <= R; ++q) wk[q] = ldu(w0p, q);
            if constexpr (FULLW) walk_full<SEG, CHK>(nearv, own, farv, vg.far == 2, wk, far_issue, far_fetch, pre8, emit8);
            else walk<SEG, DIR * P, CHK>(x0, nearv, farv, vg.far == 2, wk, far_issue, far_fetch, pre8, emit8);
            if constexpr (DEFER) {                   // epilogue phase: the thread's own cells (nobody else reads them before the step's last barrier)
#pragma unroll
                for (int p0 = 0; p0 < SEG; p0 += CHK) {
                    double v[CHK];
#pragma unroll
                    for (int j = 0; j < CHK; ++j) { v[j] = x0[DIR * (p0 + j) * P]; al8[j] = alS[p0 + j]; }
                    epilogue8<DIR>(Q, x0, m0p, pt0, ptn0, invn, p0, v, scale, rc, cc, r0, c < clim, lt0);
                }
            }
            if (REC_AHEAD && !REC_IN_PASS && k + 1 < Q.T) begin_step(Q, k + 1);
        }
        BLR_INL void v_walk(const ResParams &Q, int k) {
            const Geo vg = vgeo();
            if (vg.seg == 0) v_walk_d<-1>(Q, k, vg); else v_walk_d<1>(Q, k, vg);
        }

        // the first executed step has no transition: its input is src0 (prior / uniform), scale 1
        template <int DIR>
        BLR_INL void first_step_d(const ResParams &Q, const Geo &vg) {
            predicted_sum(Q, 0, 1.0);
            Rec rc{1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0, 0, 0};
            const int r0 = first_pos(vg.seg, DIR), c = vg.line;
            const ColC cc{lds[LDS_COL + c], lds[LDS_COL + TC + c], lds[LDS_COL + 2 * TC + c]};
#pragma unroll
            for (int q = 0; q < 5; ++q) sums[q] = 0.0;
            double *x0 = lds + r0 * P + c;
            const double *m0p = lds + LDS_M0 + r0;
            const long long g0 = (long long)(i0 + r0) * Q.n1 + (j0 + c);
            double *pt0 = f_post(Q) ? f_post(Q) + (long long)time_of(Q, 0) * Q.n0 * Q.n1 + g0 : nullptr;
            const double *s = Q.src0 + g0;
            const double *lt0 = TAB ? Q.lik + (long long)time_of(Q, 0) * Q.n0 * Q.n1 + g0 : nullptr;
#pragma unroll 1
            for (int p0 = 0; p0 < SEG; p0 += CHK) {
                double v[CHK];
#pragma unroll
                for (int j = 0; j < CHK; ++j) v[j] = (!PAD || (c < clim && r0 + DIR * (p0 + j) < rlim)) ? s[(long long)(DIR * (p0 + j)) * Q.n1] : 0.0;
                load_alpha8<DIR>(pt0, nullptr, Q.n1, p0, r0, c < clim);
                epilogue8<DIR>(Q, x0, m0p, pt0, nullptr, BWD ? 1.0 / Q.n_first : 1.0, p0, v, 1.0, rc, cc, r0, c < clim, lt0);
            }
            if (REC_AHEAD && Q.T > 1) begin_step(Q, 1);
        }
        // PAD, after the step's last barrier: the R rows / columns beyond the grid's true edge take the mirror image of the new state (half-
        // sample reflection: cell e + 1 + j <- cell e - j; the corner from both).  Sources are cells of the grid only, so one phase.
        BLR_INL void mirror_edges() {
            if (!PAD) return;
            const unsigned t = (unsigned)launder(tid);
            if (clim < TC) {                         // columns clim .. clim + R - 1 of rows 0 .. min(TR, rlim + R) - 1
                const int nr = min_(TR, rlim + R);
                for (unsigned idx = t; idx < (unsigned)(nr * R); idx += NT) {
                    const int r = (int)(idx / R), j = (int)(idx % R);
                    const int rs = r < rlim ? r : 2 * rlim - 1 - r;
                    if (clim + j < TC && clim - 1 - j >= 0 && rs >= 0) lds[r * P + clim + j] = lds[rs * P + clim - 1 - j];
                }
            }
            if (rlim < TR) {                         // rows rlim .. rlim + R - 1 of the columns inside the grid
                for (unsigned idx = t; idx < (unsigned)(clim * R); idx += NT) {
                    const int c = (int)(idx / R), j = (int)(idx % R);
                    if (rlim + j < TR && rlim - 1 - j >= 0) lds[(rlim + j) * P + c] = lds[(rlim - 1 - j) * P + c];
                }
            }
        }
        BLR_INL void first_step(const ResParams &Q) { const Geo vg = vgeo(); if (vg.seg == 0) first_step_d<-1>(Q, vg); else first_step_d<1>(Q, vg); }
    };

    // ---- the lagged global sums: one tagged granule pair per tile, step and sum -----------------------------------------------------
    BLR_INL static void publish_sum(const ResParams &Q, int tile, int k, int which, double value) {
        unsigned long long bits;
#ifdef BLR_EMULATE
        std::memcpy(&bits, &value, 8);
#else
        bits = (unsigned long long)__double_as_longlong(value);
#endif
        const unsigned long long tag = (unsigned long long)(unsigned)(k + 1) << 32;
        unsigned long long *g = Q.gran + ((long long)(k % NSLOT) * Q.ntiles + tile) * 4 + 2 * which;
        st_u64(g, tag | (bits & 0xffffffffull));
        st_u64(g + 1, tag | (bits >> 32));
    }
};

}  // namespace blr

#ifndef BLR_EMULATE
#include "blhip_kernels.hpp"

namespace blr {

// block barrier that orders LDS accesses only: the write-through strip stores (and the stored sequence's stores / loads) stay in
// flight across it.  (hipcc 7.2 compiles __syncthreads() to the same `s_waitcnt lgkmcnt(0); s_barrier` in this kernel -- checked in the
// ISA -- but a fence over all address spaces is allowed to wait for vmcnt(0); the explicit form does not depend on that.  What did
// wait for the stores' acknowledgements in round 2 was the drain in front of the epoch flags, gone with the flags.)
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// The kernel arguments arrive as 16-register tuples (s_load_dwordx16) and the register allocator spills and restores a tuple as ONE unit:
// with ~100 live scalars in the step loop a single field of the argument struct costs a 16-lane v_readlane burst per use (measured on the
// two-chain fold kernel, blhip_chainres.hpp: 304 -> 137 v_readlane per chain-step).  own_sgpr() copies a value into a scalar of its own --
// the empty asm makes the copy a separate value the allocator can place, spill or rematerialise by itself; a pointer is copied as an
// integer and handed back through the GLOBAL address space (behind the asm the compiler no longer sees that it came from a kernel
// argument and would fall back to flat loads / stores).
template <class T>
__device__ __forceinline__ T own_sgpr(T v) { asm volatile("" : "+s"(v)); return v; }
__device__ __forceinline__ double own_sgpr(double v) {
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    asm volatile("" : "+s"(b));
    return __longlong_as_double((long long)b);
}
template <class T>
__device__ __forceinline__ T *own_sgpr(T *p) {
    unsigned long long v = (unsigned long long)p;
    asm volatile("" : "+s"(v));
    return (T *)(T __attribute__((address_space(1))) *)v;
}
// (Applied to this file's own kernel it changes nothing -- 633 -> 612 v_readlane per step loop: what spills here are lane masks of the
//  control flow, not argument tuples; measured on the ISA, not kept.)
// The kernel's arguments, read AGAIN from the kernel-argument segment through a pointer the optimiser cannot see through.  Called at the
// top of every step: nothing of the ~50 dwords is loop-invariant then, nothing has to be kept across the loop -- which the allocator does
// by parking it in vector-register lanes and fetching it back with v_readlane (a vector-ALU slot each) inside the walks.  Four scalar loads
// per step instead.  Static v_readlane of the step loop / vector instructions: 128 x 128 forward 323 -> 123 / 4754 -> 4537 (applied), 128 x 128
// backward 700 -> 200 (no measurable gain), 64 x 64 forward 82 -> 35 (slower), 64 x 64 backward 190 -> 433.
template <bool RELOAD, class ARGS>
BLR_INL ARGS step_args(const ARGS &a) {         // (ARGS: the kernel's ONE by-value argument, at offset 0 of the segment)
#ifndef BLR_EMULATE
    if constexpr (RELOAD) {
        static_assert(sizeof(ARGS) % 4 == 0, "copied as dwords");
        typedef const unsigned __attribute__((address_space(4))) *ka_t;      // (the constant address space: scalar loads)
        unsigned long long kav = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kav));
        const ka_t src = (ka_t)kav;
        ARGS r;
        unsigned *const dst = reinterpret_cast<unsigned *>(&r);
#pragma unroll
        for (unsigned i = 0; i < sizeof(ARGS) / 4; ++i) dst[i] = src[i];
        return r;
    }
#endif
    return a;
}

template <int TR, int TC, int SEG, int CHK, bool BWD, int MODE = 0, bool PAD = false, bool TAB = false>
__global__ __launch_bounds__(TR *TC / SEG) void resident_kernel(const ResParams Qarg) {
    using K = Res<TR, TC, SEG, CHK, BWD, MODE, PAD, TAB>;
    constexpr bool RELOAD_ARGS = !K::ONE && !BWD;     // (measured, same box: 2048^2 forward step 8.89 -> 8.78 us; the one-chunk forward kernels LOSE 2.6 % with it -- C3 4.60 -> 4.72 us -- although their static reloads drop 82 -> 35: four scalar loads + a wait at the top of a 4.6-us step)
    const ResParams &Q = Qarg;                        // (set-up and the lambdas below; the step loop has its own, see step_args)
    constexpr int NT = K::NT, NW = K::NW;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *misc = lds + K::LDS_MISC;
    double *red = lds + K::LDS_RED;
    const int tid = threadIdx.x;
    typename K::Thread th;
    th.init(Q, blockIdx.x, tid, lds);
    // (PAD: row coordinates continue the lattice beyond the grid -- the likelihood recurrence of a column runs through its padded
    //  rows; padded columns take the last column's constants: their cells are masked)
    for (int e = tid; e < TR; e += NT)
        lds[K::LDS_M0 + e] = (!PAD || th.i0 + e < Q.n0) ? Q.m0[th.i0 + e] : Q.m0[Q.n0 - 1] + (double)(th.i0 + e - (Q.n0 - 1)) * Q.step0;
    for (int e = tid; e < TC; e += NT) {
        const int c = PAD ? min_(th.j0 + e, Q.n1 - 1) : th.j0 + e;
        lds[K::LDS_COL + e] = Q.m1[c]; lds[K::LDS_COL + TC + e] = Q.colA[c]; lds[K::LDS_COL + 2 * TC + e] = Q.colB[c];
    }
    if (PAD) for (int e = tid; e < K::LDS_TILE; e += NT) lds[e] = 0.0;       // (cells no pass writes before it reads them)
    if (tid == 0) { misc[1] = 0.0; misc[2] = 0.0; }
    const int gw = th.gather_wave();
    __syncthreads();

#ifdef BLR_PROF
    // development builds: shader-clock stamps of four waves of one interior tile (two edge-segment waves, two interior ones)
    const int prof_slot = tid == 0 ? 0 : (tid == 128 ? 1 : (tid == 256 ? 2 : (tid == NT - 64 ? 3 : -1)));
    const bool prof_me = Q.prof && th.tile == Q.ntiles / 2 + Q.tc / 2 && prof_slot >= 0;
#define BLR_STAMP(i) do { if (prof_me && k >= 8 && k < 24) Q.prof[prof_slot * 256 + (k - 8) * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define BLR_STAMP(i) do { } while (0)
#endif
    // Bookkeeping of a finished step (one lane): the waves' sums (LDS, complete since the step's last barrier, not overwritten
    // before the next step's) added in a fixed order -> the host's partial sums, and the tile's lagged sum for the other tiles.
    // It runs half a step LATER, in a wave that is early at the barrier after the axis-1 pass (multi-chunk shapes: wave 0, an edge
    // segment with issue priority over its SIMD partner; one-chunk shapes: an interior segment that neither waits for a neighbour nor
    // gathers the lagged sum).
    constexpr int BOOK_WAVE = K::ONE ? (NW >= 2 ? NW / 2 - 1 : 0) : 0;
    const bool booker = K::WAVE_UNIFORM_SEG_H ? uni(tid >> 6) == BOOK_WAVE : (tid >> 6) == BOOK_WAVE;
    auto book = [&](int kb) {                         // (the whole wave: lane w fetches wave w's sums, one tree adds them)
        constexpr int NV = BWD ? 5 : 3;
        const int lane = tid & 63;
        double tot[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            tot[q] = 0.0;
            if (!BWD && q > 0 && !K::f_means(Q)) continue;
            tot[q] = blk::wave_sum(lane < NW ? red[lane * NV + q] : 0.0);
        }
        if (lane == 0) {
            double *out = Q.psum + (long long)K::Thread::time_of(Q, kb) * NRED * Q.ntiles + th.tile;
            if (BWD) {
#pragma unroll
                for (int q = 0; q < 5; ++q) out[(long long)q * Q.ntiles] = tot[q];
                K::publish_sum(Q, th.tile, kb, 0, tot[2]);
            } else {
                out[0] = tot[0];
                if (K::f_means(Q)) { out[3LL * Q.ntiles] = tot[1]; out[4LL * Q.ntiles] = tot[2]; }
                K::publish_sum(Q, th.tile, kb, 0, tot[0]);
            }
        }
    };
    const int n_steps = Q.T;
    for (int k = 0; k < n_steps; ++k) {
        const ResParams Qstep = step_args<RELOAD_ARGS>(Qarg);
        const ResParams &Q = Qstep;                   // (hides the outer one inside the step)
        BLR_STAMP(0);
        // (see launder_uniform; the one-chunk tiles only -- static v_readlane of the step loop, 64 x 64 forward 144 -> 82, C3's forward step
        //  4.73 -> 4.65 us; the 128 x 128 forward kernel answers with MORE of them: 323 -> 490, 1.90 -> 1.95 ms per 200 steps)
        if constexpr (K::ONE) { th.ti = launder_uniform(th.ti); th.tj = launder_uniform(th.tj); th.tile = launder_uniform(th.tile); }
        if (!K::REC_AHEAD && k > 0) th.begin_step(Q, k);
        if (k == 0) {
            th.begin_step(Q, k);
            th.first_step(Q);
        } else {
            th.h_preread();
            BLR_STAMP(1);
            if (k >= Q.lag && gw >= 0) th.gather_issue(Q, k - Q.lag);
            BLR_STAMP(2);
            lds_barrier();
            BLR_STAMP(3);
            th.h_walk(Q, k);
            th.alpha_issue(Q, k);                     // (requested BEFORE the axis-1 pass instead: 21.9 - 22.2 vs 20.8 - 21.3 us, profiles/r04_notes.md)
            BLR_STAMP(4);
            if (booker) book(k - 1);
            BLR_STAMP(5);
            if (k >= Q.lag && gw >= 0) {              // (in the early waves' slack before the barrier)
                double part[K::NG];
                th.gather_finish(Q, k - Q.lag, part);
                const double ws = blk::wave_sum(part[0]);
                if ((tid & 63) == 0) {
                    misc[8 + gw] = ws;
                    unsigned *cnt = reinterpret_cast<unsigned *>(misc + 2);
                    const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (old == (unsigned)K::GW - 1u) { *cnt = 0u; th.combine_shares(); }       // the last one adds the shares (fixed order)
                }
            }
            BLR_STAMP(6);
            lds_barrier();
            BLR_STAMP(7);
            th.publish_rows(Q, k);
            BLR_STAMP(8);
            th.v_preread();
            BLR_STAMP(9);
            lds_barrier();
            BLR_STAMP(10);
            th.v_walk(Q, k);
            BLR_STAMP(11);
        }
        // ---- sums of the step: partials for the host, the lagged sums for the other tiles.  ONE barrier: every wave leaves its
        //      sums in LDS, then all threads copy the edge columns out while thread 0 adds the waves' sums (fixed order) ----------
        if (th.dead) misc[1] = 1.0;
        {
            constexpr int NV = BWD ? 5 : 3;
            double v[NV];
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q] = th.sums[BWD ? q : (q == 0 ? 0 : q + 2)];      // forward: N, M0, M1
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                if (!BWD && q > 0 && !K::f_means(Q)) break;
                const double ws = blk::wave_sum(v[q]);
                if ((tid & 63) == 0) red[(tid >> 6) * NV + q] = ws;
            }
            BLR_STAMP(12);
            lds_barrier();                            // the tile's new state is complete in LDS, the waves' sums and misc[1] are final
            BLR_STAMP(13);
            if (PAD && (th.rlim < TR || th.clim < TC)) {        // (block-uniform) the mirror image beyond the grid's edge, before anyone reads the tile again
                th.mirror_edges();
                lds_barrier();
            }
            th.publish_cols(Q, k);
            BLR_STAMP(14);
        }
        if (misc[1] != 0.0) return;                   // a wait timed out somewhere in this block: uniform exit (host falls back)
    }
    if (booker) book(Q.T - 1);
}


#ifndef BLR_EMULATE
// ---- is the chip ours?  The resident kernels need EVERY block of a launch on the chip at the same time: one block of 512 threads
// with up to 256 registers and up to 160 KB of LDS per CU.  On a GPU another process is using (or a partitioned one) some never
// arrive; a launch then sits out its whole time-out before the fit falls back (>= 0.25 s -- a 1000 x cliff for a millisecond fit).
// This probe asks the question in microseconds: a grid of one block per CU with the resident kernels' footprint (512 threads, the whole
// register file -- the empty asm touches v255 --, `lds` bytes of dynamic LDS); lane 0 of every block counts itself in and waits until
// all have, at most `timeout_ticks`.  out[0] = blocks that arrived, out[1] = 1 if a block gave up, out[8 + b] = 1 + the XCD block b ran
// on (HW_REG_XCC_ID): the both-axes kernels' plain-store exchange relies on blocks b, b + 8, b + 16, .. sharing an XCD.
template <int FOOTPRINT = 0>          // (a template: this header is part of every translation unit of the library, the kernel of the one that launches it)
__global__ __launch_bounds__(512, 1) void residency_probe_kernel(unsigned *out, unsigned nblocks, unsigned long long timeout_ticks) {
    extern __shared__ __attribute__((aligned(16))) double lds_probe[];
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    if (threadIdx.x == 0) {
        lds_probe[0] = 0.0;
        const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu;          // HW_REG_XCC_ID, 4 bits
        out[8 + blockIdx.x] = xcc + 1u;                       // (the host compares the blocks of a residue class: the ids need not BE the residues)
        __hip_atomic_fetch_add((gu32 *)(unsigned long long)&out[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = now_ticks();
        while (ld_flag(&out[0]) < nblocks) {
            if (ld_flag(&out[1]) != 0u) break;
            if (now_ticks() - t0 > timeout_ticks) { st_flag(&out[1], 1u); break; }
            nap();
        }
    }
}
#endif

}  // namespace blr
#endif
