// 1-D grids, ONE launch per pass: the K-steps-per-launch scheme of blhip_fused1d.hpp with the launch boundary replaced by a
// point-to-point hand-off inside a persistent kernel (BASELINE C2: 4096 cells, T = 10 000 -- 1250 launches of ~15 us per pass).
// Measured: 42.1 -> 36.2 ms per C2 fit (1.86 / 2.07 -> 1.64 / 1.75 us per step); what goes is the launch latency and the dependent
// loads of a launch's prologue, the ~1.4 us of a step itself (a latency chain with one cell per thread) stay.
//
// A "superstep" = what one launch of bl1f::fused1d_kernel does: a block owns TJ cells, loads them plus a halo of K * LW cells per side,
// advances K steps in LDS (the halo is recomputed redundantly: the valid range shrinks by LW per step), stores the owned cells of every
// step and the per-step partial sums.  Between supersteps a block needs
//   * the final state of the cells in its halo (its 1 - 3 neighbours per side), and
//   * the sum of that state over the WHOLE grid (lazy normalisation: the next superstep starts from state / sum) = nblk partial sums.
// Both travel as data-tagged granules (two 8-byte words per double, {tag << 32 | half}, tag = producing superstep + 1; write-through
// stores, sc1 loads: no flag, no fence, no barrier -- as the chain-resident kernels' sums), double-buffered by superstep parity: a block
// that publishes superstep s + 2 has gathered EVERY block's sum of superstep s + 1, which they published after reading superstep s.
// Results are bit-identical to the launch-per-K path (same arithmetic in the same order, the same raw sums for the host's bookkeeping).
// Bounded spins + abort word as in blhip_resident.hpp; the host falls back to the launch-per-K path when a launch gives up, and uses
// this kernel only when all blocks of a launch fit on the chip at once (nblk x chains <= CUs).
#pragma once
#include "blhip_fused1d.hpp"
#include "blhip_resident.hpp"

namespace bl1p {

using blk::NRED;
using blk::SRC_PREV;
constexpr int NT = bl1f::NT;

struct P1Params {
    int n, TJ, nblk, LW, K, dir, T, B, d, rec_len, store, means;
    const double *shared[5];
    const double *src0; long long src0_stride;     // what the FIRST step consumes if its source kind is SRC_PREV (a resumed / carried state)
    double *dst; long long dst_stride;             // state after the last step (or nullptr)
    double *post; long long post_stride;           // (B, T, n) stored rows or nullptr (evidence only)
    const unsigned char *srckind;                  // [T][B]
    const int *tap;                                // [T][B] tap-set id, -1 = identity
    const double *taps; const int *tap_off; const int *tap_lw;
    const double *wtab; const int *lwtab;          // [T][B][LW + 1] / [T][B]: the steps' weights and radii spelled out (build_wtab_kernel), or nullptr:
                                                   //   one load per staged element instead of the chain tap -> tap_lw / tap_off -> taps
    int prev_slot;                                 // the sum a superstep's successor normalises by (0: forward, 2: backward)
    double *psum;                                  // [T][B][NRED][nblk]
    const double *m1, *colA, *rec, *lik;
    unsigned long long *xch;                       // [2][B][n][2]    tagged halves of the state after a superstep
    unsigned long long *gran;                      // [2][B][nblk][2] tagged halves of a block's partial sum of the superstep's last step
    unsigned *abort_word;
    unsigned long long timeout_ticks;
};

// one tagged double (q0, q1: a first attempt requested earlier): -> value; false = timed out / another block gave up
__device__ __forceinline__ bool finish_tagged(const P1Params &P, const unsigned long long *g, unsigned long long want,
                                              unsigned long long q0, unsigned long long q1, double &v) {
    bool alive = true;
    // (the wave waits as one: blr::wave_all)
    auto there = [&]() { return ((unsigned)((q0 >> 32) == want) & (unsigned)((q1 >> 32) == want)) != 0u; };
    if (!blr::wave_all(there())) {
        const unsigned long long t0 = blr::now_ticks();
        for (unsigned spins = 1;; ++spins) {
            blr::nap();
            q0 = blr::ld_u64(g); q1 = blr::ld_u64(g + 1);
            if (blr::wave_all(there())) break;
            if ((spins & 255u) == 0u) {
                if (blr::uni((int)blr::ld_flag(P.abort_word)) != 0) { alive = false; break; }
                if (blr::now_ticks() - t0 > P.timeout_ticks) { blr::st_flag(P.abort_word, 1u); alive = false; break; }
            }
        }
    }
    v = __longlong_as_double((long long)((q0 & 0xffffffffull) | (q1 << 32)));
    return alive;
}
__device__ __forceinline__ bool fetch_tagged(const P1Params &P, const unsigned long long *g, unsigned long long want, double &v) {
    const unsigned long long q0 = blr::ld_u64(g), q1 = blr::ld_u64(g + 1);
    return finish_tagged(P, g, want, q0, q1, v);
}
__device__ __forceinline__ void publish_tagged(unsigned long long *g, unsigned long long tag, double v) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    blr::st_u64(g, (tag << 32) | (bits & 0xffffffffull));
    blr::st_u64(g + 1, (tag << 32) | (bits >> 32));
}

// the weights and radii of every (step, chain) spelled out: [TB][LW + 1] / [TB] (the same expression the kernels stage their weights with)
static __global__ void build_wtab_kernel(const int *tap, const int *tap_lw, const int *tap_off, const double *taps, long long TB, int LW,
                                  double *wtab, int *lwtab) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= TB * (LW + 1)) return;
    const long long tb = i / (LW + 1);
    const int k = (int)(i - tb * (LW + 1));
    const int tp = tap[tb];
    const int lw = tp >= 0 ? tap_lw[tp] : 0;
    wtab[i] = k <= lw ? (lw > 0 ? taps[tap_off[tp] + k] : 1.0) : 0.0;
    if (k == 0) lwtab[tb] = lw;
}

template <int OM, bool BWD>
__global__ __launch_bounds__(NT) void persist1d_kernel(const P1Params P) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int n = P.n, halo = P.K * P.LW, W = P.TJ + 2 * halo;
    double *cur = lds, *nxt = lds + W, *g1s = lds + 2 * W, *cAs = lds + 3 * W, *als = lds + 4 * W;
    double *wls = als + (BWD ? P.K * P.TJ : 0);                   // [K][LW + 1] stencil weights of the K steps
    double *recs = wls + P.K * (P.LW + 1);                        // [K][rec_len] data records of the K steps
    double *red = recs + P.K * P.rec_len;
    int *meta = (int *)(red + 4 * (NT / 64) + 2);                 // [K] source kind, [K] radius
    double *part = red + 4 * (NT / 64) + 2 + P.K;                 // [K][3][TJ] per-cell terms of the K steps' sums
    int *gave_up = (int *)(part + (size_t)P.K * 3 * P.TJ);        // (the allocation's spare words; __syncthreads_or would cost static LDS)
    const int b = blockIdx.y, blkid = blockIdx.x, tid = threadIdx.x;
    const int j0 = blkid * P.TJ, tw = min(P.TJ, n - j0);
    double *post = P.post ? P.post + (long long)b * P.post_stride : nullptr;

    blk::StepParams Q{};
    Q.d = P.d; Q.n1 = n; Q.m0 = nullptr; Q.m1 = P.m1;

    // per-cell constants of the window: the same for every superstep
    for (int e = tid; e < W; e += NT) {
        const int j = blk::reflect(j0 - halo + e, n);
        g1s[e] = P.m1[j];
        if (OM == blk::OM_POISSON) cAs[e] = P.colA[j];
    }

    const int NS = (P.T + P.K - 1) / P.K;
    for (int s = 0; s < NS; ++s) {
        const int t_first = P.dir > 0 ? s * P.K : P.T - 1 - s * P.K;
        const int Ks = min(P.K, P.dir > 0 ? P.T - t_first : t_first + 1);
        if (tid == 0) *gave_up = 0;
        __syncthreads();                                           // the previous superstep is done with LDS
        // ---- everything the superstep needs from memory is REQUESTED first and consumed afterwards (one round trip, not five): a first
        //      attempt at what the other blocks publish (this thread's partial sum and window cell of superstep s - 1: requested whatever
        //      the source kind turns out to be, harmless), then the steps' metadata / weights / data records / stored forward rows -------
        const unsigned long long want = (unsigned long long)(unsigned)s;                 // published by superstep s - 1
        const int par = (s - 1) & 1;
        const unsigned long long *gp = P.gran + ((((long long)par * P.B + b) * P.nblk + min(tid, P.nblk - 1)) << 1);
        const unsigned long long *xp = P.xch + ((((long long)par * P.B + b) * n + blk::reflect(j0 - halo + min(tid, W - 1), n)) << 1);
        unsigned long long gq0 = 0ull, gq1 = 0ull, xq0 = 0ull, xq1 = 0ull;
        if (s > 0) { gq0 = blr::ld_u64(gp); gq1 = blr::ld_u64(gp + 1); xq0 = blr::ld_u64(xp); xq1 = blr::ld_u64(xp + 1); }
        const int kind0 = P.srckind[(long long)t_first * P.B + b];
        constexpr int MW = 2, MA = 4;                              // staged elements per thread held in registers (the rest: loaded in place)
        const int nw = Ks * (P.LW + 1), na = BWD ? Ks * P.TJ : 0;
        double wreg[MW], areg[MA];
        int kreg[MW], lreg[MW];
#pragma unroll
        for (int r = 0; r < MW; ++r) {
            const int e = tid + r * NT;
            wreg[r] = 0.0; kreg[r] = 0; lreg[r] = 0;
            if (P.wtab && e < nw) {
                const int st = e / (P.LW + 1), k = e - st * (P.LW + 1);
                const long long tb = (long long)(t_first + P.dir * st) * P.B + b;
                wreg[r] = P.wtab[tb * (P.LW + 1) + k];
                if (k == 0) { kreg[r] = P.srckind[tb]; lreg[r] = P.lwtab[tb]; }
            }
        }
#pragma unroll
        for (int r = 0; r < MA; ++r) {
            const int e = tid + r * NT;
            areg[r] = 0.0;
            if (e < na) {
                const int st = e / P.TJ, c = e - st * P.TJ;
                areg[r] = c < tw ? post[(long long)(t_first + P.dir * st) * n + j0 + c] : 0.0;
            }
        }
        for (int e = tid; e < Ks * P.rec_len; e += NT) {
            const int st = e / P.rec_len, k = e - st * P.rec_len;
            recs[e] = P.rec[(long long)(t_first + P.dir * st) * P.rec_len + k];
        }
        // ---- consume ------------------------------------------------------------------------------------------------------------------------
#pragma unroll
        for (int r = 0; r < MW; ++r) {
            const int e = tid + r * NT;
            if (P.wtab && e < nw) {
                wls[e] = wreg[r];
                const int st = e / (P.LW + 1);
                if (e - st * (P.LW + 1) == 0) { meta[st] = kreg[r]; meta[P.K + st] = lreg[r]; }
            }
        }
        for (int e = P.wtab ? tid + MW * NT : tid; e < nw; e += NT) {      // (no table, or more elements than the registers hold)
            const int st = e / (P.LW + 1), k = e - st * (P.LW + 1);
            const long long tb = (long long)(t_first + P.dir * st) * P.B + b;
            const int tp = P.tap[tb];
            const int lw = tp >= 0 ? P.tap_lw[tp] : 0;
            wls[e] = k <= lw ? (lw > 0 ? P.taps[P.tap_off[tp] + k] : 1.0) : 0.0;
            if (k == 0) { meta[st] = P.srckind[tb]; meta[P.K + st] = lw; }
        }
        if (BWD) {
#pragma unroll
            for (int r = 0; r < MA; ++r) {
                const int e = tid + r * NT;
                if (e < na) als[e] = areg[r];
            }
            for (int e = tid + MA * NT; e < na; e += NT) {
                const int st = e / P.TJ, c = e - st * P.TJ;
                als[e] = c < tw ? post[(long long)(t_first + P.dir * st) * n + j0 + c] : 0.0;
            }
        }
        // ---- the window of the input state ------------------------------------------------------------------------------------------------
        bool alive = true;
        double scale = 1.0;
        if (s > 0) {
            // the sums of ALL blocks, also when this superstep restarts from a shared distribution and has no use for them: every block
            // has then finished reading the buffers of superstep s - 2 before anybody publishes superstep s into them
            double v[1] = {0.0};
            if (tid < P.nblk) {
                double x;
                alive = finish_tagged(P, gp, want, gq0, gq1, x) && alive;
                v[0] += x;
            }
            for (int k = tid + NT; k < P.nblk; k += NT) {
                double x;
                alive = fetch_tagged(P, P.gran + ((((long long)par * P.B + b) * P.nblk + k) << 1), want, x) && alive;
                v[0] += x;
            }
            blk::block_sums<1, NT / 64>(v, red);
            scale = 1.0 / v[0];
        }
        if (s == 0 || kind0 != SRC_PREV) {
            const double *src = kind0 == SRC_PREV ? P.src0 + (long long)b * P.src0_stride : P.shared[kind0];
            for (int e = tid; e < W; e += NT) cur[e] = src[blk::reflect(j0 - halo + e, n)];
        } else {
            if (tid < W) {
                double x;
                alive = finish_tagged(P, xp, want, xq0, xq1, x) && alive;
                cur[tid] = x * scale;
            }
            for (int e = tid + NT; e < W; e += NT) {
                const int j = blk::reflect(j0 - halo + e, n);
                double x;
                alive = fetch_tagged(P, P.xch + ((((long long)par * P.B + b) * n + j) << 1), want, x) && alive;
                cur[e] = x * scale;
            }
        }
        if (!alive) *gave_up = 1;
        __syncthreads();
        if (*gave_up) return;                                      // a block gave up (the abort word is set): the host repeats the pass

        for (int st = 0; st < Ks; ++st) {
            const int t = t_first + P.dir * st;
            if (st > 0) __syncthreads();                           // previous step's writes of nxt / reads of cur are done
            const int kind = meta[st], lw = meta[P.K + st];
            const double *wl = wls + st * (P.LW + 1);
            if (st > 0 && kind != SRC_PREV) {                      // restart from a shared distribution
                const double *src = P.shared[kind];
                for (int e = tid; e < W; e += NT) cur[e] = src[blk::reflect(j0 - halo + e, n)];
                __syncthreads();
            }
            Q.rec = recs + st * P.rec_len;
            Q.lik = P.lik ? P.lik + (long long)t * n : nullptr;
            double *row = post ? post + (long long)t * n : nullptr;
            double *pt = part + (size_t)st * 3 * P.TJ;
            const int lo = (st + 1) * P.LW, hi = W - (st + 1) * P.LW;    // cells that are still exact after this step
            bl1f::advance_cells<OM, BWD>(Q, n, j0, halo, tw, P.TJ, lo, hi, cur, nxt, g1s, cAs, wl, lw, als + st * P.TJ, row, pt, P.store != 0, tid);
            double *tmp = cur; cur = nxt; nxt = tmp;
        }
        __syncthreads();
        // ---- hand the superstep's result over: first the state the neighbours wait for, then the sum of its last step ----------------
        const unsigned long long tag = (unsigned long long)(unsigned)(s + 1);
        const bool more = s + 1 < NS;
        if (more) {
            unsigned long long *x = P.xch + ((((long long)(s & 1) * P.B + b) * n + j0) << 1);
            for (int c = tid; c < tw; c += NT) publish_tagged(x + 2 * c, tag, cur[halo + c]);
        }
        // ---- the sums of the Ks steps: wave w takes the (step, slot) pairs w, w + 8, ...; fixed order -> deterministic ----------------
        for (int pr = tid >> 6; pr < 4 * Ks; pr += NT / 64) {
            const int st = pr >> 2, k = pr & 3, lane = tid & 63;
            if (!(k == 0 || BWD || (k == 3 && P.means))) continue;
            const double *pt = part + (size_t)st * 3 * P.TJ + (k == 3 ? 0 : k) * P.TJ;
            double acc = 0.0;
            if (k == 3) { for (int c = lane; c < tw; c += 64) acc = fma(pt[c], g1s[halo + c], acc); }
            else { for (int c = lane; c < tw; c += 64) acc += pt[c]; }
            acc = blk::wave_sum(acc);
            if (lane == 0) {
                const long long tb = (long long)(t_first + P.dir * st) * P.B + b;
                P.psum[(tb * NRED + k) * P.nblk + blkid] = acc;
                if (more && st == Ks - 1 && k == P.prev_slot)
                    publish_tagged(P.gran + ((((long long)(s & 1) * P.B + b) * P.nblk + blkid) << 1), tag, acc);
            }
        }
    }
    if (P.dst) {
        double *d = P.dst + (long long)b * P.dst_stride + j0;
        for (int c = tid; c < tw; c += NT) d[c] = cur[halo + c];
    }
}

}  // namespace bl1p
