// The time-resident kernel's 64 x 64 tile with BOTH stencils on the matrix pipe (round 5).
//
// blr::resident_kernel<64, 64, 8, 8, ...> (blhip_resident.hpp: BASELINE C3, 1024 x 1024, T = 2000) runs its two 17-tap passes on the
// vector ALU: a thread owns 8 cells of a row / column and walks them with a register window.  Its walks are issue-bound at two waves per
// SIMD (one fp64 vector instruction per wave every ~16 cycles: half the pipe's rate; profiles/r04_notes.md: 0.54 / 0.61 busy), and two
// LDS pre-read phases with a barrier each stand in front of them.  The chain-resident kernels (blhip_chainres.hpp) apply the same
// stencil as banded Toeplitz products on v_mfma_f64_4x4x4_4b, which run at the pipe's rate with two waves per SIMD.  This kernel is the
// tile kernel with that product:
//
//  * SAME protocol as resident_kernel, so the host side (blhip_fit_paths.hpp: ResidentRun -- strips, granules, lag, predicted sums,
//    partial-sum slots, give-up) is untouched and the two kernels are interchangeable per launch: tagged column strips of the new
//    state after a step, tagged row strips of the axis-1-filtered tile in the middle of it, the lagged sum gathered by half of the waves;
//  * the tile [64][65] and a second one for the axis-1 pass's output (a wave's product reads columns other waves write), the halos in
//    LDS arrays of their own ([64][8] raw columns left / right, [8][65] filtered rows top / bottom) filled from the neighbours' strips
//    -- or, at a grid edge, with the mirror image -- while the waves run the products that do not need them: every wave owns one
//    INTERIOR product tile and one EDGE product tile per pass and does the interior one first;
//  * axis 1 (along the columns) as OUT^T = W1 X^T: the B operand is read with the row stride (pitch 65: conflict-free); axis 0 as
//    OUT = W0 Y with chain_kernel's epilogue behind it (stride-4 likelihood recurrence down the lane's rows, lagged scale, sums).
// Flavours: evidence-only forward (MODE 1), full-fit forward (MODE 2), backward.  Everything else (forward-only fits with their in-kernel
// normalisation, padded tiles, tabulated likelihoods, other tile shapes) keeps resident_kernel.
#pragma once
#include "blhip_resident.hpp"
#include "blhip_chainres.hpp"

namespace blr {

template <bool BWD, int MODE>
struct ResM {
    using K = Res<64, 64, 8, 8, BWD, MODE, false, false>;
    static constexpr int TR = 64, TC = 64, P = K::P, NT = 512, NW = 8, NK = 8, R0 = 8;
    static constexpr int LDS_Y = K::LDS_DOUBLES;             // [TR][P]   axis-1-filtered tile
    static constexpr int LDS_HL = LDS_Y + TR * P;            // [TR][R]   raw columns -8 .. -1
    static constexpr int LDS_HR = LDS_HL + TR * R;           // [TR][R]   raw columns 64 .. 71
    static constexpr int LDS_HT = LDS_HR + TR * R;           // [R][P]    filtered rows -8 .. -1
    static constexpr int LDS_HB = LDS_HT + R * P;            // [R][P]    filtered rows 64 .. 71
    static constexpr int LDS_A0 = LDS_HB + R * P;            // [NK][16]  band of axis 0
    static constexpr int LDS_A1 = LDS_A0 + NK * 16;          // [NK][16]  band of axis 1
    static constexpr int LDS_DOUBLES = LDS_A1 + NK * 16 + 8;
};

template <bool BWD, int MODE>
__global__ __launch_bounds__(512, 1) void resident_mfma_kernel(const ResParams Q) {
    using M = ResM<BWD, MODE>;
    using K = typename M::K;
    constexpr int TR = M::TR, TC = M::TC, P = M::P, NT = M::NT, NW = M::NW, NK = M::NK;
    static_assert(MODE == 0 ? BWD : (MODE == 1 || MODE == 2), "evidence-only / full-fit forward, backward");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *const X = lds, *const Y = lds + M::LDS_Y, *const HL = lds + M::LDS_HL, *const HR = lds + M::LDS_HR;
    double *const HT = lds + M::LDS_HT, *const HB = lds + M::LDS_HB, *const A0 = lds + M::LDS_A0, *const A1 = lds + M::LDS_A1;
    double *misc = lds + K::LDS_MISC;
    double *red = lds + K::LDS_RED;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    typename K::Thread th;
    th.init(Q, blockIdx.x, tid, lds);
    for (int e = tid; e < TR; e += NT) lds[K::LDS_M0 + e] = Q.m0[th.i0 + e];
    for (int e = tid; e < TC; e += NT) { lds[K::LDS_COL + e] = Q.m1[th.j0 + e]; lds[K::LDS_COL + TC + e] = Q.colA[th.j0 + e]; lds[K::LDS_COL + 2 * TC + e] = Q.colB[th.j0 + e]; }
    for (int e = tid; e < NK * 16; e += NT) {
        const int a = blc::band_distance16(e, M::R0);
        A0[e] = a <= R ? Q.w0[a] : 0.0;
        A1[e] = a <= R ? Q.w1[a] : 0.0;
    }
    if (tid == 0) { misc[1] = 0.0; misc[2] = 0.0; }
    const int gw = th.gather_wave();
    __syncthreads();

    auto fresh_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
    typedef const double __attribute__((address_space(3))) *lds_cp;
    auto band_ptr = [&](const double *Ab, int l) {
        const unsigned aoff = (unsigned)(((l >> 4) << 2) | (l & 3)) * 8u;
        return (lds_cp)((const char __attribute__((address_space(3))) *)(lds_cp)Ab + aoff);
    };
    // the wave's product tiles of a pass: an interior one (needs no halo) first, an edge one second
    //   axis 1: interior = column tiles 1, 2 / edge = 0, 3, row tile wv >> 1;    axis 0: interior = row tiles 1, 2 / edge = 0, 3, column tile wv >> 1
    // ---- axis 1: Y[row][col] = sum_j w1[|j|] X[row][col + j]  as  OUT^T = W1 X^T -------------------------------------------------------
    auto axis1_tile = [&](int rt, int ct) {
        const int l = fresh_lane(), g = l >> 4, cc = l & 15;
        const double *xrow = X + (16 * rt + cc) * P;
        double Bv[NK];
#pragma unroll
        for (int kb = 0; kb < NK; ++kb) {
            const int col = 16 * ct - 8 + 4 * kb + g;                       // -8 .. 71
            if (kb < 2 && ct == 0) Bv[kb] = HL[(16 * rt + cc) * R + col + 8];
            else if (kb >= NK - 2 && ct == 3) Bv[kb] = HR[(16 * rt + cc) * R + col - 64];
            else Bv[kb] = xrow[col];
        }
        const blc::d4 acc = blc::band_products<NK, 0, NK, 16>(band_ptr(A1, l), Bv);
#pragma unroll
        for (int r = 0; r < 4; ++r) Y[(16 * rt + cc) * P + 16 * ct + g + 4 * r] = acc[r];
    };
    // ---- axis 0 + epilogue ---------------------------------------------------------------------------------------------------------------
    double scale = 1.0, invn = 1.0;
    double mq_c[2] = {1.0, 1.0}, iq_c[2] = {1.0, 1.0}, dn_prev[2] = {-1.0, -1.0};
    int nq_c[2] = {0, 0};
    double al[2][4];                                 // backward: the stored forward state of the wave's two product tiles
    // axis 0: slot 0 = the tiles of the tile's first / last 16 COLUMNS (their new state's edge columns go out as soon as they are done), slot 1 = the others
    auto tile_of_slot = [&](int slot, int &rt, int &ct) { ct = slot == 0 ? (wv & 1) * 3 : 1 + (wv & 1); rt = wv >> 1; };
    auto row_base = [&](int k, int rt, int ct, int l) {          // global pointer of the lane's first cell (row 16 rt + g, column 16 ct + c) at the step's time
        return th.row_ptr(Q, k, 16 * rt + (l >> 4), 16 * ct + (l & 15));
    };
    auto epilogue_tile = [&](int k, int slot, int rt, int ct, blc::d4 acc) {
        const int l = fresh_lane(), g = l >> 4, c = l & 15;
        const int row = 16 * rt + g, col = 16 * ct + c;
        const double g1 = lds[K::LDS_COL + col], cA = lds[K::LDS_COL + TC + col], cB = lds[K::LDS_COL + 2 * TC + col];
        const double *m0p = lds + K::LDS_M0;
        // anchors of the stride-4 likelihood recurrence down the lane's rows (observationModels.py:566-567; blhip_chainres.hpp)
        double mE, mR, iE = 1.0, iR = 1.0;
        int nE, nR;
        {
            const double mu0 = m0p[row], mu4 = m0p[row + 4];
            double a0 = 0.0, s1 = 0.0, dn = 0.0;
#pragma unroll
            for (int q = 0; q < DMAX; ++q) {
                const double x = th.xd[q];
                if (x == x) {
                    const double dq = x - mu0;
                    a0 = fma(-(dq * dq), cA, a0) - cB;
                    s1 += (x - mu0) + (x - mu4);
                    dn += 1.0;
                }
            }
            const double d1 = cA * (mu4 - mu0) * s1;
            blmath::exp_mn(a0, mE, nE);
            blmath::exp_mn(d1, mR, nR);
            if (dn != dn_prev[slot]) {                     // (per slot: the wave's two product tiles sit in different columns, cA is the column's)
                int tmp;
                blmath::exp_mn(-32.0 * cA * dn * Q.step0 * Q.step0, mq_c[slot], nq_c[slot]);
                if (BWD) blmath::exp_mn(32.0 * cA * dn * Q.step0 * Q.step0, iq_c[slot], tmp);
                dn_prev[slot] = dn;
            }
            if (BWD) { iE = blmath::inv_m(mE); iR = blmath::inv_m(mR); }
            else mE *= scale;
        }
        double *pt = K::f_post(Q) ? row_base(k, rt, ct, l) : nullptr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double Lv = ldexp(mE, nE);
            double keep;
            if (!BWD) {
                const double a = acc[r] * Lv;
                keep = a;
                if (K::f_store(Q)) st_stream(pt + (long long)(4 * r) * Q.n1, a);
                th.sums[0] += a;
            } else {
                const double beta = acc[r] * scale;
                const double pp = al[slot][r] * beta;
                const double cn = beta * Lv;
                const double pl = Lv == 0.0 ? nan_() : ldexp(pp * iE, -nE);          // p / L; 0 / 0 -> NaN (core.py:463)
                keep = cn;
                st_stream(pt + (long long)(4 * r) * Q.n1, pp * invn);             // (invn = 1 / predicted sum: stored normalised)
                th.sums[0] += pp; th.sums[1] += pl; th.sums[2] += cn;
                th.sums[3] = fma(pp, m0p[row + 4 * r], th.sums[3]); th.sums[4] = fma(pp, g1, th.sums[4]);
            }
            X[(row + 4 * r) * P + col] = keep;
            mE *= mR; nE += nR;
            mR *= mq_c[slot]; nR += nq_c[slot];
            if (BWD) { iE *= iR; iR *= iq_c[slot]; }
        }
    };
    auto axis0_tile = [&](int k, int slot) {
        int rt, ct;
        tile_of_slot(slot, rt, ct);
        const int l = fresh_lane(), g = l >> 4, c = l & 15;
        double Bv[NK];
#pragma unroll
        for (int kb = 0; kb < NK; ++kb) {
            const int row = 16 * rt - 8 + 4 * kb + g;                       // -8 .. 71
            if (kb < 2 && rt == 0) Bv[kb] = HT[(row + 8) * P + 16 * ct + c];
            else if (kb >= NK - 2 && rt == 3) Bv[kb] = HB[(row - 64) * P + 16 * ct + c];
            else Bv[kb] = Y[row * P + 16 * ct + c];
        }
        epilogue_tile(k, slot, rt, ct, blc::band_products<NK, 0, NK, 16>(band_ptr(A0, l), Bv));
    };
    // ---- halos: neighbours' strips (tagged elements) or the mirror image at a grid edge -> LDS -----------------------------------------
    // two elements per thread: e = tid + 512 j -> side = e >> 9, line = (e >> 6) & 7 (towards the tile: 7 is next to it), pos = e & 63
    Tq hq[2];
    auto wait_two = [&](Rsrc rs, const unsigned (&off)[2], const bool (&live)[2], unsigned tag) {
        auto ok = [&]() { return (!live[0] || tq_ok(hq[0], tag)) && (!live[1] || tq_ok(hq[1], tag)); };
        if (th.dead || ok()) return;
        const unsigned long long t0 = now_ticks();
        for (unsigned spins = 1;; ++spins) {
            nap();
#pragma unroll
            for (int j = 0; j < 2; ++j) if (live[j]) hq[j] = ld_tq(rs, off[j]);
            if (ok()) return;
            if ((spins & 255u) == 0u) {
                if (ld_flag(Q.abort_word) != 0u) { th.dead = true; return; }
                if (now_ticks() - t0 > Q.timeout_ticks) { st_flag(Q.abort_word, 1u); th.dead = true; return; }
            }
        }
    };
    unsigned hoff[2];
    bool hlive[2];
    // raw edge columns of the neighbours' state of step k - 1 (tag k): element [(k - 1) & 1][nb][side][cc][row]
    auto cols_issue = [&](int k, Rsrc rs) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned e = (unsigned)tid + 512u * (unsigned)j, side = e >> 9, line = (e >> 6) & 7u, row = e & 63u;
            const bool have = side == 0 ? th.tj > 0 : th.tj < th.tc - 1;
            const int nb = side == 0 ? th.tile - 1 : th.tile + 1;
            hlive[j] = have;
            hoff[j] = (unsigned)((((((k - 1) & 1) * Q.ntiles + nb) * 2 + (side == 0 ? 1 : 0)) * R + (int)line) * TR + (int)row) * 8u;
            if (have) hq[j] = ld_tq(rs, hoff[j]);
        }
    };
    auto cols_finish = [&](int k, Rsrc rs) {
        wait_two(rs, hoff, hlive, tag_bit((unsigned)k, false));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned e = (unsigned)tid + 512u * (unsigned)j, side = e >> 9, line = (e >> 6) & 7u, row = e & 63u;
            // mirror at the grid edge (half-sample reflection): column -1 - m <- column m;  column 64 + m <- column 63 - m
            const double v = hlive[j] ? tq_value(hq[j]) : (side == 0 ? X[row * P + (7 - line)] : X[row * P + (63 - line)]);
            (side == 0 ? HL : HR)[row * R + line] = v;
        }
    };
    // axis-1-filtered edge rows of the neighbours at THIS step (tag k + 1): element [k & 1][nb][side][rr][col]
    auto rows_issue = [&](int k, Rsrc rs) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned e = (unsigned)tid + 512u * (unsigned)j, side = e >> 9, line = (e >> 6) & 7u, col = e & 63u;
            const bool have = side == 0 ? th.ti > 0 : th.ti < th.tr - 1;
            const int nb = side == 0 ? th.tile - th.tc : th.tile + th.tc;
            hlive[j] = have;
            hoff[j] = (unsigned)(((((k & 1) * Q.ntiles + nb) * 2 + (side == 0 ? 1 : 0)) * R + (int)line) * TC + (int)col) * 8u;
            if (have) hq[j] = ld_tq(rs, hoff[j]);
        }
    };
    auto rows_finish = [&](int k, Rsrc rs) {
        wait_two(rs, hoff, hlive, tag_bit((unsigned)(k + 1), true));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned e = (unsigned)tid + 512u * (unsigned)j, side = e >> 9, line = (e >> 6) & 7u, col = e & 63u;
            const double v = hlive[j] ? tq_value(hq[j]) : (side == 0 ? Y[(7 - line) * P + col] : Y[(63 - line) * P + col]);
            (side == 0 ? HT : HB)[line * P + col] = v;
        }
    };

    // Bookkeeping of a finished step (the whole wave: lane w fetches wave w's sums): resident_kernel's
    constexpr int BOOK_WAVE = NW / 2 - 1;
    const bool booker = wv == BOOK_WAVE;
    auto book = [&](int kb) {
        constexpr int NV = BWD ? 5 : 3;
        double tot[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            tot[q] = 0.0;
            if (!BWD && q > 0) continue;
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
                K::publish_sum(Q, th.tile, kb, 0, tot[0]);
            }
        }
    };
    auto load_alpha = [&](int k) {
        if (!BWD) return;
#pragma unroll
        for (int slot = 0; slot < 2; ++slot) {
            int rt, ct;
            tile_of_slot(slot, rt, ct);
            const double *pt = row_base(k, rt, ct, fresh_lane());
#pragma unroll
            for (int r = 0; r < 4; ++r) al[slot][r] = ld_stream(pt + (long long)(4 * r) * Q.n1);
        }
    };

    // (option resident_prof: shader-clock stamps of an interior tile, waves 0 and 5, steps 8 .. 23)
    const bool prof_me = Q.prof != nullptr && th.tile == Q.ntiles / 2 + Q.tc / 2 && lane == 0 && (wv == 0 || wv == 5);
#define BLM_STAMP(i) do { if (prof_me && k >= 8 && k < 24) Q.prof[(wv ? 256 : 0) + (k - 8) * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
    for (int k = 0; k < Q.T; ++k) {
#pragma unroll
        for (int q = 0; q < 5; ++q) th.sums[q] = 0.0;
        BLM_STAMP(0);
        if (k == 0) {
            // the first executed step has no transition: its input is src0 (prior / uniform), scale 1
            th.begin_step(Q, 0);
            th.predicted_sum(Q, 0, 1.0);
            scale = 1.0;
            invn = BWD ? 1.0 / Q.n_first : 1.0;
            load_alpha(0);
#pragma unroll
            for (int slot = 0; slot < 2; ++slot) {
                int rt, ct;
                tile_of_slot(slot, rt, ct);
                const int l = fresh_lane();
                const double *s = Q.src0 + (long long)(th.i0 + 16 * rt + (l >> 4)) * Q.n1 + (th.j0 + 16 * ct + (l & 15));
                blc::d4 acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = s[(long long)(4 * r) * Q.n1];
                epilogue_tile(0, slot, rt, ct, acc);
                if (slot == 0) { lds_barrier(); th.publish_cols(Q, 0); }
            }
        } else {
            const Rsrc rsc = strip_rsrc(Q.cols, Q.cols_bytes), rsr = strip_rsrc(Q.rows, Q.rows_bytes);
            // Every hand-off gets most of a pass of slack: the producer computes what it hands over FIRST and publishes it in the middle
            // of its pass, the consumer needs it at the beginning of its next pass.
            // ---- the neighbours' raw edge columns (published in the middle of their previous axis-0 pass) -> LDS ---------------------------
            cols_issue(k, rsc);
            if (k >= Q.lag && gw >= 0) th.gather_issue(Q, k - Q.lag);
            cols_finish(k, rsc);
            BLM_STAMP(1);
            load_alpha(k);                                 // (backward: this step's stored state, a pass ahead of the epilogue)
            lds_barrier();
            BLM_STAMP(2);
            // ---- axis 1, first the tiles of the first / last 16 ROWS: their filtered edge rows go out ------------------------------------------
            axis1_tile((wv & 1) * 3, wv >> 1);
            BLM_STAMP(3);
            lds_barrier();
            BLM_STAMP(4);
            if (!(Q.dbg & 2)) {
            th.template publish_strip<TC>(rsr, ((k & 1) * Q.ntiles + th.tile) * 2 * R * TC, tag_bit((unsigned)(k + 1), true), th.ti > 0, th.ti < th.tr - 1,
                                          [&](int side, unsigned rr, unsigned col) { return Y[((side ? TR - R : 0) + rr) * P + col]; });
            rows_issue(k, rsr);
            }
            axis1_tile(1 + (wv & 1), wv >> 1);
            BLM_STAMP(5);
            if (booker) book(k - 1);
            if (k >= Q.lag && gw >= 0) {                   // the lagged sum (resident_kernel's: half of the waves gather, the last one combines)
                double part[K::NG];
                th.gather_finish(Q, k - Q.lag, part);
                const double ws = blk::wave_sum(part[0]);
                if (lane == 0) {
                    misc[8 + gw] = ws;
                    unsigned *cnt = reinterpret_cast<unsigned *>(misc + 2);
                    const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (old == (unsigned)K::GW - 1u) { *cnt = 0u; th.combine_shares(); }
                }
            }
            BLM_STAMP(6);
            if (Q.dbg & 2) {
                lds_barrier();
            th.template publish_strip<TC>(rsr, ((k & 1) * Q.ntiles + th.tile) * 2 * R * TC, tag_bit((unsigned)(k + 1), true), th.ti > 0, th.ti < th.tr - 1,
                                          [&](int side, unsigned rr, unsigned col) { return Y[((side ? TR - R : 0) + rr) * P + col]; });
            rows_issue(k, rsr);
            }
            rows_finish(k, rsr);                           // (the mirror image at a grid edge reads rows 0 .. 7 / 56 .. 63 of Y: complete since the last barrier)
            BLM_STAMP(7);
            lds_barrier();                                 // Y is complete, the scale is there, the row halos are in LDS
            BLM_STAMP(8);
            // ---- axis 0 + epilogue, first the tiles of the first / last 16 COLUMNS: the new state's edge columns go out ------------------------
            scale = th.lagged_inverse(Q, k, 0);
            th.predicted_sum(Q, k, scale);
            invn = BWD ? 1.0 / th.npred : 1.0;
            axis0_tile(k, 0);
            BLM_STAMP(9);
            lds_barrier();
            if (!(Q.dbg & 1)) th.publish_cols(Q, k);
            axis0_tile(k, 1);
            BLM_STAMP(10);
        }
        if (k + 1 < Q.T) th.begin_step(Q, k + 1);         // the next step's data record (scalar loads)
        // ---- sums of the step: resident_kernel's --------------------------------------------------------------------------------------------
        if (th.dead) misc[1] = 1.0;
        {
            constexpr int NV = BWD ? 5 : 3;
            double v[NV];
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q] = th.sums[BWD ? q : (q == 0 ? 0 : q + 2)];
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                if (!BWD && q > 0) break;
                const double ws = blk::wave_sum(v[q]);
                if (lane == 0) red[wv * NV + q] = ws;
            }
            lds_barrier();                                 // the tile's new state is complete in LDS, the waves' sums and misc[1] are final
            BLM_STAMP(11);
            if ((Q.dbg & 1) && k > 0) th.publish_cols(Q, k);
        }
        if (misc[1] != 0.0) return;                        // a wait timed out somewhere in this block: uniform exit (host falls back)
    }
    if (booker) book(Q.T - 1);
#undef BLM_STAMP
}

}  // namespace blr
