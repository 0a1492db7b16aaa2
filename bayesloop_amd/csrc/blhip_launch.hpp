// The launch tables: which kernel instantiation a (model, radius, direction, flavour) tuple selects -- every family but the chain-resident slices
// (blhip_chain_tu.hip).  Every launch goes through BL_LAUNCH (blhip_err.hpp): the kernel registry behind blhip_kernel_census.
// Part of libblhip's host side: included by blhip.hip INSIDE its anonymous namespace (one translation unit; the split is by subject, not by linkage).
#pragma once

struct Tile {
    int TI, TJ, LW0, LW1, tiles_i, tiles_j, nblk;
    size_t lds_bytes;
};

size_t lds_need(int TI, int TJ, int LW0, int LW1) {
    const size_t pitch = (size_t)TJ + 2 * LW1;
    return ((size_t)(TI + 2 * LW0) * pitch + (size_t)TI * pitch + 32) * sizeof(double);
}

Tile choose_tile(const blhip_ctx *ctx, const Geometry &g, int LW0, int LW1, bool whole_row = false) {
    Tile t{};
    size_t cap = (size_t)64 * 1024;
    int TI, TJ;
    if (g.n0 == 1 && whole_row) {                // (a two-stage spline shift: one block per chain holds the whole row)
        TI = 1;
        TJ = g.n1;
        cap = 160 * 1024 - 512;
    } else if (g.n0 == 1) {
        TI = 1;
        TJ = g.n1 <= 65536 ? 256 : 1024;
    } else {
        TI = 16;
        TJ = 128;
    }
    TI = std::max(1, std::min(TI, g.n0));
    TJ = std::max(1, std::min(TJ, g.n1));
    while (lds_need(TI, TJ, LW0, LW1) > cap) {
        if (TI > 4 && (TI >= TJ / 4 || TJ <= 32)) TI = (TI + 1) / 2;
        else if (TJ > 16) TJ = (TJ + 1) / 2;
        else if (TI > 1) TI = (TI + 1) / 2;
        else break;
    }
    if (lds_need(TI, TJ, LW0, LW1) > 160 * 1024 - 512)
        fail("filter radius (%d, %d) too large for the fused step kernel (needs %zu B of LDS)", LW0, LW1,
             lds_need(TI, TJ, LW0, LW1));
    t.TI = TI; t.TJ = TJ; t.LW0 = LW0; t.LW1 = LW1;
    t.tiles_i = (g.n0 + TI - 1) / TI;
    t.tiles_j = (g.n1 + TJ - 1) / TJ;
    t.nblk = t.tiles_i * t.tiles_j;
    t.lds_bytes = lds_need(TI, TJ, LW0, LW1);
    return t;
}

template <int OM, int MODE, bool MEANS>
void launch_step_t(hipStream_t s, const StepParams &P, const Tile &t, int B) {
    arm_kernel(reinterpret_cast<const void *>(&step_kernel<OM, MODE, MEANS>));
    BL_LAUNCH((step_kernel<OM, MODE, MEANS>), dim3(t.nblk, B), dim3(NTHREADS), t.lds_bytes, s, P);
}

template <int OM>
void launch_step_om(hipStream_t s, const StepParams &P, const Tile &t, int B, int mode, bool means) {
    if (mode == MODE_FWD) {
        if (means) launch_step_t<OM, MODE_FWD, true>(s, P, t, B);
        else launch_step_t<OM, MODE_FWD, false>(s, P, t, B);
    } else if (mode == MODE_BWD) {
        launch_step_t<OM, MODE_BWD, true>(s, P, t, B);
    } else {
        // (blk::MODE_FILTER -- the transition alone -- has no caller: the models' plug-in calls run a resumed forward step with a flat
        //  likelihood; its four instantiations were pruned in round 6)
        fail("internal: generic step kernel launched in mode %d", mode);
    }
}

void launch_step(hipStream_t s, int om, const StepParams &P, const Tile &t, int B, int mode, bool means) {
    switch (om) {
        case BLHIP_OM_POISSON: launch_step_om<OM_POISSON>(s, P, t, B, mode, means); break;
        case BLHIP_OM_GAUSSIAN: launch_step_om<OM_GAUSSIAN>(s, P, t, B, mode, means); break;
        case BLHIP_OM_GAUSSIAN_MEAN: launch_step_om<OM_GAUSSIAN_MEAN>(s, P, t, B, mode, means); break;
        case BLHIP_OM_TABLE: launch_step_om<OM_TABLE>(s, P, t, B, mode, means); break;
        default: fail("unknown observation model %d", om);
    }
    HIPCHECK(hipGetLastError());
}


// ---- fast path (blhip_fast.hpp): 2-D grids, axis-0 radius <= 40, axis-1 radius <= 8 -------------------------------
constexpr int FAST_R0_MAX = 40;

template <int OM, int MODE, int R0>
void launch_fast_r(hipStream_t s, const blf::FastParams &P, bool H, int nchains) {
    constexpr bool G = OM == OM_GAUSSIAN;
    const dim3 grid(P.fnblk, nchains), block(NTHREADS);
    if (G && P.use_rec) {
        if (H) BL_LAUNCH((blf::fast_step_kernel<OM, MODE, R0, true, G>), grid, block, 0, s, P);
        else BL_LAUNCH((blf::fast_step_kernel<OM, MODE, R0, false, G>), grid, block, 0, s, P);
    } else {
        if (H) BL_LAUNCH((blf::fast_step_kernel<OM, MODE, R0, true, false>), grid, block, 0, s, P);
        else BL_LAUNCH((blf::fast_step_kernel<OM, MODE, R0, false, false>), grid, block, 0, s, P);
    }
}

template <int OM, int MODE>
void launch_fast_om(hipStream_t s, const blf::FastParams &P, int R0, bool H, int nchains) {
    switch (R0) {
        case 0: launch_fast_r<OM, MODE, 0>(s, P, H, nchains); break;
        case 8: launch_fast_r<OM, MODE, 8>(s, P, H, nchains); break;
        case 16: launch_fast_r<OM, MODE, 16>(s, P, H, nchains); break;
        case 24: launch_fast_r<OM, MODE, 24>(s, P, H, nchains); break;
        case 32: launch_fast_r<OM, MODE, 32>(s, P, H, nchains); break;
        case 40: launch_fast_r<OM, MODE, 40>(s, P, H, nchains); break;
        default: fail("fast path: bad radius bucket %d", R0);
    }
}

// ---- matrix-pipe path (blhip_mfma.hpp): stencils as banded products on v_mfma_f64_16x16x4, K = 16 + 2*R0 = 4*NK ----------
template <int OM, int MODE, int NK, bool H>
void launch_mfma_k(hipStream_t s, const blf::FastParams &P, int nchains) {
    const dim3 grid(P.mnblk, nchains), block(H ? blm::NT_H : blm::NT_V);
    constexpr bool G = OM == OM_GAUSSIAN;
    if constexpr (!H) {
        if (P.mlean) {     // whole tile groups inside the grid, 32-bit offsets (blhip_mfma.hpp: LEAN)
            if (G && P.use_rec) BL_LAUNCH((blm::mfma_step_kernel<OM, MODE, NK, G, false, true>), grid, block, 0, s, P);
            else BL_LAUNCH((blm::mfma_step_kernel<OM, MODE, NK, false, false, true>), grid, block, 0, s, P);
            return;
        }
    }
    if (G && P.use_rec) BL_LAUNCH((blm::mfma_step_kernel<OM, MODE, NK, G, H, false>), grid, block, 0, s, P);
    else BL_LAUNCH((blm::mfma_step_kernel<OM, MODE, NK, false, H, false>), grid, block, 0, s, P);
}

template <int OM, int MODE>
void launch_mfma_om(hipStream_t s, const blf::FastParams &P, int R0, bool H, int nchains) {
    if (H) {
        switch (R0) {
            case 0: launch_mfma_k<OM, MODE, 4, true>(s, P, nchains); break;
            case 8: launch_mfma_k<OM, MODE, 8, true>(s, P, nchains); break;
            case 16: launch_mfma_k<OM, MODE, 12, true>(s, P, nchains); break;
            case 24: launch_mfma_k<OM, MODE, 16, true>(s, P, nchains); break;
            case 32: launch_mfma_k<OM, MODE, 20, true>(s, P, nchains); break;
            case 40: launch_mfma_k<OM, MODE, 24, true>(s, P, nchains); break;
            default: fail("mfma path: bad radius bucket %d", R0);
        }
    } else {
        switch (R0) {
            case 8: launch_mfma_k<OM, MODE, 8, false>(s, P, nchains); break;
            case 16: launch_mfma_k<OM, MODE, 12, false>(s, P, nchains); break;
            case 24: launch_mfma_k<OM, MODE, 16, false>(s, P, nchains); break;
            case 32: launch_mfma_k<OM, MODE, 20, false>(s, P, nchains); break;
            case 40: launch_mfma_k<OM, MODE, 24, false>(s, P, nchains); break;
            default: fail("mfma path: bad radius bucket %d", R0);
        }
    }
}

void launch_mfma(hipStream_t s, int om, int mode, const blf::FastParams &P, int R0, bool H, int nchains) {
    if (om == BLHIP_OM_GAUSSIAN) {
        if (mode == MODE_FWD) launch_mfma_om<OM_GAUSSIAN, MODE_FWD>(s, P, R0, H, nchains);
        else launch_mfma_om<OM_GAUSSIAN, MODE_BWD>(s, P, R0, H, nchains);
    } else {
        if (mode == MODE_FWD) launch_mfma_om<OM_TABLE, MODE_FWD>(s, P, R0, H, nchains);
        else launch_mfma_om<OM_TABLE, MODE_BWD>(s, P, R0, H, nchains);
    }
    HIPCHECK(hipGetLastError());
}

void launch_hwide(hipStream_t s, const blh::HParams &P, int nchains) {
    const size_t lds = blh::lds_bytes(P.lwmax);
    arm_kernel(reinterpret_cast<const void *>(&blh::hwide_kernel));
    const dim3 grid((unsigned)(((P.n0 + blh::RB - 1) / blh::RB) * P.tiles_j), (unsigned)nchains);
    BL_LAUNCH(blh::hwide_kernel, grid, dim3(blh::NT), lds, s, P);
    HIPCHECK(hipGetLastError());
}

void launch_vwide(hipStream_t s, const blh::HParams &P, int nchains) {
    const size_t lds = blh::vlds_bytes(P.lwmax);
    arm_kernel(reinterpret_cast<const void *>(&blh::vwide_kernel));
    const dim3 grid((unsigned)(((P.n0 + blh::RV - 1) / blh::RV) * P.tiles_j), (unsigned)nchains);
    BL_LAUNCH(blh::vwide_kernel, grid, dim3(blh::NT), lds, s, P);
    HIPCHECK(hipGetLastError());
}

void launch_fast(hipStream_t s, int om, int mode, const blf::FastParams &P, int R0, bool H, int nchains) {
    if (om == BLHIP_OM_GAUSSIAN) {
        if (mode == MODE_FWD) launch_fast_om<OM_GAUSSIAN, MODE_FWD>(s, P, R0, H, nchains);
        else launch_fast_om<OM_GAUSSIAN, MODE_BWD>(s, P, R0, H, nchains);
    } else {
        if (mode == MODE_FWD) launch_fast_om<OM_TABLE, MODE_FWD>(s, P, R0, H, nchains);
        else launch_fast_om<OM_TABLE, MODE_BWD>(s, P, R0, H, nchains);
    }
    HIPCHECK(hipGetLastError());
}

struct FastRange { int start, count, R0; bool H; int key; bool pre; };

// chains of one step ordered by (axis-0 radius bucket, axis-1 class); one launch per non-empty group.  Axis-1 classes: 0 = no filter,
// 1 = a filter the fused kernels apply themselves (radius <= 8), 2 = (h_fused_max >= 0 only) one wider than h_fused_max (0: any): the group
// runs behind the axis-1 pre-pass (FastRange::pre) and its step kernel without an axis-1 part.
constexpr int NKEYS = 18;
void bucket_step(const int *tap0, const int *tap1, const std::vector<int> &lw, int B, int *order, std::vector<FastRange> &ranges,
                 int min_chains, int h_fused_max = -1) {
    int cnt[NKEYS] = {0};
    int promote[NKEYS];
    for (int k = 0; k < NKEYS; ++k) promote[k] = k;
    auto key0 = [&](int b) {
        const int l0 = tap0[b] >= 0 ? lw[tap0[b]] : 0;
        const int bucket = l0 == 0 ? 0 : (l0 + 7) / 8;      // 0..5
        const int hc = tap1[b] < 0 ? 0 : ((h_fused_max >= 0 && (h_fused_max == 0 || lw[tap1[b]] > h_fused_max)) ? 2 : 1);
        return bucket * 3 + hc;
    };
    auto key = [&](int b) { int k = key0(b); while (promote[k] != k) k = promote[k]; return k; };
    for (int b = 0; b < B; ++b) cnt[key0(b)]++;
    // a bucket with only a few chains cannot fill the chip: promote its chains to the next larger radius bucket
    // (zero-padded weights make that exact); keys are bucket * 3 + class
    for (int h = 0; h < 3; ++h)
        for (int bk = 0; bk < 5; ++bk) {
            const int k = bk * 3 + h;
            if (cnt[k] > 0 && cnt[k] < min_chains) {
                int up = -1;
                for (int b2 = bk + 1; b2 < 6; ++b2) if (cnt[b2 * 3 + h] > 0) { up = b2 * 3 + h; break; }
                if (up >= 0) { promote[k] = up; cnt[up] += cnt[k]; cnt[k] = 0; }
            }
        }
    int start[NKEYS], acc = 0;
    ranges.clear();
    for (int k = 0; k < NKEYS; ++k) {
        start[k] = acc;
        if (cnt[k]) ranges.push_back(FastRange{acc, cnt[k], (k / 3) * 8, (k % 3) == 1, k, (k % 3) == 2});
        acc += cnt[k];
    }
    for (int b = 0; b < B; ++b) order[start[key(b)]++] = b;
}

// ---- 1-D path, K time steps per launch (blhip_fused1d.hpp) ---------------------------------------------------------------
template <int OM>
void launch_fused1d_om(hipStream_t s, const bl1f::F1Params &P, bool bwd, size_t lds) {
    const dim3 grid(P.nblk, P.B), block(bl1f::NT);
    if (bwd) {
        arm_kernel(reinterpret_cast<const void *>(&bl1f::fused1d_kernel<OM, true>));
        BL_LAUNCH((bl1f::fused1d_kernel<OM, true>), grid, block, lds, s, P);
    } else {
        arm_kernel(reinterpret_cast<const void *>(&bl1f::fused1d_kernel<OM, false>));
        BL_LAUNCH((bl1f::fused1d_kernel<OM, false>), grid, block, lds, s, P);
    }
}

void launch_fused1d(hipStream_t s, int om, const bl1f::F1Params &P, bool bwd, size_t lds) {
    switch (om) {
        case BLHIP_OM_POISSON: launch_fused1d_om<OM_POISSON>(s, P, bwd, lds); break;
        case BLHIP_OM_GAUSSIAN_MEAN: launch_fused1d_om<OM_GAUSSIAN_MEAN>(s, P, bwd, lds); break;
        case BLHIP_OM_TABLE: launch_fused1d_om<OM_TABLE>(s, P, bwd, lds); break;
        default: fail("fused 1-D path: observation model %d", om);
    }
    HIPCHECK(hipGetLastError());
}

// ---- 1-D grids, batches of chains: one block per chain, all T steps in one launch (blhip_chain1d.hpp) -------------------------------
template <int OM, int M>
void launch_chain1d_m(hipStream_t s, const bl1f::F1Params &P, bool bwd, size_t lds) {
    if (bwd) {
        arm_kernel(reinterpret_cast<const void *>(&bl1c::chain1d_kernel<OM, true, M>));
        BL_LAUNCH((bl1c::chain1d_kernel<OM, true, M>), dim3((unsigned)P.B), dim3(bl1c::NT), lds, s, P);
    } else {
        arm_kernel(reinterpret_cast<const void *>(&bl1c::chain1d_kernel<OM, false, M>));
        BL_LAUNCH((bl1c::chain1d_kernel<OM, false, M>), dim3((unsigned)P.B), dim3(bl1c::NT), lds, s, P);
    }
}
template <int OM, int CL, int M = 1>
void launch_chain1d_cl(hipStream_t s, const bl1f::F1Params &P, bool bwd, size_t lds) {
    if (bwd) {
        arm_kernel(reinterpret_cast<const void *>(&bl1c::chain1d_kernel<OM, true, M, CL>));
        BL_LAUNCH((bl1c::chain1d_kernel<OM, true, M, CL>), dim3((unsigned)P.B), dim3(bl1c::NT), lds, s, P);
    } else {
        arm_kernel(reinterpret_cast<const void *>(&bl1c::chain1d_kernel<OM, false, M, CL>));
        BL_LAUNCH((bl1c::chain1d_kernel<OM, false, M, CL>), dim3((unsigned)P.B), dim3(bl1c::NT), lds, s, P);
    }
}
// programs with Deterministic steps (CL 1) / with RegimeSwitch, NotEqual clamps too (CL 2; without a Deterministic step: rows longer than a
// block with two cells per thread, as the plain flavour)
template <int OM>
void launch_chain1d_shift(hipStream_t s, const bl1f::F1Params &P, bool bwd, size_t lds, int m) {
    if (!P.limit) launch_chain1d_cl<OM, 1>(s, P, bwd, lds);
    else if (m == 2 && P.no_shift) launch_chain1d_cl<OM, 2, 2>(s, P, bwd, lds);
    else launch_chain1d_cl<OM, 2>(s, P, bwd, lds);
}
template <int OM>
void launch_chain1d_om(hipStream_t s, const bl1f::F1Params &P, bool bwd, size_t lds, int m) {
    if (P.cmode) launch_chain1d_shift<OM>(s, P, bwd, lds, m);
    else if (m == 2) launch_chain1d_m<OM, 2>(s, P, bwd, lds);
    else launch_chain1d_m<OM, 1>(s, P, bwd, lds);
}

// the (T, n) likelihood table every chain of a 1-D batch shares (bl1c::lik1d_table_kernel: the in-kernel function, evaluated once)
void build_lik1d_table(hipStream_t s, int om, const bl1f::F1Params &P, double *out) {
    const dim3 grid((unsigned)std::min(16, (P.n + 255) / 256), (unsigned)P.T);
    if (om == BLHIP_OM_POISSON) BL_LAUNCH((bl1c::lik1d_table_kernel<OM_POISSON>), grid, dim3(256), 0, s, P, out);
    else if (om == BLHIP_OM_GAUSSIAN_MEAN) BL_LAUNCH((bl1c::lik1d_table_kernel<OM_GAUSSIAN_MEAN>), grid, dim3(256), 0, s, P, out);
    else fail("internal: shared 1-D likelihood table for observation model %d", om);
    HIPCHECK(hipGetLastError());
}

// cells per thread: 2 adjacent ones (sharing their stencil operands) for rows longer than a block, else 1 (option chain1d_pair: 0 / 1 force)
void launch_chain1d(hipStream_t s, int om, const bl1f::F1Params &P, bool bwd, int pair_mode) {
    const size_t lds = bl1c::lds_doubles(P.n, P.LW, P.cmode != nullptr) * sizeof(double);
    const int m = pair_mode == 0 ? 1 : ((pair_mode == 1 || P.n > bl1c::NT) ? 2 : 1);
    switch (om) {
        case BLHIP_OM_POISSON: launch_chain1d_om<OM_POISSON>(s, P, bwd, lds, m); break;
        case BLHIP_OM_GAUSSIAN_MEAN: launch_chain1d_om<OM_GAUSSIAN_MEAN>(s, P, bwd, lds, m); break;
        case BLHIP_OM_TABLE: launch_chain1d_om<OM_TABLE>(s, P, bwd, lds, m); break;
        default: fail("chain-resident 1-D path: observation model %d", om);
    }
    HIPCHECK(hipGetLastError());
}

template <int OM>
void launch_persist1d_om(hipStream_t s, const bl1p::P1Params &P, bool bwd, size_t lds) {
    const dim3 grid(P.nblk, P.B), block(bl1p::NT);
    if (bwd) {
        arm_kernel(reinterpret_cast<const void *>(&bl1p::persist1d_kernel<OM, true>));
        BL_LAUNCH((bl1p::persist1d_kernel<OM, true>), grid, block, lds, s, P);
    } else {
        arm_kernel(reinterpret_cast<const void *>(&bl1p::persist1d_kernel<OM, false>));
        BL_LAUNCH((bl1p::persist1d_kernel<OM, false>), grid, block, lds, s, P);
    }
}

void launch_persist1d(hipStream_t s, int om, const bl1p::P1Params &P, bool bwd, size_t lds) {
    switch (om) {
        case BLHIP_OM_POISSON: launch_persist1d_om<OM_POISSON>(s, P, bwd, lds); break;
        case BLHIP_OM_GAUSSIAN_MEAN: launch_persist1d_om<OM_GAUSSIAN_MEAN>(s, P, bwd, lds); break;
        case BLHIP_OM_TABLE: launch_persist1d_om<OM_TABLE>(s, P, bwd, lds); break;
        default: fail("persistent 1-D path: observation model %d", om);
    }
    HIPCHECK(hipGetLastError());
}

// ---- time-resident path (blhip_resident.hpp): one launch for all time steps of a single-chain 2-D fit ----------------------------
struct ResidentPlan {
    int TR = 0, TC = 0, SEG = 0, tr = 0, tc = 0, ntiles = 0, NT = 0;
    bool pad = false;            // the grid does not fill its last tile row / column (PAD kernels)
    size_t lds_bytes = 0;
};

template <int TR, int TC, int SEG, int CHK, bool BWD, int MODE, bool PAD = false, bool TAB = false>
void launch_resident_k(hipStream_t s, const blr::ResParams &Q) {
    const size_t lds = (size_t)blr::Res<TR, TC, SEG, CHK, BWD, MODE, PAD, TAB>::LDS_DOUBLES * sizeof(double);
    arm_kernel(reinterpret_cast<const void *>(&blr::resident_kernel<TR, TC, SEG, CHK, BWD, MODE, PAD, TAB>));
    BL_LAUNCH((blr::resident_kernel<TR, TC, SEG, CHK, BWD, MODE, PAD, TAB>), dim3(Q.ntiles), dim3(TR * TC / SEG), lds, s, Q);
}

// tabulated likelihood (blr::Res TAB; the one-chunk shapes): backward, evidence-only forward, every other forward pass (flags at run time)
template <int TR, int TC, int SEG, int CHK>
void launch_resident_tab(hipStream_t s, const blr::ResParams &Q, bool bwd, bool pad) {
    const bool evid = !bwd && !Q.store && !Q.means && !Q.normalise && !Q.post;
    if (pad) {
        if (bwd) launch_resident_k<TR, TC, SEG, CHK, true, 0, true, true>(s, Q);
        else if (evid) launch_resident_k<TR, TC, SEG, CHK, false, 1, true, true>(s, Q);
        else launch_resident_k<TR, TC, SEG, CHK, false, 0, true, true>(s, Q);
    } else {
        if (bwd) launch_resident_k<TR, TC, SEG, CHK, true, 0, false, true>(s, Q);
        else if (evid) launch_resident_k<TR, TC, SEG, CHK, false, 1, false, true>(s, Q);
        else launch_resident_k<TR, TC, SEG, CHK, false, 0, false, true>(s, Q);
    }
}

template <int TR, int TC, int SEG, int CHK>
void launch_resident_t(hipStream_t s, const blr::ResParams &Q, bool bwd, bool pad = false) {
    // forward pass of an evidence-only fit (nothing stored, no means, no rows to normalise) / of a full fit (every state stored, no
    // means, no rows to normalise): / of a forward-only fit: the flavours with compile-time flags (blr::Res MODE 1 / 2 / 3); padded grids: flags at run time
    const bool evid = !bwd && !Q.store && !Q.means && !Q.normalise && !Q.post;
    const bool fullfwd = !bwd && Q.store && !Q.means && !Q.normalise && Q.post;
    const bool fwdonly = !bwd && Q.store && Q.means && Q.normalise && Q.post;
    if (pad) {                   // grids that do not fill their last tile row / column
        if (bwd) {
            // (full fits of padded 128 x 128 grids keep the launch-per-step kernels -- ResidentRun::setup: the kernel spilled 231 registers
            //  and lost to them; it is not instantiated any more, round 6)
            if constexpr (SEG != CHK) fail("internal: time-resident backward launch on a padded grid of 128 x 128 tiles");
            else launch_resident_k<TR, TC, SEG, CHK, true, 0, true>(s, Q);
        }
        else if (evid) launch_resident_k<TR, TC, SEG, CHK, false, 1, true>(s, Q);
        else launch_resident_k<TR, TC, SEG, CHK, false, 0, true>(s, Q);
        return;
    }
    if (bwd) launch_resident_k<TR, TC, SEG, CHK, true, 0>(s, Q);
    else if (evid) launch_resident_k<TR, TC, SEG, CHK, false, 1>(s, Q);
    else if (fullfwd) launch_resident_k<TR, TC, SEG, CHK, false, 2>(s, Q);
    else if (fwdonly) {
        // (the multi-chunk shape spills 48 VGPRs with the compile-time flavour, 6 without: it keeps the flags at run time)
        if constexpr (SEG == CHK) launch_resident_k<TR, TC, SEG, CHK, false, 3>(s, Q);
        else launch_resident_k<TR, TC, SEG, CHK, false, 0>(s, Q);
    }
    else fail("internal: time-resident forward launch that is neither evidence-only, nor storing, nor forward-only");      // (the run-time flavour of the one-chunk shapes had no caller: pruned in round 6)
}

// tile shapes: {rows, columns, segment length, outputs per chunk}.  One wave issues an fp64 instruction only every ~12 cycles
// (tools/ubench/fp64_banks.hip), so more waves per SIMD help.  The 128 x 128 tile runs with 512 threads (segments of 32, chunks of 8,
// ~200 registers, 2 waves per SIMD).  (A 1024-thread shape -- option resident_threads128 -- was never selected by a test or a workload
// and measured no faster: pruned in round 5, profiles/r05_kernel_census.txt.)
void launch_resident(hipStream_t s, const ResidentPlan &rp, const blr::ResParams &Q, bool bwd) {
    if (Q.lik) {
        if (rp.TR == 64) launch_resident_tab<64, 64, 8, 8>(s, Q, bwd, rp.pad);
        else if (rp.TR == 32 && rp.TC == 64) launch_resident_tab<32, 64, 8, 8>(s, Q, bwd, rp.pad);
        else if (rp.TR == 32) launch_resident_tab<32, 32, 8, 8>(s, Q, bwd, rp.pad);
        else fail("internal: time-resident launch with a likelihood table on a %d x %d tile", rp.TR, rp.TC);
        HIPCHECK(hipGetLastError());
        return;
    }
    if (rp.TR == 128) launch_resident_t<128, 128, 32, 8>(s, Q, bwd, rp.pad);
    else if (rp.TR == 64) launch_resident_t<64, 64, 8, 8>(s, Q, bwd, rp.pad);
    else if (rp.TC == 64) launch_resident_t<32, 64, 8, 8>(s, Q, bwd, rp.pad);
    else launch_resident_t<32, 32, 8, 8>(s, Q, bwd, rp.pad);
    HIPCHECK(hipGetLastError());
}

// ---- chain-resident kernels (blhip_chainres.hpp): compiled as slices of blhip_chain_tu.hip (blhip_chain_launch.hpp) -----------------------
void launch_chain(hipStream_t s, const blc::ChainParams &Q, int nk, int ntw, bool bwd, bool store, bool pad = false) {
    if (Q.lik) {                         // tabulated likelihood (blc::chain_kernel TAB): geometries of <= 512 rows, radius <= 40
        if (nk > 24 || ntw > 4) fail("internal: chain-resident launch with a likelihood table outside its envelope");
        if (pad) { if (ntw >= 3) fail("internal: padded chain-resident launch with a likelihood table on %d tiles per wave", ntw); else blcl::chain_ntw12_tab_pad(s, Q, nk, ntw, bwd, store); }
        else if (ntw >= 3) blcl::chain_ntw34_tab(s, Q, nk, ntw, bwd, store); else blcl::chain_ntw12_tab(s, Q, nk, ntw, bwd, store);
        HIPCHECK(hipGetLastError());
        return;
    }
    const bool wide = nk > 24;           // bands beyond radius 40 (NK = 26 .. 44): slices of their own
    if (ntw == 4) {
        if (wide) { if (bwd) blcl::chain_ntw4_bwd_wide(s, Q, nk, store, pad); else blcl::chain_ntw4_fwd_wide(s, Q, nk, store, pad); }
        else { if (bwd) blcl::chain_ntw4_bwd(s, Q, nk, store, pad); else blcl::chain_ntw4_fwd(s, Q, nk, store, pad); }
    } else if (ntw == 8) {               // 1024 rows: one copy of the strip in LDS (blc::chain_kernel TALL)
        if (wide) { if (bwd) blcl::chain_ntw8_bwd_wide(s, Q, nk, store, pad); else blcl::chain_ntw8_fwd_wide(s, Q, nk, store, pad); }
        else { if (bwd) blcl::chain_ntw8_bwd_narrow(s, Q, nk, store, pad); else blcl::chain_ntw8_fwd_narrow(s, Q, nk, store, pad); }
    } else if (ntw == 3) {
        if (wide) blcl::chain_ntw3_wide(s, Q, nk, bwd, store, pad); else blcl::chain_ntw3(s, Q, nk, bwd, store, pad);
    } else if (ntw == 2 || ntw == 1) {
        if (wide) blcl::chain_ntw12_wide(s, Q, nk, ntw, bwd, store, pad);
        else if (ntw == 2) blcl::chain_ntw2(s, Q, nk, bwd, store, pad);
        else blcl::chain_ntw1(s, Q, nk, bwd, store, pad);
    } else fail("internal: chain-resident kernel with %d tiles per wave", ntw);
    HIPCHECK(hipGetLastError());
}

// walks on both parameters (blc::chainax_kernel, blhip_chainax.hpp)
void launch_chainax(hipStream_t s, const blc::ChainParams &Q, int nk, int ntw, bool bwd, bool store, bool pad) {
    if (ntw == 4) { if (pad) blcl::chainax_ntw4_pad(s, Q, nk, bwd, store); else blcl::chainax_ntw4(s, Q, nk, bwd, store); }
    else blcl::chainax_ntw12_pad(s, Q, nk, ntw, bwd, store);      // (these kernels take exact 128 / 256 grids too)
    HIPCHECK(hipGetLastError());
}

// backward pass with the fused fold, two chains per block (blc::chain_fold2_kernel)
bool fold2_shape(int ntw) { return ntw >= 1 && ntw <= 4; }

void launch_fold2(hipStream_t s, const blc::ChainParams &Q, int nk, int ntw, bool pad = false) {
    if (nk > 24) { if (ntw >= 3) blcl::fold2_ntw34_wide(s, Q, nk, ntw, pad); else blcl::fold2_ntw12_wide(s, Q, nk, ntw, pad); }
    else if (ntw >= 3) blcl::fold2_ntw34(s, Q, nk, ntw, pad);
    else blcl::fold2_ntw12(s, Q, nk, ntw, pad);
    HIPCHECK(hipGetLastError());
}

// the smallest supported tile whose tile grid fits the chip (every tile = one co-resident block)
// (Two 256-thread blocks per CU -- 512 tiles of 32 x 64 for the 1024^2 grid, so that one block computes while the other waits for a
// strip -- was tried: the 512 blocks were not all co-resident, the hand-off waits timed out and the fit fell back.  One tile per CU.)
// A grid that does not fill its last tile row / column runs the PAD kernels (blhip_resident.hpp): the remainder of such an axis and the
// padding behind it must both be at least one stencil radius (the mirror image beyond the true edge lives inside the last tile and is
// made of that tile's own cells).  Grids whose sizes are multiples of a tile shape are preferred (no masks).
bool plan_resident(int n0, int n1, int cus, ResidentPlan &rp) {
    constexpr int seg128 = 32, min_tile = 32;
    constexpr bool allow_pad = true;
    // preference: 64 x 64 tiles first (measured on 128^2 .. 512^2 grids, tools/tile_probe.py: 5.7 / 6.7 us per forward / backward step
    // against 9.0 / 10.1 us with 32 x 32 tiles -- two waves per block are too few to hide the hand-offs -- and 10.8 / 20.8 us with 128 x 128),
    // whole tiles before a padded last tile row / column of the same shape; 128 x 128 only when the smaller shapes need more than one tile per CU
    const int shapes[4][3] = {{64, 64, 8}, {32, 64, 8}, {32, 32, 8}, {128, 128, seg128}};
    for (const auto &sh : shapes)
        for (int pass = 0; pass < (allow_pad ? 2 : 1); ++pass) {
            auto fits = [&](int n, int t) { const int rem = n % t; return pass == 0 ? rem == 0 : (rem == 0 || (n > t && rem >= blr::R && t - rem >= blr::R)); };
            if (sh[0] < min_tile && sh[1] < 2 * min_tile) continue;
            if (!fits(n0, sh[0]) || !fits(n1, sh[1])) continue;
            const int tr = (n0 + sh[0] - 1) / sh[0], tc = (n1 + sh[1] - 1) / sh[1];
            const long long nt = (long long)tr * tc;
            if (nt > cus) continue;
            rp.TR = sh[0]; rp.TC = sh[1]; rp.SEG = sh[2]; rp.tr = tr; rp.tc = tc; rp.ntiles = (int)nt;
            rp.NT = sh[0] * sh[1] / sh[2];
            rp.pad = (n0 % sh[0]) != 0 || (n1 % sh[1]) != 0;
            return true;
        }
    return false;
}
