// libblhip: C-ABI (include/blhip.h) + host orchestration of the forward-backward recursion on one MI355X.
//
// The host side owns what the reference's Study.fit loop owns (bayesloop/core.py:330-486): the order of steps, the
// evidence bookkeeping and the per-step transition program; the arithmetic on the grid runs in the HIP kernels of
// blhip_kernels.hpp / blhip_fast.hpp.
#include <hip/hip_runtime.h>

#include <cxxabi.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <exception>
#include <map>
#include <memory>
#include <unordered_map>
#include <mutex>
#include <set>
#include <string>
#include <system_error>
#include <chrono>
#include <thread>
#include <vector>

#include "../../include/blhip.h"
#include "blhip_kernels.hpp"
#include "blhip_fast.hpp"
#include "blhip_mfma.hpp"
#include "blhip_hwide.hpp"
#include "blhip_fused1d.hpp"
#include "blhip_chain1d.hpp"
#include "blhip_persist1d.hpp"
#include "blhip_resident.hpp"
#include "blhip_chainres.hpp"
#include "blhip_chain_launch.hpp"
#include "blhip_nd.hpp"

using namespace blk;

#include "blhip_host.hpp"
#include "blhip_comm.hpp"

namespace {

struct TapTable {
    std::vector<double> w;       // concatenated half kernels: w[off + k], k = 0..lw
    std::vector<int> off, lw, lw2;   // lw2: axis-1 radius of a dense 2-D kernel (0 for 1-D tap sets)
    std::map<std::pair<int, double>, int> index;   // (internal axis, normed sigma) -> id
    std::map<std::tuple<double, double, double>, int> index2;   // dense kernels: (ns1, ns2, rho) -> id

    // Deterministic (transitionModels.py:581, :600): scipy.ndimage.shift(order=3, mode='nearest') by d grid cells = stencil
    // out[i] = sum_m K[m] ext[i + m], K[m] = eta(-d - m) with the cardinal cubic spline eta(u) = sum_n sqrt(3) pole^|n| beta3(u - n)
    // (prefilter impulse response x B-spline), |m| <= ceil|d| + 34, over the extension blk::extend_index(rule 2) builds.
    // Exact for |d| <= 12 (beyond, SciPy extends the COEFFICIENTS by their edge values: oracle/bl_oracle.py).  Stored with all
    // 2 lw + 1 weights; lw2 = -1 marks the asymmetric layout.
    struct PairHash { size_t operator()(const std::pair<int, double> &k) const { unsigned long long b; std::memcpy(&b, &k.second, 8); return (size_t)((b * 0x9E3779B97F4A7C15ull) >> 7) ^ (size_t)k.first; } };
    struct DblHash { size_t operator()(double d) const { unsigned long long b; std::memcpy(&b, &d, 8); return (size_t)((b * 0x9E3779B97F4A7C15ull) >> 7); } };
    std::unordered_map<std::pair<int, double>, int, PairHash> index_shift;      // (looked up once per step and chain inside a Deterministic segment)
    int get_shift(int axis, double d) {
        auto key = std::make_pair(axis, d);
        auto it = index_shift.find(key);
        if (it != index_shift.end()) return it->second;
        const double pole = std::sqrt(3.0) - 2.0, gain = -6.0 * pole / (1.0 - pole * pole);
        const int r = (int)std::ceil(std::fabs(d)) + 34;
        const int id = (int)off.size();
        off.push_back((int)w.size());
        lw.push_back(r);
        lw2.push_back(-1);
        for (int m = -r; m <= r; ++m) {
            const double u = -d - (double)m, n0 = std::floor(u);
            double eta = 0.0;
            for (int k = -1; k <= 2; ++k) {
                const double n = n0 + k, a = std::fabs(u - n);
                const double b3 = a < 1.0 ? 2.0 / 3.0 - a * a + a * a * a / 2.0 : (a < 2.0 ? (2.0 - a) * (2.0 - a) * (2.0 - a) / 6.0 : 0.0);
                eta += gain * std::pow(pole, std::fabs(n)) * b3;
            }
            w.push_back(eta);
        }
        index_shift[key] = id;
        return id;
    }

    // The same shift for |d| > 12 grid cells per step (1-D grids; the reference's published break-point study shifts by up to 334 cells
    // per step: docs/source/tutorials/changepointstudy.ipynb).  Beyond its 12 pre-padded samples SciPy extends the spline COEFFICIENTS by
    // their edge values, which no shift-invariant stencil reproduces: the step kernel then works in the two stages of
    // oracle/bl_oracle.py: spline_shift_nearest -- prefilter the padded row (symmetric, radius 34, weights below), then evaluate the cubic
    // B-spline at the shifted coordinates with the coefficient index clamped.  Stored as [d, g(0) .. g(34)]; lw = 12 + 34 is the halo the
    // row needs, lw2 = -2 marks the layout.
    std::unordered_map<double, int, DblHash> index_bigshift;
    int get_bigshift(double d) {
        auto it = index_bigshift.find(d);
        if (it != index_bigshift.end()) return it->second;
        const double pole = std::sqrt(3.0) - 2.0, gain = -6.0 * pole / (1.0 - pole * pole);
        const int id = (int)off.size();
        off.push_back((int)w.size());
        lw.push_back(12 + 34);
        lw2.push_back(-2);
        w.push_back(d);
        for (int m = 0; m <= 34; ++m) w.push_back(gain * std::pow(pole, (double)m));
        index_bigshift[d] = id;
        return id;
    }

    // AlphaStableRandomWalk.createKernel (transitionModels.py:196-240) for an axis of n points: k[d], d = 0 .. n-1, of the
    // inverse real DFT (numpy.fft.irfft) of exp(-|c w|^alpha) sampled at m = int(3n/2 + 1) points of [0, pi]; the reference's
    // roll + 3x zero padding + fftconvolve(mode='same') (:233-260) is out[i] = sum_j in[j] k[|i - j|] inside the grid
    std::map<std::tuple<int, double, double, int>, int> index_as;
    int get_alphastable(int axis, double c, double alpha, int n) {
        auto key = std::make_tuple(axis, c, alpha, n);
        auto it = index_as.find(key);
        if (it != index_as.end()) return it->second;
        const int m = (int)(3.0 * n / 2.0 + 1.0), K = 2 * (m - 1);
        std::vector<double> X(m);
        for (int q = 0; q < m; ++q) X[q] = std::exp(-std::pow(std::fabs(c * (M_PI * q / (m - 1))), alpha));
        const int id = (int)off.size();
        off.push_back((int)w.size());
        lw.push_back(n - 1);
        lw2.push_back(0);
        for (int j = 0; j < n; ++j) {
            long double acc = X[0] + ((j & 1) ? -X[m - 1] : X[m - 1]);
            for (int q = 1; q < m - 1; ++q)
                acc += 2.0L * X[q] * std::cos(2.0L * (long double)M_PIl * (long double)(((long long)j * q) % K) / (long double)K);
            w.push_back((double)(acc / K));
        }
        index_as[key] = id;
        return id;
    }

    // BivariateRandomWalk.createKernel (transitionModels.py:898-911): bivariate normal density on the integer lattice
    // |x| <= 3 ceil(ns1), |y| <= 3 ceil(ns2), normalised to sum 1 (the density's own constant cancels); row-major
    int get2d(double ns1, double ns2, double rho) {
        auto key = std::make_tuple(ns1, ns2, rho);
        auto it = index2.find(key);
        if (it != index2.end()) return it->second;
        const int r0 = 3 * (int)std::ceil(ns1), r1 = 3 * (int)std::ceil(ns2);
        std::vector<double> k((size_t)(2 * r0 + 1) * (2 * r1 + 1));
        double sum = 0.0;
        for (int a = -r0; a <= r0; ++a)
            for (int b = -r1; b <= r1; ++b) {
                const double x = a, y = b;
                const double q = (x * x / (ns1 * ns1) - 2.0 * rho * x * y / (ns1 * ns2) + y * y / (ns2 * ns2)) / (2.0 * (1.0 - rho * rho));
                const double v = std::exp(-q);
                k[(size_t)(a + r0) * (2 * r1 + 1) + (b + r1)] = v;
                sum += v;
            }
        const int id = (int)off.size();
        off.push_back((int)w.size());
        lw.push_back(r0);
        lw2.push_back(r1);
        for (double v : k) w.push_back(v / sum);
        index2[key] = id;
        return id;
    }

    // SciPy's kernel: lw = int(4 sd + 0.5); phi = exp(-0.5/sd^2 x^2); phi / sum(phi)   (_filters.py, gaussian_filter1d)
    int get(int axis, double ns) {
        auto key = std::make_pair(axis, ns);
        auto it = index.find(key);
        if (it != index.end()) return it->second;
        const int r = (int)(4.0 * ns + 0.5);
        int id = -1;
        if (r > 0) {
            std::vector<double> phi(2 * r + 1);
            const double s2 = ns * ns;
            double sum = 0.0;
            for (int k = -r; k <= r; ++k) {
                phi[k + r] = std::exp(-0.5 / s2 * (double)(k * k));
                sum += phi[k + r];
            }
            id = (int)off.size();
            off.push_back((int)w.size());
            lw.push_back(r);
            lw2.push_back(0);
            for (int k = 0; k <= r; ++k) w.push_back(phi[r + k] / sum);
        }
        index[key] = id;
        return id;
    }
};

struct Geometry {
    int n0, n1;          // internal rows / cols
    int axis_map[2];     // ABI parameter index -> internal axis
    long long G;
};

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

void validate(const blhip_problem *p, int64_t n_chains, const double *op_values) {
    if (!p) fail("problem is NULL");
    if (p->ndim < 1 || p->ndim > BLHIP_MAX_DIM) fail("ndim must be 1 .. %d (got %d)", BLHIP_MAX_DIM, p->ndim);
    if (p->ndim > 2) {            // the plain N-D path (blhip_nd.hpp)
        if (p->obs_model != BLHIP_OM_TABLE) fail("grids with %d parameters need a caller-evaluated likelihood table (BLHIP_OM_TABLE)", p->ndim);
        for (int k = 0; k < p->n_ops; ++k) {
            const blhip_op &op = p->ops[k];
            const bool ok = op.kind == BLHIP_OP_GRW || op.kind == BLHIP_OP_STATIC || (op.kind == BLHIP_OP_CHANGEPOINT && !(op.flags & 1));
            if (!ok || op.segment >= 0)
                fail("grids with %d parameters support GaussianRandomWalk / Static / ChangePoint transition models (op %d has kind %d)", p->ndim, k, op.kind);
        }
    }
    for (int k = 0; k < p->ndim; ++k) {
        if (p->n[k] < 1) fail("grid size n[%d] = %lld", k, (long long)p->n[k]);
        if (!p->marginal[k]) fail("marginal[%d] is NULL", k);
        if (p->n[k] > (1ll << 30)) fail("grid axis too long");
    }
    if (p->T < 1) fail("T must be >= 1");
    if (!p->data || !p->timestamps || !p->prior) fail("data / timestamps / prior must not be NULL");
    if (n_chains < 1) fail("n_chains must be >= 1");
    if (p->n_ops < 0 || (p->n_ops > 0 && !p->ops)) fail("bad transition program");
    if (p->n_ops > 0 && !op_values) fail("op_values is NULL");
    bool has_cp = false;
    for (int k = 0; k < p->n_ops; ++k) {
        const blhip_op &op = p->ops[k];
        if (op.kind == BLHIP_OP_GRW) {
            if (op.axis < 0 || op.axis >= p->ndim) fail("GRW op %d: axis %d out of range", k, op.axis);
        } else if (op.kind == BLHIP_OP_CHANGEPOINT) {
            has_cp = true;
        } else if (op.kind == BLHIP_OP_INDEPENDENT) {
            if (!p->indep_prior) fail("INDEPENDENT op needs indep_prior");
        } else if (op.kind == BLHIP_OP_DETERMINISTIC) {
            if (op.axis < 0 || op.axis >= p->ndim) fail("DETERMINISTIC op %d: axis %d out of range", k, op.axis);
            for (int64_t q = 1; q <= 2 * p->T; ++q)
                if (k + q >= p->n_ops || p->ops[k + q].kind != BLHIP_OP_DETERMINISTIC_ARG)
                    fail("DETERMINISTIC op %d must be followed by 2 T = %lld DETERMINISTIC_ARG ops (the shifts per step)", k, (long long)(2 * p->T));
        } else if (op.kind == BLHIP_OP_ALPHASTABLE) {
            if (op.axis < 0 || op.axis >= p->ndim) fail("ALPHASTABLE op %d: axis %d out of range", k, op.axis);
            if (k + 1 >= p->n_ops || p->ops[k + 1].kind != BLHIP_OP_ALPHASTABLE_ARG)
                fail("ALPHASTABLE op %d must be followed by an ALPHASTABLE_ARG op (alpha)", k);
        } else if (op.kind == BLHIP_OP_BIVARIATE) {
            if (p->ndim != 2) fail("BIVARIATE op %d needs a 2-parameter grid", k);
            if (k + 2 >= p->n_ops || p->ops[k + 1].kind != BLHIP_OP_BIVARIATE_ARG || p->ops[k + 2].kind != BLHIP_OP_BIVARIATE_ARG)
                fail("BIVARIATE op %d must be followed by two BIVARIATE_ARG ops (sigma2, rho)", k);
        } else if (op.kind != BLHIP_OP_STATIC && op.kind != BLHIP_OP_REGIMESWITCH && op.kind != BLHIP_OP_BREAKPOINT &&
                   op.kind != BLHIP_OP_NOTEQUAL && op.kind != BLHIP_OP_BIVARIATE_ARG && op.kind != BLHIP_OP_ALPHASTABLE_ARG &&
                   op.kind != BLHIP_OP_DETERMINISTIC_ARG) {
            fail("op %d: unknown kind %d", k, op.kind);
        }
    }
    if (has_cp && !p->reset_prior) fail("CHANGEPOINT op needs reset_prior");
    switch (p->obs_model) {
        case BLHIP_OM_POISSON:
            if (p->ndim != 1) fail("Poisson model has 1 parameter");
            if (p->seg_len != 1) fail("Poisson model has segment length 1");
            break;
        case BLHIP_OM_GAUSSIAN:
            if (p->ndim != 2) fail("Gaussian model has 2 parameters");
            if (p->seg_len != 1) fail("Gaussian model has segment length 1");
            break;
        case BLHIP_OM_GAUSSIAN_MEAN:
            if (p->ndim != 1) fail("GaussianMean model has 1 parameter");
            if (p->seg_len != 1 || p->data_dim != 2) fail("GaussianMean data must be (T, 1, 2)");
            break;
        case BLHIP_OM_TABLE:
            if (!p->lik) fail("BLHIP_OM_TABLE needs lik");
            break;
        case BLHIP_OM_BERNOULLI: case BLHIP_OM_WHITE_NOISE:
            if (p->ndim != 1 || p->seg_len != 1) fail("Bernoulli / white-noise models have 1 parameter and segment length 1");
            break;
        case BLHIP_OM_LAPLACE:
            if (p->ndim != 2 || p->seg_len != 1) fail("Laplace model has 2 parameters and segment length 1");
            break;
        case BLHIP_OM_AR1: case BLHIP_OM_SCALED_AR1:
            if (p->ndim != 2 || p->seg_len != 2) fail("AR1 models have 2 parameters and segment length 2");
            break;
        default: fail("unknown observation model %d", p->obs_model);
    }
    if (p->data_dim < 1) fail("data_dim must be >= 1");
}

// per-step records consumed by blk::likelihood<>
void build_records(const blhip_problem *p, std::vector<double> &rec, int &rec_len, int &d) {
    const int64_t T = p->T;
    const int dd = p->data_dim;
    if (p->obs_model == BLHIP_OM_GAUSSIAN) {
        d = dd; rec_len = dd;
        rec.assign(p->data, p->data + T * dd);
    } else if (p->obs_model == BLHIP_OM_GAUSSIAN_MEAN) {
        d = 1; rec_len = 3;
        rec.resize(T * 3);
        for (int64_t t = 0; t < T; ++t) {
            const double x = p->data[t * 2], s = p->data[t * 2 + 1];
            const bool miss = std::isnan(x) || std::isnan(s);
            rec[t * 3 + 0] = miss ? std::numeric_limits<double>::quiet_NaN() : x;
            rec[t * 3 + 1] = 1.0 / (2.0 * s * s);
            rec[t * 3 + 2] = 0.5 * std::log(2.0 * M_PI * s * s);
        }
    } else if (p->obs_model == BLHIP_OM_POISSON) {
        d = dd; rec_len = 2 * dd;
        rec.resize(T * 2 * dd);
        for (int64_t t = 0; t < T; ++t)
            for (int k = 0; k < dd; ++k) {
                const double c = p->data[t * dd + k];
                double f = 1.0;
                if (!std::isnan(c)) {
                    if (c < 0 || c != std::floor(c)) fail("Poisson data must be non-negative integers (step %lld)", (long long)t);
                    for (double q = 2.0; q <= c; q += 1.0) f *= q;
                }
                rec[(t * dd + k) * 2] = c;
                rec[(t * dd + k) * 2 + 1] = f;
            }
    } else {
        d = 1; rec_len = 1;
        rec.assign(T, 0.0);
    }
}

struct ChainProgram {
    // per (step, chain): source kind, tap ids per internal axis, clamp mode/limit (RegimeSwitch); forward and backward
    std::vector<unsigned char> kindF, kindB, cmodeF, cmodeB;
    std::vector<int> tapF0, tapF1, tapB0, tapB1;
    std::vector<double> limitF, limitB;
    int LW0 = 0, LW1 = 0;
    bool has_clamp = false;
    bool whole_row = false;      // a two-stage spline shift (Deterministic, |d| > 12): a block needs the whole row of a 1-D grid
    bool other_clamp = false;    // has_clamp for another reason than a Deterministic model's shift (mode 6)
    bool dense_clamp = false;    // ... than a shift or the clamps of RegimeSwitch / NotEqual: AlphaStable- / BivariateRandomWalk (modes 5 / 4: zero boundary, dense kernels)
    bool has_shift = false;      // a Deterministic model
};

struct StepProg {
    unsigned char kind = SRC_PREV, cmode = 0;   // cmode: 0 none, 1 clamp the source (before the stencil), 2 clamp after it
    int t0 = -1, t1 = -1;
    double limit = 0.0;
};

void build_program(const blhip_problem *p, const Geometry &g, int64_t c0, int64_t B, const double *op_values,
                   TapTable &taps, ChainProgram &prog, bool resume) {
    const int64_t T = p->T;
    const int nops = p->n_ops;
    const size_t nT = (size_t)T * B;
    prog.kindF.assign(nT, SRC_PREV); prog.kindB.assign(nT, SRC_PREV);
    prog.cmodeF.assign(nT, 0); prog.cmodeB.assign(nT, 0);
    prog.limitF.assign(nT, 0.0); prog.limitB.assign(nT, 0.0);
    prog.tapF0.assign(nT, -1); prog.tapF1.assign(nT, -1);
    prog.tapB0.assign(nT, -1); prog.tapB1.assign(nT, -1);
    prog.LW0 = prog.LW1 = 0;
    prog.has_clamp = false;
    prog.whole_row = false;
    prog.other_clamp = false;
    prog.dense_clamp = false;
    prog.has_shift = false;
    double dV = 1.0;
    for (int k = 0; k < p->ndim; ++k) dV *= p->lattice[k];
    // the ops a step's program is made of (the *_ARG ops only carry values of the op in front of them: a Deterministic model has 2 T of
    // them, and the per-(step, chain) walk below would spend its time skipping them -- 23 400 chains x 41 steps x 2 x 87 ops measured)
    std::vector<int> real_ops;
    for (int k = 0; k < nops; ++k) {
        const int kind = p->ops[k].kind;
        if (kind != BLHIP_OP_DETERMINISTIC_ARG && kind != BLHIP_OP_BIVARIATE_ARG && kind != BLHIP_OP_ALPHASTABLE_ARG) real_ops.push_back(k);
    }
    // (a chain's T steps are T entries B apart in each of the ten arrays: written chain by chain that is one cache line per entry --
    //  half of the 30 ms this function took for the 23 400 chains x 41 steps of the published break-point study.  The steps of GROUP
    //  chains are collected first and written out as runs of GROUP consecutive entries)
    // (no more entries than the batch has chains: constructing 64 x T records for the ONE chain of a long single-chain fit -- C2: T = 10 000,
    //  2 x 15 MB -- was 8 ms of its 43-ms fit)
    constexpr int GROUP = 64;
    const size_t group_rows = (size_t)std::min<int64_t>(GROUP, std::max<int64_t>(B, 1));
    std::vector<StepProg> gF(group_rows * T), gB(group_rows * T);
    auto flush_group = [&](int64_t b0, int64_t nb) {
        for (int64_t t = 0; t < T; ++t) {
            const size_t k0 = (size_t)t * B + b0;
            for (int64_t q = 0; q < nb; ++q) {
                const StepProg &f = gF[(size_t)q * T + t], &r = gB[(size_t)q * T + t];
                prog.kindF[k0 + q] = f.kind; prog.tapF0[k0 + q] = f.t0; prog.tapF1[k0 + q] = f.t1; prog.cmodeF[k0 + q] = f.cmode; prog.limitF[k0 + q] = f.limit;
                prog.kindB[k0 + q] = r.kind; prog.tapB0[k0 + q] = r.t0; prog.tapB1[k0 + q] = r.t1; prog.cmodeB[k0 + q] = r.cmode; prog.limitB[k0 + q] = r.limit;
            }
        }
    };
    std::vector<int> op_tap(nops, -1), op_axis(nops, -1);
    // (see `at` below) the ops that bound segments or fire at a time stamp; per chain: the program of each segment, walked once
    std::vector<int> bound_ops;
    for (int k : real_ops)
        if (p->ops[k].kind == BLHIP_OP_BREAKPOINT || p->ops[k].kind == BLHIP_OP_CHANGEPOINT) bound_ops.push_back(k);
    std::vector<StepProg> seg_prog(bound_ops.size() + 1);
    std::vector<char> seg_cached(bound_ops.size() + 1, 0), det_in_seg(bound_ops.size() + 1, 0);
    for (int64_t b = 0; b < B; ++b) {
        const double *val = op_values ? op_values + (c0 + b) * nops : nullptr;
        // tap ids of this chain's GRW ops
        std::fill(op_tap.begin(), op_tap.end(), -1); std::fill(op_axis.begin(), op_axis.end(), -1);
        bool time_dependent = false;
        for (int k = 0; k < nops; ++k) {
            const blhip_op &op = p->ops[k];
            if (op.kind == BLHIP_OP_GRW) {
                const int ax = g.axis_map[op.axis];
                const double ns = val[k] / p->lattice[op.axis];            // transitionModels.py:108
                op_axis[k] = ax;
                op_tap[k] = (ns > 0.0) ? taps.get(ax, ns) : -1;            // :110-113 (sigma <= 0: copy)
                if (std::isnan(ns)) fail("chain %lld: GRW sigma is NaN", (long long)(c0 + b));
            } else if (op.kind == BLHIP_OP_CHANGEPOINT || op.kind == BLHIP_OP_BREAKPOINT) {
                time_dependent = true;
            } else if (op.kind == BLHIP_OP_REGIMESWITCH || op.kind == BLHIP_OP_NOTEQUAL) {
                prog.has_clamp = true; prog.other_clamp = true;
            } else if (op.kind == BLHIP_OP_DETERMINISTIC) {
                time_dependent = true;                       // a different shift at every step
                op_axis[k] = g.axis_map[op.axis];
                prog.has_clamp = true;                       // (mode 6 of the generic kernel)
                prog.has_shift = true;
            } else if (op.kind == BLHIP_OP_ALPHASTABLE) {
                const double c = val[k] / p->lattice[op.axis], alpha = val[k + 1];          // transitionModels.py:170-176
                if (std::isnan(c) || std::isnan(alpha)) fail("chain %lld: AlphaStableRandomWalk parameters are NaN", (long long)(c0 + b));
                op_axis[k] = g.axis_map[op.axis];
                op_tap[k] = taps.get_alphastable(op_axis[k], c, alpha, (int)p->n[op.axis]);
                prog.has_clamp = true; prog.other_clamp = true; prog.dense_clamp = true;      // (mode 5 of the generic kernel: zero boundary + renormalisation)
            } else if (op.kind == BLHIP_OP_BIVARIATE) {
                // transitionModels.py:881-885; a singular covariance makes scipy.stats.multivariate_normal raise in the reference
                const double n1 = val[k] / p->lattice[0], n2 = val[k + 1] / p->lattice[1], rho = val[k + 2];
                if (!(n1 > 0.0) || !(n2 > 0.0) || !(std::fabs(rho) < 1.0))
                    fail("chain %lld: BivariateRandomWalk needs sigma1, sigma2 > 0 and |rho| < 1", (long long)(c0 + b));
                op_tap[k] = taps.get2d(n1, n2, rho);
                prog.has_clamp = true; prog.other_clamp = true; prog.dense_clamp = true;      // (mode 4 of the generic kernel: dense kernel + renormalisation)
            }
        }
        // the transition from one step to the next, evaluated at time stamp tau (list order, transitionModels.py:645-649)
        auto run = [&](double tau, bool have_tau, int64_t step = -1, bool fwd = true) {
            StepProg sp;
            int seg = 0;                                                   // active sub-model of a serial model (:768)
            if (have_tau)
                for (int k : real_ops) {
                    const blhip_op &op = p->ops[k];
                    if ((op.kind == BLHIP_OP_BREAKPOINT || (op.kind == BLHIP_OP_CHANGEPOINT && (op.flags & 1))) && val[k] <= tau) seg++;
                }
            bool filtered = false;
            for (int k : real_ops) {
                const blhip_op &op = p->ops[k];
                if (op.segment >= 0 && op.segment != seg) continue;
                switch (op.kind) {
                    case BLHIP_OP_GRW: {
                        if (op_tap[k] < 0) break;
                        if (sp.cmode == 2) fail("a GaussianRandomWalk after a RegimeSwitch in one combined model is not supported");
                        if (sp.cmode == 4 || sp.cmode == 5)
                            fail("a GaussianRandomWalk combined with a Bivariate- / AlphaStableRandomWalk is not supported");
                        if (sp.cmode == 6 && (op_axis[k] == 0 ? sp.t0 : sp.t1) >= 0)
                            fail("a GaussianRandomWalk and a Deterministic model on the same parameter are not supported");
                        int &slot = op_axis[k] == 0 ? sp.t0 : sp.t1;
                        if (slot >= 0)
                            fail("two GaussianRandomWalk ops on the same parameter in one combined model are not supported");
                        slot = op_tap[k];
                        filtered = true;
                        break;
                    }
                    case BLHIP_OP_CHANGEPOINT:
                        if (!(op.flags & 1) && have_tau && tau == val[k]) {      // transitionModels.py:300-312
                            sp = StepProg(); sp.kind = SRC_RESET; filtered = false;
                        }
                        break;
                    case BLHIP_OP_INDEPENDENT:                                    // transitionModels.py:351-360
                        sp = StepProg(); sp.kind = SRC_INDEP; filtered = false;
                        break;
                    case BLHIP_OP_DETERMINISTIC: {                                // transitionModels.py:571-583, :585-602
                        if (step < 0) break;                                      // (the time-independent template program)
                        const double dd = val[k + 1 + (fwd ? step : T + step)] / p->lattice[op.axis];
                        if (std::isnan(dd)) fail("chain %lld: Deterministic shift of step %lld is NaN", (long long)(c0 + b), (long long)step);
                        if (std::fabs(dd) > 12.0 && (g.n0 != 1 || (double)g.n1 > 16000.0))
                            fail("chain %lld, step %lld: Deterministic model shifts by %.3g grid cells in one time step; on grids with two "
                                 "parameters (and 1-D grids beyond 16000 points) the fused kernel supports up to 12 (SciPy's pre-padding)",
                                 (long long)(c0 + b), (long long)step, dd);
                        int &slot = op_axis[k] == 0 ? sp.t0 : sp.t1;
                        if (slot >= 0 || (sp.cmode != 0 && sp.cmode != 6))
                            fail("a Deterministic model combined with another model acting on the same parameter / a clamp is not supported");
                        if (dd != 0.0) {                                          // zero shift: identity (its renormalisation is a no-op)
                            if (std::fabs(dd) > 12.0) { slot = taps.get_bigshift(dd); prog.whole_row = true; }
                            else slot = taps.get_shift(op_axis[k], dd);
                            sp.cmode = 6;
                        }
                        filtered = true;
                        break;
                    }
                    case BLHIP_OP_ALPHASTABLE: {                                  // transitionModels.py:167-187
                        if (sp.cmode != 0 || filtered)
                            fail("an AlphaStableRandomWalk combined with another model acting on the same step is not supported");
                        sp.cmode = 5;
                        (op_axis[k] == 0 ? sp.t0 : sp.t1) = op_tap[k];
                        filtered = true;
                        break;
                    }
                    case BLHIP_OP_BIVARIATE:                                      // transitionModels.py:880-891
                        if (sp.cmode != 0 || filtered)
                            fail("a BivariateRandomWalk combined with another model acting on the same step is not supported");
                        sp.cmode = 4;
                        sp.t0 = op_tap[k];
                        filtered = true;
                        break;
                    case BLHIP_OP_NOTEQUAL:                                       // transitionModels.py:462-471
                        if (sp.cmode != 0 || filtered)
                            fail("a NotEqual model after another model acting on the same step is not supported");
                        if (sp.kind != SRC_PREV) fail("a NotEqual model right after a change-point / independent restart is not supported");
                        sp.cmode = 3;
                        sp.limit = std::pow(10.0, val[k]) * dV;
                        break;
                    case BLHIP_OP_REGIMESWITCH:                                   // transitionModels.py:405-410
                        if (sp.cmode != 0) fail("two RegimeSwitch models acting at the same time are not supported");
                        sp.cmode = filtered ? 2 : 1;
                        sp.limit = std::pow(10.0, val[k]) * dV;
                        break;
                    default: break;
                }
            }
            if (have_tau)
                for (int k : real_ops) {                                          // serial change-points, :801-813
                    const blhip_op &op = p->ops[k];
                    if (op.kind == BLHIP_OP_CHANGEPOINT && (op.flags & 1) && tau == val[k]) { sp = StepProg(); sp.kind = SRC_RESET; }
                }
            if (sp.t0 >= 0) prog.LW0 = std::max(prog.LW0, taps.lw[sp.t0]);
            if (sp.cmode == 4) prog.LW1 = std::max(prog.LW1, taps.lw2[sp.t0]);
            if (sp.t1 >= 0) prog.LW1 = std::max(prog.LW1, taps.lw[sp.t1]);
            return sp;
        };
        const StepProg stat = run(0.0, false);       // the program when nothing depends on the time stamp
        // A step's program depends on its time stamp through (1) the active sub-model of a serial model = how many break- / serial
        // change-points lie at or before it, (2) a change-point AT it, (3) the step index of a Deterministic model in the active part.
        // Steps that share (1), have no (2) and no (3) share their program: it is walked once per chain and segment -- two thirds of the
        // (chain, step) pairs of the published break-point study sit in Static segments (build_program 28 -> 15 ms of a 92-ms fit).
        auto at = [&](double tau, int64_t step, bool fwd) -> StepProg {
            int seg = 0;
            bool event = false;
            for (int k : bound_ops) {
                const blhip_op &op = p->ops[k];
                const bool serial = op.kind == BLHIP_OP_BREAKPOINT || (op.flags & 1);
                if (serial && val[k] <= tau) seg++;
                if (op.kind == BLHIP_OP_CHANGEPOINT && tau == val[k]) event = true;
            }
            if (event || det_in_seg[seg]) return run(tau, true, step, fwd);
            if (!seg_cached[seg]) { seg_prog[seg] = run(tau, true, step, fwd); seg_cached[seg] = 1; }
            return seg_prog[seg];
        };
        if (time_dependent) {
            std::fill(seg_cached.begin(), seg_cached.end(), 0);
            std::fill(det_in_seg.begin(), det_in_seg.end(), 0);
            for (int k : real_ops)
                if (p->ops[k].kind == BLHIP_OP_DETERMINISTIC)
                    for (size_t sg = 0; sg < det_in_seg.size(); ++sg)
                        if (p->ops[k].segment < 0 || (size_t)p->ops[k].segment == sg) det_in_seg[sg] = 1;
        }
        for (int64_t t = 0; t < T; ++t) {
            // forward step t consumes T_fwd(post_{t-1}, ts[t-1])   core.py:411
            StepProg f; f.kind = SRC_PRIOR;
            if (t > 0) f = time_dependent ? at(p->timestamps[t - 1], t, true) : stat;
            else if (resume) f = run(p->resume_time, true, 0, true);   // continues a carried state (OnlineStudy.step, core.py:2164-2165)
            // backward step t consumes T_bwd(beta_{t+1} L_{t+1}, ts[t+1]) = T_fwd(., ts[t+1] - 1)   core.py:467, transitionModels.py:316-317
            StepProg r; r.kind = SRC_UNIFORM;
            if (t < T - 1) r = time_dependent ? at(p->timestamps[t + 1] - 1.0, t, false) : stat;
            gF[(size_t)(b % GROUP) * T + t] = f; gB[(size_t)(b % GROUP) * T + t] = r;
        }
        if (b % GROUP == GROUP - 1 || b == B - 1) flush_group(b - b % GROUP, b % GROUP + 1);
    }
}

// normalise the kept posterior rows (core.py:389 / :441) (eagerly at the end of the fit, or on first access with option lazy_normalise)
void ensure_post_scaled(blhip_ctx *ctx) {
    if (!ctx->post_valid || ctx->post_scaled) return;
    HIPCHECK(hipSetDevice(ctx->device));
    const long long G = ctx->post_G;
    const unsigned gx = (unsigned)std::min<long long>((G + NTHREADS - 1) / NTHREADS, 4096);
    // (rows [post_row0, post_row1) only: the time-resident kernel has normalised the others itself)
    const int64_t r0 = ctx->post_row0, nrows = ctx->post_row1 - ctx->post_row0;
    for (int64_t b = 0; b < ctx->post_chains && nrows > 0; ++b)
        BL_LAUNCH(scale_rows_kernel, dim3(gx, (unsigned)nrows), dim3(NTHREADS), 0, ctx->stream,
                           ctx->post.as<double>() + ((size_t)b * ctx->post_T + r0) * G, G, ctx->postinv.as<double>() + b * ctx->post_T + r0);
    HIPCHECK(hipGetLastError());
    ctx->post_scaled = true;
}

// axis-0 radius bucket of a chain as bucket_step() will see it (0: no filter, k: radius in (8 (k - 1), 8 k]); -1 if it cannot be told
// from the op values alone
int chain_bucket(const blhip_problem *p, const double *val) {
    int r0 = 0;
    for (int k = 0; k < p->n_ops; ++k) {
        const blhip_op &op = p->ops[k];
        if (op.kind == BLHIP_OP_GRW) {
            if (p->ndim == 2 && op.axis == 0) {
                const double ns = val[k] / p->lattice[0];
                if (!(ns >= 0.0) || ns > 1e6) return -1;
                r0 = std::max(r0, (int)(4.0 * ns + 0.5));
            }
        } else if (op.kind != BLHIP_OP_STATIC) {
            return -1;                               // (other models: their launches are not bucketed by radius)
        }
    }
    return r0 == 0 ? 0 : (r0 + 7) / 8;
}

// -> start index of every batch (+ n_chains at the end): equal shares of at most Bmax chains, each cut moved to the nearest change of
// radius bucket within the slack the memory budget leaves
std::vector<int64_t> plan_batches(const blhip_problem *p, int64_t n_chains, const double *op_values, int64_t Bmax, bool align) {
    // batches of a multiple of 8 chains: whole launches of the chain-resident kernel (8 chains per launch on 512-column grids)
    if (Bmax >= 16) Bmax -= Bmax % 8;
    const int64_t nbatch = (n_chains + Bmax - 1) / Bmax;
    int64_t even = (n_chains + nbatch - 1) / nbatch;
    if (nbatch > 1 && even >= 16) even = std::min(Bmax, (even + 7) / 8 * 8);
    std::vector<int64_t> start;
    for (int64_t c = 0; c < n_chains; c += even) start.push_back(c);
    start.push_back(n_chains);
    if (!align || (int64_t)start.size() != nbatch + 1 || nbatch < 2 || !op_values || p->n_ops == 0) return start;
    std::vector<int> bucket(n_chains);
    for (int64_t c = 0; c < n_chains; ++c) {
        bucket[c] = chain_bucket(p, op_values + c * p->n_ops);
        if (bucket[c] < 0) return start;
    }
    for (int64_t b = 1; b < nbatch; ++b) {
        // candidates: bucket changes between the previous cut and the next one; the batches on both sides must stay <= Bmax
        int64_t best = -1;
        for (int64_t c = start[b - 1] + 1; c < start[b + 1]; ++c) {
            if (bucket[c] == bucket[c - 1]) continue;
            if (c - start[b - 1] > Bmax || start[b + 1] - c > Bmax) continue;
            if (best < 0 || std::llabs(c - start[b]) < std::llabs(best - start[b])) best = c;
        }
        if (best >= 0) start[b] = best;
    }
    return start;
}

// One more cut where the axis-0 radius of a hyper-grid crosses the largest band of the matrix-pipe / chain-resident kernels (40): the
// chains below it keep those kernels, the chains above it take the column pre-pass (blh::vwide_kernel) -- without the cut ONE wide chain
// would route its whole batch through the pre-pass.  Only for grids sorted that way (every chain before the cut <= 40 < every chain after).
void split_wide_axis0(const blhip_problem *p, int64_t n_chains, const double *op_values, std::vector<int64_t> &start, int r_max) {
    if (p->ndim != 2 || !op_values || p->n_ops == 0 || n_chains < 2) return;
    auto radius0 = [&](int64_t c) {
        int r0 = 0;
        for (int k = 0; k < p->n_ops; ++k) {
            const blhip_op &op = p->ops[k];
            if (op.kind == BLHIP_OP_GRW && op.axis == 0) {
                const double ns = op_values[c * p->n_ops + k] / p->lattice[0];
                if (!(ns >= 0.0) || ns > 1e6) return -1;
                r0 = std::max(r0, (int)(4.0 * ns + 0.5));
            }
        }
        return r0;
    };
    int64_t cut = -1;
    for (int64_t c = 0; c < n_chains; ++c) {
        const int r = radius0(c);
        if (r < 0) return;
        if (r > r_max) { if (cut < 0) cut = c; }
        else if (cut >= 0) return;                   // a narrow chain after a wide one: not sorted by radius
    }
    if (cut <= 0) return;
    for (int64_t v : start) if (v == cut) return;
    start.insert(std::upper_bound(start.begin(), start.end(), cut), cut);
}

// ---- the phases of a fit: flags, shared tables (upload), memory plan; then per batch: program, geometry, metadata, forward pass +
//      evidence bookkeeping, backward pass + bookkeeping, carried states / average posterior / kept posterior, results ------------
struct FitFlags {
    bool evidence_only, forward_only, full, keep, accumulate, resume, carry;
};

FitFlags decode_flags(blhip_ctx *ctx, const blhip_problem *p, uint32_t flags, const double *log_w) {
    FitFlags f{};
    f.evidence_only = flags & BLHIP_EVIDENCE_ONLY;
    f.forward_only = (flags & BLHIP_FORWARD_ONLY) && !f.evidence_only;
    f.full = !f.evidence_only && !f.forward_only;
    f.keep = (flags & BLHIP_KEEP_POSTERIOR) && !f.evidence_only;
    f.accumulate = (flags & BLHIP_ACCUMULATE) && !f.evidence_only;
    f.resume = flags & BLHIP_RESUME;
    f.carry = flags & BLHIP_CARRY;
    if (f.resume || f.carry) {
        if (f.full) fail("BLHIP_RESUME / BLHIP_CARRY need a forward-only or evidence-only fit");
        if (p->carry_slot < 0) fail("carry_slot must be >= 0");
    }
    if (f.accumulate && !ctx->acc_active) fail("BLHIP_ACCUMULATE without blhip_accum_begin");
    if (f.accumulate && !log_w) fail("BLHIP_ACCUMULATE needs log_chain_weight");
    return f;
}

// what every chain of the call shares, resident in HBM for the duration of the call
struct DeviceTables {
    double *m0, *m1, *colA, *colB, *rec, *prior, *reset, *uniform, *indep, *lik;
    int rec_len, d;
};

// upload: marginal grids, per-column likelihood constants, per-step data records, prior(s); the (T, G) likelihood table of the
// closed-form table models is built on the device (table_model != 0), a caller-evaluated one (BLHIP_OM_TABLE) is copied
DeviceTables upload_tables(blhip_ctx *ctx, const blhip_problem *p, const Geometry &g, const FitFlags &ff, int table_model) {
    hipStream_t st = ctx->stream;
    const int64_t T = p->T;
    const long long G = g.G;
    DeviceTables D{};
    std::vector<double> rec;
    build_records(p, rec, D.rec_len, D.d);
    std::vector<double> colA(g.n1, 0.0), colB(g.n1, 0.0);
    const double *mcol = p->ndim == 1 ? p->marginal[0] : p->marginal[1];
    if (p->obs_model == BLHIP_OM_GAUSSIAN)
        for (int j = 0; j < g.n1; ++j) {
            const double s = mcol[j];
            colA[j] = 1.0 / (2.0 * s * s);
            colB[j] = 0.5 * std::log(2.0 * M_PI * s * s);
        }
    if (p->obs_model == BLHIP_OM_POISSON)
        for (int j = 0; j < g.n1; ++j) colA[j] = std::exp(-mcol[j]);

    size_t tb = 0;
    tb += carve_size(sizeof(double) * std::max(1, g.n0)) + 3 * carve_size(sizeof(double) * g.n1);
    tb += carve_size(sizeof(double) * rec.size()) + 4 * carve_size(sizeof(double) * G);
    ctx->tables.ensure(tb);
    char *cur = ctx->tables.as<char>();
    D.m0 = carve<double>(cur, std::max(1, g.n0));
    D.m1 = carve<double>(cur, g.n1);
    D.colA = carve<double>(cur, g.n1);
    D.colB = carve<double>(cur, g.n1);
    D.rec = carve<double>(cur, rec.size());
    D.prior = carve<double>(cur, G);
    D.reset = carve<double>(cur, G);
    D.uniform = carve<double>(cur, G);
    D.indep = carve<double>(cur, G);
    if (p->ndim == 2) HIPCHECK(hipMemcpyAsync(D.m0, p->marginal[0], sizeof(double) * g.n0, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(D.m1, mcol, sizeof(double) * g.n1, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(D.colA, colA.data(), sizeof(double) * g.n1, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(D.colB, colB.data(), sizeof(double) * g.n1, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(D.rec, rec.data(), sizeof(double) * rec.size(), hipMemcpyHostToDevice, st));
    // (the prior of a study that is fitted again: the caller's token says the content is the one this place already holds)
    const bool prior_resident = p->prior_token != 0 && p->prior_token == ctx->prior_token && D.prior == ctx->prior_dev && G == ctx->prior_G;
    if (!prior_resident) HIPCHECK(hipMemcpyAsync(D.prior, p->prior, sizeof(double) * G, hipMemcpyHostToDevice, st));
    ctx->prior_token = p->prior_token; ctx->prior_dev = D.prior; ctx->prior_G = G;
    if (p->reset_prior) HIPCHECK(hipMemcpyAsync(D.reset, p->reset_prior, sizeof(double) * G, hipMemcpyHostToDevice, st));
    if (p->indep_prior) HIPCHECK(hipMemcpyAsync(D.indep, p->indep_prior, sizeof(double) * G, hipMemcpyHostToDevice, st));
    if (ff.full) {
        // beta_T = 1/G   core.py:424-425 -- or the caller's backward message (blhip_problem.backward_init)
        if (p->backward_init) HIPCHECK(hipMemcpyAsync(D.uniform, p->backward_init, sizeof(double) * G, hipMemcpyHostToDevice, st));
        else BL_LAUNCH(fill_kernel, dim3(256), dim3(256), 0, st, D.uniform, G, 1.0 / (double)G);
    }
    D.lik = nullptr;
    if (p->obs_model == BLHIP_OM_TABLE) {
        ctx->likbuf.ensure(sizeof(double) * T * G);
        D.lik = ctx->likbuf.as<double>();
        if (table_model) {
            const size_t nd = (size_t)T * p->seg_len * p->data_dim;
            ctx->databuf.ensure(nd * 8);
            HIPCHECK(hipMemcpyAsync(ctx->databuf.p, p->data, nd * 8, hipMemcpyHostToDevice, st));
            const unsigned gx = (unsigned)std::min<long long>((G + NTHREADS - 1) / NTHREADS, 2048);
            BL_LAUNCH(lik_table_kernel, dim3(gx, (unsigned)T), dim3(NTHREADS), 0, st, table_model, D.lik, (long long)G, g.n1,
                               p->ndim, D.m0, D.m1, ctx->databuf.as<double>(), p->seg_len, p->data_dim);
            HIPCHECK(hipGetLastError());
        } else {
            HIPCHECK(hipMemcpyAsync(D.lik, p->lik, sizeof(double) * T * G, hipMemcpyHostToDevice, st));
        }
    }
    sync_stream(ctx, st);   // the host vectors above go out of use
    return D;
}

// memory plan: how many chains fit one batch (state ping-pong + the stored sequence + partial sums per chain within the budget)
constexpr int CHAIN_MIN_ROWS = 32;            // smallest grid (rows) the chain-resident kernels take (on the 128-row geometry)
constexpr int CHAIN_R0_MAX = 80;               // widest band of the chain-resident kernels (ring of 44 entries); option chain_wide = 0: FAST_R0_MAX
constexpr int CHAIN_TALL_ROWS = 1024;         // ... and the one geometry beyond 512 rows: grids of 513 .. 1024 rows (option chain_tall = 0: off)
inline bool chain_rows_ok(int n0) { return n0 >= CHAIN_MIN_ROWS && n0 <= CHAIN_TALL_ROWS; }
inline bool chain_tall(int n0) { return n0 > 512 && n0 <= CHAIN_TALL_ROWS; }

int64_t chains_per_batch(blhip_ctx *ctx, const blhip_problem *p, const Geometry &g, const FitFlags &ff, int64_t n_chains, int post_buffers) {
    const int64_t T = p->T;
    const long long G = g.G;
    size_t free_b = 0, total_b = 0;
    HIPCHECK(hipMemGetInfo(&free_b, &total_b));
    // (every reusable buffer of the context counts as available, so that the plan -- and with it the buffer sizes -- is the same from
    //  fit to fit: a plan that changed between two fits of one study re-allocated the 100-GB sequence buffer, 5 s)
    double budget = std::min((double)free_b + (double)ctx->state.cap + (double)ctx->post.cap + (double)ctx->post2.cap + (double)ctx->accpart.cap,
                             ctx->option("mem_budget_bytes", 0.70 * (double)total_b)) * 0.9;
    // the chain-resident kernels lay their sequences out on a padded geometry (rows 128 / 256 / 512, columns a multiple of 16)
    double Gk = (double)G;
    if (p->ndim == 2 && chain_rows_ok(g.n0))
        Gk = (double)((g.n0 + 127) / 128 * 128) * (double)((g.n1 + blc::WCOL - 1) / blc::WCOL * blc::WCOL);
    {   // walks on both parameters: the transposing kernels lay their sequences out on a SQUARE geometry (blhip_chainax.hpp)
        bool w0 = false, w1 = false;
        for (int k = 0; k < p->n_ops; ++k)
            if (p->ops[k].kind == BLHIP_OP_GRW) { if (g.axis_map[p->ops[k].axis] == 0) w0 = true; else w1 = true; }
        if (p->ndim == 2 && w0 && w1 && std::max(g.n0, g.n1) <= 512) {
            const double n = std::max(g.n0, g.n1) <= 128 ? 128.0 : (std::max(g.n0, g.n1) <= 256 ? 256.0 : 512.0);
            Gk = std::max(Gk, n * n);
        }
    }
    // the partial accumulators of the fused fold (ChainRun::setup: one (T, G) slot per block column of a launch) come out of the same memory
    // -- only where the chain-resident path can be taken at all (else they are never allocated: a narrow grid with a long series
    //    gave up its whole budget to 128 slots it never used and ran one chain per batch), and never more than half of the budget
    if (ff.accumulate && ff.full && p->ndim == 2 && p->obs_model == BLHIP_OM_GAUSSIAN && chain_rows_ok(g.n0) &&
        g.n1 >= 1 && g.n1 <= 16 * blc::MAX_STRIPS && ctx->option("chain_resident", 1.0) != 0.0 && ctx->resident_ok) {
        const double slots = std::max(1, std::min(ctx->num_cus, 256) / ((g.n1 + blc::WCOL - 1) / blc::WCOL));
        budget -= std::min(0.5 * budget, std::min<double>(slots, (double)n_chains) * (double)T * Gk * 8.0);
    }
    const double per_chain = (ff.evidence_only ? 2.0 : (double)post_buffers * (double)T + 2.0) * Gk * 8.0 +
                             (double)T * NRED * 8.0 * 2 * 64.0 /*partials, rough*/;
    int64_t Bmax = (int64_t)std::max(1.0, std::floor(budget / per_chain));
    // (1-D grids: a chain is a few KB and a batch of the chain-resident 1-D kernel costs a fixed ~1 ms of launches, syncs and read-backs --
    //  the reference's published break-point study, 23 400 chains: 61 ms with batches of 1024, 52 with 4096, 54 with 8192)
    Bmax = std::min<int64_t>(Bmax, (int64_t)ctx->option("max_batch", p->ndim == 1 ? 4096 : 1024));
    Bmax = std::min<int64_t>(Bmax, 65535);
    if (ff.keep && n_chains > Bmax) fail("BLHIP_KEEP_POSTERIOR: %lld chains do not fit in device memory at once", (long long)n_chains);
    if ((ff.resume || ff.carry) && n_chains > Bmax) fail("carried states: %lld chains do not fit in one batch", (long long)n_chains);
    if (ff.resume) {
        auto it = ctx->carry.find(p->carry_slot);
        if (it == ctx->carry.end() || !it->second.valid) fail("BLHIP_RESUME: carry slot %d holds no state", p->carry_slot);
        if (it->second.chains != n_chains || it->second.G != G)
            fail("BLHIP_RESUME: carry slot %d holds %lld chains x %lld cells, the call has %lld x %lld", p->carry_slot,
                 (long long)it->second.chains, (long long)it->second.G, (long long)n_chains, (long long)G);
    }
    return Bmax;
}

// which kernel family runs a batch and with what block geometry (segment lengths from a small cost model: long segments read every
// element once + 2 R0 halo rows per segment, short ones give enough blocks to fill 256 CUs when there are few chains)
struct GeometryPlan {
    bool shift1d = false;        // the chain-resident 1-D kernel's flavours with spline shifts (Deterministic steps) / clamps (RegimeSwitch, NotEqual)
    bool clamp1d = false;        // ... the one with clamps (CL = 2)
    bool fast = false, fused1d = false, use_mfma = false;
    bool chain1d = false;         // 1-D batches: one block per chain runs the whole pass (blhip_chain1d.hpp); bookkeeping of a K = 1 fused pass
    bool wideH = false;           // axis-1 walks wider than the fused kernels' halo: row filter as a pre-pass per step (blhip_hwide.hpp)
    bool wideV = false;           // axis-0 walks wider than the matrix-pipe kernels' largest band: column filter as a pre-pass, no stencil left
    bool hSplit = false;          // wideH: chains of the batch whose axis-1 filter is absent (or narrow: hFusedMax > 0) keep their fused kernels
    int hFusedMax = 0;
    int64_t fusedK = 1;
    int f1_TJ = 128;
    Tile tile{};
    int fastS = 0, fast_nseg = 1, fast_fnblk = 1, mS = 0, m_nseg = 1, m_tiles_j = 1, m_nblk = 1;
};

GeometryPlan plan_geometry(blhip_ctx *ctx, const blhip_problem *p, const Geometry &g, const ChainProgram &prog, int64_t B, int d,
                       bool resume, bool carry, const TapTable *taps = nullptr) {
    GeometryPlan gp;
    const int64_t T = p->T;
    // fast path (blhip_fast.hpp) when the whole batch qualifies, otherwise the generic LDS-tile kernel
    const bool wide_h_ok = prog.LW1 <= blh::HW_MAX && prog.LW1 < g.n1 && ctx->option("wide_h", 1.0) != 0.0;
    const bool wide_v = prog.LW0 > FAST_R0_MAX && prog.LW0 <= blh::VW_MAX && prog.LW0 < g.n0 && wide_h_ok && ctx->option("wide_v", 1.0) != 0.0;
    gp.fast = p->ndim == 2 && (p->obs_model == BLHIP_OM_GAUSSIAN || p->obs_model == BLHIP_OM_TABLE) &&
                      ctx->option("fast", 1.0) != 0.0 && !prog.has_clamp && (prog.LW0 <= FAST_R0_MAX || wide_v) &&
                      (prog.LW1 <= blf::R1MAX || wide_h_ok) &&
                      g.n0 >= (wide_v ? 0 : ((prog.LW0 + 7) / 8) * 8) + 2 * blf::CH && g.n1 >= 2 * blf::R1MAX && d <= blf::DMAX;
    gp.wideV = gp.fast && wide_v;
    gp.wideH = gp.fast && (prog.LW1 > blf::R1MAX || (gp.wideV && prog.LW1 > 0));       // (with a column pre-pass every filter runs as a pre-pass)
    // Chains of such a batch WITHOUT an axis-1 filter (a hyper-grid that includes the width 0) skip the pre-pass: it would be a copy.
    // wide_h_fused_max = 8: chains with a narrow filter keep the fused kernels' own axis-1 part as well -- 15 % fewer bytes on
    // extra.c4_both_axes but no faster (the fused both-axes kernels at radii up to 40 are bound by the fp64 pipe: 5.44e10 -> 5.47e10), so
    // the default sends every filter through the pre-pass.
    bool any_narrow = false;
    gp.hFusedMax = std::min(blf::R1MAX, std::max(0, (int)ctx->option("wide_h_fused_max", 0.0)));
    if (gp.wideH && !gp.wideV && taps && ctx->option("wide_h_split", 1.0) != 0.0) {
        bool any_none = false;
        for (size_t e = 0; e < prog.tapF1.size() && !(any_narrow && any_none); ++e) {
            const int k = prog.tapF1[e];
            if (k < 0) any_none = true; else if (gp.hFusedMax > 0 && taps->lw[k] <= gp.hFusedMax) any_narrow = true;
        }
        gp.hSplit = any_narrow || any_none;
    }
    // (programs whose only clamp mode is a Deterministic model's spline shift: the chain-resident kernel has a flavour for them
    //  -- bl1c::chain1d_kernel SHIFT --, the K-steps-per-launch and persistent kernels have not: chain1d or the generic kernel)
    // (round 6: ... and the one with the clamps of RegimeSwitch / NotEqual -- bl1c::chain1d_kernel CL = 2; the dense zero-boundary kernels of
    //  the AlphaStable walk keep the generic kernel)
    const bool shift1d = p->ndim == 1 && prog.has_clamp && !prog.dense_clamp && ctx->option("chain1d_shift", 1.0) != 0.0 &&
                         (!prog.other_clamp || ctx->option("chain1d_clamp", 1.0) != 0.0);
    if (p->ndim == 1 && !gp.fast && (!prog.has_clamp || shift1d) && ctx->option("fuse1d", 8.0) >= 1.0 &&
        (p->obs_model == BLHIP_OM_POISSON || p->obs_model == BLHIP_OM_GAUSSIAN_MEAN || p->obs_model == BLHIP_OM_TABLE)) {
        gp.f1_TJ = 128;
        gp.fusedK = std::max<int64_t>(1, std::min<int64_t>((int64_t)ctx->option("fuse1d", 8.0), T));
        // keep the redundantly recomputed halo (K * LW cells per side) within ~4x the owned cells and the window in LDS
        while (gp.fusedK > 1 && (gp.fusedK * prog.LW1 > 2 * gp.f1_TJ || (size_t)(gp.f1_TJ + 2 * gp.fusedK * prog.LW1) * 32 > 96 * 1024)) --gp.fusedK;
        gp.fused1d = (size_t)(gp.f1_TJ + 2 * gp.fusedK * prog.LW1) * 32 + (size_t)gp.fusedK * gp.f1_TJ * 32 + 4096 <= 150 * 1024;
        // Batches of chains: one block per chain for the whole pass when that is cheaper per step than the alternatives.  Per step
        // (shader cycles; measured with tools/probe.py chain1d, profiles/r04_notes.md): the chain's row is filtered out of ONE CU's LDS --
        // n (2 lw + 1) 16-byte operand pairs at 128 B per clock, ~1 k cycles of barrier / sums / likelihood -- and ceil(B / CUs) blocks
        // share a CU one after the other; the K-steps-per-launch path costs a launch (~14 k cycles) every K steps, the persistent
        // one (all blocks of all chains on the chip at once) ~4 k (K > 1) / ~7 k (K = 1: a hand-off per step) per step.
        const double c1d_mode = ctx->option("chain1d", 1.0);
        if (shift1d) gp.fused1d = true;              // (decided below: without the chain-resident kernel the batch keeps the generic one)
        if (gp.fused1d && c1d_mode != 0.0 && !resume && !carry && prog.LW1 < g.n1 && g.n1 <= bl1c::NMAX &&
            bl1c::lds_doubles(g.n1, prog.LW1, shift1d) * 8 <= 150 * 1024) {
            // microseconds per time step of the whole batch, fitted to tools/probe.py chain1d (profiles/r04_notes.md): a block's step =
            // 1.5 us + 1.0 ns per cell (likelihood from the shared table; 2.2 ns with Poisson's pow() in the kernel) + 44 ps per cell and
            // tap (the stencil's operand pairs come out of ONE CU's LDS at ~9 per clock), blocks beyond the chip's capacity queue up;
            // persistent K-step kernel (all blocks of all chains on the chip at once) 1.7 us + 40 ns per cell of radius; a launch per K
            // steps 1.5 us + (29 + 0.63 radius) ps per cell of the batch
            const int cus = std::min(ctx->num_cus, 256);
            const double n = g.n1, lw = prog.LW1;
            // (rows longer than a block: two cells per thread share their operand pairs, 21 ps per cell and tap -- launch_chain1d)
            const double tap_us = g.n1 > bl1c::NT && true ? 2.1e-5 : 4.4e-5;
            const double est_c1d = (double)((B + cus - 1) / cus) * (1.5 + n * (B >= 4 ? 0.0010 : 0.0022) + n * (2.0 * lw + 1.0) * tap_us);
            const int nblk_f = (g.n1 + gp.f1_TJ - 1) / gp.f1_TJ;
            const bool p1d_ok = ctx->option("persist1d", 1.0) != 0.0 && (long long)nblk_f * B <= cus && T > gp.fusedK;
            const double est_other = p1d_ok ? 1.7 + 0.04 * lw : 1.5 + (double)B * n * (29.0 + 0.63 * lw) * 1e-6;
            // (shift1d: the alternative is a launch per step.  A single chain takes the kernel too when the model says so -- rows of a few
            //  hundred cells with a narrow stencil: 200 cells, radius 27: 2.4 against 2.8 us per step)
            // (clamps without a Deterministic model -- the reference's regime-switch tutorial is ONE such chain: a step of a few hundred cells
            //  costs the block ~2 us against a launch of the generic kernel)
            gp.chain1d = c1d_mode == 2.0 || (shift1d && (B >= 2 || !prog.has_shift)) || (!shift1d && est_c1d < est_other);
            if (gp.chain1d) { gp.fusedK = 1; gp.f1_TJ = g.n1; }
        }
        if (shift1d && !gp.chain1d) gp.fused1d = false;
        gp.shift1d = shift1d && gp.chain1d;
        gp.clamp1d = gp.shift1d && prog.other_clamp;
    }
    if (gp.fast) {
        gp.tile.TI = blf::CH; gp.tile.LW0 = gp.wideV ? 0 : prog.LW0; gp.tile.LW1 = (prog.LW1 > 0 && (!gp.wideH || (gp.hSplit && any_narrow))) ? blf::R1MAX : 0;
        gp.tile.TJ = blf::BW - 2 * gp.tile.LW1;
        gp.tile.tiles_j = (g.n1 + gp.tile.TJ - 1) / gp.tile.TJ;
        // rows per block segment: long segments read every element once (+ 2*R0 halo rows per segment), short ones
        // give enough blocks to fill 256 CUs when there are few chains.  Model: cost = waves * blocks_per_CU * rows.
        const long long colblocks = (long long)gp.tile.tiles_j * B;
        const int R0 = (prog.LW0 == 0 || gp.wideV) ? 0 : ((prog.LW0 + 7) / 8) * 8;
        double best = 1e300;
        const int forceS = (int)ctx->option("fast_S", 0);
        for (int k = 1; k <= 4; k *= 2) {
            const double pen = k == 1 ? 1.6 : (k == 2 ? 1.15 : 1.0);
            for (int ns = 1; ns <= std::max(1, g.n0 / 16); ++ns) {
                int S = ((g.n0 + ns - 1) / ns + blf::CH - 1) / blf::CH * blf::CH;
                const int real = (g.n0 + S - 1) / S;
                const long long blocks = colblocks * real;
                const long long waves = (blocks + 256LL * k - 1) / (256LL * k);
                const double cost = pen * (double)waves * k * (S + 2.0 * R0 + 4.0);
                if (cost < best - 1e-9) { best = cost; gp.fastS = S; gp.fast_nseg = real; }
            }
        }
        if (forceS > 0) { gp.fastS = (forceS + blf::CH - 1) / blf::CH * blf::CH; gp.fast_nseg = (g.n0 + gp.fastS - 1) / gp.fastS; }
        gp.tile.tiles_i = gp.fast_nseg;
        gp.fast_fnblk = gp.tile.tiles_j * gp.fast_nseg;
        // geometry of the matrix-pipe kernel (64-column strips, segments of mS rows, mS a multiple of 16)
        {
            gp.m_tiles_j = (g.n1 + blm::BCOL - 1) / blm::BCOL;
            const long long mcol = (long long)gp.m_tiles_j * B;
            double mbest = 1e300;
            for (int k = 1; k <= 4; ++k) {                     // resident blocks per CU
                const double pen = k == 1 ? 1.5 : (k == 2 ? 1.15 : 1.0);
                for (int ns = 1; ns <= std::max(1, g.n0 / 32); ++ns) {
                    int S = ((g.n0 + ns - 1) / ns + blm::SEG_Q - 1) / blm::SEG_Q * blm::SEG_Q;
                    if (S > blm::MS_MAX) continue;
                    const int real = (g.n0 + S - 1) / S;
                    const long long blocks = mcol * real;
                    const long long waves = (blocks + 256LL * k - 1) / (256LL * k);
                    const double cost = pen * (double)waves * k * (S + 1.0 * R0 + 24.0);
                    if (cost < mbest - 1e-9) { mbest = cost; gp.mS = S; gp.m_nseg = real; }
                }
            }
            const int forceM = (int)ctx->option("mfma_S", 0);
            if (forceM > 0) { gp.mS = std::min(blm::MS_MAX, (forceM + blm::SEG_Q - 1) / blm::SEG_Q * blm::SEG_Q); gp.m_nseg = (g.n0 + gp.mS - 1) / gp.mS; }
            if (gp.mS == 0) { gp.mS = blm::MS_MAX; gp.m_nseg = (g.n0 + gp.mS - 1) / gp.mS; }
            gp.m_nblk = gp.m_tiles_j * gp.m_nseg;
        }
        gp.use_mfma = ctx->option("mfma", 1.0) != 0.0;
        gp.tile.nblk = gp.use_mfma ? std::max(gp.fast_fnblk, gp.m_nblk) : gp.fast_fnblk;
        gp.tile.lds_bytes = 0;
    } else if (gp.fused1d) {
        gp.tile.TI = 1; gp.tile.TJ = gp.f1_TJ; gp.tile.LW0 = 0; gp.tile.LW1 = prog.LW1; gp.tile.tiles_i = 1;
        gp.tile.tiles_j = (g.n1 + gp.f1_TJ - 1) / gp.f1_TJ; gp.tile.nblk = gp.tile.tiles_j; gp.tile.lds_bytes = 0;
    } else {
        gp.tile = choose_tile(ctx, g, prog.LW0, prog.LW1, prog.whole_row);
        if (prog.whole_row && gp.tile.tiles_j != 1) fail("internal: a two-stage spline shift needs the whole row in one tile");
    }
    return gp;
}

// per-(step, chain) metadata of a batch in HBM: source kinds, tap-set ids, clamp modes, the per-step launch order of the radius buckets,
// the tap table; plus scratch the finalisation kernels use
struct DeviceMeta {
    unsigned char *kindF, *kindB, *cmodeF, *cmodeB;
    double *limitF, *limitB;
    int *tapF0, *tapF1, *tapB0, *tapB1, *orderF, *orderB;
    double *taps;
    int *off, *lw, *lw2;
    double *invN, *w, *dump;
    // host copies the launch loop reads
    std::vector<int> h_orderF, h_orderB;
    std::vector<std::vector<FastRange>> rangesF, rangesB;
};

void upload_metadata(blhip_ctx *ctx, const blhip_problem *p, const ChainProgram &prog, TapTable &taps, int64_t B, bool full, bool fast, int nblk,
                     DeviceMeta &M, bool wideH = false, bool wideV = false, bool h_split = false, int h_fused_max = 0) {
    hipStream_t st = ctx->stream;
    const int64_t T = p->T;
    const size_t nT = (size_t)T * B;
    taps.w.resize(taps.w.size() + 64, 0.0);      // zero padding: the fast kernels read up to R0 weights per tap set
    size_t mb = 4 * carve_size(nT) + 6 * carve_size(nT * sizeof(int)) + carve_size(taps.w.size() * 8 + 8) +
                3 * carve_size(taps.off.size() * 4 + 4) + 4 * carve_size(sizeof(double) * nT) + carve_size(8 * B) + carve_size(8 * 4 * NTHREADS);
    ctx->meta.ensure(mb);
    char *cur = ctx->meta.as<char>();
    M.kindF = carve<unsigned char>(cur, nT); M.kindB = carve<unsigned char>(cur, nT);
    M.cmodeF = carve<unsigned char>(cur, nT); M.cmodeB = carve<unsigned char>(cur, nT);
    M.limitF = carve<double>(cur, nT); M.limitB = carve<double>(cur, nT);
    M.tapF0 = carve<int>(cur, nT); M.tapF1 = carve<int>(cur, nT);
    M.tapB0 = carve<int>(cur, nT); M.tapB1 = carve<int>(cur, nT);
    M.orderF = carve<int>(cur, nT); M.orderB = carve<int>(cur, nT);
    M.taps = carve<double>(cur, taps.w.size() + 1);
    M.off = carve<int>(cur, taps.off.size() + 1); M.lw = carve<int>(cur, taps.off.size() + 1);
    M.lw2 = carve<int>(cur, taps.off.size() + 1);
    M.invN = carve<double>(cur, nT);
    (void)carve<double>(cur, nT);
    M.w = carve<double>(cur, B);
    M.dump = carve<double>(cur, 4 * NTHREADS);   // (the halo wave of an H block spreads its dummy stores over 8 x 64 slots)
    HIPCHECK(hipMemcpyAsync(M.kindF, prog.kindF.data(), nT, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(M.tapF0, prog.tapF0.data(), nT * 4, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(M.tapF1, prog.tapF1.data(), nT * 4, hipMemcpyHostToDevice, st));
    if (prog.has_clamp) {
        HIPCHECK(hipMemcpyAsync(M.cmodeF, prog.cmodeF.data(), nT, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(M.limitF, prog.limitF.data(), nT * 8, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(M.cmodeB, prog.cmodeB.data(), nT, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(M.limitB, prog.limitB.data(), nT * 8, hipMemcpyHostToDevice, st));
    }
    if (full) {
        HIPCHECK(hipMemcpyAsync(M.kindB, prog.kindB.data(), nT, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(M.tapB0, prog.tapB0.data(), nT * 4, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(M.tapB1, prog.tapB1.data(), nT * 4, hipMemcpyHostToDevice, st));
    }
    if (fast) {
        // a launch with fewer than ~128 blocks leaves most of the 256 CUs idle: a radius bucket with fewer blocks joins the next one
        const long long min_blocks = 128;
        const int min_chains = (int)std::min<long long>(B, (min_blocks + (long long)nblk - 1) / nblk);
        // (wideH: the axis-1 filters wider than the fused kernels' 8 columns run in the pre-pass, and the fused kernels of those chains are
        //  launched without an axis-1 part; chains of the same step with a narrow filter or none keep their fused kernels -- unless the
        //  step also has the axis-0 pre-pass, or with wide_h_split = 0: then every chain of the step goes through the pre-pass)
        const std::vector<int> no_h((wideH || wideV) ? (size_t)B : 0, -1);
        auto all_pre = [&](std::vector<FastRange> &rs) { if (wideH && !h_split) for (auto &r : rs) r.pre = true; };
        M.h_orderF.resize(nT); M.rangesF.resize(T);
        for (int64_t t = 0; t < T; ++t) {
            bucket_step(wideV ? no_h.data() : &prog.tapF0[t * B], (wideH && !h_split) ? no_h.data() : &prog.tapF1[t * B], taps.lw, (int)B, &M.h_orderF[t * B], M.rangesF[t], min_chains,
                        h_split ? h_fused_max : -1);
            all_pre(M.rangesF[t]);
        }
        HIPCHECK(hipMemcpyAsync(M.orderF, M.h_orderF.data(), nT * 4, hipMemcpyHostToDevice, st));
        if (full) {
            M.h_orderB.resize(nT); M.rangesB.resize(T);
            for (int64_t t = 0; t < T; ++t) {
                bucket_step(wideV ? no_h.data() : &prog.tapB0[t * B], (wideH && !h_split) ? no_h.data() : &prog.tapB1[t * B], taps.lw, (int)B, &M.h_orderB[t * B], M.rangesB[t], min_chains,
                            h_split ? h_fused_max : -1);
                all_pre(M.rangesB[t]);
            }
            HIPCHECK(hipMemcpyAsync(M.orderB, M.h_orderB.data(), nT * 4, hipMemcpyHostToDevice, st));
        }
    }
    if (!taps.w.empty()) {
        HIPCHECK(hipMemcpyAsync(M.taps, taps.w.data(), taps.w.size() * 8, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(M.off, taps.off.data(), taps.off.size() * 4, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(M.lw, taps.lw.data(), taps.lw.size() * 4, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(M.lw2, taps.lw2.data(), taps.lw2.size() * 4, hipMemcpyHostToDevice, st));
    }
}

// host-side results of one batch of chains
struct BatchOutcome {
    std::vector<double> logE, local, means, invN;    // (B,), (B, T), (B, ndim, T), (B, T): 1 / row sum of the stored sequence
    std::vector<int64_t> abort_step;
    std::vector<int32_t> abort_phase;
};

// BLHIP_CARRY: keep every chain's filtered distribution of the last step, normalised (core.py:2173)
void store_carry(blhip_ctx *ctx, const blhip_problem *p, int64_t B, long long G, const double *redF, const double *fin, long long fstr,
                 double *d_w, bool has_clamp) {
    hipStream_t st = ctx->stream;
    const int64_t T = p->T;
    blhip_ctx::Carry &cs = ctx->carry[p->carry_slot];
    cs.buf.ensure((size_t)B * G * 8);
    std::vector<double> inv(B);
    for (int64_t b = 0; b < B; ++b) inv[b] = 1.0 / redF[((size_t)(T - 1) * B + b) * NRED];
    HIPCHECK(hipMemcpyAsync(d_w, inv.data(), B * 8, hipMemcpyHostToDevice, st));
    const unsigned gx = (unsigned)std::min<long long>((G + NTHREADS - 1) / NTHREADS, 4096);
    BL_LAUNCH(carry_store_kernel, dim3(gx, (unsigned)B), dim3(NTHREADS), 0, st, cs.buf.as<double>(), fin, fstr, G, d_w);
    sync_stream(ctx, st);
    cs.chains = B; cs.G = G; cs.valid = true;
    cs.maxv.clear();
    if (has_clamp)                               // clamp batches run the generic kernel, which reports the state maximum
        for (int64_t b = 0; b < B; ++b) cs.maxv.push_back(redF[((size_t)(T - 1) * B + b) * NRED + 6] * inv[b]);
}

// fold the batch into the average posterior (core.py:1358-1366): linear accumulator with a running reference exponent.
// Two halves, so that the kernel can be launched later than the bookkeeping is done (overlapped folds, see do_fit): prepare_fold
// turns the batch's evidences into weights (staged in h_w / h_invN, which must stay valid until the launch has consumed them) and
// advances the accumulator's reference; launch_fold copies them to the device and runs the pass on `st`.
struct FoldJob {
    bool pending = false;
    const double *d_post = nullptr;
    int64_t B = 0;
    double *h_w = nullptr, *h_invN = nullptr;        // host staging (B), (T * B)
    double *d_w = nullptr, *d_invN = nullptr;        // device copies
    double r = 0.0;                                  // factor of what the accumulator already holds
    int first = 0;
    int parity = 0;
    int sm_n0 = 0;                                   // > 0: d_post is in the chain-resident kernel's strip-major layout (rows per strip)
    int pad_n0p = 0, pad_n0 = 0, pad_n1 = 0;         // > 0: ... on a padded geometry (rows per strip n0p; the grid's true sizes)
    int pad_ax = 0;                                  //      ... in the alternating layouts of the both-axes kernels (blk::ax_layout_b)
    long long pad_step = 0;                          //      doubles per time step of a chain's sequence there
};

bool prepare_fold(blhip_ctx *ctx, int64_t T, int64_t B, const BatchOutcome &out, const double *log_w_batch, FoldJob &job) {
    double newref = ctx->acc_logref;
    std::vector<double> lw(B, -INFINITY);
    std::vector<char> valid(B, 0);
    for (int64_t b = 0; b < B; ++b) {
        // np.isfinite(logEvidence) guard (core.py:1358); a zero hyper-prior contributes log(0) = -inf, i.e. nothing
        valid[b] = out.abort_step[b] < 0 && std::isfinite(out.logE[b]) && std::isfinite(log_w_batch[b]);
        if (!valid[b]) continue;
        lw[b] = out.logE[b] + log_w_batch[b];
        if (lw[b] > newref) newref = lw[b];
    }
    if (!std::isfinite(newref)) return false;
    int nfold = 0;
    for (int64_t b = 0; b < B; ++b) {
        job.h_w[b] = valid[b] ? std::exp(lw[b] - newref) : 0.0;
        nfold += valid[b] ? 1 : 0;
    }
    std::memcpy(job.h_invN, out.invN.data(), (size_t)T * B * 8);
    job.r = ctx->acc_first ? 0.0 : std::exp(ctx->acc_logref - newref);
    job.first = ctx->acc_first ? 1 : 0;
    job.B = B;
    ctx->timing.accumulate_launches += 1;
    ctx->acc_logref = newref;
    ctx->acc_first = false;
    ctx->acc_folded += nfold;
    return true;
}

void launch_fold(blhip_ctx *ctx, int64_t T, long long G, const FoldJob &job, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    const int64_t B = job.B;
    HIPCHECK(hipMemcpyAsync(job.d_w, job.h_w, B * 8, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(job.d_invN, job.h_invN, (size_t)T * B * 8, hipMemcpyHostToDevice, st));
    HIPCHECK(hipEventRecord(ev0, st));
    if (job.pad_n0p > 0) {
        BL_LAUNCH(accumulate_pad_kernel, dim3((unsigned)((G + NTHREADS - 1) / NTHREADS), (unsigned)T), dim3(NTHREADS), 0, st, ctx->acc,
                           job.d_post, (long long)T * job.pad_step, (int)B, job.pad_n0, job.pad_n1, (int)T, job.d_w, job.d_invN, job.r, job.first,
                           job.pad_n0p, job.pad_step, job.pad_ax);
    } else if (job.sm_n0 == 0 && B >= 16 && ((G / 2 + NTHREADS - 1) / NTHREADS) * T < 1024) {        // small grids: too few blocks with a thread per cell
        BL_LAUNCH(accumulate_small_kernel, dim3((unsigned)((G + 63) / 64), (unsigned)T), dim3(NTHREADS), 0, st, ctx->acc, job.d_post,
                           (long long)T * G, (int)B, G, (int)T, job.d_w, job.d_invN, job.r, job.first);
    } else if ((G & 1) == 0 && ((uintptr_t)ctx->acc & 15) == 0) {
        const unsigned gx2 = (unsigned)((G / 2 + NTHREADS - 1) / NTHREADS);
        BL_LAUNCH(accumulate2_kernel, dim3(gx2, (unsigned)T), dim3(NTHREADS), 0, st, ctx->acc, job.d_post,
                           (long long)T * G, (int)B, G, (int)T, job.d_w, job.d_invN, job.r, job.first, job.sm_n0);
    } else {
        if (job.sm_n0 > 0) fail("internal: strip-major sequences need an even number of cells and a 16-byte aligned accumulator");
        const unsigned gx = (unsigned)std::min<long long>((G + NTHREADS - 1) / NTHREADS, 4096);
        BL_LAUNCH(accumulate_kernel, dim3(gx, (unsigned)T), dim3(NTHREADS), 0, st, ctx->acc, job.d_post,
                           (long long)T * G, (int)B, G, (int)T, job.d_w, job.d_invN, job.r, job.first);
    }
    HIPCHECK(hipEventRecord(ev1, st));
}

// the whole fold on the main stream, waited for
// (later_ev: do not wait -- the fold stays in front of whatever the stream runs next, e.g. the next batch's metadata uploads and forward
//  pass; its two timing events are appended for the caller to read after the stream has drained.  The page-locked staging of the weights is
//  reused by the next batch's fold only after that batch's passes have been waited for, on the same stream.)
void fold_accumulate(blhip_ctx *ctx, int64_t T, long long G, int64_t B, const BatchOutcome &out, const double *log_w_batch, const double *d_post,
                     double *d_w, double *d_invN, int sm_n0 = 0, const FoldJob *layout = nullptr, std::vector<hipEvent_t> *later_ev = nullptr) {
    ctx->pinA.ensure(((size_t)B + (size_t)T * B) * 8);
    FoldJob job;
    if (layout) job = *layout;
    job.sm_n0 = sm_n0;
    job.h_w = ctx->pinA.as<double>(); job.h_invN = job.h_w + B;
    job.d_w = d_w; job.d_invN = d_invN; job.d_post = d_post;
    if (!prepare_fold(ctx, T, B, out, log_w_batch, job)) return;
    if (later_ev) {
        hipEvent_t e0, e1;
        HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
        later_ev->push_back(e0); later_ev->push_back(e1);
        launch_fold(ctx, T, G, job, ctx->stream, e0, e1);
        return;
    }
    launch_fold(ctx, T, G, job, ctx->stream, ctx->ev[4], ctx->ev[5]);
    sync_stream(ctx, ctx->stream);
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5]));
    ctx->timing.accumulate_ms += ms;
}

// the batch's sequence stays on the device as the kept posterior; rows [row0, row1) still carry their raw sums (core.py:389 / :441)
void keep_posterior(blhip_ctx *ctx, const Geometry &g, int64_t T, int64_t B, const BatchOutcome &out, int64_t row0, int64_t row1) {
    hipStream_t st = ctx->stream;
    ctx->postinv.ensure((size_t)T * B * 8);
    HIPCHECK(hipMemcpyAsync(ctx->postinv.p, out.invN.data(), (size_t)T * B * 8, hipMemcpyHostToDevice, st));
    sync_stream(ctx, st);
    ctx->post_valid = true; ctx->post_scaled = false; ctx->post_chains = B; ctx->post_T = T; ctx->post_G = g.G;
    ctx->post_row0 = row0; ctx->post_row1 = row1;
    ctx->post_n0 = g.n0; ctx->post_n1 = g.n1;
    // normalised now, as part of the fit (core.py:441 is inside Study.fit)
    ensure_post_scaled(ctx); sync_stream(ctx, st);
}

void write_results(blhip_result *res, const blhip_problem *p, int64_t c0, int64_t B, const BatchOutcome &out, bool with_means) {
    if (!res) return;
    const int64_t T = p->T;
    for (int64_t b = 0; b < B; ++b) {
        if (res->log_evidence) res->log_evidence[c0 + b] = out.logE[b];
        if (res->abort_step) res->abort_step[c0 + b] = out.abort_step[b];
        if (res->abort_phase) res->abort_phase[c0 + b] = out.abort_phase[b];
        if (res->local_evidence)
            std::memcpy(res->local_evidence + (size_t)(c0 + b) * T, &out.local[(size_t)b * T], T * 8);
        if (res->posterior_mean && with_means)
            std::memcpy(res->posterior_mean + (size_t)(c0 + b) * p->ndim * T, &out.means[(size_t)b * p->ndim * T], (size_t)p->ndim * T * 8);
    }
}

// The chain-resident path (blhip_chainres.hpp): which chains of the batch run together, in which order, with which band width.
struct ChainResPlan {
    int ntw = 0, strips = 0, cpr = 0;            // product tiles per wave, strips per chain, chains per launch
    int n0p = 0, n1p = 0;                        // the geometry the kernels work on: rows 128 / 256 / 512, columns a multiple of 16
    bool pad = false;                            // the grid is smaller than that (padded cells hold zeros; sequences private to the fit only)
    int r0_max = 40;                             // widest axis-0 radius the launch may carry (set by the caller: 80 for 1024 rows)
    bool has_reset = false;                      // change points: some steps consume the reset distribution
    bool mixed = false;                          // ... in chains that also filter (random walk + change point in one model)
    std::vector<unsigned char> ckF, ckB;         // [T][B] what a step of the chain kernels consumes: SRC_PREV / SRC_RESET (| 0x80: unfiltered)
    std::vector<int> order, tap_id;              // chains sorted by stencil radius; the chain's axis-0 kernel (-1: none)
    bool allow_ax1 = false;                      // (set by the caller) walks on the second parameter too may be planned: blc::chainax_kernel
    bool ax1 = false;                            // ... and some chain has one: the batch runs the transposing kernels (square exact geometry)
    std::vector<int> tap_id1;                    // the chain's axis-1 kernel (-1: none)
    std::vector<int> round_start, round_nk;      // launches: order[round_start[r] .. round_start[r + 1]), band blocks NK
};

// every chain: prior, then the SAME axis-0 kernel at every step, nothing on axis 1 (forward; mirrored backward)
bool plan_chainres(const Geometry &g, const ChainProgram &prog, const TapTable &taps, int64_t B, int64_t T, bool full, int cus, ChainResPlan &cp) {
    // any grid of 32 .. 512 rows: the kernels work on the next geometry of 128 / 256 / 384 / 512 rows x a multiple of 16 columns;
    // 513 .. 1024 rows: 1024 rows x a multiple of 16 columns (one copy of the strip in LDS: blc::chain_kernel TALL; change-point batches keep their state in registers there too)
    if (!chain_rows_ok(g.n0)) return false;
    cp.n0p = (g.n0 + 127) / 128 * 128;
    cp.n1p = (g.n1 + blc::WCOL - 1) / blc::WCOL * blc::WCOL;
    if (chain_tall(g.n0)) cp.n0p = CHAIN_TALL_ROWS;
    cp.pad = cp.n0p != g.n0 || cp.n1p != g.n1;
    cp.strips = cp.n1p / blc::WCOL;
    cp.ntw = cp.n0p / (blc::NW * blc::TM);
    if (cp.strips > blc::MAX_STRIPS || cp.strips > cus) return false;
    cp.cpr = cus / cp.strips;
    cp.tap_id.assign(B, -1);
    cp.tap_id1.assign(B, -1);
    cp.ax1 = false;
    // walks on both parameters (blhip_chainax.hpp): an exact square geometry (the blocks of a chain change between column strips and row strips)
    // -- the next square geometry of 128 / 256 / 512 rows = columns that holds the grid (PAD kernels where it is larger)
    const int ax_n = std::max(g.n0, g.n1) <= 128 ? 128 : (std::max(g.n0, g.n1) <= 256 ? 256 : 512);
    const bool ax1_geom = cp.allow_ax1 && std::max(g.n0, g.n1) <= 512 && g.n0 >= 32 && g.n1 >= 32 && ax_n / blc::WCOL <= cus;
    std::vector<int> lw(B, 0);
    cp.ckF.assign((size_t)T * B, (unsigned char)SRC_PREV);
    cp.ckB.assign((size_t)T * B, (unsigned char)SRC_PREV);
    for (int64_t b = 0; b < B; ++b) {
        if (prog.kindF[b] != SRC_PRIOR || prog.tapF0[b] >= 0 || prog.tapF1[b] >= 0) return false;
        // the chain's band: the kernel of the first step that filters (every filtering step must use the same one)
        int k0 = -1, k1 = -1;
        for (int64_t t = 1; t < T && k0 < 0; ++t) k0 = prog.tapF0[(size_t)t * B + b];
        for (int64_t t = 0; t + 1 < T && k0 < 0 && full; ++t) k0 = prog.tapB0[(size_t)t * B + b];
        if (ax1_geom) {
            for (int64_t t = 1; t < T && k1 < 0; ++t) k1 = prog.tapF1[(size_t)t * B + b];
            for (int64_t t = 0; t + 1 < T && k1 < 0 && full; ++t) k1 = prog.tapB1[(size_t)t * B + b];
        }
        // a step either continues from the previous state through the chain's band, or RESTARTS from the reset distribution (a
        // change point, transitionModels.py:300-312) -- through the band (the change point comes before the random walk in the
        // combined model's list) or unfiltered (it comes after: the walk's output is discarded)
        auto classify = [&](unsigned char kind, int t0, int t1, unsigned char &out) {
            // (k1 = -1 unless the both-axes kernels may be planned; a restart passes through BOTH of the chain's bands or through none)
            if (kind == SRC_PREV && t0 == k0 && t1 == k1) { out = (unsigned char)SRC_PREV; return true; }
            if (kind == SRC_RESET && ((t0 == k0 && t1 == k1) || (t0 < 0 && t1 < 0))) {
                const bool filters = k0 >= 0 || k1 >= 0;
                out = (unsigned char)(SRC_RESET | ((filters && t0 < 0 && t1 < 0) ? 0x80 : 0));          // bit 7: no filter at this step
                cp.has_reset = true;
                if (filters) cp.mixed = true;
                return true;
            }
            return false;
        };
        for (int64_t t = 1; t < T; ++t) {
            const size_t k = (size_t)t * B + b;
            if (!classify(prog.kindF[k], prog.tapF0[k], prog.tapF1[k], cp.ckF[k])) return false;
        }
        if (full) {
            const size_t kl = (size_t)(T - 1) * B + b;
            if (prog.kindB[kl] != SRC_UNIFORM || prog.tapB0[kl] >= 0 || prog.tapB1[kl] >= 0) return false;
            for (int64_t t = 0; t < T - 1; ++t) {
                const size_t k = (size_t)t * B + b;
                if (!classify(prog.kindB[k], prog.tapB0[k], prog.tapB1[k], cp.ckB[k])) return false;
            }
        }
        cp.tap_id[b] = k0;
        cp.tap_id1[b] = k1;
        lw[b] = k0 >= 0 ? taps.lw[k0] : 0;
        if (lw[b] > cp.r0_max || lw[b] >= g.n0) return false;          // (single-period reflection)
        if (k1 >= 0) {
            cp.ax1 = true;
            if (taps.lw[k1] >= g.n1) return false;
            lw[b] = std::max(lw[b], taps.lw[k1]);                       // (one ring length for both filters: the wider walk's)
        }
    }
    if (cp.ax1) {
        // the transposing kernels: bands of radius <= 40 on either axis (ring lengths 8 .. 24 in steps of 4; the band's rounded radius inside
        // the grid: single-period reflection); the square geometry replaces the strip geometry planned above
        for (int64_t b = 0; b < B; ++b) if (lw[b] > FAST_R0_MAX || (std::max(8, (lw[b] + 7) / 8 * 8)) >= std::min(g.n0, g.n1)) return false;
        cp.n0p = cp.n1p = ax_n;
        cp.strips = ax_n / blc::WCOL;
        cp.ntw = ax_n / (blc::NW * blc::TM);
        cp.pad = ax_n != g.n0 || ax_n != g.n1;
        cp.cpr = cus / cp.strips;
    }
    cp.order.resize(B);
    for (int64_t b = 0; b < B; ++b) cp.order[b] = (int)b;
    std::stable_sort(cp.order.begin(), cp.order.end(), [&](int a, int c) { return lw[a] < lw[c]; });
    cp.round_start.clear(); cp.round_nk.clear();
    for (int64_t s0 = 0; s0 < B; s0 += cp.cpr) {
        const int64_t s1 = std::min<int64_t>(B, s0 + cp.cpr);
        int r0 = std::max(4, (lw[cp.order[s1 - 1]] + 3) / 4 * 4);             // (a product costs 64 cycles: bands as narrow as the widest chain of the launch allows)
        if (cp.ax1) r0 = std::max(8, (r0 + 7) / 8 * 8);                        // (ring lengths 8, 12, .. 24)
        cp.round_start.push_back((int)s0);
        cp.round_nk.push_back((prog.LW0 == 0 && !cp.ax1) ? 4 : (blc::TM + 2 * r0) / 4);          // (4: the no-stencil kernel)
    }
    cp.round_start.push_back((int)B);
    return true;
}

// Undo the lagged scale of the time-resident kernel (blhip_resident.hpp): its step k divided by the sum of step k - lag, so its row
// sums are S_k; the reference's normaliser is norm_k = S_k / (S_{k-1} s_k), s_k = k >= lag ? 1 / S_{k-lag} : 1.  The sums of every step
// are rewritten to what the launch-per-step kernels (lag 1) would have reported; rowsum keeps S_k, the normaliser of the stored row.
// false: a sum near the bottom / top of the fp64 range (a run of extreme outliers times the lag) -> the caller falls back to the
// launch-per-step kernels, whose magnitudes are the reference's.
bool resident_unlag(double *redF, int64_t T, int lag, std::vector<double> &rowsum, int64_t B = 1, int64_t b = 0) {
    rowsum.assign(T, 0.0);
    for (int64_t t = 0; t < T; ++t) rowsum[t] = redF[((size_t)t * B + b) * NRED];
    for (int64_t t = 0; t < T; ++t) {
        const double St = rowsum[t];
        if (!(St > 1e-150 && St < 1e150)) return false;
        const double sk = t >= lag ? 1.0 / rowsum[t - lag] : 1.0;
        const double norm = t == 0 ? St : St / (rowsum[t - 1] * sk);
        double *r = &redF[((size_t)t * B + b) * NRED];
        r[0] = norm; r[3] *= norm / St; r[4] *= norm / St;
    }
    return true;
}

// The same for the chain-resident kernel (blhip_chainres.hpp), whose step k divides by the normaliser of step k - lag:
// s_k = S_(k-lag-1) s_(k-lag) / S_(k-lag)  (1 while k < lag; S_(-1) = 1).
// kinds (may be null): a step whose source kind is not SRC_PREV consumed a distribution of known mass instead of the previous
// state: its normaliser is S_k / s_k.
bool chain_unlag(double *redF, int64_t T, int lag, std::vector<double> &rowsum, int64_t B, int64_t b, std::vector<double> *scales = nullptr,
                 const unsigned char *kinds = nullptr, int64_t t0 = 0) {
    // t0 > 0 (blc::ChainParams::skip_prefix): the chain's own pass began at step t0 -- the rows before it are another chain's (same
    // values, that chain's scale history), the scale history of the rows from t0 on starts there (s = 1 for lag steps, S_(t0 - 1) := 1)
    rowsum.assign(T, 0.0);
    for (int64_t t = 0; t < T; ++t) rowsum[t] = redF[((size_t)t * B + b) * NRED];
    std::vector<double> s_local;
    std::vector<double> &s = scales ? *scales : s_local;
    s.assign(T, 1.0);
    for (int64_t t = 0; t < T; ++t) {
        const double St = rowsum[t];
        if (!(St > 1e-150 && St < 1e150)) return false;
        const int64_t base = t >= t0 ? t0 : 0;                                 // first step of the scale history this step belongs to
        if (t - base >= lag) s[t] = (t - lag - 1 >= base ? rowsum[t - lag - 1] : 1.0) * s[t - lag] / rowsum[t - lag];
        const bool fresh = t == 0 || (kinds && kinds[(size_t)t * B + b] != SRC_PREV);
        const double norm = fresh ? St / s[t] : St / (rowsum[t - 1] * s[t]);
        double *r = &redF[((size_t)t * B + b) * NRED];
        r[0] = norm; r[3] *= norm / St; r[4] *= norm / St;
    }
    return true;
}

// evidence bookkeeping of the forward pass on the host, in the reference's order (core.py:385-404, 417); K > 1: raw sums of the
// K-steps-per-launch 1-D kernels.  -> false if such a raw sum came near the bottom of the fp64 range (the caller repeats with K = 1)
bool forward_bookkeeping(const blhip_problem *p, const ChainProgram &prog, const double *redF, int64_t B, double dV, bool fused1d, int64_t K,
                         bool evidence_only, bool forward_only, BatchOutcome &O) {
    const int64_t T = p->T;
    O.logE.assign(B, 0.0);
    O.abort_step.assign(B, -1);
    O.abort_phase.assign(B, 0);
    O.local.assign((size_t)B * T, 0.0);
    bool raw_ok = true;
    // (step by step over all chains: the sums of a step are B consecutive records -- chain by chain every read was a cache line of
    //  its own, 8 ms of the published break-point study's 23 batches of 1017 chains x 41 steps.  Per chain the order of the
    //  operations is the one of the reference's loop)
    std::vector<double> &le = O.logE;
    for (int64_t t = 0; t < T; ++t) {
        const double *rt = redF + (size_t)t * B * NRED;
        for (int64_t b = 0; b < B; ++b) {
            if (O.abort_step[b] >= 0) continue;
            double norm = rt[b * NRED + 0];
            if (fused1d) {
                // raw sums of the K-step launches (blhip_fused1d.hpp): inner steps carry the scale of their predecessor
                if (!(norm > 1e-200)) raw_ok = false;
                if (t % K != 0 && prog.kindF[(size_t)t * B + b] == SRC_PREV) norm /= redF[((size_t)(t - 1) * B + b) * NRED];
            }
            // RegimeSwitch renormalises the clamped prior (transitionModels.py:410): alpha = (u / sum u) L
            if (prog.has_clamp && prog.cmodeF[(size_t)t * B + b]) norm /= rt[b * NRED + 1];
            if (!(norm > 0.0)) { O.abort_step[b] = t; O.abort_phase[b] = 0; le[b] = -INFINITY; continue; }
            le[b] += std::log(norm);
            O.local[(size_t)b * T + t] = norm * dV;
        }
    }
    const double ldv = std::log(dV);
    for (int64_t b = 0; b < B; ++b)
        if (O.abort_step[b] < 0) le[b] += ldv;
    O.means.clear();
    if (!evidence_only) O.means.assign((size_t)B * p->ndim * T, 0.0);
    if (forward_only) {
        for (int64_t b = 0; b < B; ++b)
            for (int64_t t = 0; t < T; ++t) {
                const double *r = &redF[((size_t)t * B + b) * NRED];
                for (int k = 0; k < p->ndim; ++k) O.means[((size_t)b * p->ndim + k) * T + t] = r[3 + k] / r[0];
            }
    }
    return raw_ok;
}

// bookkeeping of the backward pass (core.py:441-464, 480-483): abort test, local evidence, row normalisers, posterior means.
// rows_done_from >= 0: rows t >= rows_done_from were normalised by the resident kernel itself (their invN is 1).
bool backward_bookkeeping(const blhip_problem *p, const ChainProgram &prog, const double *redF, const double *redB, int64_t B, double dV,
                          bool fused1d, int64_t rows_done_from, BatchOutcome &O) {
    const int64_t T = p->T;
    bool raw_ok = true;
    for (int64_t t = T - 1; t >= 0; --t) {           // (step by step over all chains: see forward_bookkeeping)
        for (int64_t b = 0; b < B; ++b) {
            if (O.abort_step[b] >= 0) continue;
            const double *r = &redB[((size_t)t * B + b) * NRED];
            if (fused1d && !(r[0] > 1e-200)) raw_ok = false;
            // The reference tests sum(alpha_norm * beta_norm) > 0 (core.py:441).  r[0] is the same sum up to the lazily
            // dropped normalisers, which are positive -- except with signed kernels (Deterministic's cubic-spline
            // shift, AlphaStable's FFT kernel): there sum(alpha) = redF[t][0] and sum(beta) = r[5] may be negative and
            // the reference divides by them, so the sign test has to include them.
            double refnorm = r[0];
            if (prog.has_clamp) refnorm = r[0] / (redF[((size_t)t * B + b) * NRED] * (prog.cmodeB[(size_t)t * B + b] ? r[5] : 1.0));
            if (!(refnorm > 0.0)) { O.abort_step[b] = t; O.abort_phase[b] = 1; O.logE[b] = -INFINITY; continue; }
            O.local[(size_t)b * T + t] = 1.0 / ((r[1] / r[0]) * dV);                      // core.py:463-464
            O.invN[(size_t)b * T + t] = (rows_done_from >= 0 && t >= rows_done_from) ? 1.0 : 1.0 / r[0];
            for (int k = 0; k < p->ndim; ++k) O.means[((size_t)b * p->ndim + k) * T + t] = r[3 + k] / r[0];
        }
    }
    return raw_ok;
}

// ---- what the launches move and compute BY CONSTRUCTION (blhip_timing: fwd / bwd _hbm_bytes, _flops) ------------------------------------
// fp64 flop per cell of the fused epilogues (FMA = 2; ldexp, compare and select count 1): forward  a = v L (1), sum (1), the two
// recurrence products (2), ldexp (1) + the exponentials of the anchors spread over their rows (2 x ~42 flop per 16 rows: 5);
// backward: beta, p, c (3), p / L by the reciprocal recurrence (2), three sums (3), the fold's product, max, add (3), four recurrence
// products (4), ldexp (1) + four exponentials per 16 rows (10)
constexpr double EPI_FWD_FLOP = 10.0, EPI_BWD_FLOP = 26.0;
// a radius-r stencil pass per cell: on the vector ALU (SciPy's pair order) r adds + 1 product + r FMAs; as a banded product on the
// matrix pipe (16 output rows per tile) 16 + 2 r products, zeros of the band included
inline double valu_stencil_flop(int r) { return r > 0 ? 3.0 * r + 1.0 : 0.0; }
inline double band_stencil_flop(int r) { return 2.0 * (16.0 + 2.0 * r); }
inline void account(blhip_ctx *ctx, bool bwd, double bytes, double flops) {
    (bwd ? ctx->timing.bwd_hbm_bytes : ctx->timing.fwd_hbm_bytes) += bytes;
    (bwd ? ctx->timing.bwd_flops : ctx->timing.fwd_flops) += flops;
}

#include "blhip_fit_nd.hpp"       // do_fit_nd: grids with 3 and 4 parameters
#include "blhip_fit_paths.hpp"    // BatchEnv, ResidentRun, ChainRun: the resident paths of a batch

void do_fit(blhip_ctx *ctx, const blhip_problem *p_in, int64_t n_chains, const double *op_values,
            const double *log_w, uint32_t flags, blhip_result *res) {
    Trace tr(ctx->option("trace", 0.0) != 0.0);
    validate(p_in, n_chains, op_values);
    if (p_in->ndim > 2) { ctx->prior_token = 0; do_fit_nd(ctx, p_in, n_chains, op_values, log_w, flags, res); return; }      // (that path lays the tables buffer out its own way)
    // closed-form models without an in-kernel likelihood: their (T, G) table is built on the device, the step kernels
    // then see a tabulated likelihood
    blhip_problem p_local = *p_in;
    const int table_model = (p_in->obs_model >= BLHIP_OM_BERNOULLI && p_in->obs_model <= BLHIP_OM_SCALED_AR1) ? p_in->obs_model : 0;
    if (table_model) p_local.obs_model = BLHIP_OM_TABLE;
    const blhip_problem *p = &p_local;
    HIPCHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const FitFlags ff = decode_flags(ctx, p, flags, log_w);
    const bool evidence_only = ff.evidence_only, forward_only = ff.forward_only, full = ff.full, keep = ff.keep,
               accumulate = ff.accumulate, resume = ff.resume, carry = ff.carry;
    const int64_t T = p->T;

    Geometry g{};
    if (p->ndim == 1) { g.n0 = 1; g.n1 = (int)p->n[0]; g.axis_map[0] = 1; g.axis_map[1] = 1; }
    else { g.n0 = (int)p->n[0]; g.n1 = (int)p->n[1]; g.axis_map[0] = 0; g.axis_map[1] = 1; }
    g.G = (long long)g.n0 * g.n1;
    const long long G = g.G;
    if (accumulate && (ctx->acc_T != T || ctx->acc_G != G)) fail("accumulator shape mismatch");
    double dV = 1.0;
    for (int k = 0; k < p->ndim; ++k) dV *= p->lattice[k];

    ctx->post_valid = false;
    ctx->timing = blhip_timing{};
    ctx->resident_last_reason = 0;
    // a context whose resident launch once gave up tries the resident paths again after a while (one hiccup -- another process
    // holding CUs -- must not cost 1.4 - 2 x for the life of the process); the wait doubles with every give-up in a row
    if (!ctx->resident_ok && ++ctx->resident_fits_since >= ctx->resident_retry_after) {
        ctx->resident_ok = true;
        ctx->resident_fits_since = 0;
        ctx->probe_last = -1.0;                      // (re-armed: ask the chip first)
    }
    probe_residency(ctx);
    ctx->timing.xcd_order = ctx->xcd_order_ok ? 1 : 0;

    // ---- shared tables -> HBM ---------------------------------------------------------------------------------------------------
    const DeviceTables DT = upload_tables(ctx, p, g, ff, table_model);
    double *const d_m0 = DT.m0, *const d_m1 = DT.m1, *const d_colA = DT.colA, *const d_colB = DT.colB, *const d_rec = DT.rec;
    double *const d_prior = DT.prior, *const d_reset = DT.reset, *const d_uniform = DT.uniform, *const d_indep = DT.indep, *const d_lik = DT.lik;
    const int rec_len = DT.rec_len, d = DT.d;
    tr.mark("tables + H2D");

    // ---- memory plan ------------------------------------------------------------------------------------------------------------
    // The average posterior of a hyper-study is folded batch by batch (core.py:1358-1366): a pass over the batch's whole sequence at
    // the memory roof (C4: 2 x 24 ms of a 300 ms fit).  Option accum_overlap = 1 (default 0): two sequence buffers, the fold of batch b
    // runs on a second stream beside the forward pass of batch b + 1 (~4 batches instead of as few as fit).  Measured on C4: no
    // gain -- queued behind the pass's launches the fold is not started before they have finished (rocprofv3 time line), queued
    // ahead of them it slows the pass by about its own duration (13 ms fold: forward 25 -> 35 ms per batch; 1.15e11 either way).
    const bool overlap_acc = accumulate && full && ctx->option("accum_overlap", 0.0) != 0.0 && n_chains >= 64;
    if (!overlap_acc) ctx->post2.release();         // (a fit without the second stream gets the memory back)
    int64_t Bmax = chains_per_batch(ctx, p, g, ff, n_chains, overlap_acc ? 2 : 1);
    if (overlap_acc) Bmax = std::min<int64_t>(Bmax, std::max<int64_t>(32, ((n_chains + 3) / 4 + 31) / 32 * 32));
    // ---- batches: at most Bmax chains each, cut where the axis-0 radius bucket changes.  A bucket cut by a batch boundary becomes
    //      two under-filled launches per step (round 1: the 107-chain radius-24 bucket of the C4 study ran as 36 + 71 chains at
    //      4.1 TB/s); hyper-grids are usually monotone in the random-walk width, so contiguous cuts suffice.
    // (cuts on radius-bucket boundaries serve the launch-per-step kernels; grids the chain-resident kernel takes keep whole launches)
    const bool chain_shape = p->ndim == 2 && (p->obs_model == BLHIP_OM_GAUSSIAN || (p->obs_model == BLHIP_OM_TABLE && g.n0 <= 512)) && chain_rows_ok(g.n0) && g.n1 <= 16 * blc::MAX_STRIPS &&
                             ctx->option("chain_resident", 1.0) != 0.0 && ctx->resident_ok;
    std::vector<int64_t> batch_start = plan_batches(p, n_chains, op_values, Bmax, !overlap_acc && !chain_shape);
    if (overlap_acc) {                     // (equal batches, multiples of 32 chains: whole launches of the chain-resident kernel)
        batch_start.clear();
        for (int64_t c = 0; c < n_chains; c += Bmax) batch_start.push_back(c);
        batch_start.push_back(n_chains);
    }
    // (the cut: 40; grids the chain-resident kernels take: 80)
    const bool chain_wide = chain_shape && p->obs_model == BLHIP_OM_GAUSSIAN && ctx->option("chain_wide", 1.0) != 0.0;
    if (!overlap_acc && !ff.keep && !ff.resume && !ff.carry && ctx->option("wide_v", 1.0) != 0.0)
        split_wide_axis0(p, n_chains, op_values, batch_start, chain_wide ? CHAIN_R0_MAX : FAST_R0_MAX);
    const int64_t nbatch = (int64_t)batch_start.size() - 1;
    ctx->timing.batches = nbatch;

    tr.mark("memory budget");
    hipEvent_t *ev = ctx->ev;
    HIPCHECK(hipEventRecord(ev[6], st));

    double *redF = nullptr, *redB = nullptr;   // reduced sums on the host (page-locked staging of the context)
    std::vector<hipEvent_t> fold_ev;           // overlapped folds: start / end events of every batch (timing)
    FoldJob fold_job;                          // the fold of the previous batch, launched once this batch's forward pass is queued
    auto launch_pending_fold = [&]() {
        if (!fold_job.pending) return;
        hipEvent_t e0, e1;
        HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
        fold_ev.push_back(e0); fold_ev.push_back(e1);
        launch_fold(ctx, T, G, fold_job, ctx->astream, e0, e1);
        HIPCHECK(hipEventRecord(ctx->aev_done[fold_job.parity], ctx->astream));
        fold_job.pending = false;
    };
    if (overlap_acc && !ctx->astream) {
        HIPCHECK(hipStreamCreateWithFlags(&ctx->astream, hipStreamNonBlocking));
        for (auto &e : ctx->aev_done) HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    // The per-(step, chain) programs of batch bi + 1 are built by a second host thread while batch bi runs -- its passes on the GPU, its
    // bookkeeping on this thread (the reference's published break-point study: 23 batches of 1017 chains x 41 steps, 23 - 30 ms of
    // build_program of a 90 - 130-ms fit; the builder touches only its own BatchProgram and the caller's read-only inputs, no HIP call).
    // A failure of the early build is raised where the build used to be: at the top of that batch.
    struct BatchProgram { TapTable taps; ChainProgram prog; bool ready = false; std::exception_ptr err; };
    std::unique_ptr<BatchProgram> bprog[2];
    auto build_batch = [&](int64_t bj) {
        std::unique_ptr<BatchProgram> &N = bprog[bj & 1];
        N.reset(new BatchProgram());
        try { build_program(p, g, batch_start[bj], batch_start[bj + 1] - batch_start[bj], op_values, N->taps, N->prog, resume); }
        catch (...) { N->err = std::current_exception(); }
        N->ready = true;
    };
    std::thread builder;
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } builder_guard{builder};      // (also when a batch throws)
    const bool build_ahead = nbatch > 1;
    for (int64_t bi = 0; bi < nbatch; ++bi) {
        const int64_t c0 = batch_start[bi], B = batch_start[bi + 1] - c0;
        tr.mark("batch setup");
        if (builder.joinable()) builder.join();
        if (!bprog[bi & 1] || !bprog[bi & 1]->ready) build_batch(bi);
        BatchProgram &BP = *bprog[bi & 1];
        BP.ready = false;                        // (consumed: the slot is rebuilt for batch bi + 2)
        if (BP.err) std::rethrow_exception(BP.err);
        TapTable &taps = BP.taps;
        ChainProgram &prog = BP.prog;
        if (build_ahead && bi + 1 < nbatch) {
            try { builder = std::thread(build_batch, bi + 1); }
            catch (const std::system_error &) {}        // (no thread to be had: the batch is built at its own top, as before)
        }
        tr.mark("build_program");
        const GeometryPlan gp = plan_geometry(ctx, p, g, prog, B, d, resume, carry, &taps);
        const bool fast = gp.fast, fused1d = gp.fused1d, use_mfma = gp.use_mfma;
        const int64_t fusedK = gp.fusedK;
        const int f1_TJ = gp.f1_TJ;
        Tile tile = gp.tile;
        const int fastS = gp.fastS, fast_nseg = gp.fast_nseg, fast_fnblk = gp.fast_fnblk, mS = gp.mS, m_nseg = gp.m_nseg, m_tiles_j = gp.m_tiles_j,
                  m_nblk = gp.m_nblk;
        auto f1_lds = [&](int64_t K) { return (size_t)(4 * (f1_TJ + 2 * K * prog.LW1) + K * f1_TJ + K * (prog.LW1 + 1) + K * rec_len + 4 * 8 + 2 + K + 8 + K * 3 * f1_TJ) * sizeof(double); };
        ctx->timing.fwd_kernel_variant = ctx->timing.bwd_kernel_variant = fast ? 1 : (fused1d ? 4 : 0);
        ctx->timing.cells_per_launch = std::max<int64_t>(ctx->timing.cells_per_launch, B * G);

        // --- device metadata ---
        DeviceMeta M;
        upload_metadata(ctx, p, prog, taps, B, full, fast, tile.nblk, M, gp.wideH, gp.wideV, gp.hSplit, gp.hFusedMax);
        unsigned char *const d_kindF = M.kindF, *const d_kindB = M.kindB, *const d_cmodeF = M.cmodeF, *const d_cmodeB = M.cmodeB;
        double *const d_limitF = M.limitF, *const d_limitB = M.limitB;
        int *const d_tapF0 = M.tapF0, *const d_tapF1 = M.tapF1, *const d_tapB0 = M.tapB0, *const d_tapB1 = M.tapB1;
        int *const d_orderF = M.orderF, *const d_orderB = M.orderB;
        double *const d_taps = M.taps;
        int *const d_off = M.off, *const d_lw = M.lw, *const d_lw2 = M.lw2;
        double *const d_invN = M.invN, *const d_w = M.w, *const d_dump = M.dump;
        const std::vector<int> &orderF = M.h_orderF, &orderB = M.h_orderB;
        const std::vector<std::vector<FastRange>> &rangesF = M.rangesF, &rangesB = M.rangesB;

        tr.mark("buckets + metadata H2D");
        // --- state ---
        size_t psz = (size_t)T * B * NRED * tile.nblk;
        ctx->psumF.ensure(psz * 8);
        ctx->redF.ensure((size_t)T * B * NRED * 8);
        double *d_psF = ctx->psumF.as<double>();
        double *d_post = nullptr, *d_pp[2] = {nullptr, nullptr};
        ctx->state.ensure((size_t)2 * B * G * 8);
        d_pp[0] = ctx->state.as<double>();
        d_pp[1] = d_pp[0] + (size_t)B * G;
        if (!evidence_only) {
            DevBuf &pb = (overlap_acc && (bi & 1)) ? ctx->post2 : ctx->post;
            pb.ensure((size_t)B * T * G * 8);
            d_post = pb.as<double>();
            // the fold of batch bi - 2 read this buffer on the second stream: it has to be through before this batch writes it
            if (overlap_acc && bi >= 2) HIPCHECK(hipStreamWaitEvent(st, ctx->aev_done[bi & 1], 0));
        }

        // BLHIP_RESUME: step 0 reads each chain's carried (normalised) state; its "previous partial sums" add up to 1
        const double *d_carry_src = nullptr, *d_unit = nullptr;
        if (resume) {
            d_carry_src = ctx->carry[p->carry_slot].buf.as<double>();
            std::vector<double> unit((size_t)B * NRED * tile.nblk, 0.0);
            const std::vector<double> &mv = ctx->carry[p->carry_slot].maxv;
            for (int64_t b = 0; b < B; ++b)
                for (int k = 0; k < NRED; ++k)       // slot 6 = maximum of the carried state (NotEqual inverts around it)
                    unit[((size_t)b * NRED + k) * tile.nblk] = (k == 6 && (int64_t)mv.size() == B) ? mv[b] : 1.0;
            ctx->unit.ensure(unit.size() * 8);
            HIPCHECK(hipMemcpyAsync(ctx->unit.p, unit.data(), unit.size() * 8, hipMemcpyHostToDevice, st));
            sync_stream(ctx, st);
            d_unit = ctx->unit.as<double>();
        }
        StepParams P{};
        P.n0 = g.n0; P.n1 = g.n1; P.TI = tile.TI; P.TJ = tile.TJ; P.LW0 = tile.LW0; P.LW1 = tile.LW1;
        P.tiles_i = tile.tiles_i; P.tiles_j = tile.tiles_j; P.nblk = tile.nblk; P.ndim = p->ndim; P.d = d;
        P.rec_len = rec_len; P.shared[SRC_PREV] = nullptr; P.shared[SRC_PRIOR] = d_prior; P.shared[SRC_RESET] = d_reset;
        P.shared[SRC_UNIFORM] = d_uniform; P.shared[SRC_INDEP] = d_indep; P.taps = d_taps; P.tap_off = d_off; P.tap_lw = d_lw; P.tap_lw2 = d_lw2;
        P.m0 = d_m0; P.m1 = d_m1; P.colA = d_colA; P.colB = d_colB; P.chains = (int)B;
        P.prev_nblk = tile.nblk;

        blf::FastParams FP{};
        if (fast) {
            FP.n0 = g.n0; FP.n1 = g.n1; FP.TJ = tile.TJ; FP.S = fastS; FP.nseg = fast_nseg;
            FP.tiles_j = tile.tiles_j; FP.nblk = tile.nblk; FP.fnblk = fast_fnblk;
            FP.dump = d_dump;
            FP.mlean = (g.n0 % mS == 0 && (double)G * 8.0 < 4.0e9 && g.n1 < (1 << 20) && g.n0 < (1 << 24)) ? 1 : 0;
            FP.mS = mS; FP.mnseg = m_nseg; FP.mtiles_j = m_tiles_j; FP.mnblk = m_nblk;
            FP.ndim = p->ndim; FP.d = d; FP.means = forward_only ? 1 : 0;
            FP.shared[SRC_PREV] = nullptr; FP.shared[SRC_PRIOR] = d_prior; FP.shared[SRC_RESET] = d_reset;
            FP.shared[SRC_UNIFORM] = d_uniform; FP.shared[SRC_INDEP] = d_indep; P.shared[SRC_INDEP] = d_indep; FP.taps = d_taps; FP.tap_off = d_off; FP.tap_lw = d_lw;
            FP.m0 = d_m0; FP.m1 = d_m1; FP.colA = d_colA; FP.colB = d_colB; FP.prev_nblk = tile.nblk;
            // likelihood recurrence along rows needs an equally spaced row axis (to rounding)
            const double *m0h = p->marginal[0];
            const double step = (m0h[g.n0 - 1] - m0h[0]) / (double)(g.n0 - 1);
            double dev = 0.0, mx = 0.0;
            for (int i = 0; i < g.n0; ++i) {
                dev = std::max(dev, std::fabs(m0h[i] - (m0h[0] + i * step)));
                mx = std::max(mx, std::fabs(m0h[i]));
            }
            FP.step0 = step;
            FP.use_rec = (p->obs_model == BLHIP_OM_GAUSSIAN && dev <= 8.0 * 2.3e-16 * mx && ctx->option("recurrence", 1.0) != 0.0) ? 1 : 0;
        }
        // ---- the resident paths (single chain: blhip_resident.hpp; batches of chains: blhip_chainres.hpp); the launch-per-step kernels
        //      below are their fall-back ---------------------------------------------------------------------------------------------------
        BatchEnv E{};
        E.ctx = ctx; E.p = p; E.st = st; E.g = g; E.G = G; E.T = T; E.B = B; E.c0 = c0; E.d = d; E.rec_len = rec_len; E.ff = ff;
        E.DT = &DT; E.M = &M; E.prog = &prog; E.taps = &taps; E.step0 = FP.step0; E.d_post = d_post; E.log_w = log_w;
        E.chain_means = res && res->posterior_mean;
        E.overlap_acc = overlap_acc;
        ResidentRun RR;
        RR.setup(E, n_chains, fast, FP.use_rec != 0, psz);
        ChainRun CR;
        if (!RR.on) CR.setup(E, fast, FP.use_rec != 0, psz);
        if (RR.on || CR.on) { ctx->psumF.ensure(psz * 8); d_psF = ctx->psumF.as<double>(); }
        if (CR.on && CR.depad) {                 // a padded batch that hands its posteriors out: scratch sequence for the kernels, d_post stays the result
            ctx->postpad.ensure((size_t)B * T * CR.Gk * 8);
            E.d_post = ctx->postpad.as<double>();
        } else if (CR.on && (CR.cp.pad || CR.ax1) && d_post) {      // the private sequence of a padded chain-resident batch lives on the padded geometry
            DevBuf &pb = (overlap_acc && (bi & 1)) ? ctx->post2 : ctx->post;
            pb.ensure((size_t)B * T * CR.Gk * 8);
            d_post = pb.as<double>();
            E.d_post = d_post;
        }
        const bool resident = RR.on, chainres = CR.on;
        bool resident_failed = false;

        // bucket streams: fork = every bucket stream waits for the main stream; join = the main stream waits for all of them
        // (only worth it when a step really has several launches: a launch on a secondary stream costs ~9 us more
        //  than a back-to-back launch on the main stream -- measured on single-chain fits)
        size_t max_ranges = 0;
        for (const auto &r : rangesF) max_ranges = std::max(max_ranges, r.size());
        for (const auto &r : rangesB) max_ranges = std::max(max_ranges, r.size());
        const bool multistream = fast && max_ranges >= 2;
        auto fork_streams = [&]() {
            if (!multistream) return;
            HIPCHECK(hipEventRecord(ctx->fork_ev, st));
            for (auto &bs : ctx->bstream) HIPCHECK(hipStreamWaitEvent(bs, ctx->fork_ev, 0));
        };
        auto join_streams = [&]() {
            if (!multistream) return;
            for (int k = 0; k < blhip_ctx::NBS; ++k) {
                HIPCHECK(hipEventRecord(ctx->bev[k], ctx->bstream[k]));
                HIPCHECK(hipStreamWaitEvent(st, ctx->bev[k], 0));
            }
        };
        // a chain only depends on its own previous step: streams need a barrier only where bucket membership changes
        auto same_membership = [&](const std::vector<int> &order, const std::vector<std::vector<FastRange>> &ranges,
                                   int64_t ta, int64_t tb) {
            if (ranges[ta].size() != ranges[tb].size()) return false;
            for (size_t k = 0; k < ranges[ta].size(); ++k)
                if (ranges[ta][k].key != ranges[tb][k].key || ranges[ta][k].count != ranges[tb][k].count) return false;
            return std::equal(order.begin() + ta * B, order.begin() + (ta + 1) * B, order.begin() + tb * B);
        };
        std::vector<char> any_hF, any_hB, any_vF, any_vB;
        double *d_hsrc = nullptr, *d_vsrc = nullptr;
        if (gp.wideH || gp.wideV) {
            any_hF.assign(T, 0); any_hB.assign(T, 0); any_vF.assign(T, 0); any_vB.assign(T, 0);
            for (int64_t t = 0; t < T; ++t)
                for (int64_t b = 0; b < B; ++b) {
                    if (gp.wideH && prog.tapF1[t * B + b] >= 0) any_hF[t] = 1;
                    if (gp.wideH && full && prog.tapB1[t * B + b] >= 0) any_hB[t] = 1;
                    if (gp.wideV && prog.tapF0[t * B + b] >= 0) any_vF[t] = 1;
                    if (gp.wideV && full && prog.tapB0[t * B + b] >= 0) any_vB[t] = 1;
                }
            ctx->hsrc.ensure((size_t)B * G * 8 * ((gp.wideH && gp.wideV) ? 2 : 1));
            d_hsrc = ctx->hsrc.as<double>();
            d_vsrc = (gp.wideH && gp.wideV) ? d_hsrc + (size_t)B * G : d_hsrc;
        }
        constexpr int mfma_min_r0 = 8;
        // both-axes launches: the matrix-pipe kernel wins while the launch is latency-bound (few cells per CU); with the chip
        // full the vector kernel's 2 x 17 FMAs per cell beat 2 x 32 band products (measured: 1024^2 12.7 vs 14.9 us,
        // 2048^2 30.0 vs 27.7 us, 4096^2 97 vs 77 us per forward step)
        const bool mfma_h = ctx->option("mfma_h", 1.0) != 0.0;
        const double mfma_h_max_cells = ctx->option("mfma_h_max_cells", 2.5e6);
        long long n_mfma[2] = {0, 0}, n_fast[2] = {0, 0};
        constexpr bool uniform_launch = true;
        auto run_step = [&](int mode, int64_t t, const double *srcp, long long src_stride, double *dstp, long long dst_stride,
                            double *postp, long long post_stride, const double *ps_prev, int prev_slot, double *ps_out,
                            bool means) {
            if (fast) {
                blf::FastParams Q = FP;
                Q.src = srcp; Q.src_stride = src_stride; Q.dst = dstp; Q.dst_stride = dst_stride;
                Q.post = postp; Q.post_stride = post_stride;
                Q.srckind = (mode == MODE_FWD ? d_kindF : d_kindB) + t * B;
                Q.tap0 = (mode == MODE_FWD ? d_tapF0 : d_tapB0) + t * B;
                Q.tap1 = (mode == MODE_FWD ? d_tapF1 : d_tapB1) + t * B;
                Q.psum_prev = ps_prev; Q.prev_slot = prev_slot; Q.psum_out = ps_out;
                Q.rec = d_rec + t * rec_len; Q.lik = d_lik ? d_lik + (size_t)t * G : nullptr;
                const int *ord = (mode == MODE_FWD ? d_orderF : d_orderB) + t * B;
                const bool prepass_step = gp.wideH && (mode == MODE_FWD ? any_hF : any_hB)[t];
                const bool prepass_v = gp.wideV && (mode == MODE_FWD ? any_vF : any_vB)[t];
                for (const FastRange &r : (mode == MODE_FWD ? rangesF[t] : rangesB[t])) {
                    Q.chain_ids = ord + r.start;
                    Q.u_valid = 0;
                    Q.hsrc = nullptr;
                    const bool prepass = prepass_step && r.pre;
                    if (r.count == 1 && uniform_launch) {      // single-chain launch: hand the chain's metadata over by value
                        const int64_t tb = t * B;
                        const int cb = (mode == MODE_FWD ? orderF : orderB)[tb + r.start];
                        const int k0 = (mode == MODE_FWD ? prog.tapF0 : prog.tapB0)[tb + cb];
                        const int k1 = (mode == MODE_FWD ? prog.tapF1 : prog.tapB1)[tb + cb];
                        Q.u_valid = 1; Q.u_chain = cb; Q.u_kind = (mode == MODE_FWD ? prog.kindF : prog.kindB)[tb + cb];
                        Q.u_t0 = k0; Q.u_lw0 = k0 >= 0 ? taps.lw[k0] : 0; Q.u_off0 = k0 >= 0 ? taps.off[k0] : 0;
                        Q.u_t1 = k1; Q.u_lw1 = k1 >= 0 ? taps.lw[k1] : 0; Q.u_off1 = k1 >= 0 ? taps.off[k1] : 0;
                    }
                    hipStream_t ls = multistream ? ctx->bstream[r.key] : st;
                    if (prepass) {
                        // row filter of the bucket's chains -> hsrc, on the bucket's stream: the launch below consumes it instead of its
                        // sources; the buckets' streams overlap one bucket's (HBM-bound) pre-pass with another's (matrix-pipe-bound) step
                        blh::HParams HP{};
                        HP.n0 = g.n0; HP.n1 = g.n1; HP.tiles_j = (g.n1 + blh::CB - 1) / blh::CB; HP.lwmax = prog.LW1; HP.pitch = blh::pitch_for(prog.LW1);
                        HP.src = srcp; HP.src_stride = src_stride;
                        for (int k = 0; k < 5; ++k) HP.shared[k] = FP.shared[k];
                        HP.chain_ids = Q.chain_ids;
                        HP.srckind = Q.srckind; HP.tap1 = Q.tap1; HP.taps = d_taps; HP.tap_off = d_off; HP.tap_lw = d_lw; HP.dst = d_hsrc;
                        launch_hwide(ls, HP, r.count);
                        Q.hsrc = d_hsrc;
                        account(ctx, mode != MODE_FWD, (double)r.count * G * 16.0, (double)r.count * G * 2.0 * (2.0 * prog.LW1 + 8.0));
                    }
                    if (prepass_v) {
                        // column filter of the bucket's chains (after the row filter, if there is one) -> the launch below runs without a stencil
                        blh::HParams VP{};
                        VP.n0 = g.n0; VP.n1 = g.n1; VP.tiles_j = (g.n1 + blh::CV - 1) / blh::CV; VP.lwmax = prog.LW0;
                        VP.src = srcp; VP.src_stride = src_stride; VP.presrc = Q.hsrc;
                        for (int k = 0; k < 5; ++k) VP.shared[k] = FP.shared[k];
                        VP.chain_ids = Q.chain_ids;
                        VP.srckind = Q.srckind; VP.tap1 = Q.tap0; VP.taps = d_taps; VP.tap_off = d_off; VP.tap_lw = d_lw; VP.dst = d_vsrc;
                        launch_vwide(ls, VP, r.count);
                        Q.hsrc = d_vsrc;
                        account(ctx, mode != MODE_FWD, (double)r.count * G * 16.0, (double)r.count * G * 2.0 * (2.0 * prog.LW0 + 8.0));
                    }
                    const bool on_pipe = use_mfma && (r.H ? (mfma_h && (double)r.count * g.n0 * g.n1 <= mfma_h_max_cells) : r.R0 >= mfma_min_r0);
                    if (on_pipe) { launch_mfma(ls, p->obs_model, mode, Q, r.R0, r.H, r.count); ++n_mfma[mode == MODE_FWD ? 0 : 1]; }
                    else { launch_fast(ls, p->obs_model, mode, Q, r.R0, r.H, r.count); ++n_fast[mode == MODE_FWD ? 0 : 1]; }
                    // streaming kernels: state in, state out (+ stored alpha in, posterior out backward; + a tabulated likelihood)
                    const bool bw = mode != MODE_FWD;
                    const double fl = (on_pipe ? (r.R0 > 0 ? band_stencil_flop(r.R0) : 0.0) + (r.H ? band_stencil_flop(8) : 0.0)
                                               : valu_stencil_flop(r.R0) + (r.H ? valu_stencil_flop(8) : 0.0)) + (bw ? EPI_BWD_FLOP : EPI_FWD_FLOP);
                    account(ctx, bw, (double)r.count * G * ((bw ? 32.0 : 16.0) + (d_lik ? 8.0 : 0.0)), (double)r.count * G * fl);
                }
            } else {
                StepParams Q = P;
                Q.src = srcp; Q.src_stride = src_stride; Q.dst = dstp; Q.dst_stride = dst_stride;
                Q.post = postp; Q.post_stride = post_stride;
                Q.srckind = (mode == MODE_FWD ? d_kindF : d_kindB) + t * B;
                Q.tap0 = (mode == MODE_FWD ? d_tapF0 : d_tapB0) + t * B;
                Q.tap1 = (mode == MODE_FWD ? d_tapF1 : d_tapB1) + t * B;
                Q.cmode = prog.has_clamp ? (mode == MODE_FWD ? d_cmodeF : d_cmodeB) + t * B : nullptr;
                Q.limit = prog.has_clamp ? (mode == MODE_FWD ? d_limitF : d_limitB) + t * B : nullptr;
                Q.psum_prev = ps_prev; Q.prev_slot = prev_slot; Q.psum_out = ps_out;
                Q.rec = d_rec + t * rec_len; Q.lik = d_lik ? d_lik + (size_t)t * G : nullptr;
                launch_step(st, p->obs_model, Q, tile, (int)B, mode, means);
                const bool bw = mode != MODE_FWD;
                account(ctx, bw, (double)B * G * ((bw ? 32.0 : 16.0) + (d_lik ? 8.0 : 0.0)),
                        (double)B * G * (valu_stencil_flop(prog.LW0) + valu_stencil_flop(prog.LW1) + (bw ? EPI_BWD_FLOP : EPI_FWD_FLOP)));
            }
        };

        float ms = 0;
        BatchOutcome O;
        std::vector<double> &invN = O.invN;
        auto passes = [&](const int64_t K) -> bool {
        bl1f::F1Params F1{};
        if (fused1d) {
            F1.n = g.n1; F1.TJ = f1_TJ; F1.nblk = tile.nblk; F1.LW = prog.LW1; F1.T = (int)T; F1.B = (int)B; F1.d = d; F1.rec_len = rec_len;
            F1.shared[SRC_PREV] = nullptr; F1.shared[SRC_PRIOR] = d_prior; F1.shared[SRC_RESET] = d_reset;
            F1.shared[SRC_UNIFORM] = d_uniform; F1.shared[SRC_INDEP] = d_indep;
            F1.taps = d_taps; F1.tap_off = d_off; F1.tap_lw = d_lw; F1.m1 = d_m1; F1.colA = d_colA; F1.rec = d_rec; F1.lik = d_lik;
            F1.cmode = nullptr; F1.tap_lw2 = d_lw2;
        }
        tr.mark("path setup");
        // --- forward pass (core.py:372-411) ---
        // the previous batch's fold goes to the second stream FIRST: queued behind this pass's launches it was not started before
        // they had all finished (measured: rocprofv3 time line), queued ahead of them it shares the chip with them
        launch_pending_fold();
        HIPCHECK(hipEventRecord(ev[0], st));
        // 1-D grids whose blocks all fit on the chip at once: ONE persistent launch per pass (blhip_persist1d.hpp), else a launch per K steps
        // (a pass of a single superstep -- OnlineStudy.step, T <= K -- has no launch boundary to save)
        const bool c1d_now = gp.chain1d;
        const bool p1d_now = fused1d && !c1d_now && !resident_failed && ctx->resident_ok && ctx->option("persist1d", 1.0) != 0.0 && T > K &&
                             (long long)tile.nblk * B <= std::min(ctx->num_cus, 256);
        // (the chains of a batch see the same likelihood: tabulated once per batch by the in-kernel function itself, shared by both passes)
        const bool c1d_table = c1d_now && B >= 4 && !d_lik && (p->obs_model == BLHIP_OM_POISSON || p->obs_model == BLHIP_OM_GAUSSIAN_MEAN);
        double *d_lik1 = nullptr;
        if (c1d_table) {
            ctx->lik1d.ensure((size_t)T * G * 8);
            d_lik1 = ctx->lik1d.as<double>();
            bl1f::F1Params Q = F1;
            build_lik1d_table(st, p->obs_model, Q, d_lik1);
        }
        auto launch_c1d = [&](bool bwd, double *psum) {
            bl1f::F1Params Q = F1;
            if (d_lik1) Q.lik = d_lik1;
            Q.K = 1; Q.dir = bwd ? -1 : 1; Q.t_first = bwd ? (int)(T - 1) : 0; Q.psum = psum; Q.prev_slot = bwd ? 2 : 0;
            Q.srckind = bwd ? d_kindB : d_kindF; Q.tap = bwd ? d_tapB1 : d_tapF1;
            if (gp.shift1d) Q.cmode = bwd ? d_cmodeB : d_cmodeF;
            if (gp.clamp1d) { Q.limit = bwd ? d_limitB : d_limitF; Q.no_shift = prog.has_shift ? 0 : 1; }
            Q.store = (bwd || !evidence_only) ? 1 : 0; Q.means = bwd ? 1 : (forward_only ? 1 : 0);
            Q.post = (bwd || !evidence_only) ? d_post : nullptr; Q.post_stride = (long long)T * G;
            Q.src = nullptr; Q.src_stride = 0; Q.dst = nullptr; Q.dst_stride = 0;
            launch_chain1d(st, d_lik1 ? BLHIP_OM_TABLE : p->obs_model, Q, bwd, 2);
        };
        bl1p::P1Params P1{};
        unsigned *d_abort1 = nullptr;
        size_t p1d_bytes = 0;
        if (p1d_now) {
            const size_t xb = carve_size((size_t)2 * B * g.n1 * 16), gb = carve_size((size_t)2 * B * tile.nblk * 16);
            p1d_bytes = xb + gb + carve_size(64);
            ctx->p1d.ensure(p1d_bytes);
            char *pc = ctx->p1d.as<char>();
            P1.xch = carve<unsigned long long>(pc, (size_t)2 * B * g.n1 * 2);
            P1.gran = carve<unsigned long long>(pc, (size_t)2 * B * tile.nblk * 2);
            d_abort1 = carve<unsigned>(pc, 16);
            P1.abort_word = d_abort1;
            P1.timeout_ticks = (unsigned long long)(resident_timeout_s(ctx, T) * 1e8);      // wall_clock64: 100 MHz
            P1.n = F1.n; P1.TJ = F1.TJ; P1.nblk = F1.nblk; P1.LW = F1.LW; P1.K = (int)K; P1.T = F1.T; P1.B = F1.B; P1.d = F1.d; P1.rec_len = F1.rec_len;
            for (int k = 0; k < 5; ++k) P1.shared[k] = F1.shared[k];
            P1.taps = F1.taps; P1.tap_off = F1.tap_off; P1.tap_lw = F1.tap_lw; P1.m1 = F1.m1; P1.colA = F1.colA; P1.rec = F1.rec; P1.lik = F1.lik;
        }
        auto launch_p1d = [&](bool bwd, double *psum) {
            HIPCHECK(hipMemsetAsync(ctx->p1d.p, 0, p1d_bytes, st));       // tags 0, abort word 0
            bl1p::P1Params Q = P1;
            Q.dir = bwd ? -1 : 1; Q.psum = psum; Q.prev_slot = bwd ? 2 : 0;
            Q.srckind = bwd ? d_kindB : d_kindF; Q.tap = bwd ? d_tapB1 : d_tapF1;
            {   // the pass's weights spelled out per (step, chain): the kernel stages a superstep's weights with ONE load per element
                const long long TB = (long long)T * B;
                const size_t wb = carve_size((size_t)TB * (P1.LW + 1) * 8), lb = carve_size((size_t)TB * 4);
                if (wb + lb <= ((size_t)256 << 20)) {
                    ctx->p1w.ensure(wb + lb);
                    char *wc = ctx->p1w.as<char>();
                    double *wtab = carve<double>(wc, (size_t)TB * (P1.LW + 1));
                    int *lwtab = carve<int>(wc, (size_t)TB);
                    const long long ne = TB * (P1.LW + 1);
                    BL_LAUNCH(bl1p::build_wtab_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, Q.tap, d_lw, d_off, d_taps, TB, P1.LW, wtab, lwtab);
                    HIPCHECK(hipGetLastError());
                    Q.wtab = wtab; Q.lwtab = lwtab;
                }
            }
            Q.store = (bwd || !evidence_only) ? 1 : 0; Q.means = bwd ? 1 : (forward_only ? 1 : 0);
            Q.post = (bwd || !evidence_only) ? d_post : nullptr; Q.post_stride = (long long)T * G;
            Q.src0 = (!bwd && resume) ? d_carry_src : nullptr; Q.src0_stride = G;
            Q.dst = (bwd || evidence_only) ? d_pp[((T - 1) / K) & 1] : nullptr; Q.dst_stride = G;
            launch_persist1d(st, p->obs_model, Q, bwd, f1_lds(K));
        };
        if (c1d_now) {
            launch_c1d(false, d_psF);
        } else if (p1d_now) {
            launch_p1d(false, d_psF);
        } else if (fused1d) {
            for (int64_t t = 0; t < T; t += K) {
                bl1f::F1Params Q = F1;
                Q.K = (int)std::min<int64_t>(K, T - t); Q.dir = 1; Q.t_first = (int)t;
                Q.srckind = d_kindF; Q.tap = d_tapF1; Q.psum = d_psF; Q.prev_slot = 0;
                Q.psum_prev = t > 0 ? d_psF + (size_t)(t - 1) * B * NRED * tile.nblk : d_unit;
                Q.store = evidence_only ? 0 : 1; Q.means = forward_only ? 1 : 0;
                const int64_t tl = t + Q.K - 1;              // last step of this launch
                if (evidence_only) {
                    Q.post = nullptr; Q.src = d_pp[(t / K + 1) & 1]; Q.src_stride = G; Q.dst = d_pp[(t / K) & 1]; Q.dst_stride = G;
                } else {
                    Q.post = d_post; Q.post_stride = (long long)T * G; Q.dst = nullptr; Q.dst_stride = 0;
                    Q.src = d_post + (t > 0 ? (size_t)(t - 1) * G : 0); Q.src_stride = (long long)T * G;
                }
                (void)tl;
                if (t == 0 && resume) { Q.src = d_carry_src; Q.src_stride = G; }
                launch_fused1d(st, p->obs_model, Q, false, f1_lds(Q.K));
            }
        }
        if (fused1d)        // 1-D paths: a row in, a row out per step (the K-steps-per-launch kernel re-reads only halos)
            account(ctx, false, (double)B * G * T * (16.0 + (d_lik ? 8.0 : 0.0)), (double)B * G * T * (valu_stencil_flop(prog.LW1) + EPI_FWD_FLOP));
        const bool res_now = resident && !resident_failed;
        const bool cres_now = chainres && !resident_failed;
        const int nblk_now = res_now ? RR.nblk : (cres_now ? CR.cp.strips : tile.nblk);      // partial-sum slots per (step, sum) of this pass
        if (res_now) RR.launch(E, false, d_psF);
        if (cres_now) CR.pass(E, false, d_psF);
        fork_streams();
        for (int64_t t = 0; t < T && !fused1d && !res_now && !cres_now; ++t) {
            if (multistream && t > 0 && !same_membership(orderF, rangesF, t - 1, t)) { join_streams(); fork_streams(); }
            const double *srcp; double *dstp; long long sstr, dstr;
            if (evidence_only) {
                srcp = d_pp[(t + 1) & 1]; sstr = G; dstp = d_pp[t & 1]; dstr = G;
            } else {
                srcp = d_post + (t > 0 ? (size_t)(t - 1) * G : 0); sstr = (long long)T * G;
                dstp = d_post + (size_t)t * G; dstr = (long long)T * G;
            }
            if (t == 0 && resume) { srcp = d_carry_src; sstr = G; }
            run_step(MODE_FWD, t, srcp, sstr, dstp, dstr, nullptr, 0,
                     t > 0 ? d_psF + (size_t)(t - 1) * B * NRED * tile.nblk : (resume ? d_unit : d_psF), 0,
                     d_psF + (size_t)t * B * NRED * tile.nblk, forward_only);
        }
        join_streams();
        HIPCHECK(hipEventRecord(ev[1], st));
        // (one slot per sum -- the kernels that give a chain ONE block: nothing to add up, the partials ARE the sums.  The published
        //  break-point study: 2 x 6 launches of T * B * NRED = 1.3 M one-value blocks, 0.48 ms each, 5.7 of a 44-ms fit)
        if (nblk_now > 1)
            BL_LAUNCH(reduce_partials_kernel, dim3((unsigned)(T * B * NRED)), dim3(NTHREADS), 0, st, d_psF,
                               ctx->redF.as<double>(), nblk_now, NRED);
        ctx->pinF.ensure((size_t)T * B * NRED * 8);
        redF = ctx->pinF.as<double>();
        HIPCHECK(hipMemcpyAsync(redF, nblk_now > 1 ? ctx->redF.p : (const void *)d_psF, (size_t)T * B * NRED * 8, hipMemcpyDeviceToHost, st));
        tr.mark("forward pass queued");
        sync_stream(ctx, st);
        tr.mark("forward pass done + sums D2H");
        ms = 0;
        HIPCHECK(hipEventElapsedTime(&ms, ev[0], ev[1]));
        ctx->timing.forward_ms += ms;
        ctx->timing.forward_launches += T;
        if (n_mfma[0] > 0 && n_mfma[0] >= n_fast[0]) ctx->timing.fwd_kernel_variant = 3;
        if (res_now) {
            ctx->timing.fwd_kernel_variant = 5;
            if (!RR.forward_ok(E, redF)) { resident_failed = true; return false; }
        }
        if (cres_now) {
            ctx->timing.fwd_kernel_variant = 6;
            if (!CR.forward_ok(E, redF)) { resident_failed = true; return false; }
        }
        if (c1d_now) ctx->timing.fwd_kernel_variant = 9;
        if (p1d_now) {
            ctx->timing.fwd_kernel_variant = 8;
            if (resident_gave_up(ctx, st, d_abort1)) { resident_failed = true; return false; }
        }

        // (the launch-per-step backward pass of a batch of chains needs nothing of the forward pass's host bookkeeping: it is done
        //  while the GPU runs that pass -- 11 ms of the published break-point study's fit, 23 batches of 1017 chains)
        const bool late_fb = full && (!fused1d || c1d_now) && !res_now && !cres_now && !p1d_now && B >= 64;
        bool raw_ok = late_fb ? true : forward_bookkeeping(p, prog, redF, B, dV, fused1d, K, evidence_only, forward_only, O);
        tr.mark("forward checks + bookkeeping");

        // --- backward pass (core.py:424-470) ---
        invN.assign((size_t)B * T, 0.0);
        if (full) {
            ctx->psumB.ensure(psz * 8);
            ctx->redB.ensure((size_t)T * B * NRED * 8);
            double *d_psB = ctx->psumB.as<double>();
            if (fused1d && !raw_ok && K > 1) return false;
            HIPCHECK(hipEventRecord(ev[2], st));
            if (c1d_now) {
                launch_c1d(true, d_psB);
            } else if (p1d_now) {
                launch_p1d(true, d_psB);
            } else if (fused1d) {
                for (int64_t t = T - 1; t >= 0; t -= K) {
                    bl1f::F1Params Q = F1;
                    Q.K = (int)std::min<int64_t>(K, t + 1); Q.dir = -1; Q.t_first = (int)t;
                    Q.srckind = d_kindB; Q.tap = d_tapB1; Q.psum = d_psB; Q.prev_slot = 2;
                    Q.psum_prev = t < T - 1 ? d_psB + (size_t)(t + 1) * B * NRED * tile.nblk : nullptr;
                    Q.store = 1; Q.means = 1; Q.post = d_post; Q.post_stride = (long long)T * G;
                    const int64_t li = (T - 1 - t) / K;
                    Q.src = d_pp[(li + 1) & 1]; Q.src_stride = G; Q.dst = d_pp[li & 1]; Q.dst_stride = G;
                    launch_fused1d(st, p->obs_model, Q, true, f1_lds(Q.K));
                }
            }
            if (fused1d)
                account(ctx, true, (double)B * G * T * (32.0 + (d_lik ? 8.0 : 0.0)), (double)B * G * T * (valu_stencil_flop(prog.LW1) + EPI_BWD_FLOP));
            if (res_now) RR.launch(E, true, d_psB);
            if (cres_now && CR.fused) CR.prepare_fold(E, O);
            if (cres_now) CR.pass(E, true, d_psB);
            fork_streams();
            for (int64_t t = T - 1; t >= 0 && !fused1d && !res_now && !cres_now; --t) {
                if (multistream && t < T - 1 && !same_membership(orderB, rangesB, t + 1, t)) { join_streams(); fork_streams(); }
                // reads c_{t+1} and the stored alpha_t, writes c_t and posterior_t
                run_step(MODE_BWD, t, d_pp[(t + 1) & 1], G, d_pp[t & 1], G, d_post + (size_t)t * G, (long long)T * G,
                         t < T - 1 ? d_psB + (size_t)(t + 1) * B * NRED * tile.nblk : d_psB, 2,
                         d_psB + (size_t)t * B * NRED * tile.nblk, true);
            }
            join_streams();
            HIPCHECK(hipEventRecord(ev[3], st));
            if (nblk_now > 1)
                BL_LAUNCH(reduce_partials_kernel, dim3((unsigned)(T * B * NRED)), dim3(NTHREADS), 0, st, d_psB,
                                   ctx->redB.as<double>(), nblk_now, NRED);
            ctx->pinB.ensure((size_t)T * B * NRED * 8);
            redB = ctx->pinB.as<double>();
            HIPCHECK(hipMemcpyAsync(redB, nblk_now > 1 ? ctx->redB.p : (const void *)d_psB, (size_t)T * B * NRED * 8, hipMemcpyDeviceToHost, st));
            tr.mark("backward pass queued");
            if (late_fb) { raw_ok = forward_bookkeeping(p, prog, redF, B, dV, fused1d, K, evidence_only, forward_only, O); tr.mark("forward bookkeeping (behind the backward pass)"); }
            sync_stream(ctx, st);
            tr.mark("backward pass done + sums D2H");
            HIPCHECK(hipEventElapsedTime(&ms, ev[2], ev[3]));
            ctx->timing.backward_ms += ms;
            if (n_mfma[1] > 0 && n_mfma[1] >= n_fast[1]) ctx->timing.bwd_kernel_variant = 3;
            ctx->timing.backward_launches += T;
            if (res_now) {
                ctx->timing.bwd_kernel_variant = 5;
                if (!RR.backward_ok(E, redB)) { resident_failed = true; return false; }
            }
            if (c1d_now) ctx->timing.bwd_kernel_variant = 9;
            if (p1d_now) {
                ctx->timing.bwd_kernel_variant = 8;
                if (resident_gave_up(ctx, st, d_abort1)) { resident_failed = true; return false; }
            }
            if (cres_now) {
                ctx->timing.bwd_kernel_variant = 6;
                if (!CR.backward_ok(E, redB)) { resident_failed = true; return false; }
                if (CR.fused && !CR.fold(E, redB)) { resident_failed = true; return false; }
            }
            raw_ok = backward_bookkeeping(p, prog, redF, redB, B, dV, fused1d, res_now ? 0 : -1, O) && raw_ok;
            tr.mark("backward checks + fused fold + bookkeeping");
        } else if (forward_only) {
            for (int64_t b = 0; b < B; ++b)
                for (int64_t t = 0; t < T; ++t) {
                    const double n0 = res_now ? RR.rowsumF[t] : (cres_now ? CR.rowsumC[b][t] : redF[((size_t)t * B + b) * NRED]);
                    invN[(size_t)b * T + t] = (n0 != 0.0 && std::isfinite(n0)) ? 1.0 / n0 : 0.0;     // (a signed kernel can leave a negative raw sum)
                    if (res_now && t <= T - 1 - RR.RQ.lag) invN[(size_t)b * T + t] = 1.0;                // (already normalised by the resident kernel)
                }
        }

        return raw_ok || K == 1;
        };
        int64_t usedK = fusedK;
        if (!passes(fusedK)) {
            if (resident_failed) {                   // the launch-per-step kernels take over (timing of the failed attempt is dropped)
                ctx->timing.resident_fallbacks += 1;
                ctx->timing.resident_fallback_reason = ctx->resident_last_reason ? ctx->resident_last_reason : BLHIP_FALLBACK_RANGE;
                ctx->timing.forward_ms = ctx->timing.backward_ms = 0.0;
                ctx->timing.fwd_hbm_bytes = ctx->timing.bwd_hbm_bytes = ctx->timing.fwd_flops = ctx->timing.bwd_flops = 0.0;
                ctx->timing.forward_launches = ctx->timing.backward_launches = 0;
                ctx->timing.fwd_kernel_variant = ctx->timing.bwd_kernel_variant = fast ? 1 : (fused1d ? 4 : 0);
                n_mfma[0] = n_mfma[1] = n_fast[0] = n_fast[1] = 0;
                if (!passes(fusedK)) fail("internal: the launch-per-step pass failed after the resident pass gave up");
            } else { usedK = 1; passes(1); }
        } else if (resident || chainres || ctx->timing.fwd_kernel_variant == 8) {
            ctx->resident_retry_after = 8;           // a resident pass went through: the next give-up starts from the short wait again
            ctx->resident_giveups = 0;               // (the wait doubles with give-ups IN A ROW only)
        }

        if (CR.on && CR.depad && !resident_failed && !evidence_only) {
            BL_LAUNCH(depad_kernel, dim3((unsigned)((G + NTHREADS - 1) / NTHREADS), (unsigned)(B * T)), dim3(NTHREADS), 0, st, d_post,
                               ctx->postpad.as<double>(), g.n0, g.n1, CR.cp.n0p, CR.Gk, (int)T, CR.ax1 ? 1 : 0);
            HIPCHECK(hipGetLastError());
        }

        // --- carried states / average posterior / kept posterior / results ---
        if (carry) {
            const double *fin; long long fstr;
            if (!evidence_only) { fin = d_post + (size_t)(T - 1) * G; fstr = (long long)T * G; }
            else { fin = fused1d ? d_pp[((T - 1) / usedK) & 1] : d_pp[(T - 1) & 1]; fstr = G; }
            store_carry(ctx, p, B, G, redF, fin, fstr, d_w, prog.has_clamp);
        }
        if (accumulate && CR.fold_done) {
            // (the backward kernel folded this batch)
        } else if (accumulate && overlap_acc) {
            // private copies of the weights, host (page-locked: the copies must not block the host) and device: the batch metadata
            // buffers are rewritten by the next batch while the fold is pending / running
            const size_t wsz = carve_size((size_t)Bmax * 8) + carve_size((size_t)T * Bmax * 8);
            ctx->accw.ensure(2 * wsz);
            ctx->pinA.ensure(2 * wsz);
            char *wc = ctx->accw.as<char>() + (bi & 1) * wsz, *hc = ctx->pinA.as<char>() + (bi & 1) * wsz;
            fold_job.d_w = carve<double>(wc, (size_t)Bmax); fold_job.d_invN = carve<double>(wc, (size_t)T * Bmax);
            fold_job.h_w = carve<double>(hc, (size_t)Bmax); fold_job.h_invN = carve<double>(hc, (size_t)T * Bmax);
            fold_job.d_post = d_post; fold_job.parity = (int)(bi & 1);
            fold_job.sm_n0 = (CR.post_private && !resident_failed && !CR.cp.pad && !CR.ax1) ? g.n0 : 0;
            if (CR.on && (CR.cp.pad || CR.ax1) && !CR.depad && !resident_failed) { fold_job.pad_n0p = CR.cp.n0p; fold_job.pad_n0 = g.n0; fold_job.pad_n1 = g.n1; fold_job.pad_step = CR.Gk; fold_job.pad_ax = CR.ax1 ? 1 : 0; }
            fold_job.pending = prepare_fold(ctx, T, B, O, log_w + c0, fold_job);
            // launched behind the NEXT batch's forward pass (see passes); the last batch has nothing to hide behind
            if (bi == nbatch - 1) launch_pending_fold();
        } else if (accumulate) {
            FoldJob lay;
            const bool padded = CR.on && (CR.cp.pad || CR.ax1) && !CR.depad && !resident_failed;
            if (padded) { lay.pad_n0p = CR.cp.n0p; lay.pad_n0 = g.n0; lay.pad_n1 = g.n1; lay.pad_step = CR.Gk; lay.pad_ax = CR.ax1 ? 1 : 0; }
            // (many small batches: nobody waits for a batch's fold; the last one is waited for with the stream below)
            fold_accumulate(ctx, T, G, B, O, log_w + c0, d_post, d_w, d_invN, (CR.post_private && !resident_failed && !padded) ? g.n0 : 0, padded ? &lay : nullptr,
                            (nbatch >= 4 && !keep && !carry) ? &fold_ev : nullptr);
        }
        if (keep) {
            int64_t row0 = 0, row1 = T;              // rows the resident kernel normalised in place (invN = 1 there) need no pass
            if (resident && !resident_failed) {
                if (full) row1 = 0;                  // (every posterior row was stored normalised)
                else row0 = std::max<int64_t>(0, T - RR.RQ.lag);
            }
            keep_posterior(ctx, g, T, B, O, row0, row1);
        }
        tr.mark("fold / keep / carry");
        write_results(res, p, c0, B, O, !evidence_only);
    }
    if (overlap_acc || !fold_ev.empty()) {     // the last fold(s) before anybody reads the accumulator; their time from their events
        sync_stream(ctx, overlap_acc ? ctx->astream : st);
        for (size_t k = 0; k + 1 < fold_ev.size(); k += 2) {
            float fms = 0;
            HIPCHECK(hipEventElapsedTime(&fms, fold_ev[k], fold_ev[k + 1]));
            ctx->timing.accumulate_ms += fms;
        }
        for (hipEvent_t e : fold_ev) (void)hipEventDestroy(e);
    }
    tr.mark("batches done");
    HIPCHECK(hipEventRecord(ev[7], st));
    HIPCHECK(hipEventSynchronize(ev[7]));
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, ev[6], ev[7]));
    tr.mark("final sync");
    ctx->timing.total_ms = ms;
}

}  // namespace

// Page-locked result arrays.  hipHostMalloc of 16 GiB takes 2.2 s (one thread allocates, zeroes and pins four million pages) -- more
// than the copy it is meant to speed up.  Here the block is an anonymous mapping pinned in PIN_CHUNK pieces (hipHostRegister faults
// the pages in and locks them: 1.06 s for 16 GiB; more threads are slower); the read-backs below copy piece by piece, so every
// DMA lands inside one registered range.  The Python side pins in the background (bayesloop_amd/engine.py: _PinnedPool).
namespace {
constexpr size_t PIN_CHUNK = (size_t)256 << 20;
struct PinnedBlock { size_t bytes; std::vector<char> ok; };
std::mutex g_pin_mu;
std::map<void *, PinnedBlock> g_pinned;

// pieces [a, b) of a host destination such that no piece crosses a chunk boundary of a block of blhip_host_alloc (any other
// destination: one piece)
template <class F> void for_pinned_pieces(void *host, size_t bytes, F &&f) {
    char *base = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        auto it = g_pinned.upper_bound(host);
        if (it != g_pinned.begin()) {
            --it;
            if ((char *)host >= (char *)it->first && (char *)host + bytes <= (char *)it->first + it->second.bytes) base = (char *)it->first;
        }
    }
    if (!base) { f((char *)host, bytes); return; }
    char *p = (char *)host, *end = p + bytes;
    while (p < end) {
        const size_t into = (size_t)(p - base) % PIN_CHUNK;
        const size_t n = std::min<size_t>(PIN_CHUNK - into, (size_t)(end - p));
        f(p, n);
        p += n;
    }
}
}  // namespace

extern "C" {

int blhip_abi_version(void) { return BLHIP_ABI_VERSION; }

int64_t blhip_kernel_census(char *buf, int64_t cap) {
    try {
        std::vector<std::pair<std::string, unsigned long long>> rows;
        for (blreg::Entry *e = blreg::head().load(std::memory_order_acquire); e; e = e->next) {
            // typeid name of blreg::Site<&kernel>: demangled "blreg::Site<&blc::chain_kernel<4, 1, false, false, false, false>(blc::ChainParams)>"
            int status = 0;
            char *dm = abi::__cxa_demangle(e->mangled, nullptr, nullptr, &status);
            std::string name = (status == 0 && dm) ? dm : (e->mangled ? e->mangled : "?");
            std::free(dm);
            // -> "blreg::Site<&(void blc::chain_kernel<10, 1, false, false, false, false>(blc::ChainParams))>": keep the kernel with its template arguments
            const std::string pre = "blreg::Site<&(", post = ")>";
            if (name.size() > pre.size() + post.size() && name.compare(0, pre.size(), pre) == 0 && name.compare(name.size() - post.size(), post.size(), post) == 0) {
                name = name.substr(pre.size(), name.size() - pre.size() - post.size());
                if (name.compare(0, 5, "void ") == 0) name.erase(0, 5);
                if (!name.empty() && name.back() == ')') {            // the parameter list: the last top-level parenthesis group
                    int depth = 0;
                    for (size_t k = name.size(); k-- > 0;) {
                        if (name[k] == ')') ++depth;
                        else if (name[k] == '(' && --depth == 0) { name.erase(k); break; }
                    }
                }
            }
            else if (name.compare(0, 13, "blreg::Site<&") == 0 && name.back() == '>') name = name.substr(13, name.size() - 14);      // (a kernel that is no template)
            rows.emplace_back(name, e->launches.load(std::memory_order_relaxed));
        }
        std::sort(rows.begin(), rows.end());
        std::string out;
        for (const auto &r : rows) out += std::to_string(r.second) + "\t" + r.first + "\n";
        if (buf && cap > 0) {
            const size_t n = std::min<size_t>((size_t)cap - 1, out.size());
            std::memcpy(buf, out.data(), n);
            buf[n] = 0;
        }
        return (int64_t)out.size();
    } catch (...) {
        return -1;
    }
}

int blhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

blhip_ctx *blhip_create(int device) {
    blhip_ctx *ctx = nullptr;
    try {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) fail("no HIP device visible");
        if (device < 0 || device >= n) fail("device %d out of range (%d visible)", device, n);
        HIPCHECK(hipSetDevice(device));
        ctx = new blhip_ctx();
        ctx->device = device;
        HIPCHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        for (auto &e : ctx->ev) HIPCHECK(hipEventCreate(&e));
        for (auto &bs : ctx->bstream) HIPCHECK(hipStreamCreateWithFlags(&bs, hipStreamNonBlocking));
        for (auto &e : ctx->bev) HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIPCHECK(hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming));
        HIPCHECK(hipEventCreateWithFlags(&ctx->sync_ev, hipEventDisableTiming));
        hipDeviceProp_t prop;
        HIPCHECK(hipGetDeviceProperties(&prop, device));
        ctx->num_cus = prop.multiProcessorCount;
        std::string marketing = prop.name;
        if (marketing.find_first_not_of(' ') == std::string::npos) marketing = "AMD Instinct (name not reported by the driver)";
        ctx->name = marketing + " (" + prop.gcnArchName + ")";
        return ctx;
    } catch (const Fail &e) {
        g_create_error = e.msg;
    } catch (...) {
        g_create_error = "unknown error in blhip_create";
    }
    delete ctx;
    return nullptr;
}

void blhip_destroy(blhip_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)blhip_comm_destroy(ctx);
    ctx->commbuf.release(); ctx->pinC.release(); ctx->resx.release(); ctx->post2.release(); ctx->postpad.release(); ctx->accw.release(); ctx->accpart.release(); ctx->xch.release(); ctx->axlik.release(); ctx->probebuf.release(); ctx->p1d.release(); ctx->p1w.release(); ctx->lik1d.release();
    if (ctx->astream) { (void)hipStreamSynchronize(ctx->astream); (void)hipStreamDestroy(ctx->astream); }
    for (auto &e : ctx->aev_done) if (e) (void)hipEventDestroy(e);
    for (DevBuf *b : {&ctx->state, &ctx->post, &ctx->psumF, &ctx->psumB, &ctx->redF, &ctx->redB, &ctx->meta,
                      &ctx->tables, &ctx->likbuf, &ctx->small, &ctx->accum_own, &ctx->stats, &ctx->mix, &ctx->unit, &ctx->databuf, &ctx->postinv})
        b->release();
    for (auto &kv : ctx->carry) kv.second.buf.release();
    ctx->pinF.release(); ctx->pinB.release(); ctx->pinS.release(); ctx->pinA.release();
    for (auto &e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &bs : ctx->bstream)
        if (bs) { (void)hipStreamSynchronize(bs); (void)hipStreamDestroy(bs); }
    for (auto &e : ctx->bev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->fork_ev) (void)hipEventDestroy(ctx->fork_ev);
    if (ctx->sync_ev) (void)hipEventDestroy(ctx->sync_ev);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *blhip_last_error(blhip_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int blhip_device_name(blhip_ctx *ctx, char *buf, int buflen) {
    if (!ctx || !buf || buflen <= 0) return -1;
    std::snprintf(buf, (size_t)buflen, "%s", ctx->name.c_str());
    return 0;
}

int blhip_host_unlag(int scheme, double *sums, int64_t T, int lag, const unsigned char *kinds, double *scales_out) {
    if (!sums || T < 1 || lag < 1) return -1;
    try {
        std::vector<double> red((size_t)T * NRED, 0.0), rowsum, scales;
        for (int64_t t = 0; t < T; ++t) red[(size_t)t * NRED] = sums[t];
        const bool ok = scheme == 0 ? resident_unlag(red.data(), T, lag, rowsum) : chain_unlag(red.data(), T, lag, rowsum, 1, 0, &scales, kinds);
        if (!ok) return 1;
        for (int64_t t = 0; t < T; ++t) sums[t] = red[(size_t)t * NRED];
        if (scales_out)
            for (int64_t t = 0; t < T; ++t) scales_out[t] = scheme == 0 ? (t >= lag ? 1.0 / rowsum[t - lag] : 1.0) : scales[t];
        return 0;
    } catch (...) {
        return -1;
    }
}

int blhip_set_option(blhip_ctx *ctx, const char *key, double value) {
    if (!ctx || !key) return -1;
    // "resident_ok": the context's memory of a resident launch that gave up (its blocks were not all co-resident) -- settable so that
    // a caller (the tests of the fall-back) can re-arm the resident paths
    if (std::strcmp(key, "resident_ok") == 0) { ctx->resident_ok = value != 0.0; ctx->resident_fits_since = 0; if (value != 0.0) { ctx->resident_retry_after = 8; ctx->resident_giveups = 0; } return 0; }
    if (std::strcmp(key, "resident_retry_after") == 0) { ctx->resident_retry_after = std::max(1, (int)value); return 0; }
    // (a key the library never reads is an error, not a silent no-op: an A/B run over a removed option measured nothing -- ADVICE r05)
    static const char *const known[] = {
        "accum_overlap", "chain1d", "chain1d_clamp", "chain1d_shift", "chain_ax1", "chain_depad", "chain_prof", "chain_resident", "chain_resident_lag",
        "chain_table", "chain_wide", "comm_reduce_mode", "fast", "fast_S", "fold2", "fold2_cp", "fuse1d", "fuse_accumulate", "max_batch",
        "mem_budget_bytes", "mfma", "mfma_S", "mfma_h", "mfma_h_max_cells", "peer_copy_mode", "persist1d", "quiet", "recurrence", "resident",
        "resident_force_abort", "resident_lag", "resident_probe", "resident_probe_force_busy", "resident_probe_interval_s", "resident_probe_timeout_s",
        "resident_table", "resident_timeout_s", "share_prefix", "skip_prefix", "trace", "wide_h",
        "wide_h_fused_max", "wide_h_split", "wide_v"};
    bool ok = false;
    for (const char *k : known) ok = ok || std::strcmp(key, k) == 0;
    if (!ok) {
        ctx->err = std::string("blhip_set_option: unknown option '") + key + "' (README.md lists the options the library reads)";
        return -1;
    }
    ctx->opt[key] = value;
    return 0;
}

int blhip_synchronize(blhip_ctx *ctx) {
    return guarded(ctx, [&] {
        HIPCHECK(hipSetDevice(ctx->device));
        sync_stream(ctx, ctx->stream);
    });
}

int blhip_fit(blhip_ctx *ctx, const blhip_problem *problem, int64_t n_chains, const double *op_values,
              const double *log_chain_weight, uint32_t flags, blhip_result *result) {
    return guarded(ctx, [&] { do_fit(ctx, problem, n_chains, op_values, log_chain_weight, flags, result); });
}

int blhip_bandwidth_probe(blhip_ctx *ctx, int64_t bytes, int iterations, double *gb_per_s) {
    return guarded(ctx, [&] {
        if (!gb_per_s || bytes < (1 << 20) || iterations < 1) fail("blhip_bandwidth_probe: bad arguments");
        HIPCHECK(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        const long long n2 = bytes / 16;
        ScopedDevBuf a, b;          // (released on every exit path, also when a HIPCHECK below throws)
        a.ensure((size_t)n2 * 16); b.ensure((size_t)n2 * 16);
        BL_LAUNCH(fill_kernel, dim3(2048), dim3(256), 0, st, a.as<double>(), n2 * 2, 1.0);
        const unsigned gx = (unsigned)((n2 + 4 * NTHREADS - 1) / (4 * NTHREADS));
        double best = 0.0;
        for (int variant = 0; variant < 2; ++variant) {
            auto go = [&](const double2 *src, double2 *dst) {
                if (variant) BL_LAUNCH(copy16_kernel<true>, dim3(gx), dim3(NTHREADS), 0, st, src, dst, n2);
                else BL_LAUNCH(copy16_kernel<false>, dim3(gx), dim3(NTHREADS), 0, st, src, dst, n2);
            };
            go(a.as<double2>(), b.as<double2>());                                                   // warm-up
            HIPCHECK(hipEventRecord(ctx->ev[4], st));
            for (int k = 0; k < iterations; ++k) go((k & 1) ? b.as<double2>() : a.as<double2>(), (k & 1) ? a.as<double2>() : b.as<double2>());
            HIPCHECK(hipEventRecord(ctx->ev[5], st));
            HIPCHECK(hipGetLastError());
            sync_stream(ctx, st);
            float ms = 0;
            HIPCHECK(hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5]));
            best = std::max(best, 2.0 * (double)n2 * 16.0 * iterations / ((double)ms * 1e-3) / 1e9);
        }
        for (int variant = 0; variant < 2; ++variant) {
            // a store-only stream (what the no-stencil forward chain kernel does) can run above the copy rate: the calibrated peak is
            // the best of the streams measured
            auto go = [&](double2 *dst, double v) {
                if (variant) BL_LAUNCH(fill16_kernel<true>, dim3(gx), dim3(NTHREADS), 0, st, dst, n2, v);
                else BL_LAUNCH(fill16_kernel<false>, dim3(gx), dim3(NTHREADS), 0, st, dst, n2, v);
            };
            go(b.as<double2>(), 2.0);
            HIPCHECK(hipEventRecord(ctx->ev[4], st));
            for (int k = 0; k < iterations; ++k) go((k & 1) ? a.as<double2>() : b.as<double2>(), 3.0);
            HIPCHECK(hipEventRecord(ctx->ev[5], st));
            HIPCHECK(hipGetLastError());
            sync_stream(ctx, st);
            float ms = 0;
            HIPCHECK(hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5]));
            best = std::max(best, (double)n2 * 16.0 * iterations / ((double)ms * 1e-3) / 1e9);
        }
        *gb_per_s = best;
    });
}

int blhip_last_timing(blhip_ctx *ctx, blhip_timing *out) {
    if (!ctx || !out) return -1;
    *out = ctx->timing;
    out->resident_armed = ctx->resident_ok ? 1 : 0;
    return 0;
}

void *blhip_host_alloc(size_t bytes) {
    if (bytes == 0) return nullptr;
    const size_t len = (bytes + 4095) / 4096 * 4096;
    void *p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) return nullptr;
    const size_t nchunk = (len + PIN_CHUNK - 1) / PIN_CHUNK;
    PinnedBlock blk{len, std::vector<char>(nchunk, 0)};
    int nthreads = 1;      // (measured, 16 GiB: 1 thread 1.06 s, 4: 1.8 s, 16: 2.1 s -- the page-table lock serialises them; hipHostMalloc: 2.2 s)
    if (const char *e = std::getenv("BLHIP_PIN_THREADS")) nthreads = std::max(1, std::atoi(e));
    nthreads = (int)std::min<size_t>((size_t)nthreads, nchunk);
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::atomic<size_t> next{0};
    auto work = [&]() {
        (void)hipSetDevice(dev);
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= nchunk) break;
            char *c = (char *)p + k * PIN_CHUNK;
            const size_t n = std::min(PIN_CHUNK, len - k * PIN_CHUNK);
            if (hipHostRegister(c, n, hipHostRegisterDefault) == hipSuccess) blk.ok[k] = 1;
            else (void)hipGetLastError();
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    bool all = true;
    for (char o : blk.ok) all = all && o;
    if (!all) {
        for (size_t k = 0; k < nchunk; ++k) if (blk.ok[k]) (void)hipHostUnregister((char *)p + k * PIN_CHUNK);
        munmap(p, len);
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_pin_mu);
    g_pinned[p] = std::move(blk);
    return p;
}

void blhip_host_free(void *p) {
    if (!p) return;
    PinnedBlock blk;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        auto it = g_pinned.find(p);
        if (it == g_pinned.end()) return;
        blk = std::move(it->second);
        g_pinned.erase(it);
    }
    for (size_t k = 0; k < blk.ok.size(); ++k) (void)hipHostUnregister((char *)p + k * PIN_CHUNK);
    munmap(p, blk.bytes);
}

int blhip_posterior_read(blhip_ctx *ctx, int64_t chain, int64_t t0, int64_t t1, double *host_out) {
    return guarded(ctx, [&] {
        if (!ctx->post_valid) fail("no posterior kept (run blhip_fit with BLHIP_KEEP_POSTERIOR)");
        ensure_post_scaled(ctx);
        if (chain < 0 || chain >= ctx->post_chains || t0 < 0 || t1 > ctx->post_T || t0 > t1 || !host_out)
            fail("blhip_posterior_read: bad range");
        HIPCHECK(hipSetDevice(ctx->device));
        const char *src = reinterpret_cast<const char *>(ctx->post.as<double>() + ((size_t)chain * ctx->post_T + t0) * ctx->post_G);
        for_pinned_pieces(host_out, (size_t)(t1 - t0) * ctx->post_G * 8, [&](char *dst, size_t n) {
            HIPCHECK(hipMemcpyAsync(dst, src + (dst - (char *)host_out), n, hipMemcpyDeviceToHost, ctx->stream));
        });
        sync_stream(ctx, ctx->stream);
    });
}

int blhip_posterior_devptr(blhip_ctx *ctx, void **devptr, int64_t *chain_stride, int64_t *step_stride) {
    return guarded(ctx, [&] {
        if (!ctx->post_valid) fail("no posterior kept");
        ensure_post_scaled(ctx);
        sync_stream(ctx, ctx->stream);
        if (devptr) *devptr = ctx->post.p;
        if (chain_stride) *chain_stride = ctx->post_T * ctx->post_G;
        if (step_stride) *step_stride = ctx->post_G;
    });
}

int blhip_posterior_release(blhip_ctx *ctx) {
    return guarded(ctx, [&] {
        HIPCHECK(hipSetDevice(ctx->device));
        ctx->post_valid = false;
        ctx->post.release();
        ctx->post2.release();
    });
}

namespace {
struct SeqView { const double *p; int64_t T; int n0, n1; };
SeqView sequence_view(blhip_ctx *ctx, int source, int64_t chain) {
    if (source == 0) {
        if (!ctx->post_valid) fail("no posterior kept (run blhip_fit with BLHIP_KEEP_POSTERIOR)");
        if (chain < 0 || chain >= ctx->post_chains) fail("chain out of range");
        ensure_post_scaled(ctx);
        return SeqView{ctx->post.as<double>() + (size_t)chain * ctx->post_T * ctx->post_G, ctx->post_T, ctx->post_n0, ctx->post_n1};
    }
    if (source == 1) {
        if (!ctx->acc_active || !ctx->acc_final) fail("accumulator not finalised");
        return SeqView{ctx->acc, ctx->acc_T, ctx->acc_n0, ctx->acc_n1};
    }
    fail("bad source %d", source);
}
}  // namespace

int blhip_posterior_marginal(blhip_ctx *ctx, int source, int64_t chain, int keep_axis, double *host_out) {
    return guarded(ctx, [&] {
        if (!host_out) fail("host_out is NULL");
        HIPCHECK(hipSetDevice(ctx->device));
        const SeqView v = sequence_view(ctx, source, chain);
        const bool one_d = v.n0 == 1;
        if (keep_axis < 0 || keep_axis > (one_d ? 0 : 1)) fail("keep_axis out of range");
        const int nk = one_d ? v.n1 : (keep_axis == 0 ? v.n0 : v.n1);
        ctx->stats.ensure((size_t)v.T * nk * 8);
        double *d_out = ctx->stats.as<double>();
        hipStream_t st = ctx->stream;
        if (one_d) {
            HIPCHECK(hipMemcpyAsync(d_out, v.p, (size_t)v.T * nk * 8, hipMemcpyDeviceToDevice, st));
        } else if (keep_axis == 0) {
            BL_LAUNCH(marginal_rows_kernel, dim3(v.n0, (unsigned)v.T), dim3(NTHREADS), 0, st, v.p, d_out, v.n0, v.n1);
        } else {
            BL_LAUNCH(marginal_cols_kernel, dim3((v.n1 + NTHREADS - 1) / NTHREADS, (unsigned)v.T), dim3(NTHREADS), 0, st,
                               v.p, d_out, v.n0, v.n1);
        }
        HIPCHECK(hipMemcpyAsync(host_out, d_out, (size_t)v.T * nk * 8, hipMemcpyDeviceToHost, st));
        sync_stream(ctx, st);
    });
}

int blhip_posterior_time_average(blhip_ctx *ctx, int source, int64_t chain, double *host_out) {
    return guarded(ctx, [&] {
        if (!host_out) fail("host_out is NULL");
        HIPCHECK(hipSetDevice(ctx->device));
        const SeqView v = sequence_view(ctx, source, chain);
        const long long G = (long long)v.n0 * v.n1;
        ctx->stats.ensure((size_t)G * 8);
        double *d_out = ctx->stats.as<double>();
        hipStream_t st = ctx->stream;
        BL_LAUNCH(time_average_kernel, dim3((unsigned)std::min<long long>((G + NTHREADS - 1) / NTHREADS, 4096)),
                           dim3(NTHREADS), 0, st, v.p, d_out, G, (int)v.T);
        HIPCHECK(hipMemcpyAsync(host_out, d_out, (size_t)G * 8, hipMemcpyDeviceToHost, st));
        sync_stream(ctx, st);
    });
}

int blhip_carry_mix(blhip_ctx *ctx, int slot, int64_t n_chains, const double *weights, int accumulate) {
    return guarded(ctx, [&] {
        if (!weights) fail("weights is NULL");
        HIPCHECK(hipSetDevice(ctx->device));
        auto it = ctx->carry.find(slot);
        if (it == ctx->carry.end() || !it->second.valid) fail("blhip_carry_mix: carry slot %d holds no state", slot);
        blhip_ctx::Carry &cs = it->second;
        if (cs.chains != n_chains) fail("blhip_carry_mix: slot %d holds %lld chains, got %lld weights", slot, (long long)cs.chains, (long long)n_chains);
        if (accumulate && ctx->mix_G != cs.G) fail("blhip_carry_mix: accumulating %lld cells into a mix of %lld", (long long)cs.G, (long long)ctx->mix_G);
        hipStream_t st = ctx->stream;
        ctx->mix.ensure((size_t)cs.G * 8);
        ctx->mix_G = cs.G;
        ctx->small.ensure((size_t)n_chains * 8);
        HIPCHECK(hipMemcpyAsync(ctx->small.p, weights, (size_t)n_chains * 8, hipMemcpyHostToDevice, st));
        const unsigned gx = (unsigned)std::min<long long>((cs.G + NTHREADS - 1) / NTHREADS, 8192);
        BL_LAUNCH(carry_mix_kernel, dim3(gx), dim3(NTHREADS), 0, st, ctx->mix.as<double>(), cs.buf.as<double>(),
                           (long long)cs.G, (int)n_chains, ctx->small.as<double>(), accumulate ? 1 : 0);
        HIPCHECK(hipGetLastError());
        sync_stream(ctx, st);
    });
}

int blhip_carry_read(blhip_ctx *ctx, int slot, int64_t chain, double *host_out) {
    return guarded(ctx, [&] {
        if (!host_out) fail("host_out is NULL");
        HIPCHECK(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        if (chain < 0) {
            if (ctx->mix_G <= 0) fail("blhip_carry_read: no mix computed");
            HIPCHECK(hipMemcpyAsync(host_out, ctx->mix.p, (size_t)ctx->mix_G * 8, hipMemcpyDeviceToHost, st));
        } else {
            auto it = ctx->carry.find(slot);
            if (it == ctx->carry.end() || !it->second.valid) fail("blhip_carry_read: carry slot %d holds no state", slot);
            if (chain >= it->second.chains) fail("blhip_carry_read: chain %lld of %lld", (long long)chain, (long long)it->second.chains);
            HIPCHECK(hipMemcpyAsync(host_out, it->second.buf.as<double>() + (size_t)chain * it->second.G, (size_t)it->second.G * 8,
                                    hipMemcpyDeviceToHost, st));
        }
        sync_stream(ctx, st);
    });
}

int blhip_carry_write(blhip_ctx *ctx, int slot, int64_t n_chains, int64_t G, const double *host_in) {
    return guarded(ctx, [&] {
        if (!host_in || n_chains < 1 || G < 1 || slot < 0) fail("blhip_carry_write: bad arguments");
        HIPCHECK(hipSetDevice(ctx->device));
        blhip_ctx::Carry &cs = ctx->carry[slot];
        cs.buf.ensure((size_t)n_chains * G * 8);
        HIPCHECK(hipMemcpyAsync(cs.buf.p, host_in, (size_t)n_chains * G * 8, hipMemcpyHostToDevice, ctx->stream));
        sync_stream(ctx, ctx->stream);
        cs.chains = n_chains; cs.G = G; cs.valid = true;
        cs.maxv.assign(n_chains, 0.0);           // (NotEqual inverts around the maximum of the carried state)
        for (int64_t b = 0; b < n_chains; ++b) {
            double m = -INFINITY;
            for (int64_t c = 0; c < G; ++c) m = std::max(m, host_in[b * G + c]);
            cs.maxv[b] = m;
        }
    });
}

int blhip_carry_release(blhip_ctx *ctx, int slot) {
    return guarded(ctx, [&] {
        HIPCHECK(hipSetDevice(ctx->device));
        if (slot < 0) {
            for (auto &kv : ctx->carry) kv.second.buf.release();
            ctx->carry.clear(); ctx->mix.release(); ctx->mix_G = 0;
        } else {
            auto it = ctx->carry.find(slot);
            if (it != ctx->carry.end()) { it->second.buf.release(); ctx->carry.erase(it); }
        }
    });
}

int blhip_accum_begin(blhip_ctx *ctx, int64_t T, int64_t G, void *external_devptr) {
    return guarded(ctx, [&] {
        if (T < 1 || G < 1) fail("blhip_accum_begin: bad shape");
        HIPCHECK(hipSetDevice(ctx->device));
        if (external_devptr) {
            ctx->acc = reinterpret_cast<double *>(external_devptr);
        } else {
            ctx->accum_own.ensure((size_t)T * G * 8);
            ctx->acc = ctx->accum_own.as<double>();
        }
        ctx->acc_T = T; ctx->acc_G = G; ctx->acc_folded = 0;
        ctx->acc_logref = -std::numeric_limits<double>::infinity();
        ctx->acc_active = true; ctx->acc_final = false; ctx->acc_first = true;
    });
}

int blhip_accum_state(blhip_ctx *ctx, double *log_ref, void **devptr, int64_t *n_folded) {
    return guarded(ctx, [&] {
        if (!ctx->acc_active) fail("no active accumulator");
        if (log_ref) *log_ref = ctx->acc_logref;
        if (devptr) *devptr = ctx->acc;
        if (n_folded) *n_folded = ctx->acc_folded;
    });
}

int blhip_accum_rescale(blhip_ctx *ctx, double new_log_ref) {
    return guarded(ctx, [&] {
        if (!ctx->acc_active) fail("no active accumulator");
        HIPCHECK(hipSetDevice(ctx->device));
        const long long n = (long long)ctx->acc_T * ctx->acc_G;
        if (ctx->acc_first) {
            // nothing folded on this rank: contributes zeros
            BL_LAUNCH(fill_kernel, dim3(1024), dim3(256), 0, ctx->stream, ctx->acc, n, 0.0);
            ctx->acc_first = false;
        } else {
            if (new_log_ref < ctx->acc_logref) fail("blhip_accum_rescale: new reference below the current one");
            const double r = std::exp(ctx->acc_logref - new_log_ref);
            if (r != 1.0)
                BL_LAUNCH(scale_all_kernel, dim3(2048), dim3(NTHREADS), 0, ctx->stream, ctx->acc, n, r);
        }
        ctx->acc_logref = new_log_ref;
        sync_stream(ctx, ctx->stream);
    });
}

int blhip_accum_fold_host(blhip_ctx *ctx, const double *posterior, double log_weight) {
    return guarded(ctx, [&] {
        if (!ctx->acc_active) fail("blhip_accum_fold_host without blhip_accum_begin");
        if (ctx->acc_final) fail("blhip_accum_fold_host: the accumulator is finalised");
        if (!posterior) fail("posterior is NULL");
        if (!std::isfinite(log_weight)) return;                 // np.isfinite(logEvidence) guard, a zero hyper-prior (core.py:1358, :1364)
        HIPCHECK(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        const int64_t T = ctx->acc_T;
        const long long G = ctx->acc_G;
        ctx->post_valid = false;                                  // (the sequence buffer is the staging area)
        ctx->post.ensure((size_t)T * G * 8);
        for_pinned_pieces(const_cast<double *>(posterior), (size_t)T * G * 8, [&](char *src, size_t n) {
            HIPCHECK(hipMemcpyAsync(ctx->post.as<char>() + (src - (const char *)posterior), src, n, hipMemcpyHostToDevice, st));
        });
        BatchOutcome O;
        O.logE.assign(1, log_weight);
        O.abort_step.assign(1, -1);
        O.abort_phase.assign(1, 0);
        O.invN.assign((size_t)T, 1.0);                             // (rows arrive normalised)
        const double zero = 0.0;
        ctx->small.ensure(carve_size(8) + carve_size((size_t)T * 8));
        char *cur = ctx->small.as<char>();
        double *d_w = carve<double>(cur, 1), *d_invN = carve<double>(cur, (size_t)T);
        fold_accumulate(ctx, T, G, 1, O, &zero, ctx->post.as<double>(), d_w, d_invN);
    });
}

namespace {
// per-step sums [sum A, sum A grid_0, sum A grid_1] of the accumulator -> page-locked host memory (T, 3) (+ T spare doubles
// behind them); d_inv: T doubles of device scratch
struct RowStats { double *red, *d_inv; };      // red: (T, RS_W) page-locked, + T spare doubles behind it
constexpr int RS_W = 1 + bln::MAXD;
RowStats accum_row_stats(blhip_ctx *ctx, const blhip_problem *p) {
    if (!ctx->acc_active) fail("no active accumulator");
    if (!p) fail("problem is NULL");
    HIPCHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t T = ctx->acc_T, G = ctx->acc_G;
    bln::NdGrid ng{};
    ng.ndim = p->ndim;
    long long Gp = 1;
    for (int k = p->ndim - 1; k >= 0; --k) { ng.n[k] = (int)p->n[k]; ng.stride[k] = Gp; Gp *= p->n[k]; }
    ng.G = Gp;
    if (Gp != G) fail("accumulator / grid mismatch");
    const unsigned gx = (unsigned)std::min<long long>((G + NTHREADS - 1) / NTHREADS, 256);
    size_t bytes = carve_size((size_t)T * RS_W * gx * 8) + carve_size((size_t)T * RS_W * 8) + carve_size((size_t)T * 8);
    for (int k = 0; k < p->ndim; ++k) bytes += carve_size(8 * (size_t)p->n[k]);
    ctx->stats.ensure(bytes);
    char *cur = ctx->stats.as<char>();
    double *d_part = carve<double>(cur, (size_t)T * RS_W * gx);
    double *d_red = carve<double>(cur, (size_t)T * RS_W);
    double *d_inv = carve<double>(cur, (size_t)T);
    for (int k = 0; k < p->ndim; ++k) {
        double *dm = carve<double>(cur, (size_t)p->n[k]);
        HIPCHECK(hipMemcpyAsync(dm, p->marginal[k], 8 * (size_t)p->n[k], hipMemcpyHostToDevice, st));
        ng.m[k] = dm;
    }
    BL_LAUNCH(bln::row_stats_kernel, dim3(gx, (unsigned)T), dim3(NTHREADS), 0, st, ctx->acc, ng, d_part);
    BL_LAUNCH(reduce_partials_kernel, dim3((unsigned)(T * RS_W)), dim3(NTHREADS), 0, st, d_part, d_red, (int)gx, 0);
    ctx->pinS.ensure((size_t)T * (RS_W + 1) * 8);
    double *red = ctx->pinS.as<double>();
    HIPCHECK(hipMemcpyAsync(red, d_red, (size_t)T * RS_W * 8, hipMemcpyDeviceToHost, st));
    sync_stream(ctx, st);
    ctx->acc_n0 = p->ndim == 1 ? 1 : (int)p->n[0];
    ctx->acc_n1 = (int)(G / ctx->acc_n0);
    return RowStats{red, d_inv};
}
}  // namespace

int blhip_accum_row_stats(blhip_ctx *ctx, const blhip_problem *p, double *host_out) {
    return guarded(ctx, [&] {
        if (!host_out) fail("host_out is NULL");
        if (!ctx->acc_active) fail("no active accumulator");
        if (!p) fail("problem is NULL");
        const int64_t T = ctx->acc_T;
        const int w = 1 + p->ndim;
        if (ctx->acc_first) { std::fill(host_out, host_out + T * w, 0.0); return; }
        const double *red = accum_row_stats(ctx, p).red;
        for (int64_t t = 0; t < T; ++t)
            for (int k = 0; k < w; ++k) host_out[t * w + k] = red[t * RS_W + k];
    });
}

int blhip_accum_finalize(blhip_ctx *ctx, const blhip_problem *p, double *posterior_mean) {
    return guarded(ctx, [&] {
        if (!ctx->acc_active) fail("no active accumulator");
        if (ctx->acc_first) fail("blhip_accum_finalize: nothing was accumulated");
        const RowStats rs = accum_row_stats(ctx, p);
        double *red = rs.red, *d_inv = rs.d_inv;
        hipStream_t st = ctx->stream;
        const int64_t T = ctx->acc_T, G = ctx->acc_G;
        double *inv = red + (size_t)T * RS_W;
        for (int64_t t = 0; t < T; ++t) {
            inv[t] = 1.0 / red[t * RS_W];                                        // core.py:1379-1382
            if (posterior_mean)
                for (int k = 0; k < p->ndim; ++k) posterior_mean[k * T + t] = red[t * RS_W + 1 + k] / red[t * RS_W];   // :1416-1419
        }
        HIPCHECK(hipMemcpyAsync(d_inv, inv, T * 8, hipMemcpyHostToDevice, st));
        const unsigned gs = (unsigned)std::min<long long>((G + NTHREADS - 1) / NTHREADS, 4096);
        BL_LAUNCH(scale_rows_kernel, dim3(gs, (unsigned)T), dim3(NTHREADS), 0, st, ctx->acc, (long long)G, d_inv);
        sync_stream(ctx, st);
        ctx->acc_final = true;
    });
}

int blhip_accum_read(blhip_ctx *ctx, int64_t t0, int64_t t1, double *host_out) {
    return guarded(ctx, [&] {
        if (!ctx->acc_active) fail("no active accumulator");
        if (t0 < 0 || t1 > ctx->acc_T || t0 > t1 || !host_out) fail("blhip_accum_read: bad range");
        HIPCHECK(hipSetDevice(ctx->device));
        const char *src = reinterpret_cast<const char *>(ctx->acc + (size_t)t0 * ctx->acc_G);
        for_pinned_pieces(host_out, (size_t)(t1 - t0) * ctx->acc_G * 8, [&](char *dst, size_t n) {
            HIPCHECK(hipMemcpyAsync(dst, src + (dst - (char *)host_out), n, hipMemcpyDeviceToHost, ctx->stream));
        });
        sync_stream(ctx, ctx->stream);
    });
}

int blhip_accum_end(blhip_ctx *ctx) {
    return guarded(ctx, [&] {
        ctx->acc_active = false;
        ctx->acc = nullptr;
    });
}

}  // extern "C"
