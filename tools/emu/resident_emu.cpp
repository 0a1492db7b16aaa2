// Host emulation of blr::resident_kernel (bayesloop_amd/csrc/blhip_resident.hpp): the per-thread phase functions of the
// device kernel, compiled with g++ (-DBLR_EMULATE) and run sequentially (all tiles through a phase, then the next phase),
// against a direct dense evaluation of the same forward / backward recursion.  Checks the tile / segment / halo / strip
// index logic and the lag bookkeeping before any GPU time is spent.  Development aid only: nothing ships from here.
//   g++ -O2 -std=c++17 -DBLR_EMULATE -I bayesloop_amd/csrc tools/emu/resident_emu.cpp -o /tmp/resident_emu && /tmp/resident_emu
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "blhip_resident.hpp"

using namespace blr;

static int reflect(int i, int n) {
    const int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i >= n ? p - 1 - i : i;
}

struct Problem {
    int n0, n1, T, d;
    std::vector<double> w0, w1, m0, m1, colA, colB, rec, prior;
    double step0;
};

static std::vector<double> taps(double ns) {
    std::vector<double> w(R + 1, 0.0);
    const int r = (int)(4.0 * ns + 0.5);
    if (r == 0) { w[0] = 1.0; return w; }
    double sum = 0.0;
    std::vector<double> phi(2 * r + 1);
    for (int k = -r; k <= r; ++k) { phi[k + r] = std::exp(-0.5 / (ns * ns) * k * k); sum += phi[k + r]; }
    for (int k = 0; k <= r; ++k) w[k] = phi[r + k] / sum;
    return w;
}

static void filter(std::vector<double> &x, int n0, int n1, const std::vector<double> &w, int axis) {
    std::vector<double> y(x.size());
    for (int i = 0; i < n0; ++i)
        for (int j = 0; j < n1; ++j) {
            double s = w[0] * x[(size_t)i * n1 + j];
            for (int k = R; k >= 1; --k) {
                const double a = axis == 0 ? x[(size_t)reflect(i - k, n0) * n1 + j] : x[(size_t)i * n1 + reflect(j - k, n1)];
                const double b = axis == 0 ? x[(size_t)reflect(i + k, n0) * n1 + j] : x[(size_t)i * n1 + reflect(j + k, n1)];
                s += (a + b) * w[k];
            }
            y[(size_t)i * n1 + j] = s;
        }
    x.swap(y);
}

static std::vector<double> likelihood(const Problem &p, int t) {
    std::vector<double> L((size_t)p.n0 * p.n1, 1.0);
    for (int i = 0; i < p.n0; ++i)
        for (int j = 0; j < p.n1; ++j)
            for (int q = 0; q < p.d; ++q) {
                const double x = p.rec[(size_t)t * p.d + q];
                if (x == x) L[(size_t)i * p.n1 + j] *= std::exp(-(x - p.m0[i]) * (x - p.m0[i]) * p.colA[j] - p.colB[j]);
            }
    return L;
}

template <int TR, int TC, int SEG, int CHK = 8, bool PAD = false>
static int run(int tr, int tc, int T, int lag, unsigned seed, bool one_axis, int pad0 = 0, int pad1 = 0) {
    using KF = Res<TR, TC, SEG, CHK, false, false, PAD>;
    using KB = Res<TR, TC, SEG, CHK, true, false, PAD>;
    Problem p;
    p.n0 = tr * TR - pad0; p.n1 = tc * TC - pad1; p.T = T; p.d = 1;              // (PAD: the grid does not fill its last tile row / column)
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    p.w0 = taps(one_axis ? 0.05 : 1.9); p.w1 = taps(2.05);
    p.m0.resize(p.n0); p.m1.resize(p.n1); p.colA.resize(p.n1); p.colB.resize(p.n1);
    p.step0 = 16.0 / (p.n0 - 1);
    for (int i = 0; i < p.n0; ++i) p.m0[i] = -8.0 + i * p.step0;
    for (int j = 0; j < p.n1; ++j) {
        p.m1[j] = 4.0 * (j + 1) / (p.n1 + 1);
        p.colA[j] = 1.0 / (2.0 * p.m1[j] * p.m1[j]);
        p.colB[j] = 0.5 * std::log(2.0 * M_PI * p.m1[j] * p.m1[j]);
    }
    p.rec.resize(T);
    for (int t = 0; t < T; ++t) p.rec[t] = 3.0 * (U(rng) - 0.5);
    if (T > 3) p.rec[2] = std::nan("");               // a missing data point: likelihood 1
    const size_t G = (size_t)p.n0 * p.n1;
    p.prior.resize(G);
    double ps = 0.0;
    for (auto &v : p.prior) { v = 0.1 + U(rng); ps += v; }
    for (auto &v : p.prior) v /= ps;

    // ---- direct evaluation (the reference's recursion, dense) ------------------------------------------------------------------
    std::vector<std::vector<double>> alpha(T), post(T);
    std::vector<double> norm(T), locB(T);
    {
        std::vector<double> a = p.prior;
        for (int t = 0; t < T; ++t) {
            const auto L = likelihood(p, t);
            double s = 0.0;
            for (size_t c = 0; c < G; ++c) { a[c] *= L[c]; s += a[c]; }
            norm[t] = s;
            for (auto &v : a) v /= s;
            alpha[t] = a;
            filter(a, p.n0, p.n1, p.w0, 0);
            filter(a, p.n0, p.n1, p.w1, 1);
        }
        std::vector<double> beta(G, 1.0 / G);
        for (int t = T - 1; t >= 0; --t) {
            const auto L = likelihood(p, t);
            post[t].resize(G);
            double s = 0.0, sl = 0.0;
            for (size_t c = 0; c < G; ++c) { post[t][c] = alpha[t][c] * beta[c]; s += post[t][c]; }
            for (size_t c = 0; c < G; ++c) { post[t][c] /= s; sl += post[t][c] / L[c]; }
            locB[t] = 1.0 / sl;
            for (size_t c = 0; c < G; ++c) beta[c] *= L[c];
            filter(beta, p.n0, p.n1, p.w0, 0);
            filter(beta, p.n0, p.n1, p.w1, 1);
            double bs = 0.0;
            for (auto v : beta) bs += v;
            for (auto &v : beta) v /= bs;
        }
    }

    // ---- emulation ---------------------------------------------------------------------------------------------------------------
    const int ntiles = tr * tc;
    std::vector<double> gpost((size_t)T * G, 0.0), uniform(G, 1.0 / G);
    std::vector<double> cols((size_t)2 * ntiles * 2 * R * TR), rows((size_t)2 * ntiles * 2 * R * TC);      // tagged 8-byte elements
    std::vector<unsigned long long> gran((size_t)NSLOT * ntiles * 4);
    unsigned abort_word = 0;
    ResParams Q{};
    Q.n0 = p.n0; Q.n1 = p.n1; Q.tr = tr; Q.tc = tc; Q.ntiles = ntiles; Q.T = T; Q.d = 1; Q.rec_len = 1; Q.lag = lag;
    Q.post = gpost.data(); Q.w0 = p.w0.data(); Q.w1 = p.w1.data(); Q.m0 = p.m0.data(); Q.m1 = p.m1.data();
    Q.colA = p.colA.data(); Q.colB = p.colB.data(); Q.rec = p.rec.data(); Q.step0 = p.step0;
    Q.cols = cols.data(); Q.rows = rows.data(); Q.cols_bytes = (unsigned)(cols.size() * 8); Q.rows_bytes = (unsigned)(rows.size() * 8); Q.gran = gran.data();
    Q.abort_word = &abort_word; Q.timeout_ticks = 0;

    auto pass = [&](auto tag, std::vector<double> &psum) {
        using K = decltype(tag);
        constexpr int NT = K::NT;
        std::fill(cols.begin(), cols.end(), 0.0); std::fill(rows.begin(), rows.end(), 0.0);
        std::fill(gran.begin(), gran.end(), 0ull);
        std::vector<std::vector<double>> lds(ntiles, std::vector<double>(K::LDS_DOUBLES, 0.0));
        std::vector<typename K::Thread> th((size_t)ntiles * NT);
        std::vector<int> block_of_tile(ntiles);
        for (int b = 0; b < ntiles; ++b)
            for (int t = 0; t < NT; ++t) {
                auto &x = th[(size_t)b * NT + t];
                x.init(Q, b, t, lds[b].data());
                for (int e = 0; e < TR; ++e) lds[b][K::LDS_M0 + e] = x.i0 + e < p.n0 ? p.m0[x.i0 + e] : p.m0[p.n0 - 1] + (x.i0 + e - (p.n0 - 1)) * p.step0;
                for (int e = 0; e < TC; ++e) {
                    const int c = std::min(x.j0 + e, p.n1 - 1);
                    lds[b][K::LDS_COL + e] = p.m1[c]; lds[b][K::LDS_COL + TC + e] = p.colA[c]; lds[b][K::LDS_COL + 2 * TC + e] = p.colB[c];
                }
            }
        psum.assign((size_t)T * NRED * ntiles, 0.0);
        Q.psum = psum.data();
        for (int k = 0; k < T; ++k) {
            const int tt = K::Thread::time_of(Q, k);
            for (auto &x : th) x.begin_step(Q, k);
            if (k == 0) {
                for (auto &x : th) x.first_step(Q);
            } else {
                for (auto &x : th) x.h_preread();
                if (k >= lag) for (auto &x : th) if (x.gather_wave() >= 0) x.gather_issue(Q, k - lag);
                for (auto &x : th) x.h_walk(Q, k);
                for (auto &x : th) x.alpha_issue(Q, k);
                for (auto &x : th) x.publish_rows(Q, k);
                for (auto &x : th) x.v_preread();
                if (k >= lag)
                    for (int b = 0; b < ntiles; ++b) {
                        for (int w = 0; w < K::NW; ++w) {
                            const int gw = th[(size_t)b * NT + w * 64].gather_wave();
                            if (gw < 0) continue;
                            double part[K::NG] = {};
                            for (int lane = 0; lane < 64; ++lane) {
                                double a[K::NG];
                                th[(size_t)b * NT + w * 64 + lane].gather_finish(Q, k - lag, a);
                                for (int g2 = 0; g2 < K::NG; ++g2) part[g2] += a[g2];
                            }
                            lds[b][K::LDS_MISC + 8 + gw] = part[0];
                        }
                        th[(size_t)b * NT].combine_shares();
                    }
                for (auto &x : th) x.v_walk(Q, k);
            }
            for (int b = 0; b < ntiles; ++b) {
                double v[5] = {0, 0, 0, 0, 0};
                for (int t = 0; t < NT; ++t)
                    for (int q = 0; q < 5; ++q) v[q] += th[(size_t)b * NT + t].sums[q];
                const int tile = th[(size_t)b * NT].tile;
                for (int q = 0; q < 5; ++q) psum[((size_t)tt * NRED + q) * ntiles + tile] = v[q];
                K::publish_sum(Q, tile, k, 0, K::BWD ? v[2] : v[0]);
                if (K::BWD) K::publish_sum(Q, tile, k, 1, v[0]);
            }
            for (auto &x : th) x.mirror_edges();
            for (auto &x : th) x.publish_cols(Q, k);
            for (auto &x : th) if (x.dead) { std::printf("dead thread\n"); return false; }
        }
        return true;
    };

    int bad = 0;
    // (cells of the normalised distributions: the suite's bar, |dp| <= 1e-12 + 1e-9 p -- a cell 1e-200 of the maximum carries the
    //  rounding of an exponent of ~ -500, a few 1e-11 relative after a 32-row recurrence; scalars keep their tight relative bars)
    auto check = [&](const char *what, double got, double want, double tol) {
        const bool cell = what[0] == 'a' || what[0] == 'p';
        if (!(std::fabs(got - want) <= (cell ? 1e-9 : tol) * std::fabs(want) + (cell ? 1e-12 : 1e-300)) && !(got != got && want != want)) {
            if (bad < 10) std::printf("  MISMATCH %s: got %.17g want %.17g\n", what, got, want);
            ++bad;
        }
    };
    std::vector<double> psF, psB;
    Q.src0 = p.prior.data(); Q.store = 1; Q.means = 1; Q.normalise = 0;
    if (!pass(KF{}, psF)) return 1;
    // undo the lag: S_k actual sums; norm_0 = S_0, norm_k = S_k / (S_{k-1} s_k), s_k = k >= lag ? 1 / S_{k-lag} : 1
    std::vector<double> S(T);
    for (int t = 0; t < T; ++t) { double s = 0.0; for (int b = 0; b < ntiles; ++b) s += psF[((size_t)t * NRED) * ntiles + b]; S[t] = s; }
    for (int t = 0; t < T; ++t) {
        const double sk = t >= lag ? 1.0 / S[t - lag] : 1.0;
        const double nt = t == 0 ? S[0] : S[t] / (S[t - 1] * sk);
        check("norm", nt, norm[t], 1e-12);
        double m0 = 0.0, mref = 0.0;
        for (int b = 0; b < ntiles; ++b) m0 += psF[((size_t)t * NRED + 3) * ntiles + b];
        for (size_t c = 0; c < G; ++c) mref += alpha[t][c] * p.m0[c / p.n1];
        check("mean0", m0 / S[t], mref, 1e-11);
        for (size_t c = 0; c < G; ++c) check("alpha", gpost[(size_t)t * G + c] / S[t], alpha[t][c], 1e-10);
    }
    {   // forward-only style: rows normalised in the kernel, `lag` steps behind (the last `lag` rows are left raw)
        std::vector<double> keep = gpost, psF2;
        Q.normalise = 1;
        if (!pass(KF{}, psF2)) return 1;
        for (int t = 0; t < T; ++t)
            for (size_t c = 0; c < G; ++c) check("alpha (in-kernel normalisation)", gpost[(size_t)t * G + c] / (t <= T - 1 - lag ? 1.0 : S[t]), alpha[t][c], 1e-10);
        gpost = keep;
    }
    // backward: posteriors stored normalised by the predicted sums (forward scales + the last forward row sum)
    std::vector<double> sfwd(T);
    for (int t = 0; t < T; ++t) sfwd[t] = t >= lag ? 1.0 / S[t - lag] : 1.0;
    Q.src0 = uniform.data(); Q.normalise = 0; Q.sfwd = sfwd.data(); Q.n_first = S[T - 1] / (double)G;
    if (!pass(KB{}, psB)) return 1;
    for (int t = 0; t < T; ++t) {
        double N = 0.0, Sl = 0.0;
        for (int b = 0; b < ntiles; ++b) { N += psB[((size_t)t * NRED) * ntiles + b]; Sl += psB[((size_t)t * NRED + 1) * ntiles + b]; }
        check("localEvidence", 1.0 / (Sl / N), locB[t], 1e-11);
        for (size_t c = 0; c < G; ++c) check("posterior", gpost[(size_t)t * G + c], post[t][c], 1e-10);
        (void)N;
    }
    std::printf("tile %dx%d seg %d, grid %dx%d (%d tiles), T=%d, lag=%d, %s: %s (%d mismatches)\n", TR, TC, SEG, p.n0, p.n1, ntiles, T, lag,
                one_axis ? "axis 1 only" : "both axes", bad ? "FAIL" : "ok", bad);
    return bad != 0;
}

int main() {
    int rc = 0;
    rc |= run<32, 32, 8>(2, 3, 7, 2, 1, false);
    rc |= run<32, 32, 8>(1, 1, 5, 1, 2, false);
    rc |= run<32, 32, 8>(3, 2, 9, 3, 3, true);
    rc |= run<64, 64, 8>(2, 2, 6, 2, 4, false);
    rc |= run<64, 64, 16>(2, 1, 6, 2, 5, false);
    rc |= run<128, 128, 32>(1, 2, 5, 2, 6, false);
    rc |= run<128, 128, 16, 4>(2, 1, 5, 2, 8, false);  // 1024 threads, 4-output chunks
    rc |= run<64, 64, 16, 4>(2, 2, 6, 3, 9, false);
    rc |= run<32, 32, 8, 4>(3, 3, 7, 1, 10, true);
    rc |= run<32, 64, 8>(3, 2, 7, 2, 11, false);        // rectangular tile (two blocks per CU on the device)
    rc |= run<32, 32, 8>(4, 4, 12, 2, 7, false);      // 16 tiles: the XCD-friendly block -> tile map is a permutation
    // grids that do not fill their last tile row / column (PAD kernels): mirror image beyond the true edge, masked cells
    rc |= run<32, 32, 8, 8, true>(3, 2, 8, 2, 12, false, 12, 0);        // 84 x 64: padded rows only
    rc |= run<32, 32, 8, 8, true>(2, 3, 8, 2, 13, false, 0, 20);        // 64 x 76: padded columns only
    rc |= run<32, 32, 8, 8, true>(3, 3, 9, 2, 14, false, 8, 24);        // 88 x 72: both, smallest padding / smallest remainder
    rc |= run<64, 64, 8, 8, true>(2, 2, 7, 3, 15, false, 24, 40);       // 104 x 88
    rc |= run<32, 64, 8, 8, true>(3, 2, 7, 2, 16, true, 16, 30);        // 80 x 98, rectangular tiles, one filtered axis
    rc |= run<128, 128, 32, 8, true>(2, 1, 5, 2, 17, false, 56, 28);    // 200 x 100: multi-chunk segments
    return rc;
}
