// The transition program of a batch: tap tables of the models' kernels (SciPy's gaussian_filter1d weights, spline shifts, alpha-stable and bivariate
// kernels), validation of a blhip_problem, the per-step data records, and build_program -- per (step, chain) what a step consumes and filters with
// (reference: transitionModels.py:96-118, :289-317, :632-662, :756-818; core.py:411, :467).
// Part of libblhip's host side: included by blhip.hip INSIDE its anonymous namespace (one translation unit; the split is by subject, not by linkage).
#pragma once

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
