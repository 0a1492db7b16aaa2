"""Seeded random study configurations (case dictionaries in the format of tests/cases.py) shared by the GPU parity tests
(HIP path vs oracle) and the CPU test that cross-checks the oracle against the imported reference."""
import numpy as np

import cases
import tolerances


def random_case(seed):
    """A seeded random configuration (study kind, ragged grid sizes, stencil radii from 0 to ~45 cells, missing data,
    fit flags) small enough for the oracle to finish in about a second."""
    rng = np.random.default_rng(1000 + seed)
    kind = ['1d_poisson', '1d_gm', '2d_axis0', '2d_both', '2d_axis1', 'hyper_axis0', 'hyper_both', 'cp', 'aligned_axis0', 'aligned_hyper'][seed % 10]
    T = int(rng.integers(1, 21))
    flags = [dict(), dict(), dict(forwardOnly=True), dict(evidenceOnly=True)][int(rng.integers(0, 4))]
    nan_at = sorted(set(int(x) for x in rng.integers(0, T, size=int(rng.integers(0, 3))))) if T > 2 else []
    data = ('series_nan', 500 + seed, T, nan_at) if nan_at else ('series', 500 + seed, T)

    def sigma(span, n, radius):              # random-walk width whose stencil radius is about `radius` cells
        return max(radius, 0.3) / 4.0 * span / max(n - 1, 1)

    if kind == '1d_poisson':
        n = int(rng.integers(2, 12000))
        counts = rng.poisson(3.0, T)
        return dict(study='Study', data=counts, om=('Poisson', [('rate', ('oint', 0, 8, n))], 'default'),
                    tm=('GRW', 'sigma', sigma(8, n, rng.integers(0, 46)), 'rate', None), fit=flags)
    if kind == '1d_gm':
        n = int(rng.integers(2, 12000))
        return dict(study='Study', data=('gm', 600 + seed, T), om=('GaussianMean', [('mean', ('cint', -6, 6, n))], 'default'),
                    tm=('GRW', 'sigma', sigma(12, n, rng.integers(0, 46)), 'mean', None), fit=flags)
    if kind.startswith('aligned'):           # whole tiles: the 32-bit-offset (LEAN) matrix-pipe kernels, several column blocks
        n0, n1 = 32 * int(rng.integers(1, 9)), 16 * int(rng.integers(1, 17))
        om = ('Gaussian', [('mean', ('cint', -5, 5, n0)), ('std', ('oint', 0, 3, n1))], 'default')
        if kind == 'aligned_axis0':
            return dict(study='Study', data=data, om=om, tm=('GRW', 's1', sigma(10, n0, rng.integers(8, 46)), 'mean', None), fit=flags)
        return dict(study='HyperStudy', data=data, om=om, fit=flags,
                    tm=('GRW', 'sigma', ('cint', sigma(10, n0, 6), sigma(10, n0, 45), int(rng.integers(2, 6))), 'mean', None))
    big = kind in ('2d_axis0', '2d_both', '2d_axis1')
    n0, n1 = int(rng.integers(3, 421 if big else 201)), int(rng.integers(3, 421 if big else 201))
    om = ('Gaussian', [('mean', ('cint', -5, 5, n0)), ('std', ('oint', 0, 3, n1))], 'default')
    r0, r1 = int(rng.integers(0, 46)), int(rng.integers(0, 9))
    if kind == '2d_axis0':
        return dict(study='Study', data=data, om=om, tm=('GRW', 's1', sigma(10, n0, r0), 'mean', None), fit=flags)
    if kind == '2d_axis1':
        return dict(study='Study', data=data, om=om, tm=('GRW', 's2', sigma(3, n1 + 2, r1), 'std', None), fit=flags)
    if kind == '2d_both':
        return dict(study='Study', data=data, om=om, fit=flags,
                    tm=('Combined', [('GRW', 's1', sigma(10, n0, r0), 'mean', None), ('GRW', 's2', sigma(3, n1 + 2, r1), 'std', None)]))
    nh = int(rng.integers(2, 7))
    if kind == 'hyper_axis0':
        return dict(study='HyperStudy', data=data, om=om, fit=flags,
                    tm=('GRW', 'sigma', ('cint', 0, sigma(10, n0, 45), nh), 'mean', None))
    if kind == 'hyper_both':
        return dict(study='HyperStudy', data=data, om=om, fit=flags,
                    tm=('Combined', [('GRW', 's1', ('cint', 0, sigma(10, n0, r0 + 1), nh), 'mean', None),
                                     ('GRW', 's2', sigma(3, n1 + 2, r1), 'std', None)]))
    T = max(T, 6)
    return dict(study='ChangepointStudy', data=('series_jump', 700 + seed, T, T // 2, 2.0), om=om, tm=('ChangePoint', 'tc', 'all', None))


def random_model_case(seed):
    """Seeded random configurations of the transition / observation models beyond GRW + Gaussian/Poisson (generic kernel,
    device-built likelihood tables): ragged small grids, random hyper-parameter values."""
    rng = np.random.default_rng(5000 + seed)
    kind = ['rs_1d', 'rs_2d', 'ne_1d', 'bivariate', 'alphastable', 'deterministic', 'serial', 'independent',
            'bernoulli', 'laplace', 'whitenoise', 'ar1', 'scaledar1'][seed % 13]
    T = int(rng.integers(2, 13))
    n = int(rng.integers(5, 400))
    n0, n1 = int(rng.integers(4, 70)), int(rng.integers(4, 70))
    pois = ('Poisson', [('rate', ('oint', 0, 8, n))], 'default')
    g2 = ('Gaussian', [('mean', ('cint', -5, 5, n0)), ('std', ('oint', 0, 3, n1))], 'default')
    counts = rng.poisson(3.0, T)
    ser = ('series', 900 + seed, T)
    s1d = float(rng.uniform(0.02, 0.6))
    tol = None
    if kind == 'rs_1d':
        c = dict(study='Study', data=counts, om=pois, tm=('Combined', [('GRW', 's', s1d, 'rate', None), ('RS', 'p', float(rng.uniform(-7, -1)), None)]))
    elif kind == 'rs_2d':
        order = [('RS', 'p', float(rng.uniform(-7, -2)), None), ('GRW', 's1', float(rng.uniform(0.1, 0.8)), 'mean', None)]
        c = dict(study='Study', data=ser, om=g2, tm=('Combined', order if seed % 2 else order[::-1]))
    elif kind == 'ne_1d':
        c = dict(study='HyperStudy', data=counts, om=pois, tm=('NE', 'p', [float(x) for x in sorted(rng.uniform(-7, -1, 3))], None))
    elif kind == 'bivariate':
        c = dict(study='Study', data=ser, om=g2, tm=('Bivariate', 's1', float(rng.uniform(0.2, 0.9)), 's2', float(rng.uniform(0.05, 0.3)),
                                                       'rho', float(rng.uniform(-0.8, 0.8))))
    elif kind == 'alphastable':
        c = dict(study='Study', data=counts, om=pois, tm=('AlphaStable', 'c', float(rng.uniform(0.05, 0.4)), 'alpha', float(rng.uniform(0.8, 2.0)), 'rate'))
        tol = cases.FFT_TOL
    elif kind == 'deterministic':
        c = dict(study='HyperStudy', data=ser, om=g2, tm=('Deterministic', ['quadratic', 'drift'][seed % 2], ['mean', 'std'][(seed // 2) % 2]))
        # a dozen cubic-spline shifts of a distribution that runs into the grid edge leave ringing (negative lobes); the
        # reference then divides by a sum that can be 1e-2 .. 1e-3 of the mass (or negative), which amplifies any rounding
        # difference by that factor per step: seen 8e-10 relative in one of 6000 configurations
        tol = tolerances.DETERMINISTIC_FUZZ_TOL
    elif kind == 'serial':
        tb = int(rng.integers(1, T - 1)) if T > 2 else 0
        c = dict(study='Study', data=counts, om=pois,
                 tm=('Serial', [('GRW', 'sa', s1d, 'rate', None), ('BreakPoint', 'tb', tb, None), ('Static',)]))
    elif kind == 'independent':
        c = dict(study='Study', data=counts, om=pois, tm=('Independent',))
    elif kind == 'bernoulli':
        c = dict(study='Study', data=rng.integers(0, 2, T).astype(float), om=('Bernoulli', [('p', ('oint', 0, 1, n))], 'default'),
                 tm=('GRW', 's', float(rng.uniform(0.01, 0.2)), 'p', None))
    elif kind == 'laplace':
        c = dict(study='Study', data=ser, om=('Laplace', [('mu', ('cint', -4, 4, n0)), ('b', ('oint', 0, 3, n1))], 'default'),
                 tm=('GRW', 's', float(rng.uniform(0.1, 0.6)), 'mu', None))
    elif kind == 'whitenoise':
        c = dict(study='Study', data=ser, om=('WhiteNoise', [('std', ('oint', 0, 3, n))], 'default'),
                 tm=('GRW', 's', float(rng.uniform(0.02, 0.3)), 'std', None))
    elif kind == 'ar1':
        c = dict(study='Study', data=ser, om=('AR1', [('rho', ('oint', -1, 1, n0)), ('sigma', ('oint', 0, 3, n1))], 'default'),
                 tm=('GRW', 's', float(rng.uniform(0.02, 0.3)), 'rho', None))
    else:
        c = dict(study='Study', data=ser, om=('ScaledAR1', [('rho', ('oint', -1, 1, n0)), ('sigma', ('oint', 0, 3, n1))], 'default'),
                 tm=('GRW', 's', float(rng.uniform(0.05, 0.4)), 'sigma', None))
    return c, tol


def random_hyper_case(seed):
    """Seeded random hyper-studies: two hyper-parameters, hyper-priors (array / function), observation-model priors,
    multi-dimensional data, serial models with break- and change-points, RegimeSwitch inside a ChangepointStudy."""
    rng = np.random.default_rng(9000 + seed)
    kind = ['two_hp', 'hyper_prior_array', 'hyper_prior_function', 'om_prior', 'multidim', 'serial_cps', 'cp_grw_rs', 'two_cp', 'cps_1d'][seed % 9]
    T = int(rng.integers(4, 15))
    n0, n1 = int(rng.integers(4, 90)), int(rng.integers(4, 90))
    g2 = ('Gaussian', [('mean', ('cint', -5, 5, n0)), ('std', ('oint', 0.2, 3, n1))], ['default', 'inv_s3', 'inv_s_2d'][int(rng.integers(0, 3))])
    ser = ('series', 1300 + seed, T)
    flags = [dict(), dict(), dict(forwardOnly=True), dict(evidenceOnly=True)][int(rng.integers(0, 4))]
    sg = lambda k: ('cint', float(rng.uniform(0.0, 0.1)), float(rng.uniform(0.2, 0.9)), k)
    if kind == 'cps_1d':        # 1-D grid: the K-steps-per-launch kernels with restarts at every offset inside a launch
        T1 = int(rng.integers(6, 41))
        n = int(rng.integers(3, 2500))
        jump = int(rng.integers(2, T1 - 2))
        counts = np.concatenate([rng.poisson(2.0, jump), rng.poisson(5.0, T1 - jump)])
        tm = ('ChangePoint', 'tc', 'all', None) if seed % 2 else ('Combined', [('GRW', 's', float(rng.uniform(0.02, 0.5)), 'rate', None),
                                                                              ('ChangePoint', 'tc', ('arange', 1, T1 - 1, int(rng.integers(1, 4))), None)])
        return dict(study='ChangepointStudy', data=counts, om=('Poisson', [('rate', ('oint', 0, 9, n))], 'default'), tm=tm)
    if kind == 'two_hp':
        return dict(study='HyperStudy', data=ser, om=g2, fit=flags,
                    tm=('Combined', [('GRW', 's1', sg(int(rng.integers(2, 5))), 'mean', None),
                                     ('GRW', 's2', ('cint', 0.01, float(rng.uniform(0.05, 0.3)), int(rng.integers(2, 4))), 'std', None)]))
    if kind == 'hyper_prior_array':
        k = int(rng.integers(2, 6))
        return dict(study='HyperStudy', data=ser, om=g2, fit=flags,
                    tm=('GRW', 's1', sg(k), 'mean', ('array', [float(x) for x in rng.uniform(0.1, 1.0, k)])))
    if kind == 'hyper_prior_function':
        return dict(study='HyperStudy', data=ser, om=g2, fit=flags,
                    tm=('GRW', 's1', ('cint', 0.05, float(rng.uniform(0.2, 0.9)), int(rng.integers(2, 6))), 'mean', 'inv_s'))
    if kind == 'om_prior':
        n = int(rng.integers(3, 500))
        if (seed // 9) % 2:      # array prior: a single fit (the reference normalises and then MUTATES the user's array in place,
            #                      core.py:208-221 / :382, so its hyper-studies depend on the chain order; not replicated)
            return dict(study='Study', data=rng.poisson(3.0, T), fit=flags, om=('Poisson', [('rate', ('oint', 0, 8, n))], ('ones', n)),
                        tm=('GRW', 's', float(rng.uniform(0.1, 1.0)), 'rate', None))
        return dict(study='HyperStudy', data=rng.poisson(3.0, T), fit=flags, om=('Poisson', [('rate', ('oint', 0, 8, n))], 'inv_x'),
                    tm=('GRW', 's', ('cint', 0.0, float(rng.uniform(0.1, 1.0)), int(rng.integers(2, 7))), 'rate', None))
    if kind == 'multidim':
        return dict(study='HyperStudy', data=('series2d', 1400 + seed, max(T, 5)), om=g2, fit=flags,
                    tm=('GRW', 's1', sg(int(rng.integers(2, 5))), 'mean', None))
    if kind == 'serial_cps':
        T = max(T, 8)
        return dict(study='ChangepointStudy', data=('series_jump', 1500 + seed, T, T // 2, 1.5), om=g2,
                    tm=('Serial', [('GRW', 'sa', float(rng.uniform(0.05, 0.4)), 'mean', None),
                                   ('BreakPoint', 'b1', ('arange', 1, T - 2, int(rng.integers(1, 3))), None), ('Static',)]))
    if kind == 'cp_grw_rs':
        T = max(T, 6)
        return dict(study='ChangepointStudy', data=('series_jump', 1600 + seed, T, T // 3, 2.0), om=g2,
                    tm=('Combined', [('GRW', 's1', float(rng.uniform(0.05, 0.4)), 'mean', None),
                                     ('RS', 'p', float(rng.uniform(-7, -2)), None), ('ChangePoint', 'tc', 'all', None)]))
    T = max(T, 9)
    return dict(study='ChangepointStudy', data=('series_jump', 1700 + seed, T, T // 2, 2.0), om=g2,
                tm=('Combined', [('ChangePoint', 't1', ('arange', 1, T - 1, 3), None), ('ChangePoint', 't2', ('arange', 2, T - 1, 3), None)]))


def random_online_case(seed):
    """Seeded random OnlineStudy: 1-3 competing transition models with random hyper-grids (and a random model prior), stepped
    over a short series (reference core.py:2062-2226)."""
    rng = np.random.default_rng(13000 + seed)
    two_d = seed % 2 == 1
    T = int(rng.integers(3, 11))
    if two_d:
        n0, n1 = int(rng.integers(4, 60)), int(rng.integers(4, 60))
        om = ('Gaussian', [('mean', ('cint', -5, 5, n0)), ('std', ('oint', 0.2, 3, n1))], 'default')
        target, data = 'mean', ('series', 1900 + seed, T)
    else:
        n = int(rng.integers(4, 600))
        om = ('Poisson', [('rate', ('oint', 0, 8, n))], 'default')
        target, data = 'rate', [int(x) for x in rng.poisson(3.0, T)]
    pool = [
        lambda k: ('Static',),
        lambda k: ('GRW', 's%d' % k, [float(x) for x in sorted(rng.uniform(0.02, 0.8, int(rng.integers(1, 5))))], target, None),
        lambda k: ('RS', 'p%d' % k, [float(x) for x in sorted(rng.uniform(-7, -1, int(rng.integers(1, 4))))], None),
        lambda k: ('Independent',),
        lambda k: ('ChangePoint', 'tc%d' % k, [int(x) for x in sorted(set(rng.integers(0, T, 2)))], None),
        lambda k: ('Combined', [('GRW', 'sa%d' % k, float(rng.uniform(0.05, 0.5)), target, None),
                                ('RS', 'q%d' % k, [float(x) for x in sorted(rng.uniform(-6, -2, 2))], None)]),
        lambda k: ('NE', 'ne%d' % k, [float(x) for x in sorted(rng.uniform(-7, -2, 2))], None),
    ]
    nm = int(rng.integers(1, 4))
    picks = rng.choice(len(pool), size=nm, replace=False)
    models = [('M%d' % k, pool[int(p)](k)) for k, p in enumerate(picks)]
    c = dict(om=om, models=models, data=data)
    if nm > 1 and seed % 3 == 0:
        w = rng.uniform(0.1, 1.0, nm)
        c['tm_prior'] = [float(x) for x in w / w.sum()]
    return c


def random_chain_resident_case(seed, ragged=False, tall=False):
    """Seeded random studies inside the envelope of the chain-resident kernel (blhip_chainres.hpp): hyper-studies over the width of one
    random walk on the first parameter (radius 0 .. 40; every fourth study up to 79, hyper-priors, observation-model priors, missing / multi-dimensional data,
    every fit mode) and change-point studies without a stencil, on grids of 128 / 256 / 512 rows x a multiple of 16 columns."""
    rng = np.random.default_rng(12000 + seed)
    n0 = [128, 128, 256, 512][int(rng.integers(0, 4))]
    n1 = 16 * int(rng.integers(1, 9 if n0 < 512 else 5))
    T = int(rng.integers(1, 15))
    if ragged:
        # any grid of 48 .. 512 rows x any number of columns: the kernels work on the next geometry of 128 / 256 / 512 rows x a multiple
        # of 16 columns (padded cells hold zeros, the stencil reflects at the grid's true last row)
        # (>= 64 rows x >= 16 columns: what the launch-per-step column kernels -- the fall-back of the resident paths -- need for a
        #  radius of up to 40)
        n0 = int(rng.integers(64, 513)) if seed % 5 else [64, 127, 129, 255, 257, 511][int(rng.integers(0, 6))]
        n1 = int(rng.integers(16, 130 if n0 <= 256 else 70))
    if tall:
        # 513 .. 1024 rows: the 1024-row geometry (one copy of the strip in LDS), bands up to radius 80; hyper-studies only (the kernels
        # of that geometry always filter), exact (1024 x a multiple of 16) every fourth seed
        n0 = 1024 if seed % 4 == 0 else int(rng.integers(513, 1025))
        n1 = 16 * int(rng.integers(1, 4)) if seed % 4 == 0 else int(rng.integers(16, 50))
        T = int(rng.integers(1, 9))
    lo, hi = -float(rng.uniform(3, 8)), float(rng.uniform(3, 8))
    prior = ['default', 'inv_s3', 'inv_s_2d'][int(rng.integers(0, 3))]
    om = ('Gaussian', [('mean', ('cint', lo, hi, n0)), ('std', ('oint', float(rng.uniform(0.0, 0.3)), float(rng.uniform(1.5, 4)), n1))], prior)
    kind = ['hyper', 'hyper', 'hyper_nan', 'hyper_2d', 'hyper_prior', 'changepoints'][seed % 6]
    if tall and kind == 'changepoints':
        kind = 'hyper'
    if kind == 'changepoints':
        T = max(T, 6)
        tm = ('ChangePoint', 'tc', 'all' if seed % 4 else ('arange', 1, T - 1, int(rng.integers(1, 4))), None)
        flags = [dict(), dict(evidenceOnly=True)][int(rng.integers(0, 2))]
        return dict(study='ChangepointStudy', data=('series_jump', 1700 + seed, T, T // 2, float(rng.uniform(-2, 2))), om=om, tm=tm, fit=flags)
    lattice = (hi - lo) / (n0 - 1)
    smax = float(rng.uniform(0.5, 9.8)) * lattice                    # radius int(4 sigma / lattice + 0.5) <= 39
    k = int(rng.integers(2, 41 if n0 * n1 <= 128 * 64 else 13))
    if tall:
        smax = float(rng.uniform(0.5, 19.8)) * lattice               # ... <= 79
        k = int(rng.integers(2, 8))
    elif seed % 4 == 3 and n0 > 90:
        smax = float(rng.uniform(10.0, 19.8)) * lattice              # (every fourth study: the wide bands, radius 41 .. 79)
        k = min(k, 9)
    sig = ('cint', 0.0 if seed % 3 == 0 else float(rng.uniform(0.0, 0.3)) * smax, smax, k)
    flags = [dict(), dict(), dict(forwardOnly=True), dict(evidenceOnly=True)][int(rng.integers(0, 4))]
    # (padded grids: evidence-only fits and full fits fold in the backward kernel; forward-only fits hand their filtered distributions out
    #  through the de-padding copy)
    data = ('series', 1800 + seed, T)
    hp = None
    if kind == 'hyper_nan' and T >= 3:
        data = ('series_nan', 1800 + seed, T, sorted(set(int(x) for x in rng.integers(0, T, int(rng.integers(1, 3))))))
    if kind == 'hyper_2d':
        data = ('series2d', 1800 + seed, max(T, 5))
    if kind == 'hyper_prior':
        hp = ('array', [float(x) for x in rng.uniform(0.1, 1.0, k)]) if seed % 2 else 'inv_s'
        if hp == 'inv_s':
            sig = ('cint', 0.05 * smax, smax, k)
    return dict(study='HyperStudy', data=data, om=om, fit=flags, tm=('GRW', 's1', sig, 'mean', hp))


def random_chain_mixed_case(seed):
    """Seeded random studies with a random walk on the first parameter AND a change point in one combined model (either list order;
    transitionModels.py:645-649 applies the sub-models in list order in both directions): restarts inside filtering chains of the
    chain-resident kernels, on aligned and padded grids, full and evidence-only fits, hyper- and change-point studies."""
    rng = np.random.default_rng(15000 + seed)
    if seed % 2:
        n0 = [128, 256, 384, 512][int(rng.integers(0, 4))]
        n1 = 16 * int(rng.integers(1, 5))
    else:
        n0 = int(rng.integers(64, 400))
        n1 = int(rng.integers(16, 70))
    T = int(rng.integers(6, 16))
    lo, hi = -float(rng.uniform(3, 8)), float(rng.uniform(3, 8))
    om = ('Gaussian', [('mean', ('cint', lo, hi, n0)), ('std', ('oint', float(rng.uniform(0.0, 0.3)), float(rng.uniform(1.5, 4)), n1))], 'default')
    lattice = (hi - lo) / (n0 - 1)
    smax = float(rng.uniform(0.5, 9.0)) * lattice
    nsig = int(rng.integers(1, 5))
    sig = float(smax) if nsig == 1 else ('cint', float(rng.uniform(0.0, 0.4)) * smax, smax, nsig)
    grw = ('GRW', 'sigma', sig, 'mean', None)
    cp = ('ChangePoint', 'tChange', ('arange', int(rng.integers(1, 3)), T - 1, int(rng.integers(1, 4))), None)
    order = [cp, grw] if rng.integers(0, 2) else [grw, cp]
    study = 'ChangepointStudy' if seed % 3 else 'HyperStudy'
    flags = [dict(), dict(), dict(evidenceOnly=True)][int(rng.integers(0, 3))]
    return dict(study=study, data=('series_jump', 1900 + seed, T, T // 2, float(rng.uniform(-2, 2))), om=om, tm=('Combined', order), fit=flags)


def random_wide_axis1_case(seed):
    """Seeded studies whose random walk on the SECOND parameter is wider than the fused kernels' 8-column halo (radius 9 .. 64 grid steps here; the kernel takes up to 256):
    the axis-1 pre-pass (blhip_hwide.hpp) in front of the streaming kernels -- single fits, hyper-studies over that width (small and wide
    radii in one batch), walks on both parameters, change points on top, ragged grids, missing data, every fit flag."""
    rng = np.random.default_rng(9000 + seed)
    kind = ['study_both', 'study_axis1', 'hyper_both', 'hyper_axis1', 'cp_walk', 'hyper_pairs', 'study_wide_v', 'hyper_wide_v'][seed % 8]
    T = int(rng.integers(2, 15))
    flags = [dict(), dict(), dict(forwardOnly=True), dict(evidenceOnly=True)][int(rng.integers(0, 4))]
    nan_at = sorted(set(int(x) for x in rng.integers(0, T, size=int(rng.integers(0, 3))))) if T > 3 else []
    data = ('series_nan', 800 + seed, T, nan_at) if nan_at else ('series', 800 + seed, T)
    n0, n1 = int(rng.integers(40, 301)), int(rng.integers(24, 421))
    om = ('Gaussian', [('mean', ('cint', -5, 5, n0)), ('std', ('oint', 0, 3, n1))], 'default')

    def sigma(span, n, radius):
        return max(radius, 0.3) / 4.0 * span / max(n - 1, 1)

    r1 = int(rng.integers(9, min(64, n1 - 2) + 1))
    r0 = int(rng.integers(0, 41))
    if kind.endswith('wide_v'):              # ... and on the FIRST parameter wider than the matrix-pipe kernels' band: the column pre-pass
        n0 = int(rng.integers(150, 421))
        om = ('Gaussian', [('mean', ('cint', -5, 5, n0)), ('std', ('oint', 0, 3, n1))], 'default')
        r0 = int(rng.integers(41, min(128, n0 - 2) + 1))
        r1 = int(rng.integers(0, min(64, n1 - 2) + 1))
    s1, s2 = sigma(10, n0, r0), sigma(3, n1 + 2, r1)
    nh = int(rng.integers(2, 6))
    if kind == 'study_wide_v':
        return dict(study='Study', data=data, om=om, fit=flags,
                    tm=('Combined', [('GRW', 's1', sigma(10, n0, r0), 'mean', None), ('GRW', 's2', sigma(3, n1 + 2, r1), 'std', None)]))
    if kind == 'hyper_wide_v':
        return dict(study='HyperStudy', data=data, om=om, fit=flags, tm=('GRW', 's1', ('cint', 0, sigma(10, n0, r0), nh), 'mean', None))
    if kind == 'study_both':
        return dict(study='Study', data=data, om=om, fit=flags,
                    tm=('Combined', [('GRW', 's1', s1, 'mean', None), ('GRW', 's2', s2, 'std', None)]))
    if kind == 'study_axis1':
        return dict(study='Study', data=data, om=om, tm=('GRW', 's2', s2, 'std', None), fit=flags)
    if kind == 'hyper_both':
        return dict(study='HyperStudy', data=data, om=om, fit=flags,
                    tm=('Combined', [('GRW', 's1', s1, 'mean', None), ('GRW', 's2', ('cint', 0, s2, nh), 'std', None)]))
    if kind == 'hyper_axis1':
        return dict(study='HyperStudy', data=data, om=om, fit=flags, tm=('GRW', 's2', ('cint', sigma(3, n1 + 2, 2), s2, nh), 'std', None))
    if kind == 'hyper_pairs':
        return dict(study='HyperStudy', data=data, om=om, fit=flags,
                    tm=('Combined', [('GRW', 's1', ('cint', 0, s1, int(rng.integers(2, 4))), 'mean', None),
                                     ('GRW', 's2', ('cint', 0, s2, int(rng.integers(2, 4))), 'std', None)]))
    T = max(T, 6)
    return dict(study='ChangepointStudy', data=('series_jump', 900 + seed, T, T // 2, 2.0), om=om, fit=dict(),
                tm=('Combined', [('ChangePoint', 'tc', ('arange', 1, T - 1, 2), None), ('GRW', 's2', s2, 'std', None)]))


def _resident_plan(n0, n1, cus=256):
    """Python replica of plan_resident (blhip.hip): -> (TR, TC, padded) or None."""
    for TR, TC in [(64, 64), (32, 64), (32, 32), (128, 128)]:
        for allow_pad in (False, True):
            def fits(n, t):
                rem = n % t
                return rem == 0 or (allow_pad and n > t and rem >= 8 and t - rem >= 8)
            if fits(n0, TR) and fits(n1, TC) and -(-n0 // TR) * -(-n1 // TC) <= cus:
                return TR, TC, bool(n0 % TR or n1 % TC)
    return None


def random_resident_case(seed):
    """Seeded single-chain 2-D studies on grids the time-resident kernel takes -- most of them NOT whole tiles (PAD kernels: masked cells,
    mirror image beyond the true edge): random sizes 64 .. 700, radii 0 .. 8 per axis, every fit flag, missing data."""
    rng = np.random.default_rng(21000 + seed)
    while True:
        n0, n1 = int(rng.integers(64, 701)), int(rng.integers(64, 701))
        plan = _resident_plan(n0, n1)
        if plan is not None and (plan[2] or seed % 4 == 0):
            break
    T = int(rng.integers(1, 13))
    flags = [dict(), dict(), dict(forwardOnly=True), dict(evidenceOnly=True)][int(rng.integers(0, 4))]
    if plan[0] == 128 and plan[2] and not flags:        # (full fits on padded 128 x 128 tiles keep the launch-per-step kernels: blhip_fit_paths.hpp)
        flags = dict(forwardOnly=True) if seed % 2 else dict(evidenceOnly=True)
    nan_at = sorted(set(int(x) for x in rng.integers(0, T, size=int(rng.integers(0, 3))))) if T > 3 else []
    data = ('series_nan', 1200 + seed, T, nan_at) if nan_at else ('series', 1200 + seed, T)
    om = ('Gaussian', [('mean', ('cint', -8, 8, n0)), ('std', ('oint', 0, 4, n1))], 'default')
    u0, u1 = float(rng.uniform(0.0, 2.0)), float(rng.uniform(0.0, 2.0))      # widths in lattice units: radius int(4 u + 0.5) <= 8
    s1, s2 = u0 * 16.0 / (n0 - 1), u1 * 4.0 / (n1 + 1)
    kind = seed % 3
    if kind == 0:
        tm = ('Combined', [('GRW', 's1', s1, 'mean', None), ('GRW', 's2', s2, 'std', None)])
    elif kind == 1:
        tm = ('GRW', 's1', s1, 'mean', None)
    else:
        tm = ('GRW', 's2', s2, 'std', None)
    return dict(study='Study', data=data, om=om, tm=tm, fit=flags)


def random_both_axes_square_case(seed):
    """Seeded studies with random walks on BOTH parameters (radii 0 .. 40 on either axis) of a square grid of 128 or 256 points per axis -- or
    of any grid of 48 .. 300 points per axis, which runs padded inside the next square geometry: the
    shapes the transposing chain-resident kernels take (blhip_chainax.hpp: blc::chainax_kernel) -- single fits, hyper-studies over one
    width, over both (pairs), a walk on the second parameter only, missing data, every fit flag."""
    rng = np.random.default_rng(12000 + seed)
    kind = ['study_both', 'hyper_pairs', 'hyper_both', 'hyper_axis1', 'study_axis1', 'hyper_pairs', 'cp_both', 'study_cp_after', 'cp_both_after'][seed % 9]
    n = [128, 256, 128][seed % 3]
    n0 = n1 = n
    if seed % 4 >= 2:                       # ragged grids inside the square geometry (PAD kernels): any sizes, also very different ones
        n0, n1 = int(rng.integers(48, 300)), int(rng.integers(48, 300))
        n = max(n0, n1)
    T = int(rng.integers(3, 9 if n > 128 else 13))
    flags = [dict(), dict(forwardOnly=True), dict(evidenceOnly=True), dict()][int(rng.integers(0, 4))]
    nan_at = sorted(set(int(x) for x in rng.integers(0, T, size=int(rng.integers(0, 3))))) if T > 3 else []
    data = ('series_nan', 1300 + seed, T, nan_at) if nan_at else ('series', 1300 + seed, T)
    om = ('Gaussian', [('mean', ('cint', -5, 5, n0)), ('std', ('oint', 0, 3, n1))], 'default')
    names = ('mean', 'std')
    if seed % 7 == 5:                       # another observation model: its likelihood comes out of a table in both layouts
        om = ('Laplace', [('mu', ('cint', -5, 5, n0)), ('b', ('oint', 0, 3, n1))], 'default')
        names = ('mu', 'b')

    def sigma(span, npts, radius):
        return max(radius, 0.3) / 4.0 * span / max(npts - 1, 1)
    r0, r1 = int(rng.integers(0, 41)), int(rng.integers(1, 41))
    if kind == 'hyper_axis1':
        r1 = max(r1, 3)                      # (the hyper-grid of that kind starts at radius 2: an ascending grid)
    s1, s2 = sigma(10, n0, r0), sigma(3, n1 + 2, r1)
    nh = int(rng.integers(2, 5))
    if kind == 'study_both':
        tm = ('Combined', [('GRW', 's1', s1, names[0], None), ('GRW', 's2', s2, names[1], None)])
        return dict(study='Study', data=data, om=om, fit=flags, tm=tm)
    if kind in ('cp_both', 'cp_both_after', 'study_cp_after'):
        # change points (transitionModels.py:289-317) beside the two walks: in front of them the restart passes through both bands,
        # behind them it is consumed unfiltered; as a ChangepointStudy over every time step, or one fixed change point in a Study
        T = max(T, 6)
        walks = [('GRW', 's1', s1, names[0], None), ('GRW', 's2', s2, names[1], None)]
        if kind == 'study_cp_after':
            return dict(study='Study', data=('series_jump', 1300 + seed, T, T // 2, 2.0), om=om, fit=flags,
                        tm=('Combined', walks + [('ChangePoint', 'tc', float(T // 2), None)]))
        cp = ('ChangePoint', 'tc', ('arange', 1, T - 1, 2), None)
        return dict(study='ChangepointStudy', data=('series_jump', 1300 + seed, T, T // 2, 2.0), om=om, fit=dict(),
                    tm=('Combined', [cp] + walks if kind == 'cp_both' else walks + [cp]))
    if kind == 'study_axis1':
        return dict(study='Study', data=data, om=om, fit=flags, tm=('GRW', 's2', s2, names[1], None))
    if kind == 'hyper_both':
        return dict(study='HyperStudy', data=data, om=om, fit=flags,
                    tm=('Combined', [('GRW', 's1', s1, names[0], None), ('GRW', 's2', ('cint', 0, s2, nh), names[1], None)]))
    if kind == 'hyper_axis1':
        return dict(study='HyperStudy', data=data, om=om, fit=flags, tm=('GRW', 's2', ('cint', sigma(3, n1 + 2, 2), s2, nh), names[1], None))
    return dict(study='HyperStudy', data=data, om=om, fit=flags,
                tm=('Combined', [('GRW', 's1', ('cint', 0, s1, int(rng.integers(2, 4))), names[0], None),
                                 ('GRW', 's2', ('cint', 0, s2, int(rng.integers(2, 4))), names[1], None)]))
