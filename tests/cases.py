"""
Neutral descriptions of the parity cases.

The same description builds a study with the *reference* package (only inside
``tests/golden/gen_golden.py``, in the build container) and with ``bayesloop_amd`` (in the tests),
because both expose the same ``bl.Study / bl.HyperStudy / bl.ChangepointStudy / bl.om / bl.tm`` surface.
:func:`oracle_call` turns a description into a call of the CPU oracle.

Case ids starting with ``kat_`` are the reference's own known-answer tests for the path
(tests/test_transitionmodels.py, tests/test_observationmodels.py, tests/test_study.py,
tests/test_hyperstudy.py in the reference); ``c1..c5`` are the BASELINE.json configs at fixture size.
"""
from __future__ import annotations

import numpy as np

# ----------------------------------------------------------------------------------------------------------
# named priors (callables cannot be stored in fixtures; both sides look them up here)
# ----------------------------------------------------------------------------------------------------------
PRIORS = {
    'inv_s3': lambda m, s: 1 / s ** 3,
    'inv_x': lambda x: 1. / x,
    'inv_s': lambda s: 1. / s,
    'inv_s_2d': lambda m, s: 1. / s,
    'inv_s_3d': lambda df, m, s: 1. / s,
}

# the reference convolves with scipy.signal.fftconvolve (AlphaStableRandomWalk) / shifts with a recursive spline prefilter
# (Deterministic): its own round-off is ~1e-17 ABSOLUTE, not relative to the (possibly tiny) posterior value
from tolerances import FFT_TOL, WIDE_FILTER_2D_TOL   # noqa: E402  (every tolerance exception is registered in tests/tolerances.py)

COAL = np.array([5, 4, 1, 0, 4, 3, 4, 0, 6, 3, 3, 4, 0, 2, 6, 3, 3, 5, 4, 5, 3, 1, 4,
                 4, 1, 5, 5, 3, 4, 2, 5, 2, 2, 3, 4, 2, 1, 3, 2, 2, 1, 1, 1, 1, 3, 0,
                 0, 1, 0, 1, 1, 0, 0, 3, 1, 0, 3, 2, 2, 0, 1, 1, 1, 0, 1, 0, 1, 0, 0,
                 0, 2, 1, 0, 0, 0, 1, 1, 0, 2, 3, 3, 1, 1, 2, 1, 1, 1, 1, 2, 3, 3, 0,
                 0, 0, 1, 4, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0])
COAL_T = np.arange(1852, 1962)


def series(seed, T, jump_at=None, jump=0.0):
    """Synthetic mean-tracking series of SURVEY.md section 8(d)."""
    rng = np.random.default_rng(seed)
    mu = np.cumsum(rng.normal(0, 0.02, T))
    if jump_at is not None:
        mu[jump_at:] += jump
    return mu + rng.normal(0, 1.0, T)


def gm_data(seed, T):
    """GaussianMean data (value, std) of config C2."""
    x = series(seed, T)
    return np.stack([x, np.ones(T)], 1)


def _g(kind, a, b, n):
    return (kind, a, b, n)


D15 = np.array([1, 2, 3, 4, 5])
D10100 = np.array([1, 0, 1, 0, 0])
GM5 = np.array([[1, 0.5], [0, 0.4], [1, 0.3], [0, 0.2], [0, 0.1]])

G2020 = ('Gaussian', [('mean', _g('cint', 0, 6, 20)), ('sigma', _g('oint', 0, 2, 20))], 'inv_s3')


def gauss2d(n, lo=-8, hi=8, smax=4):
    return ('Gaussian', [('mean', _g('cint', lo, hi, n)), ('std', _g('oint', 0, smax, n))], 'default')


CASES = {
    # four-parameter grid (tests/golden/nd4_reference.npz): scipy.stats.johnsonsu(a, b, loc, scale), walks on two of the parameters
    'nd4_reference': dict(study='Study', data=('series', 96, 6),
                          om=('SciPy:johnsonsu', [('a', _g('cint', -1.0, 1.0, 5)), ('b', _g('cint', 0.8, 2.5, 4)), ('loc', _g('cint', -2.0, 2.0, 11)),
                                                  ('scale', _g('oint', 0.3, 2.0, 9))], 'default'),
                          tm=('Combined', [('GRW', 's_loc', 0.5, 'loc', None), ('GRW', 's_b', 0.4, 'b', None)])),
    # three-parameter grid through the reference's SciPy plug-in (tests/golden/nd3_reference.npz)
    'nd3_reference': dict(study='Study', data=('series', 95, 8),
                          om=('SciPy:t', [('df', _g('cint', 2.0, 9.0, 5)), ('loc', _g('cint', -3.0, 3.0, 16)), ('scale', _g('oint', 0.2, 2.5, 12))], 'default'),
                          tm=('Combined', [('GRW', 's_loc', 0.5, 'loc', None), ('GRW', 's_scale', 0.2, 'scale', None)])),
    # --- reference tests/test_transitionmodels.py:9-21, 40-52, 82-94
    'kat_static': dict(study='Study', data=D15, om=('Poisson', [('rate', _g('oint', 0, 6, 100))], 'default'),
                       tm=('Static',), kat=-10.372209708143769),
    'kat_grw': dict(study='Study', data=D15, om=('Poisson', [('rate', _g('oint', 0, 6, 100))], 'default'),
                    tm=('GRW', 'sigma', 0.2, 'rate', None), kat=-10.323144246611964),
    'kat_changepoint': dict(study='Study', data=D15, om=('Poisson', [('rate', _g('oint', 0, 6, 100))], 'default'),
                            tm=('ChangePoint', 't_change', 2, None), kat=-12.894336092378385),
    # --- reference tests/test_observationmodels.py:138-148, 150-160, 174-184
    'kat_poisson': dict(study='Study', data=D10100, om=('Poisson', [('rate', _g('oint', 0, 1, 100))], 'default'),
                        tm=('Static',), kat=-4.433708287229158),
    'kat_gaussian': dict(study='Study', data=D10100,
                         om=('Gaussian', [('mu', _g('oint', 0, 1, 100)), ('std', _g('oint', 0, 1, 100))], 'inv_s3'),
                         tm=('Static',), kat=-12.430583625665736),
    'kat_gaussianmean': dict(study='Study', data=GM5, om=('GaussianMean', [('mu', _g('oint', 0, 1, 100))], 'default'),
                             tm=('Static',), kat=-6.3333705075036226),
    # --- reference tests/test_observationmodels.py:126-220: the closed-form models whose likelihood table is built on the device
    'kat_bernoulli': dict(study='Study', data=D10100, om=('Bernoulli', [('p', _g('oint', 0, 1, 100))], 'default'),
                          tm=('Static',), kat=-4.3494298741972859),
    'kat_laplace': dict(study='Study', data=D10100, om=('Laplace', [('mu', None), ('b', None)], 'default'),
                        tm=('Static',), kat=-10.658573159),
    'kat_whitenoise': dict(study='Study', data=D10100, om=('WhiteNoise', [('std', _g('oint', 0, 1, 100))], 'default'),
                           tm=('Static',), kat=-6.8161638661444073),
    'kat_ar1': dict(study='Study', data=D10100, om=('AR1', [('rho', _g('oint', -1, 1, 100)), ('sigma', _g('oint', 0, 1, 100))], 'default'),
                    tm=('Static',), kat=-4.3291291450463421),
    'kat_scaledar1': dict(study='Study', data=D10100,
                          om=('ScaledAR1', [('rho', _g('oint', -1, 1, 100)), ('sigma', _g('oint', 0, 1, 100))], 'default'),
                          tm=('Static',), kat=-4.4178639067800738),
    'laplace_grw_2d': dict(study='Study', data=('series', 71, 12),
                           om=('Laplace', [('mu', _g('cint', -4, 4, 60)), ('b', _g('oint', 0, 3, 44))], 'default'),
                           tm=('Combined', [('GRW', 's1', 0.3, 'mu', None), ('GRW', 's2', 0.1, 'b', None)])),
    'whitenoise_multidim_nan': dict(study='Study', data=('series2d', 72, 10), om=('WhiteNoise', [('std', _g('oint', 0, 3, 300))], 'default'),
                                    tm=('GRW', 's', 0.05, 'std', None)),
    'scaledar1_hyper': dict(study='HyperStudy', data=('series', 73, 14),
                            om=('ScaledAR1', [('rho', _g('oint', -1, 1, 40)), ('sigma', _g('oint', 0, 3, 36))], 'default'),
                            tm=('GRW', 's', _g('cint', 0.02, 0.2, 4), 'rho', None)),
    'bernoulli_changepoint': dict(study='Study', data=np.array([1, 1, 0, 1, 1, 0, 0, 0, 1, 0, 0, 0]),
                                  om=('Bernoulli', [('p', _g('oint', 0, 1, 200))], 'default'), tm=('ChangePoint', 'tc', 5, None)),
    # --- reference tests/test_study.py:32-52 (default 1000-pt estimated grid), :80-100 (array prior), :102-...
    'kat_study_1hp': dict(study='Study', data=D15, om=('Poisson', [('rate', None)], 'default'),
                          tm=('GRW', 'sigma', 0.1, 'rate', None), kat=-10.4337420351, kat_decimal=2),
    'kat_study_prior_array': dict(study='Study', data=D15,
                                  om=('Poisson', [('rate', _g('oint', 0, 6, 1000))], ('ones', 1000)),
                                  tm=('GRW', 'sigma', 0.1, 'rate', None), kat=-10.0866227472, kat_decimal=2),
    'kat_study_prior_function': dict(study='Study', data=D15,
                                     om=('Poisson', [('rate', _g('oint', 0, 6, 1000))], 'inv_x'),
                                     tm=('GRW', 'sigma', 0.1, 'rate', None)),
    # --- reference tests/test_hyperstudy.py:10-30, 32-59, 104-187
    'kat_hyper_0hp': dict(study='HyperStudy', data=D15, om=G2020, tm=('Static',), kat=-16.1946904707),
    'kat_hyper_1hp': dict(study='HyperStudy', data=D15, om=G2020,
                          tm=('GRW', 'sigma', _g('cint', 0, 0.2, 2), 'mean', None), kat=-16.0629517262,
                          kat_hpd=[0.43828499, 0.56171501]),
    'kat_hyper_prior_array': dict(study='HyperStudy', data=D15, om=G2020,
                                  tm=('GRW', 'sigma', _g('cint', 0, 0.2, 2), 'mean', ('array', [0.2, 0.8])),
                                  kat=-15.9915077133, kat_hpd=[0.16322581, 0.83677419]),
    'kat_hyper_prior_function': dict(study='HyperStudy', data=D15, om=G2020,
                                     tm=('GRW', 'sigma', _g('cint', 0.1, 0.3, 2), 'mean', 'inv_s'),
                                     kat=-15.9898700147, kat_hpd=[0.61609973, 0.38390027]),
    # --- BASELINE.json configs
    'c1_coal': dict(study='Study', data=COAL, timestamps=COAL_T,
                    om=('Poisson', [('rate', _g('oint', 0, 6, 200))], 'default'),
                    tm=('GRW', 'sigma', 0.2, 'rate', None), kat=-171.68672187433867, kat_decimal=9),
    'c1_coal_hyper': dict(study='HyperStudy', data=COAL, timestamps=COAL_T,
                          om=('Poisson', [('rate', _g('oint', 0, 6, 200))], 'default'),
                          tm=('GRW', 'sigma', _g('cint', 0, 1, 20), 'rate', None), kat=-172.6703099789132,
                          kat_decimal=9),
    'c1_coal_changepoint': dict(study='ChangepointStudy', data=COAL, timestamps=COAL_T,
                                om=('Poisson', [('rate', _g('oint', 0, 6, 200))], 'default'),
                                tm=('ChangePoint', 'tChange', 'all', None)),
    'c2_small': dict(study='Study', data=('gm', 20260927, 300),
                     om=('GaussianMean', [('mean', _g('cint', -8, 8, 4096))], 'default'),
                     tm=('GRW', 'sigma', 0.02, 'mean', None)),
    'c2_full': dict(study='Study', data=('gm', 20260927, 10000),
                    om=('GaussianMean', [('mean', _g('cint', -8, 8, 4096))], 'default'),
                    tm=('GRW', 'sigma', 0.02, 'mean', None), kat=-14319.827959661387, kat_decimal=7,
                    store='sparse', slow=True),
    'c3_small': dict(study='Study', data=('series', 3, 10), om=gauss2d(128),
                     tm=('Combined', [('GRW', 's1', 0.24, 'mean', None), ('GRW', 's2', 0.064, 'std', None)])),
    'c3_t10': dict(study='Study', data=('series', 3, 10), om=gauss2d(1024),
                   tm=('Combined', [('GRW', 's1', 0.03, 'mean', None), ('GRW', 's2', 0.008, 'std', None)]),
                   kat=-17.470610045317024, kat_decimal=9, store='sparse', slow=True),
    'c3_missing': dict(study='Study', data=('series_nan', 3, 12, [4, 7]), om=gauss2d(64),
                       tm=('Combined', [('GRW', 's1', 0.4, 'mean', None), ('GRW', 's2', 0.1, 'std', None)])),
    'c3_forward_only': dict(study='Study', data=('series', 3, 10), om=gauss2d(96),
                            tm=('Combined', [('GRW', 's1', 0.3, 'mean', None), ('GRW', 's2', 0.09, 'std', None)]),
                            fit=dict(forwardOnly=True)),
    'c3_evidence_only': dict(study='Study', data=('series', 3, 10), om=gauss2d(96),
                             tm=('Combined', [('GRW', 's1', 0.3, 'mean', None), ('GRW', 's2', 0.09, 'std', None)]),
                             fit=dict(evidenceOnly=True)),
    'c3_std_first': dict(study='Study', data=('series', 3, 8), om=gauss2d(80),
                         tm=('Combined', [('GRW', 's2', 0.09, 'std', None), ('GRW', 's1', 0.3, 'mean', None)])),
    'c3_one_axis': dict(study='Study', data=('series', 3, 8), om=gauss2d(80),
                        tm=('GRW', 's2', 0.11, 'std', None)),
    'wide_filter': dict(study='Study', data=D15, om=('Poisson', [('rate', _g('oint', 0, 6, 40))], 'default'),
                        tm=('GRW', 'sigma', 3.5, 'rate', None)),          # lw = 97 > n = 40: multi-period reflect
    'wide_filter_2d': dict(study='Study', data=('series', 7, 6), om=gauss2d(24, -3, 3, 2),
                           tm=('Combined', [('GRW', 's1', 2.0, 'mean', None), ('GRW', 's2', 1.1, 'std', None)]),
                           # std values down to 0.08 put DENORMAL likelihood values (not yet zero) on the grid at step 0;
                           # the reference's backward localEvidence = 1/sum(post/L) is then dominated by cells whose
                           # alpha*L product keeps only a few significant bits, i.e. the golden value itself is only
                           # defined to ~1e-4 (any change of operation order moves it) -> looser bar for that one number
                           tol=WIDE_FILTER_2D_TOL),
    'tiny_sigma': dict(study='Study', data=D15, om=('Poisson', [('rate', _g('oint', 0, 6, 100))], 'default'),
                       tm=('GRW', 'sigma', 0.005, 'rate', None)),         # sigma/delta < 0.125 -> lw = 0 (identity)
    'zero_sigma': dict(study='Study', data=D15, om=('Poisson', [('rate', _g('oint', 0, 6, 100))], 'default'),
                       tm=('GRW', 'sigma', 0.0, 'rate', None)),
    'multidim_data': dict(study='Study', data=('series2d', 11, 9), om=gauss2d(48, -4, 4, 3),
                          tm=('GRW', 's1', 0.2, 'mean', None)),
    'abort_forward': dict(study='Study', data=np.array([0.2, 0.1, 500.0, 0.3]), om=gauss2d(32, -2, 2, 0.5),
                          tm=('Static',)),                                # zero normaliser at step 2 -> logE = -inf
    'c4_small': dict(study='HyperStudy', data=('series', 4, 32), om=gauss2d(128),
                     tm=('GRW', 'sigma', _g('cint', 0, 0.3, 16), 'mean', None), kat=-48.69144546025129,
                     kat_decimal=9, store='sparse'),
    'c4_small_evidence': dict(study='HyperStudy', data=('series', 4, 32), om=gauss2d(128),
                              tm=('GRW', 'sigma', _g('cint', 0, 0.3, 16), 'mean', None), fit=dict(evidenceOnly=True)),
    'c4_2hp': dict(study='HyperStudy', data=('series', 4, 12), om=gauss2d(48, -4, 4, 3),
                   tm=('Combined', [('GRW', 's1', _g('cint', 0, 0.4, 4), 'mean', None),
                                    ('GRW', 's2', _g('cint', 0.02, 0.2, 3), 'std', 'inv_s')])),
    'c4_njobs3': dict(study='HyperStudy', data=('series', 4, 12), om=gauss2d(48, -4, 4, 3),
                      tm=('GRW', 'sigma', _g('cint', 0, 0.3, 7), 'mean', None), fit=dict(nJobs=3)),
    'c5_small': dict(study='ChangepointStudy', data=('series_jump', 5, 64, 32, 2.0), om=gauss2d(128),
                     tm=('ChangePoint', 'tChange', ('arange', 3, 63, 4), None), kat=-105.38594028835217,
                     kat_decimal=9, store='sparse'),
    'c5_cp_grw': dict(study='ChangepointStudy', data=('series_jump', 5, 24, 12, 2.0), om=gauss2d(40, -4, 6, 3),
                      tm=('Combined', [('ChangePoint', 'tChange', ('arange', 2, 22, 3), None),
                                       ('GRW', 'sigma', _g('cint', 0.05, 0.25, 3), 'mean', None)])),
    'c5_grw_cp': dict(study='ChangepointStudy', data=('series_jump', 5, 24, 12, 2.0), om=gauss2d(40, -4, 6, 3),
                      tm=('Combined', [('GRW', 'sigma', 0.2, 'mean', None),
                                       ('ChangePoint', 'tChange', ('arange', 2, 22, 3), None)])),
    'c5_two_cp': dict(study='ChangepointStudy', data=('series_jump', 5, 14, 7, 2.0), om=gauss2d(24, -4, 6, 3),
                      tm=('Combined', [('ChangePoint', 't1', ('arange', 1, 12, 2), None),
                                       ('ChangePoint', 't2', ('arange', 1, 12, 2), None)])),
    # --- rows of SURVEY.md 8(f): RegimeSwitch, Independent, SerialTransitionModel + BreakPoint
    # reference tests/test_transitionmodels.py:96-108, :110-122, :139-161
    'kat_regimeswitch': dict(study='Study', data=D15, om=('Poisson', [('rate', _g('oint', 0, 6, 100))], 'default'),
                             tm=('RS', 'p_min', -3, None), kat=-10.372866559561402),
    # reference tests/test_transitionmodels.py:22-37 (decimal=3 there: SciPy's spline boundary handling changed over time)
    'kat_deterministic': dict(study='HyperStudy', data=D15, om=('Poisson', [('rate', _g('oint', 0, 6, 100))], 'default'),
                              tm=('Deterministic', 'linear_kat', 'rate'), kat=-9.4050089375418136, kat_decimal=3, tol=FFT_TOL),
    'deterministic_2d_hyper': dict(study='HyperStudy', data=('series', 95, 9), om=gauss2d(40, -5, 5, 3),
                                   tm=('Deterministic', 'quadratic', 'mean'), tol=FFT_TOL),
    'deterministic_grw_2d': dict(study='Study', data=('series', 96, 8), om=gauss2d(36, -5, 5, 3),
                                 tm=('Combined', [('Deterministic', 'drift', 'std'), ('GRW', 's', 0.3, 'mean', None)]), tol=FFT_TOL),
    # reference tests/test_transitionmodels.py:68-80
    'kat_alphastable': dict(study='Study', data=D15, om=('Poisson', [('rate', _g('oint', 0, 6, 100))], 'default'),
                            tm=('AlphaStable', 'c', 0.2, 'alpha', 1.5, 'rate'), kat=-10.122384638661309, tol=FFT_TOL),
    'alphastable_2d_hyper': dict(study='HyperStudy', data=('series', 91, 8), om=gauss2d(36, -5, 5, 3),
                                 tm=('AlphaStable', 'c', [0.1, 0.3], 'alpha', [1.0, 1.7, 2.0], 'mean'), tol=FFT_TOL),
    'alphastable_axis1_2d': dict(study='Study', data=('series', 92, 7), om=gauss2d(30, -5, 5, 3),
                                 tm=('AlphaStable', 'c', 0.08, 'alpha', 1.3, 'std'), tol=FFT_TOL),
    # reference tests/test_transitionmodels.py:53-66
    'kat_bivariate': dict(study='Study', data=D15, om=('Gaussian', [('mu', _g('oint', 0, 6, 20)), ('sigma', _g('oint', 0, 2, 20))], 'default'),
                          tm=('Bivariate', 'sigma1', 1., 'sigma2', 0.1, 'rho', 0.5), kat=-7.330706514472251),
    'bivariate_hyper': dict(study='HyperStudy', data=('series', 81, 8), om=gauss2d(40, -5, 5, 3),
                            tm=('Bivariate', 's1', [0.3, 0.6], 's2', 0.15, 'rho', [-0.4, 0.0, 0.7])),
    # reference tests/test_transitionmodels.py:124-136
    'kat_notequal': dict(study='Study', data=D15, om=('Poisson', [('rate', _g('oint', 0, 6, 100))], 'default'),
                         tm=('NE', 'p_min', -3, None), kat=-10.569099863134156),
    'notequal_2d_hyper': dict(study='HyperStudy', data=('series', 61, 9), om=gauss2d(36, -5, 5, 3),
                              tm=('NE', 'p_min', [-6., -4., -2.], None)),
    'notequal_before_grw_2d': dict(study='Study', data=('series', 62, 8), om=gauss2d(30, -5, 5, 3),
                                   tm=('Combined', [('NE', 'p_min', -4, None), ('GRW', 's', 0.4, 'mean', None)])),
    'kat_independent': dict(study='Study', data=D15, om=('Poisson', [('rate', _g('oint', 0, 6, 100))], 'default'),
                            tm=('Independent',), kat=-11.087360077190617),
    'kat_nested': dict(study='Study', data=D15, om=('Poisson', [('rate', _g('oint', 0, 6, 100))], 'default'),
                       tm=('Serial', [('Static',), ('ChangePoint', 't_change', 1, None),
                                      ('Combined', [('GRW', 'sigma', 0.2, 'rate', None), ('RS', 'p_min', -3, None)]),
                                      ('BreakPoint', 't_break', 3, None), ('Independent',)]), kat=-13.269918024215237),
    # reference tests/test_study.py:54-78 and tests/test_hyperstudy.py:61-103
    'kat_study_2hp': dict(study='Study', data=D15, om=('Poisson', [('rate', None)], 'default'),
                          tm=('Combined', [('GRW', 'sigma', 0.1, 'rate', None), ('RS', 'log10pMin', -3, None)]),
                          kat=-10.4342948181, kat_decimal=2),
    'kat_hyper_2hp': dict(study='HyperStudy', data=D15, om=G2020,
                          tm=('Combined', [('GRW', 'sigma', _g('cint', 0, 0.2, 2), 'mean', None),
                                           ('RS', 'log10pMin', [-3, -1], None)]), kat=-10.7601875492),
    # reference tests/test_changepointstudy.py:10-52
    'kat_changepointstudy': dict(study='ChangepointStudy', data=D15, om=G2020,
                                 tm=('Serial', [('Static',), ('ChangePoint', 'ChangePoint', [0, 1], None),
                                                ('Combined', [('GRW', 'sigma', _g('cint', 0, 0.2, 2), 'mean', None),
                                                              ('RS', 'log10pMin', [-3, -1], None)]),
                                                ('BreakPoint', 'BreakPoint', 'all', None), ('Static',)]),
                                 kat=-15.072007461556161),
    'rs_before_grw_2d': dict(study='Study', data=('series', 12, 9), om=gauss2d(40, -4, 4, 3),
                             tm=('Combined', [('RS', 'p', -4.5, None), ('GRW', 's1', 0.3, 'mean', None),
                                              ('GRW', 's2', 0.1, 'std', None)])),
    'rs_after_grw_2d': dict(study='HyperStudy', data=('series', 12, 9), om=gauss2d(40, -4, 4, 3),
                            tm=('Combined', [('GRW', 's1', _g('cint', 0.1, 0.5, 3), 'mean', None),
                                             ('RS', 'p', [-6, -3], None)])),
    'serial_breakpoints_2d': dict(study='ChangepointStudy', data=('series_jump', 5, 16, 8, 2.0), om=gauss2d(32, -4, 6, 3),
                                  tm=('Serial', [('GRW', 'sa', 0.1, 'mean', None), ('BreakPoint', 'b1', ('arange', 2, 14, 3), None),
                                                 ('Static',), ('BreakPoint', 'b2', ('arange', 3, 15, 3), None),
                                                 ('Combined', [('GRW', 'sb', 0.4, 'mean', None), ('GRW', 'sc', 0.1, 'std', None)])])),
    # a Deterministic sub-model INSIDE a serial model counts its time from the break-point that starts its segment (reference
    # transitionModels.py:770-776): the shape of the reference's published break-point study (docs tutorial changepointstudy.ipynb,
    # bench.py: coal_breakpoints) in small, and a NON-linear function of time, for which the offset matters
    'serial_deterministic_bp': dict(study='ChangepointStudy', data=('coal', 14), om=('Poisson', [('rate', _g('oint', 0, 6, 60))], 'default'),
                                    tm=('Serial', [('Static',), ('BreakPoint', 't_1', 'all', None), ('Deterministic', 'slopes', 'rate'),
                                                   ('BreakPoint', 't_2', 'all', None), ('Static',)]), tol=FFT_TOL),
    'serial_deterministic_offset': dict(study='HyperStudy', data=('coal', 12), om=('Poisson', [('rate', _g('oint', 0, 6, 64))], 'default'),
                                        tm=('Serial', [('GRW', 's', 0.2, 'rate', None), ('BreakPoint', 'tb', [3, 5, 8], None),
                                                       ('Deterministic', 'quad_off', 'rate')]), tol=FFT_TOL),
    'cp_nonunit_time': dict(study='Study', data=D15, timestamps=np.array([0., 2., 4., 6., 8.]),
                            om=('Poisson', [('rate', _g('oint', 0, 6, 100))], 'default'),
                            tm=('ChangePoint', 't_change', 4., None)),   # forward fires at t=4, backward at t-1: never
}


def _linear_kat(t, a=[1, 2]):                 # reference tests/test_transitionmodels.py:26-27
    return 0.5 + 0.2 * a * t


def _quadratic(t, a=0.01, b=np.array([-0.1, 0.05, 0.2])):
    return a * (t ** 2) + b * t


def _drift(t, slope=0.15):
    return slope * t


def _slopes(t, slope=np.array([-0.3, -0.1, 0.0])):
    return t * slope


def _quad_off(t, a=np.array([0.004, 0.012])):
    return -a * t ** 2


FUNCS = {'linear_kat': _linear_kat, 'quadratic': _quadratic, 'drift': _drift, 'slopes': _slopes, 'quad_off': _quad_off}


# ---- OnlineStudy (reference core.py:1963-2226, tests/test_onlinestudy.py) ----------------------
G20 = ('Gaussian', [('mean', _g('cint', 0, 6, 20)), ('sigma', _g('oint', 0, 2, 20))])
ONLINE_CASES = {
    # reference tests/test_onlinestudy.py:10-32 (setTM only: the model is added on the first step)
    'online_kat_static': dict(om=G20 + ('inv_s3',), set_tm=('Static',), data=[1, 2, 3, 4, 5], kat=-16.1946904707),
    # reference tests/test_onlinestudy.py:34-84
    'online_kat_2tm': dict(om=G20 + ('inv_s_2d',),
                           models=[('T1', ('Combined', [('GRW', 's1', [0.25, 0.5], 'mean', ('sympy_exp', 0.5)),
                                                        ('GRW', 's2', _g('cint', 0, 0.2, 2), 'sigma', ('array', [0.2, 0.8]))])),
                                   ('T2', ('Independent',))],
                           tm_prior=[0.9, 0.1], data=[1, 2, 3, 4, 5], kat=-9.46900822686),
    'online_poisson_3tm': dict(om=('Poisson', [('rate', _g('oint', 0, 6, 150))], 'default'),
                               models=[('static', ('Static',)),
                                       ('walk', ('GRW', 'sigma', _g('cint', 0.05, 0.4, 6), 'rate', None)),
                                       ('switch', ('RS', 'log10pMin', [-5, -3], None))],
                               data=COAL[:30].tolist()),
    'online_gauss2d': dict(om=('Gaussian', [('mean', _g('cint', -5, 5, 48)), ('std', _g('oint', 0, 3, 40))], 'default'),
                           models=[('walk', ('GRW', 'sm', [0.0, 0.1, 0.3], 'mean', None)),
                                   ('both', ('Combined', [('GRW', 'a', 0.2, 'mean', None),
                                                          ('GRW', 'b', _g('cint', 0.02, 0.1, 3), 'std', None)])),
                                   ('cp', ('ChangePoint', 'tc', [-1, 3], None))],
                           tm_prior=[0.5, 0.3, 0.2], data=('series', 41, 12)),
    'online_notequal': dict(om=('Poisson', [('rate', _g('oint', 0, 6, 120))], 'default'),
                            models=[('static', ('Static',)), ('different', ('NE', 'log10pMin', [-7., -3.], None)),
                                    ('walk+different', ('Combined', [('NE', 'q', -5., None), ('GRW', 'sg', 0.2, 'rate', None)]))],
                            data=COAL[30:52].tolist()),
    'online_deterministic': dict(om=('Poisson', [('rate', _g('oint', 0, 6, 120))], 'default'),
                                 models=[('static', ('Static',)), ('drift', ('Deterministic', 'quadratic', 'rate'))],
                                 data=COAL[60:75].tolist()),
    # three parameters (the reference's SciPy plug-in: Student's t with df, loc, scale; core.py:2062-2226 has no limit on the grid's dimensions)
    'online_scipy_t3': dict(om=('SciPy:t', [('df', _g('cint', 2.0, 8.0, 4)), ('loc', _g('cint', -3.0, 3.0, 14)), ('scale', _g('oint', 0.2, 2.5, 10))], 'default'),
                            models=[('static', ('Static',)),
                                    ('walks', ('Combined', [('GRW', 's_loc', [0.3, 0.6], 'loc', None), ('GRW', 's_scale', 0.2, 'scale', None)])),
                                    ('cp', ('ChangePoint', 'tc', [-1, 2], None))],
                            tm_prior=[0.4, 0.4, 0.2], data=('series', 97, 7)),
    'online_ar1_wait': dict(om=('AR1', [('rho', _g('oint', -1, 1, 30)), ('sigma', _g('oint', 0, 1, 25))], 'default'),
                            models=[('static', ('Static',)), ('walk', ('GRW', 's', [0.05, 0.1], 'rho', None))],
                            data=[1, 0, 1, 0, 0, 1]),
}


def online_data(c):
    d = c['data']
    return list(make_data(d)) if isinstance(d, tuple) else list(d)


def build_online(bl, case, storeHistory=True):
    """OnlineStudy of the given package, models added, no data yet."""
    import contextlib, io
    c = ONLINE_CASES[case] if isinstance(case, str) else case
    with contextlib.redirect_stdout(io.StringIO()):
        S = bl.OnlineStudy(storeHistory=storeHistory, silent=True)
        S.setOM(make_om(bl, c['om']), silent=True)
        if 'set_tm' in c:
            S.setTM(make_tm(bl, c['set_tm']), silent=True)
        for name, spec in c.get('models', []):
            S.addTransitionModel(name, make_tm(bl, spec))
        if c.get('tm_prior') is not None:
            S.setTransitionModelPrior(c['tm_prior'], silent=True)
    return S


def make_data(spec):
    if isinstance(spec, np.ndarray):
        return spec
    kind = spec[0]
    if kind == 'gm':
        return gm_data(spec[1], spec[2])
    if kind == 'series':
        return series(spec[1], spec[2])
    if kind == 'series_nan':
        x = series(spec[1], spec[2])
        x[list(spec[3])] = np.nan
        return x
    if kind == 'series_jump':
        return series(spec[1], spec[2], jump_at=spec[3], jump=spec[4])
    if kind == 'coal':                       # the years 1870 .. of the coal-mining counts (the data of the reference's tutorials)
        return COAL[18:18 + spec[1]]
    if kind == 'series2d':
        a = series(spec[1], spec[2])
        b = series(spec[1] + 1, spec[2])
        b[3] = np.nan
        return np.stack([a, b], 1)
    raise ValueError(spec)


def make_values(bl, v):
    if v is None or isinstance(v, (int, float, str)):
        return v
    if isinstance(v, tuple) and v[0] in ('cint', 'oint'):
        return getattr(bl, v[0])(v[1], v[2], v[3])
    if isinstance(v, tuple) and v[0] == 'arange':
        return np.arange(v[1], v[2], v[3])
    return np.asarray(v)


def make_prior(p):
    if p is None or p == 'default':
        return p
    if isinstance(p, str):
        return PRIORS[p]
    if p[0] == 'ones':
        return np.ones(p[1])
    if p[0] == 'array':
        return np.array(p[1], dtype=float)
    if p[0] == 'sympy_exp':                      # reference tests/test_onlinestudy.py:41
        import sympy.stats as stats
        return stats.Exponential('e', p[1])
    raise ValueError(p)


def make_tm(bl, spec):
    kind = spec[0]
    if kind == 'Static':
        return bl.tm.Static()
    if kind == 'GRW':
        return bl.tm.GaussianRandomWalk(spec[1], make_values(bl, spec[2]), target=spec[3], prior=make_prior(spec[4]))
    if kind == 'ChangePoint':
        return bl.tm.ChangePoint(spec[1], make_values(bl, spec[2]), prior=make_prior(spec[3]))
    if kind == 'Combined':
        return bl.tm.CombinedTransitionModel(*[make_tm(bl, s) for s in spec[1]])
    if kind == 'RS':
        return bl.tm.RegimeSwitch(spec[1], make_values(bl, spec[2]), prior=make_prior(spec[3]))
    if kind == 'Independent':
        return bl.tm.Independent()
    if kind == 'Deterministic':
        return bl.tm.Deterministic(FUNCS[spec[1]], target=spec[2])
    if kind == 'AlphaStable':
        return bl.tm.AlphaStableRandomWalk(spec[1], make_values(bl, spec[2]), spec[3], make_values(bl, spec[4]), target=spec[5])
    if kind == 'Bivariate':
        return bl.tm.BivariateRandomWalk(spec[1], make_values(bl, spec[2]), spec[3], make_values(bl, spec[4]),
                                         spec[5], make_values(bl, spec[6]))
    if kind == 'NE':
        return bl.tm.NotEqual(spec[1], make_values(bl, spec[2]), prior=make_prior(spec[3]))
    if kind == 'BreakPoint':
        return bl.tm.BreakPoint(spec[1], make_values(bl, spec[2]), prior=make_prior(spec[3]))
    if kind == 'Serial':
        return bl.tm.SerialTransitionModel(*[make_tm(bl, s) for s in spec[1]])
    raise ValueError(spec)


def make_om(bl, spec):
    name, params, prior = spec
    args = []
    for pname, values in params:
        args += [pname, make_values(bl, values)]
    if name.startswith('SciPy:'):            # ('SciPy:<scipy.stats name>', [(parameter, values) ...], prior): any number of parameters
        import scipy.stats
        rv = getattr(scipy.stats, name.split(':')[1])
        return bl.om.SciPy(rv, *args) if prior == 'default' else bl.om.SciPy(rv, *args, prior=make_prior(prior))
    cls = getattr(bl.om, name)
    if prior == 'default':
        return cls(*args)
    return cls(*args, prior=make_prior(prior))


def build(bl, case):
    """Un-fitted study of the given package for a case description."""
    c = CASES[case] if isinstance(case, str) else case
    S = getattr(bl, c['study'])(silent=True)
    S.loadData(make_data(c['data']), timestamps=c.get('timestamps'), silent=True)
    S.set(make_om(bl, c['om']), make_tm(bl, c['tm']), silent=True)
    return S


def fit_kwargs(case):
    c = CASES[case] if isinstance(case, str) else case
    kw = dict(c.get('fit', {}))
    if c['study'] != 'ChangepointStudy' or True:
        kw['silent'] = True
    return kw


# rows of the posterior sequence kept for 'sparse' fixtures
def sparse_rows(T):
    return sorted(set([0, 1, T // 3, T // 2, T - 2, T - 1]) & set(range(T)))


def sparse_stride(grid_shape, limit=70_000):
    """Per-axis subsampling stride of the stored posterior rows for large grids."""
    stride = [1] * len(grid_shape)
    while np.prod([-(-n // s) for n, s in zip(grid_shape, stride)]) > limit:
        stride = [s * 2 for s in stride]
    return stride
