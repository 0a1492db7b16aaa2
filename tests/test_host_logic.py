"""Host-side logic of bayesloop_amd (grid/prior construction, transition-program compilation, hyper-grid plumbing,
ChangepointStudy masking, evidence algebra) on CPU: the product's study classes run with the oracle-backed TEST DOUBLE
engine (tests/oracle_engine.py) and must reproduce the reference's golden vectors."""
import numpy as np
import pytest

import bayesloop_amd as bl
import cases
import compare
import oracle_adapter as oa
from oracle_engine import OracleEngine


@pytest.fixture(autouse=True)
def oracle_engine():
    prev = bl.set_engine(OracleEngine())
    yield
    bl.set_engine(prev)


def result_of(S, case):
    c = cases.CASES[case]
    res = dict(logEvidence=S.logEvidence, localEvidence=S.localEvidence)
    kw = c.get('fit', {})
    if not kw.get('evidenceOnly', False) and np.isfinite(S.logEvidence):
        res['posteriorSequence'] = S.posteriorSequence
        res['posteriorMeanValues'] = S.posteriorMeanValues
    for key in ('logEvidenceList', 'hyperParameterDistribution', 'hyperGridValues', 'flatHyperPriorValues',
                'hyperGridConstant', 'mask'):
        if hasattr(S, key) and getattr(S, key) is not None and len(np.atleast_1d(getattr(S, key))) > 0:
            res[key] = np.asarray(getattr(S, key))
    return res


FAST = [k for k, c in cases.CASES.items() if not c.get('slow')]


@pytest.mark.parametrize('case', FAST)
def test_study_classes_reproduce_reference(case, capsys):
    S = cases.build(bl, case)
    with np.errstate(all='ignore'):
        S.fit(**{k: v for k, v in cases.fit_kwargs(case).items()})
    gold = oa.load_golden(case)
    np.testing.assert_array_equal(S.marginalGrid[0], gold['marginal0'])
    np.testing.assert_allclose(S.latticeConstant, gold['latticeConstant'], rtol=0, atol=0)
    compare.check(result_of(S, case), gold, compare.ORACLE_TOL, case_tol=cases.CASES[case].get('tol'))


def test_reference_test_expectations_through_accessors():
    """tests/test_hyperstudy.py:32-59 of the reference, verbatim expectations, via the accessor methods."""
    S = cases.build(bl, 'kat_hyper_1hp')
    S.fit(silent=True)
    np.testing.assert_allclose(S.getParameterDistributions('mean', density=False)[1][:, 5],
                               [0.017242, 0.014581, 0.012691, 0.011705, 0.011586], rtol=1e-04)
    np.testing.assert_allclose(S.getParameterMeanValues('mean'), [2.92089, 2.952597, 3., 3.047403, 3.07911], rtol=1e-05)
    np.testing.assert_almost_equal(S.logEvidence, -16.0629517262, decimal=5)
    x, p = S.getHyperParameterDistribution('sigma')
    np.testing.assert_allclose(np.array([x, p]), [[0., 0.2], [0.43828499, 0.56171501]], rtol=1e-05)


def test_study_accessors_reference_test_study():
    """tests/test_study.py:32-52 of the reference (default estimated 1000-point grid)."""
    S = cases.build(bl, 'kat_study_1hp')
    S.fit(silent=True)
    np.testing.assert_allclose(S.getParameterDistributions('rate', density=False)[1][:, 250],
                               [0.000417, 0.000386, 0.000356, 0.000336, 0.000332], rtol=1e-02)
    np.testing.assert_allclose(S.getParameterMeanValues('rate'),
                               [3.073534, 3.08179, 3.093091, 3.104016, 3.111173], rtol=1e-02)


def test_configuration_errors():
    S = bl.Study(silent=True)
    with pytest.raises(bl.ConfigurationError):
        S.fit()
    S.loadData(np.array([1, 2, 3]), silent=True)
    with pytest.raises(bl.ConfigurationError):
        S.fit()
    S.setOM(bl.om.Poisson('rate', bl.oint(0, 6, 50)), silent=True)
    with pytest.raises(bl.ConfigurationError):
        S.fit()
    with pytest.raises(bl.ConfigurationError):
        S.set(bl.om.Poisson('a', bl.oint(0, 1, 5)), bl.om.Poisson('b', bl.oint(0, 1, 5)), silent=True)
    with pytest.raises(bl.ConfigurationError):
        bl.tm.GaussianRandomWalk('sigma', 0.1)          # no target
    with pytest.raises(bl.ConfigurationError):
        bl.tm.Deterministic(lambda t, slope: slope * t, target='rate')      # hyper-parameters need default values
    with pytest.raises(bl.ConfigurationError):
        bl.tm.Deterministic(lambda t, slope=1: slope * t)                   # no target
    S.setTM(bl.tm.CombinedTransitionModel(bl.tm.GaussianRandomWalk('s', 0.1, target='rate'),
                                          bl.tm.GaussianRandomWalk('s', 0.2, target='rate')), silent=True)
    with pytest.raises(bl.ConfigurationError):
        S.fit(silent=True)                               # duplicate hyper-parameter names


def test_table_likelihood_path_matches_native_models():
    """A user-defined observation model (plug-in pdf) goes through the likelihood-table path."""
    class MyPoisson(bl.om.ObservationModel):
        def __init__(self, name, value):
            self.name = 'my poisson'
            self.segmentLength = 1
            self.multiplyLikelihoods = True
            self.parameterNames = [name]
            self.parameterValues = [value]
            self.prior = lambda x: np.sqrt(1. / x)

        def pdf(self, grid, dataSegment):
            import math
            return (grid[0] ** dataSegment[0]) * np.exp(-grid[0]) / math.factorial(int(dataSegment[0]))

    S = bl.Study(silent=True)
    S.loadData(cases.D15, silent=True)
    S.set(MyPoisson('rate', bl.oint(0, 6, 100)), bl.tm.GaussianRandomWalk('sigma', 0.2, target='rate'), silent=True)
    S.fit(silent=True)
    np.testing.assert_allclose(S.logEvidence, -10.323144246611964, rtol=1e-13)


def test_pickle_roundtrip_materialises_posterior():
    import pickle
    S = cases.build(bl, 'kat_grw')
    S.fit(silent=True)
    S2 = pickle.loads(pickle.dumps(S))
    np.testing.assert_array_equal(S2.posteriorSequence, S.posteriorSequence)
    assert S2.logEvidence == S.logEvidence


def test_device_side_reductions_equal_host_reductions():
    """getParameterDistributions / getParameterDistribution use the engine's reductions while the posterior is still on
    the device; they must equal reductions of the materialised array."""
    for case in ('c3_small', 'c4_2hp'):
        S = cases.build(bl, case)
        S.fit(silent=True)
        assert S._posterior_pending is not None
        x, m0 = S.getParameterDistributions('mean', density=False)
        x1, m1 = S.getParameterDistributions('std')
        xa, avg = S.getParameterDistribution('avg', 'mean')
        xt, pt = S.getParameterDistribution(S.formattedTimestamps[3], 'std', density=False)
        assert S._posterior_pending is not None           # nothing was materialised
        post = S.posteriorSequence
        np.testing.assert_allclose(m0, post.sum(axis=2), rtol=1e-13)
        np.testing.assert_allclose(m1, post.sum(axis=1) / S.latticeConstant[1], rtol=1e-13)
        np.testing.assert_allclose(avg, post.mean(axis=0).sum(axis=1) / S.latticeConstant[0], rtol=1e-13)
        np.testing.assert_allclose(pt, post[3].sum(axis=0), rtol=1e-13)


def test_simulate_matches_reference_golden_host_side():
    """Study.simulate (core.py:566-597) over the test double (row / time-average reads of the posterior handle)."""
    gold = oa.load_golden('simulate')
    for case in ('c1_coal', 'kat_gaussian'):
        S = cases.build(bl, case)
        with np.errstate(all='ignore'):
            S.fit(**cases.fit_kwargs(case))
        x = gold[case + '_x']
        np.testing.assert_allclose(S.simulate(x), gold[case + '_avg'], rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(S.simulate(x, t=float(gold[case + '_t'])), gold[case + '_at'], rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(S.simulate(x, density=True), gold[case + '_avg_density'], rtol=1e-9, atol=1e-300)
    S = bl.Study(silent=True)
    S.loadData(np.array([1, 0, 1, 0, 0]), silent=True)
    S.set(bl.om.AR1('rho', bl.oint(-1, 1, 20), 'sigma', bl.oint(0, 1, 20)), bl.tm.Static(), silent=True)
    with pytest.raises(NotImplementedError):
        S.simulate([0.])


def test_scipy_sympy_numpy_observation_models_reference_kats():
    """bl.om.SymPy / SciPy / NumPy (reference tests/test_observationmodels.py:11-120): plug-in likelihoods through the
    table path; the reference's own log-evidence values, same tolerance (decimal=5)."""
    scipy_stats = pytest.importorskip('scipy.stats')
    sympy_stats = pytest.importorskip('sympy.stats')
    from sympy import Symbol

    def logE(L, data=np.array([1, 2, 3, 4, 5])):
        S = bl.Study(silent=True)
        S.loadData(data, silent=True)
        S.setOM(L, silent=True)
        S.setTM(bl.tm.Static(), silent=True)
        S.fit(silent=True)
        return S.logEvidence

    rate = Symbol('rate', positive=True)
    np.testing.assert_almost_equal(logE(bl.om.SymPy(sympy_stats.Poisson('poisson', rate), 'rate', bl.oint(0, 7, 100))),
                                   -10.238278174965238, decimal=5)
    mu, std = Symbol('mu'), Symbol('std', positive=True)
    np.testing.assert_almost_equal(logE(bl.om.SymPy(sympy_stats.Normal('norm', mu, std), 'mu', bl.cint(0, 7, 200), 'std',
                                                    bl.oint(0, 1, 200), prior=lambda x, y: 1.)), -13.663836264357226, decimal=5)
    np.testing.assert_almost_equal(logE(bl.om.SciPy(scipy_stats.poisson, 'mu', bl.oint(0, 7, 100), fixedParameters={'loc': 0})),
                                   -10.238278174965238, decimal=5)
    np.testing.assert_almost_equal(logE(bl.om.SciPy(scipy_stats.norm, 'loc', bl.cint(0, 7, 200), 'scale', bl.oint(0, 1, 200))),
                                   -13.663836264357225, decimal=5)

    def likelihood(data, mu):
        x, s = data
        return np.exp((x - mu) ** 2. / (2 * s ** 2.)) / np.sqrt(2 * np.pi * s ** 2.)
    np.testing.assert_almost_equal(logE(bl.om.NumPy(likelihood, 'mu', bl.oint(0, 7, 100)),
                                        np.array([[1, 0.5], [2, 0.5], [3, 0.5], [4, 1.], [5, 1.]])), 148.92056578058387, decimal=5)
    # the symbolic Jeffreys prior the reference intends (1/sqrt(rate) for a Poisson rate)
    from bayesloop_amd.observationModels import jeffreys_prior_of
    expr, f = jeffreys_prior_of(sympy_stats.Poisson('poisson', rate))
    assert str(expr) == '1/sqrt(rate)' and abs(f(4.0) - 0.5) < 1e-15
    with pytest.raises(bl.ConfigurationError):
        bl.om.SciPy(np.random, 'mu', bl.oint(0, 7, 100))
    with pytest.raises(bl.ConfigurationError):
        bl.om.SciPy(scipy_stats.norm, 'wrong', bl.oint(0, 7, 100))


def test_save_load_roundtrip(tmp_path):
    """bl.save / bl.load (reference fileIO.py:10-37, tests/test_fileio.py): a fitted study with a lambda prior survives."""
    S = bl.HyperStudy(silent=True)
    S.loadData(np.array([1, 2, 3, 4, 5]), silent=True)
    S.setOM(bl.om.Gaussian('mean', bl.cint(0, 6, 20), 'sigma', bl.oint(0, 2, 20), prior=lambda m, s: 1 / s ** 3), silent=True)
    S.setTM(bl.tm.GaussianRandomWalk('s', [0.1, 0.3], target='mean'), silent=True)
    S.fit(silent=True)
    path = str(tmp_path / 'study.bl')
    bl.save(path, S)
    R = bl.load(path)
    assert R.logEvidence == S.logEvidence
    np.testing.assert_array_equal(R.posteriorSequence, S.posteriorSequence)
    np.testing.assert_array_equal(R.hyperParameterDistribution, S.hyperParameterDistribution)
    assert R.observationModel.prior(1.0, 2.0) == 1 / 8.
    R.fit(silent=True)                     # the loaded study can be fitted again
    assert abs(R.logEvidence - S.logEvidence) < 1e-12


def test_jeffreys_helpers():
    """bl.getJeffreysPrior / bl.computeJeffreysPriorAR1 (reference bayesloop/jeffreys.py): the closed form of Uhlig's prior."""
    sympy_stats = pytest.importorskip('sympy.stats')
    from sympy import Symbol
    expr, f = bl.getJeffreysPrior(sympy_stats.Poisson('p', Symbol('rate', positive=True)))
    assert str(expr) == '1/sqrt(rate)' and abs(f(4.0) - 0.5) < 1e-15
    data = np.array([0.3, -0.2, 0.5, 0.1, -0.4])
    for om, scaled in ((bl.om.AR1, False), (bl.om.ScaledAR1, True)):
        S = bl.Study(silent=True)
        S.loadData(data, silent=True)
        S.setOM(om('rho', bl.oint(-1, 1, 30), 'sigma', bl.oint(0, 2, 25)), silent=True)
        p = bl.computeJeffreysPriorAR1(S, t=2)
        r, s = S.grid
        if scaled:
            s = s * np.sqrt(1 - r ** 2)
        want = np.exp(-data[1] ** 2 * (1 - r ** 2) / (2 * s ** 2)) / s ** 2 * np.sqrt(4 * r ** 2 / (1 - r ** 2) + 2 * (len(data) + 1))
        np.testing.assert_allclose(p, want / want.sum(), rtol=1e-13)
        assert abs(p.sum() - 1) < 1e-12
    S = bl.Study(silent=True)
    S.loadData(data, silent=True)
    S.setOM(bl.om.Poisson('rate', bl.oint(0, 6, 20)), silent=True)
    with pytest.raises(bl.ConfigurationError):
        bl.computeJeffreysPriorAR1(S)


def test_accessors_accept_keyword_arguments():
    """The reference's accessors take their arguments by keyword as well (core.py:864-1002, 1535-1694); the plot decorator must
    not swallow them (every call below used to raise TypeError: missing positional argument)."""
    S = cases.build(bl, 'kat_hyper_1hp')
    S.fit(silent=True)
    x0, p0 = S.getParameterDistribution(S.formattedTimestamps[2], 'mean')
    x1, p1 = S.getParameterDistribution(t=S.formattedTimestamps[2], name='mean')
    x2, p2 = S.getParameterDistribution(S.formattedTimestamps[2], name='mean', density=False)
    np.testing.assert_array_equal(p0, p1)
    np.testing.assert_allclose(p2, p0 * S.latticeConstant[0], rtol=1e-13)
    np.testing.assert_array_equal(S.getParameterDistributions(name='mean')[1], S.getParameterDistributions('mean')[1])
    np.testing.assert_array_equal(S.getPDs(name='mean', density=False)[1], S.getParameterDistributions('mean', density=False)[1])
    np.testing.assert_array_equal(S.getHyperParameterDistribution(name='sigma')[1], S.getHyperParameterDistribution('sigma')[1])
    np.testing.assert_array_equal(S.getHPD(name='sigma')[1], S.getHPD('sigma')[1])
    S2 = cases.build(bl, 'c4_2hp')
    S2.fit(silent=True)
    names = list(S2.flatHyperParameterNames[:2])
    a = S2.getJointHyperParameterDistribution(names)
    b = S2.getJointHyperParameterDistribution(names=names)
    for u, v in zip(a, b):
        np.testing.assert_array_equal(u, v)
    C = cases.build(bl, 'kat_changepointstudy')
    C.fit(silent=True)
    cp = [n for n in C.flatHyperParameterNames]
    if len(cp) >= 2:
        np.testing.assert_array_equal(C.getDurationDistribution(names=cp[:2])[1], C.getDurationDistribution(cp[:2])[1])


def test_evidence_only_refit_keeps_the_previous_posterior():
    """Study.fit(evidenceOnly=True) after a full fit: the reference keeps the previous posteriorSequence (core.py:355-356)."""
    S = cases.build(bl, 'c1_coal')
    S.fit(silent=True)
    logE = S.logEvidence
    S.fit(evidenceOnly=True, silent=True)
    assert S.logEvidence == logE
    gold = oa.load_golden('c1_coal')
    np.testing.assert_allclose(S.posteriorSequence, gold['posteriorSequence'], rtol=1e-9, atol=1e-15)
    x, p = S.getParameterDistribution(1900, 'rate')
    assert np.isfinite(p).all()


def test_online_study_survives_pickling():
    """A pickled / copied OnlineStudy carries its per-chain filter states along and continues stepping independently of the
    original (reference fileIO.py:10-37)."""
    import pickle
    data = np.array([1.0, 2.0, 3.0, 2.0, 4.0, 3.0])

    def make():
        O = bl.OnlineStudy(storeHistory=True, silent=True)
        O.set(bl.om.Poisson('rate', bl.oint(0, 6, 50)), silent=True)
        O.add('static', bl.tm.Static())
        O.add('grw', bl.tm.GaussianRandomWalk('sigma', [0.1, 0.3], target='rate'))
        return O
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        A = make()
        for x in data[:3]:
            A.step(x)
        B = pickle.loads(pickle.dumps(A))
        for x in data[3:]:
            B.step(x)             # the copy goes on ...
        for x in data[3:]:
            A.step(x)             # ... and so does the original, from its own state
        R = make()
        for x in data:
            R.step(x)
    for O in (A, B):
        np.testing.assert_allclose(O.logEvidence, R.logEvidence, rtol=1e-12)
        np.testing.assert_allclose(O.marginalizedPosterior, R.marginalizedPosterior, rtol=1e-12)
        np.testing.assert_allclose(O.transitionModelDistribution, R.transitionModelDistribution, rtol=1e-12)
    del B
    with contextlib.redirect_stdout(io.StringIO()):
        A.step(2.0)               # releasing the copy's slots must not touch the original's
    assert np.isfinite(A.logEvidence)


# ---- the scale algebra of the resident kernels (include/blhip.h: blhip_host_unlag) -- pure host code of libblhip.so ---------------------

def _simulate_sums(norms, lag, scheme, kinds=None):
    """What a resident kernel reports: S_k = (S_(k-1) or 1 at a restart) * s_k * n_k with the kernel's rule for s_k."""
    T = len(norms)
    S, s = np.zeros(T), np.ones(T)
    for k in range(T):
        if k >= lag:
            s[k] = 1.0 / S[k - lag] if scheme == 0 else (S[k - lag - 1] if k - lag - 1 >= 0 else 1.0) * s[k - lag] / S[k - lag]
        fresh = k == 0 or (kinds is not None and kinds[k] != 0)
        S[k] = (1.0 if fresh else S[k - 1]) * s[k] * norms[k]
    return S, s


@pytest.mark.parametrize('scheme,lag', [(0, 1), (0, 2), (1, 2), (1, 3), (1, 4)])
def test_host_recovers_the_normalisers_from_lagged_sums(scheme, lag):
    import ctypes
    from bayesloop_amd import _abi
    lib = _abi.load()
    rng = np.random.default_rng(100 * scheme + lag)
    T = 400
    norms = np.exp(rng.normal(-3.0, 1.5, T))                    # what core.py:385 would see
    S, s = _simulate_sums(norms, lag, scheme)
    sums, scales = S.copy(), np.zeros(T)
    assert lib.blhip_host_unlag(scheme, _abi.dptr(sums), T, lag, None, _abi.dptr(scales)) == 0
    np.testing.assert_allclose(sums, norms, rtol=1e-12)
    np.testing.assert_allclose(scales, s, rtol=1e-12)


def test_scale_from_the_lagged_normaliser_stays_bounded_where_the_lagged_sum_diverges():
    """Dividing by the SUM of step k - lag is a feedback loop x_k = x_(k-1) - x_(k-lag) + nu in the log domain: bounded (period 6) for
    lag 2, exponentially unstable for lag 3 (|roots of r^3 - r^2 + 1| = 1.15).  Dividing by the NORMALISER of step k - lag leaves
    the product of the last `lag` normalisers in the state, whatever the lag (blhip_chainres.hpp)."""
    rng = np.random.default_rng(7)
    T = 400
    norms = np.exp(rng.normal(-2.0, 0.3, T))
    with np.errstate(over='ignore', invalid='ignore', divide='ignore'):
        S_sum3, _ = _simulate_sums(norms, 3, 0)
    assert not np.all(np.isfinite(S_sum3)) or np.nanmax(np.abs(np.log(S_sum3[np.isfinite(S_sum3) & (S_sum3 > 0)]))) > 300.0
    S_sum2, _ = _simulate_sums(norms, 2, 0)
    assert np.max(np.abs(np.log(S_sum2))) < 40.0
    for lag in (2, 3, 4):
        S, _ = _simulate_sums(norms, lag, 1)
        want = np.array([np.prod(norms[max(0, k - lag + 1):k + 1]) for k in range(T)])
        np.testing.assert_allclose(S, want, rtol=1e-10)


def test_host_unlag_with_restarts_and_out_of_range_sums():
    import ctypes
    from bayesloop_amd import _abi
    lib = _abi.load()
    rng = np.random.default_rng(11)
    T, lag = 60, 4
    norms = np.exp(rng.normal(-3.0, 1.0, T))
    kinds = np.zeros(T, dtype=np.uint8)
    kinds[[17, 18, 40]] = 2                                     # blk::SRC_RESET: change points (transitionModels.py:300-312)
    S, s = _simulate_sums(norms, lag, 1, kinds)
    sums = S.copy()
    assert lib.blhip_host_unlag(1, _abi.dptr(sums), T, lag, kinds.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), None) == 0
    np.testing.assert_allclose(sums, norms, rtol=1e-12)
    bad = S.copy()
    bad[30] = 1e-200                                            # a run of extreme outliers: the fit falls back to the launch-per-step kernels
    assert lib.blhip_host_unlag(1, _abi.dptr(bad), T, lag, None, None) == 1


# ---- observation models with three parameters: the Python surface (grid, program, accessors) through the test double -------------------

def test_three_parameter_models_host_logic():
    pytest.importorskip('scipy.stats')
    om = ('SciPy:t', [('df', ('cint', 2.0, 9.0, 5)), ('loc', ('cint', -3.0, 3.0, 12)), ('scale', ('oint', 0.2, 2.5, 10))], 'default')
    c = dict(study='Study', data=('series', 82, 8), om=om,
             tm=('Combined', [('GRW', 's_loc', 0.6, 'loc', None), ('GRW', 's_scale', 0.25, 'scale', None)]))
    S = cases.build(bl, c)
    S.fit(silent=True)
    with np.errstate(all='ignore'):
        want = oa.run(c)
    assert list(S.gridSize) == [5, 12, 10] and len(S.grid) == 3 and S.grid[0].shape == (5, 12, 10)
    assert abs(S.logEvidence - want['logEvidence']) <= 1e-12 * abs(want['logEvidence'])
    np.testing.assert_allclose(np.asarray(S.posteriorSequence), want['posteriorSequence'], rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(S.posteriorMeanValues, want['posteriorMeanValues'], rtol=1e-12)
    for k, name in enumerate(S.observationModel.parameterNames):           # marginals of every parameter (core.py:915, 979-980)
        axes = tuple(a + 1 for a in range(3) if a != k)
        np.testing.assert_allclose(S.getParameterDistributions(name, density=False)[1], want['posteriorSequence'].sum(axis=axes), rtol=1e-12)
    x, p = S.getParameterDistribution(3, 'loc', density=False)
    np.testing.assert_allclose(p, want['posteriorSequence'][3].sum(axis=(0, 2)), rtol=1e-12)

    h = dict(study='HyperStudy', data=('series', 89, 7), om=om, tm=('GRW', 's_loc', ('cint', 0, 1.2, 5), 'loc', None))
    H = cases.build(bl, h)
    H.fit(silent=True)
    with np.errstate(all='ignore'):
        wh = oa.run(h)
    assert abs(H.logEvidence - wh['logEvidence']) <= 1e-12 * abs(wh['logEvidence'])
    np.testing.assert_allclose(H.hyperParameterDistribution, wh['hyperParameterDistribution'], rtol=1e-10)
    np.testing.assert_allclose(H.posteriorMeanValues, wh['posteriorMeanValues'], rtol=1e-10)

    # transition models the N-D path does not run are refused BEFORE any device work (as every configuration problem)
    bad = dict(study='Study', data=('series', 82, 6), om=om, tm=('RS', 'log10pMin', -4, None))
    B = cases.build(bl, bad)
    with pytest.raises(bl.exceptions.ConfigurationError):
        B.fit(silent=True)


def test_deterministic_shifts_of_many_parameter_sets_equal_the_per_set_evaluation():
    """Deterministic.shifts_many (one broadcast call of the user's function over parameter sets x time stamps: what a hyper-study over
    break-points evaluates per unique (parameters, offset) pair) returns exactly what shifts() returns per set -- for a function that
    broadcasts, and, through the fall-back, for one written for scalars only (reference transitionModels.py:573-577, :592-596, :770-776)."""
    import numpy as np
    import bayesloop_amd as bl
    rng = np.random.default_rng(5)
    ts = np.arange(1851., 1892.)

    def linear(t, slope=0.0):
        return slope * t

    def scalar_only(t, slope=0.0, level=0.0):
        if np.ndim(t) != 0 or np.ndim(slope) != 0:          # (a function that refuses arrays)
            raise TypeError('scalars only')
        return level + slope * max(t, 0.0)

    for fn, names in ((linear, ['slope']), (scalar_only, ['slope', 'level'])):
        m = bl.tm.Deterministic(fn, target='rate')
        rows = rng.uniform(-2, 2, size=(7, len(names)))
        offs = rng.integers(1855, 1880, 7).astype(float)
        for t_offsets in (None, offs):
            many = m.shifts_many(names, rows, ts, -1.0, t_offsets=t_offsets)
            assert many.shape == (7, 2 * len(ts))
            for u in range(7):
                one = m.shifts(dict(zip(names, rows[u])), ts, -1.0, t_offset=None if t_offsets is None else t_offsets[u])
                assert np.array_equal(many[u], one, equal_nan=True)


def test_deterministic_function_that_takes_arrays_but_is_not_elementwise():
    """Advisor finding (round 4): a user function that accepts an array of time stamps need not be elementwise (here it subtracts the
    FIRST stamp it is handed).  The reference calls it per scalar (transitionModels.py:573-577): the vectorised evaluation is cross-checked
    against scalar calls and falls back to them."""
    def f(t, slope=0.0):
        t = np.asarray(t, dtype=float)
        return slope * (t - t.ravel()[0]) + slope * t          # array input: depends on the batch; scalar input: slope * t
    m = bl.tm.Deterministic(f, target='rate')
    ts = np.arange(10.)
    got = m.shifts({'slope': 0.5}, ts, -1.0, t_offset=0.0)
    ref = bl.tm.Deterministic(lambda t, slope=0.0: slope * t, target='rate').shifts({'slope': 0.5}, ts, -1.0, t_offset=0.0)
    assert np.array_equal(got, ref)


def test_online_study_refuses_user_defined_transition_models():
    """Advisor finding (round 4): OnlineStudy has no host-transition path; a user-defined model (or a built-in subclass that overrides
    computeForwardPrior) must raise ConfigurationError, not TypeError / silently run the base class's device program."""
    from bayesloop_amd.exceptions import ConfigurationError

    class Mine(bl.tm.TransitionModel):
        hyperParameterNames = []
        hyperParameterValues = []

        def computeForwardPrior(self, posterior, t):
            return posterior

    class Override(bl.tm.GaussianRandomWalk):
        def computeForwardPrior(self, posterior, t):
            return posterior
    import contextlib, io
    for tm in (Mine(), Override('sigma', 0.1, target='rate'), bl.tm.CombinedTransitionModel(bl.tm.Static(), Override('sigma', 0.1, target='rate'))):
        with contextlib.redirect_stdout(io.StringIO()):
            O = bl.OnlineStudy(storeHistory=True, silent=True)
            O.set(bl.om.Poisson('rate', bl.oint(0, 6, 20)), silent=True)
            O.add('m', tm)
            with pytest.raises(ConfigurationError, match='user-defined'):
                O.step(1.0)
