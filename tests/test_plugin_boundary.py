"""The transition-model PLUG-IN boundary of the reference -- ``computeForwardPrior(posterior, t)`` / ``computeBackwardPrior(posterior, t)``
(bayesloop/transitionModels.py:49-63; called at core.py:411, :467) -- against goldens the reference itself produced
(tests/golden/gen_plugin_golden.py; the model definitions shared by both sides live in tests/plugin_models.py):

* user-defined transition models, written as the reference documents them, fitted through ``bl.Study`` / ``bl.HyperStudy``
  (host-side transition between device steps, ``Study._fitHostTransition``);
* the built-in models' own methods called directly on normalised and un-normalised distributions (one-step device programs).

Every case runs on CPU over the oracle-backed test double (host logic) and, marked ``gpu``, through the HIP engine.
"""
import os

import numpy as np
import pytest

import bayesloop_amd as bl
import plugin_models as pm
from oracle_engine import OracleEngine

HERE = os.path.dirname(os.path.abspath(__file__))
FITS = np.load(os.path.join(HERE, 'golden', 'plugin_fits.npz'))
DIRECT = np.load(os.path.join(HERE, 'golden', 'plugin_direct.npz'))
M = pm.make(bl.tm)
STUDY_NAMES = list(pm.studies(bl, M))
DIRECT_NAMES = list(pm.direct_calls(bl))


@pytest.fixture
def double():
    prev = bl.set_engine(OracleEngine())
    yield
    bl.set_engine(prev)


def check_fit(name):
    S, kw = pm.studies(bl, M)[name]
    S.fit(silent=True, **kw)
    want = float(FITS[name + '/logEvidence'])
    assert abs(S.logEvidence - want) <= 1e-9 * abs(want), (S.logEvidence, want)
    le = FITS[name + '/localEvidence']
    assert np.array_equal(np.isnan(S.localEvidence), np.isnan(le))
    np.testing.assert_allclose(np.asarray(S.localEvidence)[~np.isnan(le)], le[~np.isnan(le)], rtol=1e-9, atol=0)
    if name + '/posteriorSequence' in FITS.files:
        p = FITS[name + '/posteriorSequence']
        got = np.asarray(S.posteriorSequence)
        assert got.shape == p.shape
        assert np.all(np.abs(got - p) <= 1e-12 + 1e-9 * np.abs(p)), np.max(np.abs(got - p))
        np.testing.assert_allclose(np.asarray(S.posteriorMeanValues), FITS[name + '/posteriorMeanValues'], rtol=1e-9, atol=1e-12)
    if name + '/hyperParameterDistribution' in FITS.files:
        np.testing.assert_allclose(S.hyperParameterDistribution, FITS[name + '/hyperParameterDistribution'], rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(np.asarray(S.logEvidenceList, dtype=float), FITS[name + '/logEvidenceList'], rtol=1e-9)


def check_direct(name):
    S, model, calls = pm.direct_calls(bl)[name]
    S.setTransitionModel(model, silent=True)
    for k, (method, kind, t) in enumerate(calls):
        x = pm.distribution(kind, S.gridSize, seed=k)
        fn = model.computeForwardPrior if method == 'fwd' else model.computeBackwardPrior
        got = np.asarray(fn(x.copy(), t), dtype=float)
        want = DIRECT['%s/%d' % (name, k)]
        assert got.shape == want.shape
        # (AlphaStable / Deterministic: the reference's own FFT / spline round-off is ~1e-17 absolute, tests/tolerances.py FFT_FLOOR)
        assert np.all(np.abs(got - want) <= 1e-15 + 1e-9 * np.abs(want)), (name, k, method, kind, t, np.max(np.abs(got - want)))


@pytest.mark.parametrize('name', STUDY_NAMES)
def test_user_defined_transition_models_host_logic(double, name):
    check_fit(name)


@pytest.mark.parametrize('name', DIRECT_NAMES)
def test_built_in_models_answer_the_plug_in_calls_host_logic(double, name):
    check_direct(name)


@pytest.mark.gpu
@pytest.mark.parametrize('name', STUDY_NAMES)
def test_user_defined_transition_models_on_the_gpu(name):
    check_fit(name)


@pytest.mark.gpu
@pytest.mark.parametrize('name', DIRECT_NAMES)
def test_built_in_models_answer_the_plug_in_calls_on_the_gpu(name):
    check_direct(name)


def test_which_models_take_the_host_path():
    from bayesloop_amd.transitionModels import needs_host_transition as host
    assert not host(bl.tm.GaussianRandomWalk('s', 0.1, target='x'))
    assert not host(bl.tm.CombinedTransitionModel(bl.tm.GaussianRandomWalk('s', 0.1, target='x'), bl.tm.RegimeSwitch('p', -3)))
    assert not host(bl.tm.SerialTransitionModel(bl.tm.Static(), bl.tm.BreakPoint('b', 3), bl.tm.Static()))
    assert host(M['LeakyRandomWalk']('s', 0.1, 'l', 0.1, target='x'))
    assert host(M['CappedRandomWalk']('s', 0.1, target='x'))                   # a built-in subclass overriding the transition
    assert host(bl.tm.CombinedTransitionModel(bl.tm.Static(), M['CoolingWalk']('s', 0.1, 't', 3., target='x')))

    class Duck:                      # not even a subclass: the reference's boundary is duck-typed
        hyperParameterNames, hyperParameterValues, prior = [], [], None

        def computeForwardPrior(self, posterior, t):
            return posterior

        def computeBackwardPrior(self, posterior, t):
            return posterior
    assert host(Duck())


def test_the_host_path_is_announced_once(double, capfd):
    S, kw = pm.studies(bl, M)['cooling_poisson_full']
    bl.Study._host_transition_announced = bl.Study._host_transition_announced - {'CoolingWalk'}
    S.fit(silent=True, evidenceOnly=True)
    S.fit(silent=True, evidenceOnly=True)
    err = capfd.readouterr().err
    assert err.count('has no device program') == 1 and 'Cooling random walk' in err


def test_hyper_study_over_a_user_defined_model_builds_its_average_posterior(double):
    """Round 5 refused everything but evidenceOnly here; the reference averages posteriors for any model (core.py:1349-1366).  The
    goldens of `leaky_hyper_full` / `_fwdonly` / `combined_gauss_hyper_full` (parametrised above) pin the values; here: the accessors the
    reference's users call on the result work on the device-resident average."""
    S, kw = pm.studies(bl, M)['leaky_hyper_full']
    S.fit(silent=True)
    p = np.asarray(S.posteriorSequence)
    assert p.shape == (36, 90) and np.allclose(p.sum(axis=1), 1.0, rtol=1e-12)
    np.testing.assert_allclose(np.asarray(S.posteriorMeanValues)[0], (p * S.grid[0]).sum(axis=1), rtol=1e-10)
    x, d = S.getParameterDistribution(3, 'rate', density=False)
    np.testing.assert_allclose(d, p[3], rtol=1e-12, atol=1e-300)
    S.fit(silent=True, evidenceOnly=True)          # (an evidence-only fit afterwards leaves the average in place, as the reference does)
    assert np.array_equal(np.asarray(S.posteriorSequence), p)
