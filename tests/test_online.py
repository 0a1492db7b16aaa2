"""OnlineStudy (reference core.py:1963-2226, tests/test_onlinestudy.py).

CPU: the oracle's restatement of OnlineStudy.step against goldens generated from the reference, and the product's host
logic (model compilation, evidence bookkeeping, accessors) over the oracle test double.  GPU (-m gpu): the same goldens
through the HIP kernels with device-resident carried states."""
import contextlib
import io
import os

import numpy as np
import pytest

import cases
import oracle_adapter as oa
import bayesloop_amd as bl

RTOL, ATOL = 1e-9, 1e-13


def check_online(got, gold, n_models):
    assert abs(got['logEvidence'] - float(gold['logEvidence'])) <= RTOL * abs(float(gold['logEvidence']))
    for key in ('posteriorSequence', 'posteriorMeanValues', 'transitionModelSequence', 'localTransitionModelSequence'):
        np.testing.assert_allclose(np.asarray(got[key]), gold[key], rtol=RTOL, atol=ATOL, err_msg=key)
    for i in range(n_models):
        np.testing.assert_allclose(np.asarray([h[i] for h in got['hyperParameterSequence']]), gold['hyperParameterSequence%d' % i],
                                   rtol=RTOL, atol=ATOL)
        if 'parameterPosterior' in got:
            np.testing.assert_allclose(np.asarray(got['parameterPosterior'][i]), gold['parameterPosterior%d' % i], rtol=RTOL, atol=ATOL)
        if 'logEvidenceList' in got:
            np.testing.assert_allclose(np.asarray(got['logEvidenceList'][i]), gold['logEvidenceList%d' % i], rtol=RTOL)


@pytest.mark.parametrize('case', list(cases.ONLINE_CASES))
def test_oracle_online_step_matches_reference_golden(case):
    gold = oa.load_golden(case)
    got = oa.run_online(case)
    kat = cases.ONLINE_CASES[case].get('kat')
    if kat is not None:
        np.testing.assert_almost_equal(got['logEvidence'], kat, decimal=5)      # reference tests/test_onlinestudy.py:30, :82
    check_online(got, gold, int(gold['n_models']))


def run_product(case):
    c = cases.ONLINE_CASES[case]
    S = cases.build_online(bl, case)
    with contextlib.redirect_stdout(io.StringIO()):
        for d in cases.online_data(c):
            S.step(d)
    got = dict(logEvidence=S.logEvidence, posteriorSequence=S.posteriorSequence, posteriorMeanValues=S.posteriorMeanValues,
               transitionModelSequence=S.transitionModelSequence, localTransitionModelSequence=S.localTransitionModelSequence,
               hyperParameterSequence=S.hyperParameterSequence, parameterPosterior=S.parameterPosterior,
               logEvidenceList=S.logEvidenceList)
    return S, got


def reference_accessor_checks(S, case):
    """The assertions of the reference's own tests (tests/test_onlinestudy.py), verbatim values."""
    if case == 'online_kat_static':
        np.testing.assert_allclose(S.getParameterDistributions('mean', density=False)[1][:, 5],
                                   [0.0053811, 0.38690331, 0.16329865, 0.04887604, 0.01334921], rtol=1e-05)
        np.testing.assert_allclose(S.getParameterMeanValues('mean'), [0.96310103, 1.5065597, 2.00218465, 2.500366, 3.], rtol=1e-05)
        np.testing.assert_almost_equal(S.logEvidence, -16.1946904707, decimal=5)
    if case == 'online_kat_2tm':
        np.testing.assert_allclose(S.getCurrentTransitionModelDistribution(local=False)[1], [0.49402616, 0.50597384], rtol=1e-05)
        np.testing.assert_allclose(S.getCurrentTransitionModelDistribution(local=True)[1], [0.81739495, 0.18260505], rtol=1e-05)
        np.testing.assert_allclose(S.getCurrentHyperParameterDistribution('s2')[1], [0.19047162, 0.80952838], rtol=1e-05)
        np.testing.assert_allclose(S.getParameterDistributions('mean', density=False)[1][:, 5],
                                   [0.05825921, 0.20129444, 0.07273516, 0.02125759, 0.0039255], rtol=1e-05)
        np.testing.assert_allclose(S.getParameterMeanValues('mean'),
                                   [1.0771838, 1.71494272, 2.45992376, 3.34160617, 4.39337253], rtol=1e-05)
        np.testing.assert_almost_equal(S.logEvidence, -9.46900822686, decimal=5)
        # accessors beyond the reference's tests: consistency with the stored history
        x, p = S.getHyperParameterDistributions('s1')
        assert p.shape == (5, 2) and np.allclose(p.sum(axis=1), 1.0)
        np.testing.assert_allclose(S.getTransitionModelProbabilities('T2'), np.array(S.transitionModelSequence)[:, 1])
        np.testing.assert_allclose(S.getCurrentParameterMeanValue('mean'), S.getParameterMeanValues('mean')[-1])
        np.testing.assert_allclose(S.getHyperParameterMeanValues('s2')[-1], S.getHyperParameterMeanValue(4, 's2'))
        np.testing.assert_allclose(S.getParameterDistribution(4, 'mean')[1], S.getCurrentParameterDistribution('mean')[1])
        tmp = S.transitionModelPosterior
        np.testing.assert_allclose(np.tensordot(S.transitionModelDistribution, tmp, axes=(0, 0)), S.marginalizedPosterior, rtol=1e-12)


@pytest.mark.parametrize('case', list(cases.ONLINE_CASES))
def test_online_study_host_logic_over_oracle_engine(case):
    from oracle_engine import OracleEngine
    prev = bl.engine._engine if hasattr(bl.engine, '_engine') else None
    bl.set_engine(OracleEngine())
    try:
        S, got = run_product(case)
        gold = oa.load_golden(case)
        check_online(got, gold, int(gold['n_models']))
        reference_accessor_checks(S, case)
    finally:
        bl.set_engine(prev)


def test_online_study_errors():
    from oracle_engine import OracleEngine
    prev = bl.engine._engine if hasattr(bl.engine, '_engine') else None
    bl.set_engine(OracleEngine())
    try:
        S = bl.OnlineStudy(silent=True)
        S.setOM(bl.om.Poisson('rate', bl.oint(0, 6, 50)), silent=True)
        with pytest.raises(bl.exceptions.ConfigurationError):
            S.step(1)                                                     # no transition model (core.py:2070-2071)
        with pytest.raises(NotImplementedError):
            S.fit()
        with contextlib.redirect_stdout(io.StringIO()):
            S.addTransitionModel('a', bl.tm.GaussianRandomWalk('sigma', [0.1, 0.2], target='rate'))
            S.addTransitionModel('b', bl.tm.GaussianRandomWalk('sigma', 0.3, target='rate'))
            with pytest.raises(bl.exceptions.ConfigurationError):
                S.step(1)                                                 # duplicate hyper-parameter names (:2083-2085)
        with pytest.raises(bl.exceptions.ConfigurationError):
            S.setTransitionModelPrior([1.0])
        S2 = bl.OnlineStudy(silent=True)
        S2.setOM(bl.om.Poisson('rate', bl.oint(0, 6, 50)), silent=True)
        with contextlib.redirect_stdout(io.StringIO()):
            S2.addTransitionModel('a', bl.tm.Static())
            S2.step(3)
        with pytest.raises(bl.exceptions.PostProcessingError):
            S2.getParameterMeanValues('rate')                             # needs storeHistory=True (:2546-2549)
        assert S2.getCurrentParameterMeanValue('rate') > 0
    finally:
        bl.set_engine(prev)


@pytest.mark.gpu
@pytest.mark.parametrize('case', list(cases.ONLINE_CASES))
def test_online_study_on_device_matches_reference_golden(case):
    S, got = run_product(case)
    gold = oa.load_golden(case)
    check_online(got, gold, int(gold['n_models']))
    reference_accessor_checks(S, case)


@pytest.mark.gpu
def test_online_study_large_grid_matches_oracle_and_offline_fit():
    """512 x 512 grid, wide walk (matrix-pipe kernels through the resume path): the online filter of a single chain equals
    Study.fit(forwardOnly=True) step by step (same recursion), and the evidence equals the offline evidence."""
    x = cases.series(51, 10)
    om = lambda: bl.om.Gaussian('mean', bl.cint(-8, 8, 512), 'std', bl.oint(0, 4, 512))
    S = bl.OnlineStudy(storeHistory=True, silent=True)
    S.setOM(om(), silent=True)
    with contextlib.redirect_stdout(io.StringIO()):
        S.addTransitionModel('walk', bl.tm.GaussianRandomWalk('s', [0.1, 0.6], target='mean'))
        for d in x:
            S.step(d)
    for j, s in enumerate([0.1, 0.6]):
        F = bl.Study(silent=True)
        F.loadData(x, silent=True)
        F.set(om(), bl.tm.GaussianRandomWalk('s', s, target='mean'), silent=True)
        F.fit(forwardOnly=True, silent=True)
        assert abs(S.logEvidenceList[0][j] - F.logEvidence) <= 1e-10 * abs(F.logEvidence)
        np.testing.assert_allclose(S.parameterPosterior[0][j], F.posteriorSequence[-1], rtol=1e-9, atol=1e-14)


def run_product_case(c):
    S = cases.build_online(bl, c)
    with contextlib.redirect_stdout(io.StringIO()):
        for d in cases.online_data(c):
            S.step(d)
    return S


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(int(os.environ.get('BLHIP_FUZZ_SEEDS', 40))))
def test_online_study_seeded_random_models_match_oracle(seed):
    """Random sets of competing transition models with random hyper-grids (tests/random_cases.py) on the device vs the oracle's
    restatement of OnlineStudy.step (itself checked against the reference on the same seeds, test_oracle_vs_reference_random.py)."""
    import random_cases
    c = random_cases.random_online_case(seed)
    S = run_product_case(c)
    with np.errstate(all='ignore'):
        w = oa.run_online(c)
    assert abs(S.logEvidence - w['logEvidence']) <= RTOL * abs(w['logEvidence'])
    for key in ('posteriorSequence', 'posteriorMeanValues', 'transitionModelSequence', 'localTransitionModelSequence'):
        np.testing.assert_allclose(np.asarray(getattr(S, key), dtype=float), np.asarray(w[key], dtype=float), rtol=RTOL, atol=ATOL, err_msg=key)
    for i in range(len(S.transitionModels)):
        np.testing.assert_allclose(np.asarray([h[i] for h in S.hyperParameterSequence], dtype=float),
                                   np.asarray([h[i] for h in w['hyperParameterSequence']], dtype=float), rtol=RTOL, atol=ATOL)


@pytest.mark.gpu
def test_online_study_with_wide_walk_on_the_second_parameter_matches_oracle():
    """The resume path (one forward step per data point from carried states) through the axis-1 pre-pass (blhip_hwide.hpp): walks on 'std'
    with stencil radii 32 and 81 beside a Static model."""
    c = dict(om=('Gaussian', [('mean', ('cint', -5, 5, 64)), ('std', ('oint', 0, 3, 120))], 'default'),
             models=[('wide', ('Combined', [('GRW', 'a', 0.3, 'mean', None), ('GRW', 'b', [0.2, 0.5], 'std', None)])),
                     ('static', ('Static',))],
             tm_prior=[0.6, 0.4], data=('series', 61, 8))
    S = run_product_case(c)
    with np.errstate(all='ignore'):
        w = oa.run_online(c)
    assert abs(S.logEvidence - w['logEvidence']) <= RTOL * abs(w['logEvidence'])
    for key in ('posteriorSequence', 'posteriorMeanValues', 'transitionModelSequence', 'localTransitionModelSequence'):
        np.testing.assert_allclose(np.asarray(getattr(S, key), dtype=float), np.asarray(w[key], dtype=float), rtol=RTOL, atol=ATOL, err_msg=key)
