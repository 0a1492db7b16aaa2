"""bl.Parser / Study.eval (reference bayesloop/parser.py, tests/test_parser.py): the reference's known answers through this
package's own parser, plus -- where the reference is mounted -- random queries against the reference's parser."""
import contextlib
import io
import math
import os
import sys
import warnings

import numpy as np
import pytest

import bayesloop_amd as bl
from oracle_engine import OracleEngine

D15 = np.array([1, 2, 3, 4, 5])


@pytest.fixture(autouse=True)
def oracle_engine(request):
    if request.node.get_closest_marker('gpu'):
        yield
        return
    prev = bl.set_engine(OracleEngine())
    yield
    bl.set_engine(prev)


def two_studies(b):
    S = b.Study()
    S.loadData(D15)
    S.setOM(b.om.Poisson('rate', b.oint(0, 6, 50)))
    S.setTM(b.tm.Static())
    S2 = b.Study()
    S2.loadData(D15)
    S2.setOM(b.om.Poisson('rate2', b.oint(0, 6, 50)))
    S2.setTM(b.tm.GaussianRandomWalk('sigma', 0.2, target='rate2'))
    with contextlib.redirect_stdout(io.StringIO()):
        S.fit()
        S2.fit()
    return S, S2


def check_reference_known_answers():
    S, S2 = two_studies(bl)
    P = bl.Parser(S, S2)
    np.testing.assert_almost_equal(P('log(rate2@1*2*1.2) + 4 + rate@2^2 > 20', silent=True), 0.19606860326174191, decimal=5)   # test_parser.py:26
    np.testing.assert_almost_equal(P('log(rate2*2*1.2) + 4 + rate^2 > 20', t=3, silent=True), 0.19772797081330246, decimal=5)  # :28
    x, p = P('log(rate2@1*2*1.2)+ 4 + rate@2^2', silent=True)
    np.testing.assert_allclose(p[100:105], [0.00732, 0.007495, 0.005775, 0.003511, 0.003949], rtol=1e-03)                      # :47-49

    H = bl.HyperStudy(silent=True)
    H.loadData(D15, silent=True)
    H.set(bl.om.Poisson('rate', bl.oint(0, 6, 50)), bl.tm.GaussianRandomWalk('sigma', bl.cint(0, 0.2, 5), target='rate'), silent=True)
    H.fit(silent=True)
    np.testing.assert_almost_equal(H.eval('exp(0.99*log(sigma))+1 > 1.1', silent=True), 0.60696006616644793, decimal=5)        # :60-63

    for history, n, query in ((True, 5, 'exp(0.99*log(sigma@2))+1 > 1.1'), (False, 3, 'exp(0.99*log(sigma))+1 > 1.1')):
        O = bl.OnlineStudy(storeHistory=history, silent=True)
        O.setOM(bl.om.Poisson('rate', bl.oint(0, 6, 50)), silent=True)
        with contextlib.redirect_stdout(io.StringIO()):
            O.add('gradual', bl.tm.GaussianRandomWalk('sigma', bl.cint(0, 0.2, 5), target='rate'))
            O.add('static', bl.tm.Static())
            for d in np.arange(n):
                O.step(d)
        np.testing.assert_almost_equal(O.eval(query, silent=True), 0.61228433813735061, decimal=5)                             # :74-90


def test_reference_known_answers_host_logic():
    check_reference_known_answers()


@pytest.mark.gpu
def test_reference_known_answers_gpu():
    check_reference_known_answers()


def test_errors():
    S, S2 = two_studies(bl)
    with pytest.raises(bl.ConfigurationError):
        bl.Parser()
    with pytest.raises(bl.ConfigurationError):
        bl.Parser(S, S)                                   # duplicate names
    P = bl.Parser(S, S2)
    with pytest.raises(bl.ConfigurationError):
        P('rate + 1 > 2', silent=True)                    # no time stamp
    with pytest.raises(bl.ConfigurationError):
        P('rate@1 > 1 > 0', silent=True)                  # two relations
    with pytest.raises(bl.ConfigurationError):
        P('nonsense@1 > 1', silent=True)


REFERENCE = '/root/reference'
QUERIES = [
    ('rate@2 > 1.5', None), ('rate@0 + rate@4 < 6', None), ('rate2@1 - rate2@3 > 0', None), ('rate@1*rate2@1 >= 4', None),
    ('sqrt(rate@3) + exp(-rate2@0) < 2.1', None), ('rate^2 - 2*rate2 > 1', 2), ('-rate + 3 > 0', 1), ('rate/rate2 <= 1', 4),
    ('log(rate2@2*2)^2 > 0.5', None), ('2^-rate@1 < 0.3', None), ('(rate@1 + rate2@2)*(rate@3 - 1) > 2', None),
    ('abs(rate@0 - 3) + gamma(rate2@4 + 1) > 8', None), ('rate@4 == rate@4', None),
]
DISTRIBUTIONS = ['rate@1 + rate2@2', 'log(rate2@1*2*1.2)+ 4 + rate@2^2', 'rate@0*rate@3', 'exp(rate@1/3) - rate2@0']


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, 'bayesloop')), reason='reference not mounted')
def test_random_queries_against_the_reference_parser():
    pytest.importorskip('pyparsing')
    sys.path.insert(0, REFERENCE)
    np.math = math
    try:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            import bayesloop as ref
        Sr, S2r = two_studies(ref)
        S, S2 = two_studies(bl)
        Pr, P = ref.Parser(Sr, S2r), bl.Parser(S, S2)
        compared = 0
        for q, t in QUERIES:
            try:
                with contextlib.redirect_stdout(io.StringIO()), np.errstate(all='ignore'):
                    want = Pr(q, t=t, silent=True)
            except (ValueError, TypeError, IndexError):
                P(q, t=t, silent=True)                     # e.g. a sign in front of a parameter: the reference's evaluator trips, this one must not
                continue
            got = P(q, t=t, silent=True)
            assert abs(got - want) <= 1e-9 + 1e-9 * abs(want), (q, got, want)
            compared += 1
        assert compared >= 11
        x, p = P('sqrt(rate2@4)', silent=True)              # a bare function call (the reference's evaluator cannot do this one)
        assert abs(p.sum() - 1.0) < 0.05 and np.all(np.diff(x) > 0)
        for q in DISTRIBUTIONS:
            with contextlib.redirect_stdout(io.StringIO()), np.errstate(all='ignore'):
                xr, pr = Pr(q, silent=True)
            x, p = P(q, silent=True)
            np.testing.assert_allclose(x, xr, rtol=1e-12, atol=0)
            np.testing.assert_allclose(p, pr, rtol=1e-9, atol=1e-15)
    finally:
        sys.path.remove(REFERENCE)


def hyper_studies(b):
    H = b.HyperStudy()
    H.loadData(D15)
    H.setOM(b.om.Gaussian('mean', b.cint(0, 6, 12), 'std', b.oint(0.2, 2, 9)))
    H.setTM(b.tm.CombinedTransitionModel(b.tm.GaussianRandomWalk('sigma', b.cint(0, 0.4, 4), target='mean'),
                                         b.tm.RegimeSwitch('pmin', [-5, -2])))
    C = b.ChangepointStudy()
    C.loadData(np.array([1, 1, 2, 1, 4, 5, 4, 5]))
    C.setOM(b.om.Poisson('lam', b.oint(0, 8, 40)))
    C.setTM(b.tm.ChangePoint('tc', 'all'))
    O = b.OnlineStudy(storeHistory=True)
    O.setOM(b.om.Poisson('nu', b.oint(0, 6, 30)))
    with contextlib.redirect_stdout(io.StringIO()):
        H.fit()
        C.fit()
        O.add('walk', b.tm.GaussianRandomWalk('sw', b.cint(0.05, 0.4, 4), target='nu'))
        O.add('static', b.tm.Static())
        for d in [2, 3, 1, 4, 2]:
            O.step(d)
    return H, C, O


HYPER_QUERIES = [
    ('sigma > 0.15', None), ('sigma*2 + pmin < -3.9', None), ('mean@2*sigma > 0.5', None), ('mean@1 + std@1 > 3', None),
    ('tc >= 3', None), ('tc - lam@2 > 0', None), ('lam@6 - lam@1 > 2', None), ('sw@3 > 0.2', None), ('nu@4*sw@4 < 0.6', None),
    ('exp(sigma) + tc/4 > 1.9', None), ('sw + sigma > 0.4', 2), ('mean*std - nu < 1', 3),
]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, 'bayesloop')), reason='reference not mounted')
def test_hyper_parameter_queries_against_the_reference_parser():
    pytest.importorskip('pyparsing')
    sys.path.insert(0, REFERENCE)
    np.math = math
    try:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            import bayesloop as ref
        with np.errstate(all='ignore'), warnings.catch_warnings():
            warnings.simplefilter('ignore')
            Pr, P = ref.Parser(*hyper_studies(ref)), bl.Parser(*hyper_studies(bl))
        compared = 0
        for q, t in HYPER_QUERIES:
            try:
                with contextlib.redirect_stdout(io.StringIO()), np.errstate(all='ignore'):
                    want = Pr(q, t=t, silent=True)
            except (ValueError, TypeError, IndexError):
                P(q, t=t, silent=True)
                continue
            got = P(q, t=t, silent=True)
            assert abs(got - want) <= 1e-9 + 1e-9 * abs(want), (q, got, want)
            compared += 1
        assert compared >= 10
    finally:
        sys.path.remove(REFERENCE)
