"""Study.optimize (reference core.py:488-565; tests/test_study.py:146-176, :317-347): COBYLA over evidence-only fits.
The assertions are the reference's own (same tolerances).  CPU: host logic over the oracle test double; GPU: the device."""
import contextlib
import io

import numpy as np
import pytest

import bayesloop_amd as bl


def optimize_1d():
    import sympy.stats as stats
    S = bl.Study(silent=True)
    S.loadData(np.array([1, 2, 3, 4, 5]), silent=True)
    S.setOM(bl.om.Poisson('rate', bl.oint(0, 6, 1000), prior=stats.Exponential('expon', 1.)), silent=True)
    S.setTM(bl.tm.CombinedTransitionModel(bl.tm.GaussianRandomWalk('sigma', 2.1, target='rate'),
                                          bl.tm.RegimeSwitch('log10pMin', -3)), silent=True)
    with contextlib.redirect_stdout(io.StringIO()):
        S.optimize()
    np.testing.assert_allclose(S.getParameterDistributions('rate', density=False)[1][:, 250],
                               [1.820641e-03, 2.083830e-03, 7.730833e-04, 1.977125e-04, 9.441302e-05], rtol=1e-02)
    np.testing.assert_allclose(S.getParameterMeanValues('rate'), [1.015955, 2.291846, 3.36402, 4.113622, 4.390356], rtol=1e-02)
    np.testing.assert_almost_equal(S.logEvidence, -9.47362827569, decimal=2)
    np.testing.assert_almost_equal(S.getHyperParameterValue('sigma'), 2.11216289063, decimal=2)
    np.testing.assert_almost_equal(S.getHyperParameterValue('log10pMin'), -3.0, decimal=3)


def optimize_2d():
    S = bl.Study(silent=True)
    S.loadData(np.array([1, 2, 3, 4, 5]), silent=True)
    S.setOM(bl.om.Gaussian('mean', bl.cint(0, 6, 20), 'sigma', bl.oint(0, 2, 20), prior=lambda m, s: 1 / s ** 3), silent=True)
    S.setTM(bl.tm.CombinedTransitionModel(bl.tm.GaussianRandomWalk('sigma', 1.07, target='mean'),
                                          bl.tm.RegimeSwitch('log10pMin', -3.90)), silent=True)
    with contextlib.redirect_stdout(io.StringIO()):
        S.optimize()
    np.testing.assert_allclose(S.getParameterDistributions('mean', density=False)[1][:, 5],
                               [9.903855e-03, 1.887901e-02, 8.257234e-05, 5.142727e-06, 2.950377e-06], rtol=1e-02)
    np.testing.assert_allclose(S.getParameterMeanValues('mean'), [0.979099, 1.951689, 3.000075, 4.048376, 5.020886], rtol=1e-02)
    np.testing.assert_almost_equal(S.logEvidence, -8.010466752050611, decimal=2)
    np.testing.assert_almost_equal(S.getHyperParameterValue('sigma'), 1.065854087589326, decimal=2)
    np.testing.assert_almost_equal(S.getHyperParameterValue('log10pMin'), -4.039735868499399, decimal=2)


@pytest.mark.parametrize('flow', [optimize_1d, optimize_2d])
def test_optimize_host_logic_over_oracle_engine(flow):
    from oracle_engine import OracleEngine
    prev = bl.set_engine(OracleEngine())
    try:
        flow()
    finally:
        bl.set_engine(prev)


@pytest.mark.gpu
@pytest.mark.parametrize('flow', [optimize_1d, optimize_2d])
def test_optimize_on_device(flow):
    flow()
