"""Plot methods (reference tests/test_plot.py: every call below must draw without raising; Agg backend, nothing is shown)."""
import contextlib
import io

import numpy as np
import pytest

matplotlib = pytest.importorskip('matplotlib')
matplotlib.use('Agg')
import matplotlib.pyplot as plt   # noqa: E402

import bayesloop_amd as bl   # noqa: E402
from oracle_engine import OracleEngine   # noqa: E402

D15 = np.array([1, 2, 3, 4, 5])


@pytest.fixture(autouse=True)
def oracle_engine():
    prev = bl.set_engine(OracleEngine())
    yield
    bl.set_engine(prev)
    plt.close('all')


def drawn():
    ax = plt.gca()
    return len(ax.images) + len(ax.lines) + len(ax.collections) + len(ax.patches)


def test_plot_study():
    S = bl.Study(silent=True)
    S.loadData(D15, silent=True)
    S.set(bl.om.Poisson('rate', bl.oint(0, 6, 100)), bl.tm.Static(), silent=True)
    S.fit(silent=True)
    S.plot('rate')
    ax = plt.gca()
    assert len(ax.images) == 1 and len(ax.lines) == 1 and ax.get_ylabel() == 'rate'
    np.testing.assert_allclose(ax.lines[0].get_ydata(), S.getParameterMeanValues('rate'))
    plt.close()
    S.plot('rate', t=2)
    assert drawn() >= 1 and plt.gca().get_ylabel() == 'probability density'
    plt.close()
    x, p = S.getParameterDistributions('rate', plot=True, color='r')
    assert len(plt.gca().images) == 1 and p.shape == (5, 100)


def test_plot_hyperstudy():
    S = bl.HyperStudy(silent=True)
    S.loadData(D15, silent=True)
    S.set(bl.om.Poisson('rate', bl.oint(0, 6, 100)), bl.tm.GaussianRandomWalk('sigma', bl.cint(0, 0.2, 5), target='rate'), silent=True)
    S.fit(silent=True)
    S.plot('rate'); plt.close()
    S.plot('rate', t=2); plt.close()
    S.plot('sigma')
    ax = plt.gca()
    assert len(ax.patches) == 5 and ax.get_xlabel() == 'sigma'
    np.testing.assert_allclose([b.get_height() for b in ax.patches], S.getHyperParameterDistribution('sigma')[1])


def test_plot_changepointstudy():
    S = bl.ChangepointStudy(silent=True)
    S.loadData(D15, silent=True)
    T = bl.tm.SerialTransitionModel(bl.tm.Static(), bl.tm.ChangePoint('t1', 'all'),
                                    bl.tm.GaussianRandomWalk('sigma', bl.cint(0, 0.2, 3), target='rate'),
                                    bl.tm.ChangePoint('t2', 'all'), bl.tm.Static())
    S.set(bl.om.Poisson('rate', bl.oint(0, 6, 100)), T, silent=True)
    with contextlib.redirect_stdout(io.StringIO()):
        S.fit(silent=True)
    S.plot('rate'); plt.close()
    S.plot('rate', t=2); plt.close()
    S.plot('sigma'); plt.close()
    d, p = S.getDD(['t1', 't2'], plot=True)
    assert len(plt.gca().patches) == len(d)
    plt.close()
    S.getJHPD(['t1', 'sigma'], plot=True)
    assert drawn() >= 1


def test_plot_onlinestudy():
    S = bl.OnlineStudy(storeHistory=True, silent=True)
    S.setOM(bl.om.Poisson('rate', bl.oint(0, 6, 50)), silent=True)
    with contextlib.redirect_stdout(io.StringIO()):
        S.add('gradual', bl.tm.GaussianRandomWalk('sigma', bl.cint(0, 0.2, 5), target='rate'))
        S.add('static', bl.tm.Static())
        for d in np.arange(5):
            S.step(d)
    for args, kw in ((('rate',), {}), (('rate',), dict(t=2)), (('sigma',), {}), (('sigma',), dict(t=2)), (('gradual',), {}),
                     (('gradual',), dict(local=True))):
        S.plot(*args, **kw)
        assert drawn() >= 1, (args, kw)
        plt.close()
    S2 = bl.OnlineStudy(storeHistory=False, silent=True)
    S2.setOM(bl.om.Poisson('rate', bl.oint(0, 6, 50)), silent=True)
    with contextlib.redirect_stdout(io.StringIO()):
        S2.add('gradual', bl.tm.GaussianRandomWalk('sigma', bl.cint(0, 0.2, 5), target='rate'))
        S2.step(2)
    S2.plot('rate'); plt.close()
    S2.plot('sigma'); plt.close()
    with pytest.raises(bl.PostProcessingError):
        S2.plot('rate', t=0)
