"""The expectations of the reference's own API-level tests (tests/test_study.py, test_hyperstudy.py,
test_changepointstudy.py of christophmark/bayesloop), checked through the public classes of this package.

The numbers are the reference's known answers (same tolerances as there); the set-up code is this repo's.  Every entry
runs twice: on CPU with the oracle-backed test-double engine (host logic: priors, hyper-grids, accessors) and, marked
``gpu``, with the HIP engine (the product path).
"""
import numpy as np
import pytest

import bayesloop_amd as bl
from oracle_engine import OracleEngine

D15 = np.array([1, 2, 3, 4, 5])


def _stats():
    return pytest.importorskip('sympy.stats')


def gauss(prior):
    return bl.om.Gaussian('mean', bl.cint(0, 6, 20), 'sigma', bl.oint(0, 2, 20), prior=prior)


def inv_s3(m, s):
    return 1 / s ** 3


def grw_rs(sigma, pmin, target):
    return bl.tm.CombinedTransitionModel(bl.tm.GaussianRandomWalk('sigma', sigma, target=target),
                                         bl.tm.RegimeSwitch('log10pMin', pmin))


def serial(cp_prior=None, grw_values=None, grw_prior=None, bp_prior=None):
    return bl.tm.SerialTransitionModel(
        bl.tm.Static(),
        bl.tm.ChangePoint('ChangePoint', [0, 1], prior=cp_prior),
        bl.tm.CombinedTransitionModel(bl.tm.GaussianRandomWalk('sigma', grw_values, target='mean', prior=grw_prior),
                                      bl.tm.RegimeSwitch('log10pMin', [-3, -1])),
        bl.tm.BreakPoint('BreakPoint', 'all', prior=bp_prior),
        bl.tm.Static())


# id -> (study class, observation model factory, transition model factory, expectations); ref = reference file:line
ENTRIES = {
    # ---- tests/test_study.py:9-177 (one-parameter model, default 1000-point grid)
    'study1_0hp': dict(cls='Study', om=lambda: bl.om.Poisson('rate'), tm=lambda: bl.tm.Static(), param='rate', col=250,
                       dist=[0.00034] * 5, dist_rtol=1e-3, mean=[3.09761] * 5, logE=-10.4463425036, ref='test_study.py:10'),
    'study1_1hp': dict(cls='Study', om=lambda: bl.om.Poisson('rate'),
                       tm=lambda: bl.tm.GaussianRandomWalk('sigma', 0.1, target='rate'), param='rate', col=250,
                       dist=[0.000417, 0.000386, 0.000356, 0.000336, 0.000332],
                       mean=[3.073534, 3.08179, 3.093091, 3.104016, 3.111173], logE=-10.4337420351, ref='test_study.py:32'),
    'study1_2hp': dict(cls='Study', om=lambda: bl.om.Poisson('rate'), tm=lambda: grw_rs(0.1, -3, 'rate'), param='rate', col=250,
                       dist=[0.000412, 0.000376, 0.000353, 0.000336, 0.000332],
                       mean=[2.942708, 3.002756, 3.071995, 3.103038, 3.111179], logE=-10.4342948181, ref='test_study.py:54'),
    'study1_prior_array': dict(cls='Study', om=lambda: bl.om.Poisson('rate', bl.oint(0, 6, 1000), prior=np.ones(1000)),
                               tm=lambda: bl.tm.GaussianRandomWalk('sigma', 0.1, target='rate'), param='rate', col=250,
                               dist=[0.000221, 0.000202, 0.000184, 0.000172, 0.000172],
                               mean=[3.174159, 3.180812, 3.190743, 3.200642, 3.20722], logE=-10.0866227472,
                               ref='test_study.py:80'),
    'study1_prior_function': dict(cls='Study', om=lambda: bl.om.Poisson('rate', bl.oint(0, 6, 1000), prior=lambda x: 1. / x),
                                  tm=lambda: bl.tm.GaussianRandomWalk('sigma', 0.1, target='rate'), param='rate', col=250,
                                  dist=[0.000437, 0.000401, 0.000366, 0.000342, 0.000337],
                                  mean=[2.967834, 2.977838, 2.990624, 3.002654, 3.010419], logE=-11.3966589329,
                                  ref='test_study.py:102'),
    'study1_prior_sympy': dict(cls='Study', sympy=True,
                               om=lambda: bl.om.Poisson('rate', bl.oint(0, 6, 1000), prior=_stats().Exponential('expon', 1.)),
                               tm=lambda: bl.tm.GaussianRandomWalk('sigma', 0.1, target='rate'), param='rate', col=250,
                               dist=[0.000881, 0.00081, 0.00074, 0.00069, 0.000674],
                               mean=[2.627709, 2.643611, 2.661415, 2.677185, 2.687023], logE=-11.1819034242,
                               ref='test_study.py:124'),
    # ---- tests/test_study.py:179-315 (two-parameter model)
    'study2_0hp': dict(cls='Study', om=lambda: gauss(inv_s3), tm=lambda: bl.tm.Static(), param='mean', col=5,
                       dist=[0.013349] * 5, mean=[3.] * 5, logE=-16.1946904707, ref='test_study.py:180'),
    'study2_1hp': dict(cls='Study', om=lambda: gauss(inv_s3), tm=lambda: bl.tm.GaussianRandomWalk('sigma', 0.1, target='mean'),
                       param='mean', col=5, dist=[0.013547, 0.013428, 0.013315, 0.013241, 0.013232],
                       mean=[2.995242, 2.997088, 3., 3.002912, 3.004758], logE=-16.1865343702, ref='test_study.py:202'),
    'study2_2hp': dict(cls='Study', om=lambda: gauss(inv_s3), tm=lambda: grw_rs(0.1, -3, 'mean'), param='mean', col=5,
                       dist=[0.018848, 0.149165, 0.025588, 0.006414, 0.005426],
                       mean=[1.005987, 2.710129, 3.306985, 3.497192, 3.527645], logE=-14.3305753098, ref='test_study.py:224'),
    'study2_prior_array': dict(cls='Study', om=lambda: gauss(np.ones((20, 20))),
                               tm=lambda: bl.tm.GaussianRandomWalk('sigma', 0.1, target='mean'), param='mean', col=5,
                               dist=[0.02045, 0.020327, 0.020208, 0.020128, 0.020115],
                               mean=[2.99656, 2.997916, 3., 3.002084, 3.00344], logE=-10.9827282104, ref='test_study.py:250'),
    'study2_prior_function': dict(cls='Study', om=lambda: gauss(lambda m, s: 1. / s),
                                  tm=lambda: bl.tm.GaussianRandomWalk('sigma', 0.1, target='mean'), param='mean', col=5,
                                  dist=[0.018242, 0.018119, 0.018001, 0.017921, 0.01791],
                                  mean=[2.996202, 2.997693, 3., 3.002307, 3.003798], logE=-11.9842221343,
                                  ref='test_study.py:272'),
    'study2_prior_sympy': dict(cls='Study', sympy=True,
                               om=lambda: gauss([_stats().Uniform('u', 0, 6), _stats().Exponential('e', 2.)]),
                               tm=lambda: bl.tm.GaussianRandomWalk('sigma', 0.1, target='mean'), param='mean', col=5,
                               dist=[0.014305, 0.014183, 0.014066, 0.01399, 0.01398],
                               mean=[2.995526, 2.997271, 3., 3.002729, 3.004474], logE=-12.4324853153, ref='test_study.py:294'),
    # ---- tests/test_hyperstudy.py:9-187
    'hyper_0hp': dict(cls='HyperStudy', om=lambda: gauss(inv_s3), tm=lambda: bl.tm.Static(), param='mean', col=5,
                      dist=[0.013349] * 5, dist_rtol=1e-4, mean=[3.] * 5, mean_rtol=1e-5, logE=-16.1946904707, decimal=5,
                      ref='test_hyperstudy.py:10'),
    'hyper_2hp': dict(cls='HyperStudy', om=lambda: gauss(inv_s3), tm=lambda: grw_rs(bl.cint(0, 0.2, 2), [-3, -1], 'mean'),
                      param='mean', col=5, dist=[0.005589, 0.112966, 0.04335, 0.00976, 0.002909], dist_rtol=1e-4,
                      mean=[0.963756, 2.105838, 2.837739, 3.734359, 4.595412], mean_rtol=1e-5, logE=-10.7601875492, decimal=5,
                      hpd=('sigma', [[0., 0.2], [0.48943645, 0.51056355]], 1e-5),
                      joint=(['log10pMin', 'sigma'], [[-3., -1.], [0., 0.2]], [[0.00701834, 0.0075608], [0.48241812, 0.50300274]]),
                      ref='test_hyperstudy.py:61'),
    'hyper_prior_array': dict(cls='HyperStudy', om=lambda: gauss(inv_s3),
                              tm=lambda: bl.tm.GaussianRandomWalk('sigma', bl.cint(0, 0.2, 2), target='mean', prior=np.array([0.2, 0.8])),
                              param='mean', col=5, dist=[0.019149, 0.015184, 0.012369, 0.0109, 0.010722], dist_rtol=1e-4,
                              mean=[2.882151, 2.929385, 3., 3.070615, 3.117849], mean_rtol=1e-4, logE=-15.9915077133, decimal=5,
                              hpd=('sigma', [[0., 0.2], [0.16322581, 0.83677419]], 1e-5), ref='test_hyperstudy.py:105'),
    'hyper_prior_function': dict(cls='HyperStudy', om=lambda: gauss(inv_s3),
                                 tm=lambda: bl.tm.GaussianRandomWalk('sigma', bl.cint(0.1, 0.3, 2), target='mean', prior=lambda s: 1. / s),
                                 param='mean', col=5, dist=[0.025476, 0.015577, 0.012088, 0.010889, 0.010749], dist_rtol=1e-4,
                                 mean=[2.858477, 2.915795, 3., 3.084205, 3.141523], mean_rtol=1e-4, logE=-15.9898700147, decimal=5,
                                 hpd=('sigma', [[0.1, 0.3], [0.61609973, 0.38390027]], 1e-5), ref='test_hyperstudy.py:133'),
    'hyper_prior_sympy': dict(cls='HyperStudy', sympy=True, om=lambda: gauss(inv_s3),
                              tm=lambda: bl.tm.GaussianRandomWalk('sigma', bl.cint(0, 0.2, 2), target='mean',
                                                                  prior=_stats().Exponential('e', 1.)),
                              param='mean', col=5, dist=[0.016898, 0.014472, 0.012749, 0.011851, 0.011742], dist_rtol=1e-4,
                              mean=[2.927888, 2.95679, 3., 3.04321, 3.072112], mean_rtol=1e-4, logE=-17.0866290887, decimal=5,
                              hpd=('sigma', [[0., 0.2], [0.487971, 0.512029]], 1e-5), ref='test_hyperstudy.py:161'),
    # ---- tests/test_changepointstudy.py:9-100
    'cps_1cp_1bp_2hp': dict(cls='ChangepointStudy', om=lambda: gauss(inv_s3), tm=lambda: serial(grw_values=bl.cint(0, 0.2, 2)),
                            param='mean', col=5, dist=[0.012437, 0.030168, 0.01761, 0.001731, 0.001731],
                            mean=[0.968022, 1.956517, 3.476958, 4.161028, 4.161028], logE=-15.072007461556161, decimal=5,
                            hpd=('sigma', [[0., 0.2], [0.4963324, 0.5036676]], 1e-2),
                            duration=(['ChangePoint', 'BreakPoint'], [[1., 2., 3.], [0.01039273, 0.49395867, 0.49564861]]),
                            ref='test_changepointstudy.py:10'),
    'cps_hyperpriors': dict(cls='ChangepointStudy', sympy=True, om=lambda: gauss(inv_s3),
                            tm=lambda: serial(cp_prior=np.array([0.3, 0.7]), grw_values=bl.oint(0, 0.2, 2), grw_prior=lambda s: 1. / s,
                                              bp_prior=_stats().Normal('Normal', 3., 1.)),
                            param='mean', col=5, dist=[0.033729, 0.050869, 0.020636, 0.001647, 0.001647],
                            mean=[0.98944, 1.927195, 3.349921, 4.213695, 4.213695], logE=-15.709534690217343, decimal=5,
                            hpd=('sigma', [[0.06666667, 0.13333333], [0.66515107, 0.33484893]], 1e-2),
                            duration=(['ChangePoint', 'BreakPoint'], [[1., 2., 3.], [0.00373717, 0.40402616, 0.59223667]]),
                            ref='test_changepointstudy.py:56'),
}


def run_entry(name):
    e = ENTRIES[name]
    if e.get('sympy'):
        _stats()
    S = getattr(bl, e['cls'])(silent=True)
    S.loadData(D15, silent=True)
    S.set(e['om'](), e['tm'](), silent=True)
    with np.errstate(all='ignore'):
        S.fit(silent=True)
    np.testing.assert_allclose(S.getParameterDistributions(e['param'], density=False)[1][:, e['col']], e['dist'],
                               rtol=e.get('dist_rtol', 1e-2), err_msg=e['ref'])
    np.testing.assert_allclose(S.getParameterMeanValues(e['param']), e['mean'], rtol=e.get('mean_rtol', 1e-2), err_msg=e['ref'])
    np.testing.assert_almost_equal(S.logEvidence, e['logE'], decimal=e.get('decimal', 2), err_msg=e['ref'])
    if 'hpd' in e:
        pname, expect, rtol = e['hpd']
        x, p = S.getHyperParameterDistribution(pname)
        np.testing.assert_allclose(np.array([x, p]), expect, rtol=rtol, err_msg=e['ref'])
    if 'joint' in e:
        names, xy, pj = e['joint']
        x, y, p = S.getJointHyperParameterDistribution(names)
        np.testing.assert_allclose(np.array([x, y]), xy, rtol=1e-5, err_msg=e['ref'])
        np.testing.assert_allclose(p, pj, rtol=1e-5, err_msg=e['ref'])
    if 'duration' in e:
        names, expect = e['duration']
        d, p = S.getDurationDistribution(names)
        np.testing.assert_allclose(np.array([d, p]), expect, rtol=1e-2, err_msg=e['ref'])


@pytest.mark.parametrize('name', sorted(ENTRIES))
def test_reference_expectations_host_logic(name):
    prev = bl.set_engine(OracleEngine())
    try:
        run_entry(name)
    finally:
        bl.set_engine(prev)


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(ENTRIES))
def test_reference_expectations_gpu(name):
    run_entry(name)
