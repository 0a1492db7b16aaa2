"""Development probe: a FULL fit on a 2048 x 2048 grid (the 128 x 128 backward kernel of the time-resident path).  usage: python tools/full2048_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayesloop_amd as bl
from bench import series

eng = bl.get_engine()
T = 48
for rep in range(2):
    S = bl.Study(silent=True)
    S.loadData(series(3, T), silent=True)
    S.set(bl.om.Gaussian('mean', bl.cint(-8, 8, 2048), 'std', bl.oint(0, 4, 2048)),
          bl.tm.CombinedTransitionModel(bl.tm.GaussianRandomWalk('s1', 0.015, target='mean'), bl.tm.GaussianRandomWalk('s2', 0.004, target='std')), silent=True)
    t0 = time.time(); S.fit(silent=True); eng.synchronize(); dt = time.time() - t0
    tm = S.lastTiming
    print('fit %.3f s' % dt, 'variants', tm['fwd_kernel_variant'], tm['bwd_kernel_variant'], 'fwd %.2f us/step  bwd %.2f us/step' % (1e3 * tm['forward_ms'] / T, 1e3 * tm['backward_ms'] / T),
          'logE %.10f' % S.logEvidence, 'fallbacks', tm['resident_fallbacks'])
    S._posterior_pending = None
    eng.release_posterior()
