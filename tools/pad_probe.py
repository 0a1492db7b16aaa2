import os, sys, time
sys.path.insert(0, os.getcwd())
import bayesloop_amd as bl
from bench import series
eng = bl.get_engine()
T = 48
for n0, n1 in ((2000, 1100), (1000, 1000)):
    for pad in (1, 0, 1):
        eng.set_option('resident_pad', pad)
        S = bl.Study(silent=True)
        S.loadData(series(3, T), silent=True)
        S.set(bl.om.Gaussian('mean', bl.cint(-8, 8, n0), 'std', bl.oint(0, 4, n1)),
              bl.tm.CombinedTransitionModel(bl.tm.GaussianRandomWalk('s1', 16.0 / n0 * 1.9, target='mean'), bl.tm.GaussianRandomWalk('s2', 4.0 / n1 * 1.9, target='std')), silent=True)
        S.fit(silent=True); eng.synchronize()
        tm = S.lastTiming
        print(n0, n1, 'resident_pad', pad, 'variants', tm['fwd_kernel_variant'], tm['bwd_kernel_variant'], 'fwd %.2f us/step  bwd %.2f us/step' % (1e3 * tm['forward_ms'] / T, 1e3 * tm['backward_ms'] / T), 'logE %.9f' % S.logEvidence)
        S._posterior_pending = None
        eng.release_posterior()
