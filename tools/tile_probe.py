import os, sys, time
sys.path.insert(0, os.getcwd())
import bayesloop_amd as bl
from bench import series
eng = bl.get_engine()
T = 200
for n in (128, 256, 384, 512, 500, 768):
    for mt in (32, 64, 128):
        eng.set_option('resident_min_tile', mt)
        S = bl.Study(silent=True); S.loadData(series(3, T), silent=True)
        S.set(bl.om.Gaussian('mean', bl.cint(-8, 8, n), 'std', bl.oint(0, 4, n)),
              bl.tm.CombinedTransitionModel(bl.tm.GaussianRandomWalk('s1', 16.0 / n * 1.9, target='mean'), bl.tm.GaussianRandomWalk('s2', 4.0 / n * 1.9, target='std')), silent=True)
        S.fit(silent=True); eng.synchronize(); tm = S.lastTiming
        print(n, 'min_tile', mt, 'variants', tm['fwd_kernel_variant'], tm['bwd_kernel_variant'], 'fwd %.2f bwd %.2f us/step' % (1e3 * tm['forward_ms'] / T, 1e3 * tm['backward_ms'] / T), flush=True)
        S._posterior_pending = None; eng.release_posterior()
