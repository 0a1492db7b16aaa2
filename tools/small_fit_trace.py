import os, sys, time
sys.path.insert(0, os.getcwd())
import bayesloop_amd as bl
from bench import series
eng = bl.get_engine()
def mk():
    S = bl.Study(silent=True); S.loadData(series(3, 400), silent=True)
    S.set(bl.om.Gaussian('mean', bl.cint(-8, 8, 256), 'std', bl.oint(0, 4, 256)),
          bl.tm.CombinedTransitionModel(bl.tm.GaussianRandomWalk('s1', 0.12, target='mean'), bl.tm.GaussianRandomWalk('s2', 0.03, target='std')), silent=True)
    return S
S = mk(); S.fit(silent=True)
eng.set_option('trace', 1)
S = mk(); t0 = time.time(); S.fit(silent=True); print('wall %.2f ms' % ((time.time() - t0) * 1e3), S.lastTiming['forward_ms'], S.lastTiming['backward_ms'], S.lastTiming['total_ms'])
eng.set_option('trace', 0)
import cProfile, pstats
S = mk(); pr = cProfile.Profile(); pr.enable(); S.fit(silent=True); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
