"""Wall time of small fits (what a user of the reference's tutorials runs): per-call overhead of fit() next to its kernel time."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bayesloop_amd as bl
from bench import series
eng = bl.get_engine()

def timeit(mk, kw, n=5):
    S = mk(); S.fit(**kw)
    ts = []
    for _ in range(n):
        S = mk(); t0 = time.perf_counter(); S.fit(**kw); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, S.lastTiming

def study2d(n, T):
    def mk():
        S = bl.Study(silent=True); S.loadData(series(3, T), silent=True)
        S.set(bl.om.Gaussian('mean', bl.cint(-8, 8, n), 'std', bl.oint(0, 4, n)),
              bl.tm.CombinedTransitionModel(bl.tm.GaussianRandomWalk('s1', 0.12, target='mean'), bl.tm.GaussianRandomWalk('s2', 0.03, target='std')), silent=True)
        return S
    return mk

def study1d(n, T):
    def mk():
        S = bl.Study(silent=True); S.loadData(np.random.default_rng(1).poisson(3.0, T).astype(float), silent=True)
        S.set(bl.om.Poisson('rate', bl.oint(0, 6, n)), bl.tm.GaussianRandomWalk('s', 0.2, target='rate'), silent=True)
        return S
    return mk

def hyper1d(n, T, k):
    def mk():
        S = bl.HyperStudy(silent=True); S.loadData(np.random.default_rng(1).poisson(3.0, T).astype(float), silent=True)
        S.set(bl.om.Poisson('rate', bl.oint(0, 6, n)), bl.tm.GaussianRandomWalk('s', bl.cint(0, 1.0, k), target='rate'), silent=True)
        return S
    return mk

for name, mk in (('Study 1-D 1000 pts, T=110 (coal mining)', study1d(1000, 110)), ('HyperStudy 1-D 1000 pts x 20 widths, T=110', hyper1d(1000, 110, 20)),
                 ('Study 2-D 200x200, T=100', study2d(200, 100)), ('Study 2-D 256x256, T=400', study2d(256, 400))):
    ms, tm = timeit(mk, dict(silent=True))
    print('%-46s wall %7.2f ms  engine total %7.2f  fwd %6.2f bwd %6.2f  variant %d/%d' % (name, ms, tm['total_ms'], tm['forward_ms'], tm['backward_ms'], tm['fwd_kernel_variant'], tm['bwd_kernel_variant']))
import cProfile, pstats
S = study1d(1000, 110)(); pr = cProfile.Profile(); pr.enable(); S.fit(silent=True); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
