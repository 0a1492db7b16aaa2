"""Development aid: wall time of a set of representative fits (not the bench workloads) -- run it with two builds of the library
(BLHIP_LIBRARY=...) to spot a regression outside what bench.py times.  usage: python tools/regress_sweep.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayesloop_amd as bl
from bench import series

eng = bl.get_engine()


def gauss(n0, n1):
    return bl.om.Gaussian('mean', bl.cint(-8, 8, n0), 'std', bl.oint(0, 4, n1))


def run(name, make, **kw):
    best = 1e9
    for rep in range(3):
        S = make()
        t0 = time.time(); S.fit(silent=True, **kw); eng.synchronize(); best = min(best, time.time() - t0)
        tm = S.lastTiming
        S._posterior_pending = None
        eng.release_posterior()
    print('%-34s %8.2f ms   variants %d/%d  logE %.9f' % (name, best * 1e3, tm['fwd_kernel_variant'], tm['bwd_kernel_variant'], S.logEvidence), flush=True)


def study(n0, n1, T, tm, seed=3):
    def make():
        S = bl.Study(silent=True); S.loadData(series(seed, T), silent=True); S.set(gauss(n0, n1), tm(), silent=True); return S
    return make


def hyper(n0, n1, T, tm, seed=4):
    def make():
        S = bl.HyperStudy(silent=True); S.loadData(series(seed, T), silent=True); S.set(gauss(n0, n1), tm(), silent=True); return S
    return make


grw2 = lambda a, b: (lambda: bl.tm.CombinedTransitionModel(bl.tm.GaussianRandomWalk('s1', a, target='mean'), bl.tm.GaussianRandomWalk('s2', b, target='std')))
run('study 2048^2 full T=32', study(2048, 2048, 32, grw2(0.015, 0.004)))
run('study 2048^2 forwardOnly T=32', study(2048, 2048, 32, grw2(0.015, 0.004)), forwardOnly=True)
run('study 1024^2 forwardOnly T=200', study(1024, 1024, 200, grw2(0.03, 0.008)), forwardOnly=True)
run('study 512^2 full T=400', study(512, 512, 400, grw2(0.06, 0.016)))
run('study 256^2 full T=400', study(256, 256, 400, grw2(0.12, 0.03)))
run('study 1000^2 full T=100 (ragged)', study(1000, 1000, 100, grw2(0.03, 0.008)))
run('study 1024x512 axis0 r=30 T=100', study(1024, 512, 100, lambda: bl.tm.GaussianRandomWalk('s', 0.12, target='mean')))
run('hyper 256^2 x64 T=200 full', hyper(256, 256, 200, lambda: bl.tm.GaussianRandomWalk('s', bl.cint(0, 0.3, 64), target='mean')))
run('hyper 256^2 x64 T=200 evidence', hyper(256, 256, 200, lambda: bl.tm.GaussianRandomWalk('s', bl.cint(0, 0.3, 64), target='mean')), evidenceOnly=True)
run('hyper 200x200 x32 T=300 full', hyper(200, 200, 300, lambda: bl.tm.GaussianRandomWalk('s', bl.cint(0, 0.4, 32), target='mean')))
run('hyper 512^2 x32 T=100 forwardOnly', hyper(512, 512, 100, lambda: bl.tm.GaussianRandomWalk('s', bl.cint(0, 0.3, 32), target='mean')), forwardOnly=True)
run('hyper 128^2 2hp 8x4 T=100', hyper(128, 128, 100, lambda: bl.tm.CombinedTransitionModel(bl.tm.GaussianRandomWalk('a', bl.cint(0, 0.4, 8), target='mean'), bl.tm.GaussianRandomWalk('b', bl.cint(0, 0.05, 4), target='std'))))


def cp(n, T):
    def make():
        x = series(5, T); x[T // 2:] += 2.0
        S = bl.ChangepointStudy(silent=True); S.loadData(x, silent=True); S.set(gauss(n, n), bl.tm.ChangePoint('tc', 'all'), silent=True); return S
    return make


run('changepoint 256^2 T=120 all', cp(256, 120))


def one_d(n, T):
    def make():
        S = bl.Study(silent=True); S.loadData(np.stack([series(7, T), np.ones(T)], 1), silent=True)
        S.set(bl.om.GaussianMean('mean', bl.cint(-8, 8, n)), bl.tm.GaussianRandomWalk('sigma', 0.02, target='mean'), silent=True); return S
    return make


run('1-D GaussianMean 4096 T=2000', one_d(4096, 2000))
run('1-D GaussianMean 65536 T=300', one_d(65536, 300))
