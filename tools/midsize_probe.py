"""What a mid-size hyper-study costs through the launch-per-step kernels (ragged grid) and through the chain-resident path (the next
aligned grid): python tools/midsize_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bayesloop_amd as bl
eng = bl.get_engine()
def series(seed, T):
    rng = np.random.default_rng(seed); mu = np.cumsum(rng.normal(0, 0.02, T)); return mu + rng.normal(0, 1.0, T)
def run(n0, n1, nh, T, full=True):
    S = bl.HyperStudy(silent=True); S.loadData(series(4, T), silent=True)
    S.set(bl.om.Gaussian('mean', bl.cint(-8, 8, n0), 'std', bl.oint(0, 4, n1)), bl.tm.GaussianRandomWalk('sigma', bl.cint(0, 0.3, nh), target='mean'), silent=True)
    kw = dict(silent=True) if full else dict(silent=True, evidenceOnly=True)
    S.fit(**kw); ts = []
    for _ in range(3):
        t0 = time.perf_counter(); S.fit(**kw); eng.synchronize(); ts.append(time.perf_counter() - t0)
    t = S.lastTiming
    print('%4d x %4d  %3d chains  T %4d  %s: %7.2f ms  %.2e cell-steps/s  variants %d/%d' % (n0, n1, nh, T, 'full' if full else 'evid', min(ts) * 1e3, n0 * n1 * nh * T / min(ts), t['fwd_kernel_variant'], t['bwd_kernel_variant']), flush=True)
    S._posterior_pending = None; eng.release_posterior()
for full in (True, False):
    for (a, b) in ((200, 200), (256, 208), (100, 100), (128, 112), (300, 300), (512, 304), (500, 500), (512, 512)):
        run(a, b, 32, 500, full)
