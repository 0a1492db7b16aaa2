"""Development probe: a hyper-study over a random-walk width on a 1024-row grid (radii up to 77 > the matrix-pipe kernels' 40) with the
column pre-pass (wide_v = 1) and on the generic kernel (wide_v = 0).  usage: python tools/widev_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayesloop_amd as bl
from bench import series

eng = bl.get_engine()
n0, n1, nh, T = 1024, 512, 64, 64
for wv in (1, 0, 1):
    eng.set_option('wide_v', wv)
    S = bl.HyperStudy(silent=True)
    S.loadData(series(4, T), silent=True)
    S.set(bl.om.Gaussian('mean', bl.cint(-8, 8, n0), 'std', bl.oint(0, 4, n1)),
          bl.tm.GaussianRandomWalk('sigma', bl.cint(0, 0.3, nh), target='mean'), silent=True)
    t0 = time.time(); S.fit(silent=True); eng.synchronize(); dt = time.time() - t0
    tm = S.lastTiming
    print('wide_v', wv, 'fit %.3f s' % dt, '%.3g cells*steps/s' % (n0 * n1 * nh * T / dt), 'logE %.12f' % S.logEvidence,
          'variant', tm['fwd_kernel_variant'], 'fwd %.1f ms bwd %.1f ms' % (tm['forward_ms'], tm['backward_ms']))
    S._posterior_pending = None
    eng.release_posterior()
