"""Both-axes hyper-studies on square grids: the transposing chain-resident kernels (blhip_chainax.hpp) against the launch-per-step path
(option chain_ax1 = 0) of the same build.  python tools/ax_probe.py [n ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bayesloop_amd as bl
import bench
eng = bl.get_engine()
for n in [int(a) for a in sys.argv[1:]] or [128, 256, 512]:
    T, nh1, nh2 = 200, 16, 4
    for mode in (1, 0):
        eng.set_option('chain_ax1', mode)
        S = bl.HyperStudy(silent=True)
        S.loadData(bench.series(4, T), silent=True)
        S.set(bl.om.Gaussian('mean', bl.cint(-8, 8, n), 'std', bl.oint(0, 4, n)),
              bl.tm.CombinedTransitionModel(bl.tm.GaussianRandomWalk('s1', bl.cint(0, 0.3, nh1), target='mean'),
                                            bl.tm.GaussianRandomWalk('s2', bl.cint(0, 0.06, nh2), target='std')), silent=True)
        S.fit(silent=True)
        t0 = time.perf_counter()
        for _ in range(3):
            S.fit(silent=True)
        eng.synchronize()
        dt = (time.perf_counter() - t0) / 3
        tm = S.lastTiming
        print('n %4d chain_ax1 %d: %.2f ms per fit, %.3g cell-steps/s, variants %d/%d, fwd %.1f bwd %.1f ms, logE %.12f' % (
            n, mode, dt * 1e3, n * n * T * nh1 * nh2 / dt, tm['fwd_kernel_variant'], tm['bwd_kernel_variant'], tm['forward_ms'], tm['backward_ms'], S.logEvidence))
    eng.set_option('chain_ax1', 1)
