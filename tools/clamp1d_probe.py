"""Clamps (RegimeSwitch / NotEqual) on 1-D grids: the chain-resident kernel's clamp flavour (bl1c::chain1d_kernel CL = 2, option chain1d_clamp = 1,
the default) against the launch-per-step generic kernel (chain1d_clamp = 0).  The reference's regime-switch tutorial shape: coal-mining counts,
Poisson rate on 1000 points.    python tools/clamp1d_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayesloop_amd as bl

eng = bl.get_engine()


def build(kind, n):
    S = (bl.HyperStudy if kind != 'study' else bl.Study)(silent=True)
    S.loadExampleData(silent=True)
    if kind == 'study':
        tm = bl.tm.CombinedTransitionModel(bl.tm.GaussianRandomWalk('sigma', 0.1, target='accident_rate'), bl.tm.RegimeSwitch('log10pMin', -7))
    elif kind == 'hyper_rs':
        tm = bl.tm.CombinedTransitionModel(bl.tm.GaussianRandomWalk('sigma', bl.cint(0.05, 0.3, 8), target='accident_rate'),
                                           bl.tm.RegimeSwitch('log10pMin', bl.cint(-10, -3, 16)))
    else:
        tm = bl.tm.NotEqual('log10pMin', bl.cint(-10, -3, 64))
    S.set(bl.om.Poisson('accident_rate', bl.oint(0, 6, n)), tm, silent=True)
    return S


for kind, n in (('study', 1000), ('study', 200), ('hyper_rs', 1000), ('hyper_ne', 1000)):
    row = []
    for opt in (1, 0):
        eng.set_option('chain1d_clamp', opt)
        S = build(kind, n)
        S.fit(silent=True); S.fit(silent=True)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); S.fit(silent=True); eng.synchronize(); ts.append(time.perf_counter() - t0)
        row.append((min(ts) * 1e3, S.lastTiming['fwd_kernel_variant'], S.logEvidence))
    eng.set_option('chain1d_clamp', 1)
    print('%-9s n = %4d: chain-resident %.2f ms (kernel %d)   launch per step %.2f ms (kernel %d)   x %.2f   logE %.10f / %.10f' % (
        kind, n, row[0][0], row[0][1], row[1][0], row[1][1], row[1][0] / row[0][0], row[0][2], row[1][2]))
