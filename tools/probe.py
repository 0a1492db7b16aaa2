"""Development probes on the GPU box, one file (round 4: the per-topic probe scripts of rounds 2 / 3 folded into sub-commands).
    python tools/probe.py resident [--n 1024] [--T 200] [--modes evid,fwdonly,full] [--opts a=1,b=2;a=0,b=2] [--reps 2]
        single-chain 2-D fits (time-resident kernel): per-step kernel time from blhip_last_timing for every option set
    python tools/probe.py hyper [--n0 512 --n1 512 --nh 64 --T 128] [--opts ...]      a HyperStudy over one random-walk width
Prints one line per (mode, option set, repetition)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bayesloop_amd as bl  # noqa: E402
from bench import series  # noqa: E402


def option_sets(text):
    out = []
    for part in (text or '').split(';'):
        out.append([(kv.split('=')[0], float(kv.split('=')[1])) for kv in part.split(',') if '=' in kv])
    return out or [[]]


def timed(S, kw, T):
    eng = bl.get_engine()
    t0 = time.time()
    S.fit(silent=True, **kw)
    eng.synchronize()
    dt = time.time() - t0
    tm = S.lastTiming
    line = 'wall %.1f ms  variants %d/%d  fwd %.2f  bwd %.2f us/step  fallbacks %d  logE %.10f' % (
        1e3 * dt, tm['fwd_kernel_variant'], tm['bwd_kernel_variant'], 1e3 * tm['forward_ms'] / T, 1e3 * tm['backward_ms'] / T,
        tm['resident_fallbacks'], S.logEvidence)
    S._posterior_pending = None
    eng.release_posterior()
    return line


def with_options(opts, fn):
    eng = bl.get_engine()
    for k, v in opts:
        eng.set_option(k, v)
    try:
        return fn()
    finally:
        pass


def resident(a):
    kws = dict(evid=dict(evidenceOnly=True), fwdonly=dict(forwardOnly=True), full={})
    n1 = a.n1 or a.n
    for mode in a.modes.split(','):
        for opts in option_sets(a.opts):
            for rep in range(a.reps):
                def run():
                    S = bl.Study(silent=True)
                    S.loadData(series(3, a.T), silent=True)
                    S.set(bl.om.Gaussian('mean', bl.cint(-8, 8, a.n), 'std', bl.oint(0, 4, n1)),
                          bl.tm.CombinedTransitionModel(bl.tm.GaussianRandomWalk('s1', 16.0 / a.n * 1.9, target='mean'),
                                                        bl.tm.GaussianRandomWalk('s2', 4.0 / n1 * 2.05, target='std')), silent=True)
                    return timed(S, kws[mode], a.T)
                print('resident %dx%d T=%d %-7s %-40s %s' % (a.n, n1, a.T, mode, ','.join('%s=%g' % kv for kv in opts), with_options(opts, run)), flush=True)


def hyper(a):
    for opts in option_sets(a.opts):
        for rep in range(a.reps):
            def run():
                S = bl.HyperStudy(silent=True)
                S.loadData(series(4, a.T), silent=True)
                S.set(bl.om.Gaussian('mean', bl.cint(-8, 8, a.n0), 'std', bl.oint(0, 4, a.n1)),
                      bl.tm.GaussianRandomWalk('sigma', bl.cint(0, 0.3, a.nh), target='mean'), silent=True)
                return timed(S, dict(evidenceOnly=True) if a.evid else {}, a.T)
            print('hyper %dx%d x %d T=%d %-40s %s' % (a.n0, a.n1, a.nh, a.T, ','.join('%s=%g' % kv for kv in opts), with_options(opts, run)), flush=True)


def chain1d(a):
    """1-D hyper-studies (Poisson rate, coal-mining counts repeated to T steps): the chain-resident 1-D kernel against the other 1-D paths"""
    coal = np.array([4, 5, 4, 1, 0, 4, 3, 4, 0, 6, 3, 3, 4, 0, 2, 6, 3, 3, 5, 4, 5, 3, 1, 4, 4, 1, 5, 5, 3, 4, 2, 5, 2, 2, 3, 4, 2, 1, 3, 2])
    for n in [int(x) for x in a.ns.split(',')]:
        for lw in [int(x) for x in a.lws.split(',')]:
            for B in [int(x) for x in a.Bs.split(',')]:
                for opts in option_sets(a.opts):
                    def run():
                        S = bl.HyperStudy(silent=True)
                        S.loadData(np.resize(coal, a.T), silent=True)
                        delta = 6.0 / (n + 1)
                        s_hi = (lw + 0.4) / 4.0 * delta             # radius int(4 sigma / delta + 0.5) = lw for the widest chain
                        S.set(bl.om.Poisson('rate', bl.oint(0, 6, n)), bl.tm.GaussianRandomWalk('sigma', np.linspace(0.5 * s_hi, s_hi, B), target='rate'), silent=True)
                        S.fit(silent=True)
                        return timed(S, {}, a.T)
                    print('chain1d n=%d lw=%d B=%d T=%d %-28s %s' % (n, lw, B, a.T, ','.join('%s=%g' % kv for kv in opts), with_options(opts, run)), flush=True)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest='cmd', required=True)
    r = sub.add_parser('resident')
    r.add_argument('--n', type=int, default=1024); r.add_argument('--n1', type=int, default=0); r.add_argument('--T', type=int, default=200)
    r.add_argument('--modes', default='evid,fwdonly,full'); r.add_argument('--opts', default=''); r.add_argument('--reps', type=int, default=2)
    h = sub.add_parser('hyper')
    h.add_argument('--n0', type=int, default=512); h.add_argument('--n1', type=int, default=512); h.add_argument('--nh', type=int, default=64)
    h.add_argument('--T', type=int, default=128); h.add_argument('--evid', action='store_true'); h.add_argument('--opts', default=''); h.add_argument('--reps', type=int, default=2)
    c = sub.add_parser('chain1d')
    c.add_argument('--ns', default='200,1000,4000'); c.add_argument('--lws', default='8,27,133'); c.add_argument('--Bs', default='2,20,256,1000')
    c.add_argument('--T', type=int, default=110); c.add_argument('--opts', default='chain1d=2;chain1d=0')
    a = ap.parse_args()
    dict(resident=resident, hyper=hyper, chain1d=chain1d)[a.cmd](a)
