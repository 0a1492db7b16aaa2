#!/usr/bin/env python
"""
Generates the golden vectors under tests/golden/ by IMPORTING THE REFERENCE (``/root/reference``), which exists only
in the build container.  Run:  python tests/golden/gen_golden.py [case ...]

The reference is never edited and never copied; two harness-side shims are needed to run it on numpy 2 /
without pathos:
  * ``numpy.math = math``  (Poisson.pdf calls ``np.math.factorial``, observationModels.py:502)
  * a stub ``pathos.multiprocessing.ProcessPool`` whose ``map`` pickles the bound method per item, so that
    ``HyperStudy.fit(nJobs>1)`` (core.py:1307-1340) can be exercised in-process.
Result *attributes* are read directly (the accessors ``getParameterDistributions`` & co. break on numpy 2).

A fixture is data only: inputs (grids, data, prior arrays) and the reference's outputs.
"""
import io
import math
import os
import sys
import types
import warnings
import contextlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))           # tests/
warnings.filterwarnings('ignore')
np.math = math


def install_pathos_stub():
    import cloudpickle

    class ProcessPool:
        def __init__(self, nodes=1):
            self.nodes = nodes

        def map(self, f, *iterables):
            out = []
            for args in zip(*iterables):
                g = cloudpickle.loads(cloudpickle.dumps(f))
                out.append(cloudpickle.loads(cloudpickle.dumps(g(*args))))
            return out

        def close(self): pass
        def join(self): pass
        def terminate(self): pass
        def restart(self): pass

    pathos = types.ModuleType('pathos')
    mp = types.ModuleType('pathos.multiprocessing')
    mp.ProcessPool = ProcessPool
    pathos.multiprocessing = mp
    sys.modules['pathos'] = pathos
    sys.modules['pathos.multiprocessing'] = mp


install_pathos_stub()
sys.path.insert(0, '/root/reference')
import bayesloop as bl   # noqa: E402  (the reference)
import cases             # noqa: E402

FULL_LIMIT = 300_000     # store the full posterior sequence below this many doubles


def run(case):
    c = cases.CASES[case]
    S0 = cases.build(bl, case)
    out = {}
    out['marginal_count'] = len(S0.marginalGrid)
    for k, m in enumerate(S0.marginalGrid):
        out['marginal%d' % k] = np.asarray(m)
    out['latticeConstant'] = np.asarray(S0.latticeConstant, dtype=float)
    with contextlib.redirect_stdout(io.StringIO()):
        out['prior'] = np.array(S0._computePrior(silent=True), dtype=float)

    S = cases.build(bl, case)
    kw = cases.fit_kwargs(case)
    with contextlib.redirect_stdout(io.StringIO()):
        with np.errstate(all='ignore'):
            S.fit(**kw)
    out['rawData'] = np.asarray(S.rawData, dtype=float)
    out['formattedTimestamps'] = np.asarray(S.formattedTimestamps, dtype=float)
    out['logEvidence'] = np.float64(S.logEvidence)
    out['localEvidence'] = np.asarray(S.localEvidence, dtype=float)
    aborted = not np.isfinite(S.logEvidence) and c['study'] == 'Study'
    if not kw.get('evidenceOnly', False) and not aborted:
        out['posteriorMeanValues'] = np.asarray(S.posteriorMeanValues, dtype=float)
        post = np.asarray(S.posteriorSequence, dtype=float)
        T = post.shape[0]
        if post.size <= FULL_LIMIT and c.get('store') != 'sparse':
            out['posteriorSequence'] = post
        else:
            rows = cases.sparse_rows(T)
            out['posteriorRowsIndex'] = np.array(rows)
            stride = cases.sparse_stride(post.shape[1:])
            out['posteriorRowsStride'] = np.array(stride)
            out['posteriorRows'] = post[rows][(slice(None),) + tuple(slice(None, None, s) for s in stride)]
            if post.ndim == 3:
                out['marginalSequence0'] = post.sum(axis=2)
                out['marginalSequence1'] = post.sum(axis=1)
    if c['study'] in ('HyperStudy', 'ChangepointStudy') and len(S.hyperGridValues) > 1:
        out['logEvidenceList'] = np.asarray(S.logEvidenceList, dtype=float)
        out['hyperParameterDistribution'] = np.asarray(S.hyperParameterDistribution, dtype=float)
        out['hyperGridValues'] = np.asarray(S.hyperGridValues, dtype=float)
        out['flatHyperPriorValues'] = np.asarray(S.flatHyperPriorValues, dtype=float)
        out['hyperGridConstant'] = np.asarray(S.hyperGridConstant, dtype=float)
    if c['study'] == 'ChangepointStudy':
        out['allHyperGridValues'] = np.asarray(S.allHyperGridValues, dtype=float)
        out['mask'] = np.asarray(S.mask, dtype=bool)
    return out


def run_online(case):
    """OnlineStudy.step over the case's data: everything the study exposes after every step (storeHistory=True)."""
    c = cases.ONLINE_CASES[case]
    S = cases.build_online(bl, case)
    out = {}
    with contextlib.redirect_stdout(io.StringIO()):
        with np.errstate(all='ignore'):
            for d in cases.online_data(c):
                S.step(d)
    for k, m in enumerate(S.marginalGrid):
        out['marginal%d' % k] = np.asarray(m)
    out['rawData'] = np.asarray(S.rawData, dtype=float)
    out['logEvidence'] = np.float64(S.logEvidence)
    out['posteriorSequence'] = np.asarray(S.posteriorSequence, dtype=float)
    out['posteriorMeanValues'] = np.asarray(S.posteriorMeanValues, dtype=float)
    out['transitionModelSequence'] = np.asarray(S.transitionModelSequence, dtype=float)
    out['localTransitionModelSequence'] = np.asarray(S.localTransitionModelSequence, dtype=float)
    out['n_models'] = len(S.transitionModels)
    for i in range(len(S.transitionModels)):
        out['hyperParameterSequence%d' % i] = np.asarray([h[i] for h in S.hyperParameterSequence], dtype=float)
        out['logEvidenceList%d' % i] = np.asarray(S.logEvidenceList[i], dtype=float)
        out['parameterPosterior%d' % i] = np.asarray(S.parameterPosterior[i], dtype=float)
    out['hyperLogEvidenceList'] = np.asarray(S.hyperLogEvidenceList, dtype=float)
    return out


SIMULATE = {'c1_coal': np.arange(0, 9), 'kat_gaussian': np.linspace(-2, 8, 11), 'c4_small': np.linspace(-3, 3, 7)}


def run_simulate():
    """Study.simulate (core.py:566-597) on three fitted cases: time-averaged and at one time stamp, probability and density."""
    out = {}
    for case, x in SIMULATE.items():
        S = cases.build(bl, case)
        with contextlib.redirect_stdout(io.StringIO()):
            with np.errstate(all='ignore'):
                S.fit(**cases.fit_kwargs(case))
        t = S.formattedTimestamps[len(S.formattedTimestamps) // 2]
        out[case + '_x'] = np.asarray(x, dtype=float)
        out[case + '_t'] = np.float64(t)
        out[case + '_avg'] = np.asarray(S.simulate(x), dtype=float)
        out[case + '_at'] = np.asarray(S.simulate(x, t=t), dtype=float)
        out[case + '_avg_density'] = np.asarray(S.simulate(x, density=True), dtype=float)
    return out


def main():
    if sys.argv[1:] == ['simulate']:
        np.savez_compressed(os.path.join(HERE, 'simulate.npz'), **run_simulate())
        print('simulate.npz written')
        return
    names = sys.argv[1:] or (list(cases.CASES) + list(cases.ONLINE_CASES))
    for name in [n for n in names if n in cases.ONLINE_CASES]:
        out = run_online(name)
        path = os.path.join(HERE, name + '.npz')
        np.savez_compressed(path, **out)
        kat = cases.ONLINE_CASES[name].get('kat')
        msg = '' if kat is None else '  (reference test value %r, diff %.2e)' % (kat, abs(out['logEvidence'] - kat))
        print('%-26s logE=%r  %6.1f kB%s' % (name, float(out['logEvidence']), os.path.getsize(path) / 1e3, msg))
    for name in [n for n in names if n in cases.CASES]:
        out = run(name)
        path = os.path.join(HERE, name + '.npz')
        np.savez_compressed(path, **out)
        kat = cases.CASES[name].get('kat')
        msg = ''
        if kat is not None:
            msg = '  (reference test value %r, diff %.2e)' % (kat, abs(out['logEvidence'] - kat))
        print('%-26s logE=%r  %6.1f kB%s' % (name, float(out['logEvidence']), os.path.getsize(path) / 1e3, msg))


if __name__ == '__main__':
    main()
