"""One-off: the seeded random configurations of tests/random_cases.py through every alternative kernel path (engine options)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bayesloop_amd as bl, cases, compare, oracle_adapter as oa, random_cases

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
eng = bl.get_engine()
OPTS = [dict(), dict(fast=0), dict(mfma=0), dict(mfma_h=0), dict(recurrence=0),
        dict(fuse1d=0), dict(fuse1d=1), dict(fuse1d=3), dict(max_batch=2)]
DEFAULTS = dict(fast=1, mfma=1, mfma_h=1, recurrence=1, fuse1d=8, max_batch=1024, chain1d=1, chain1d_shift=1)
if os.environ.get('FUZZ_OPTS') == 'chain1d':       # round 4: the chain-resident 1-D kernel's flavours (pairs, spline shifts, shared table)
    OPTS = [dict(), dict(chain1d=2), dict(chain1d=0), dict(chain1d=2, chain1d_shift=0), dict(chain1d=2, max_batch=2)]

def result_of(S, c):
    res = dict(logEvidence=S.logEvidence, localEvidence=S.localEvidence)
    if not c.get('fit', {}).get('evidenceOnly', False) and np.isfinite(S.logEvidence):
        res['posteriorSequence'] = S.posteriorSequence; res['posteriorMeanValues'] = S.posteriorMeanValues
    for key in ('logEvidenceList', 'hyperParameterDistribution'):
        if hasattr(S, key) and getattr(S, key) is not None and len(np.atleast_1d(getattr(S, key))) > 0:
            res[key] = np.asarray(getattr(S, key))
    return res

def ill(S, want):
    with np.errstate(all='ignore'):
        liks = [np.asarray(S.observationModel.processedPdf(S.grid, seg), dtype=float) for seg in S.formattedData]
    return any(((L > 0) & (L < 2.3e-308)).any() for L in liks) or bool(np.isnan(np.asarray(want['localEvidence'], dtype=float)).any())

bad = 0; runs = 0
for gen in (random_cases.random_case, random_cases.random_hyper_case, random_cases.random_model_case):
    for seed in range(N):
        c = gen(seed); tol = None
        if isinstance(c, tuple): c, tol = c
        with np.errstate(all='ignore'):
            want = oa.run(c)
        for o in OPTS:
            for k, v in DEFAULTS.items(): eng.set_option(k, v)
            for k, v in o.items(): eng.set_option(k, v)
            S = cases.build(bl, c)
            try:
                with np.errstate(all='ignore'):
                    S.fit(**cases.fit_kwargs(c))
                got = result_of(S, c)
                gold = dict(logEvidence=want['logEvidence'], localEvidence=want['localEvidence'])
                for k in ('posteriorSequence', 'posteriorMeanValues', 'logEvidenceList', 'hyperParameterDistribution'):
                    if k in want and want[k] is not None and k in got and (k != 'posteriorMeanValues' or len(want[k])):
                        gold[k] = np.asarray(want[k])
                if 'logEvidenceList' in want and not np.all(np.isfinite(np.asarray(want['logEvidenceList'], dtype=float))):
                    got['localEvidence'] = gold['localEvidence']
                t = dict(tol or {})
                if ill(S, want): t['local_rtol'] = 2e-2
                compare.check(got, gold, compare.GPU_TOL, case_tol=t or None)
            except Exception as e:
                bad += 1
                print('FAIL', gen.__name__, seed, o, str(e)[:160].replace('\n', ' '), flush=True)
            runs += 1
print('runs', runs, 'failures', bad)
