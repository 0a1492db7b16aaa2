"""The published break-point study (bench.py: coal_breakpoints) on the GPU against the reference's run of it: which chains stop on either
side, the user-visible logEvidence.  python tools/coal_check.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import contextlib, io
import bayesloop_amd as bl
import bench
gold = np.load(os.path.join(ROOT, 'tests', 'golden', 'bench_coal_breakpoints_full.npz'))
S, kw, units, desc = bench.make_study(bl, 'coal_breakpoints')
with np.errstate(all='ignore'), contextlib.redirect_stdout(io.StringIO()):
    S.fit(silent=True)
gl, rl = gold['logEvidenceList'], np.asarray(S.logEvidenceList, dtype=float)
fg, fr = np.isfinite(gl), np.isfinite(rl)
print('stops: reference %d, here %d, both %d, only reference %d, only here %d' % ((~fg).sum(), (~fr).sum(), (~fg & ~fr).sum(), (~fg & fr).sum(), (fg & ~fr).sum()))
both = fg & fr
print('max rel err per chain (both finite): %.3g' % np.max(np.abs(rl[both] - gl[both]) / np.abs(gl[both])))
print('logEvidence %.12f reference %.12f rel %.3g' % (S.logEvidence, float(gold['logEvidence']), abs(S.logEvidence - float(gold['logEvidence'])) / abs(float(gold['logEvidence']))))
print('timing', {k: v for k, v in S.lastTiming.items() if k in ('forward_ms', 'backward_ms', 'total_ms', 'fwd_kernel_variant', 'bwd_kernel_variant')})
post, want = np.asarray(S.posteriorSequence), gold['posteriorSequence']
print('average posterior: max |diff| %.3g (max value %.3g), max rel diff where want > 1e-6: %.3g' % (np.max(np.abs(post - want)), want.max(), np.max((np.abs(post - want) / np.maximum(want, 1e-300))[want > 1e-6])))
print('posterior means: max |diff| %.3g' % np.max(np.abs(np.asarray(S.posteriorMeanValues) - gold['posteriorMeanValues'])))
d2, p2 = S.getDurationDistribution(['t_1', 't_2'])
keep = np.isin(gold['durations'], d2)
ref_dd = gold['durationDistribution'][keep] / gold['durationDistribution'][keep].sum()
print('duration distribution: max |diff| %.3g, max rel %.3g' % (np.max(np.abs(p2 - ref_dd)), np.max(np.abs(p2 - ref_dd) / ref_dd)))
only_here = np.where(fg & ~fr)[0]
print('stops only here:', only_here, np.asarray(S.hyperGridValues)[only_here].tolist(), 'reference logE of it', gl[only_here])
