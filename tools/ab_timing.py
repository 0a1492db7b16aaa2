"""Kernel times of one bench workload on several builds of the library in ONE call (no parity gate: timing experiments whose results may be wrong):
    python tools/ab_timing.py <workload> <lib> [<lib> ...]      ("new" = the product build; others: bayesloop_amd/libblhip_<lib>.so)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, io, contextlib
sys.path.insert(0, %r)
import bayesloop_amd as bl, bench
eng = bl.get_engine()
S, kw, units, desc = bench.make_study(bl, sys.argv[1])
with contextlib.redirect_stdout(io.StringIO()):
    S.fit(**kw)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); S.fit(**kw); eng.synchronize(); ts.append(time.perf_counter() - t0)
t = S.lastTiming
print('%%-14s %%-8s fit %%.2f ms  forward %%.2f ms  backward %%.2f ms  fallbacks %%d  logE %%r' %% (sys.argv[1], sys.argv[2], min(ts) * 1e3, t['forward_ms'], t['backward_ms'], t['resident_fallbacks'], S.logEvidence))
''' % ROOT
w = sys.argv[1]
for rep in range(2):
    for lib in sys.argv[2:]:
        path = os.path.join(ROOT, 'bayesloop_amd', 'libblhip.so' if lib == 'new' else 'libblhip_%s.so' % lib)
        subprocess.run([sys.executable, '-c', CHILD, w, lib], env=dict(os.environ, BLHIP_LIBRARY=path))
