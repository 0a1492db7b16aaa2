"""Where a fresh process's first fit goes: import, engine (library + context), first / second / third fit of a bench workload, with the library's trace.
usage: python tools/cold_start.py [workload]"""
import os, sys, time, io, contextlib
t0 = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bayesloop_amd as bl
t1 = time.perf_counter()
eng = bl.get_engine()
t2 = time.perf_counter()
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
S, kw, units, desc = bench.make_study(bl, wl)
t3 = time.perf_counter()
eng.set_option('trace', 1.0)
ts = []
for i in range(3):
    a = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        S.fit(**kw)
    eng.synchronize()
    ts.append(time.perf_counter() - a)
    sys.stderr.write('---- fit %d: %.1f ms\n' % (i, ts[-1] * 1e3))
    if i == 0: eng.set_option('trace', 0.0)
print('import %.1f ms, engine %.1f ms, study %.1f ms, fits %s ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, ' '.join('%.1f' % (t * 1e3) for t in ts)))
