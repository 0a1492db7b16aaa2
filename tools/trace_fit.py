import os, sys, io, time, contextlib
sys.path.insert(0, '/root/repo')
import bayesloop_amd as bl
import bench
eng = bl.get_engine()
for wl in sys.argv[1:]:
    S, kw, units, desc = bench.make_study(bl, wl)
    with contextlib.redirect_stdout(io.StringIO()):
        S.fit(**kw); S.fit(**kw)
    ts=[]
    for _ in range(5):
        t0 = time.perf_counter(); S.fit(**kw); eng.synchronize(); ts.append(time.perf_counter() - t0)
    print(wl, 'fit wall ms:', ' '.join('%.2f' % (t * 1e3) for t in ts), flush=True)
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in S.lastTiming.items() if not isinstance(v, (list, dict))}, flush=True)
    eng.set_option('trace', 1)
    S.fit(**kw)
    eng.set_option('trace', 0)
    sys.stderr.flush()
