"""Where a bench workload's fit() spends its wall time: lastTiming + cProfile of the host side.  python tools/fit_profile.py <workload> [n_lines]"""
import os, sys, io, time, contextlib, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bayesloop_amd as bl
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else 'coal_breakpoints'
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 35
eng = bl.get_engine()
S, kw, units, desc = bench.make_study(bl, wl)
with contextlib.redirect_stdout(io.StringIO()):
    S.fit(**kw); S.fit(**kw)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); S.fit(**kw); eng.synchronize(); ts.append(time.perf_counter() - t0)
print('fit wall ms:', ' '.join('%.2f' % (t * 1e3) for t in ts))
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in S.lastTiming.items() if not isinstance(v, (list, dict))})
pr = cProfile.Profile()
with contextlib.redirect_stdout(io.StringIO()):
    pr.enable(); S.fit(**kw); eng.synchronize(); pr.disable()
st = pstats.Stats(pr, stream=sys.stdout); st.sort_stats(os.environ.get('SORT', 'cumulative')).print_stats(nl)
