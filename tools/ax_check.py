import sys, time
sys.path.insert(0, '/root/repo')
import bayesloop_amd as bl, bench
eng = bl.get_engine()
S, kw, units, desc = bench.make_study(bl, 'c4_both_axes')
S.fit(**kw)
for _ in range(2):
    t0 = time.perf_counter(); S.fit(**kw); eng.synchronize(); print('c4_both_axes fit ms %.1f' % ((time.perf_counter() - t0) * 1e3), {k: S.lastTiming[k] for k in ('resident_probe', 'xcd_order', 'forward_ms', 'backward_ms')})
