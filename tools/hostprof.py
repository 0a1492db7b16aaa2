"""cProfile of one C4 fit (host-side overheads around the kernels)."""
import cProfile, pstats, sys, os, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, bayesloop_amd as bl
name = sys.argv[1] if len(sys.argv) > 1 else 'c4'
S, kw, units, desc = bench.make_study(bl, name)
S.fit(**kw)
t0 = time.perf_counter(); S.fit(**kw); bl.get_engine().synchronize(); dt = time.perf_counter() - t0
print('wall %.1f ms; device timeline %.1f ms (fwd %.1f bwd %.1f acc %.1f)' % (dt * 1e3, S.lastTiming['total_ms'], S.lastTiming['forward_ms'], S.lastTiming['backward_ms'], S.lastTiming['accumulate_ms']))
pr = cProfile.Profile(); pr.enable(); S.fit(**kw); bl.get_engine().synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(18); print(s.getvalue()[:3500])
