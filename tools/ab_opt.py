"""A/B of one engine option on bench workloads inside one process, interleaved:  python tools/ab_opt.py <option> <v0,v1,..> <workload> [...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import contextlib, io
import bayesloop_amd as bl
import bench
opt, vals, wls = sys.argv[1], [float(v) for v in sys.argv[2].split(',')], sys.argv[3:]
eng = bl.get_engine()
for wl in wls:
    S, kw, units, desc = bench.make_study(bl, wl)
    with contextlib.redirect_stdout(io.StringIO()):
        S.fit(**kw)
    for rep in range(3):
        for v in vals:
            eng.set_option(opt, v)
            with contextlib.redirect_stdout(io.StringIO()):
                S.fit(**kw)
                t0 = time.perf_counter(); S.fit(**kw); eng.synchronize(); dt = time.perf_counter() - t0
            tm = S.lastTiming
            nb = max(1, int(tm.get('batches', 1)))
            print('%-14s %s=%g: fit %.2f ms, forward %.3f us, backward %.3f us per step, fallbacks %d, logE %.12f' % (
                wl, opt, v, dt * 1e3, tm['forward_ms'] * 1e3 / (desc['T'] * nb), tm.get('backward_ms', 0.0) * 1e3 / (desc['T'] * nb), tm.get('resident_fallbacks', 0), S.logEvidence))
