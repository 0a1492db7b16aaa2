"""Phase stamps of blr::resident_mfma_kernel on a bench workload (option resident_prof):  python tools/mfma_prof.py [workload]"""
import os, sys, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bayesloop_amd as bl
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else 'c3'
eng = bl.get_engine()
S, kw, units, desc = bench.make_study(bl, wl)
eng.set_option('resident_mfma', 1.0)
with contextlib.redirect_stdout(io.StringIO()):
    S.fit(**kw)
eng.set_option('resident_prof', 1.0)
with contextlib.redirect_stdout(io.StringIO()):
    S.fit(**kw)
