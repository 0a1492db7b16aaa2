"""How long does it take to get a page-locked host block for the posterior sequence, and how fast is the read-back into it?
usage (GPU box): python tools/pin_probe.py [GiB]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bayesloop_amd as bl
from bayesloop_amd import _abi
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 16.0
n = int(gib * 2 ** 30)
eng = bl.get_engine()
lib = _abi.load()
for th in (1, 4, 8, 16, 32):
    os.environ['BLHIP_PIN_THREADS'] = str(th)
    t0 = time.perf_counter(); p = lib.blhip_host_alloc(n); t1 = time.perf_counter()
    print('threads %2d: blhip_host_alloc(%.0f GiB) %.3f s -> %s' % (th, gib, t1 - t0, 'ok' if p else 'FAILED'), flush=True)
    t0 = time.perf_counter(); lib.blhip_host_free(p); print('   free %.3f s' % (time.perf_counter() - t0), flush=True)
# a fit whose posterior is ~gib: 1024 x 1024 grid, T chosen
T = max(2, int(n // (1024 * 1024 * 8)))
rng = np.random.default_rng(3)
S = bl.Study(silent=True); S.loadData(np.cumsum(rng.normal(0, 0.02, T)) + rng.normal(0, 1, T), silent=True)
S.set(bl.om.Gaussian('mean', bl.cint(-8, 8, 1024), 'std', bl.oint(0, 4, 1024)),
      bl.tm.CombinedTransitionModel(bl.tm.GaussianRandomWalk('s1', 0.03, target='mean'), bl.tm.GaussianRandomWalk('s2', 0.008, target='std')), silent=True)
for pinned in ('1', '0', '1', '1'):
    eng._pinned.enabled = pinned == '1'
    S.fit(silent=True); eng.synchronize()
    t0 = time.perf_counter(); post = S.posteriorSequence; dt = time.perf_counter() - t0
    print('pinned=%s: read-back of %.1f GiB in %.3f s = %.1f GB/s' % (pinned, post.nbytes / 2 ** 30, dt, post.nbytes / dt / 1e9), flush=True)
    del post; S.posteriorSequence = None
