"""Time one rank's share of the C4 workload on a single GPU: contiguous chunks (np.array_split) vs strided assignment."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayesloop_amd as bl

def series(seed, T):
    rng = np.random.default_rng(seed); mu = np.cumsum(rng.normal(0, 0.02, T)); return mu + rng.normal(0, 1.0, T)

n, T, nh, N = 512, 256, 512, int(sys.argv[1]) if len(sys.argv) > 1 else 8
sig = bl.cint(0, 0.3, nh)
eng = bl.get_engine()
for kv in os.environ.get('BLHIP_OPTS', '').split(','):
    if '=' in kv: eng.set_option(kv.split('=')[0], float(kv.split('=')[1]))
only = os.environ.get('SHARD_ONLY')
def run(vals, tag):
    S = bl.HyperStudy(silent=True); S.loadData(series(4, T), silent=True)
    S.set(bl.om.Gaussian('mean', bl.cint(-8, 8, n), 'std', bl.oint(0, 4, n)),
          bl.tm.GaussianRandomWalk('sigma', list(vals), target='mean'), silent=True)
    S.fit(silent=True)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); S.fit(silent=True); eng.synchronize(); ts.append(time.perf_counter() - t0)
    S._posterior_pending = None; eng.release_posterior()
    print('%-28s %3d chains  %.1f ms  (%.3e cell-steps/s)' % (tag, len(vals), min(ts) * 1e3, n * n * T * len(vals) / min(ts)), flush=True)
if not only: run(sig, 'all 512')
parts = np.array_split(np.arange(nh), N)
for r in (() if only else (0, N // 2, N - 1)):
    run(sig[parts[r]], 'contiguous chunk %d/%d' % (r, N))
for r in (0, N - 1):
    run(sig[r::N], 'strided %d::%d' % (r, N))
