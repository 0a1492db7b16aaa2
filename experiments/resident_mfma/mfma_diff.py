import sys, io, contextlib
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import bayesloop_amd as bl
import cases
eng = bl.get_engine()
res = {}
for v in (0.0, 1.0):
    eng.set_option('resident_mfma', v)
    S = cases.build(bl, 'c3_small')
    with contextlib.redirect_stdout(io.StringIO()):
        S.fit(**cases.fit_kwargs('c3_small'))
    res[v] = (np.array(S.posteriorSequence), float(S.logEvidence), dict(S.lastTiming))
print('logE', res[0.0][1], res[1.0][1], res[1.0][2].get('resident_fallbacks'), res[1.0][2].get('path'))
a, b = res[0.0][0], res[1.0][0]
print(a.shape)
for t in range(a.shape[0]):
    d = np.abs(a[t] - b[t]) / (np.abs(a[t]).max() + 1e-300)
    n0, n1 = d.shape
    blk = d.reshape(n0 // 16, 16, n1 // 16, 16).max(axis=(1, 3))
    print('t', t, 'max rel', d.max())
    if d.max() > 1e-9:
        print((blk > 1e-9).astype(int))
        i, j = np.unravel_index(np.argmax(d), d.shape); print('worst cell', i, j, a[t, i, j], b[t, i, j])
