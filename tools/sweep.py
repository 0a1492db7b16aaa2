"""Kernel sweep on the GPU: forward-step time vs grid size / segment length (prints us per launch and algorithmic GB/s)."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayesloop_amd as bl

def series(seed, T):
    rng = np.random.default_rng(seed); mu = np.cumsum(rng.normal(0, 0.02, T)); return mu + rng.normal(0, 1.0, T)

def study(n, T, s1f=1.0, s2f=1.0, two=True):
    S = bl.Study(silent=True); S.loadData(series(3, T), silent=True)
    tms = [bl.tm.GaussianRandomWalk('s1', 0.03 * 1024 / n * s1f, target='mean')]
    if two: tms.append(bl.tm.GaussianRandomWalk('s2', 0.008 * 1024 / n * s2f, target='std'))
    S.set(bl.om.Gaussian('mean', bl.cint(-8, 8, n), 'std', bl.oint(0, 4, n)), bl.tm.CombinedTransitionModel(*tms), silent=True)
    return S

eng = bl.get_engine()
for kv in os.environ.get('BLHIP_OPTS', '').split(','):
    if '=' in kv: eng.set_option(kv.split('=')[0], float(kv.split('=')[1]))
def run(n, T, opts, mode, **kw):
    for k, v in opts.items(): eng.set_option(k, v)
    S = study(n, T, **kw)
    fit = dict(silent=True, evidenceOnly=True) if mode == 'fwd' else dict(silent=True)
    S.fit(**fit); S.fit(**fit)
    t = S.lastTiming
    S._posterior_pending = None; eng.release_posterior()
    out = 'n=%5d T=%3d %-22s fwd %8.2f us %6.0f GB/s' % (n, T, opts, t['forward_ms'] * 1e3 / t['forward_launches'], 16.0 * n * n / (t['forward_ms'] * 1e-3 / t['forward_launches']) / 1e9)
    if mode != 'fwd':
        out += '   bwd %8.2f us %6.0f GB/s' % (t['backward_ms'] * 1e3 / t['backward_launches'], 32.0 * n * n / (t['backward_ms'] * 1e-3 / t['backward_launches']) / 1e9)
    print(out, flush=True)
    for k in opts: eng.set_option(k, 0 if k == 'fast_S' else 1)

which = sys.argv[1] if len(sys.argv) > 1 else 'size'
if which == 'size':
    for n in (512, 1024, 2048, 4096, 8192):
        run(n, 40 if n < 8192 else 12, {}, 'fwd')
    for n in (1024, 2048, 4096):
        run(n, 24, {}, 'full')
    for S_ in (16, 24, 32, 40, 48, 64, 80, 128, 256):
        run(2048, 40, {'fast_S': S_}, 'fwd')
    for S_ in (32, 64, 128, 256, 512, 1024):
        run(4096, 40, {'fast_S': S_}, 'fwd')
    run(2048, 40, {}, 'fwd', two=False)
    run(4096, 40, {}, 'fwd', two=False)
    run(4096, 40, {'recurrence': 0}, 'fwd')
    run(4096, 40, {'fast': 0}, 'fwd')
if which == 'one':
    n = int(sys.argv[2]); two = sys.argv[3] != '1d'
    run(n, 40, {}, 'fwd', two=two)
    run(n, 16, {}, 'full', two=two)
if which == 'bucket':
    # C4-like: 512x512 grid, 64 chains, GRW on 'mean' only, all chains in one radius bucket
    n, T, nh = 512, int(os.environ.get('SWEEP_T', 32)), 64
    for lw in ([int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (0, 4, 8, 12, 16, 24, 32, 38)):
        sig = (lw / 4.0) * (16.0 / (n - 1)) if lw else 0.0
        S = bl.HyperStudy(silent=True); S.loadData(series(4, T), silent=True)
        vals = bl.cint(sig * 0.97 + 1e-9, sig + 1e-9, nh) if lw else bl.cint(1e-6, 2e-6, nh)
        S.set(bl.om.Gaussian('mean', bl.cint(-8, 8, n), 'std', bl.oint(0, 4, n)),
              bl.tm.GaussianRandomWalk('sigma', vals, target='mean'), silent=True)
        S.fit(silent=True); S.fit(silent=True)
        t = S.lastTiming; cells = n * n * nh
        S._posterior_pending = None; eng.release_posterior()
        print('lw0~%2d  fwd %8.1f us %5.0f GB/s   bwd %8.1f us %5.0f GB/s  acc %6.1f ms total %6.1f ms' % (
            lw, t['forward_ms'] * 1e3 / t['forward_launches'], 16.0 * cells / (t['forward_ms'] * 1e-3 / t['forward_launches']) / 1e9,
            t['backward_ms'] * 1e3 / t['backward_launches'], 32.0 * cells / (t['backward_ms'] * 1e-3 / t['backward_launches']) / 1e9,
            t['accumulate_ms'], t['total_ms']), flush=True)
if which == 'S':
    n = int(sys.argv[2])
    for S_ in [int(x) for x in sys.argv[3].split(',')]:
        run(n, 24, {'fast_S': S_}, 'full')
    run(n, 24, {}, 'full')
if which == 'ms':
    for n in (1024, 2048):
        run(n, 24, {'multistream': 0}, 'full')
        run(n, 24, {}, 'full')
        run(n, 24, {'multistream': 0}, 'full')
        run(n, 24, {}, 'full')
if which == 'shape':
    # same cell count per chain (2^18), different row lengths: does the 512-byte column strip of a block cost DRAM efficiency?
    T, nh = 32, 64
    lw = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    for n0, n1 in ((4096, 64), (2048, 128), (1024, 256), (512, 512), (256, 1024), (128, 2048)):
        sig = (lw / 4.0) * (16.0 / (n0 - 1))
        S = bl.HyperStudy(silent=True); S.loadData(series(4, T), silent=True)
        S.set(bl.om.Gaussian('mean', bl.cint(-8, 8, n0), 'std', bl.oint(0, 4, n1)),
              bl.tm.GaussianRandomWalk('sigma', bl.cint(sig * 0.97 + 1e-9, sig + 1e-9, nh), target='mean'), silent=True)
        S.fit(silent=True); S.fit(silent=True)
        t = S.lastTiming; cells = n0 * n1 * nh
        S._posterior_pending = None; eng.release_posterior()
        print('%4d x %4d lw0~%2d  fwd %8.1f us %5.0f GB/s   bwd %8.1f us %5.0f GB/s' % (
            n0, n1, lw, t['forward_ms'] * 1e3 / t['forward_launches'], 16.0 * cells / (t['forward_ms'] * 1e-3 / t['forward_launches']) / 1e9,
            t['backward_ms'] * 1e3 / t['backward_launches'], 32.0 * cells / (t['backward_ms'] * 1e-3 / t['backward_launches']) / 1e9), flush=True)
