import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import bayesloop_amd as bl, cases, random_cases
eng = bl.get_engine()
eng.set_option('quiet', 0)
for seed in (1490, 1737, 1755, 1939, 1947, 2004, 2442):
    c = random_cases.random_chain_resident_case(seed)
    row = []
    for opts in ({}, dict(fold2=0), dict(fold2_cp=0), dict(fuse_accumulate=0)):
        for k, v in opts.items(): eng.set_option(k, v)
        eng.set_option('resident_ok', 1)
        S = cases.build(bl, c)
        with np.errstate(all='ignore'):
            S.fit(**cases.fit_kwargs(c))
        t = S.lastTiming
        row.append((opts, t['fwd_kernel_variant'], t['bwd_kernel_variant'], t['resident_fallbacks']))
        for k in opts: eng.set_option(k, 1)
    print(seed, c['study'], c.get('tm'), 'T', len(S.formattedData), 'grid', S.gridSize, row, flush=True)
