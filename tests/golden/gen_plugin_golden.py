#!/usr/bin/env python
"""Golden vectors of the transition-model PLUG-IN boundary (reference bayesloop/transitionModels.py:49-63, called at core.py:411, :467):
the reference itself fitted with the user-defined transition models of tests/plugin_models.py, and the reference's built-in models'
own computeForwardPrior / computeBackwardPrior called directly.  Imports /root/reference (build container only); writes
tests/golden/plugin_*.npz -- data only (inputs travel as the shared definitions in tests/plugin_models.py).
    python tests/golden/gen_plugin_golden.py"""
import math
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
warnings.filterwarnings('ignore')
np.math = math                      # Poisson.pdf calls np.math.factorial (observationModels.py:502)
sys.path.insert(0, '/root/reference')
import bayesloop as bl              # noqa: E402  (the reference)
import plugin_models as pm          # noqa: E402

M = pm.make(bl.tm)
out = {}
for name, (S, kw) in pm.studies(bl, M).items():
    S.fit(silent=True, **kw)
    out[name + '/logEvidence'] = np.float64(S.logEvidence)
    out[name + '/localEvidence'] = np.asarray(S.localEvidence, dtype=float)
    if not kw.get('evidenceOnly'):
        out[name + '/posteriorSequence'] = np.asarray(S.posteriorSequence, dtype=float)
        out[name + '/posteriorMeanValues'] = np.asarray(S.posteriorMeanValues, dtype=float)
    if hasattr(S, 'hyperParameterDistribution') and len(np.atleast_1d(S.hyperParameterDistribution)):
        out[name + '/hyperParameterDistribution'] = np.asarray(S.hyperParameterDistribution, dtype=float)
        out[name + '/logEvidenceList'] = np.asarray(S.logEvidenceList, dtype=float)
    print('%-26s logE = %.12f' % (name, S.logEvidence))
np.savez_compressed(os.path.join(HERE, 'plugin_fits.npz'), **out)

out = {}
for name, (S, model, calls) in pm.direct_calls(bl).items():
    S.setTransitionModel(model, silent=True)
    for k, (method, kind, t) in enumerate(calls):
        x = pm.distribution(kind, S.gridSize, seed=k)
        fn = model.computeForwardPrior if method == 'fwd' else model.computeBackwardPrior
        out['%s/%d' % (name, k)] = np.asarray(fn(x.copy(), t), dtype=float)
    print('%-18s %d calls' % (name, len(calls)))
np.savez_compressed(os.path.join(HERE, 'plugin_direct.npz'), **out)
