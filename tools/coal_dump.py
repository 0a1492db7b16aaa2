import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bayesloop_amd as bl
import bench
S, kw, units, desc = bench.make_study(bl, 'coal_breakpoints')
t0 = time.time()
with np.errstate(all='ignore'):
    S.fit(silent=True)
print('fit', time.time() - t0, S.logEvidence, S.lastTiming)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
with np.errstate(all='ignore'):
    S.fit(silent=True)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
os.makedirs('gpurun_out/r04d', exist_ok=True)
np.savez_compressed('gpurun_out/r04d/coal_gpu.npz', logEvidenceList=np.asarray(S.logEvidenceList, dtype=float), hpd=np.asarray(S.hyperParameterDistribution),
                    localEvidence=np.asarray(S.localEvidence), means=np.asarray(S.posteriorMeanValues), post=np.asarray(S.posteriorSequence), hv=np.asarray(S.hyperGridValues))
