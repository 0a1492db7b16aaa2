import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
import bayesloop_amd as bl
if len(sys.argv) > 1 and sys.argv[1] == 'oracle':
    from oracle_engine import OracleEngine
    bl.set_engine(OracleEngine())
for slope, t1, t2 in ((-2.0, 1888, 1891), (-2.0, 1890, 1893), (-1.5172413793103448, 1874, 1878)):
    S = bl.Study(silent=True)
    S.loadExampleData(silent=True)
    mask = (S.rawTimestamps >= 1870) * (S.rawTimestamps <= 1910)
    S.rawTimestamps = S.rawTimestamps[mask]; S.rawData = S.rawData[mask]
    S.set(bl.om.Poisson('accident_rate', bl.oint(0, 6, 1000)),
          bl.tm.SerialTransitionModel(bl.tm.Static(), bl.tm.BreakPoint('t_1', t1), bl.tm.Deterministic(lambda t, slope=slope: t * slope, target='accident_rate'),
                                      bl.tm.BreakPoint('t_2', t2), bl.tm.Static()), silent=True)
    with np.errstate(all='ignore'):
        S.fit(silent=True)
    print(slope, t1, t2, 'logE', S.logEvidence)
    if np.isfinite(S.logEvidence):
        P = np.asarray(S.posteriorSequence)
        for i in range(16, 23):
            print('   ', i, S.formattedTimestamps[i], 'sum', P[i].sum(), 'min', P[i].min(), 'max', P[i].max(), 'local', S.localEvidence[i])
