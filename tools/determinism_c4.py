"""The full-size C4 fit four times in one process: log-evidence, per-chain evidences, posterior means and four rows of the average
posterior must be bit-identical run to run (every sum of the chain-resident kernel has a fixed order; a race on a tagged sum or on a
partial accumulator would show as a difference).  python tools/determinism_c4.py"""
import sys, os, numpy as np, hashlib
sys.path.insert(0, os.getcwd())
import bench, bayesloop_amd as bl
S, kw, units, desc = bench.make_study(bl, 'c4', None)
sigs = []
for i in range(4):
    S.fit(**kw)
    post = S._posterior_pending
    rows = [post.row(t) for t in (0, 17, 128, 255)]
    h = hashlib.sha256(b''.join(r.tobytes() for r in rows)).hexdigest()[:16]
    sigs.append((repr(S.logEvidence), hashlib.sha256(np.asarray(S.logEvidenceList).tobytes()).hexdigest()[:16], h,
                 hashlib.sha256(np.asarray(S.posteriorMeanValues).tobytes()).hexdigest()[:16]))
    print(sigs[-1], flush=True)
print('bitwise identical:', all(s == sigs[0] for s in sigs))
