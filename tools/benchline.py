"""One-line summary of a bench.py JSON line:  python tools/benchline.py file.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['config']['workload'][:40], {k: round(v['avg_launch_us'], 2) for k, v in d['kernels'].items()}, 'value %.3g' % d['value'], 'relerr', d['log_evidence_rel_err'])
