"""One-line summary of a bench.py JSON line (the compact line or the full record bench_detail.json):  python tools/benchline.py file.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]) if not open(sys.argv[1]).read().lstrip().startswith('{\n') else json.load(open(sys.argv[1]))
k = d['kernels']
us = {kk: round(v, 2) for kk, v in k.items() if kk.endswith('_us')} if 'fwd_us' in k or 'bwd_us' in k else {kk: round(v['avg_launch_us'], 2) for kk, v in k.items()}
print(d['config']['workload'][:40], us, 'value %.3g' % d['value'], 'relerr', d['log_evidence_rel_err'])
