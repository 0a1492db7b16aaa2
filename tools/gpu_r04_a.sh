#!/bin/bash
# round 4, call A: the one-hand-off resident kernel -- parity first, then C3 A/B against the classic scheme inside ONE call (same box)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04a; mkdir -p $out
timeout 900 python -m pytest tests -q -m gpu -n 4 > $out/pytest_gpu.txt 2>&1; tail -15 $out/pytest_gpu.txt
for opt in resident_onex=1 resident_onex=0 resident_onex=1 resident_onex=0; do
  timeout 300 python bench.py --workload c3 --steps 3 --warmup 1 --no-extra --no-cpu --no-pmc --no-e2e --opt $opt > $out/c3_$opt.json 2> $out/c3_$opt.err
  python - $out/c3_$opt.json $opt <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('c3', sys.argv[2], '%.4g' % d['value'], {k: round(v['avg_launch_us'], 2) for k, v in d['kernels'].items()}, d.get('log_evidence_rel_err'), d.get('resident_fallbacks'))
except Exception as e:
    print('c3', sys.argv[2], 'FAILED', e)
P
done
BLHIP_LIBRARY=$PWD/bayesloop_amd/libblhip_prof.so timeout 300 python bench.py --workload c3 --steps 1 --warmup 1 --no-extra --no-cpu --no-pmc --no-e2e > $out/c3_prof.json 2> $out/c3_prof.err; grep "blr prof" $out/c3_prof.err | tail -8
BLHIP_LIBRARY=$PWD/bayesloop_amd/libblhip_prof.so timeout 300 python bench.py --workload c3 --steps 1 --warmup 1 --no-extra --no-cpu --no-pmc --no-e2e --opt resident_onex=0 > $out/c3_prof0.json 2> $out/c3_prof0.err; grep "blr prof" $out/c3_prof0.err | tail -8
