#!/bin/bash
# usage (on the GPU box): tools/ab_r03.sh <out dir> <workload> <lib> [<lib> ...]  -- the same bench workload on several builds of the
# library (bayesloop_amd/libblhip_<lib>.so; "new" = the product build) inside ONE call (same box): A/B numbers that can be compared
out=$1; w=$2; shift 2
mkdir -p $out
for lib in "$@"; do
  p=bayesloop_amd/libblhip_$lib.so; [ "$lib" = new ] && p=bayesloop_amd/libblhip.so
  for rep in 1 2; do
    BLHIP_LIBRARY=$PWD/$p timeout 300 python bench.py --workload $w --steps 3 --warmup 1 --no-extra --no-cpu --no-pmc --no-e2e > $out/${w}_${lib}_$rep.json 2> $out/${w}_${lib}_$rep.err
    python - $out/${w}_${lib}_$rep.json $lib $w <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[3], sys.argv[2], '%.4g' % d['value'], {k: round(v['avg_launch_us'], 2) for k, v in d['kernels'].items()}, d.get('log_evidence_rel_err'), d.get('resident_fallbacks'))
except Exception as e:
    print(sys.argv[3], sys.argv[2], 'FAILED', e)
P
  done
done
