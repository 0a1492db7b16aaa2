#!/bin/bash
# usage (on the GPU box): tools/ab.sh <workload> <lib> [<lib> ...]  -- the same bench workload on several builds of the library
# (bayesloop_amd/libblhip_<lib>.so; "new" = the product build) inside ONE call (same box), interleaved twice
w=$1; shift
for rep in 1 2; do for lib in "$@"; do
  p=bayesloop_amd/libblhip_$lib.so; [ "$lib" = new ] && p=bayesloop_amd/libblhip.so
  BLHIP_LIBRARY=$PWD/$p timeout 300 python bench.py --workload $w --steps 3 --warmup 1 --no-extra --no-cpu --no-pmc --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], sys.argv[2], '%.4g' % d['value'], '%.3f ms' % d['ms_per_step'], d.get('log_evidence_rel_err'))" $w $lib
done; done
