#!/bin/bash
# round 4, call o: the four-chain no-stencil fold kernel (C5): A/B against fold4 = 0, the change-point / chain-resident tests
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04o; mkdir -p $out
for o in "" "fold4=0" "" "fold4=0"; do
  BLHIP_ENGINE_OPTS=$o timeout 300 python bench.py --workload c5 --steps 2 --warmup 1 --no-extra --no-cpu --no-pmc --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1] or 'fold4=1', '%.4g' % d['value'], {k: round(v['avg_launch_us'],2) for k,v in d['kernels'].items()}, d.get('log_evidence_rel_err'), d['kernels']['backward'].get('hbm', {}).get('bytes_per_cell_step'))" "$o"
done | tee $out/c5.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -n 4 -k "cres or CHAINRES or changepoint or c5 or chain_res or prefix" > $out/pytest.txt 2>&1; tail -4 $out/pytest.txt
