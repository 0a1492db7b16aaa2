cd $GRAFT_REPO_ROOT
o=gpurun_out/r03p; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "prefix or c5 or cp or change or padded or walk_plus" 2>&1 | tail -12 | tee $o/t.txt
for sp in 0 1 0 1; do
timeout 600 python bench.py --workload c5 --steps 2 --warmup 1 --no-extra --no-cpu --no-pmc --no-e2e --opt share_prefix=$sp > $o/c5_$sp.json 2> $o/c5_$sp.err
python - $sp <<'P'
import json,sys
d=json.loads(open('gpurun_out/r03p/c5_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print('share_prefix', sys.argv[1], '%.4g'%d['value'], round(d['ms_per_step'],1), {k:(round(v['avg_launch_us'],1), round(v['hbm']['bytes_per_cell_step'],2)) for k,v in d['kernels'].items()}, d['log_evidence_rel_err'], d['resident_fallbacks'])
P
done
