#!/bin/bash
# round 4, call g: the two-phase axis-0 pass of the multi-chunk backward kernel (A/B: product / uniform segments in the backward kernel too /
# fused epilogue as before), the resident + padded suites, PMC passes of the 2048^2 forward kernel
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04g; mkdir -p $out
for rep in 1 2; do for lib in libblhip.so libblhip_ub.so libblhip_nodefer.so; do
  echo $lib; BLHIP_LIBRARY=$PWD/bayesloop_amd/$lib timeout 300 python tools/full2048_probe.py 2>&1 | tail -1
done; done | tee $out/full2048.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -n 4 -k "resident or RESIDENT or res_ or pad or 2048 or fuzz" > $out/pytest_res.txt 2>&1; tail -3 $out/pytest_res.txt
tools/prof_cmd.sh r04_fwd2048 python bench.py --workload fwd2048 --steps 2 --warmup 1 --no-extra --no-cpu --no-pmc --no-e2e > $out/prof.txt 2>&1; tail -3 $out/prof.txt
python tools/pmc_json.py gpurun_out/prof_r04_fwd2048 > $out/fwd2048_pmc_raw.json
