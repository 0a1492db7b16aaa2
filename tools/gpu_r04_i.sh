#!/bin/bash
# round 4, call i: C4 / C5 with the compile-time prefix test in the filtering chain kernels; forward-only resident fits (MODE 3)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04i; mkdir -p $out
tools/ab_r04.sh c4 new | tee $out/c4.txt
tools/ab_r04.sh c5 new | tee $out/c5.txt
python tools/probe.py resident --n 1024 --T 200 --modes fwdonly,full --reps 2 2>&1 | grep -v WARN | tee $out/probe.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -n 4 -k "resident or RESIDENT or res_ or forward_only or fwdonly or CHAINRES or chain" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
