#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04l; mkdir -p $out
tools/gpu_r04_j.sh 2>&1 | tail -20
python tools/hostprof.py coal_breakpoints 2>&1 | grep -v "WARN\|Stopping" | head -16
timeout 900 python -m pytest tests -q -m gpu -n 4 -x > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
