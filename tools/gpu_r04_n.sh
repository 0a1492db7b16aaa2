#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04n; mkdir -p $out
FUZZ_OPTS=chain1d timeout 1500 python tools/fuzz_options.py 70 2>&1 | grep -v "WARN\|Stopping" | tail -25 | tee $out/fuzz.txt
