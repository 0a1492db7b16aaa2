#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04e; mkdir -p $out
timeout 900 python -m pytest tests -q -m gpu -n 4 > $out/pytest.txt 2>&1; tail -6 $out/pytest.txt
python tools/probe.py chain1d --ns 200,1000,4000 --lws 8,133 --Bs 20,256,1000 --opts "chain1d=2;chain1d=1" 2>&1 | grep -v WARN | tee $out/probe.txt
