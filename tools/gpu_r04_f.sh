#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04f; mkdir -p $out
timeout 900 python -m pytest tests -q -m gpu -n 4 > $out/pytest.txt 2>&1; tail -5 $out/pytest.txt
tools/ab_r04.sh fwd2048 new segv; tools/ab_r04.sh c3 new segv
python tools/full2048_probe.py 2>&1 | tail -1
python tools/probe.py chain1d --ns 1000,4000 --lws 8,133 --Bs 256,1000 --opts "chain1d=2,chain1d_pair=2;chain1d=2,chain1d_pair=0" 2>&1 | grep -v WARN | tee $out/probe.txt
