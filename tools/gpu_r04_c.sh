#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c; mkdir -p $out
timeout 1200 python -m pytest tests -q -m gpu -n 4 > $out/pytest.txt 2>&1; tail -12 $out/pytest.txt
timeout 300 python bench.py --workload coal_breakpoints --steps 2 --warmup 1 --no-extra --no-cpu --no-pmc --no-e2e > $out/coal.json 2> $out/coal.err; tail -c 1800 $out/coal.json; tail -3 $out/coal.err
