#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04b; mkdir -p $out
timeout 900 python -m pytest tests -q -m gpu -n 4 -k "resident or full_chip or plug_in or user_defined or published or golden" > $out/pytest.txt 2>&1; tail -8 $out/pytest.txt
python tools/probe.py resident --n 1024 --T 300 --opts "resident_onex=1;resident_onex=0" 2>&1 | grep -v WARN | tee $out/probe1024.txt
BLHIP_LIBRARY=$PWD/bayesloop_amd/libblhip_prof.so python tools/probe.py resident --n 1024 --T 64 --modes evid,full --reps 1 --opts "resident_onex=1" 2>&1 | grep "blr prof" | tee $out/prof_onex.txt
timeout 300 python bench.py --workload coal_breakpoints --steps 2 --warmup 1 --no-extra --no-cpu --no-pmc --no-e2e > $out/coal.json 2> $out/coal.err; tail -c 1500 $out/coal.json; tail -3 $out/coal.err
