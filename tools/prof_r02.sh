#!/bin/bash
# usage (on the GPU box): tools/prof_r02.sh <workload> [bench args]  -- kernel trace/stats + the two HBM-traffic PMC passes (separate runs,
# as MI355X_MICROARCH.md prescribes) of `python bench.py --workload <w> --steps 1 --warmup 1` into gpurun_out/prof_r02_<w>/
w=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_r02_$w
mkdir -p $out
args="--workload $w --steps 1 --warmup 1 --no-extra --no-cpu $@"
rocprofv3 --kernel-trace --stats -f csv -d $out/trace -o t -- python bench.py $args > $out/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $out/pmc1 -o p -- python bench.py $args > $out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $out/pmc2 -o p -- python bench.py $args > $out/pmc2.log 2>&1
tail -1 $out/trace.log | head -c 600; echo
find $out -name "*kernel_stats.csv" | head -3
