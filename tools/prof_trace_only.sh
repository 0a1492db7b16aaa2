#!/bin/bash
# usage (on the GPU box): tools/prof_trace_only.sh <workload> <tag> [bench args] -- rocprofv3 --kernel-trace --stats of one bench workload, nothing else
w=$1; tag=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/trace_${w}_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats -f csv -d $out -o t -- python bench.py --workload $w --steps 5 --warmup 2 --no-extra --no-cpu --no-pmc --no-e2e "$@" > $out/log.txt 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
head -4 $f | cut -c1-200
