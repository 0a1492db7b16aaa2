#!/bin/bash
# usage (on the GPU box): tools/prof_all.sh <tag> <workload> ...  -- tools/prof_workload.sh for every workload, then the traffic JSON and the
# kernel-stats CSVs under gpurun_out/ as <tag>_traffic.json / <tag>_<workload>_kernel_stats.csv (copy what is to be judged into profiles/)
tag=$1; shift
cd $GRAFT_REPO_ROOT
for w in "$@"; do timeout 900 tools/prof_workload.sh $w; done
python tools/traffic.py $(for w in "$@"; do echo gpurun_out/prof_$w; done) > gpurun_out/${tag}_traffic.json 2> gpurun_out/${tag}_traffic.err; tail -3 gpurun_out/${tag}_traffic.err
for w in "$@"; do f=$(find gpurun_out/prof_$w/trace -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/${tag}_${w}_kernel_stats.csv; head -3 $f | cut -c1-200; done
