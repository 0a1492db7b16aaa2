#!/bin/bash
cd $GRAFT_REPO_ROOT
for w in c4 fwd2048 c3 c5 coal_hyper1000; do timeout 600 tools/prof_r04.sh $w; done
python tools/traffic_r04.py gpurun_out/prof_r04_c4 gpurun_out/prof_r04_c5 gpurun_out/prof_r04_c3 gpurun_out/prof_r04_fwd2048 gpurun_out/prof_r04_coal_hyper1000 > gpurun_out/r04_traffic.json 2> gpurun_out/r04_traffic.err; tail -3 gpurun_out/r04_traffic.err; head -c 1500 gpurun_out/r04_traffic.json
for w in c4 fwd2048 c3 c5 coal_hyper1000; do f=$(find gpurun_out/prof_r04_$w/trace -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r04_${w}_kernel_stats.csv; head -4 $f | cut -c1-220; done
