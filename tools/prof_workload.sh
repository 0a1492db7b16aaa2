#!/bin/bash
# usage (on the GPU box): tools/prof_workload.sh <workload> [bench args]  -- kernel trace/stats + the two HBM-traffic PMC passes (separate runs,
# as MI355X_MICROARCH.md prescribes) of `python bench.py --workload <w> --steps 1 --warmup 1` into gpurun_out/prof_<w>/
w=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$w
mkdir -p $out
args="--workload $w --steps 1 --warmup 1 --no-extra --no-cpu --no-pmc --no-e2e $@"
rocprofv3 --kernel-trace --stats -f csv -d $out/trace -o t -- python bench.py $args > $out/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $out/pmc1 -o p -- python bench.py $args > $out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $out/pmc2 -o p -- python bench.py $args > $out/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -f csv -d $out/pmc3 -o p -- python bench.py $args > $out/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -f csv -d $out/pmc4 -o p -- python bench.py $args > $out/pmc4.log 2>&1
tail -1 $out/trace.log | head -c 400; echo
