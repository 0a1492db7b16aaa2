#!/bin/bash
# usage: tools/prof.sh <tag> <bench args...>   -- kernel trace + PMC passes into gpurun_out/prof_<tag>/
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats -f csv -d $out/trace -o t -- python bench.py "$@" --no-extra --no-cpu > $out/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $out/pmc1 -o p -- python bench.py "$@" --no-extra --no-cpu > $out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -f csv -d $out/pmc2 -o p -- python bench.py "$@" --no-extra --no-cpu > $out/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -f csv -d $out/pmc3 -o p -- python bench.py "$@" --no-extra --no-cpu > $out/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE -f csv -d $out/pmc4 -o p -- python bench.py "$@" --no-extra --no-cpu > $out/pmc4.log 2>&1
find $out -name "*.csv" | head -20
python tools/prof_summary.py $out
