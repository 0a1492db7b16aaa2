#!/bin/bash
# usage: tools/prof_cmd.sh <tag> <command...>   -- kernel trace + PMC passes of an arbitrary command
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats -f csv -d $out/trace -o t -- "$@" > $out/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $out/pmc1 -o p -- "$@" > $out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -f csv -d $out/pmc2 -o p -- "$@" > $out/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -f csv -d $out/pmc3 -o p -- "$@" > $out/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE -f csv -d $out/pmc4 -o p -- "$@" > $out/pmc4.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC -f csv -d $out/pmc5 -o p -- "$@" > $out/pmc5.log 2>&1
python tools/prof_summary.py $out
