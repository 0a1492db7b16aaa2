cd $GRAFT_REPO_ROOT
o=gpurun_out/r03q; mkdir -p $o
tools/ab_r03.sh $o c5 base new 2>&1 | tee $o/ab.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "prefix or c5 or cp or change or padded or walk_plus or chain" 2>&1 | tail -5 | tee -a $o/ab.txt
