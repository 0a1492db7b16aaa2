cd $GRAFT_REPO_ROOT
o=gpurun_out/r03z9; mkdir -p $o
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $o/t.txt
