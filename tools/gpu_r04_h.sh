#!/bin/bash
# round 4, call h: stored state requested before the axis-1 pass (A/B), the full-fit forward flavour (MODE 2) on C3 / 2048^2, full GPU suite
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04h; mkdir -p $out
for rep in 1 2; do for lib in libblhip.so libblhip_early.so; do
  echo $lib; BLHIP_LIBRARY=$PWD/bayesloop_amd/$lib timeout 300 python tools/full2048_probe.py 2>&1 | tail -1
done; done | tee $out/full2048.txt
tools/ab_r04.sh c3 new early | tee $out/c3.txt
timeout 1200 python -m pytest tests -q -m gpu -n 4 > $out/pytest.txt 2>&1; tail -5 $out/pytest.txt
