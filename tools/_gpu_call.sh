cd $GRAFT_REPO_ROOT
o=gpurun_out/r03z5; mkdir -p $o
tools/ab_r03.sh $o fwd2048 base new 2>&1 | tee $o/ab.txt
tools/ab_r03.sh $o c3 base new 2>&1 | tee -a $o/ab.txt
for w in fwd2048 c3; do
BLHIP_LIBRARY=$PWD/bayesloop_amd/libblhip_prof.so timeout 300 python bench.py --workload $w --steps 1 --warmup 1 --no-extra --no-cpu --no-pmc --no-e2e > $o/prof_$w.json 2> $o/prof_$w.err; grep "blr prof" $o/prof_$w.err | tail -8 | tee -a $o/ab.txt
done
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "resident and not chain" 2>&1 | tail -5 | tee -a $o/ab.txt
