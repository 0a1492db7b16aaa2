cd $GRAFT_REPO_ROOT
o=gpurun_out/r03f; mkdir -p $o
python __graft_entry__.py --smoke > $o/smoke.txt 2>&1; tail -6 $o/smoke.txt
timeout 1500 python bench.py --steps 5 --warmup 2 > $o/bench.json 2> $o/bench.err; tail -c 600 $o/bench.json
for w in c4 c5 c3 fwd2048 c4_both_axes; do tools/prof_r03.sh $w > /dev/null 2>&1; done
ls gpurun_out/prof_r03_*/trace/t_kernel_stats.csv
