cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_c4_serial
BLHIP_OPTS=multistream=0 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof_c4_serial/trace -o t -- python bench.py --steps 1 --warmup 1 --no-extra --no-cpu > gpurun_out/prof_c4_serial/trace.log 2>&1
tail -1 gpurun_out/prof_c4_serial/trace.log | cut -c1-100
BLHIP_OPTS=multistream=0 python bench.py --steps 2 --warmup 1 --no-extra --no-cpu 2>/dev/null | tail -1 > gpurun_out/bench_c4_serial.json
python bench.py --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/bench_full.json
