python -m pytest tests -m gpu -q -x --timeout=900 2>&1 | tail -3
for n in 1024 2048; do python tools/sweep.py one $n 2d | tail -1; done
python bench.py --steps 2 --warmup 1 --no-cpu 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('c4', d['value'], d['roofline']['achieved'], d['kernels']['forward']['achieved'])
for k, v in d['extra'].items(): print(k, v['value'], {kk: (vv['avg_launch_us'], vv['achieved']) for kk, vv in v['kernels'].items()})
"
