#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04k; mkdir -p $out
python - <<'PY' 2> $out/trace_raw.txt | grep -v WARN
import sys, time; sys.path.insert(0, '.')
import bench, bayesloop_amd as bl
S, kw, units, desc = bench.make_study(bl, 'c4')
S.fit(**kw)
eng = bl.get_engine()
t0 = time.perf_counter(); S.fit(**kw); eng.synchronize(); print('wall', time.perf_counter() - t0)
eng.set_option('trace', 1)
t0 = time.perf_counter(); S.fit(**kw); eng.synchronize(); print('wall (trace)', time.perf_counter() - t0)
for k, v in sorted(S.lastTiming.items()): print(k, v)
PY
python - <<'PY' | tee $out/trace.txt
import re, collections
acc = collections.OrderedDict()
for l in open('gpurun_out/r04k/trace_raw.txt'):
    m = re.match(r'\[blhip trace\] (.*?)\s+([0-9.]+) ms', l)
    if m: acc[m.group(1)] = acc.get(m.group(1), 0.0) + float(m.group(2))
for k, v in acc.items(): print('%-45s %8.2f ms' % (k, v))
print('total', sum(acc.values()))
PY
python tools/hostprof.py c4 2>&1 | grep -v WARN | head -40
