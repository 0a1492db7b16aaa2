#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04j; mkdir -p $out
python - <<'PY' 2> $out/trace_raw.txt | grep -v WARN
import sys, time; sys.path.insert(0, '.')
import bench, bayesloop_amd as bl
S, kw, units, desc = bench.make_study(bl, 'coal_breakpoints')
S.fit(**kw)
bl.get_engine().set_option('trace', 1)
S.fit(**kw)
PY
python - <<'PY' | tee $out/trace.txt
import re, collections
acc = collections.OrderedDict()
for l in open('gpurun_out/r04j/trace_raw.txt'):
    m = re.match(r'\[blhip trace\] (.*?)\s+([0-9.]+) ms', l)
    if m: acc[m.group(1)] = acc.get(m.group(1), 0.0) + float(m.group(2))
for k, v in acc.items(): print('%-45s %8.2f ms' % (k, v))
print('total', sum(acc.values()))
PY
