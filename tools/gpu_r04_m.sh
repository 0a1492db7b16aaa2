#!/bin/bash
# round 4, call m: the default bench line (all extras), the new Deterministic-on-chain1d tests
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04m; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -n 4 -k "deterministic_steps or published" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
timeout 1500 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 600 $out/bench.err; python tools/benchline.py $out/bench.json 2>/dev/null | head -40
