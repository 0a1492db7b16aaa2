echo "== base"; python tools/sweep.py bucket 8,24,38 2>&1 | tail -3
for v in NOMFMA NOSTORE NOLOAD; do echo "== $v"; BLHIP_LIBRARY=bayesloop_amd/libblhip_$v.so python tools/sweep.py bucket 8,24,38 2>&1 | tail -3; done
echo "== base again"; python tools/sweep.py bucket 8,24,38 2>&1 | tail -3
