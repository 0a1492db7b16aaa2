"""Summarise -Rpass-analysis=kernel-resource-usage output (stdin or file) per kernel: VGPR / AGPR / scratch / occupancy / LDS."""
import re, subprocess, sys
lines = open(sys.argv[1]).read().split('\n') if len(sys.argv) > 1 else sys.stdin.read().split('\n')
pat = sys.argv[2] if len(sys.argv) > 2 else ''
rows, cur = [], None
for l in lines:
    m = re.search(r'Function Name: (\S+)', l)
    if m:
        cur = {'name': m.group(1)}; rows.append(cur); continue
    for key, short in (('VGPRs', 'v'), ('AGPRs', 'a'), ('ScratchSize [bytes/lane]', 's'), ('Occupancy [waves/SIMD]', 'o'), ('LDS Size [bytes/block]', 'l')):
        m = re.search(r' ' + re.escape(key) + r': (\d+)', l)
        if m and cur is not None and short not in cur:
            cur[short] = int(m.group(1))
names = subprocess.run(['c++filt'], input='\n'.join(r['name'] for r in rows), capture_output=True, text=True).stdout.split('\n')
for r, n in zip(rows, names):
    n = re.sub(r'\(.*$', '', n).replace('void ', '')
    if pat in n:
        print('%-60s vgpr=%3d agpr=%3d scratch=%d occ=%d lds=%d' % (n, r.get('v', -1), r.get('a', -1), r.get('s', -1), r.get('o', -1), r.get('l', -1)))
