"""Extract one kernel's body from a hipcc -S listing and print the instruction stream (compacted) around a mnemonic."""
import re, sys, collections
s = open(sys.argv[1]).read()
sym = sys.argv[2]
m = re.search(r'^(%s):[^\n]*\n(.*?)\n\s*s_endpgm' % re.escape(sym), s, re.S | re.M)
body = m.group(2).split('\n')
out = []
for l in body:
    l = l.strip()
    if not l or l.startswith(';') or (l.startswith('.') and not l.startswith('.LBB')):
        continue
    out.append(re.sub(r'\s+', ' ', l.split(';')[0]).strip())
open(sys.argv[3], 'w').write('\n'.join(out))
print(len(out), collections.Counter(x.split()[0] for x in out).most_common(30))
