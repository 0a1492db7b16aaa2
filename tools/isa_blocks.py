"""Static instruction statistics of the largest loop of every kernel in a `hipcc -save-temps` listing whose mangled name contains one
of the given substrings: length, branches, MFMA / vector / scalar instruction counts, v_readlane (scalar spill reloads).
usage: python tools/isa_blocks.py file-hip-amdgcn-amd-amdhsa-gfx950.s chain_fold2 resident_kernel
(Static counts include every path of the loop body: a wave executes one side of each uniform branch.)"""
import re,collections,sys
s=open(sys.argv[1]).read()
for m0 in re.finditer(r'\n(_ZN\w+):', s):
    name=m0.group(1)
    if not any(k in name for k in sys.argv[2:]): continue
    i=m0.start(); j=s.index('.Lfunc_end',i)
    L=[l.strip() for l in s[i:j].split('\n')]
    L=[l for l in L if l and not l.startswith(';')]
    labels={l.split(':')[0]:k for k,l in enumerate(L) if re.match(r'^\.?\w+:',l)}
    loops=[]
    for k,l in enumerate(L):
        m=re.match(r's_cbranch_\w+ (\S+)|s_branch (\S+)',l)
        if m:
            t=m.group(1) or m.group(2)
            if t in labels and labels[t]<k: loops.append((k-labels[t],labels[t],k,t))
    loops.sort(reverse=True)
    n,a,b,t=loops[0]
    body=L[a:b]
    c=collections.Counter(x.split()[0] for x in body if not re.match(r'^\.?\w+:',x))
    print(name[:70],'\n  loop',n,'branches',sum(v for k,v in c.items() if k.startswith('s_cbranch') or k=='s_branch'),'mfma',sum(v for k,v in c.items() if 'mfma' in k),'valu',sum(v for k,v in c.items() if k.startswith('v_') and 'mfma' not in k),'readlane',c['v_readlane_b32'],'salu',sum(v for k,v in c.items() if k.startswith('s_')))
