"""Lists the kernels of one translation unit whose INNERMOST multiply loops contain scratch (spill) instructions:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1048576 --cuda-device-only -S -o tu.s csrc/<tu>.hip
    python tools/isa_scratch_scan.py tu.s      ->  kernel name [(loop instructions, v_mad_u64_u32, scratch ops), ...]"""
import re, sys, subprocess
from collections import Counter
path=sys.argv[1]
lines=open(path).read().split('\n')
# split by function
funcs=[]; cur=None
for i,l in enumerate(lines):
    m=re.match(r'^(_ZN3pai\w+):\s+; @',l)
    if m: cur=[m.group(1), i, None]; funcs.append(cur)
    if l.startswith('.Lfunc_end') and cur and cur[2] is None: cur[2]=i
for name,a0,b0 in funcs:
    if b0 is None: continue
    body=lines[a0:b0]
    labels={}
    for i,l in enumerate(body):
        m=re.match(r'^(\.LBB\d+_\d+):',l)
        if m: labels[m.group(1)]=i
    loops=[]
    for i,l in enumerate(body):
        m=re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)',l) or re.search(r's_branch\s+(\.LBB\d+_\d+)',l)
        if m and m.group(1) in labels and labels[m.group(1)]<i: loops.append((labels[m.group(1)],i))
    # innermost loops with mads
    bad=[]
    for a,b in loops:
        if any(a<=a2 and b2<=b and (a,b)!=(a2,b2) for a2,b2 in loops): 
            inner=False
        else: inner=True
        ins=[x.strip().split()[0] for x in body[a:b+1] if x.startswith('\t') and not x.strip().startswith(('.',';'))]
        mad=sum(1 for x in ins if x.startswith('v_mad_u64')); sc=sum(1 for x in ins if x.startswith('scratch'))
        if inner and mad>50 and sc>0: bad.append((len(ins),mad,sc))
    if bad:
        dn=subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip()[:110]
        print(dn, bad)
