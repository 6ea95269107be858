"""Instruction histogram of the hottest loop of a kernel in a `hipcc --cuda-device-only -S` listing.
usage: isa_hist.py <file.s> [kernel name fragment] [top N]   (the bootstrap kernels are VALU-issue bound)"""
import re,collections,sys
lines=open(sys.argv[1]).read().split('\n')
pat=sys.argv[2] if len(sys.argv)>2 else 'pbs_kernel'
start=[i for i,l in enumerate(lines) if re.match(r'_Z\w*'+pat+r'\w*:',l)]
s=start[0]
end=next(i for i in range(s,len(lines)) if lines[i].strip().startswith('s_endpgm'))
body=lines[s:end]
labels={}
for i,l in enumerate(body):
    m=re.match(r'(\.LBB\d+_\d+):',l)
    if m: labels[m.group(1)]=i
loops=[]
for i,l in enumerate(body):
    m=re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)',l) or re.search(r's_branch (\.LBB\d+_\d+)',l)
    if m and m.group(1) in labels and labels[m.group(1)]<i: loops.append((i-labels[m.group(1)],labels[m.group(1)],i))
loops.sort(reverse=True); print(loops[:3])
n,a,b=loops[0]
cnt=collections.Counter()
for l in body[a:b]:
    l=l.strip()
    if not l or l.startswith(';') or l.startswith('.') : continue
    cnt[l.split()[0]]+=1
tot=sum(cnt.values()); print('total',tot)
groups=collections.Counter()
for k,v in cnt.items():
    if 'f64' in k: g='f64'
    elif k.startswith('ds_'): g='lds'
    elif k.startswith('global_') or k.startswith('buffer_') or k.startswith('scratch_') or k.startswith('flat_'): g='mem:'+k.split('_')[0]
    elif k.startswith('s_'): g='scalar'
    elif k.startswith('v_'): g='valu-other'
    else: g='other'
    groups[g]+=v
print(dict(groups))
for k,v in cnt.most_common(int(sys.argv[3]) if len(sys.argv)>3 else 40): print(f'{k:28s}{v}')
