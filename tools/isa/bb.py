"""Per-basic-block instruction counts of one kernel in a `hipcc -S` listing.
usage: python tools/isa/bb.py kernels.s <mangled kernel name prefix>
prints VALU / SALU / LDS / VMEM / barrier counts and branch targets per block, then the register and LDS totals."""
import re,sys
name=sys.argv[2]
lines=open(sys.argv[1]).read().split('\n')
start=[i for i,l in enumerate(lines) if l.startswith(name) and l.rstrip().endswith(name+':') or (l.startswith(name) and ':' in l and '@' in l)]
i0=start[0]
i1=next(i for i in range(i0+1,len(lines)) if lines[i].startswith('.Lfunc_end'))
blocks=[];cur=None
for ln in lines[i0:i1]:
    s=ln.strip()
    if re.match(r'^\.LBB\d+_\d+:',s) or s.startswith('_Z'):
        cur=[s.split(':')[0][-12:],0,0,0,0,[],0];blocks.append(cur);continue
    if cur is None or not s or s.startswith(';') or s.startswith('.'):continue
    op=s.split()[0]
    if op.startswith('v_'):cur[1]+=1
    elif op.startswith('s_'):
        cur[2]+=1
        if 'branch' in op: cur[5].append(s.split()[-1])
        if op=='s_barrier':cur[6]+=1
    elif op.startswith('ds_'):cur[3]+=1
    elif op.startswith('global_') or op.startswith('flat_') or op.startswith('buffer_'):cur[4]+=1
for b in blocks:
    print("%-12s V%4d S%4d DS%3d G%2d bar%d  -> %s"%(b[0],b[1],b[2],b[3],b[4],b[6],','.join(b[5])))
print("total V",sum(b[1] for b in blocks),"S",sum(b[2] for b in blocks))
for l in lines[i1:i1+60]:
    if 'vgpr_count' in l or 'sgpr_count' in l or 'NumVgprs' in l or 'Occupancy' in l or 'LDSByteSize' in l or 'ScratchSize' in l: print(l.strip())
