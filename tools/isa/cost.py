"""VALU issue-cycle estimate per basic block of one kernel in a `hipcc -S` listing, with the two issue classes measured
in profiles/r01_valu_rate.txt (2.7 / 4.5 cycles per wave64 instruction).  Multiply by per-block trip counts of the workload
to get cycles per wave.
usage: python tools/isa/cost.py kernels.s <mangled kernel name prefix> [min VALU per block to print]"""
import re,sys
# measured issue costs (profiles/r01_valu_rate.txt), cycles @2.4GHz per wave64 instr
FAST={'v_add_u32','v_sub_u32','v_subrev_u32','v_and_b32','v_or_b32','v_xor_b32','v_lshrrev_b32','v_mov_b32','v_add_u16','v_sub_u16','v_min_i16','v_max_i16','v_min_u16','v_max_u16','v_bitop3_b32','v_mul_f32','v_fma_f32','v_fmac_f32','v_add_f32','v_sub_f32','v_not_b32','v_ashrrev_i32','v_bitop3_b16','v_mul_lo_u16','v_lshrrev_b16','v_accvgpr_write_b32','v_accvgpr_read_b32'}
def cost(op):
    base=op.replace('_e32','').replace('_e64','').replace('_sdwa','').replace('_dpp','')
    if op.endswith('_dpp') or op.endswith('_sdwa'): return 4.5
    if base in FAST: return 2.7
    if base.startswith('v_pk_add_f32') or base=='v_max3_i16': return 8.5
    if 'f64' in base: return 16
    return 4.5
name=sys.argv[2]
lines=open(sys.argv[1]).read().split('\n')
i0=[i for i,l in enumerate(lines) if l.startswith(name) and '@' in l][0]
i1=next(i for i in range(i0+1,len(lines)) if lines[i].startswith('.Lfunc_end'))
blocks=[];cur=None
from collections import Counter
tot=Counter()
for ln in lines[i0:i1]:
    s=ln.strip()
    if re.match(r'^\.LBB\d+_\d+:',s) or s.startswith('_Z'):
        cur=[s.split(':')[0][-12:],0,0.0,0,0,Counter()];blocks.append(cur);continue
    if cur is None or not s or s.startswith(';') or s.startswith('.'):continue
    op=s.split()[0]
    if op.startswith('v_'):
        cur[1]+=1;cur[2]+=cost(op);cur[5][op]+=1
    elif op.startswith('s_'):cur[3]+=1
    elif op.startswith('ds_'):cur[4]+=1
for b in blocks:
    if b[1]>=int(sys.argv[3]) if len(sys.argv)>3 else 8:
        slow=[(k,v) for k,v in b[5].items() if cost(k)>3]
        print("%-12s V%4d cyc%6.0f S%3d DS%3d  slow: %s"%(b[0],b[1],b[2],b[3],b[4],' '.join('%s:%d'%(k.replace('_e32','').replace('_e64',''),v) for k,v in sorted(slow,key=lambda x:-x[1]))))
