"""CPU statistics behind DESIGN.md 7.4 / profiles/r04_fast_ab.txt (VERDICT r03 #1a "measure first"): on the bench frames, how many
pixels pass K-FAST's phase-A test (6-bit SWAR compass test) against the exact compass test, stronger necessary tests (3 / 4 / 8
opposite pairs) and the true FAST-9 corner count; survivors and corners per cell; phase B's lane occupancy with per-cell queues and
with queues pooled over runs of cells.  Uses the oracle for the pyramid levels (tools only; nothing here is product code).
usage: python tools/fast_pass_rates.py"""

# ---- part 1: pass rates of the necessary tests
import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import orb_oracle_py as O
from orb_slam3_detailed_comments_kor_amd import synth
from collections import Counter
tot = Counter()
offs=[(3,0),(3,1),(2,2),(1,3),(0,3),(-1,3),(-2,2),(-3,1),(-3,0),(-3,-1),(-2,-2),(-1,-3),(0,-3),(1,-3),(2,-2),(3,-1)]
for seed in range(1234, 1238):
    img = synth.make_frame(480, 752, seed)
    ex = O.Extractor(1000,1.2,8,20,7)
    ex.extract(img,(0,0))
    for lvl in range(8):
        L = ex.level(lvl).astype(np.int32)
        h,w = L.shape
        for t in (20,):
            c = L[3:-3,3:-3]
            ring=np.stack([L[3+dy:h-3+dy,3+dx:w-3+dx] for dy,dx in offs],0)
            tb=(t+1)>>2; cq=c>>2
            for name,B,Dk in (("x",ring>c+t,ring<c-t),("q",(ring>>2)>=cq+tb,(ring>>2)<=cq-tb)):
                def pairs(M,ks):
                    acc=np.ones(M.shape[1:],bool)
                    for k in ks: acc&=(M[k]|M[k+8])
                    return acc
                for nm,ks in (("2",(0,4)),("4",(0,2,4,6)),("8",range(8)),("4b",(0,4,1,5)),("3",(0,4,2))):
                    tot[name+nm]+= (pairs(B,ks)|pairs(Dk,ks)).sum()
            B=ring>c+t; Dk=ring<c-t
            def arc9(M):
                M2=np.concatenate([M,M[:8]],0)
                acc=np.zeros(M.shape[1:],bool)
                for k in range(16): acc|=M2[k:k+9].all(0)
                return acc
            tot['corner']+=(arc9(B)|arc9(Dk)).sum(); tot['px']+=c.size
print({k:round(v/tot['px'],4) for k,v in sorted(tot.items())})

# ---- part 2: per-cell statistics and pooled occupancy
import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import orb_oracle_py as O
from orb_slam3_detailed_comments_kor_amd import synth
offs=[(3,0),(3,1),(2,2),(1,3),(0,3),(-1,3),(-2,2),(-3,1),(-3,0),(-3,-1),(-2,-2),(-1,-3),(0,-3),(1,-3),(2,-2),(3,-1)]
nqs=[]; ncs=[]; rows=[]
for seed in range(1234, 1238):
    img = synth.make_frame(480, 752, seed)
    ex = O.Extractor(1000,1.2,8,20,7); ex.extract(img,(0,0))
    for lvl in range(8):
        L = ex.level(lvl)[19:-19,19:-19].astype(np.int32)
        h,w = L.shape; t=20
        c = L[3:-3,3:-3]; ring=np.stack([L[3+dy:h-3+dy,3+dx:w-3+dx] for dy,dx in offs],0)
        tb=(t+1)>>2; cq=c>>2
        B=(ring>>2)>=cq+tb; D=(ring>>2)<=cq-tb
        a6=((B[0]|B[8])&(B[4]|B[12]))|((D[0]|D[8])&(D[4]|D[12]))
        Bx=ring>c+t; Dx=ring<c-t
        def arc9(M):
            M2=np.concatenate([M,M[:8]],0); acc=np.zeros(M.shape[1:],bool)
            for k in range(16): acc|=M2[k:k+9].all(0)
            return acc
        cor=arc9(Bx)|arc9(Dx)
        # cells: zone coords start at x=16 (minBorder) ; a6 index (y-3,x-3)
        minB=16; maxBX=w-16; maxBY=h-16
        W=35; width=maxBX-minB; height=maxBY-minB
        nC=width//W; nR=height//W; wC=int(np.ceil(width/nC)); hC=int(np.ceil(height/nR))
        for i in range(nR):
            iniY=minB+i*hC; maxY=min(iniY+hC+6,maxBY)
            if iniY>=maxBY-3: continue
            rowq=[]
            for j in range(nC):
                iniX=minB+j*wC; maxX=min(iniX+wC+6,maxBX)
                if iniX>=maxBX-6: continue
                z=a6[iniY:maxY-6, iniX:maxX-6]; zc=cor[iniY:maxY-6, iniX:maxX-6]
                nqs.append(z.sum()); ncs.append(zc.sum()); rowq.append(z.sum())
            rows.append(rowq)
nqs=np.array(nqs); ncs=np.array(ncs)
print("cells",len(nqs),"mean nq",nqs.mean(),"mean corners",ncs.mean(), "empty-at-ini cells", (ncs==0).mean())
wr=np.ceil(nqs/64); print("phase B wave-rounds/cell",wr.mean(),"lane occupancy",nqs.sum()/(64*wr.sum()))
wrc=np.ceil(ncs/64); print("if only corners went on: rounds",wrc.mean(),"occ",ncs.sum()/(64*np.maximum(wrc,0).sum()))
for G in (2,3,4,5):
    tot=0; used=0
    for r in rows:
        for k in range(0,len(r),G):
            q=sum(r[k:k+G]); tot+=np.ceil(q/64)*64; used+=q
    print("pooled",G,"cells: occupancy",used/tot, "wave-rounds per cell", tot/64/len(nqs))
print(np.percentile(nqs,[10,25,50,75,90,99]))
