"""CPU emulation of k_fps_lazy's protocol (csrc/fps.hip): how many samples does one exchange of the workgroups' T largest keys
decide?  500 k points on a sphere, 5 000 samples: nb = 128, T = 4 -> 145 exchanges, 34.5 samples each on average (median 30, 1..96);
T = 8 -> 80 exchanges; nb = 64 -> 193.  usage: python tools/diag/fps_lazy_emul.py [workgroups] [T]"""
import numpy as np, sys
P=500000; NS=5000; nb=int(sys.argv[1]) if len(sys.argv)>1 else 128; T=int(sys.argv[2]) if len(sys.argv)>2 else 4
rng=np.random.default_rng(3)
p=rng.standard_normal((P,3)).astype(np.float32); p/=np.linalg.norm(p,axis=1,keepdims=True)
wg=(np.arange(P)//1024)%nb
order=np.argsort(wg,kind='stable'); bounds=np.searchsorted(wg[order],np.arange(nb+1))
mind=np.full(P,np.inf,np.float32)
def apply(c):
    global mind
    d=((p-p[c])**2).sum(1).astype(np.float32); mind=np.minimum(mind,d)
apply(0); done=1; rounds=[]; 
while done<NS:
    # lists
    ek=[];ei=[]
    for j in range(nb):
        idx=order[bounds[j]:bounds[j+1]]
        if len(idx)==0: continue
        k=min(T,len(idx))
        top=idx[np.argpartition(-mind[idx],k-1)[:k]]
        top=top[np.lexsort((top,-mind[top]))]
        ek.append(mind[top].copy()); ei.append(top)
    B=[e[-1] if len(e)==T else -1 for e in ek]
    m=0
    while done+m<NS and m<10000:
        best=[e.max() for e in ek]
        c=max(best); blk=max([B[j] for j in range(len(ek)) if best[j]<B[j]],default=-1)
        if not c>blk: break
        j=int(np.argmax(best)); q=int(np.argmax(ek[j])); w=ei[j][q]
        for jj in range(len(ek)):
            d=((p[ei[jj]]-p[w])**2).sum(1).astype(np.float32); ek[jj]=np.minimum(ek[jj],d)
        apply(w); m+=1
    rounds.append(m); done+=m
r=np.array(rounds); print("nb",nb,"T",T,"rounds",len(r),"mean m %.1f"%r.mean(),"median",np.median(r),"min",r.min(),"max",r.max(), "first rounds",r[:12])
