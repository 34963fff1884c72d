"""Off-line reading of dump_lists.py's file (a fixed-merger failing build: haz_top_ns): every lane of every dumped tile is merged
on the host with the swap chain in the kernel's order (slices ascending, the merging slice's own list first in its registers) and
the swaps taken by the TIE rule are recorded as (slice, entry index in that slice's list, level); then failing lanes against
those events.  Result (3 runs, 98 failing lanes): every failing lane has a tie swap during the insertion of ENTRY 0 of another
slice's list, none of the 22 116 lanes without one fails -- profiles/r06_spill_repro_5.txt.
usage: python analyse_dump.py DUMP.pt"""
import sys
import torch, collections
d=torch.load(sys.argv[1])
K=8
def push(lst, c):
    """swap chain on a sorted list of (z,id); returns list of events (level, kind) where kind = 'lt' or 'tie'"""
    ev=[]
    for j in range(K):
        if j>=len(lst): 
            lst.append(c); c=None; break
        s=lst[j]
        if c[0] < s[0]:
            lst[j]=c; c=s; ev.append((j,'lt'))
        elif c[0]==s[0] and c[1]<s[1]:
            lst[j]=c; c=s; ev.append((j,'tie'))
    return ev
stat=collections.Counter()
detail=collections.Counter()
for r,run in enumerate(d['runs']):
    for t,info in run['tiles'].items():
        L=info['lists']; ns=info['slices']; got=info['got']; want=info['want']
        for lane in range(256):
            lists=[]
            for s in range(ns):
                z=L[s,0,:,lane]; ids=L[s,2,:,lane].view(torch.int32)
                lists.append([(float(z[j]),int(ids[j])) for j in range(8) if float(z[j])<3e38])
            own=list(lists[ns-1])
            while len(own)<K: own.append((3.4028234663852886e38,0x7fffffff))
            tie_at=[]   # (slice, entry, level)
            for s in range(ns-1):
                for e,c in enumerate(lists[s]):
                    ev=push(own,c); own=own[:K]
                    for (lvl,kind) in ev:
                        if kind=='tie': tie_at.append((s,e,lvl))
            bad=bool((got[lane]!=want[lane]).any())
            e0=any(e==0 for (_,e,_) in tie_at)
            stat[(bad, 'tie-swap at entry 0' if e0 else ('tie-swap elsewhere' if tie_at else 'no tie-swap'))]+=1
            if tie_at:
                for (s,e,lvl) in tie_at: detail[(bad,e,lvl)]+=1
for k,v in sorted(stat.items()): print(k,v)
print("by (bad, entry, level):")
for k,v in sorted(detail.items()): print("  ",k,v)
