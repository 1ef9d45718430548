#!/usr/bin/env python3
"""What would a HIERARCHICAL replication of the top of the elimination tree buy a sharded solve? (host-only estimate, round 6)
VERDICT r05 item 6(i): instead of replicating every top front on all N ranks, a separator is reduced and factorised only inside the group of ranks
that hold the subtrees below it (rank pairs / quads / all). This tool deals the ranks down the chosen tree recursively (ranks proportional to subtree
flops, LPT where a node has more children than ranks) and prints, per world size, every rank's share of the factorisation flops (its subtrees plus the
fronts of its whole root path) and the bytes of top fronts it would exchange per linear solve.
Result (profiles/r06_shard_hier_stats.txt): the flops sit in the fronts NEAR THE TOP (large separators with 2 000-8 000-unknown borders), and a rank's
root path carries them whatever the grouping — 12-agent map at 8 ranks: busiest rank 57 % of all flops, 1.4 GB exchanged per solve; 5-agent map: 46 %.
The <= 25 % the verdict asks for is out of reach of any replication scheme: it needs the big fronts themselves factorised distributed.
usage: python tools/shard_hier_stats.py a12 | mh12345"""
import sys, ctypes as C, numpy as np
sys.path.insert(0,'/root/repo')
from covins_amd import backend, mapdata, synth
name=sys.argv[1]
m = synth.make_map(synth.config_named(name))
prob,_ = mapdata.flatten_gba(m, visual_only=False, loop_loss=True)
lib=backend.lib(); h=C.c_void_p(); s=prob.as_struct(); opt=backend.default_options()
rc=lib.covgpu_nd_plan_create(C.byref(opt),C.byref(s),0,C.byref(h)); assert rc==0
info=(C.c_int64*16)(); lib.covgpu_nd_plan_info(h,info); nn=info[0]
parent=np.zeros(nn,np.int32); level=np.zeros(nn,np.int32); optr=np.zeros(nn+1,np.int32); sptr=np.zeros(nn+1,np.int32)
ov=np.zeros(max(info[3],1),np.int32); sv=np.zeros(max(info[4],1),np.int32)
ip=lambda a:a.ctypes.data_as(C.POINTER(C.c_int32))
lib.covgpu_nd_plan_arrays(h,ip(parent),ip(level),ip(optr),ip(ov),ip(sptr),ip(sv))
dim=lambda vs: sum(9 if v&1 else 6 for v in vs)
own=np.array([dim(ov[optr[n]:optr[n+1]]) for n in range(nn)],float); st=np.array([dim(sv[sptr[n]:sptr[n+1]]) for n in range(nn)],float)
w=own**3/3+own**2*st+own*st**2
child=[[] for _ in range(nn)]
for n in range(nn):
    if parent[n]>=0: child[parent[n]].append(n)
sub=w.copy()
for n in range(nn-1,-1,-1):
    if parent[n]>=0: sub[parent[n]]+=sub[n]
tot=w.sum()
for world in (2,4,8):
    load=np.zeros(world); exch=np.zeros(world)
    def assign(n,lo,hi):
        if hi-lo==1:
            load[lo]+=sub[n]; return
        load[lo:hi]+=w[n]; exch[lo:hi]+=0.5*(own[n]+st[n])**2*8
        ch=sorted(child[n],key=lambda c:-sub[c])
        r=hi-lo
        if len(ch)==0: return
        if len(ch)>=r:
            bins=[0.0]*r; binc=[[] for _ in range(r)]
            for c in ch:
                b=int(np.argmin(bins)); bins[b]+=sub[c]; binc[b].append(c)
            for b in range(r):
                for c in binc[b]: assign(c,lo+b,lo+b+1)
            return
        # fewer children than ranks: ranks proportional to weight, at least one each
        cnt=[1]*len(ch); rem=r-len(ch)
        for _ in range(rem):
            k=int(np.argmax([sub[c]/cnt[i] for i,c in enumerate(ch)])); cnt[k]+=1
        a=lo
        for c,k in zip(ch,cnt): assign(c,a,a+k); a+=k
    roots=[n for n in range(nn) if parent[n]<0]
    assert len(roots)==1
    assign(roots[0],0,world)
    print(name,'world',world,'busiest share %.3f'%(load.max()/tot),'mean %.3f'%(load.mean()/tot),'shares',np.round(load/tot,3),'exchange MB per rank',np.round(exch/1e6,1))
