"""How much of what ordering a bin's members by norm can give do the shells already give?  numpy only: for far query groups
(128 queries sharing a nearest centre other than the bin's) and the exact sweep-1 threshold, the share of a bin's tiles
behind the last needed one, for several shell definitions and for an exact sort by norm.
Result (120k x 140 x 32, five coverage columns): 16 shells 63.7 %, 32 shells 66.3 %, quantile shells 66.4-67.0 %, exact sort
67.5 %: the norm bound is used up, more skipping needs a directional bound (tile centres)."""
import sys, numpy as np
sys.path.insert(0,'/root/repo')
import chbin_amd
from chbin_amd import synth
N,D,B,m=120000,140,32,5
X,initial,true=synth.make_synthetic(N,D,B,S=5,seed=0)
rng=np.random.default_rng(0)
cent=np.stack([X[true==c].mean(0) for c in range(B)])
near=np.argmin(np.stack([((X-cent[c])**2).sum(1) for c in range(B)],1),1)
bins=rng.choice(B,6,replace=False)
pairs={c:rng.choice([x for x in range(B) if x!=c],4,replace=False) for c in bins}
def run(kind,nsh):
    skipped=tot=0
    for c in bins:
        memb=X[true==c]; z=memb-cent[c]; nr=np.linalg.norm(z,axis=1)
        seeds=nr[:60]
        if kind=='seed': sh=np.minimum((nsh*nr/(1.25*seeds.max())).astype(int),nsh-1)
        elif kind=='true': sh=np.minimum((nsh*nr/(1.0001*nr.max())).astype(int),nsh-1)
        elif kind=='range':   # uniform over the seeds' range, widened
            lo,hi=0.8*seeds.min(),1.25*seeds.max(); sh=np.clip(((nr-lo)/(hi-lo)*nsh).astype(int),0,nsh-1)
        elif kind=='q':      # quantile cuts from the seeds
            cuts=np.quantile(seeds,np.arange(1,nsh)/nsh); sh=np.searchsorted(cuts,nr)
        elif kind=='qall':   # quantile cuts from all members
            cuts=np.quantile(nr,np.arange(1,nsh)/nsh); sh=np.searchsorted(cuts,nr)
        if kind=='exact': order=np.argsort(-nr)
        else: order=np.argsort(-sh,kind='stable')
        nr_o=nr[order]; T=(len(nr_o)+31)//32
        tmax=np.array([nr_o[t*32:(t+1)*32].max() for t in range(T)])
        suf=np.maximum.accumulate(tmax[::-1])[::-1]
        rows=memb[order]
        for cc in pairs[c]:
            qq=np.flatnonzero(near==cc)[:128]
            if len(qq)<128: continue
            d=np.linalg.norm(rows[None,:,:]-X[qq][:,None,:],axis=2)
            tau=np.sort(d,axis=1)[:,m-1]
            zq=np.linalg.norm(X[qq]-cent[c],axis=1)
            need=(zq[:,None]-suf[None,:])<=tau[:,None]
            anyneed=need.any(0)
            K=max(3,(np.flatnonzero(anyneed).max()+1) if anyneed.any() else 3); K=min(T,K+2)
            skipped+=T-K; tot+=T
    return round(skipped/tot,3)
for kind,nsh in (('seed',16),('seed',32),('true',32),('range',32),('q',16),('q',32),('qall',32),('qall',64),('exact',0)):
    print(kind,nsh,run(kind,nsh),flush=True)
