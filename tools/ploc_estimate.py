#!/usr/bin/env python3
"""Would a better builder help the LBVH on BASELINE configs[4]'s soup?  (CPU, numpy; an estimate made before building anything.)

Surface-area cost of the binary hierarchy -- sum over internal nodes of area(node) / area(root), the expected number of
internal nodes a random ray enters -- for (a) the Morton radix tree the library builds (csrc/pt_bvh.hip) and (b) PLOC
(Meister & Bittner 2018: mutual nearest neighbours by merged surface area within a window of the Morton order), on the ~51 000
triangles of the soup whose centres lie in a cube of side 2 (the soup's own density and triangle sizes).
usage: python tools/ploc_estimate.py [side]      -> profiles/r03/lbvh_steps.txt"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oclpathtracer_amd import scene
t,m = scene.make_soup(1_000_000)
p1=t['p1'][36:,:3]; p2=t['p2'][36:,:3]; p3=t['p3'][36:,:3]
c=(np.minimum(np.minimum(p1,p2),p3)+np.maximum(np.maximum(p1,p2),p3))*0.5
side=float(sys.argv[1]) if len(sys.argv)>1 else 2.0
sel=np.all((c>np.array([-1.0,1.0,-4.0]))&(c<np.array([-1.0,1.0,-4.0])+side),axis=1)
p1,p2,p3=p1[sel],p2[sel],p3[sel]
lo=np.minimum(np.minimum(p1,p2),p3).astype(np.float64); hi=np.maximum(np.maximum(p1,p2),p3).astype(np.float64)
n=len(lo); print("triangles",n)
cen=(lo+hi)/2
smin=cen.min(0); sext=(cen.max(0)-smin).max()
q=np.clip(((cen-smin)/sext*1024).astype(np.int64),0,1023)
def expand(v):
    v=(v|(v<<16))&0x030000FF; v=(v|(v<<8))&0x0300F00F; v=(v|(v<<4))&0x030C30C3; v=(v|(v<<2))&0x09249249; return v
mort=(expand(q[:,0])<<2)|(expand(q[:,1])<<1)|expand(q[:,2])
order=np.argsort(mort,kind='stable'); mort=mort[order]; lo=lo[order]; hi=hi[order]
def area(l,h):
    d=np.maximum(h-l,0); return 2*(d[...,0]*d[...,1]+d[...,1]*d[...,2]+d[...,0]*d[...,2])
rootA=area(lo.min(0),hi.max(0))
# ---- LBVH (recursive split on highest differing bit; ties -> middle) ----
sys.setrecursionlimit(100000)
keys=(mort.astype(np.uint64)<<np.uint64(32))|np.arange(n,dtype=np.uint64)
def lbvh_cost():
    tot=0.0; stack=[(0,n)]
    # iterative: compute boxes via prefix? just do recursion returning boxes
    def rec(a,b):
        nonlocal tot
        if b-a==1: return lo[a],hi[a]
        x=int(keys[a])^int(keys[b-1]); bit=x.bit_length()-1
        mask=(int(keys[a])>>bit)
        # first index where bit set: keys sorted => binary search
        pref=(int(keys[a])>>bit)<<bit | (1<<bit)
        s=int(np.searchsorted(keys[a:b],np.uint64((int(keys[a])>>(bit+1)<<(bit+1))|(1<<bit))))+a
        l1,h1=rec(a,s); l2,h2=rec(s,b)
        l=np.minimum(l1,l2); h=np.maximum(h1,h2)
        tot+=area(l,h)
        return l,h
    rec(0,n); return tot/rootA
t0=time.time(); cl=lbvh_cost(); print("LBVH  internal SAH sum %.2f (%.1fs)"%(cl,time.time()-t0))
# ---- PLOC ----
def ploc(r):
    L=lo.copy(); H=hi.copy(); tot=0.0; it=0
    while len(L)>1:
        k=len(L); best=np.full(k,np.inf); nn=np.full(k,-1)
        for off in range(1,r+1):
            if off>=k: break
            a=area(np.minimum(L[:-off],L[off:]),np.maximum(H[:-off],H[off:]))
            # pair (i, i+off)
            idx=np.arange(k-off)
            upd=a<best[idx]; best[idx[upd]]=a[upd]; nn[idx[upd]]=idx[upd]+off
            idx2=idx+off
            upd=a<best[idx2]; best[idx2[upd]]=a[upd]; nn[idx2[upd]]=idx[upd]
        i=np.arange(k)
        mutual=(nn[nn]==i)&(i<nn)
        tot+=best[mutual].sum()
        j=nn[mutual]; ii=i[mutual]
        L[ii]=np.minimum(L[ii],L[j]); H[ii]=np.maximum(H[ii],H[j])
        keep=np.ones(k,bool); keep[j]=False
        L=L[keep]; H=H[keep]; it+=1
    return tot/rootA,it
for r in (8,16,32):
    t0=time.time(); cp,it=ploc(r); print("PLOC r=%d internal SAH sum %.2f in %d iterations (%.1fs)  ratio to LBVH %.3f"%(r,cp,it,time.time()-t0,cp/cl))
