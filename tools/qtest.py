import sys; sys.path.insert(0,'cuda-pathtracer_amd/python'); sys.path.insert(0,'tests')
import numpy as np, ptamd, util, orc
from ptamd import scenes
from util import bits
for name,b,ext,yoff in [('cornell',util.cornell('sky'),400.0,100.0),('atrium',util.atrium(detail=0.35),1500.0,300.0)]:
    P=ptamd.default_params(64,64)
    r=b.renderer(P)
    rays=scenes.random_rays(2000000,seed=11,extent=ext); rays['o'][:,1]+=yoff
    # axis-parallel and grid-aligned rays too
    rays['d'][:1000,2]=0; rays['d'][1000:2000,0]=0; rays['d'][2000:3000,1]=0
    n=np.linalg.norm(rays['d'][:3000],axis=1,keepdims=True); rays['d'][:3000]/=n
    exact=r.trace_closest(rays); fast=r.trace_closest_fast(rays)
    bad=np.nonzero((exact['tri']!=fast['tri'])|(bits(exact['t'])!=bits(fast['t'])))[0]
    print(name,'hits',(exact['tri']!=0xffffffff).sum(),'mismatches',len(bad))
    for i in bad[:5]: print(i,rays[i],exact[i],fast[i])
    r.close()
