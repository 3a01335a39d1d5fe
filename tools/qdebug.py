"""Diagnostic: one ray through the production and the reference-order traversal, with the nodes each visits."""
import sys; sys.path.insert(0,'cuda-pathtracer_amd/python'); sys.path.insert(0,'tests')
import numpy as np, ptamd, util, orc
from ptamd import scenes
from util import bits
b=util.cornell('sky')
P=scenes.cornell_params(ptamd.default_params(37,29))
def run(mode,pool=None,f0=1,f1=6):
    r=b.renderer(P); r.set_mode(mode); r.set_max_bounces(12)
    if pool: r.set_pool(*pool)
    r.render(P,f0,f1); a=r.read_accum(); r.close(); return a
ref=run(1)
tot=0
for rep in range(6):
    for pool in [(64,1),(320,3),(320,2),(512,5),(1<<14,64)]:
        a=run(0,pool)
        bad=np.argwhere(bits(a)!=bits(ref)); tot+=len(bad)
        if len(bad): print(rep,pool,'diff',len(bad))
print('total diffs',tot)
