"""Production traversal kernel (both node widths) against the reference-order kernel on many random rays."""
import sys; sys.path.insert(0,'cuda-pathtracer_amd/python'); sys.path.insert(0,'tests')
import numpy as np, ptamd, util
from ptamd import scenes
from util import bits
tot = 0
for name, b, ext in (('atrium', util.atrium(detail=0.5), 1400.0), ('soup', util.soup(200000, seed=3, extent=400.0, edge=8.0), 420.0), ('cornell', util.cornell('sky'), 300.0)):
    P = ptamd.default_params(64, 64)
    r = b.renderer(P)
    rays = scenes.random_rays(2000000, seed=9, extent=ext)
    slow = r.trace_closest(rays)
    for width in (4, 2):
        r.set_node_width(width)
        fast = r.trace_closest_fast(rays)
        bad = int((fast['tri'] != slow['tri']).sum() + (bits(fast['t']) != bits(slow['t'])).sum())
        tot += bad
        print(name, 'width', width, 'hits', int((slow['tri'] != 0xFFFFFFFF).sum()), 'mismatches', bad, 'ms', r.counters()['last_ms'])
    r.close()
print('total mismatches', tot)
