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
# axis-parallel and nearly parallel rays (zero and denormal direction components)
b = util.atrium(detail=0.5)
r = b.renderer(ptamd.default_params(64, 64))
rays = scenes.random_rays(600000, seed=10, extent=1400.0)
k = np.arange(len(rays))
for a in range(3):
    rays['d'][k % 7 == a, a] = 0.0
    rays['d'][k % 7 == a + 3, a] = 1e-41
rays['d'][k % 7 == 6] = [0.0, -1.0, -0.0]
n = np.linalg.norm(rays['d'], axis=1, keepdims=True); rays['d'] /= np.maximum(n, 1e-30)
slow = r.trace_closest(rays); fast = r.trace_closest_fast(rays)
bad = int((fast['tri'] != slow['tri']).sum() + (bits(fast['t']) != bits(slow['t'])).sum())
print('axis-parallel rays: hits', int((slow['tri'] != 0xFFFFFFFF).sum()), 'mismatches', bad, 'ms', r.counters()['last_ms'])
