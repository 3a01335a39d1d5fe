"""GPU-built LBVH against the reference tree: closest hits must be identical, build time and traversal rate reported."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'cuda-pathtracer_amd', 'python')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ptamd, util
from ptamd import scenes, lib
from util import bits
cases = [('cornell', util.cornell('sky'), 300.0, 200000), ('atrium', util.atrium(detail=0.5), 1400.0, 1000000),
         ('soup200k', util.soup(200000, seed=3, extent=400.0, edge=8.0), 420.0, 1000000)]
one = util.soup(1, seed=11, extent=50.0, edge=30.0)
cases.append(('single', one, 80.0, 20000))
two = util.soup(2, seed=12, extent=50.0, edge=30.0)
cases.append(('pair', two, 80.0, 20000))
dup = util.soup(64, seed=13, extent=20.0, edge=30.0); dup.tris[:] = dup.tris[0]; dup2 = util.Bundle(dup.tris.copy(), dup.mats)
cases.append(('64 copies of one triangle', dup2, 80.0, 20000))
for name, b, ext, nr in cases:
    P = ptamd.default_params(64, 64)
    rays = scenes.random_rays(nr, seed=9, extent=ext)
    r = b.renderer(P)
    want = r.trace_closest(rays)
    r.trace_closest_fast(rays); t_ref = None
    r.trace_closest_fast(rays); ms_ref = r.counters()['last_ms']
    t0 = time.perf_counter()
    r.upload_scene(b.tris, b.mats, None, None, flavour=lib.PT_BVH_GPU_LBVH)
    t1 = time.perf_counter()
    got = r.trace_closest(rays)
    fast = r.trace_closest_fast(rays); fast = r.trace_closest_fast(rays); ms = r.counters()['last_ms']
    ok = np.array_equal(got['tri'], want['tri']) and np.array_equal(bits(got['t']), bits(want['t'])) and np.array_equal(fast['tri'], want['tri']) and np.array_equal(bits(fast['t']), bits(want['t']))
    print('%-28s %8d tris: upload+GPU build+flatten %.3f s, exact %s, hits %d, production traversal %.2f ms (reference tree %.2f ms)'
          % (name, len(b.tris), t1 - t0, ok, int((want['tri'] != 0xFFFFFFFF).sum()), ms, ms_ref), flush=True)
    r.close()
