"""Traversal cost of the device-built tree against the host-built SAH tree on the bench atrium: node visits and
triangle tests per ray of the production walk, box tests of the reference-order walk, render time of 16 spp at 1080p."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'cuda-pathtracer_amd', 'python')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ptamd, util
from ptamd import scenes, lib
b = util.atrium()
P = scenes.atrium_params(ptamd.default_params(1920, 1080))
r = b.renderer(P)
for name, flavour in (('host SAH (reference tree)', None), ('device LBVH', lib.PT_BVH_GPU_LBVH), ('device PLOC', lib.PT_BVH_GPU_PLOC)):
    if flavour is not None:
        t0 = time.perf_counter(); r.upload_scene(b.tris, b.mats, None, None, flavour=flavour); r.sync(); tb = time.perf_counter() - t0
    else:
        tb = float('nan')
    out = {}
    for mode in (1, 2):
        r.enable_counters(mode); r.reset_counters(); r.clear_accum(); r.render(P, 0, 4); out[mode] = r.counters()
    r.enable_counters(0)
    r.clear_accum(); r.render(P, 0, 16); r.sync()
    r.clear_accum(); t0 = time.perf_counter(); r.render(P, 0, 16); r.sync(); dt = time.perf_counter() - t0
    c1, c2 = out[1], out[2]
    print('%-28s upload+build %.2f s | reference walk %.1f box tests/ray | production walk %.2f nodes, %.2f triangle tests/ray | 16 spp in %.1f ms'
          % (name, tb, c1['box_tests'] / c1['rays'], c2['wide_nodes'] / c2['rays'], c2['wide_tris'] / c2['rays'], dt * 1e3), flush=True)
r.close()
