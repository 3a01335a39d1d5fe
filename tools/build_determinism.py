"""Repeated tree builds on the device (radix and agglomerative flavours) must give the same tree: same walk counters, same node count."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'cuda-pathtracer_amd', 'python')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ptamd, util
from ptamd import scenes, lib
b = util.soup(500000, seed=4, extent=800.0, edge=8.0)
P = ptamd.default_params(64, 48); P.rayOrigin = ptamd.Float3(0, 0, 1500.0); P.isRendering = 1
r = b.renderer(P)
for fl in (lib.PT_BVH_GPU_LBVH, lib.PT_BVH_GPU_PLOC):
    seen = []
    for rep in range(3):
        r.upload_scene(b.tris, b.mats, None, None, flavour=fl)
        r.enable_counters(2); r.reset_counters(); r.clear_accum(); r.render(P, 0, 2); c = r.counters(); r.enable_counters(0)
        seen.append((c['wide_nodes'], c['wide_tris'], r.scene_bytes()))
    print(fl, seen, 'deterministic' if len(set(seen)) == 1 else 'DIFFERS')
