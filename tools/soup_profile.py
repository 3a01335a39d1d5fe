"""The 10 M-triangle soup at 1080p, 16 spp, tree built on the GPU: the workload whose traversal does not fit the
caches. Run under rocprofv3 (kernel trace, FETCH_SIZE, WRITE_SIZE passes) to get the memory-side traffic of the
traversal kernel; prints the ray and test counts the algorithmic byte figure needs."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'cuda-pathtracer_amd', 'python')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ptamd, util
from ptamd import scenes, lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
count = len(sys.argv) > 3 and sys.argv[3] == 'count'
tris, mats = scenes.soup(n, seed=2, extent=5000.0, edge=5.0)
env = scenes.sky_envmap(256, 128)
env4 = np.concatenate([env, np.zeros(env.shape[:2] + (1,), np.float32)], -1)
cdf, total = lib.build_env_cdf(env4)
P = ptamd.default_params(1920, 1080)
P.rayOrigin = ptamd.Float3(0., 0., 9000.)
P.isRendering = 1
r = ptamd.Renderer(0)
r.init(P)
r.upload_scene(tris, mats, None, None, flavour=lib.PT_BVH_GPU_LBVH)
r.upload_envmap(env4, cdf, total, P)
r.set_max_bounces(512)
print('scene bytes', r.scene_bytes(), flush=True)
r.render(P, 0, 1)
if count:
    r.enable_counters(True); r.reset_counters(); r.render(P, 0, spp); c = r.counters(); r.enable_counters(False)
    print('rays %d box %d tri %d samples %d' % (c['rays'], c['box_tests'], c['tri_tests'], c['samples']), flush=True)
r.clear_accum()
r.stage_timing(True)
t0 = time.perf_counter()
r.render(P, 0, spp)
dt = time.perf_counter() - t0
ms, launches = r.stage_timing(False)
print('render %d spp: %.1f ms; extend %.3f ms x %d launches' % (spp, dt * 1e3, ms / max(launches, 1), launches), flush=True)
r.close()
