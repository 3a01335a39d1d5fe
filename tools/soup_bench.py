"""Triangle-soup configuration (n random small triangles in a 10000^3 cube, 64 materials): host build time,
closest-hit rate of the production traversal kernel on random rays, and render rate at 1080p."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'cuda-pathtracer_amd', 'python')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ptamd, util
from ptamd import scenes, lib
from util import bits
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
t0 = time.time()
tris, mats = scenes.soup(n, seed=2, extent=5000.0, edge=5.0)
t1 = time.time()
nodes, idx = lib.host_build_bvh(tris)
t2 = time.time()
print('soup %d: generate %.1f s, host BVH build %.1f s (%d nodes)' % (n, t1 - t0, t2 - t1, len(nodes)), flush=True)
env = scenes.sky_envmap(256, 128)
env4 = np.concatenate([env, np.zeros(env.shape[:2] + (1,), np.float32)], -1)
cdf, total = lib.build_env_cdf(env4)
P = ptamd.default_params(1920, 1080)
P.rayOrigin = ptamd.Float3(0., 0., 9000.)
P.isRendering = 1
r = ptamd.Renderer(0)
r.init(P)
t3 = time.time()
r.upload_scene(tris, mats, idx, nodes)
r.upload_envmap(env4, cdf, total, P)
print('upload + flatten %.1f s, scene bytes %s' % (time.time() - t3, r.scene_bytes()), flush=True)
rays = scenes.random_rays(4000000, seed=5, extent=5000.0)
for width in (4, 2):
    r.set_node_width(width)
    h = r.trace_closest_fast(rays); h = r.trace_closest_fast(rays)
    ms = r.counters()['last_ms']
    print('width %d: %d rays, %d hits, %.2f ms -> %.1f Mrays/s' % (width, len(rays), (h['tri'] != 0xFFFFFFFF).sum(), ms, len(rays) / ms / 1e3), flush=True)
ref = r.trace_closest(rays[:500000])
r.set_node_width(4)
h = r.trace_closest_fast(rays[:500000])
print('exact vs reference-order kernel:', bool(np.array_equal(h['tri'], ref['tri']) and np.array_equal(bits(h['t']), bits(ref['t']))), flush=True)
t5 = time.time()
r.upload_scene(tris, mats, None, None, flavour=lib.PT_BVH_GPU_LBVH)
print('GPU LBVH: upload + build + flatten %.1f s' % (time.time() - t5), flush=True)
r.set_node_width(4)
h2 = r.trace_closest_fast(rays); h2 = r.trace_closest_fast(rays)
ms = r.counters()['last_ms']
print('LBVH tree: %.2f ms -> %.1f Mrays/s, same hits as the reference tree: %s' % (ms, len(rays) / ms / 1e3, bool(np.array_equal(h2['tri'], r.trace_closest_fast(rays)['tri']) and np.array_equal(h2['tri'][:500000], ref['tri']) and np.array_equal(bits(h2['t'][:500000]), bits(ref['t'])))), flush=True)
r.upload_scene(tris, mats, idx, nodes)
r.set_max_bounces(512)
r.render(P, 0, 1)
r.enable_counters(True); r.reset_counters(); r.render(P, 0, spp); c = r.counters(); r.enable_counters(False)
r.clear_accum()
t4 = time.perf_counter(); r.render(P, 0, spp); dt = time.perf_counter() - t4
print('render 1920x1080 x %d spp: %.1f ms, %.1f Mrays/s, %.2f rays/sample, box tests/ray %.1f, tri tests/ray %.1f'
      % (spp, dt * 1e3, c['rays'] / dt / 1e6, c['rays'] / c['samples'], c['box_tests'] / c['rays'], c['tri_tests'] / c['rays']), flush=True)
acc = r.read_accum()
print('mean radiance %.4f, finite %s' % (acc.mean(), bool(np.isfinite(acc).all())))
