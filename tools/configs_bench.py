"""Throughput of the benchmark configurations other than the headline one, on one GPU.
1: Cornell box 256x256, 16 frames, 4 bounces (also run whole on the CPU oracle)
4: soup of n random triangles, 1920x1080, 16 spp
5: atrium with 30 % glossy metals and 20 % glass, sun/sky gradient instead of the environment map, 16 bounces, 1080p, 64 spp"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'cuda-pathtracer_amd', 'python')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ptamd, util, orc
from ptamd import scenes, lib
from util import bits

def timed(r, P, f0, f1):
    r.render(P, f0, f1)
    r.enable_counters(True); r.reset_counters(); r.clear_accum(); r.render(P, f0, f1); c = r.counters(); r.enable_counters(False)
    r.clear_accum()
    t0 = time.perf_counter(); r.render(P, f0, f1); dt = time.perf_counter() - t0
    return c, dt

# ---- 1
b = util.cornell('black')
P = scenes.cornell_params(ptamd.default_params(256, 256))
r = b.renderer(P); r.set_max_bounces(4)
c, dt = timed(r, P, 0, 16)
acc = r.read_accum(); r.close()
t0 = time.perf_counter(); want, st = orc.render(b.oracle(P), P, 0, 16, 4); dt_cpu = time.perf_counter() - t0
print('config 1 cornell 256x256 x16, 4 bounces: GPU %.1f ms (%.1f Mrays/s), CPU oracle %.0f ms (%.2f Mrays/s, %d threads), bit-identical %s'
      % (dt * 1e3, c['rays'] / dt / 1e6, dt_cpu * 1e3, st['rays'] / dt_cpu / 1e6, orc.max_threads(), bool(np.array_equal(bits(acc), bits(want)))), flush=True)
# ---- 5
b = util.atrium(glossy=0.3, glass=0.2, sky_mode=1)
P = scenes.atrium_params(ptamd.default_params(1920, 1080))
r = b.renderer(P); r.set_max_bounces(16)
c, dt = timed(r, P, 0, 64)
print('config 5 atrium glossy/glass, sun-sky, 1080p x64, 16 bounces: %.1f ms, %.1f Mrays/s, %.1f Msamples/s, %.2f rays/sample'
      % (dt * 1e3, c['rays'] / dt / 1e6, c['samples'] / dt / 1e6, c['rays'] / c['samples']), flush=True)
r.close()
# ---- 4
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
tris, mats = scenes.soup(n, seed=2, extent=5000.0, edge=5.0)
env = scenes.sky_envmap(256, 128)
env4 = np.concatenate([env, np.zeros(env.shape[:2] + (1,), np.float32)], -1)
cdf, total = lib.build_env_cdf(env4)
P = ptamd.default_params(1920, 1080); P.rayOrigin = ptamd.Float3(0., 0., 9000.); P.isRendering = 1
r = ptamd.Renderer(0); r.init(P)
t0 = time.perf_counter()
r.upload_scene(tris, mats, None, None, flavour=lib.PT_BVH_GPU_LBVH)
t_build = time.perf_counter() - t0
r.upload_envmap(env4, cdf, total, P)
c, dt = timed(r, P, 0, 16)
print('config 4 soup %d triangles (GPU-built tree, %.1f s with upload), 1080p x16: %.1f ms, %.1f Mrays/s, %.1f Msamples/s, %.2f rays/sample, %.0f box tests/ray'
      % (n, t_build, dt * 1e3, c['rays'] / dt / 1e6, c['samples'] / dt / 1e6, c['rays'] / c['samples'], c['box_tests'] / c['rays']), flush=True)
r.close()
