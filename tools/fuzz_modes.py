"""Random render configurations through both kernel organisations (stage kernels over queues / one kernel per sample):
the accumulators must be bit-identical whatever the image size, frame range, samples in flight, lanes, drain threshold,
bounce cap, sky mode or scene. usage: tools/fuzz_modes.py [cases] [seed]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'cuda-pathtracer_amd', 'python')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ptamd, util
from ptamd import scenes
from util import bits
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bundles = {'cornell_sky': (util.cornell('sky'), scenes.cornell_params), 'cornell_black': (util.cornell('black'), scenes.cornell_params),
           'atrium': (util.atrium(detail=0.25), scenes.atrium_params), 'atrium_glass_sunsky': (util.atrium(detail=0.25, glossy=0.3, glass=0.3, sky_mode=1), scenes.atrium_params)}
bad = 0
for case in range(n_cases):
    name = list(bundles)[int(rng.integers(len(bundles)))]
    b, mkparams = bundles[name]
    W, H = int(rng.integers(1, 200)), int(rng.integers(1, 120))
    if rng.random() < 0.4:
        W, H = W // 8 * 8 + 8, H // 8 * 8 + 8                       # tiled sample order
    f0 = int(rng.integers(0, 5)); frames = int(rng.integers(1, 9))
    pool = int(rng.choice([64, 256, 700, 4096, 65536, 1 << 27])); chunk = int(rng.choice([1, 2, 3, 64]))
    lanes = int(rng.integers(1, 3)); drain = int(rng.choice([0, 64, 5000, 262144])); bounces = int(rng.choice([1, 2, 5, 12, 40]))
    P = mkparams(ptamd.default_params(W, H))
    if rng.random() < 0.3:
        P.DOF_strenght = 2.0
    r = b.renderer(P, 0)
    r.set_max_bounces(bounces); r.set_pool(pool, chunk); r.set_lanes(lanes); r.set_drain(drain)
    r.set_mode(1); r.clear_accum(); r.render(P, f0, f0 + frames); want = r.read_accum()
    r.set_mode(0); r.clear_accum(); r.render(P, f0, f0 + frames); got = r.read_accum()
    it = r.wavefront_stats()['iterations']
    r.close()
    ok = np.array_equal(bits(got), bits(want)) and np.isfinite(got).all()
    bad += not ok
    print('%-20s %3dx%-3d frames [%d,%d) pool %9d chunk %2d lanes %d drain %6d bounces %2d iterations %4d  %s'
          % (name, W, H, f0, f0 + frames, pool, chunk, lanes, drain, bounces, it, 'ok' if ok else 'DIFFERENT'), flush=True)
print('%d of %d cases differ' % (bad, n_cases))
sys.exit(1 if bad else 0)
