"""A small render through the wavefront kernels against the single-kernel mode, with timings (variant smoke check)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'cuda-pathtracer_amd', 'python')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ptamd, util
from ptamd import scenes
from util import bits
b = util.atrium(detail=0.3, env_size=(64, 32))
for W, H, n in ((72, 40, 4), (320, 200, 4), (640, 360, 8)):
    P = scenes.atrium_params(ptamd.default_params(W, H))
    r = b.renderer(P)
    r.set_max_bounces(16)
    t0 = time.perf_counter(); r.render(P, 0, n); a = r.read_accum(); t1 = time.perf_counter()
    r.set_mode(1); r.clear_accum(); r.render(P, 0, n); c = r.read_accum(); t2 = time.perf_counter()
    print('%dx%d x%d: wavefront %.1f ms, single kernel %.1f ms, equal %s' % (W, H, n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, bool(np.array_equal(bits(a), bits(c)))), flush=True)
    r.close()
