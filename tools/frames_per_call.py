import sys, time, os
ROOT = '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd()
sys.path.insert(0, os.path.join(ROOT, 'cuda-pathtracer_amd', 'python')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ptamd, util
from ptamd import scenes
b = util.atrium()
W, H = 1920, 1080
P = scenes.atrium_params(ptamd.default_params(W, H))
r = b.renderer(P)
for k in (1, 2, 4, 8, 16, 32, 64):
    r.clear_accum(); r.render(P, 0, k)
    t=time.perf_counter(); r.clear_accum(); r.render(P, 100, 100+k); dt=time.perf_counter()-t
    print('%d frames in one call: %.2f ms per frame' % (k, dt/k*1e3))
