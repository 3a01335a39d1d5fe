"""Wall time of the first pt_render_frame calls of a view (ramp of the look-ahead batches)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'cuda-pathtracer_amd', 'python')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ptamd, util
from ptamd import scenes
b = util.atrium()
P = scenes.atrium_params(ptamd.default_params(1920, 1080))
r = b.renderer(P)
for f in range(3):
    P.frameIndex = f
    t0 = time.perf_counter(); r.render_frame(P); print('frame', f, (time.perf_counter() - t0) * 1e3, 'ms')
