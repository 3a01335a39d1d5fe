"""Small renders with one and two lanes: where does the second lane start to pay?"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'cuda-pathtracer_amd', 'python')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ptamd, util
from ptamd import scenes
b = util.atrium()
P = scenes.atrium_params(ptamd.default_params(1920, 1080))
r = b.renderer(P)
for frames in (1, 2, 4, 8, 16):
    for lanes in (1, 2):
        r.set_lanes(lanes)
        r.render(P, 0, frames)
        n = max(2, 32 // frames)
        t0 = time.perf_counter()
        for k in range(n): r.render(P, frames * k, frames * (k + 1))
        dt = time.perf_counter() - t0
        print('frames per call %2d lanes %d: %.2f ms per call' % (frames, lanes, dt / n * 1e3), flush=True)
