"""Latency of one call that renders 1, 2 or 4 frames of the 1080p atrium alone, against the drain threshold (TRACE=1: three
renders of FRAMES frames (default 1) and nothing else, for a kernel trace: tools/one_frame_trace.py)."""
import sys, time, os
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, 'cuda-pathtracer_amd', 'python')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ptamd, util
from ptamd import scenes
b = util.atrium()
W, H = 1920, 1080
P = scenes.atrium_params(ptamd.default_params(W, H))
r = b.renderer(P)
if os.environ.get('TRACE'):
    k = int(os.environ.get('FRAMES', '1'))
    for rep in range(3):
        r.clear_accum(); r.render(P, rep * k, rep * k + k)
    r.close(); sys.exit(0)
for drain in (262144, 524288, 1048576, 2097152, 65536):
    r.set_drain(drain)
    for k in (1, 2, 4):
        r.clear_accum(); r.render(P, 0, k)
        ts = []
        for rep in range(5):
            r.clear_accum(); t = time.perf_counter(); r.render(P, 100 + rep * k, 100 + rep * k + k); ts.append(time.perf_counter() - t)
        st = r.wavefront_stats()
        print('drain %8d  %d frame(s) per call: %.2f ms per call (min of 5)' % (drain, k, min(ts) * 1e3), flush=True)
