"""Diagnostic: per-section cycle stamps of the extend kernel (library built with -DPT_STAMPS=1)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'cuda-pathtracer_amd', 'python')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ptamd, util
from ptamd import scenes, lib
b = util.atrium()
P = scenes.atrium_params(ptamd.default_params(1920, 1080))
r = b.renderer(P)
r.render(P, 0, 4)
d = (C.c_ulonglong * 16)()
lib.check(lib.lib().pt_debug_words(r.h, d))
r.render(P, 4, 20)
lib.check(lib.lib().pt_debug_words(r.h, d))
refill, box, leaf, iters, lanes, leafph, waves = [d[i] for i in range(7)]
tot = refill + box + leaf
print('waves %d  iterations/wave %.0f  cycles/iteration %.0f  (refill+vote %.0f, interior step %.0f, leaf phase %.0f)' % (waves, iters / waves, tot / iters, refill / iters, box / iters, leaf / iters))
print('descending lanes per iteration %.1f   leaf phases per iteration %.3f' % (lanes / iters, leafph / iters))
