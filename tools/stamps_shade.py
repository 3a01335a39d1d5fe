"""Diagnostic: where the waves of the shade kernel spend their time (library built with -DPT_STAMPS_SHADE=1)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'cuda-pathtracer_amd', 'python')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ptamd, util
from ptamd import scenes, lib
b = util.atrium()
P = scenes.atrium_params(ptamd.default_params(1920, 1080))
r = b.renderer(P)
r.set_lanes(1)
r.render(P, 0, 2)
d = (C.c_ulonglong * 16)()
lib.check(lib.lib().pt_debug_words(r.h, d))
r.render(P, 2, 34)
lib.check(lib.lib().pt_debug_words(r.h, d))
waves = d[0]
names = ['state loads', 'resolve_hit (triangle, material, texture taps)', 'shadow branch', 'path branch + vertex store', 'bsdf queue sort', 'regenerate', 'state stores']
total = sum(d[1:8])
for i, nm in enumerate(names):
    print('%-48s %5.1f %%   %8.0f clocks per wave' % (nm, 100.0 * d[1 + i] / total, d[1 + i] / waves))
print('per wave: %.1f live lanes, %.1f on a visibility ray, %.1f missed, %.1f finished' % (d[8] / waves, d[9] / waves, d[10] / waves, d[11] / waves))
