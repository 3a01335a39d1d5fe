"""Diagnostic (library built with -DPT_STAMPS=2): one of the rays that took the most passes in the extend kernel."""
import ctypes as C, os, sys, struct
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'cuda-pathtracer_amd', 'python')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ptamd, util
from ptamd import scenes, lib
b = util.atrium()
P = scenes.atrium_params(ptamd.default_params(1920, 1080))
r = b.renderer(P)
r.set_lanes(1)
d = (C.c_ulonglong * 16)()
for f in range(0, 8, 2):
    r.render(P, f, f + 2)
    lib.check(lib.lib().pt_debug_words(r.h, d))
    fl = lambda v: struct.unpack('f', struct.pack('I', v & 0xffffffff))[0]
    print('o', [fl(d[i]) for i in (4, 5, 6)], 'd', [fl(d[i]) for i in (7, 8, 9)], 'passes', d[10], 't', fl(d[11]), 'max', d[2], flush=True)
