"""Where a wave of the path stage kernel spends its time (a -DPT_STAMPS_PATH=1 build, PTAMD_LIB pointing at it)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'cuda-pathtracer_amd', 'python')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ptamd, util
from ptamd import scenes, lib
b = util.atrium()
P = scenes.atrium_params(ptamd.default_params(1920, 1080))
r = b.renderer(P); r.set_max_bounces(512); r.set_lanes(1)
d = (C.c_ulonglong * 16)()
r.render(P, 0, 4)
lib.check(lib.lib().pt_debug_words(r.h, d))
r.render(P, 4, 20)
lib.check(lib.lib().pt_debug_words(r.h, d))
names = ['segment grab', 'record + hit loads', 'triangle + material loads', 'diffuse map', 'other maps', 'vertex (medium, light sample)', 'finish + regenerate', 'queue positions', 'stores']
tot = float(sum(d[k] for k in range(9)))
print('wave-segments', d[9], 'clocks per wave-segment %.0f' % (tot / max(d[9], 1)))
for k, n in enumerate(names):
    print('%-32s %5.1f %%  %8.0f clocks' % (n, 100.0 * d[k] / tot, d[k] / max(d[9], 1)))
r.close()
