"""Diagnostic: per-section cycle stamps of the extend kernel (library built with -DPT_STAMPS=1)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'cuda-pathtracer_amd', 'python')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ptamd, util
from ptamd import scenes, lib
b = util.atrium()
P = scenes.atrium_params(ptamd.default_params(1920, 1080))
r = b.renderer(P)
pool = int(os.environ.get('POOL', '2097152')); tune = [int(x) for x in os.environ.get('TUNE', '24,24,5').split(',')]
r.set_pool(pool, 64); r.set_extend_tuning(*tune)
print('pool', pool, 'tune', tune)
r.render(P, 0, 4)
d = (C.c_ulonglong * 16)()
lib.check(lib.lib().pt_debug_words(r.h, d))
r.render(P, 4, 20)
lib.check(lib.lib().pt_debug_words(r.h, d))
total, refills, _, iters, lanes, leafph, waves, leaflanes, idle = [d[i] for i in range(9)]
print('waves %d  iterations/wave %.0f  cycles/iteration %.0f  descending lanes/iteration %.1f  leaf passes/iteration %.3f  refills/iteration %.3f'
      % (waves, iters / waves, total / iters, lanes / iters, leafph / iters, refills / iters))
print('lanes in a leaf pass %.1f  idle lanes/iteration %.1f  lanes waiting at a leaf/iteration %.1f' % (leaflanes / max(leafph, 1), idle / iters, 64 - lanes / iters - idle / iters))
tail, tail_idle, tail_desc = d[9], d[10], d[11]
print('tail (no more blocks to take): %.1f %% of the iterations, idle lanes %.1f, descending %.1f; before the tail: idle %.1f, descending %.1f'
      % (100.0 * tail / iters, tail_idle / max(tail, 1), tail_desc / max(tail, 1), (idle - tail_idle) / max(iters - tail, 1), (lanes - tail_desc) / max(iters - tail, 1)))
print('time in the tail: %.1f %% of the wave cycles' % (100.0 * d[12] / total))
print('rays that took >= 64 / 128 / 256 passes: %d / %d / %d; longest %d passes' % (d[13], d[14], d[15], d[2]))
