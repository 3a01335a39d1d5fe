"""Diagnostic: where the production walk and exhaustive search disagree for rays that start far from a flat floor."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'cuda-pathtracer_amd', 'python'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ptamd, orc, util
from util import bits
from test_gpu_parity import _flat_floor
from ptamd.types import RAY_DTYPE

extent = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
tris = _flat_floor(8, extent)
b = util.Bundle(tris, ptamd.default_materials(1))
rng = np.random.default_rng(41)
parts, tags = [], []
for far in (2.0, 50.0, 100.0, 1000.0, 30000.0):
    n = 4000
    rays = np.zeros(n, dtype=RAY_DTYPE)
    target = np.zeros((n, 3), np.float32)
    target[:, :2] = rng.uniform(-extent, extent, (n, 2))
    o = target.copy()
    o[:, 2] = far * extent
    o[n // 3:2 * n // 3, :2] += rng.uniform(-far * extent, far * extent, (n // 3, 2)).astype(np.float32)
    o[2 * n // 3:, 0] += min(far, 100.0) * extent * 40.0
    d = target.astype(np.float64) - o.astype(np.float64)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:n // 3] = [0.0, 0.0, -1.0]
    rays['o'] = o; rays['d'] = d
    parts.append(rays)
    tags += [(far, 0)] * (n // 3) + [(far, 1)] * (2 * n // 3 - n // 3) + [(far, 2)] * (n - 2 * n // 3)
rays = np.concatenate(parts)
tags = np.array(tags)
r = b.renderer(ptamd.default_params(64, 64))
fast = r.trace_closest_fast(rays)
slow = r.trace_closest(rays)
brute = orc.closest_brute(b.oracle(), rays)
lost = (fast['tri'] != brute['tri']) | (bits(fast['t']) != bits(brute['t']))
lost_s = (slow['tri'] != brute['tri']) | (bits(slow['t']) != bits(brute['t']))
for far in (2.0, 50.0, 100.0, 1000.0, 30000.0):
    for k in range(3):
        m = (tags[:, 0] == far) & (tags[:, 1] == k)
        print('far %7g kind %d: n %d brute hits %d  fast differs %d (fast miss, brute hit: %d; other tri same t: %d)  slow differs %d' % (
            far, k, m.sum(), (brute['tri'][m] != 0xFFFFFFFF).sum(), lost[m].sum(),
            (lost & (fast['tri'] == 0xFFFFFFFF) & (brute['tri'] != 0xFFFFFFFF))[m].sum(),
            (lost & (fast['tri'] != brute['tri']) & (bits(fast['t']) == bits(brute['t'])))[m].sum(), lost_s[m].sum()))
idx = np.nonzero(lost)[0][:6]
for i in idx:
    print(i, 'o', rays['o'][i], 'd', rays['d'][i], 'fast', fast['tri'][i], fast['t'][i], 'brute', brute['tri'][i], brute['t'][i], 'slow', slow['tri'][i], slow['t'][i])
r.close()
