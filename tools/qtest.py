"""Production traversal kernel (both node widths) against the reference-order kernel on many random rays."""
import sys; sys.path.insert(0,'cuda-pathtracer_amd/python'); sys.path.insert(0,'tests')
import numpy as np, ptamd, util
from ptamd import scenes
from util import bits
tot = 0
for name, b, ext in (('atrium', util.atrium(detail=0.5), 1400.0), ('soup', util.soup(200000, seed=3, extent=400.0, edge=8.0), 420.0), ('cornell', util.cornell('sky'), 300.0)):
    P = ptamd.default_params(64, 64)
    r = b.renderer(P)
    rays = scenes.random_rays(2000000, seed=9, extent=ext)
    slow = r.trace_closest(rays)
    for width in (4, 2):
        r.set_node_width(width)
        fast = r.trace_closest_fast(rays)
        bad = int((fast['tri'] != slow['tri']).sum() + (bits(fast['t']) != bits(slow['t'])).sum())
        tot += bad
        print(name, 'width', width, 'hits', int((slow['tri'] != 0xFFFFFFFF).sum()), 'mismatches', bad, 'ms', r.counters()['last_ms'])
    r.close()
print('total mismatches', tot)
# axis-parallel and nearly parallel rays (zero and denormal direction components)
b = util.atrium(detail=0.5)
r = b.renderer(ptamd.default_params(64, 64))
rays = scenes.random_rays(600000, seed=10, extent=1400.0)
k = np.arange(len(rays))
for a in range(3):
    rays['d'][k % 7 == a, a] = 0.0
    rays['d'][k % 7 == a + 3, a] = 1e-41
rays['d'][k % 7 == 6] = [0.0, -1.0, -0.0]
n = np.linalg.norm(rays['d'], axis=1, keepdims=True); rays['d'] /= np.maximum(n, 1e-30)
slow = r.trace_closest(rays); fast = r.trace_closest_fast(rays)
bad = int((fast['tri'] != slow['tri']).sum() + (bits(fast['t']) != bits(slow['t'])).sum())
print('axis-parallel rays: hits', int((slow['tri'] != 0xFFFFFFFF).sum()), 'mismatches', bad, 'ms', r.counters()['last_ms'])
r.close()
# rays as a render makes them: origins 0.01 above a surface point (pathtracer.cu:467,480), aimed at vertices, edge points and
# centres of other triangles (hits on shared edges and corners tie at equal distance: the lowest triangle index must win),
# along the directions of an environment map's texel grid (exactly axis-aligned components), and along the surface itself
from ptamd.types import RAY_DTYPE
for name, b in (('atrium', util.atrium(detail=0.5)), ('atrium full', util.atrium()), ('cornell', util.cornell('sky'))):
    rng = np.random.default_rng(5)
    T = b.tris
    n = 1500000
    i = rng.integers(0, len(T), n); j = rng.integers(0, len(T), n)
    u = rng.uniform(0, 1, n); v = rng.uniform(0, 1, n); f = u + v > 1; u[f] = 1 - u[f]; v[f] = 1 - v[f]
    u[::7] = 0.0; v[::11] = 0.0
    a, e1, e2 = T['a'][i].astype(np.float64), (T['b'][i] - T['a'][i]).astype(np.float64), (T['c'][i] - T['a'][i]).astype(np.float64)
    p = a + e1 * u[:, None] + e2 * v[:, None]
    nrm = np.cross(e1, e2); nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
    nrm *= np.where(rng.uniform(0, 1, n) > 0.5, 1.0, -1.0)[:, None]
    o = p + nrm * 0.01
    kind = np.arange(n) % 5
    tgt = np.where((kind == 0)[:, None], T['a'][j], np.where((kind == 1)[:, None], 0.5 * (T['b'][j].astype(np.float64) + T['c'][j]), (T['a'][j].astype(np.float64) + T['b'][j] + T['c'][j]) / 3.0))
    d = tgt - o
    th = np.floor(rng.uniform(0, 512, n)) / 512.0 * np.pi; ph = np.floor(rng.uniform(0, 1024, n)) / 1024.0 * 2 * np.pi
    ph[::3] = 0.0
    env = np.stack([-np.sin(th) * np.cos(ph), np.cos(th), -np.sin(th) * np.sin(ph)], 1)
    d[kind == 3] = env[kind == 3]
    # grazing, not coplanar: a ray lying exactly in a triangle's plane makes triangleIntersect's result pure rounding
    # (|det| just above its 1e-4 threshold), which no bounding-volume scheme, the reference's included, reproduces
    along = e1 * rng.uniform(-1, 1, n)[:, None] + e2 * rng.uniform(-1, 1, n)[:, None]
    along += nrm * (np.linalg.norm(along, axis=1) * 10.0 ** rng.uniform(-4, -2, n))[:, None]
    d[kind == 4] = along[kind == 4]
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)
    rays = np.zeros(n, RAY_DTYPE); rays['o'] = o.astype(np.float32); rays['d'] = d.astype(np.float32)
    r = b.renderer(ptamd.default_params(64, 64))
    slow = r.trace_closest(rays); fast = r.trace_closest_fast(rays)
    bad = (fast['tri'] != slow['tri']) | (bits(fast['t']) != bits(slow['t']))
    # a ray lying in the plane of the triangle the exhaustive search reports makes triangleIntersect's result pure rounding
    # (|det| just above its 1e-4 threshold; u = v = 0 comes out): no bounding-volume scheme, the reference's included,
    # reproduces those, so they are counted apart
    hit = slow['tri'] != 0xFFFFFFFF
    tn = np.zeros((n, 3)); ti = slow['tri'][hit].astype(np.int64)
    tn[hit] = np.cross((T['b'][ti] - T['a'][ti]).astype(np.float64), (T['c'][ti] - T['a'][ti]).astype(np.float64))
    tn /= np.maximum(np.linalg.norm(tn, axis=1, keepdims=True), 1e-30)
    coplanar = hit & (np.abs(np.sum(tn * d, axis=1)) < 1e-5)
    print(name, 'rays in the plane of the triangle they hit: %d, of which mismatching %d' % (int(coplanar.sum()), int((bad & coplanar).sum())))
    bad &= ~coplanar
    tot += int(bad.sum())
    print(name, 'surface rays: hits', int((slow['tri'] != 0xFFFFFFFF).sum()), 'mismatches', int(bad.sum()), 'ms', r.counters()['last_ms'])
    if bad.any():
        k = np.nonzero(bad)[0][:5]
        print('  first:', [(int(x), rays['o'][x].tolist(), rays['d'][x].tolist(), (int(slow['tri'][x]), float(slow['t'][x])), (int(fast['tri'][x]), float(fast['t'][x]))) for x in k])
    r.close()
print('total mismatches', tot)
