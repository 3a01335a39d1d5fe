"""Drives tools/libw8sim.so: node steps / triangle tests per ray of candidate traversal schemes on the atrium."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'cuda-pathtracer_amd', 'python')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ptamd, util, orc
from ptamd import scenes, lib
from ptamd.types import RAY_DTYPE
L = C.CDLL(os.path.join(ROOT, 'tools', 'libw8sim.so'))
which = sys.argv[1] if len(sys.argv) > 1 else 'atrium'
if which == 'atrium':
    b = util.atrium()
    P = scenes.atrium_params(ptamd.default_params(1920, 1080))
    tris, nodes, idx = b.tris, b.nodes, b.idx
    S = b.oracle(P)
else:
    n = int(which)
    tris, mats = scenes.soup(n, seed=2)
    nodes, idx = lib.host_build_bvh(tris)
    P = ptamd.default_params(1920, 1080); P.rayOrigin = ptamd.Float3(0., 0., 9000.)
    S = orc.Scene(tris, mats, idx, nodes)
rng = np.random.default_rng(1)
N = 40000
ww, uu, vv = lib.camera_basis(P)
px = rng.uniform(-0.5, 0.5, N) * 1920 / 1080; py = rng.uniform(-0.5, 0.5, N)
d = uu[None] * px[:, None] + vv[None] * py[:, None] + ww[None] * P.fov
d /= np.linalg.norm(d, axis=1, keepdims=True)
prim = np.zeros(N, RAY_DTYPE); prim['o'] = [P.rayOrigin.x, P.rayOrigin.y, P.rayOrigin.z]; prim['d'] = d
h = orc.closest_bvh(S, prim)
ok = h['tri'] != 0xFFFFFFFF
hp = prim['o'][ok] + prim['d'][ok] * h['t'][ok][:, None]
def hemi(n):
    v = rng.normal(0, 1, (n, 3)); return v / np.linalg.norm(v, axis=1, keepdims=True)
sec = np.zeros(len(hp), RAY_DTYPE); sec['d'] = hemi(len(hp)); sec['o'] = hp + sec['d'] * 0.05
h2 = orc.closest_bvh(S, sec); ok2 = h2['tri'] != 0xFFFFFFFF
hp2 = sec['o'][ok2] + sec['d'][ok2] * h2['t'][ok2][:, None]
ter = np.zeros(len(hp2), RAY_DTYPE); ter['d'] = hemi(len(hp2)); ter['d'][:, 1] = np.abs(ter['d'][:, 1]); ter['o'] = hp2 + ter['d'] * 0.05
rays = np.concatenate([prim, sec, ter, ter])
if which == 'atrium':
    env = np.zeros(len(hp), RAY_DTYPE); th = rng.uniform(0.2, 1.6, len(hp)); env['d'] = np.stack([-np.sin(th), np.cos(th), -0.0 * th], 1); env['o'] = hp; env['o'][:, 2] += 0.01 * np.sign(-hp[:, 2]); env['o'][:, 1] += 0.01
    rays = np.concatenate([rays, env])      # bounce and shadow rays dominate a render
print('rays', len(prim), len(sec), len(ter))
flat = np.zeros((len(rays), 8), np.float32); flat[:, 0:3] = rays['o']; flat[:, 4:7] = rays['d']
out = (C.c_double * 8)()
def run(W, merge, quant, octs, policy):
    L.w8sim(tris.ctypes.data_as(C.c_void_p), C.c_int(len(tris)), nodes.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p),
            flat.ctypes.data_as(C.c_void_p), C.c_int(len(flat)), C.c_int(W), C.c_int(merge), C.c_int(quant), C.c_int(octs), C.c_int(policy), out)
    print('W%d merge%d quant%d oct%d policy%d: steps %.2f tris %.2f stale %.2f pushes %.2f maxsp %d nodes %d longest %d' % (W, merge, quant, octs, policy, out[0], out[1], out[2], out[3], out[4], out[5], out[7]), flush=True)
run(4, 2, 13, 0, 0)     # local 8-bit frames on tight boxes
run(4, 2, 613, 0, 0)    # 6 bits
run(4, 2, 1013, 0, 0)   # 10 bits
run(4, 2, 1213, 0, 0)   # 12 bits
run(4, 2, 1613, 0, 0)   # 16 bits
run(4, 2, 10, 0, 0)     # fp32 tight boxes
run(6, 2, 13, 0, 0)
run(8, 2, 13, 0, 0)
run(8, 2, 1613, 0, 0)
