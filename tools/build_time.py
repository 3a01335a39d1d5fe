"""Wall time of pt_build_bvh on the 10-million-triangle soup (GPU tree build + flattening + upload) and of the host builder's tree upload."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'cuda-pathtracer_amd', 'python')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ptamd
from ptamd import scenes, lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
tris, mats = scenes.soup(n, seed=2)
P = ptamd.default_params(64, 64)
r = ptamd.Renderer(0); r.init(P)
import ctypes as C
from ptamd.lib import _p, check
t = time.perf_counter(); check(lib.lib().pt_upload_triangles(r.h, None, _p(tris), C.c_int(n))); check(lib.lib().pt_upload_materials(r.h, _p(mats), C.c_int(len(mats)))); print('upload triangles %.2f s' % (time.perf_counter() - t), flush=True)
for name, flavour in (('GPU_LBVH', lib.PT_BVH_GPU_LBVH), ('GPU_PLOC', lib.PT_BVH_GPU_PLOC), ('GPU_LBVH_VIA_HOST', lib.PT_BVH_GPU_LBVH_VIA_HOST)):
    for rep in range(2):
        t = time.perf_counter(); check(lib.lib().pt_build_bvh(r.h, C.c_int(flavour))); r.sync(); print('pt_build_bvh(%s) %.3f s' % (name, time.perf_counter() - t), flush=True)
r.close()
