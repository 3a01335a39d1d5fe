"""ctypes binding of oracle/liboracle.so (test infrastructure: the CPU restatement of the reference)."""
import ctypes as C
import os
import numpy as np

import ptamd
from ptamd.types import TRI_DTYPE, NODE_DTYPE, MAT_DTYPE, RAY_DTYPE, Params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_HIT = np.dtype([('t', '<f4'), ('tri', '<u4'), ('u', '<f4'), ('v', '<f4')])


class OrcTexture(C.Structure):
    _fields_ = [('texels', C.c_void_p), ('w', C.c_int), ('h', C.c_int), ('comps', C.c_int)]


class OrcScene(C.Structure):
    _fields_ = [('tris', C.c_void_p), ('n_tris', C.c_int), ('tri_idx', C.c_void_p), ('nodes', C.c_void_p),
                ('n_nodes', C.c_int), ('mats', C.c_void_p), ('n_mats', C.c_int), ('textures', C.c_void_p),
                ('n_textures', C.c_int), ('env_rgba', C.c_void_p), ('env_cdf', C.c_void_p), ('env_w', C.c_int),
                ('env_h', C.c_int), ('env_sum', C.c_float), ('sky_mode', C.c_int), ('sun', C.c_float * 3)]


class OrcStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ('rays', 'samples', 'box_tests', 'tri_tests', 'shade_vertices',
                                          'env_samples', 'tex_taps', 'tex_taps_r')]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ROOT, 'oracle', 'liboracle.so')
        if not os.path.exists(path):
            raise RuntimeError('oracle/liboracle.so missing: run `make -C oracle` (or __graft_entry__.build())')
        _lib = C.CDLL(path)
        _lib.orc_build_bvh.restype = C.c_int
        _lib.orc_render.restype = C.c_int
        _lib.orc_max_threads.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Scene:
    """Host arrays of one scene in the reference's layouts, viewable by the oracle."""

    def __init__(self, tris, mats, tri_idx=None, nodes=None, textures=(), env_rgba=None, env_cdf=None,
                 env_sum=0.0, sky_mode=0, sun=(0.0, 1.0, 0.0)):
        self.tris = np.ascontiguousarray(tris, dtype=TRI_DTYPE)
        self.mats = np.ascontiguousarray(mats, dtype=MAT_DTYPE)
        n = len(self.tris)
        self.tri_idx = np.arange(n, dtype=np.int32) if tri_idx is None else np.ascontiguousarray(tri_idx, np.int32)
        self.nodes = nodes
        self.nodes_used = 0 if nodes is None else len(nodes)
        self.textures = [np.ascontiguousarray(t, dtype=np.float32) for t in textures]   # each (h, w, comps)
        self.env_rgba = None if env_rgba is None else np.ascontiguousarray(env_rgba, dtype=np.float32)  # (h, w, 4)
        self.env_cdf = None if env_cdf is None else np.ascontiguousarray(env_cdf, dtype=np.float32)
        self.env_sum = float(env_sum)
        self.sky_mode = sky_mode
        self.sun = tuple(float(x) for x in sun)

    def build_bvh_reference(self):
        """buildBVH as the reference does it (slow exhaustive sweeps)."""
        n = len(self.tris)
        self.nodes = np.zeros(max(2 * n - 1, 1), dtype=NODE_DTYPE)
        self.tri_idx = np.arange(n, dtype=np.int32)
        self.nodes_used = lib().orc_build_bvh(_p(self.tris), _p(self.tri_idx), C.c_int(n), _p(self.nodes))
        return self.nodes_used

    def set_env(self, rgba):
        self.env_rgba = np.ascontiguousarray(rgba, dtype=np.float32)
        h, w, _ = self.env_rgba.shape
        self.env_cdf = np.zeros((h, w), dtype=np.float32)
        s = C.c_float(0)
        lib().orc_build_cdf(_p(self.env_rgba), C.c_int(w), C.c_int(h), _p(self.env_cdf), C.byref(s))
        self.env_sum = s.value

    def c_scene(self):
        self._tex_structs = (OrcTexture * max(len(self.textures), 1))()
        for i, t in enumerate(self.textures):
            h, w, c = t.shape
            self._tex_structs[i] = OrcTexture(t.ctypes.data, w, h, c)
        s = OrcScene()
        s.tris = self.tris.ctypes.data
        s.n_tris = len(self.tris)
        s.tri_idx = self.tri_idx.ctypes.data
        s.nodes = self.nodes.ctypes.data if self.nodes is not None else None
        s.n_nodes = self.nodes_used
        s.mats = self.mats.ctypes.data
        s.n_mats = len(self.mats)
        s.textures = C.cast(self._tex_structs, C.c_void_p)
        s.n_textures = len(self.textures)
        if self.env_rgba is not None:
            s.env_rgba = self.env_rgba.ctypes.data
            s.env_cdf = self.env_cdf.ctypes.data
            s.env_h, s.env_w = self.env_rgba.shape[:2]
        s.env_sum = self.env_sum
        s.sky_mode = self.sky_mode
        s.sun = (C.c_float * 3)(*self.sun)
        return s


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return list(o)


def uniform_stream(pixel, frame, n):
    out = np.zeros(n, dtype=np.float32)
    lib().orc_uniform_stream(C.c_uint32(pixel), C.c_int(frame), C.c_int(n), _p(out))
    return out


def closest_brute(scene, rays):
    rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
    out = np.zeros(len(rays), dtype=ORC_HIT)
    cs = scene.c_scene()
    lib().orc_closest_brute(C.byref(cs), _p(rays), C.c_int(len(rays)), _p(out))
    return out


def closest_bvh(scene, rays, counted=False):
    rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
    out = np.zeros(len(rays), dtype=ORC_HIT)
    nb = np.zeros(len(rays), dtype=np.uint32)
    nt = np.zeros(len(rays), dtype=np.uint32)
    cs = scene.c_scene()
    lib().orc_closest_bvh(C.byref(cs), _p(rays), C.c_int(len(rays)), _p(out), _p(nb), _p(nt))
    return (out, nb, nt) if counted else out


def closest_reference_order(scene, rays):
    rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
    out = np.zeros(len(rays), dtype=ORC_HIT)
    cs = scene.c_scene()
    lib().orc_closest_reference_order(C.byref(cs), _p(rays), C.c_int(len(rays)), _p(out))
    return out


def render(scene, params, frame_begin, frame_end, max_bounces, accum=None, nthreads=0):
    W, H = params.windowSize.x, params.windowSize.y
    if accum is None:
        accum = np.zeros((H, W, 3), dtype=np.float32)
    st = OrcStats()
    cs = scene.c_scene()
    lib().orc_render(C.byref(cs), C.byref(params), C.c_int(frame_begin), C.c_int(frame_end), C.c_int(max_bounces),
                     _p(accum), C.byref(st), C.c_int(nthreads))
    return accum, st.as_dict()


def pixel_sample(scene, params, x, y, frame, max_bounces):
    rgb = (C.c_float * 3)()
    st = OrcStats()
    cs = scene.c_scene()
    lib().orc_pixel_sample(C.byref(cs), C.byref(params), C.c_int(x), C.c_int(y), C.c_int(frame), C.c_int(max_bounces),
                           rgb, C.byref(st))
    return np.array(list(rgb), dtype=np.float32), st.as_dict()


def resolve_display(accum, n_frames):
    H, W, _ = accum.shape
    out = np.zeros((H, W, 4), dtype=np.uint8)
    lib().orc_resolve_display(_p(np.ascontiguousarray(accum, np.float32)), C.c_int(W), C.c_int(H), C.c_int(n_frames), _p(out))
    return out


def post_process(rgba, params):
    rgba = np.ascontiguousarray(rgba, np.uint8)
    H, W, _ = rgba.shape
    out = np.zeros_like(rgba)
    lib().orc_post_process(_p(rgba), C.c_int(W), C.c_int(H), C.byref(params), _p(out))
    return out


def agx(rgb):
    i = (C.c_float * 3)(*[float(x) for x in rgb])
    o = (C.c_float * 3)()
    lib().orc_agx(i, o)
    return np.array(list(o), dtype=np.float32)


def tex2d(tex, u, v):
    t = np.ascontiguousarray(tex, dtype=np.float32)
    h, w, c = t.shape
    ts = OrcTexture(t.ctypes.data, w, h, c)
    o = (C.c_float * 4)()
    lib().orc_tex2d(C.byref(ts), C.c_float(u), C.c_float(v), o)
    return np.array(list(o)[:c], dtype=np.float32)


def camera_basis(params):
    ww, uu, vv = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)()
    lib().orc_camera_basis(C.byref(params), ww, uu, vv)
    return np.array(list(ww), np.float32), np.array(list(uu), np.float32), np.array(list(vv), np.float32)


def max_threads():
    return lib().orc_max_threads()


def detmath_probe(x, y):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    out = np.zeros((len(x), 12), dtype=np.float32)
    lib().orc_detmath_probe(_p(x), _p(y), C.c_int(len(x)), _p(out))
    return out


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def bsdf_eval(mat, inside, V, N, L):
    """DisneyEval_2 (disney.cuh:234-353) on one material record: returns (f[3], pdf)."""
    m = np.ascontiguousarray(mat, dtype=MAT_DTYPE).reshape(1)
    f = (C.c_float * 3)()
    pdf = C.c_float()
    lib().orc_bsdf_eval(_p(m), C.c_int(int(inside)), _f3(V), _f3(N), _f3(L), f, C.byref(pdf))
    return np.array(list(f), np.float32), pdf.value


def bsdf_sample(mat, inside, V, N, pixel, frame):
    """DisneySample_2 (disney.cuh:355-460) with the random stream of (pixel, frame): returns (L, f, pdf, inside')."""
    m = np.ascontiguousarray(mat, dtype=MAT_DTYPE).reshape(1)
    L, f = (C.c_float * 3)(), (C.c_float * 3)()
    pdf, ins = C.c_float(), C.c_int()
    lib().orc_bsdf_sample(_p(m), C.c_int(int(inside)), _f3(V), _f3(N), C.c_uint32(pixel), C.c_int(frame), L, f, C.byref(pdf), C.byref(ins))
    return np.array(list(L), np.float32), np.array(list(f), np.float32), pdf.value, bool(ins.value)


def bsdf_eval_batch(mat, inside, V, N, L):
    """DisneyEval_2 for an array of directions L[n, 3]: returns (f[n, 3], pdf[n])."""
    m = np.ascontiguousarray(mat, dtype=MAT_DTYPE).reshape(1)
    L = np.ascontiguousarray(L, np.float32)
    n = len(L)
    f, pdf = np.zeros((n, 3), np.float32), np.zeros(n, np.float32)
    lib().orc_bsdf_eval_batch(_p(m), C.c_int(int(inside)), _f3(V), _f3(N), _p(L), C.c_int(n), _p(f), _p(pdf))
    return f, pdf


def bsdf_sample_batch(mat, inside, V, N, pixel0, n, frame):
    """DisneySample_2 with the random streams of pixels pixel0 .. pixel0 + n - 1: returns (L[n, 3], f[n, 3], pdf[n], inside'[n])."""
    m = np.ascontiguousarray(mat, dtype=MAT_DTYPE).reshape(1)
    L, f = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    pdf, ins = np.zeros(n, np.float32), np.zeros(n, np.int32)
    lib().orc_bsdf_sample_batch(_p(m), C.c_int(int(inside)), _f3(V), _f3(N), C.c_uint32(pixel0), C.c_int(n), C.c_int(frame), _p(L), _p(f), _p(pdf), _p(ins))
    return L, f, pdf, ins.astype(bool)


def env_eval(scene, d):
    out = (C.c_float * 4)()
    cs = scene.c_scene()
    lib().orc_env_eval(C.byref(cs), _f3(d), out)
    return np.array(list(out), np.float32)


def env_sample(scene, pixel, frame):
    dp, col = (C.c_float * 4)(), (C.c_float * 3)()
    cs = scene.c_scene()
    lib().orc_env_sample(C.byref(cs), C.c_uint32(pixel), C.c_int(frame), dp, col)
    return np.array(list(dp), np.float32), np.array(list(col), np.float32)


def sky_gradient(d, sun):
    out = (C.c_float * 3)()
    lib().orc_sky_gradient(_f3(d), _f3(sun), out)
    return np.array(list(out), np.float32)
