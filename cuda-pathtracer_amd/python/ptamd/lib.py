"""ctypes binding of libptamd.so, the C-ABI declared in include/pt_capi.h.

This is plumbing for tests and the benchmark: every call goes straight to the shared library.
There is no Python or CPU fallback; a missing library or a missing GPU raises.
"""
import ctypes as C
import os
import sys
import numpy as np

from .types import TRI_DTYPE, NODE_DTYPE, MAT_DTYPE, RAY_DTYPE, HIT_DTYPE, Params, Counters, Float3

_PKG = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB_PATH = os.environ.get('PTAMD_LIB') or os.path.join(_PKG, 'libptamd.so')     # PTAMD_LIB: try an experimental build

PT_SKY_ENVMAP, PT_SKY_SUNSKY = 0, 1
PT_BVH_REFERENCE = 0
PT_BVH_GPU_LBVH = 1
PT_BVH_GPU_LBVH_VIA_HOST = 2
PT_BVH_GPU_PLOC = 3
PT_MODE_WAVEFRONT, PT_MODE_SINGLE_KERNEL = 0, 1
PT_NOALLOC_MAX_RAYS = 262144     # include/pt_capi.h

EXPORTS = [
    'pt_last_error', 'pt_create', 'pt_destroy', 'pt_set_stream', 'pt_init', 'pt_clear_accum', 'pt_upload_triangles',
    'pt_upload_materials', 'pt_upload_bvh', 'pt_build_bvh', 'pt_render_frame', 'pt_render_frame_async', 'pt_frame_fetch', 'pt_set_frame_lookahead', 'pt_upload_texture', 'pt_upload_envmap',
    'pt_set_sky_mode', 'pt_set_env_guides', 'pt_set_max_bounces', 'pt_set_mode', 'pt_set_pool', 'pt_set_drain', 'pt_wavefront_stats', 'pt_set_lanes', 'pt_set_extend_tuning', 'pt_set_node_width', 'pt_render_headless', 'pt_group_create', 'pt_group_frames', 'pt_group_render', 'pt_group_destroy', 'pt_sync', 'pt_read_accum', 'pt_write_accum',
    'pt_accum_device_ptr', 'pt_bind_accum', 'pt_resolve_display', 'pt_save_png', 'pt_resolve_display_post', 'pt_save_pfm', 'pt_save_exr', 'pt_load_pfm', 'pt_trace_closest', 'pt_trace_closest_dev',
    'pt_trace_closest_counted', 'pt_trace_closest_fast', 'pt_trace_closest_fast_dev', 'pt_trace_closest_noalloc_dev', 'pt_debug_words', 'pt_get_counters', 'pt_reset_counters', 'pt_enable_counters', 'pt_stage_timing', 'pt_stage_times', 'pt_scene_bytes',
    'pt_detmath_probe', 'pt_bsdf_probe', 'pt_unorm8_probe', 'pt_set_texture_packing', 'pt_read_obj', 'pt_read_obj_general', 'pt_host_scene_free', 'pt_host_scene_counts', 'pt_host_scene_triangles',
    'pt_host_scene_tri_idx', 'pt_host_scene_materials', 'pt_host_scene_material_name', 'pt_host_scene_texture_info',
    'pt_host_build_bvh', 'pt_load_hdr', 'pt_load_texture_png', 'pt_load_texture_image', 'pt_build_env_cdf', 'pt_free', 'pt_camera_basis',
]

_lib = None


class PtError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PtError('%s is missing: build it with `make -C cuda-pathtracer_amd` (or __graft_entry__.build())' % LIB_PATH)
        # torch ships its own copy of the HIP runtime (soname libamdhip64.so.7, asked for as libamdhip64.so). Loaded first,
        # that copy also satisfies this library's libamdhip64.so.7 and the process holds one runtime; loaded after this
        # library it becomes a second runtime beside /opt/rocm's and torch then finds no GPU. So, where torch is installed
        # and not yet imported, it is imported here (PTAMD_NO_TORCH_FIRST=1 skips that).
        if 'torch' not in sys.modules and os.environ.get('PTAMD_NO_TORCH_FIRST') != '1':
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        L.pt_last_error.restype = C.c_char_p
        L.pt_host_scene_triangles.restype = C.c_void_p
        L.pt_host_scene_tri_idx.restype = C.c_void_p
        L.pt_host_scene_materials.restype = C.c_void_p
        L.pt_host_scene_material_name.restype = C.c_char_p
        L.pt_free.argtypes = [C.c_void_p]
        L.pt_destroy.argtypes = [C.c_void_p]
        L.pt_host_scene_free.argtypes = [C.c_void_p]
        L.pt_group_destroy.argtypes = [C.c_void_p]
        for n in ('pt_host_scene_triangles', 'pt_host_scene_tri_idx', 'pt_host_scene_materials'):
            getattr(L, n).argtypes = [C.c_void_p]
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise PtError('libptamd error %d: %s' % (rc, lib().pt_last_error().decode('utf-8', 'replace')))


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# ------------------------------------------------------------------ host-side producers (no GPU)
class HostScene:
    """Result of pt_read_obj: triangles, index array, materials and the texture files the MTL names."""

    def __init__(self, obj_path, model_dir, sub_dir='', general=False, scale=100.0):
        h = C.c_void_p()
        if general:
            check(lib().pt_read_obj_general(obj_path.encode(), model_dir.encode(), sub_dir.encode(), C.c_float(scale), C.byref(h)))
        else:
            check(lib().pt_read_obj(obj_path.encode(), model_dir.encode(), sub_dir.encode(), C.byref(h)))
        try:
            nt, nm, nx = C.c_int(), C.c_int(), C.c_int()
            check(lib().pt_host_scene_counts(h, C.byref(nt), C.byref(nm), C.byref(nx)))
            self.tris = np.ctypeslib.as_array(C.cast(lib().pt_host_scene_triangles(h), C.POINTER(C.c_uint8)),
                                              (nt.value * TRI_DTYPE.itemsize,)).view(TRI_DTYPE).copy()
            self.tri_idx = np.ctypeslib.as_array(C.cast(lib().pt_host_scene_tri_idx(h), C.POINTER(C.c_int32)), (nt.value,)).copy()
            self.mats = np.ctypeslib.as_array(C.cast(lib().pt_host_scene_materials(h), C.POINTER(C.c_uint8)),
                                              (nm.value * MAT_DTYPE.itemsize,)).view(MAT_DTYPE).copy()
            self.mat_names = [lib().pt_host_scene_material_name(h, C.c_int(i)).decode('utf-8', 'replace') for i in range(nm.value)]
            self.texture_requests = []
            for i in range(nx.value):
                path, comps = C.c_char_p(), C.c_int()
                check(lib().pt_host_scene_texture_info(h, C.c_int(i), C.byref(path), C.byref(comps)))
                self.texture_requests.append((path.value.decode(), comps.value))
        finally:
            lib().pt_host_scene_free(h)


def host_build_bvh(tris, flavour=PT_BVH_REFERENCE):
    """buildBVH on host arrays: returns (nodes[:used], tri_idx); tris['origin'] is filled in place."""
    n = len(tris)
    assert tris.dtype == TRI_DTYPE and tris.flags['C_CONTIGUOUS']
    nodes = np.zeros(2 * n - 1 + 1, dtype=NODE_DTYPE)
    idx = np.arange(n, dtype=np.int32)
    used = C.c_int()
    check(lib().pt_host_build_bvh(_p(tris), _p(idx), C.c_int(n), C.c_int(flavour), _p(nodes), C.byref(used)))
    return nodes[:used.value].copy(), idx


def load_hdr(path):
    ptr, w, h = C.c_void_p(), C.c_int(), C.c_int()
    check(lib().pt_load_hdr(path.encode(), C.byref(ptr), C.byref(w), C.byref(h)))
    try:
        a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), (h.value, w.value, 4)).copy()
    finally:
        lib().pt_free(ptr)
    return a


def load_texture_png(path, comps):
    ptr, w, h = C.c_void_p(), C.c_int(), C.c_int()
    check(lib().pt_load_texture_png(path.encode(), C.c_int(comps), C.byref(ptr), C.byref(w), C.byref(h)))
    try:
        a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), (h.value, w.value, comps)).copy()
    finally:
        lib().pt_free(ptr)
    return a


def load_texture_image(path, comps):
    ptr, w, h = C.c_void_p(), C.c_int(), C.c_int()
    check(lib().pt_load_texture_image(path.encode(), C.c_int(comps), C.byref(ptr), C.byref(w), C.byref(h)))
    try:
        a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), (h.value, w.value, comps)).copy()
    finally:
        lib().pt_free(ptr)
    return a


def build_env_cdf(rgba):
    rgba = np.ascontiguousarray(rgba, dtype=np.float32)
    h, w, _ = rgba.shape
    cdf = np.zeros((h, w), dtype=np.float32)
    s = C.c_float()
    check(lib().pt_build_env_cdf(_p(rgba), C.c_int(w), C.c_int(h), _p(cdf), C.byref(s)))
    return cdf, s.value


def camera_basis(params):
    ww, uu, vv = Float3(), Float3(), Float3()
    check(lib().pt_camera_basis(C.byref(params), C.byref(ww), C.byref(uu), C.byref(vv)))
    f = lambda v: np.array([v.x, v.y, v.z], dtype=np.float32)
    return f(ww), f(uu), f(vv)


def save_png(path, rgba):
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    h, w, _ = rgba.shape
    check(lib().pt_save_png(path.encode(), _p(rgba), C.c_int(w), C.c_int(h)))


def save_pfm(path, rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w, _ = rgb.shape
    check(lib().pt_save_pfm(path.encode(), _p(rgb), C.c_int(w), C.c_int(h)))


def save_exr(path, rgb, scale=1.0):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w, _ = rgb.shape
    check(lib().pt_save_exr(path.encode(), _p(rgb), C.c_int(w), C.c_int(h), C.c_float(scale)))


def load_pfm(path):
    ptr, w, h = C.c_void_p(), C.c_int(), C.c_int()
    check(lib().pt_load_pfm(path.encode(), C.byref(ptr), C.byref(w), C.byref(h)))
    try:
        a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), (h.value, w.value, 3)).copy()
    finally:
        lib().pt_free(ptr)
    return a


# ------------------------------------------------------------------ device context
class Renderer:
    """One pt_ctx bound to one GPU."""

    def __init__(self, device=0):
        self.h = C.c_void_p()
        check(lib().pt_create(C.byref(self.h), C.c_int(device)))
        self.params = None

    def close(self):
        if self.h:
            lib().pt_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_handle):
        check(lib().pt_set_stream(self.h, C.c_void_p(stream_handle)))

    def init(self, params):
        self.params = params
        check(lib().pt_init(self.h, C.byref(params)))

    def upload_texture(self, texels):
        t = np.ascontiguousarray(texels, dtype=np.float32)
        if t.ndim == 2:
            t = t[:, :, None]
        h, w, c = t.shape
        handle = C.c_uint64()
        check(lib().pt_upload_texture(self.h, _p(t), C.c_int(w), C.c_int(h), C.c_int(c), C.byref(handle)))
        return handle.value

    def upload_scene(self, tris, mats, tri_idx=None, nodes=None, flavour=PT_BVH_REFERENCE):
        tris = np.ascontiguousarray(tris, dtype=TRI_DTYPE)
        mats = np.ascontiguousarray(mats, dtype=MAT_DTYPE)
        idx = None if tri_idx is None else np.ascontiguousarray(tri_idx, dtype=np.int32)
        check(lib().pt_upload_triangles(self.h, _p(idx), _p(tris), C.c_int(len(tris))))
        check(lib().pt_upload_materials(self.h, _p(mats), C.c_int(len(mats))))
        if nodes is None:
            check(lib().pt_build_bvh(self.h, C.c_int(flavour)))
        else:
            nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
            check(lib().pt_upload_bvh(self.h, _p(nodes), C.c_int(0), C.c_int(len(nodes)), C.c_int(len(tris))))

    def upload_envmap(self, rgba, cdf, total, params=None):
        rgba = np.ascontiguousarray(rgba, dtype=np.float32)
        cdf = np.ascontiguousarray(cdf, dtype=np.float32)
        h, w, _ = rgba.shape
        check(lib().pt_upload_envmap(self.h, _p(rgba), _p(cdf), C.c_int(w), C.c_int(h), C.c_float(total),
                                     C.byref(params) if params is not None else None))

    def set_sky_mode(self, mode):
        check(lib().pt_set_sky_mode(self.h, C.c_int(mode)))

    def set_env_guides(self, on):
        """on: 0 plain bisections, 1 all guide tables, 2 without the whole-range table. Returns the verification
        bits of the uploaded environment map: 1 row/column guides passed, 2 whole-range table passed."""
        ok = C.c_int()
        check(lib().pt_set_env_guides(self.h, C.c_int(int(on)), C.byref(ok)))
        return ok.value

    def set_mode(self, mode):
        check(lib().pt_set_mode(self.h, C.c_int(mode)))

    def set_pool(self, slots, chunk_frames=64):
        check(lib().pt_set_pool(self.h, C.c_int(slots), C.c_int(chunk_frames)))

    def set_frame_lookahead(self, max_frames):
        check(lib().pt_set_frame_lookahead(self.h, C.c_int(max_frames)))

    def wavefront_stats(self):
        it, rp = C.c_uint64(0), C.c_uint64(0)
        check(lib().pt_wavefront_stats(self.h, C.byref(it), C.byref(rp)))
        return {'iterations': it.value, 'drains': rp.value}

    def set_lanes(self, lanes):
        check(lib().pt_set_lanes(self.h, C.c_int(lanes)))

    def set_drain(self, max_rays):
        check(lib().pt_set_drain(self.h, C.c_int(max_rays)))

    def set_extend_tuning(self, refill_lanes, leaf_vote, blocks_per_cu):
        check(lib().pt_set_extend_tuning(self.h, C.c_int(refill_lanes), C.c_int(leaf_vote), C.c_int(blocks_per_cu)))

    def set_node_width(self, width, leaf_merge=2):
        check(lib().pt_set_node_width(self.h, C.c_int(width), C.c_int(leaf_merge)))

    def set_max_bounces(self, n):
        check(lib().pt_set_max_bounces(self.h, C.c_int(n)))

    def enable_counters(self, on=True):
        """False/0 off, True/1 count the reference-order walk, 2 count the production walk over the wide nodes."""
        check(lib().pt_enable_counters(self.h, C.c_int(int(on))))

    def clear_accum(self):
        check(lib().pt_clear_accum(self.h))

    def render(self, params, frame_begin, frame_end, sync=True):
        check(lib().pt_render_headless(self.h, C.byref(params), C.c_int(frame_begin), C.c_int(frame_end)))
        if sync:
            self.sync()

    def render_frame(self, params):
        W, H = params.windowSize.x, params.windowSize.y
        buf = np.zeros((H, W, 4), dtype=np.uint8)
        params.pixelBuffer = buf.ctypes.data
        check(lib().pt_render_frame(self.h, C.byref(params)))
        params.pixelBuffer = None
        return buf

    def render_frame_async(self, params):
        """Queues one frame; the image is fetched later with frame_fetch (at most two frames in flight)."""
        params.pixelBuffer = None
        check(lib().pt_render_frame_async(self.h, C.byref(params)))

    def frame_fetch(self):
        """(RGBA8 image, frame index) of the oldest frame in flight."""
        W, H = self.params.windowSize.x, self.params.windowSize.y
        buf = np.zeros((H, W, 4), dtype=np.uint8)
        idx = C.c_int(-1)
        check(lib().pt_frame_fetch(self.h, buf.ctypes.data_as(C.c_void_p), None, C.byref(idx)))
        return buf, idx.value

    def sync(self):
        check(lib().pt_sync(self.h))

    def read_accum(self):
        W, H = self.params.windowSize.x, self.params.windowSize.y
        out = np.zeros((H, W, 3), dtype=np.float32)
        check(lib().pt_read_accum(self.h, _p(out)))
        return out

    def write_accum(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        check(lib().pt_write_accum(self.h, _p(a)))

    def accum_device_ptr(self):
        p = C.c_void_p()
        check(lib().pt_accum_device_ptr(self.h, C.byref(p)))
        return p.value

    def bind_accum(self, ptr):
        check(lib().pt_bind_accum(self.h, C.c_void_p(ptr)))

    def resolve_display(self, n_frames):
        W, H = self.params.windowSize.x, self.params.windowSize.y
        out = np.zeros((H, W, 4), dtype=np.uint8)
        check(lib().pt_resolve_display(self.h, C.c_int(n_frames), _p(out)))
        return out

    def resolve_display_post(self, n_frames, params):
        W, H = self.params.windowSize.x, self.params.windowSize.y
        out = np.zeros((H, W, 4), dtype=np.uint8)
        check(lib().pt_resolve_display_post(self.h, C.c_int(n_frames), C.byref(params), _p(out)))
        return out

    def trace_closest(self, rays, counted=False):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(len(rays), dtype=HIT_DTYPE)
        if counted:
            nb = np.zeros(len(rays), dtype=np.uint32)
            nt = np.zeros(len(rays), dtype=np.uint32)
            check(lib().pt_trace_closest_counted(self.h, _p(rays), C.c_int(len(rays)), _p(hits), _p(nb), _p(nt)))
            return hits, nb, nt
        check(lib().pt_trace_closest(self.h, _p(rays), C.c_int(len(rays)), _p(hits)))
        return hits

    def trace_closest_fast(self, rays):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(len(rays), dtype=HIT_DTYPE)
        check(lib().pt_trace_closest_fast(self.h, _p(rays), C.c_int(len(rays)), _p(hits)))
        return hits

    def trace_closest_dev(self, rays_ptr, n, hits_ptr):
        check(lib().pt_trace_closest_dev(self.h, C.c_void_p(rays_ptr), C.c_int(n), C.c_void_p(hits_ptr)))

    def trace_closest_fast_dev(self, rays_ptr, n, hits_ptr):
        check(lib().pt_trace_closest_fast_dev(self.h, C.c_void_p(rays_ptr), C.c_int(n), C.c_void_p(hits_ptr)))

    def trace_closest_noalloc_dev(self, rays_ptr, n, hits_ptr):
        check(lib().pt_trace_closest_noalloc_dev(self.h, C.c_void_p(rays_ptr), C.c_int(n), C.c_void_p(hits_ptr)))

    def counters(self):
        c = Counters()
        check(lib().pt_get_counters(self.h, C.byref(c)))
        return {n: getattr(c, n) for n, _ in Counters._fields_}

    def reset_counters(self):
        check(lib().pt_reset_counters(self.h))

    def stage_timing(self, on):
        """Returns (extend_ms, extend_launches) accumulated since the last call, then sets timing on/off."""
        ms, n = C.c_double(), C.c_uint64()
        check(lib().pt_stage_timing(self.h, C.c_int(1 if on else 0), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def stage_times(self):
        """({kernel: ms}, iterations) accumulated since stage timing was switched on (extend, path_hit, a slot not in use, miss, shadow, bsdf)."""
        ms, n = (C.c_double * 6)(), C.c_uint64()
        check(lib().pt_stage_times(self.h, ms, C.byref(n)))
        return dict(zip(('extend', 'path_hit', 'unused', 'miss', 'shadow', 'bsdf'), list(ms))), n.value

    def scene_bytes(self):
        a, b = C.c_uint64(), C.c_uint64()
        check(lib().pt_scene_bytes(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_texture_packing(self, on):
        check(lib().pt_set_texture_packing(self.h, C.c_int(1 if on else 0)))

    def unorm8_probe(self):
        out = np.zeros(256, dtype=np.float32)
        check(lib().pt_unorm8_probe(self.h, _p(out)))
        return out

    def bsdf_probe(self, mats, inside, V, N, L, pixel, frame):
        """Device DisneyEval_2 / DisneySample_2 per item: (eval_f[n,3], eval_pdf[n], L[n,3], f[n,3], pdf[n], inside_after[n])."""
        mats = np.ascontiguousarray(mats, dtype=MAT_DTYPE)
        n = len(mats)
        inside = np.ascontiguousarray(inside, dtype=np.int32)
        V, N, L = (np.ascontiguousarray(a, dtype=np.float32).reshape(n, 3) for a in (V, N, L))
        pixel = np.ascontiguousarray(pixel, dtype=np.uint32)
        out = np.zeros((n, 12), dtype=np.float32)
        check(lib().pt_bsdf_probe(self.h, _p(mats), _p(inside), _p(V), _p(N), _p(L), _p(pixel), C.c_int(frame), C.c_int(n), _p(out)))
        return out[:, 0:3], out[:, 3], out[:, 4:7], out[:, 7:10], out[:, 10], out[:, 11] != 0

    def detmath_probe(self, x, y):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.ascontiguousarray(y, dtype=np.float32)
        out = np.zeros((len(x), 12), dtype=np.float32)
        check(lib().pt_detmath_probe(self.h, _p(x), _p(y), C.c_int(len(x)), _p(out)))
        return out


class Group:
    """pt_group: one Renderer per GPU with the same scene; render() shards frames and reduces onto the first."""

    def __init__(self, renderers):
        self.renderers = list(renderers)
        arr = (C.c_void_p * len(self.renderers))(*[r.h for r in self.renderers])
        self.h = C.c_void_p()
        check(lib().pt_group_create(arr, C.c_int(len(self.renderers)), C.byref(self.h)))

    def render(self, params, frame_begin, frame_end):
        check(lib().pt_group_render(self.h, C.byref(params), C.c_int(frame_begin), C.c_int(frame_end)))

    def close(self):
        if self.h:
            lib().pt_group_destroy(self.h)
            self.h = C.c_void_p()


def group_frames(frame_begin, frame_end, n, r):
    a, b = C.c_int(), C.c_int()
    check(lib().pt_group_frames(C.c_int(frame_begin), C.c_int(frame_end), C.c_int(n), C.c_int(r), C.byref(a), C.byref(b)))
    return a.value, b.value
