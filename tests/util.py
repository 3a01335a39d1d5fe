"""Shared scene fixtures: each scene is generated once per session under tests/_scratch, loaded through
the product's host loaders, and exposed both as an oracle Scene and as arrays for the GPU context."""
import os
import numpy as np

import ptamd
from ptamd import lib, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRATCH = os.path.join(ROOT, 'tests', '_scratch')
ATRIUM_DETAIL = 1.073       # about 262k triangles

_cache = {}


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


class Bundle:
    """tris/mats/idx/nodes + decoded textures + env, ready for orc.Scene and ptamd.Renderer."""

    def __init__(self, tris, mats, textures=(), env=None, sky_mode=0):
        self.tris = tris
        self.mats = mats
        self.textures = list(textures)      # decoded float arrays, handle = index + 1 (already set in mats)
        self.nodes, self.idx = lib.host_build_bvh(self.tris)
        self.env = env
        self.sky_mode = sky_mode
        if env is not None:
            self.cdf, self.env_sum = lib.build_env_cdf(env)

    def oracle(self, params=None):
        import orc          # the CPU oracle is a checker: nothing on the render path imports it
        S = orc.Scene(self.tris, self.mats, self.idx, self.nodes, textures=self.textures, sky_mode=self.sky_mode)
        if self.env is not None:
            S.env_rgba = np.ascontiguousarray(self.env, np.float32)
            S.env_cdf = self.cdf
            S.env_sum = self.env_sum
        if params is not None:
            S.sun = (params.sunDirection.x, params.sunDirection.y, params.sunDirection.z)
        return S

    def renderer(self, params, device=0, pack_textures=True):
        r = ptamd.Renderer(device)
        r.init(params)
        r.set_texture_packing(pack_textures)
        handles = [r.upload_texture(t) for t in self.textures]
        assert handles == list(range(1, len(handles) + 1))
        r.upload_scene(self.tris, self.mats, self.idx, self.nodes)
        if self.env is not None:
            r.upload_envmap(self.env, self.cdf, self.env_sum, params)
        r.set_sky_mode(self.sky_mode)
        return r


def load_obj_bundle(obj_path, model_dir, env=None, sky_mode=0):
    hs = ptamd.HostScene(obj_path, model_dir)
    textures = [scenes.decode_texture(p, c) for p, c in hs.texture_requests]
    return Bundle(hs.tris, hs.mats, textures, env, sky_mode)


def cornell(env_kind='black'):
    key = ('cornell', env_kind)
    if key not in _cache:
        d = os.path.join(SCRATCH, 'cornell')
        obj = scenes.cornell(d)
        if env_kind == 'black':
            env = lib.load_hdr(os.path.join(d, 'black_4x2.hdr'))
        else:
            p = os.path.join(d, 'sky_64x32.hdr')
            scenes.write_hdr(p, scenes.sky_envmap(64, 32, sun_power=40.0))
            env = lib.load_hdr(p)
        _cache[key] = load_obj_bundle(obj, d + '/', env)
    return _cache[key]


def atrium(detail=ATRIUM_DETAIL, env_size=(1024, 512), glossy=0.0, glass=0.0, sky_mode=0):
    key = ('atrium', detail, env_size, glossy, glass, sky_mode)
    if key not in _cache:
        d = os.path.join(SCRATCH, 'atrium_%g_%g_%g' % (detail, glossy, glass))
        obj = scenes.atrium(d, detail=detail, glossy_fraction=glossy, glass_fraction=glass)
        p = os.path.join(d, 'sky.hdr')
        scenes.write_hdr(p, scenes.sky_envmap(env_size[0], env_size[1]))
        env = lib.load_hdr(p)
        _cache[key] = load_obj_bundle(obj, d + '/', env, sky_mode)
    return _cache[key]


def soup(n, seed=2, extent=5000.0, edge=5.0):
    key = ('soup', n, seed, extent, edge)
    if key not in _cache:
        tris, mats = scenes.soup(n, seed, extent=extent, edge=edge)
        env = scenes.sky_envmap(64, 32)
        env4 = np.concatenate([env, np.zeros(env.shape[:2] + (1,), np.float32)], -1)
        _cache[key] = Bundle(tris, mats, (), env4)
    return _cache[key]
