"""GPU parity tests at the sizes of the remaining benchmark configurations and of the BSDF lobes the scene
loaders never switch on: the 10-million-triangle soup at 1080p, the atrium at 3840x2160, and materials with
clearcoat, subsurface, sheen and anisotropy set on the uploaded records.  The oracle cannot render these sizes
whole, so the large cases combine size-independent properties (the two kernel organisations agree bit for
bit, frame ranges compose, the fast and the reference-order traversal agree) with oracle evaluations of a
sparse sample of rays and pixels.  Bars as everywhere: closest hits and radiance bit-exact."""
import numpy as np
import pytest

import ptamd
from ptamd import scenes, lib
from ptamd.types import default_materials
import orc
import util
from util import bits

pytestmark = pytest.mark.gpu


def _same_hits(a, b):
    return np.array_equal(a['tri'], b['tri']) and np.array_equal(bits(a['t']), bits(b['t']))


# ------------------------------------------------------------------ BSDF lobes
def _lobe_materials(n, seed):
    """Material records that reach every branch of the Disney model: each family pushes one lobe's weight up."""
    rng = np.random.default_rng(seed)
    m = default_materials(n)
    m['baseColor'] = rng.uniform(0.02, 0.95, (n, 3))
    m['roughness'] = rng.uniform(0.02, 1.0, n)
    m['anisotropic'] = rng.uniform(0.0, 1.0, n) * (rng.uniform(0, 1, n) > 0.4)
    m['metallic'] = rng.uniform(0.0, 1.0, n) * (rng.uniform(0, 1, n) > 0.5)
    m['subsurface'] = rng.uniform(0.0, 1.0, n) * (rng.uniform(0, 1, n) > 0.4)
    m['sheen'] = rng.uniform(0.0, 1.0, n) * (rng.uniform(0, 1, n) > 0.5)
    m['clearcoat'] = rng.uniform(0.0, 1.0, n) * (rng.uniform(0, 1, n) > 0.3)
    m['clearcoatRoughness'] = rng.uniform(0.0, 1.2, n)        # >= 1 takes gtr1's constant branch
    m['specTrans'] = rng.uniform(0.0, 1.0, n) * (rng.uniform(0, 1, n) > 0.6)
    m['ior'] = np.where(rng.uniform(0, 1, n) > 0.85, 1.005, rng.uniform(1.1, 2.2, n))      # some index-matched
    fam = np.arange(n) % 8
    coat = fam == 0       # clearcoat dominates the lobe pick: dark base, no metal, no glass
    m['baseColor'][coat] = 0.01; m['metallic'][coat] = 0.0; m['specTrans'][coat] = 0.0; m['clearcoat'][coat] = 1.0
    sss = fam == 1        # diffuse with full subsurface blend and sheen
    m['metallic'][sss] = 0.0; m['specTrans'][sss] = 0.0; m['subsurface'][sss] = 1.0; m['sheen'][sss] = 1.0; m['clearcoat'][sss] = 0.0
    ani = fam == 2        # anisotropic metal
    m['metallic'][ani] = 1.0; m['anisotropic'][ani] = 0.9; m['roughness'][ani] = 0.3
    gls = fam == 3        # rough glass
    m['metallic'][gls] = 0.0; m['specTrans'][gls] = 1.0; m['ior'][gls] = 1.5
    return m


def test_bsdf_lobes_on_the_device_match_the_oracle():
    """DisneyEval_2 / DisneySample_2 alone (disney.cuh:29-460), item by item on the device against the CPU
    restatement: clearcoat (gtr1, sample_gtr1, eval_clearcoat), the subsurface blend, sheen, anisotropic GGX,
    glass reflection/refraction and the index-matched pass-through all occur in the sample."""
    n = 6000
    rng = np.random.default_rng(77)
    mats = _lobe_materials(n, 5)
    unit = lambda v: (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    N = unit(rng.normal(0, 1, (n, 3)))
    V = unit(rng.normal(0, 1, (n, 3)))
    flip = np.sum(V * N, axis=1) < 0
    V[flip & (np.arange(n) % 5 != 0)] *= -1.0            # most view vectors above the surface, some below
    L = unit(rng.normal(0, 1, (n, 3)))
    inside = (rng.uniform(0, 1, n) > 0.7).astype(np.int32)
    pixel = rng.integers(0, 1 << 22, n).astype(np.uint32)
    frame = 9
    r = ptamd.Renderer()
    ef, ep, sl, sf, sp, si = r.bsdf_probe(mats, inside, V, N, L, pixel, frame)
    r.close()
    sampled_coat = sampled_pass = refracted = 0
    for i in range(n):
        wf, wp = orc.bsdf_eval(mats[i], inside[i], V[i], N[i], L[i])
        wl, wsf, wsp, wsi = orc.bsdf_sample(mats[i], inside[i], V[i], N[i], int(pixel[i]), frame)
        got = np.concatenate([ef[i], [ep[i]], sl[i], sf[i], [sp[i]]]).astype(np.float32)
        want = np.concatenate([wf, [wp], wl, wsf, [wsp]]).astype(np.float32)
        same = (bits(got) == bits(want)) | (np.isnan(got) & np.isnan(want))
        assert same.all() and bool(si[i]) == wsi, (i, got, want, mats[i])
        if wsi != bool(inside[i]):
            refracted += 1
            if abs(float(mats['ior'][i]) - 1.0) <= 0.01:
                sampled_pass += 1
        # the clearcoat lobe was picked when the third draw fell beyond the other lobes: families with a dark base,
        # no metal and no glass and full clearcoat pick it about one time in four
        if i % 8 == 0 and wsp > 0:
            sampled_coat += 1
    assert refracted > 50 and sampled_pass > 5 and sampled_coat > 100
    assert np.count_nonzero(ep > 0) > n // 3


def _set_lobes(mats, seed):
    rng = np.random.default_rng(seed)
    n = len(mats)
    m = mats.copy()
    m['clearcoat'] = rng.uniform(0.3, 1.0, n)
    m['clearcoatRoughness'] = rng.uniform(0.05, 0.6, n)
    m['subsurface'] = rng.uniform(0.2, 1.0, n)
    m['sheen'] = rng.uniform(0.0, 1.0, n)
    m['anisotropic'] = rng.uniform(0.0, 0.9, n)
    return m


@pytest.mark.parametrize('mode', [pytest.param(0, id='wavefront'), pytest.param(1, id='single_kernel')])
def test_clearcoat_subsurface_sheen_anisotropy_radiance(mode):
    """The reference's MTL reader never sets Pc / subsurface (file_reader.h:64-179 ignores them), but Material carries
    them and DisneyEval_2 / DisneySample_2 use them; here they are set on the uploaded records of the Cornell box and
    of the textured atrium, and the accumulators must be the oracle's."""
    for b, P, frames, bounces in ((util.cornell('sky'), scenes.cornell_params(ptamd.default_params(56, 40)), 4, 8),
                                  (util.atrium(detail=0.3, env_size=(64, 32), glossy=0.3, glass=0.2), scenes.atrium_params(ptamd.default_params(72, 40)), 2, 24)):
        mats = _set_lobes(b.mats, 3)
        S = b.oracle(P)
        S.mats = np.ascontiguousarray(mats)
        want, st = orc.render(S, P, 0, frames, bounces)
        r = ptamd.Renderer()
        r.init(P)
        for t in b.textures:
            r.upload_texture(t)
        r.upload_scene(b.tris, mats, b.idx, b.nodes)
        r.upload_envmap(b.env, b.cdf, b.env_sum, P)
        r.set_mode(mode)
        r.set_max_bounces(bounces)
        r.enable_counters(True)
        r.render(P, 0, frames)
        got = r.read_accum()
        ctr = r.counters()
        r.close()
        same = bits(got) == bits(want)
        assert same.all(), '%d of %d accumulator words differ' % ((~same).sum(), same.size)
        assert ctr['rays'] == st['rays'] and ctr['shade_vertices'] == st['shade_vertices']
        # the lobes matter: the image differs from the one without them
        plain, _ = orc.render(b.oracle(P), P, 0, frames, bounces)
        assert not np.array_equal(bits(plain), bits(want))


# ------------------------------------------------------------------ 3840 x 2160 (the size of the sample-partitioned configuration)
def test_4k_render_properties():
    """The 262k-triangle atrium at 3840x2160: wavefront == single kernel, frame ranges compose, a rank's frame range
    of a partitioned render starts from any frame, and a sparse sample of pixels equals the oracle."""
    b = util.atrium()
    W, H = 3840, 2160
    P = scenes.atrium_params(ptamd.default_params(W, H))
    r = b.renderer(P)
    r.render(P, 0, 3)
    wave = r.read_accum()
    assert r.wavefront_stats()['iterations'] > 0
    r.clear_accum()
    r.render(P, 0, 2)
    r.render(P, 2, 3)
    assert np.array_equal(bits(r.read_accum()), bits(wave))
    r.set_mode(1)
    r.clear_accum()
    r.render(P, 0, 3)
    assert np.array_equal(bits(r.read_accum()), bits(wave))
    # what rank 5 of 8 renders in a 4096-spp partition: frames [2560, 2562) here
    r.clear_accum()
    r.render(P, 2560, 2562)
    late_single = r.read_accum()
    r.set_mode(0)
    r.clear_accum()
    r.render(P, 2560, 2562)
    late = r.read_accum()
    r.close()
    assert np.array_equal(bits(late), bits(late_single))
    assert np.isfinite(wave).all() and wave.min() >= 0 and wave.mean() > 0.01
    S = b.oracle(P)
    rng = np.random.default_rng(18)
    for x, y in zip(rng.integers(0, W, 100), rng.integers(0, H, 100)):
        acc = np.zeros(3, np.float32)
        for f in range(3):
            rgb, _ = orc.pixel_sample(S, P, int(x), int(y), f, 512)
            acc = acc + rgb
        assert np.array_equal(bits(acc), bits(wave[y, x])), (x, y)
        acc = np.zeros(3, np.float32)
        for f in (2560, 2561):
            rgb, _ = orc.pixel_sample(S, P, int(x), int(y), f, 512)
            acc = acc + rgb
        assert np.array_equal(bits(acc), bits(late[y, x])), (x, y, 'late')


# ------------------------------------------------------------------ sun/sky, glossy metal and glass, 1080p
def test_sun_sky_glossy_glass_full_size_properties():
    """The fifth benchmark configuration at its full size: the 262k-triangle atrium with 30 % glossy metal and 20 %
    glass materials under the sun/sky model, bounce cap 16, 1920x1080. Wavefront == single kernel, frame ranges
    compose, and a sparse sample of pixels equals the oracle (vertices reach the bsdf stage without a visibility walk
    here, and glass paths toggle the medium flag)."""
    b = util.atrium(glossy=0.3, glass=0.2, sky_mode=1)
    W, H = 1920, 1080
    P = scenes.atrium_params(ptamd.default_params(W, H))
    r = b.renderer(P)
    r.set_max_bounces(16)
    r.render(P, 0, 3)
    wave = r.read_accum()
    r.clear_accum()
    r.render(P, 0, 1)
    r.render(P, 1, 3)
    assert np.array_equal(bits(r.read_accum()), bits(wave))
    r.set_mode(1)
    r.clear_accum()
    r.render(P, 0, 3)
    assert np.array_equal(bits(r.read_accum()), bits(wave))
    r.close()
    assert np.isfinite(wave).all() and wave.min() >= 0 and wave.mean() > 0.01
    S = b.oracle(P)
    rng = np.random.default_rng(28)
    for x, y in zip(rng.integers(0, W, 100), rng.integers(0, H, 100)):
        acc = np.zeros(3, np.float32)
        for f in range(3):
            rgb, _ = orc.pixel_sample(S, P, int(x), int(y), f, 16)
            acc = acc + rgb
        assert np.array_equal(bits(acc), bits(wave[y, x])), (x, y)


# ------------------------------------------------------------------ 10 million triangles
def test_ten_million_triangle_soup():
    """The large-mesh configuration: 10 M random triangles (2.4 GB of nodes and triangle records, beyond every
    cache), tree built on the GPU. Closest hits of the production kernel equal the reference-order kernel on a
    million rays and exhaustive search on a sample; the image at 1080p is the same from both kernel organisations and
    equals the oracle on a sample of pixels; the tree built by the host builder gives the same hits."""
    n = 10_000_000
    tris, mats = scenes.soup(n, seed=2, extent=5000.0, edge=5.0)
    env = scenes.sky_envmap(64, 32)
    env4 = np.concatenate([env, np.zeros(env.shape[:2] + (1,), np.float32)], -1)
    cdf, total = lib.build_env_cdf(env4)
    W, H = 1920, 1080
    P = ptamd.default_params(W, H)
    P.rayOrigin = ptamd.Float3(0.0, 0.0, 9000.0)
    P.isRendering = 1
    r = ptamd.Renderer()
    r.init(P)
    r.upload_scene(tris, mats, None, None, flavour=lib.PT_BVH_GPU_LBVH)
    r.upload_envmap(env4, cdf, total, P)
    rays = scenes.random_rays(1 << 20, seed=44, extent=5200.0)
    slow = r.trace_closest(rays)
    fast = r.trace_closest_fast(rays)
    assert _same_hits(slow, fast)
    assert (fast['tri'] != 0xFFFFFFFF).sum() > len(rays) // 4
    S = orc.Scene(tris, mats)
    S.set_env(env4)
    brute = orc.closest_brute(S, rays[:192])
    assert _same_hits(fast[:192], brute)
    # the render: both organisations, two frames
    r.set_max_bounces(512)
    r.render(P, 0, 2)
    wave = r.read_accum()
    r.set_mode(1)
    r.clear_accum()
    r.render(P, 0, 2)
    assert np.array_equal(bits(r.read_accum()), bits(wave))
    assert np.isfinite(wave).all() and wave.mean() > 0.01
    # the reference tree from the host builder: same hits through both kernels, and it lets the oracle trace paths
    nodes, idx = lib.host_build_bvh(tris)
    r.upload_scene(tris, mats, idx, nodes)
    assert _same_hits(r.trace_closest_fast(rays), fast)
    # and the agglomerative tree with its host-built top, at this size
    r.upload_scene(tris, mats, None, None, flavour=lib.PT_BVH_GPU_PLOC)
    assert _same_hits(r.trace_closest_fast(rays), fast)
    r.close()
    S = orc.Scene(tris, mats, idx, nodes)
    S.set_env(env4)
    rng = np.random.default_rng(28)
    for x, y in zip(rng.integers(0, W, 100), rng.integers(0, H, 100)):
        acc = np.zeros(3, np.float32)
        for f in range(2):
            rgb, _ = orc.pixel_sample(S, P, int(x), int(y), f, 512)
            acc = acc + rgb
        assert np.array_equal(bits(acc), bits(wave[y, x])), (x, y)
