"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on the same inputs.
Bars: closest hits bit-exact (triangle id and t bits); radiance bit-exact as well, because both sides
evaluate the same pinned arithmetic (the contract allows 1e-4 relative; the tests hold the tighter bar
and report the worst relative error if it is ever missed)."""
import numpy as np
import pytest

import ptamd
from ptamd import scenes, lib
import orc
import util
from util import bits

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-6)))


def test_detmath_bits_match_host():
    rng = np.random.default_rng(7)
    n = 1 << 18
    x = np.concatenate([rng.uniform(-20, 20, n // 2), np.exp(rng.uniform(-30, 30, n // 4)), rng.uniform(-1, 1, n // 4)]).astype(np.float32)
    y = np.concatenate([rng.uniform(-4, 4, n // 2), rng.uniform(-2, 2, n // 2)]).astype(np.float32)
    x[:8] = [0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45]
    r = ptamd.Renderer()
    dev = r.detmath_probe(x, y)
    host = orc.detmath_probe(x, y)
    names = ['sinf', 'cosf', 'expf', 'logf', 'log2f', 'powf', 'acosf', 'atan2f', 'rcp', 'sqrtf', 'div', 'pow(double)']
    for k, nme in enumerate(names):
        a, b = bits(dev[:, k]), bits(host[:, k])
        same = (a == b) | (np.isnan(dev[:, k]) & np.isnan(host[:, k]))
        assert same.all(), '%s differs on %d of %d inputs, first x=%r y=%r' % (nme, (~same).sum(), n, x[~same][0], y[~same][0])
    r.close()


def test_packed_texture_conversion_is_exact():
    """Byte-packed textures rebuild k/255 without a division; all 256 values must equal the IEEE quotient, and a
    render with packing off (fp32 texels) must give the same bits as with packing on."""
    r = ptamd.Renderer()
    got = r.unorm8_probe()
    r.close()
    want = np.arange(256, dtype=np.float32) / np.float32(255.0)
    assert np.array_equal(bits(got), bits(want))
    b = util.atrium(detail=0.3, env_size=(64, 32))
    P = scenes.atrium_params(ptamd.default_params(64, 40))
    out = []
    for packing in (True, False):
        r = ptamd.Renderer()
        r.init(P)
        r.set_texture_packing(packing)
        for t in b.textures:
            r.upload_texture(t)
        r.upload_scene(b.tris, b.mats, b.idx, b.nodes)
        r.upload_envmap(b.env, b.cdf, b.env_sum, P)
        r.set_max_bounces(8)
        r.render(P, 0, 2)
        out.append(r.read_accum())
        r.close()
    assert np.array_equal(bits(out[0]), bits(out[1])) and out[0].sum() > 0


@pytest.mark.parametrize('n_tri', [1, 5, 300, 20000])
def test_closest_hit_bit_exact_vs_brute_force(n_tri):
    extent = {1: 50.0, 5: 50.0, 300: 60.0, 20000: 250.0}[n_tri]
    b = util.soup(n_tri, seed=10 + n_tri, extent=extent, edge=8.0)
    rays = scenes.random_rays(20000, seed=3, extent=extent * 1.1)
    if n_tri <= 5:      # aim at the few triangles so that some rays hit
        c = b.tris['a'][np.arange(len(rays)) % n_tri] + 0.3 * (b.tris['b'] - b.tris['a'])[np.arange(len(rays)) % n_tri]
        d = c - rays['o']
        rays['d'] = d / np.linalg.norm(d, axis=1, keepdims=True)
    P = ptamd.default_params(64, 64)
    r = b.renderer(P)
    got = r.trace_closest(rays)
    want = orc.closest_brute(b.oracle(), rays)
    assert np.array_equal(got['tri'], want['tri'])
    assert np.array_equal(bits(got['t']), bits(want['t']))
    assert (got['tri'] != 0xFFFFFFFF).sum() > 0
    r.close()


def test_closest_hit_and_counters_vs_oracle_traversal_atrium():
    b = util.atrium(detail=0.35)
    P = scenes.atrium_params(ptamd.default_params(64, 64))
    rays = scenes.random_rays(200000, seed=5, extent=1400.0)
    rays['o'][:, 1] = np.abs(rays['o'][:, 1]) * 0.8
    r = b.renderer(P)
    got, nb, nt = r.trace_closest(rays, counted=True)
    want, wb, wt = orc.closest_bvh(b.oracle(), rays, counted=True)
    assert np.array_equal(got['tri'], want['tri'])
    assert np.array_equal(bits(got['t']), bits(want['t']))
    assert np.array_equal(nb, wb) and np.array_equal(nt, wt)
    # a sample against exhaustive search
    sub = rays[:2000]
    brute = orc.closest_brute(b.oracle(), sub)
    assert np.array_equal(got['tri'][:2000], brute['tri']) and np.array_equal(bits(got['t'][:2000]), bits(brute['t']))
    r.close()


def test_non_finite_rays_miss():
    b = util.soup(300, seed=310, extent=60.0, edge=8.0)
    rays = scenes.random_rays(64, seed=1, extent=60.0)
    rays['d'][0] = [np.nan, 0, 1]
    rays['o'][1] = [np.inf, 0, 0]
    rays['d'][2] = [0, 0, 0]
    r = b.renderer(ptamd.default_params(64, 64))
    got = r.trace_closest(rays)
    want = orc.closest_brute(b.oracle(), rays)
    assert got['tri'][0] == 0xFFFFFFFF and got['tri'][1] == 0xFFFFFFFF
    assert np.array_equal(got['tri'], want['tri']) and np.array_equal(bits(got['t']), bits(want['t']))
    r.close()


MODES = [pytest.param(0, id='wavefront'), pytest.param(1, id='single_kernel')]


def _render_both(b, P, f0, f1, bounces, mode=0, pool=None):
    r = b.renderer(P)
    r.set_mode(mode)
    if pool is not None:
        r.set_pool(*pool)
    r.set_max_bounces(bounces)
    r.enable_counters(True)
    r.render(P, f0, f1)
    acc = r.read_accum()
    ctr = r.counters()
    img = r.resolve_display(f1 - f0)
    r.close()
    want, st = orc.render(b.oracle(P), P, f0, f1, bounces)
    return acc, ctr, img, want, st


def _assert_radiance(acc, want):
    same = bits(acc) == bits(want)
    if not same.all():
        bad = (~same).sum()
        raise AssertionError('%d of %d accumulator words differ; worst relative error %g' % (bad, same.size, rel_err(acc, want)))


@pytest.mark.parametrize('mode', MODES)
def test_cornell_radiance_matches_oracle(mode):
    b = util.cornell('black')
    P = scenes.cornell_params(ptamd.default_params(64, 64))
    acc, ctr, img, want, st = _render_both(b, P, 0, 8, 4, mode)
    _assert_radiance(acc, want)
    for k in ('rays', 'samples', 'box_tests', 'tri_tests', 'shade_vertices', 'env_samples', 'tex_taps', 'tex_taps_r'):
        assert ctr[k] == st[k], k
    assert np.array_equal(img, orc.resolve_display(want, 8))
    assert want.sum() > 0


@pytest.mark.parametrize('mode', MODES)
def test_cornell_with_sky_light_long_paths(mode):
    b = util.cornell('sky')
    P = scenes.cornell_params(ptamd.default_params(48, 48))
    acc, ctr, img, want, st = _render_both(b, P, 3, 7, 512, mode)
    _assert_radiance(acc, want)
    assert ctr['rays'] == st['rays']
    assert np.array_equal(img, orc.resolve_display(want, 4))


@pytest.mark.parametrize('mode', MODES)
def test_atrium_radiance_matches_oracle(mode):
    b = util.atrium(detail=0.35, env_size=(256, 128))
    P = scenes.atrium_params(ptamd.default_params(96, 56))
    acc, ctr, img, want, st = _render_both(b, P, 0, 2, 512, mode)
    _assert_radiance(acc, want)
    for k in ('rays', 'box_tests', 'tri_tests', 'shade_vertices', 'env_samples', 'tex_taps', 'tex_taps_r'):
        assert ctr[k] == st[k] and st[k] > 0, k
    assert st['rays'] > 3 * st['samples'] and want.sum() > 0
    assert np.array_equal(img, orc.resolve_display(want, 2))


@pytest.mark.parametrize('mode', MODES)
def test_thin_lens_and_preview_modes(mode):
    b = util.cornell('sky')
    P = scenes.cornell_params(ptamd.default_params(40, 40))
    P.DOF_strenght = 4.0
    P.focalDistance = 300.0
    acc, _, _, want, _ = _render_both(b, P, 0, 3, 6, mode)
    _assert_radiance(acc, want)
    P.DOF_strenght = 0.0
    P.isRendering = 0
    acc, _, _, want, _ = _render_both(b, P, 0, 1, 6, mode)
    _assert_radiance(acc, want)


@pytest.mark.parametrize('mode', MODES)
def test_sun_sky_mode_glossy_dielectric(mode):
    b = util.atrium(detail=0.3, env_size=(64, 32), glossy=0.3, glass=0.2, sky_mode=1)
    P = scenes.atrium_params(ptamd.default_params(64, 40))
    acc, ctr, img, want, st = _render_both(b, P, 0, 2, 16, mode)
    _assert_radiance(acc, want)
    assert ctr['env_samples'] == 0 and st['env_samples'] == 0
    assert ctr['rays'] == st['rays']


@pytest.mark.parametrize('mode', MODES)
def test_index_matched_scattering_medium_in_the_cornell_box(mode):
    """pathtracer.cu:279-295, :432-451: the two boxes of the Cornell box made of an index-matched scattering medium (ior
    1.005, absorption and scattering on the scale of the room). Path rays scatter inside them (phase-function vertices with
    light samples that start inside the medium), visibility walks cross them in several segments and carry a transmittance
    other than (1, 1, 1) from segment to segment and on to the bsdf stage."""
    b = util.cornell('black')
    lo = np.minimum(np.minimum(b.tris['a'], b.tris['b']), b.tris['c']).min(0)
    hi = np.maximum(np.maximum(b.tris['a'], b.tris['b']), b.tris['c']).max(0)
    ext = float((hi - lo).max())
    centre = (b.tris['a'] + b.tris['b'] + b.tris['c']) / 3
    inside = ((centre > lo + 0.02 * ext) & (centre < hi - 0.02 * ext)).all(1) & (b.tris['matIndex'] == 0)
    assert 10 <= inside.sum() <= 24
    mats = np.concatenate([b.mats, b.mats[:1]])
    fog = len(mats) - 1
    mats['specTrans'][fog] = 1.0; mats['ior'][fog] = 1.005; mats['roughness'][fog] = 0.0; mats['G'][fog] = 0.6
    mats['sigmaA'][fog] = np.float32(0.3 / ext) * np.array([1, 2, 4], np.float32)
    mats['sigmaS'][fog] = np.float32(2.0 / ext)
    tris = b.tris.copy()
    tris['matIndex'][inside] = fog
    foggy = util.Bundle(tris, mats, b.textures, b.env, b.sky_mode)
    P = scenes.cornell_params(ptamd.default_params(56, 40))
    acc, ctr, img, want, st = _render_both(foggy, P, 0, 4, 12, mode)
    _assert_radiance(acc, want)
    for k in ('rays', 'samples', 'shade_vertices', 'env_samples'):
        assert ctr[k] == st[k], k
    plain, _ = orc.render(b.oracle(P), P, 0, 4, 12)
    assert st['rays'] > st['env_samples'] + st['shade_vertices']        # visibility walks of more than one segment
    assert want.sum() > 0 and not np.array_equal(want, plain)


@pytest.mark.parametrize('mode', MODES)
def test_emission_maps_on_every_surface(mode):
    """image_loader.h:47-89 / pathtracer.cu:166-170: an emission map replaces the material's emission by its texel, on all
    three channels. The Cornell box with a small emission map on every material (so every vertex of every path carries a
    mapped emission through the queues) against the oracle, bit for bit."""
    b = util.cornell('black')
    mats = b.mats.copy()
    glow = ((np.arange(64, dtype=np.float32).reshape(8, 8) * 37) % 61 / 255.0).astype(np.float32)[:, :, None]
    handle = len(b.textures) + 1
    mats['emissiveTexture'] = handle
    mats['use_mapKe'] = 1
    lit = util.Bundle(b.tris, mats, list(b.textures) + [glow], b.env, b.sky_mode)
    P = scenes.cornell_params(ptamd.default_params(56, 40))
    acc, ctr, img, want, st = _render_both(lit, P, 0, 4, 6, mode)
    _assert_radiance(acc, want)
    assert ctr['tex_taps_r'] == st['tex_taps_r'] and st['tex_taps_r'] >= st['shade_vertices'] > 0
    assert want.sum() > 0


@pytest.mark.parametrize('sky_mode', [pytest.param(0, id='environment_map'), pytest.param(1, id='sun_sky')])
def test_material_table_larger_than_the_kernels_local_copy(sky_mode):
    """The path-hit kernel keeps up to 64 materials and 64 textures in LDS; a scene with more takes the variant that
    reads them from memory. The atrium with its 25 materials listed three times, every triangle moved to one of the three
    copies of its material, is the same scene: same accumulators and counters as with the table of 25."""
    if sky_mode:
        b = util.atrium(detail=0.3, env_size=(64, 32), glossy=0.3, glass=0.2, sky_mode=1)
        P, bounces = scenes.atrium_params(ptamd.default_params(64, 40)), 16
    else:
        b = util.atrium(detail=0.35, env_size=(256, 128))
        P, bounces = scenes.atrium_params(ptamd.default_params(96, 56)), 512
    n = len(b.mats)
    assert n <= 64 < 3 * n
    tris = b.tris.copy()
    tris['matIndex'] = b.tris['matIndex'] + n * (np.arange(len(tris)) % 3)
    big = util.Bundle(tris, np.concatenate([b.mats, b.mats, b.mats]), b.textures, b.env, b.sky_mode)
    assert np.array_equal(big.idx, b.idx)
    out = []
    for bundle in (b, big):
        r = bundle.renderer(P)
        r.set_max_bounces(bounces)
        r.enable_counters(True)
        r.render(P, 0, 2)
        out.append((r.read_accum(), r.counters()))
        r.close()
    assert np.array_equal(bits(out[0][0]), bits(out[1][0])) and out[0][0].sum() > 0
    assert {k: v for k, v in out[0][1].items() if k != 'last_ms'} == {k: v for k, v in out[1][1].items() if k != 'last_ms'}


@pytest.mark.parametrize('pool', [(64, 1), (320, 3), (320, 2), (512, 5), (1 << 14, 64)])
def test_wavefront_pool_size_does_not_change_results(pool):
    """Tiny pools force slot regeneration, retirement and several staging chunks; odd image size
    exercises the row-major sample order."""
    b = util.cornell('sky')
    P = scenes.cornell_params(ptamd.default_params(37, 29))
    acc, ctr, img, want, st = _render_both(b, P, 1, 6, 12, 0, pool)
    _assert_radiance(acc, want)
    for k in ('rays', 'samples', 'box_tests', 'tri_tests', 'shade_vertices', 'env_samples', 'tex_taps', 'tex_taps_r'):
        assert ctr[k] == st[k], k


def test_sample_queues_shrink_when_the_device_is_short_of_memory():
    """The default asks for queue arrays for every sample of a chunk at once (456 bytes each, up to 61 GB). On a
    device that does not have them free the library halves the samples in flight until the arrays fit: more stage
    iterations, the same image."""
    import torch
    b = util.atrium(detail=0.3)
    W, H, frames = 1280, 720, 12                      # 11 million samples: 5 GB of queue arrays
    P = scenes.atrium_params(ptamd.default_params(W, H))
    r = b.renderer(P, 0)
    r.set_max_bounces(6)
    r.render(P, 0, frames)
    want = r.read_accum()
    iters_all = r.wavefront_stats()['iterations']
    r.close()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    hog = torch.empty(max(free - (5 << 29), 0), dtype=torch.uint8, device='cuda')      # leave 2.5 GB: room for the scene, the staging array and a quarter of the queues
    try:
        r = b.renderer(P, 0)
        r.set_max_bounces(6)
        r.render(P, 0, frames)
        got = r.read_accum()
        iters_short = r.wavefront_stats()['iterations']
        r.close()
    finally:
        del hog
        torch.cuda.empty_cache()
    assert np.array_equal(bits(got), bits(want))
    assert iters_short > iters_all                    # the samples went through in several turns


def test_random_configurations_agree_between_the_kernel_organisations():
    """Sixteen random combinations of image size (tiled and row-major sample order), frame range, samples in flight,
    staging chunk, lanes, drain threshold, bounce cap, thin lens, sky mode and scene: the stage kernels over the striped
    queues and the one-kernel-per-sample mode give the same accumulator bits (tools/fuzz_modes.py runs more)."""
    rng = np.random.default_rng(11)
    bundles = [(util.cornell('sky'), scenes.cornell_params), (util.atrium(detail=0.25), scenes.atrium_params),
               (util.atrium(detail=0.25, glossy=0.3, glass=0.3, sky_mode=1), scenes.atrium_params)]
    for case in range(16):
        b, mkparams = bundles[int(rng.integers(len(bundles)))]
        W, H = int(rng.integers(1, 160)), int(rng.integers(1, 100))
        if rng.random() < 0.4:
            W, H = W // 8 * 8 + 8, H // 8 * 8 + 8
        f0, frames = int(rng.integers(0, 5)), int(rng.integers(1, 7))
        pool, chunk = int(rng.choice([64, 256, 700, 4096, 65536, 1 << 27])), int(rng.choice([1, 2, 3, 64]))
        lanes, drain, bounces = int(rng.integers(1, 3)), int(rng.choice([0, 64, 5000, 262144])), int(rng.choice([1, 2, 5, 12]))
        P = mkparams(ptamd.default_params(W, H))
        if rng.random() < 0.3:
            P.DOF_strenght = 2.0
        r = b.renderer(P, 0)
        r.set_max_bounces(bounces); r.set_pool(pool, chunk); r.set_lanes(lanes); r.set_drain(drain)
        r.set_mode(1); r.clear_accum(); r.render(P, f0, f0 + frames); want = r.read_accum()
        r.set_mode(0); r.clear_accum(); r.render(P, f0, f0 + frames); got = r.read_accum()
        r.close()
        assert np.array_equal(bits(got), bits(want)), (case, W, H, f0, frames, pool, chunk, lanes, drain, bounces)


def test_render_frame_matches_reference_frame_loop():
    """pt_render_frame == render(): 1 spp per call, frame 0 overwrites, RGBA8 of accum/(frame+1)."""
    b = util.cornell('sky')
    P = scenes.cornell_params(ptamd.default_params(32, 32))
    r = b.renderer(P)
    r.set_max_bounces(5)
    S = b.oracle(P)
    acc = np.zeros((32, 32, 3), np.float32)
    for f in range(3):
        P.frameIndex = f
        img = r.render_frame(P)
        acc, _ = orc.render(S, P, f, f + 1, 5, accum=acc)
        assert np.array_equal(img, orc.resolve_display(acc, f + 1)), 'frame %d' % f
    P.frameIndex = 0        # a camera move resets the accumulation (main.cpp:45)
    img = r.render_frame(P)
    acc0, _ = orc.render(S, P, 0, 1, 5)
    assert np.array_equal(img, orc.resolve_display(acc0, 1))
    r.close()


def test_frame_ranges_compose():
    """Frames [0,2) then [2,5) give the same accumulator as [0,5)."""
    b = util.cornell('sky')
    P = scenes.cornell_params(ptamd.default_params(32, 32))
    r = b.renderer(P)
    r.set_max_bounces(4)
    r.render(P, 0, 5)
    whole = r.read_accum()
    r.clear_accum()
    r.render(P, 0, 2)
    r.render(P, 2, 5)
    parts = r.read_accum()
    assert np.array_equal(bits(whole), bits(parts))
    r.close()


def test_frames_rendered_ahead_arrive_in_order():
    """pt_render_frame renders 2, 4, ... frames per batch while the caller keeps asking for the next frame of one view
    and adds each to the accumulator when its call comes. Every call must return the image and leave the accumulator
    of the one-frame-per-call loop: through batches, a change of view, a restart at frame 0, a change of the bounce
    cap between calls (staged frames become stale) and a skipped frame index."""
    b = util.atrium(detail=0.3, env_size=(64, 32))
    P = scenes.atrium_params(ptamd.default_params(72, 40))
    plain, ahead = b.renderer(P), b.renderer(P)
    plain.set_frame_lookahead(1)
    ahead.set_frame_lookahead(8)
    for r in (plain, ahead):
        r.set_max_bounces(8)
    script = [('f', k) for k in range(14)] + [('move', 0)] + [('f', k) for k in range(5)] + [('bounces', 5)] + [('f', k) for k in range(5, 9)] + \
             [('f', 11), ('f', 12), ('f', 13), ('f', 0), ('f', 1), ('f', 2)]
    for op, k in script:
        if op == 'move':
            P.rayOrigin.x += 25.0
            continue
        if op == 'bounces':
            plain.set_max_bounces(k); ahead.set_max_bounces(k)
            continue
        P.frameIndex = k
        a = plain.render_frame(P)
        c = ahead.render_frame(P)
        assert np.array_equal(a, c), ('image', k)
        assert np.array_equal(bits(plain.read_accum()), bits(ahead.read_accum())), ('accumulator', k)
    assert ahead.wavefront_stats()['iterations'] < plain.wavefront_stats()['iterations']
    plain.close(); ahead.close()


def test_frames_rendered_ahead_stay_within_one_staging_chunk():
    """A staging chunk smaller than the look-ahead (pt_set_pool's chunk_frames) bounds the batch: every staged frame
    must still be in the staging arrays when its call comes."""
    b = util.cornell('sky')
    P = scenes.cornell_params(ptamd.default_params(40, 24))
    plain, ahead = b.renderer(P), b.renderer(P)
    plain.set_frame_lookahead(1)
    ahead.set_frame_lookahead(8)
    ahead.set_pool(512, 2)
    for r in (plain, ahead):
        r.set_max_bounces(6)
    for k in range(12):
        P.frameIndex = k
        a = plain.render_frame(P)
        c = ahead.render_frame(P)
        assert np.array_equal(a, c), ('image', k)
        assert np.array_equal(bits(plain.read_accum()), bits(ahead.read_accum())), ('accumulator', k)
    plain.close(); ahead.close()


def test_small_pools_repeat_exactly():
    """Slot regeneration near the end of a render (slots retiring while other workgroups still hold unissued
    sample ids) must not depend on scheduling: the same render ten times gives the same bits."""
    b = util.cornell('sky')
    P = scenes.cornell_params(ptamd.default_params(37, 29))
    want, _ = orc.render(b.oracle(P), P, 1, 6, 12)
    r = b.renderer(P)
    r.set_max_bounces(12)
    for rep in range(5):
        for pool in ((320, 2), (512, 5)):
            r.set_pool(*pool)
            r.clear_accum()
            r.render(P, 1, 6)
            _assert_radiance(r.read_accum(), want)
    r.close()


def test_production_traversal_kernel_matches_brute_force():
    """pt_trace_closest_fast runs the persistent-wave kernel of the render path (quantized node boxes, leaf
    votes, lane refill); its hits must be the exact ones."""
    b = util.atrium(detail=0.35)
    P = scenes.atrium_params(ptamd.default_params(64, 64))
    rays = scenes.random_rays(300000, seed=15, extent=1400.0)
    rays['o'][:, 1] = np.abs(rays['o'][:, 1]) * 0.8
    # axis-parallel and almost parallel directions (zero, negative zero and denormal components)
    k = np.arange(len(rays))
    for a in range(3):
        rays['d'][k % 11 == a, a] = 0.0
        rays['d'][k % 11 == a + 3, a] = 1e-41
    rays['d'][k % 11 == 6] = [0.0, -1.0, -0.0]
    rays['d'] /= np.maximum(np.linalg.norm(rays['d'], axis=1, keepdims=True), 1e-30)
    rays['d'][5] = [np.nan, 0, 1]; rays['o'][6] = [np.inf, 0, 0]; rays['d'][7] = [0, 0, 0]
    r = b.renderer(P)
    slow = r.trace_closest(rays)
    fast = r.trace_closest_fast(rays)
    assert np.array_equal(fast['tri'], slow['tri']) and np.array_equal(bits(fast['t']), bits(slow['t']))
    # the other node layouts of the same kernel: binary compact nodes, and wide leaves of up to 1 and 7 triangles
    r.set_node_width(2)
    alt = r.trace_closest_fast(rays)
    assert np.array_equal(alt['tri'], slow['tri']) and np.array_equal(bits(alt['t']), bits(slow['t']))
    for merge in (1, 7):
        r.set_node_width(4, merge)
        r.upload_scene(b.tris, b.mats, b.idx, b.nodes)
        alt = r.trace_closest_fast(rays)
        assert np.array_equal(alt['tri'], slow['tri']) and np.array_equal(bits(alt['t']), bits(slow['t'])), merge
    r.set_node_width(4, 2)
    r.upload_scene(b.tris, b.mats, b.idx, b.nodes)
    brute = orc.closest_brute(b.oracle(), rays[:3000])
    assert np.array_equal(fast['tri'][:3000], brute['tri']) and np.array_equal(bits(fast['t'][:3000]), bits(brute['t']))
    for n in (1, 63, 257):      # ragged sizes
        part = r.trace_closest_fast(rays[:n])
        assert np.array_equal(part['tri'], slow['tri'][:n]) and np.array_equal(bits(part['t']), bits(slow['t'][:n]))
    r.close()


def _flat_floor(n, extent):
    """An n x n grid of quads (2 n^2 triangles) in the plane z = 0 over [-extent, extent]^2."""
    from ptamd.types import TRI_DTYPE
    xs = np.linspace(-extent, extent, n + 1, dtype=np.float32)
    x0, y0 = np.meshgrid(xs[:-1], xs[:-1], indexing='ij')
    x1, y1 = np.meshgrid(xs[1:], xs[1:], indexing='ij')
    z = np.zeros(n * n, np.float32)
    p00 = np.stack([x0.ravel(), y0.ravel(), z], -1); p10 = np.stack([x1.ravel(), y0.ravel(), z], -1)
    p01 = np.stack([x0.ravel(), y1.ravel(), z], -1); p11 = np.stack([x1.ravel(), y1.ravel(), z], -1)
    tris = np.zeros(2 * n * n, dtype=TRI_DTYPE)
    tris['a'][0::2] = p00; tris['b'][0::2] = p10; tris['c'][0::2] = p11
    tris['a'][1::2] = p00; tris['b'][1::2] = p11; tris['c'][1::2] = p01
    for k in ('nA', 'nB', 'nC'):
        tris[k] = [0.0, 0.0, 1.0]
    return tris


@pytest.mark.parametrize('extent', [1.0, 700.0])
def test_thin_boxes_seen_from_far_away(extent):
    """The wide boxes of a flat floor are a few rounding errors thick, while the rounding of a ray's plane distances
    grows with the distance travelled: from 50, 100 and 1000 scene extents away entry and exit of such a box round to
    the same number. The production kernel must still reach the triangles (the hits of exhaustive search), for rays straight down the axis and for oblique ones."""
    tris = _flat_floor(8, extent)
    b = util.Bundle(tris, ptamd.default_materials(1))
    rng = np.random.default_rng(41)
    from ptamd.types import RAY_DTYPE
    parts = []
    for far in (2.0, 50.0, 100.0, 1000.0, 30000.0):
        n = 4000
        rays = np.zeros(n, dtype=RAY_DTYPE)
        target = np.zeros((n, 3), np.float32)
        target[:, :2] = rng.uniform(-extent, extent, (n, 2))
        o = target.copy()
        o[:, 2] = far * extent
        # a third straight down, a third oblique from above, a third from far off to the side at a grazing angle
        o[n // 3:2 * n // 3, :2] += rng.uniform(-far * extent, far * extent, (n // 3, 2)).astype(np.float32)
        # (grazing rays only up to 100 extents: further out the triangle test's own rounding, o - a at 10^6 units against
        # edges of a quarter unit, accepts triangles the ray passes a sixth of an edge away from, and exact-t ties with
        # such neighbours are beyond what any box can promise, the reference's padded ones included)
        o[2 * n // 3:, 0] += min(far, 100.0) * extent * 40.0
        d = target.astype(np.float64) - o.astype(np.float64)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        d[:n // 3] = [0.0, 0.0, -1.0]
        rays['o'] = o; rays['d'] = d
        parts.append(rays)
    rays = np.concatenate(parts)
    r = b.renderer(ptamd.default_params(64, 64))
    fast = r.trace_closest_fast(rays)
    brute = orc.closest_brute(b.oracle(), rays)
    assert (brute['tri'] != 0xFFFFFFFF).mean() > 0.6
    # (the reference-order walk is not held to this: boxIntersectF itself drops a box whose padded slab, 0.2 units,
    # falls below one ulp of t, intersectors.cuh:59-71, so the reference loses these hits from 10^7 units away)
    lost = (fast['tri'] != brute['tri']) | (bits(fast['t']) != bits(brute['t']))
    assert not lost.any(), 'production walk differs on %d rays, first origin %r' % (lost.sum(), rays['o'][lost][0])
    # the same floor through the trees built on the device (their wide boxes carry the same margin)
    for flavour in (lib.PT_BVH_GPU_LBVH, lib.PT_BVH_GPU_PLOC):
        r.upload_scene(b.tris, b.mats, None, None, flavour=flavour)
        alt = r.trace_closest_fast(rays)
        assert np.array_equal(alt['tri'], brute['tri']) and np.array_equal(bits(alt['t']), bits(brute['t'])), flavour
    r.close()


def test_committed_golden_vectors():
    """The HIP path against tests/golden/vectors.npz (no oracle run involved)."""
    import os
    from make_golden import bunny_bundle, cornell_bundle
    from ptamd.types import RAY_DTYPE
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'vectors.npz'))
    bb = bunny_bundle()
    assert np.array_equal(bb.nodes, g['bunny_nodes'].view(ptamd.NODE_DTYPE).ravel())
    rays = g['bunny_rays'].view(RAY_DTYPE).ravel()
    r = bb.renderer(ptamd.default_params(16, 16))
    for h in (r.trace_closest(rays), r.trace_closest_fast(rays)):
        assert np.array_equal(h['tri'], g['bunny_hit_tri']) and np.array_equal(bits(h['t']), g['bunny_hit_t_bits'])
    probe = r.detmath_probe(g['probe_x'], g['probe_y'])
    assert np.array_equal(bits(probe), g['probe_bits'])
    r.close()
    cb, P = cornell_bundle()
    for mode in (0, 1):
        r = cb.renderer(P)
        r.set_mode(mode)
        r.set_max_bounces(int(g['cornell_bounces']))
        r.enable_counters(True)
        r.render(P, 0, int(g['cornell_frames']))
        assert np.array_equal(bits(r.read_accum()), g['cornell_accum_bits'])
        assert r.counters()['rays'] == int(g['cornell_rays'])
        assert np.array_equal(r.resolve_display(int(g['cornell_frames'])), g['cornell_rgba'])
        r.close()


def test_full_size_render_properties():
    """At the benchmark's size (1920x1080, the 262k-triangle atrium) the oracle is too slow to run whole, so the
    test uses properties that do not depend on size: the two kernel organisations agree bit for bit, frame ranges
    compose, and a band of rows equals the oracle's render of those rows."""
    b = util.atrium()
    W, H = 1920, 1080
    P = scenes.atrium_params(ptamd.default_params(W, H))
    r = b.renderer(P)
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
    # 25 million samples: lanes start new samples many times over, both lanes run, the last paths are drained
    r.clear_accum()
    r.render(P, 0, 12)
    single = r.read_accum()
    r.set_mode(0)
    r.clear_accum()
    r.render(P, 0, 12)
    assert np.array_equal(bits(r.read_accum()), bits(single))
    assert r.wavefront_stats()['drains'] >= 1
    r.close()
    assert np.isfinite(wave).all() and wave.min() >= 0 and wave.mean() > 0.01
    # oracle on a sparse sample of pixels (a few hundred single-pixel evaluations)
    S = b.oracle(P)
    rng = np.random.default_rng(8)
    for x, y in zip(rng.integers(0, W, 150), rng.integers(0, H, 150)):
        acc = np.zeros(3, np.float32)
        for f in range(3):
            rgb, _ = orc.pixel_sample(S, P, int(x), int(y), f, 512)
            acc = acc + rgb
        assert np.array_equal(bits(acc), bits(wave[y, x])), (x, y)


def test_post_process_and_checkpoint_resume(tmp_path):
    """The window's post-process stage matches its restatement byte for byte, and a render resumed from a saved
    accumulator (PFM) equals the uninterrupted one."""
    import os
    b = util.cornell('sky')
    P = scenes.cornell_params(ptamd.default_params(40, 24))
    r = b.renderer(P)
    r.set_max_bounces(6)
    r.render(P, 0, 6)
    whole = r.read_accum()
    P.exposure = 1.3; P.gamma = 0.8; P.saturation = 1.25; P.mult = ptamd.Float3(1.0, 0.9, 0.8)
    img = r.resolve_display_post(6, P)
    assert np.array_equal(img, orc.post_process(orc.resolve_display(whole, 6), P))
    r.clear_accum()
    r.render(P, 0, 2)
    path = os.path.join(str(tmp_path), 'ckpt.pfm')
    lib.save_pfm(path, r.read_accum())
    r.close()
    r2 = b.renderer(P)
    r2.set_max_bounces(6)
    r2.write_accum(lib.load_pfm(path))
    r2.render(P, 2, 6)
    assert np.array_equal(bits(r2.read_accum()), bits(whole))
    r2.close()


def test_headless_front_end_end_to_end(tmp_path):
    """ptamd_render (C++ on the C-ABI: model, textures, BVH, environment, frames, PNG) against the oracle on the
    atrium with its texture maps, in one headless call and in per-frame calls with a checkpoint in between."""
    import os
    import subprocess
    exe = os.path.join(util.ROOT, 'cuda-pathtracer_amd', 'ptamd_render')
    b = util.atrium(detail=0.2, env_size=(64, 32))
    d = os.path.join(util.SCRATCH, 'atrium_0.2_0_0')
    P = scenes.atrium_params(ptamd.default_params(48, 32))
    want, _ = orc.render(b.oracle(P), P, 0, 4, 16)
    o, a = P.rayOrigin, P.cameraAngle
    common = [exe, '--obj', os.path.join(d, 'atrium.obj'), '--dir', d + '/', '--hdr', os.path.join(d, 'sky.hdr'), '--size', '48x32', '--bounces', '16',
              '--origin', '%r,%r,%r' % (o.x, o.y, o.z), '--angle', '%r,%r' % (a.x, a.y), '--fov', repr(P.fov)]
    png, pfm = os.path.join(str(tmp_path), 'a.png'), os.path.join(str(tmp_path), 'a.pfm')
    r = subprocess.run(common + ['--spp', '4', '--out', png, '--pfm', pfm], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(bits(lib.load_pfm(pfm)), bits(want))
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(png)), orc.resolve_display(want, 4))
    # two frames per-frame like the application, checkpoint, two more frames resumed
    ck = os.path.join(str(tmp_path), 'ck.pfm')
    r = subprocess.run(common + ['--spp', '2', '--frames-per-call', '1', '--out', png, '--pfm', ck], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(common + ['--spp', '2', '--resume', ck, '--resume-frames', '2', '--out', png, '--pfm', pfm], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(bits(lib.load_pfm(pfm)), bits(want))
    assert np.array_equal(np.asarray(Image.open(png)), orc.resolve_display(want, 4))
    # the trees built on the GPU give the same image
    for tree in ('lbvh', 'ploc'):
        r = subprocess.run(common + ['--spp', '4', '--tree', tree, '--out', png, '--pfm', pfm], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert 'built on the GPU' in r.stdout
        assert np.array_equal(bits(lib.load_pfm(pfm)), bits(want)), tree


@pytest.mark.parametrize('mode', MODES)
def test_degenerate_sizes_and_scenes(mode):
    """One-pixel and odd images, a scene of a single triangle (the tree is one leaf), an empty frame range, an
    empty ray batch."""
    one = util.soup(1, seed=11, extent=50.0, edge=30.0)
    P = ptamd.default_params(1, 1)
    P.isRendering = 1
    P.rayOrigin = ptamd.Float3(float(one.tris['a'][0][0]), float(one.tris['a'][0][1]), float(one.tris['a'][0][2]) + 200.0)
    for W, H in ((1, 1), (3, 5), (9, 2)):
        P.windowSize = ptamd.Int2(W, H)
        acc, ctr, img, want, st = _render_both(one, P, 0, 3, 8, mode)
        _assert_radiance(acc, want)
        assert ctr['rays'] == st['rays'] and ctr['samples'] == W * H * 3
    r = one.renderer(P)
    r.set_mode(mode)
    r.render(P, 0, 2)
    before = r.read_accum()
    r.render(P, 2, 2)                       # empty range: nothing changes
    assert np.array_equal(bits(r.read_accum()), bits(before))
    assert len(r.trace_closest(np.zeros(0, dtype=ptamd.RAY_DTYPE))) == 0
    with pytest.raises(ptamd.PtError):
        r.render(P, 3, 2)
    r.close()


def test_guided_environment_search_is_the_bisection():
    """sampleEnvMap's two bisections are answered from guide tables (one over the whole range of the random number,
    row and column tables behind it) when the upload-time check of every breakpoint passed; renders with all
    tables, without the whole-range table and with plain bisections must agree bit for bit, on maps of very
    different shape."""
    b = util.cornell('sky')
    P = scenes.cornell_params(ptamd.default_params(64, 48))
    rng = np.random.default_rng(3)
    onehot = np.zeros((50, 100, 4), np.float32); onehot[7, 93, :3] = 1000.0; onehot[30, 2, :3] = 1e-3
    maps = [b.env,
            np.concatenate([rng.uniform(0, 2, (64, 128, 3)).astype(np.float32) ** 6, np.zeros((64, 128, 1), np.float32)], -1),
            np.concatenate([np.full((17, 33, 3), 0.5, np.float32), np.zeros((17, 33, 1), np.float32)], -1),
            onehot]
    for env in maps:
        cdf, total = lib.build_env_cdf(env)
        out = []
        for on in (1, 2, 0):
            r = ptamd.Renderer(); r.init(P)
            r.upload_scene(b.tris, b.mats, b.idx, b.nodes)
            r.upload_envmap(env, cdf, total, P)
            verified = r.set_env_guides(on)
            r.set_max_bounces(6)
            r.render(P, 0, 3)
            out.append(r.read_accum())
            r.close()
        assert verified == 3, 'the guide tables of this map did not pass the check'
        assert np.array_equal(bits(out[0]), bits(out[2])) and np.array_equal(bits(out[1]), bits(out[2]))
    S = b.oracle(P)
    S.env_rgba = np.ascontiguousarray(maps[1]); S.env_cdf, S.env_sum = lib.build_env_cdf(maps[1])
    want, _ = orc.render(S, P, 0, 3, 6)
    r = ptamd.Renderer(); r.init(P)
    r.upload_scene(b.tris, b.mats, b.idx, b.nodes)
    r.upload_envmap(maps[1], S.env_cdf, S.env_sum, P)
    r.set_max_bounces(6)
    r.render(P, 0, 3)
    _assert_radiance(r.read_accum(), want)
    r.close()


def test_one_and_two_lanes_agree_and_odd_frame_counts():
    """Two concurrent lanes own alternate 64-pixel blocks of the image; one lane, two lanes, small pools and several
    staging chunks all give the oracle's accumulator."""
    b = util.atrium(detail=0.3, env_size=(64, 32))
    P = scenes.atrium_params(ptamd.default_params(72, 40))
    want, st = orc.render(b.oracle(P), P, 2, 9, 12)
    for lanes, pool in ((1, (4096, 3)), (2, (4096, 3)), (2, (1 << 20, 64)), (2, (700, 1))):
        r = b.renderer(P)
        r.set_lanes(lanes)
        r.set_pool(*pool)
        r.set_max_bounces(12)
        r.enable_counters(True)
        r.render(P, 2, 9)
        _assert_radiance(r.read_accum(), want)
        assert r.counters()['rays'] == st['rays'] and r.counters()['samples'] == st['samples']
        r.close()
    # an image whose pixel count is not a multiple of the 64-pixel blocks the lanes deal out, and a one-block image
    c = util.cornell('sky')
    for W, H in ((37, 29), (7, 5)):
        Pc = scenes.cornell_params(ptamd.default_params(W, H))
        want, st = orc.render(c.oracle(Pc), Pc, 0, 5, 6)
        r = c.renderer(Pc)
        r.set_lanes(2)
        r.set_pool(512, 2)
        r.set_max_bounces(6)
        r.render(Pc, 0, 5)
        _assert_radiance(r.read_accum(), want)
        r.close()


def test_run_down_of_the_queues_gives_the_same_image():
    """Once the global hand-out of sample ids has run dry, the ids that workgroups still hold in their chunks are
    started all at once and the queues then thin out until they are empty (drain kernel off) or short enough for
    the drain kernel. The accumulator and every counter must be those of the oracle, on one lane and on two, for
    pools far larger than the render, far smaller, and smaller than one staging chunk's frame."""
    b = util.atrium(detail=0.3, env_size=(64, 32))
    P = scenes.atrium_params(ptamd.default_params(72, 40))
    want, st = orc.render(b.oracle(P), P, 0, 6, 24)
    for lanes, pool, drain in ((1, (1 << 20, 64), 0), (2, (1 << 20, 64), 0), (1, (2048, 64), 0), (2, (4096, 2), 64), (1, (256, 64), 0), (1, (768, 64), 300)):
        r = b.renderer(P)
        r.set_lanes(lanes)
        r.set_pool(*pool)
        r.set_drain(drain)
        r.set_max_bounces(24)
        r.enable_counters(True)
        r.render(P, 0, 6)
        _assert_radiance(r.read_accum(), want)
        ctr = r.counters()
        for k in ('rays', 'samples', 'box_tests', 'tri_tests', 'shade_vertices', 'env_samples'):
            assert ctr[k] == st[k], (k, lanes, pool, drain)
        assert (r.wavefront_stats()['drains'] >= 1) == (drain > 0), (lanes, pool, drain)
        r.close()


@pytest.mark.parametrize('flavour', [pytest.param(lib.PT_BVH_GPU_LBVH, id='radix_tree'), pytest.param(lib.PT_BVH_GPU_PLOC, id='agglomerative')])
def test_gpu_built_tree_gives_the_same_image(flavour):
    """PT_BVH_GPU_LBVH: a Morton-order radix tree built on the device; PT_BVH_GPU_PLOC: clusters in Morton order merged
    with their nearest neighbours. Closest hits and therefore radiance do not depend on the tree; only traversal cost
    and the box/triangle test counters do."""
    b = util.atrium(detail=0.35, env_size=(256, 128))
    P = scenes.atrium_params(ptamd.default_params(96, 56))
    want, st = orc.render(b.oracle(P), P, 0, 2, 32)
    rays = scenes.random_rays(100000, seed=21, extent=1400.0)
    brute = orc.closest_brute(b.oracle(), rays[:3000])
    r = b.renderer(P)
    r.upload_scene(b.tris, b.mats, None, None, flavour=flavour)
    slow, fast = r.trace_closest(rays), r.trace_closest_fast(rays)
    assert np.array_equal(slow['tri'], fast['tri']) and np.array_equal(bits(slow['t']), bits(fast['t']))
    assert np.array_equal(slow['tri'][:3000], brute['tri']) and np.array_equal(bits(slow['t'][:3000]), bits(brute['t']))
    r.set_node_width(2)
    fast2 = r.trace_closest_fast(rays)
    assert np.array_equal(slow['tri'], fast2['tri']) and np.array_equal(bits(slow['t']), bits(fast2['t']))
    r.set_node_width(4)
    for mode in (0, 1):
        r.set_mode(mode)
        r.set_max_bounces(32)
        r.enable_counters(True); r.reset_counters()
        r.clear_accum()
        r.render(P, 0, 2)
        _assert_radiance(r.read_accum(), want)
        c = r.counters()
        assert c['rays'] == st['rays'] and c['shade_vertices'] == st['shade_vertices'] and c['tex_taps'] == st['tex_taps']
    r.close()
    # degenerate inputs: one triangle, two, many coincident (64 and more are built and laid out on the device)
    for n in (1, 2, 64, 300):
        s = util.soup(n, seed=40 + n, extent=20.0, edge=30.0)
        tris = s.tris.copy()
        if n >= 64:
            tris[:] = tris[0]
        rr = scenes.random_rays(5000, seed=2, extent=60.0)
        r = ptamd.Renderer(); r.init(ptamd.default_params(16, 16))
        r.upload_scene(tris, s.mats, None, None, flavour=flavour)
        got = r.trace_closest_fast(rr)
        ref = orc.closest_brute(orc.Scene(tris, s.mats), rr)
        assert np.array_equal(got['tri'], ref['tri']) and np.array_equal(bits(got['t']), bits(ref['t'])), n
        r.close()


def test_device_side_layouts_equal_the_host_flattening_of_the_same_tree():
    """pt_build_bvh(PT_BVH_GPU_LBVH) leaves binary, quantized and 4-wide nodes and the triangle records on the device
    without a round trip; PT_BVH_GPU_LBVH_VIA_HOST takes the same radix tree through pt_upload_bvh's host code.
    Same tree, same boxes: hits, and the test counts of the reference-order walk and of the production walk, agree
    exactly (the node numbering differs, which no count sees)."""
    Ps = ptamd.default_params(64, 48); Ps.rayOrigin = ptamd.Float3(0.0, 0.0, 700.0); Ps.isRendering = 1
    cases = [(util.atrium(detail=0.35, env_size=(64, 32)), 1400.0, scenes.atrium_params(ptamd.default_params(64, 48))),
             (util.soup(300000, seed=9, extent=400.0, edge=6.0), 450.0, Ps)]
    for b, extent, P in cases:
        rays = scenes.random_rays(200000, seed=5, extent=extent)
        seen = []
        for flavour in (lib.PT_BVH_GPU_LBVH, lib.PT_BVH_GPU_LBVH_VIA_HOST):
            r = b.renderer(P)       # textures and environment, then the scene again with the tree under test
            r.upload_scene(b.tris, b.mats, None, None, flavour=flavour)
            slow = r.trace_closest(rays)
            fast = r.trace_closest_fast(rays)
            assert np.array_equal(slow['tri'], fast['tri']) and np.array_equal(bits(slow['t']), bits(fast['t']))
            counts = []
            for mode in (1, 2):
                r.enable_counters(mode); r.reset_counters(); r.clear_accum()
                r.render(P, 0, 2)
                c = r.counters()
                counts.append((c['rays'], c['box_tests'], c['tri_tests'], c['wide_nodes'], c['wide_tris']))
            r.enable_counters(0)
            image = r.read_accum()
            for nw in (2, 4):       # the quantized binary nodes, then the wide ones again
                r.set_node_width(nw)
                again = r.trace_closest_fast(rays)
                assert np.array_equal(again['tri'], slow['tri']) and np.array_equal(bits(again['t']), bits(slow['t']))
            seen.append((slow, counts, r.scene_bytes(), image))
            r.close()
        (h0, c0, b0, i0), (h1, c1, b1, i1) = seen
        assert np.array_equal(h0['tri'], h1['tri']) and np.array_equal(bits(h0['t']), bits(h1['t']))
        assert np.array_equal(bits(i0), bits(i1))
        assert c0 == c1, (c0, c1)
        assert c0[0][1] > 0 and c0[1][3] > 0 and b0 == b1, (c0, b0, b1)


def test_async_frames_equal_the_blocking_frame_loop():
    """pt_render_frame_async / pt_frame_fetch (display image to pinned host memory on a copy stream, two frames in
    flight) hand out, in order, exactly the images of the blocking pt_render_frame loop, and leave the same accumulator."""
    b = util.cornell('sky')
    P = scenes.cornell_params(ptamd.default_params(96, 64))
    r = b.renderer(P)
    r.set_max_bounces(6)
    n = 11
    want = []
    for f in range(n):
        P.frameIndex = f
        want.append(r.render_frame(P))
    acc = r.read_accum()
    got = []
    for f in range(n):
        P.frameIndex = f
        r.render_frame_async(P)
        if f >= 1:
            img, idx = r.frame_fetch()
            assert idx == f - 1
            got.append(img)
    img, idx = r.frame_fetch()
    assert idx == n - 1
    got.append(img)
    for f in range(n):
        assert np.array_equal(got[f], want[f]), f
    assert np.array_equal(bits(r.read_accum()), bits(acc))
    with pytest.raises(ptamd.PtError, match='no frame in flight'):
        r.frame_fetch()
    P.frameIndex = 0
    r.render_frame_async(P); P.frameIndex = 1; r.render_frame_async(P); P.frameIndex = 2
    with pytest.raises(ptamd.PtError, match='two frames are in flight'):
        r.render_frame_async(P)
    assert r.frame_fetch()[1] == 0 and r.frame_fetch()[1] == 1
    assert np.array_equal(r.render_frame(P), want[2])
    r.close()


def test_group_of_one_gpu_and_group_errors():
    """pt_group_render on one GPU is pt_render_headless; two contexts on one GPU are refused (a group takes one
    context per GPU; the RCCL reduce between GPUs cannot be exercised on a one-GPU machine)."""
    b = util.cornell('sky')
    P = scenes.cornell_params(ptamd.default_params(40, 24))
    want, _ = orc.render(b.oracle(P), P, 2, 7, 6)
    r = b.renderer(P)
    r.set_max_bounces(6)
    g = lib.Group([r])
    g.render(P, 2, 4)
    g.render(P, 4, 7)           # calls compose onto the first context's accumulator
    _assert_radiance(r.read_accum(), want)
    g.close()
    r2 = b.renderer(P)
    with pytest.raises(ptamd.PtError, match='share GPU'):
        lib.Group([r, r2])
    r.close(); r2.close()


def test_group_of_two_ranks_through_a_stand_in_for_rccl():
    """The n > 1 branch of pt_group_render on a one-GPU machine: two contexts on GPU 0 (PTAMD_GROUP_SHARED_GPU) and
    tests/fake_rccl.c in place of RCCL (PTAMD_RCCL_LIBRARY), in a process of its own because the library reads both
    once. Exercised: ncclCommInitAll over the group's devices, one host thread per context, the frame split, the
    clearing of the other ranks' accumulators, the grouped in-place reduce onto context 0 behind each context's
    render, and destruction after the contexts. The sum is fp32 rank 0 + rank 1, so it equals the two parts added
    in numpy bit for bit, and the one-GPU render of the whole range to rounding."""
    import os
    import subprocess
    import sys
    import textwrap
    fake = os.path.join(util.ROOT, 'tests', 'libfake_rccl.so')
    assert os.path.exists(fake), 'tests/libfake_rccl.so is missing: __graft_entry__.build() compiles it'
    code = textwrap.dedent('''
        import ctypes, numpy as np, sys
        import torch      # first: its HIP runtime then serves the library and the stand-in too
        import ptamd, util
        from ptamd import scenes, lib
        b = util.cornell('sky')
        P = scenes.cornell_params(ptamd.default_params(40, 24))
        rs = [b.renderer(P) for _ in range(3)]
        for r in rs: r.set_max_bounces(6)
        a, c = lib.group_frames(2, 9, 2, 0), lib.group_frames(2, 9, 2, 1)
        assert a == (2, 5) and c == (5, 9)
        rs[2].render(P, 2, 9); whole = rs[2].read_accum()
        rs[2].clear_accum(); rs[2].render(P, *a); part0 = rs[2].read_accum()
        rs[2].clear_accum(); rs[2].render(P, *c); part1 = rs[2].read_accum()
        rs[1].render(P, 0, 1)                      # stale contents on rank 1: the group must clear them
        g = lib.Group(rs[:2])
        g.render(P, 2, 9)
        got = rs[0].read_accum()
        assert np.array_equal(got.view(np.uint32), (part0 + part1).view(np.uint32)), 'reduce is not rank 0 + rank 1'
        assert np.max(np.abs(got - whole) / np.maximum(np.abs(whole), 1e-6)) <= 1e-4
        assert np.array_equal(rs[1].read_accum().view(np.uint32), part1.view(np.uint32))      # in place: the other ranks keep their part
        g.render(P, 9, 11)                          # a second call adds onto context 0 and starts the others from zero again
        rs[2].clear_accum(); rs[2].render(P, 2, 11); whole2 = rs[2].read_accum()
        got2 = rs[0].read_accum()
        assert np.max(np.abs(got2 - whole2) / np.maximum(np.abs(whole2), 1e-6)) <= 1e-4
        fake = ctypes.CDLL(sys.argv[1])
        nr, ng = ctypes.c_int(), ctypes.c_int()
        fake.fake_rccl_calls(ctypes.byref(nr), ctypes.byref(ng))
        assert (nr.value, ng.value) == (4, 2), (nr.value, ng.value)
        for r in rs: r.close()                      # contexts first, the group after them
        g.close()
        print('group ok')
    ''')
    env = dict(os.environ, PTAMD_RCCL_LIBRARY=fake, PTAMD_GROUP_SHARED_GPU='1',
               PYTHONPATH=os.pathsep.join([os.path.join(util.ROOT, 'cuda-pathtracer_amd', 'python'), os.path.join(util.ROOT, 'tests')]))
    p = subprocess.run([sys.executable, '-c', code, fake], capture_output=True, text=True, timeout=200, env=env)
    assert p.returncode == 0 and 'group ok' in p.stdout, p.stdout[-1000:] + p.stderr[-3000:]


@pytest.mark.timeout(600)      # three bench processes, each importing torch afresh
def test_bench_two_ranks_on_one_gpu_match_one_rank():
    """bench.py itself under torch.distributed.run with two ranks sharing the GPU (gloo backend; RCCL needs one GPU per
    rank): the sample partition, the reduce onto rank 0 and the JSON line. With a fixed total sample count (strong
    scaling) two ranks render the frames one rank renders, so the reduced image agrees to rounding."""
    import json
    import os
    import socket
    import subprocess
    import sys
    bench = os.path.join(util.ROOT, 'bench.py')
    common = ['--steps', '1', '--warmup', '0', '--no-cpu', '--width', '160', '--height', '96', '--detail', '0.3', '--device', '0', '--bounces', '8']

    def run(n, extra):
        if n == 1:
            cmd = [sys.executable, bench, '--gpus', '1'] + common + extra
        else:
            with socket.socket() as s:
                s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
            cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1', '--master-port', str(port),
                   bench, '--gpus', str(n), '--backend', 'gloo'] + common + extra
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=280)
        assert p.returncode == 0, p.stderr[-2000:]
        return json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][-1])

    one = run(1, ['--total-spp', '6'])
    two = run(2, ['--total-spp', '6'])
    assert one['scaling'] == two['scaling'] == 'strong' and two['n_gpus'] == 2
    assert one['image_mean'] > 0 and abs(two['image_mean'] - one['image_mean']) <= 1e-4 * one['image_mean']
    assert two['value'] > 0 and two['roofline']['own_walk_per_ray']['wide_nodes'] > 0
    # launched plainly, without torch.distributed.run around it: the script starts its own ranks
    p = subprocess.run([sys.executable, bench, '--gpus', '2', '--backend', 'gloo', '--spp', '3'] + common, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    weak = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][-1])
    assert weak['scaling'] == 'weak' and weak['n_gpus'] == 2 and weak['image_mean'] > 0
    assert [k['kernel'] for k in weak['roofline']['kernels']] == ['wf_extend_kernel', 'wf_pathhit_kernel', 'wf_miss_kernel', 'wf_shadow_kernel', 'wf_bsdf_kernel']
    assert all(k['ms_per_step'] > 0 and k['requested_bytes_per_launch'] > 0 for k in weak['roofline']['kernels'])
    # weak mode: 3 spp per rank = frames [0, 6) in all, the same image again
    assert abs(weak['image_mean'] - one['image_mean']) <= 1e-4 * one['image_mean']
