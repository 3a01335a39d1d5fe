"""CPU tests (no GPU) of the host side of the library: the C-ABI surface, the loaders, the BVH builder
against the oracle's exhaustive restatement of the reference builder, and the multi-GPU partition logic."""
import ctypes as C
import os
import re
import struct
import zlib
import numpy as np
import pytest

import ptamd
from ptamd import scenes, lib, partition
import orc
import util
from util import bits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ C-ABI surface
def _declared_functions():
    text = open(os.path.join(ROOT, 'include', 'pt_capi.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(pt_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    names = _declared_functions()
    assert len(names) >= 45
    L = lib.lib()
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert sorted(lib.EXPORTS) == names, 'the Python binding list and the header disagree'
    # and the other way round: the library exports no pt_ function the header does not declare
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r'\b[TW] (pt_[a-z0-9_]+)$', out, flags=re.M)))
    undeclared = [n for n in exported if n not in names]
    assert not undeclared, undeclared


def test_binding_constants_equal_the_header():
    """Every PT_* integer constant the Python binding carries has the header's value."""
    text = open(os.path.join(ROOT, 'include', 'pt_capi.h')).read()
    defs = {m.group(1): int(m.group(2), 0) for m in re.finditer(r'^#define\s+(PT_[A-Z0-9_]+)\s+\(?(-?(?:0x)?[0-9a-fA-F]+)\)?\s*(?:/\*.*)?$', text, flags=re.M)}
    assert {'PT_BVH_REFERENCE', 'PT_BVH_GPU_LBVH', 'PT_BVH_GPU_LBVH_VIA_HOST', 'PT_BVH_GPU_PLOC', 'PT_NOALLOC_MAX_RAYS'} <= set(defs)
    checked = 0
    for name, value in defs.items():
        if hasattr(lib, name):
            assert getattr(lib, name) == value, name
            checked += 1
    assert checked >= 5


def test_pod_layouts_match_the_reference_structs():
    """structs.cuh:111-181 sizes as nvcc lays them out: Triangle 112, BVH_Node 32, Material 176, kernelParams 152."""
    text = open(os.path.join(ROOT, 'include', 'pt_capi.h')).read()
    for name, size in (('pt_triangle', 112), ('pt_bvh_node', 32), ('pt_material', 176), ('pt_params', 152), ('pt_ray', 32), ('pt_hit', 8)):
        assert re.search(r'PT_SIZE_CHECK\(%s,\s*%d\)' % (name, size), text), name
    assert ptamd.TRI_DTYPE.itemsize == 112 and ptamd.NODE_DTYPE.itemsize == 32 and ptamd.MAT_DTYPE.itemsize == 176
    assert C.sizeof(ptamd.Params) == 152
    assert ptamd.Params.frameIndex.offset == 72 and ptamd.Params.pixelBuffer.offset == 80 and ptamd.Params.envmap_sum.offset == 92
    assert ptamd.MAT_DTYPE.fields['diffuseTexture'][1] == 96 and ptamd.MAT_DTYPE.fields['ior'][1] == 60


def test_errors_are_reported_not_swallowed():
    L = lib.lib()
    assert L.pt_init(None, None) != 0
    assert b'context' in L.pt_last_error() or len(L.pt_last_error()) > 0
    with pytest.raises(ptamd.PtError):
        ptamd.HostScene('/nonexistent/model.obj', '/nonexistent/')
    with pytest.raises(ptamd.PtError):
        lib.load_hdr('/nonexistent/map.hdr')
    assert L.pt_render_headless(None, None, 0, 1) != 0
    assert L.pt_trace_closest(None, None, 0, None) != 0


def test_no_gpu_means_an_error_not_a_fallback():
    """Without a GPU the device entry points fail with a message; nothing falls back to the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    with pytest.raises(ptamd.PtError):
        ptamd.Renderer(0)


def test_headless_front_end_reports_errors(tmp_path):
    """ptamd_render is written against the C-ABI only; it must stop with the library's message, not fall back."""
    import subprocess
    import torch
    exe = os.path.join(ROOT, 'cuda-pathtracer_amd', 'ptamd_render')
    assert os.path.exists(exe), 'build it with make -C cuda-pathtracer_amd'
    r = subprocess.run([exe, '--obj', '/nonexistent.obj', '--dir', '/tmp/', '--hdr', 'x.hdr'], capture_output=True, text=True)
    assert r.returncode == 1 and 'cannot open model' in r.stderr
    assert subprocess.run([exe], capture_output=True).returncode == 2
    if not torch.cuda.is_available():
        util.SCRATCH = str(tmp_path)
        try:
            obj = scenes.cornell(os.path.join(str(tmp_path), 'cornell'))
        finally:
            util.SCRATCH = os.path.join(ROOT, 'tests', '_scratch')
        d = os.path.dirname(obj)
        r = subprocess.run([exe, '--obj', obj, '--dir', d + '/', '--hdr', os.path.join(d, 'black_4x2.hdr'), '--size', '16x16', '--spp', '1',
                            '--out', os.path.join(str(tmp_path), 'o.png')], capture_output=True, text=True)
        assert r.returncode == 1 and 'opening the GPU' in r.stderr and not os.path.exists(os.path.join(str(tmp_path), 'o.png'))


# ------------------------------------------------------------------ OBJ / MTL
def _write(path, text):
    with open(path, 'w') as f:
        f.write(text)


def test_obj_mtl_loader_semantics(tmp_path):
    """file_reader.h:48-425: positions scaled by 100, normals normalised, texture coordinates kept, usemtl by name,
    d -> specTrans = 1 - d, Tf -> sigmaA = 1 - Tf, map_* request textures (4 channels for colour and normal maps,
    1 for roughness / metallic / emissive), map_Bump skips its two option tokens."""
    d = str(tmp_path)
    _write(os.path.join(d, 'm.mtl'),
           'newmtl first\nKd 0.1 0.2 0.3\nPr 0.25\nPm 0.5\nKe 1 2 3\nNi 1.33 \nPs 0.125\naniso 0.75\n'
           'newmtl glass\nKd 1 1 1\nd 0.25\nTf 0.9 0.8 0.7\nTs 0.1 0.2 0.3\nPg 0.5 \n'
           'newmtl textured\nmap_Kd tex/c.png\nmap_Pr tex/r.png\nmap_Pm tex/m.png\nmap_Bump -bm 1.0 tex/n.png\nmap_Ke tex/e.png\n')
    _write(os.path.join(d, 'm.obj'),
           'mtllib m.mtl\n'
           'v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\n'
           'vt 0 0\nvt 1 0\nvt 0.25 0.75\n'
           'vn 0 0 2\nvn 0 3 0\n'
           'usemtl glass\nf 1/1/1 2/2/1 3/3/1\n'
           'usemtl textured\nf 1/1/2 3/3/2 4/2/2\n'
           'usemtl nosuchname\nf 2/1/1 3/2/1 4/3/2\n')
    hs = ptamd.HostScene(os.path.join(d, 'm.obj'), d + '/')
    assert hs.mat_names == ['first', 'glass', 'textured']
    t = hs.tris
    assert len(t) == 3 and list(t['matIndex']) == [1, 2, 0]          # unknown names fall back to material 0
    assert np.array_equal(t['a'][0], [0, 0, 0]) and np.array_equal(t['b'][0], [100, 0, 0]) and np.array_equal(t['c'][0], [0, 100, 0])
    assert np.array_equal(t['c'][1], [0, 0, 100])
    assert np.array_equal(t['nA'][0], [0, 0, 1]) and np.array_equal(t['nA'][1], [0, 1, 0]) and np.array_equal(t['nC'][2], [0, 1, 0])
    assert np.array_equal(t['tC'][0], np.array([0.25, 0.75], np.float32)) and np.array_equal(t['tB'][1], np.array([0.25, 0.75], np.float32))
    assert list(hs.tri_idx) == [0, 1, 2]
    m = hs.mats
    assert np.allclose(m['baseColor'][0], [0.1, 0.2, 0.3]) and m['roughness'][0] == 0.25 and m['metallic'][0] == 0.5
    assert np.allclose(m['emissive'][0], [1, 2, 3]) and abs(m['ior'][0] - 1.33) < 1e-6 and m['sheen'][0] == 0.125 and m['anisotropic'][0] == 0.75
    assert m['specTrans'][1] == 0.75 and np.allclose(m['sigmaA'][1], [0.1, 0.2, 0.3], atol=1e-6) and np.allclose(m['sigmaS'][1], [0.1, 0.2, 0.3])
    assert m['G'][1] == 0.5
    # defaults (structs.cuh:75-94)
    assert m['roughness'][1] == np.float32(0.01) and m['ior'][1] == 1.5 and m['alpha'][1] == 1.0
    reqs = hs.texture_requests
    assert [os.path.relpath(p, d) for p, _ in reqs] == ['tex/c.png', 'tex/r.png', 'tex/m.png', 'tex/n.png', 'tex/e.png']
    assert [c for _, c in reqs] == [4, 1, 1, 4, 1]
    mt = m[2]
    assert (mt['useTexture'], mt['use_mapPr'], mt['use_mapPm'], mt['use_mapNor'], mt['use_mapKe']) == (1, 1, 1, 1, 1)
    assert (mt['diffuseTexture'], mt['roughnessTexture'], mt['metallicTexture'], mt['normalTexture'], mt['emissiveTexture']) == (1, 2, 3, 4, 5)


def test_obj_loader_index_quirk_and_plain_faces(tmp_path):
    """file_reader.h:386-397: a position index above the triangle count is reset to vertex 0 (a reference bug that
    any mesh with more vertices than faces would hit); faces without slashes are read when no vn/vt exist."""
    d = str(tmp_path)
    _write(os.path.join(d, 'p.mtl'), 'newmtl only\nKd 0.5 0.5 0.5\n')
    _write(os.path.join(d, 'p.obj'), 'mtllib p.mtl\nv 1 1 1\nv 2 0 0\nv 0 2 0\nv 0 0 2\nv 5 5 5\nusemtl only\nf 2 3 4\nf 2 3 5\n')
    hs = ptamd.HostScene(os.path.join(d, 'p.obj'), d + '/')
    t = hs.tris
    assert len(t) == 2
    assert np.array_equal(t['a'][0], [200, 0, 0]) and np.array_equal(t['b'][0], [0, 200, 0])
    # zero-based indices 3 (vertex 4) and 4 (vertex 5) exceed nbTri = 2 -> both become vertex 0
    assert np.array_equal(t['c'][0], [100, 100, 100]) and np.array_equal(t['c'][1], [100, 100, 100])
    assert np.all(t['nA'] == 0) and np.all(t['tA'] == 0)


def test_general_obj_reader(tmp_path):
    """pt_read_obj_general: quads and pentagons fan out, corner syntaxes mix, negative indices count from the end,
    vertices may outnumber faces, missing normals become the face normal; on a file both readers accept the
    records are identical."""
    d = str(tmp_path)
    _write(os.path.join(d, 'g.mtl'), 'newmtl a\nKd 1 0 0\nnewmtl b with space\nKd 0 1 0\n')
    _write(os.path.join(d, 'g.obj'),
           '# comment\nmtllib g.mtl\n'
           'v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 1.5 0\nv 9 9 9\nv 8 8 8\nv 7 7 7\n'
           'vt 0 0\nvt 1 0 0\nvt 1 1\n'
           'vn 0 0 5\n'
           'usemtl a\nf 1 2 3 4\n'                      # quad, no normals
           'usemtl b with space\n\tf  1/1/1   2/2/1\t3/3/1 4//1 5\n'     # pentagon with mixed corners
           'f -3 -2 -1\n')                             # relative indices: vertices 6, 7, 8
    hs = ptamd.HostScene(os.path.join(d, 'g.obj'), d + '/', general=True, scale=10.0)
    t = hs.tris
    assert len(t) == 2 + 3 + 1 and list(t['matIndex']) == [0, 0, 1, 1, 1, 1]
    assert np.array_equal(t['a'][0], [0, 0, 0]) and np.array_equal(t['b'][0], [10, 0, 0]) and np.array_equal(t['c'][0], [10, 10, 0])
    assert np.array_equal(t['b'][1], [10, 10, 0]) and np.array_equal(t['c'][1], [0, 10, 0])
    assert np.all(t['nA'][:2] == [0, 0, 1]) and np.all(t['nC'][:2] == [0, 0, 1])          # face normal of a counter-clockwise quad
    assert np.array_equal(t['c'][4], [5, 15, 0]) and np.array_equal(t['tB'][2], [1, 0]) and np.array_equal(t['tC'][2], [1, 1])
    assert np.array_equal(t['nA'][2], [0, 0, 1]) and np.array_equal(t['tC'][4], [0, 0])      # corner 5 has neither uv nor normal
    assert np.array_equal(t['a'][5], [90, 90, 90]) and np.array_equal(t['c'][5], [70, 70, 70])
    assert np.all(np.isfinite(t['nA'])) and list(hs.tri_idx) == list(range(6))
    with pytest.raises(ptamd.PtError, match='out of range'):
        _write(os.path.join(d, 'bad.obj'), 'v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 4\n')
        ptamd.HostScene(os.path.join(d, 'bad.obj'), d + '/', general=True)
    # a triangulated v/vt/vn file: both readers give the same records
    obj = scenes.cornell(os.path.join(d, 'cornell'))
    a = ptamd.HostScene(obj, os.path.dirname(obj) + '/')
    b = ptamd.HostScene(obj, os.path.dirname(obj) + '/', general=True)
    assert np.array_equal(a.tris.view(np.uint8), b.tris.view(np.uint8))
    assert all(np.array_equal(a.mats[k], b.mats[k]) for k in a.mats.dtype.names) and a.mat_names == b.mat_names
    # the shipped bunny has more vertices per face than the reference reader tolerates only by luck; here it just loads
    import shutil
    shutil.copy(os.path.join(ROOT, 'tests', 'golden', 'bunny_triangle_2.obj.txt'), os.path.join(d, 'bunny.obj'))
    _write(os.path.join(d, 'Bunny_triangle_2.mtl'), 'newmtl none\n')
    bn = ptamd.HostScene(os.path.join(d, 'bunny.obj'), d + '/', general=True)
    assert len(bn.tris) == 292 and np.all(np.isfinite(bn.tris['nA'])) and np.allclose(np.linalg.norm(bn.tris['nA'], axis=1), 1, atol=1e-5)


def test_loader_rejects_broken_input(tmp_path):
    d = str(tmp_path)
    _write(os.path.join(d, 'e.mtl'), 'newmtl a\nKd 1 x 1\n')
    _write(os.path.join(d, 'e.obj'), 'mtllib e.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n')
    with pytest.raises(ptamd.PtError, match='malformed'):
        ptamd.HostScene(os.path.join(d, 'e.obj'), d + '/')
    _write(os.path.join(d, 'n.obj'), 'v 0 0 0\nv 1 0 0\n')
    with pytest.raises(ptamd.PtError, match='no faces'):
        ptamd.HostScene(os.path.join(d, 'n.obj'), d + '/')
    _write(os.path.join(d, 'm.obj'), 'mtllib missing.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n')
    with pytest.raises(ptamd.PtError, match='material library'):
        ptamd.HostScene(os.path.join(d, 'm.obj'), d + '/')


@pytest.mark.skipif(not os.path.isdir('/root/reference/Cuda Pathtracer/assets/models'), reason='reference assets not present')
def test_every_material_library_the_reference_ships_parses(tmp_path):
    import glob
    mtls = sorted(glob.glob('/root/reference/Cuda Pathtracer/assets/models/**/*.mtl', recursive=True))
    assert len(mtls) >= 20
    d = str(tmp_path)
    total = 0
    for k, mtl in enumerate(mtls):
        sub = os.path.dirname(mtl) + '/'
        obj = os.path.join(d, 'probe%d.obj' % k)
        _write(obj, 'mtllib %s\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n' % os.path.basename(mtl))
        hs = ptamd.HostScene(obj, sub)
        assert len(hs.mats) >= 1 and len(hs.mats) == len(hs.mat_names)
        assert np.all(np.isfinite(hs.mats['baseColor'])) and np.all(hs.mats['roughness'] >= 0)
        total += len(hs.mats)
    assert total > 50


# ------------------------------------------------------------------ HDR / CDF / PNG
def test_hdr_loader_decodes_rgbe(tmp_path):
    """hdr_loader.h:196-214: (mantissa + 0.5) * 2^(e - 136), zero exponent = black; run-length rows for widths 8..32767."""
    rng = np.random.default_rng(3)
    for w, h in ((4, 2), (16, 3), (64, 8)):           # the first is stored flat, the others run-length encoded
        img = (rng.uniform(0, 1, (h, w, 3)) ** 4 * 50).astype(np.float32)
        img[0, 0] = 0
        p = os.path.join(str(tmp_path), 'x%d.hdr' % w)
        scenes.write_hdr(p, img)
        got = lib.load_hdr(p)
        assert got.shape == (h, w, 4) and np.all(got[..., 3] == 0)
        assert np.all(got[0, 0] == 0)
        scale = img.max(-1, keepdims=True)          # one exponent per pixel: the error is relative to the largest channel
        assert np.all(np.abs(got[..., :3] - img) <= scale / 100 + 1e-6)
    # exact decode of a hand-made flat file: mantissas (128, 64, 0), exponent 129 -> (128.5, 64.5, 0.5) * 2^-7
    p = os.path.join(str(tmp_path), 'hand.hdr')
    with open(p, 'wb') as f:
        f.write(b'#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 1 +X 2\n' + bytes([128, 64, 0, 129, 1, 2, 3, 0]))
    got = lib.load_hdr(p)
    assert np.array_equal(got[0, 0, :3], np.array([128.5, 64.5, 0.5], np.float32) * np.float32(2.0 ** -7))
    assert np.all(got[0, 1] == 0)


@pytest.mark.skipif(not os.path.exists('/root/reference/Cuda Pathtracer/assets/cubemaps/metro_noord_1k.hdr'), reason='reference assets not present')
def test_shipped_environment_map_loads():
    env = lib.load_hdr('/root/reference/Cuda Pathtracer/assets/cubemaps/metro_noord_1k.hdr')
    assert env.shape == (512, 1024, 4) and np.all(np.isfinite(env)) and env.min() >= 0 and env.max() > 10
    cdf, total = lib.build_env_cdf(env)
    assert np.all(np.diff(cdf.ravel()) >= 0) and total > 0


def test_env_cdf_is_a_running_fp32_sum():
    """envmap.cpp:11-28: luminance weights 0.212671 / 0.715160 / 0.072169 in double, summed in float, row-major."""
    rng = np.random.default_rng(9)
    env = rng.uniform(0, 4, (8, 16, 4)).astype(np.float32)
    cdf, total = lib.build_env_cdf(env)
    run = np.float32(0)
    want = np.zeros(8 * 16, np.float32)
    flat = env.reshape(-1, 4).astype(np.float64)
    for i in range(len(want)):
        wgt = np.float32(0.212671 * flat[i, 0] + 0.715160 * flat[i, 1] + 0.072169 * flat[i, 2])
        run = wgt if i == 0 else np.float32(run + wgt)
        want[i] = run
    assert np.array_equal(bits(cdf.ravel()), bits(want)) and np.float32(total) == want[-1]


def _decode_png(path):
    data = open(path, 'rb').read()
    assert data[:8] == b'\x89PNG\r\n\x1a\n'
    pos, idat, w, h = 8, b'', 0, 0
    while pos < len(data):
        n, typ = struct.unpack('>I4s', data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        crc, = struct.unpack('>I', data[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(typ + body) & 0xffffffff
        if typ == b'IHDR':
            w, h, depth, ctype = struct.unpack('>IIBB', body[:10])
            assert depth == 8 and ctype == 6
        elif typ == b'IDAT':
            idat += body
        pos += 12 + n
    raw = zlib.decompress(idat)
    rows = np.frombuffer(raw, np.uint8).reshape(h, 1 + 4 * w)
    assert np.all(rows[:, 0] == 0)          # filter type 0 on every row
    return rows[:, 1:].reshape(h, w, 4)


def test_png_writer_round_trip(tmp_path):
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (13, 17, 4), dtype=np.uint8)
    p = os.path.join(str(tmp_path), 'shot.png')
    lib.save_png(p, img)
    assert np.array_equal(_decode_png(p), img)
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(p)), img)


def test_png_texture_decoder_matches_an_independent_decoder(tmp_path):
    """genTexture / genTexture_float (image_loader.h:5-89): RGBA8 view of the file, rows flipped, / 255."""
    from PIL import Image
    rng = np.random.default_rng(1)
    d = str(tmp_path)
    for mode, (w, h) in (('L', (13, 7)), ('RGB', (16, 9)), ('RGBA', (5, 5)), ('LA', (8, 3)), ('P', (12, 12)), ('1', (17, 5))):
        if mode == '1':
            img = Image.fromarray((rng.integers(0, 2, (h, w)) * 255).astype(np.uint8)).convert('1')
        elif mode == 'P':
            img = Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).convert('P')
        else:
            ch = {'L': 1, 'RGB': 3, 'RGBA': 4, 'LA': 2}[mode]
            a = rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
            img = Image.fromarray(a[:, :, 0] if ch == 1 else a, mode)
        p = os.path.join(d, mode + '.png')
        img.save(p)
        for comps in (4, 1):
            got = lib.load_texture_png(p, comps)
            assert got.shape == (h, w, comps) and np.array_equal(got, scenes.decode_texture(p, comps)), (mode, comps)
    bad = os.path.join(d, 'bad.png')
    raw = bytearray(open(os.path.join(d, 'RGB.png'), 'rb').read())
    raw[40] ^= 0xff
    open(bad, 'wb').write(bytes(raw))
    with pytest.raises(ptamd.PtError):
        lib.load_texture_png(bad, 4)
    with pytest.raises(ptamd.PtError):
        lib.load_texture_png(os.path.join(d, 'missing.png'), 4)
    import glob
    for p in sorted(glob.glob('/root/reference/Cuda Pathtracer/assets/models/**/*.png', recursive=True))[:6]:      # the textures the reference ships
        assert np.array_equal(lib.load_texture_png(p, 4), scenes.decode_texture(p, 4)), p


def test_jpeg_texture_decoder_agrees_with_libjpeg(tmp_path):
    """pt_load_texture_image on JPEG files (the reference reads them through sf::Image, image_loader.h:8,52):
    baseline and progressive, 4:4:4 / 4:2:2 / 4:2:0, grey, restart intervals, odd sizes. Decoders differ in the last
    bits of the inverse transform and the chroma filter, so the bar is libjpeg's pixels within 3 levels, 0.1 on average;
    the texture array is that image with rows flipped, / 255, alpha 1. Parity with the reference's own decoder is unpinned."""
    from PIL import Image
    rng = np.random.default_rng(3)
    y, x = np.mgrid[0:203, 0:317]
    a = np.stack([127 + 100 * np.sin(x / 17.0 + y / 29.0), 127 + 100 * np.cos(x / 11.0 - y / 7.0), 127 + 90 * np.sin((x + y) / 23.0)], -1)
    a = np.clip(a + rng.normal(0, 12, a.shape), 0, 255).astype(np.uint8)
    d = str(tmp_path)
    cases = [('444', a, dict(quality=90, subsampling=0)), ('422', a, dict(quality=90, subsampling=1)), ('420', a, dict(quality=50, subsampling=2)),
             ('420p', a, dict(quality=90, subsampling=2, progressive=True)), ('444p', a, dict(quality=75, subsampling=0, progressive=True)),
             ('grey', a[:, :, 0], dict(quality=85)), ('greyp', a[:, :, 0], dict(quality=85, progressive=True)),
             ('1x1', a[:1, :1], dict(quality=90)), ('7x5', a[:5, :7], dict(quality=90, subsampling=2)), ('opt', a, dict(quality=90, subsampling=2, optimize=True)),
             ('rst', a, dict(quality=90, subsampling=2, restart_marker_blocks=3))]
    for name, img, kw in cases:
        p = os.path.join(d, name + '.jpg')
        Image.fromarray(img).save(p, 'JPEG', **kw)
        ref = np.asarray(Image.open(p).convert('RGB')).astype(np.int32)
        got = lib.load_texture_image(p, 4)
        assert got.shape == ref.shape[:2] + (4,) and np.all(got[:, :, 3] == 1.0), name
        px = got[::-1, :, :3] * np.float32(255)
        assert np.array_equal(px, np.rint(px)), name                       # exact k / 255 values
        diff = np.abs(px.astype(np.int32) - ref)
        assert diff.max() <= 3 and diff.mean() < 0.1, (name, diff.max(), diff.mean())
        assert np.array_equal(lib.load_texture_image(p, 1)[:, :, 0], got[:, :, 0]), name
    # a PNG goes through the same entry point
    Image.fromarray(a).save(os.path.join(d, 'a.png'))
    assert np.array_equal(lib.load_texture_image(os.path.join(d, 'a.png'), 4), lib.load_texture_png(os.path.join(d, 'a.png'), 4))
    # damaged and unsupported files are errors or images, never crashes
    raw = open(os.path.join(d, '420p.jpg'), 'rb').read()
    for cut in (2, 20, 200, len(raw) // 2, len(raw) - 2):
        open(os.path.join(d, 'cut.jpg'), 'wb').write(raw[:cut])
        try:
            t = lib.load_texture_image(os.path.join(d, 'cut.jpg'), 4)
            assert t.shape == (203, 317, 4)
        except ptamd.PtError:
            pass
    noisy = bytearray(raw)
    for k in rng.integers(600, len(raw) - 2, 40):
        noisy[k] = int(rng.integers(0, 255))
    open(os.path.join(d, 'noisy.jpg'), 'wb').write(bytes(noisy))
    try:
        lib.load_texture_image(os.path.join(d, 'noisy.jpg'), 4)
    except ptamd.PtError:
        pass
    # a Huffman table whose code lengths over-subscribe the code space (255 codes of one bit) is refused
    base = open(os.path.join(d, '444.jpg'), 'rb').read()
    at = base.index(b'\xff\xc4')
    seg_len = int.from_bytes(base[at + 2:at + 4], 'big')
    bad_counts = bytes([255, 1] + [0] * 14)
    bad = base[:at + 5] + bad_counts + base[at + 21:]
    bad = bad[:at + 2] + (2 + 1 + 16 + 256).to_bytes(2, 'big') + bad[at + 4:at + 21] + bytes(256) + bad[at + 2 + seg_len:]
    open(os.path.join(d, 'dht.jpg'), 'wb').write(bad)
    with pytest.raises(ptamd.PtError, match='Huffman'):
        lib.load_texture_image(os.path.join(d, 'dht.jpg'), 4)
    Image.fromarray(a).convert('CMYK').save(os.path.join(d, 'cmyk.jpg'), 'JPEG')
    with pytest.raises(ptamd.PtError):
        lib.load_texture_image(os.path.join(d, 'cmyk.jpg'), 4)
    import glob
    for p in sorted(glob.glob('/root/reference/Cuda Pathtracer/assets/models/**/*.jpg', recursive=True)):      # the JPEG textures the reference ships
        ref = np.asarray(Image.open(p).convert('RGB')).astype(np.int32)
        px = np.rint(lib.load_texture_image(p, 4)[::-1, :, :3] * 255).astype(np.int32)
        diff = np.abs(px - ref)
        assert diff.max() <= 3 and diff.mean() < 0.1, (p, diff.max(), diff.mean())


def test_pfm_round_trip_is_lossless(tmp_path):
    rng = np.random.default_rng(6)
    img = rng.normal(0, 100, (7, 11, 3)).astype(np.float32)
    img[0, 0] = [np.inf, -0.0, 1e-42]
    p = os.path.join(str(tmp_path), 'accum.pfm')
    lib.save_pfm(p, img)
    raw = open(p, 'rb').read()
    assert raw.startswith(b'PF\n11 7\n-1.0\n') and len(raw) == len(b'PF\n11 7\n-1.0\n') + 7 * 11 * 12
    # rows are stored bottom-up
    assert np.array_equal(np.frombuffer(raw[-11 * 12:], '<f4').view(np.uint32), bits(img[0]).ravel())
    back = lib.load_pfm(p)
    assert np.array_equal(bits(back), bits(img))
    with open(p, 'wb') as f:
        f.write(b'PF\n11 7\n1.0\n' + b'\0' * 10)
    with pytest.raises(ptamd.PtError):
        lib.load_pfm(p)


# ------------------------------------------------------------------ BVH builder
def test_exr_writer_is_readable_by_the_format_rules(tmp_path):
    """pt_save_exr: single-part scan-line OpenEXR, no compression, float B/G/R channels; parsed back here by the
    published layout (magic, version, attribute list, line offset table, one block per scan line)."""
    rng = np.random.default_rng(9)
    img = rng.normal(0, 50, (5, 7, 3)).astype(np.float32)
    p = str(tmp_path / 'a.exr')
    lib.save_exr(p, img, 0.25)
    raw = open(p, 'rb').read()
    assert struct.unpack_from('<ii', raw, 0) == (20000630, 2)
    pos, attrs = 8, {}
    while raw[pos] != 0:
        e = raw.index(b'\0', pos); name = raw[pos:e].decode(); pos = e + 1
        e = raw.index(b'\0', pos); typ = raw[pos:e].decode(); pos = e + 1
        size, = struct.unpack_from('<i', raw, pos); pos += 4
        attrs[name] = (typ, raw[pos:pos + size]); pos += size
    pos += 1
    assert attrs['compression'] == ('compression', b'\0') and attrs['lineOrder'] == ('lineOrder', b'\0')
    assert struct.unpack('<4i', attrs['dataWindow'][1]) == (0, 0, 6, 4) == struct.unpack('<4i', attrs['displayWindow'][1])
    ch, names, q = attrs['channels'][1], [], 0
    while ch[q] != 0:
        e = ch.index(b'\0', q); names.append(ch[q:e].decode()); q = e + 1
        assert struct.unpack_from('<i4xii', ch, q) == (2, 1, 1); q += 16
    assert names == ['B', 'G', 'R']
    offsets = struct.unpack_from('<5Q', raw, pos)
    out = np.zeros_like(img)
    for y, off in enumerate(offsets):
        yy, nbytes = struct.unpack_from('<ii', raw, off)
        assert yy == y and nbytes == 7 * 3 * 4
        row = np.frombuffer(raw, '<f4', 21, off + 8).reshape(3, 7)
        out[y, :, 2], out[y, :, 1], out[y, :, 0] = row[0], row[1], row[2]
    assert offsets[-1] + 8 + 84 == len(raw)
    assert np.array_equal(out, img * np.float32(0.25))
    with pytest.raises(ptamd.PtError):
        lib.save_exr(str(tmp_path / 'no_such_dir' / 'a.exr'), img)


@pytest.mark.parametrize('n,seed', [(1, 1), (2, 2), (3, 3), (37, 4), (500, 5), (3000, 6)])
def test_host_builder_reproduces_the_reference_builder(n, seed):
    """The product's builder (binned sweeps, parallel subtrees) must emit the tree BVH_builder.cpp:12-128 builds
    (exhaustive 99-plane sweeps, restated in the oracle): same nodes, same order, same index permutation."""
    tris, mats = scenes.soup(n, seed=seed, extent=200.0, edge=15.0)
    if n >= 37:          # duplicates and coincident centroids
        tris[5] = tris[4]
        tris['a'][7] = tris['a'][6]; tris['b'][7] = tris['c'][6]; tris['c'][7] = tris['b'][6]
    mine = tris.copy()
    nodes, idx = lib.host_build_bvh(mine)
    S = orc.Scene(tris.copy(), mats)
    used = S.build_bvh_reference()
    assert used == len(nodes)
    assert np.array_equal(S.tri_idx, idx)
    assert np.array_equal(S.nodes[:used].view(np.uint8), nodes.view(np.uint8))
    assert np.array_equal(bits(S.tris['origin']), bits(mine['origin']))


def test_host_builder_on_structured_meshes():
    b = util.cornell('black')
    S = orc.Scene(b.tris.copy(), b.mats)
    used = S.build_bvh_reference()
    assert used == len(b.nodes) and np.array_equal(S.nodes[:used].view(np.uint8), b.nodes.view(np.uint8)) and np.array_equal(S.tri_idx, b.idx)
    a = util.atrium(detail=0.12, env_size=(32, 16))
    assert len(a.tris) > 2000
    S = orc.Scene(a.tris.copy(), a.mats)
    used = S.build_bvh_reference()
    assert used == len(a.nodes) and np.array_equal(S.nodes[:used].view(np.uint8), a.nodes.view(np.uint8)) and np.array_equal(S.tri_idx, a.idx)


def test_host_builder_rejects_bad_arguments():
    L = lib.lib()
    used = C.c_int()
    assert L.pt_host_build_bvh(None, None, C.c_int(0), C.c_int(0), None, C.byref(used)) != 0
    tris, _ = scenes.soup(4, seed=1, extent=10.0, edge=1.0)
    nodes = np.zeros(8, ptamd.NODE_DTYPE); idx = np.arange(4, dtype=np.int32)
    assert L.pt_host_build_bvh(tris.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p), C.c_int(4), C.c_int(99),
                               nodes.ctypes.data_as(C.c_void_p), C.byref(used)) != 0


# ------------------------------------------------------------------ scene generators
def test_scene_generators_are_deterministic(tmp_path):
    a = scenes.atrium(os.path.join(str(tmp_path), 'a'), detail=0.1)
    b = scenes.atrium(os.path.join(str(tmp_path), 'b'), detail=0.1)
    assert open(a).read() == open(b).read()
    hs = ptamd.HostScene(a, os.path.dirname(a) + '/')
    assert len(hs.mats) == 25 and len(hs.tris) > 1000
    assert len(hs.texture_requests) > 10 and all(os.path.exists(p) for p, _ in hs.texture_requests)
    tex = scenes.decode_texture(*hs.texture_requests[0])
    assert tex.dtype == np.float32 and tex.ndim == 3 and 0 <= tex.min() and tex.max() <= 1


# ------------------------------------------------------------------ multi-GPU partition
def test_frame_partition_covers_every_frame_once():
    for total in (0, 1, 7, 64, 4096):
        for world in (1, 2, 3, 8):
            ranges = [partition.split_frames(total, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1
    seen = set()
    for step in range(3):
        for r in range(4):
            a, b = partition.step_frames(step, 4, r, 16)
            assert b - a == 16
            seen.update(range(a, b))
    assert seen == set(range(3 * 4 * 16))
    with pytest.raises(ValueError):
        partition.split_frames(8, 2, 2)


def _gloo_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        util.SCRATCH = os.path.join(out_dir, 'rank%d' % rank)       # each rank writes its own copy of the scene files
        b = util.cornell('sky')
        P = scenes.cornell_params(ptamd.default_params(20, 20))
        S = b.oracle(P)
        total_frames = 6
        f0, f1 = partition.split_frames(total_frames, world, rank)
        acc, _ = orc.render(S, P, f0, f1, 5, nthreads=2)          # the CPU oracle stands in for the device renderer
        t = torch.from_numpy(acc)
        partition.reduce_accum(t, dst=0)
        if rank == 0:
            whole, _ = orc.render(S, P, 0, total_frames, 5, nthreads=2)
            got = t.numpy()
            err = np.max(np.abs(got - whole) / np.maximum(np.abs(whole), 1e-3))
            img_a = orc.resolve_display(got, total_frames).astype(np.int32)
            img_b = orc.resolve_display(whole, total_frames).astype(np.int32)
            np.save(os.path.join(out_dir, 'result.npy'), np.array([err, np.abs(img_a - img_b).max(), whole.sum()]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_sample_partition_with_gloo_reduce(tmp_path):
    """World size 2 on CPU: each rank renders its frame range, the accumulators are summed on rank 0 and
    agree with the one-process render to fp32 rounding (the only difference is the order of the additions)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_gloo_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    err, byte_diff, total = np.load(os.path.join(str(tmp_path), 'result.npy'))
    assert total > 0 and err <= 1e-4 and byte_diff <= 1


def test_group_frame_ranges_partition_a_render():
    """pt_group_frames (the C++ multi-GPU path) deals frames as partition.split_frames does: disjoint, in order, complete."""
    for begin, end, n in ((0, 4096, 8), (5, 1029, 8), (0, 7, 4), (3, 3, 2), (0, 1, 3), (100, 164, 1)):
        edges = [lib.group_frames(begin, end, n, r) for r in range(n)]
        assert edges[0][0] == begin and edges[-1][1] == end
        for r in range(n):
            a, b = partition.split_frames(end - begin, n, r)
            assert edges[r] == (begin + a, begin + b)
            if r:
                assert edges[r][0] == edges[r - 1][1]
    with pytest.raises(ptamd.PtError):
        lib.group_frames(0, 4, 2, 2)
