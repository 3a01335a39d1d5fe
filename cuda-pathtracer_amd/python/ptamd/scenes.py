"""Seeded synthetic scenes, written as OBJ/MTL (+ PNG textures, + Radiance HDR) so that they enter the
renderer through the same loaders a user scene would (pt_read_obj, pt_load_hdr).

The reference repository ships no meshes, so every benchmark configuration is generated here:
  cornell()  -- 36-triangle box scene lit by an emissive quad
  atrium()   -- a Sponza-class colonnaded hall, about 262k triangles, 25 materials with texture maps
  soup()     -- N random small triangles (arrays only; the OBJ path would mangle it, see below)

Loader constraints honoured by the writers (file_reader.h:245-404): tokens separated by single spaces,
faces are triangles in v/vt/vn form, and no vertex index exceeds the triangle count (larger indices
are reset to vertex 0 by the reference loader).
"""
import os
import struct
import numpy as np

from .types import TRI_DTYPE, MAT_DTYPE, default_materials


# ---------------------------------------------------------------------------------------------- mesh helpers
class Mesh:
    def __init__(self):
        self.v, self.vt, self.vn = [], [], []      # arrays (n,3), (n,2), (n,3)
        self.groups = []                           # (material name, faces (m,3) global 0-based indices)
        self.nv = 0

    def add(self, mat, v, vt, vn, faces):
        v = np.asarray(v, np.float64).reshape(-1, 3)
        vt = np.asarray(vt, np.float64).reshape(-1, 2)
        vn = np.asarray(vn, np.float64).reshape(-1, 3)
        assert len(v) == len(vt) == len(vn)
        self.v.append(v); self.vt.append(vt); self.vn.append(vn)
        self.groups.append((mat, np.asarray(faces, np.int64).reshape(-1, 3) + self.nv))
        self.nv += len(v)

    def n_tris(self):
        return sum(len(f) for _, f in self.groups)

    def write(self, obj_path, mtl_name):
        """v, vt and vn are indexed independently in OBJ: each list is de-duplicated on its own."""
        def uniq(arr):
            a32 = np.ascontiguousarray(np.concatenate(arr).astype(np.float32))
            u, inv = np.unique(a32, axis=0, return_inverse=True)
            return u, inv.reshape(-1)
        v, iv = uniq(self.v); vt, it = uniq(self.vt); vn, inn = uniq(self.vn)
        assert len(v) <= self.n_tris() + 1, 'the reference loader resets vertex indices above the triangle count'
        with open(obj_path, 'w') as f:
            f.write('mtllib %s\n' % mtl_name)
            f.write(''.join('v %.9g %.9g %.9g\n' % tuple(p) for p in v))
            f.write(''.join('vt %.9g %.9g\n' % tuple(p) for p in vt))
            f.write(''.join('vn %.9g %.9g %.9g\n' % tuple(p) for p in vn))
            for mat, faces in self.groups:
                f.write('usemtl %s\n' % mat)
                a, b, c = iv[faces] + 1, it[faces] + 1, inn[faces] + 1
                f.write(''.join('f %d/%d/%d %d/%d/%d %d/%d/%d\n' % (a[k, 0], b[k, 0], c[k, 0], a[k, 1], b[k, 1], c[k, 1], a[k, 2], b[k, 2], c[k, 2])
                                for k in range(len(faces))))


def grid(origin, du, dv, nu, nv, uv_scale=(1.0, 1.0), height=None):
    """(nu x nv) quads spanned by du, dv from origin; optional height(u,v)->offset along the normal."""
    origin, du, dv = (np.asarray(a, np.float64) for a in (origin, du, dv))
    n = np.cross(du, dv); n /= np.linalg.norm(n)
    u = np.linspace(0, 1, nu + 1); w = np.linspace(0, 1, nv + 1)
    U, W = np.meshgrid(u, w, indexing='xy')
    P = origin + U[..., None] * du + W[..., None] * dv
    N = np.broadcast_to(n, P.shape).copy()
    if height is not None:
        Hh = height(U, W)
        P = P + Hh[..., None] * n
        # finite-difference normals
        e = 1e-3
        Hu = (height(U + e, W) - Hh) / e
        Hw = (height(U, W + e) - Hh) / e
        tu = du + Hu[..., None] * n
        tw = dv + Hw[..., None] * n
        N = np.cross(tu, tw); N /= np.linalg.norm(N, axis=-1, keepdims=True)
    T = np.stack([U * uv_scale[0], W * uv_scale[1]], -1)
    idx = np.arange((nu + 1) * (nv + 1)).reshape(nv + 1, nu + 1)
    a, b, c, d = idx[:-1, :-1], idx[:-1, 1:], idx[1:, 1:], idx[1:, :-1]
    faces = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)])
    return P.reshape(-1, 3), T.reshape(-1, 2), N.reshape(-1, 3), faces


def revolve(center, profile_r, profile_y, nseg, uv_scale=(1.0, 1.0)):
    """Surface of revolution about the y axis through center; profile arrays give radius and height."""
    center = np.asarray(center, np.float64)
    r = np.asarray(profile_r, np.float64); y = np.asarray(profile_y, np.float64)
    th = np.linspace(0, 2 * np.pi, nseg + 1)
    R, TH = np.meshgrid(r, th, indexing='ij')
    Y = np.broadcast_to(y[:, None], R.shape)
    P = np.stack([center[0] + R * np.cos(TH), center[1] + Y, center[2] + R * np.sin(TH)], -1)
    dr = np.gradient(r); dy = np.gradient(y)
    nr = dy / np.maximum(np.hypot(dr, dy), 1e-12); ny = -dr / np.maximum(np.hypot(dr, dy), 1e-12)
    N = np.stack([nr[:, None] * np.cos(TH), np.broadcast_to(ny[:, None], R.shape), nr[:, None] * np.sin(TH)], -1)
    T = np.stack([TH / (2 * np.pi) * uv_scale[0], np.broadcast_to(np.linspace(0, 1, len(r))[:, None], R.shape) * uv_scale[1]], -1)
    nr_, ns_ = len(r), nseg + 1
    idx = np.arange(nr_ * ns_).reshape(nr_, ns_)
    a, b, c, d = idx[:-1, :-1], idx[:-1, 1:], idx[1:, 1:], idx[1:, :-1]
    faces = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)])
    return P.reshape(-1, 3), T.reshape(-1, 2), N.reshape(-1, 3), faces


def sphere(center, radius, nseg, nring, bump=None):
    ph = np.linspace(0, np.pi, nring + 1)
    r = radius * np.sin(ph); y = -radius * np.cos(ph)
    P, T, N, F = revolve(center, r, y, nseg)
    if bump is not None:
        c = np.asarray(center, np.float64)
        d = P - c
        l = np.linalg.norm(d, axis=1, keepdims=True)
        dirn = d / np.maximum(l, 1e-12)
        P = c + dirn * (l + bump(dirn))
        N = dirn
    else:
        N = (P - np.asarray(center)) / radius
    return P, T, N, F


# ---------------------------------------------------------------------------------------------- textures / envmap
def _write_png(path, arr_u8):
    from PIL import Image
    Image.fromarray(arr_u8).save(path)


def decode_texture(path, comps):
    """What genTexture / genTexture_float produce (image_loader.h:5-89): rows flipped, /255, RGBA or R."""
    from PIL import Image
    img = np.asarray(Image.open(path).convert('RGBA'), dtype=np.uint8)
    img = img[::-1]
    f = img.astype(np.float32) / np.float32(255.0)
    return np.ascontiguousarray(f if comps == 4 else f[:, :, :1])


def write_hdr(path, rgb):
    """Radiance RGBE writer: run-length encoded scanlines for widths 8..32767, flat otherwise."""
    rgb = np.asarray(rgb, np.float32)
    h, w, _ = rgb.shape
    m = rgb.max(axis=2)
    e = np.zeros((h, w), np.int32)
    nz = m > 1e-32
    mant, ex = np.frexp(m[nz])
    scale = np.zeros((h, w), np.float64)
    scale[nz] = mant * 256.0 / m[nz]
    e[nz] = ex + 128
    rgbe = np.zeros((h, w, 4), np.uint8)
    rgbe[..., :3] = np.clip(rgb * scale[..., None], 0, 255).astype(np.uint8)
    rgbe[..., 3] = np.where(nz, e, 0).astype(np.uint8)
    with open(path, 'wb') as f:
        f.write(b'#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n' % (h, w))
        if w < 8 or w > 0x7fff:
            f.write(rgbe.tobytes())
            return
        for y in range(h):
            f.write(bytes([2, 2, (w >> 8) & 0xff, w & 0xff]))
            for c in range(4):
                row = rgbe[y, :, c]
                x = 0
                out = bytearray()
                while x < w:
                    # run detection
                    run = 1
                    while x + run < w and run < 127 and row[x + run] == row[x]:
                        run += 1
                    if run >= 4:
                        out += bytes([128 + run, int(row[x])]); x += run
                    else:
                        start = x
                        while x < w and x - start < 128:
                            run = 1
                            while x + run < w and run < 4 and row[x + run] == row[x]:
                                run += 1
                            if run >= 4:
                                break
                            x += 1
                        out += bytes([x - start]) + row[start:x].tobytes()
                f.write(bytes(out))


def sky_envmap(w, h, seed=3, sun_dir=(1.0, 0.7, -1.0), sun_power=300.0):
    """Procedural lat-long sky: gradient + ground + a small bright sun + a few soft area lights."""
    rng = np.random.default_rng(seed)
    v, u = np.meshgrid((np.arange(h) + 0.5) / h, (np.arange(w) + 0.5) / w, indexing='ij')
    theta = v * np.pi
    phi = u * 2 * np.pi
    d = np.stack([-np.sin(theta) * np.cos(phi), np.cos(theta), -np.sin(theta) * np.sin(phi)], -1)
    up = np.clip(d[..., 1], 0, 1)
    sky = np.stack([0.25 + 0.35 * (1 - up), 0.45 + 0.3 * (1 - up), 0.9 - 0.1 * (1 - up)], -1) * 1.2
    ground = np.array([0.12, 0.1, 0.08])
    img = np.where(d[..., 1:2] > 0, sky, ground)
    s = np.asarray(sun_dir, np.float64); s /= np.linalg.norm(s)
    cs = (d * s).sum(-1)
    img = img + sun_power * np.exp((cs - 1) / 0.0008)[..., None] * np.array([1.0, 0.9, 0.75])
    for _ in range(4):
        c = rng.normal(size=3); c[1] = abs(c[1]) + 0.2; c /= np.linalg.norm(c)
        img = img + 3.0 * np.exp(((d * c).sum(-1) - 1) / 0.02)[..., None] * rng.uniform(0.5, 1.0, 3)
    return img.astype(np.float32)


# ---------------------------------------------------------------------------------------------- Cornell box
def cornell(out_dir):
    """5 walls + 2 boxes + ceiling emitter, 36 triangles, 4 materials; OBJ units are metres (x100 on load)."""
    os.makedirs(out_dir, exist_ok=True)
    m = Mesh()

    def quad(mat, p0, p1, p2, p3):
        P = np.array([p0, p1, p2, p3], np.float64)
        n = np.cross(P[1] - P[0], P[3] - P[0]); n /= np.linalg.norm(n)
        m.add(mat, P, [[0, 0], [1, 0], [1, 1], [0, 1]], [n] * 4, [[0, 1, 2], [0, 2, 3]])

    def box(mat, c, half, rot):
        cs, sn = np.cos(rot), np.sin(rot)
        corners = []
        for dy in (0, 1):
            for dx, dz in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
                x, z = dx * half[0], dz * half[2]
                corners.append([c[0] + cs * x - sn * z, c[1] + dy * half[1] * 2, c[2] + sn * x + cs * z])
        cnr = np.array(corners)
        faces = [(0, 1, 2, 3), (4, 7, 6, 5), (0, 4, 5, 1), (1, 5, 6, 2), (2, 6, 7, 3), (3, 7, 4, 0)]
        for fc in faces:
            quad(mat, *[cnr[i] for i in fc])

    quad('white', [-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1])          # floor
    quad('white', [-1, 2, -1], [-1, 2, 1], [1, 2, 1], [1, 2, -1])          # ceiling
    quad('white', [-1, 0, -1], [-1, 2, -1], [1, 2, -1], [1, 0, -1])        # back
    quad('red', [-1, 0, -1], [-1, 0, 1], [-1, 2, 1], [-1, 2, -1])          # left
    quad('green', [1, 0, -1], [1, 2, -1], [1, 2, 1], [1, 0, 1])            # right
    box('white', [-0.35, 0, -0.3], [0.3, 0.6, 0.3], 0.3)
    box('white', [0.4, 0, 0.3], [0.3, 0.3, 0.3], -0.3)
    quad('light', [-0.25, 1.98, -0.25], [0.25, 1.98, -0.25], [0.25, 1.98, 0.25], [-0.25, 1.98, 0.25])
    with open(os.path.join(out_dir, 'cornell.mtl'), 'w') as f:
        f.write('newmtl white\nKd 0.73 0.73 0.73\nPr 1.0\n')
        f.write('newmtl red\nKd 0.65 0.05 0.05\nPr 1.0\n')
        f.write('newmtl green\nKd 0.12 0.45 0.15\nPr 1.0\n')
        f.write('newmtl light\nKd 0.78 0.78 0.78\nKe 15 15 15\nPr 1.0\n')
    m.write(os.path.join(out_dir, 'cornell.obj'), 'cornell.mtl')
    write_hdr(os.path.join(out_dir, 'black_4x2.hdr'), np.zeros((2, 4, 3), np.float32))
    return os.path.join(out_dir, 'cornell.obj')


def cornell_params(params):
    """Camera of the Cornell configuration: origin (0,100,350), looking down -z, fov 1."""
    from .types import Float3
    params.rayOrigin = Float3(0., 100., 350.)
    params.cameraAngle = Float3(0., 0., 0.)
    params.fov = 1.0
    params.isRendering = 1
    return params


# ---------------------------------------------------------------------------------------------- atrium
def _textures(out_dir, seed):
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(out_dir, 'tex'), exist_ok=True)
    n = 256
    yy, xx = np.meshgrid(np.arange(n), np.arange(n), indexing='ij')

    def noise(scale, octaves=4):
        out = np.zeros((n, n))
        for o in range(octaves):
            k = scale * (2 ** o)
            g = rng.uniform(0, 1, (k + 1, k + 1))
            gx = xx * k / n; gy = yy * k / n
            x0 = gx.astype(int); y0 = gy.astype(int)
            fx = gx - x0; fy = gy - y0
            fx = fx * fx * (3 - 2 * fx); fy = fy * fy * (3 - 2 * fy)
            a = g[y0 % k, x0 % k]; b = g[y0 % k, (x0 + 1) % k]; c = g[(y0 + 1) % k, x0 % k]; d = g[(y0 + 1) % k, (x0 + 1) % k]
            out += ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy) / (2 ** o)
        return out / out.max()

    def save(name, rgba):
        _write_png(os.path.join(out_dir, 'tex', name), np.clip(rgba * 255 + 0.5, 0, 255).astype(np.uint8))

    nz = noise(8)
    brick = ((yy // 32) % 2 * 32 + xx) % 64
    mortar = ((brick < 3) | (yy % 32 < 3)).astype(float)
    base = np.stack([0.55 + 0.25 * nz, 0.32 + 0.15 * nz, 0.25 + 0.1 * nz], -1)
    base = base * (1 - mortar[..., None]) + 0.7 * mortar[..., None]
    save('brick_diff.png', np.concatenate([base, np.ones((n, n, 1))], -1))
    hgt = (1 - mortar) * (0.8 + 0.2 * nz)
    dx = np.roll(hgt, -1, 1) - np.roll(hgt, 1, 1); dy = np.roll(hgt, -1, 0) - np.roll(hgt, 1, 0)
    nrm = np.stack([-dx * 2, -dy * 2, np.ones_like(dx)], -1); nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    save('brick_nor.png', np.concatenate([nrm * 0.5 + 0.5, np.ones((n, n, 1))], -1))
    r = 0.35 + 0.5 * noise(16)
    save('rough_a.png', np.stack([r, r, r, np.ones_like(r)], -1))
    chk = (((xx // 32) + (yy // 32)) % 2).astype(float)
    tile = np.stack([0.8 - 0.5 * chk, 0.78 - 0.5 * chk, 0.72 - 0.45 * chk], -1) * (0.85 + 0.15 * noise(4)[..., None])
    save('tile_diff.png', np.concatenate([tile, np.ones((n, n, 1))], -1))
    tr = 0.08 + 0.3 * chk + 0.1 * noise(8)
    save('tile_rough.png', np.stack([tr, tr, tr, np.ones_like(tr)], -1))
    marble = 0.5 + 0.5 * np.sin(xx / 9.0 + 6 * noise(4))
    mcol = np.stack([0.85 - 0.2 * marble, 0.83 - 0.22 * marble, 0.8 - 0.2 * marble], -1)
    save('marble_diff.png', np.concatenate([mcol, np.ones((n, n, 1))], -1))
    cloth_a = ((np.sin(xx / 3.0) * np.sin(yy / 3.0)) > -0.55).astype(float)
    stripes = 0.5 + 0.5 * np.sin(yy / 10.0)
    ccol = np.stack([0.7 * stripes + 0.15, 0.1 + 0.1 * stripes, 0.12 + 0.5 * (1 - stripes)], -1)
    save('cloth_diff.png', np.concatenate([ccol, cloth_a[..., None]], -1))
    mm = (noise(4) > 0.55).astype(float)
    save('metal_mask.png', np.stack([mm, mm, mm, np.ones_like(mm)], -1))
    glow = np.clip(1.2 - np.hypot(xx - n / 2, yy - n / 2) / (n / 2.5), 0, 1) * 6.0 / 6.0
    save('lamp_emis.png', np.stack([glow, glow, glow, np.ones_like(glow)], -1))


def atrium(out_dir, seed=1, detail=1.0, glossy_fraction=0.0, glass_fraction=0.0):
    """Colonnaded hall, about 262k triangles at detail=1 (scales ~detail^2). Returns the OBJ path.

    glossy_fraction / glass_fraction rewrite that share of the 25 materials to `Pm 1, Pr 0.1` and to
    `d 0, Ni 1.5, Pr 0` (the sun/sky + glossy/dielectric configuration)."""
    rng = np.random.default_rng(seed)
    os.makedirs(out_dir, exist_ok=True)
    _textures(out_dir, seed)
    s = lambda k: max(2, int(round(k * detail)))
    m = Mesh()
    LX, LZ, HY = 30.0, 14.0, 12.0       # hall length (x), width (z), height
    # floor, gallery floors, walls
    m.add('floor_tile', *grid([-LX / 2, 0, -LZ / 2], [LX, 0, 0], [0, 0, LZ], s(160), s(80), uv_scale=(30, 14)))
    m.add('gallery', *grid([-LX / 2, 5.0, -LZ / 2], [LX, 0, 0], [0, 0, 3.0], s(96), s(12), uv_scale=(30, 3)))
    m.add('gallery', *grid([-LX / 2, 5.0, LZ / 2 - 3.0], [LX, 0, 0], [0, 0, 3.0], s(96), s(12), uv_scale=(30, 3)))
    m.add('brick_n', *grid([-LX / 2, 0, -LZ / 2], [0, HY, 0], [LX, 0, 0], s(48), s(120), uv_scale=(6, 15)))
    m.add('brick_s', *grid([-LX / 2, 0, LZ / 2], [LX, 0, 0], [0, HY, 0], s(120), s(48), uv_scale=(15, 6)))
    m.add('brick_e', *grid([LX / 2, 0, -LZ / 2], [0, HY, 0], [0, 0, LZ], s(48), s(56), uv_scale=(6, 7)))
    m.add('brick_w', *grid([-LX / 2, 0, -LZ / 2], [0, 0, LZ], [0, HY, 0], s(56), s(48), uv_scale=(7, 6)))
    # colonnades: 2 rows x 11 columns, with arches between neighbours
    ncol = 11
    xs = np.linspace(-LX / 2 + 2.0, LX / 2 - 2.0, ncol)
    prof_y = np.concatenate([np.linspace(0, 0.4, s(4)), np.linspace(0.45, 4.3, s(22)), np.linspace(4.35, 5.0, s(6))])
    prof_r = np.concatenate([np.linspace(0.55, 0.42, s(4)), 0.36 - 0.03 * np.linspace(0, 1, s(22)), np.linspace(0.36, 0.6, s(6))])
    k = 0
    for zi, z in enumerate((-LZ / 2 + 3.0, LZ / 2 - 3.0)):
        for xi, x in enumerate(xs):
            m.add('column_%d' % (k % 4), *revolve([x, 0, z], prof_r, prof_y, s(40), uv_scale=(2, 4)))
            k += 1
            if xi + 1 < ncol:
                # arch: half torus between this column and the next
                cx = 0.5 * (x + xs[xi + 1]); R = 0.5 * (xs[xi + 1] - x); r = 0.22
                a = np.linspace(0, np.pi, s(28) + 1); b = np.linspace(0, 2 * np.pi, s(14) + 1)
                A, B = np.meshgrid(a, b, indexing='ij')
                P = np.stack([cx - (R + r * np.cos(B)) * np.cos(A), 5.0 + (R * 0.55 + r * np.cos(B)) * np.sin(A), z + r * np.sin(B)], -1)
                N = np.stack([-np.cos(B) * np.cos(A), np.cos(B) * np.sin(A), np.sin(B)], -1)
                T = np.stack([A / np.pi * 3, B / (2 * np.pi)], -1)
                idx = np.arange(P.shape[0] * P.shape[1]).reshape(P.shape[:2])
                q = (idx[:-1, :-1], idx[:-1, 1:], idx[1:, 1:], idx[1:, :-1])
                F = np.concatenate([np.stack([q[0], q[1], q[2]], -1).reshape(-1, 3), np.stack([q[0], q[2], q[3]], -1).reshape(-1, 3)])
                m.add('arch', P.reshape(-1, 3), T.reshape(-1, 2), N.reshape(-1, 3), F)
    # drapes hanging across the nave
    for i in range(6):
        x0 = -LX / 2 + 3.5 + i * 4.6
        ph = rng.uniform(0, 6.28)
        amp = rng.uniform(0.15, 0.35)
        m.add('cloth_%d' % (i % 3), *grid([x0, 9.5, -2.6], [0.0, -4.5, 0.0], [0.0, 0.0, 5.2], s(72), s(72), uv_scale=(3, 3),
                                          height=lambda U, W, ph=ph, amp=amp: amp * np.sin(9 * W + ph) * (0.3 + U) + 0.08 * np.sin(23 * U + ph)))
    # objects on the floor
    for i in range(6):
        x = -LX / 2 + 4.0 + i * 4.4
        z = (-1) ** i * 1.6
        mat = ['gold', 'steel', 'glass', 'copper', 'fog_glass', 'plastic'][i]
        fr = rng.uniform(3, 7)
        m.add(mat, *sphere([x, 0.9, z], 0.85, s(72), s(36), bump=(lambda d, fr=fr: 0.06 * np.sin(fr * d[:, 0:1] * 3) * np.sin(fr * d[:, 1:2] * 3)) if i % 2 == 0 else None))
    # lamps on the gallery undersides
    for i in range(8):
        x = -LX / 2 + 2.5 + i * 3.6
        for z in (-LZ / 2 + 1.5, LZ / 2 - 1.5):
            m.add('lamp', *grid([x, 4.95, z - 0.3], [0.6, 0, 0], [0, 0, 0.6], 2, 2))
    mats = {
        'floor_tile': 'Kd 0.8 0.8 0.8\nPr 0.3\nmap_Kd tex/tile_diff.png\nmap_Pr tex/tile_rough.png\n',
        'gallery': 'Kd 0.6 0.55 0.5\nPr 0.8\nmap_Kd tex/marble_diff.png\n',
        'brick_n': 'Kd 0.6 0.4 0.3\nPr 0.9\nmap_Kd tex/brick_diff.png\nmap_Bump -bm 1.0 tex/brick_nor.png\n',
        'brick_s': 'Kd 0.6 0.4 0.3\nPr 0.9\nmap_Kd tex/brick_diff.png\nmap_Pr tex/rough_a.png\n',
        'brick_e': 'Kd 0.6 0.4 0.3\nPr 0.85\nmap_Kd tex/brick_diff.png\nmap_Bump -bm 1.0 tex/brick_nor.png\nmap_Pr tex/rough_a.png\n',
        'brick_w': 'Kd 0.55 0.42 0.33\nPr 0.95\nmap_Kd tex/brick_diff.png\n',
        'column_0': 'Kd 0.82 0.8 0.75\nPr 0.6\nmap_Kd tex/marble_diff.png\n',
        'column_1': 'Kd 0.75 0.73 0.7\nPr 0.5\n',
        'column_2': 'Kd 0.7 0.7 0.72\nPr 0.4\nmap_Pr tex/rough_a.png\n',
        'column_3': 'Kd 0.8 0.75 0.65\nPr 0.7\nPs 0.3\n',
        'arch': 'Kd 0.7 0.65 0.6\nPr 0.75\n',
        'cloth_0': 'Kd 0.7 0.1 0.1\nPr 0.9\nPs 0.8\nmap_Kd tex/cloth_diff.png\n',
        'cloth_1': 'Kd 0.1 0.3 0.7\nPr 0.95\nPs 0.5\n',
        'cloth_2': 'Kd 0.2 0.6 0.2\nPr 0.85\nmap_Kd tex/cloth_diff.png\n',
        'gold': 'Kd 1.0 0.77 0.34\nPr 0.15\nPm 1.0\n',
        'steel': 'Kd 0.8 0.8 0.85\nPr 0.3\nPm 1.0\naniso 0.6\n',
        'glass': 'Kd 1.0 1.0 1.0\nPr 0.02\nd 0.05\nNi 1.5 \n',
        'copper': 'Kd 0.95 0.64 0.54\nPr 0.4\nPm 0.9\nmap_Pm tex/metal_mask.png\n',
        'fog_glass': 'Kd 1.0 1.0 1.0\nPr 0.0\nd 0.0\nNi 1.005 \nTf 0.995 0.99 0.98\nTs 0.004 0.004 0.004\nPg 0.6 \n',
        'plastic': 'Kd 0.2 0.5 0.9\nPr 0.25\n',
        'lamp': 'Kd 0.9 0.9 0.9\nKe 12 11 9\nPr 1.0\nmap_Ke tex/lamp_emis.png\n',
        'spare_0': 'Kd 0.5 0.5 0.5\nPr 0.5\n',
        'spare_1': 'Kd 0.4 0.4 0.4\nPr 0.6\n',
        'spare_2': 'Kd 0.3 0.3 0.3\nPr 0.7\n',
        'spare_3': 'Kd 0.2 0.2 0.2\nPr 0.8\n',
    }
    names = list(mats.keys())
    assert len(names) == 25
    order = rng.permutation(len(names))
    ng, nd = int(round(glossy_fraction * len(names))), int(round(glass_fraction * len(names)))
    for j in order[:ng]:
        if names[j] != 'lamp':
            mats[names[j]] += 'Pm 1\nPr 0.1\n'
    for j in order[ng:ng + nd]:
        if names[j] != 'lamp':
            mats[names[j]] += 'd 0\nNi 1.5 \nPr 0\n'
    with open(os.path.join(out_dir, 'atrium.mtl'), 'w') as f:
        for nme in names:
            f.write('newmtl %s\n%s' % (nme, mats[nme]))
    m.write(os.path.join(out_dir, 'atrium.obj'), 'atrium.mtl')
    return os.path.join(out_dir, 'atrium.obj')


def atrium_params(params):
    """Camera inside the nave looking along +x."""
    from .types import Float3
    params.rayOrigin = Float3(-1250., 330., 40.)
    params.cameraAngle = Float3(float(np.float32(np.pi / 2)), -0.08, 0.)
    params.fov = 1.0
    params.isRendering = 1
    return params


# ---------------------------------------------------------------------------------------------- soup
def soup(n, seed=2, n_mats=64, extent=5000.0, edge=5.0):
    """n random small triangles as host arrays (already in scene units).

    The OBJ route is not used: three fresh vertices per triangle would exceed the triangle count and
    the reference loader would reset those indices to vertex 0 (file_reader.h:386-397)."""
    rng = np.random.default_rng(seed)
    tris = np.zeros(n, dtype=TRI_DTYPE)
    c = rng.uniform(-extent, extent, (n, 3)).astype(np.float32)
    e1 = rng.normal(0, edge, (n, 3)).astype(np.float32)
    e2 = rng.normal(0, edge, (n, 3)).astype(np.float32)
    tris['a'] = c; tris['b'] = c + e1; tris['c'] = c + e2
    nrm = np.cross(e1.astype(np.float64), e2.astype(np.float64))
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-20)
    for k in ('nA', 'nB', 'nC'):
        p = nrm + rng.normal(0, 0.1, (n, 3))
        p /= np.linalg.norm(p, axis=1, keepdims=True)
        tris[k] = p.astype(np.float32)
    tris['tA'] = rng.uniform(0, 1, (n, 2)); tris['tB'] = rng.uniform(0, 1, (n, 2)); tris['tC'] = rng.uniform(0, 1, (n, 2))
    tris['matIndex'] = rng.integers(0, n_mats, n)
    mats = default_materials(n_mats)
    mats['baseColor'] = rng.uniform(0.2, 0.9, (n_mats, 3))
    mats['roughness'] = rng.uniform(0.1, 1.0, n_mats)
    mats['metallic'] = (rng.uniform(0, 1, n_mats) > 0.7).astype(np.float32)
    return tris, mats


def random_rays(n, seed, extent, inside_fraction=1.0):
    """Rays with origins uniform in a cube and uniformly distributed directions."""
    from .types import RAY_DTYPE
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, dtype=RAY_DTYPE)
    rays['o'] = rng.uniform(-extent, extent, (n, 3))
    d = rng.normal(0, 1, (n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays['d'] = d
    return rays
