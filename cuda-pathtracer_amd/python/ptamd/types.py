"""Numpy / ctypes views of the C-ABI PODs declared in include/pt_capi.h.

Layouts follow the reference's structs (Cuda Pathtracer/core/CUDA/utility/structs.cuh:111-181).
"""
import ctypes as C
import numpy as np

TRI_DTYPE = np.dtype([('a', '<f4', 3), ('b', '<f4', 3), ('c', '<f4', 3), ('origin', '<f4', 3),
                      ('nA', '<f4', 3), ('nB', '<f4', 3), ('nC', '<f4', 3),
                      ('tA', '<f4', 2), ('tB', '<f4', 2), ('tC', '<f4', 2), ('matIndex', '<i4')])
NODE_DTYPE = np.dtype([('aabbMin', '<f4', 3), ('aabbMax', '<f4', 3), ('leftFirst', '<i4'), ('triCount', '<i4')])
MAT_DTYPE = np.dtype({
    'names': ['baseColor', 'emissive', 'roughness', 'anisotropic', 'metallic', 'subsurface', 'sheen', 'clearcoat',
              'clearcoatRoughness', 'specTrans', 'alpha', 'ior', 'G', 'sigmaA', 'sigmaS',
              'diffuseTexture', 'useTexture', 'roughnessTexture', 'use_mapPr', 'metallicTexture', 'use_mapPm',
              'emissiveTexture', 'use_mapKe', 'normalTexture', 'use_mapNor'],
    'formats': [('<f4', 3), ('<f4', 3), '<f4', '<f4', '<f4', '<f4', '<f4', '<f4', '<f4', '<f4', '<f4', '<f4', '<f4',
                ('<f4', 3), ('<f4', 3), '<u8', 'u1', '<u8', 'u1', '<u8', 'u1', '<u8', 'u1', '<u8', 'u1'],
    'offsets': [0, 12, 24, 28, 32, 36, 40, 44, 48, 52, 56, 60, 64, 68, 80,
                96, 104, 112, 120, 128, 136, 144, 152, 160, 168],
    'itemsize': 176})
RAY_DTYPE = np.dtype([('o', '<f4', 3), ('_p0', '<f4'), ('d', '<f4', 3), ('_p1', '<f4')])
HIT_DTYPE = np.dtype([('t', '<f4'), ('tri', '<u4')])
assert TRI_DTYPE.itemsize == 112 and NODE_DTYPE.itemsize == 32 and MAT_DTYPE.itemsize == 176
assert RAY_DTYPE.itemsize == 32 and HIT_DTYPE.itemsize == 8


class Float3(C.Structure):
    _fields_ = [('x', C.c_float), ('y', C.c_float), ('z', C.c_float)]


class Int2(C.Structure):
    _fields_ = [('x', C.c_int), ('y', C.c_int)]


class Params(C.Structure):
    """pt_params == kernelParams (structs.cuh:149-181), 152 bytes."""
    _fields_ = [('windowSize', Int2), ('rayOrigin', Float3), ('rayDirectionZ', Float3), ('cameraAngle', Float3),
                ('cameraSpeed', C.c_float), ('sunDirection', Float3), ('focalDistance', C.c_float),
                ('DOF_strenght', C.c_float), ('fov', C.c_float), ('frameIndex', C.c_int), ('_pad0', C.c_uint32),
                ('pixelBuffer', C.c_void_p), ('isRendering', C.c_uint8), ('_p1', C.c_uint8 * 3),
                ('envmap_sum', C.c_float), ('envMap_size', Int2), ('cubeMap', C.c_uint64), ('envMap_cdf', C.c_uint64),
                ('mult', Float3), ('gamma', C.c_float), ('contrast', C.c_float), ('saturation', C.c_float),
                ('exposure', C.c_float), ('_pad1', C.c_uint32)]


assert C.sizeof(Params) == 152


class Counters(C.Structure):
    _fields_ = [('rays', C.c_uint64), ('samples', C.c_uint64), ('box_tests', C.c_uint64), ('tri_tests', C.c_uint64),
                ('shade_vertices', C.c_uint64), ('env_samples', C.c_uint64), ('tex_taps', C.c_uint64), ('tex_taps_r', C.c_uint64),
                ('wide_nodes', C.c_uint64), ('wide_tris', C.c_uint64), ('path_rays', C.c_uint64), ('last_ms', C.c_double)]


def default_params(width, height):
    """The application defaults of the reference (main.cpp:181-202), at a chosen window size."""
    p = Params()
    p.windowSize = Int2(width, height)
    p.rayDirectionZ = Float3(0., 0., -1.)
    p.cameraAngle = Float3(0., 0., 0.)
    p.rayOrigin = Float3(0., 1.5, 5.)
    p.cameraSpeed = 720.
    p.focalDistance = 200.
    p.DOF_strenght = 0.
    p.fov = 1.
    p.frameIndex = 0
    p.isRendering = 0
    p.mult = Float3(1., 1., 1.)
    p.gamma = p.saturation = p.contrast = p.exposure = 1.
    n = np.float32(1.) / np.sqrt(np.float32(1. + 0.49 + 1.))
    p.sunDirection = Float3(float(n), float(np.float32(0.7) * n), float(-n))
    return p


def default_materials(n):
    """n default-constructed materials (structs.cuh:75-94)."""
    m = np.zeros(n, dtype=MAT_DTYPE)
    m['roughness'] = 0.01
    m['ior'] = 1.5
    m['alpha'] = 1.0
    return m
