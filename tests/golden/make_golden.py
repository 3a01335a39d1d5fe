#!/usr/bin/env python3
"""Writes tests/golden/vectors.npz from the CPU oracle (run from the repository root after `make -C oracle`).

The reference ships no tests, golden images or expected outputs, so these vectors are NOT reference outputs:
they freeze what the oracle computes today on fixed inputs, so that a later edit, compiler or flag change
that alters a single bit is caught, and so that the GPU tests have a target that does not depend on running
the oracle.  Inputs that are not regenerated deterministically from code (the rays) are stored in the file.
bunny_triangle_2.obj.txt is the one mesh the reference ships (assets/models/Bunny_triangle_2.txt), kept as data.
"""
import os
import shutil
import sys
import tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'cuda-pathtracer_amd', 'python'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import ptamd  # noqa: E402
from ptamd import scenes  # noqa: E402
from ptamd.types import RAY_DTYPE  # noqa: E402

_cache = {}


def bunny_bundle():
    import util
    if 'bunny' not in _cache:
        d = tempfile.mkdtemp(prefix='bunny_')
        try:
            shutil.copy(os.path.join(HERE, 'bunny_triangle_2.obj.txt'), os.path.join(d, 'Bunny_triangle_2.txt'))
            with open(os.path.join(d, 'Bunny_triangle_2.mtl'), 'w') as f:
                f.write('newmtl none\n')
            hs = ptamd.HostScene(os.path.join(d, 'Bunny_triangle_2.txt'), d + '/')
        finally:
            shutil.rmtree(d, ignore_errors=True)
        _cache['bunny'] = util.Bundle(hs.tris, hs.mats)
    return _cache['bunny']


def cornell_bundle():
    import util
    b = util.cornell('black')
    P = scenes.cornell_params(ptamd.default_params(16, 16))
    return b, P


def main():
    import orc
    from util import bits
    out = {}
    out['stream_pixel'], out['stream_frame'] = 4242, 3
    out['stream'] = orc.uniform_stream(4242, 3, 64)
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-20, 20, 128), np.exp(rng.uniform(-30, 30, 64)), rng.uniform(-1, 1, 64)]).astype(np.float32)
    y = rng.uniform(-4, 4, 256).astype(np.float32)
    out['probe_x'], out['probe_y'] = x, y
    out['probe_bits'] = bits(orc.detmath_probe(x, y))
    bb = bunny_bundle()
    out['bunny_nodes'] = bb.nodes.view(np.uint8).reshape(len(bb.nodes), -1)
    out['bunny_idx'] = bb.idx
    rays = np.zeros(512, dtype=RAY_DTYPE)
    centre = (bb.nodes['aabbMin'][0] + bb.nodes['aabbMax'][0]) / 2
    half = (bb.nodes['aabbMax'][0] - bb.nodes['aabbMin'][0]) / 2
    o = rng.normal(0, 1, (512, 3)); o /= np.linalg.norm(o, axis=1, keepdims=True)
    rays['o'] = centre + o * half.max() * 2.5
    target = centre + rng.uniform(-1, 1, (512, 3)) * half * 0.9
    d = target - rays['o']
    rays['d'] = d / np.linalg.norm(d, axis=1, keepdims=True)
    out['bunny_rays'] = rays.view(np.uint8).reshape(512, -1)
    h = orc.closest_brute(bb.oracle(), rays)
    out['bunny_hit_tri'] = h['tri']
    out['bunny_hit_t_bits'] = bits(h['t'])
    cb, P = cornell_bundle()
    frames, bounces = 8, 4
    acc, st = orc.render(cb.oracle(P), P, 0, frames, bounces)
    out['cornell_frames'], out['cornell_bounces'] = frames, bounces
    out['cornell_accum_bits'] = bits(acc)
    out['cornell_rays'] = st['rays']
    out['cornell_rgba'] = orc.resolve_display(acc, frames)
    np.savez_compressed(os.path.join(HERE, 'vectors.npz'), **out)
    print('bunny hits %d of 512; cornell rays %d, mean radiance %.4f' % ((h['tri'] != 0xFFFFFFFF).sum(), st['rays'], acc.mean()))


if __name__ == '__main__':
    main()
