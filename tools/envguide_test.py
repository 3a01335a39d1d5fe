"""Guided environment-map search against the bisection: renders with guides on and off must agree bit for bit."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'cuda-pathtracer_amd', 'python')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ptamd, util
from ptamd import scenes, lib
from util import bits
b = util.cornell('sky')
P = scenes.cornell_params(ptamd.default_params(96, 64))
rng = np.random.default_rng(3)
maps = {'sky64x32': b.env,
        'noise256x128': np.concatenate([rng.uniform(0, 2, (128, 256, 3)).astype(np.float32) ** 6, np.zeros((128, 256, 1), np.float32)], -1),
        'constant33x17': np.concatenate([np.full((17, 33, 3), 0.5, np.float32), np.zeros((17, 33, 1), np.float32)], -1),
        'onehot100x50': None, 'sky1024x512': None}
oh = np.zeros((50, 100, 4), np.float32); oh[7, 93, :3] = 1000.0; oh[30, 2, :3] = 1e-3; maps['onehot100x50'] = oh
e = scenes.sky_envmap(1024, 512); maps['sky1024x512'] = np.concatenate([e, np.zeros(e.shape[:2] + (1,), np.float32)], -1)
for name, env in maps.items():
    cdf, total = lib.build_env_cdf(env)
    out = []
    for on in (True, False):
        r = ptamd.Renderer(); r.init(P)
        r.upload_scene(b.tris, b.mats, b.idx, b.nodes)
        r.upload_envmap(env, cdf, total, P)
        ok = r.set_env_guides(on)
        r.set_max_bounces(6)
        r.render(P, 0, 4); out.append(r.read_accum()); r.close()
    print(name, 'verified', ok, 'identical', bool(np.array_equal(bits(out[0]), bits(out[1]))), 'mean', float(out[0].mean()))
