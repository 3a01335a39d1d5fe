#!/usr/bin/env python3
"""Benchmark of the render hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

Workload: the Sponza-class atrium (about 262k triangles, 25 PBR materials
with texture maps, 1024x512 HDR environment), 1920x1080, 64 samples per pixel per GPU per step, bounce
cap 512 as in the reference.  One step = clear the accumulator, render this rank's 64 frames, sum the
per-GPU accumulators on rank 0 (RCCL reduce, N > 1 only), resolve to RGBA8.  Ranks render disjoint
frame ranges (sample partition), so work per GPU is fixed as N grows: weak scaling.

Prints one JSON line: metric Mrays/s (a ray is one closest-hit query: camera, bounce or shadow-walk
segment), with samples/s, the roofline of the dominant kernel and the CPU oracle timed beside it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'cuda-pathtracer_amd', 'python'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E datasheet peak


def algorithmic_bytes(c):
    """Bytes the path must move per the layout-independent accounting: per ray 32 in + 8 out, 32 per
    ray/box test, 48 per ray/triangle test; per shaded vertex 64 (normals, uv, material index) + 64
    (material); 64 per 4-channel bilinear lookup, 16 per 1-channel lookup; 24 per sample (accumulator)."""
    return (40 * c['rays'] + 32 * c['box_tests'] + 48 * c['tri_tests'] + 128 * c['shade_vertices']
            + 64 * c['tex_taps'] + 16 * c['tex_taps_r'] + 24 * c['samples'])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--width', type=int, default=1920)
    ap.add_argument('--height', type=int, default=1080)
    ap.add_argument('--spp', type=int, default=64, help='samples per pixel per GPU per step')
    ap.add_argument('--bounces', type=int, default=512)
    ap.add_argument('--detail', type=float, default=None, help='atrium tessellation (default: about 262k triangles)')
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU oracle baseline')
    ap.add_argument('--cpu-spp', type=int, default=12, help='frames the CPU oracle renders for the baseline (about 15 s on 16 cores)')
    ap.add_argument('--mode', choices=['wavefront', 'single_kernel'], default='wavefront')
    ap.add_argument('--pool', type=int, default=0, help='wavefront pool slots (0 = library default)')
    ap.add_argument('--chunk-frames', type=int, default=0, help='frames per staging chunk (0 = library default)')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl is RCCL; gloo lets several ranks rehearse on one GPU)')
    ap.add_argument('--device', type=int, default=-1, help='GPU to use (default: LOCAL_RANK)')
    ap.add_argument('--fp32-textures', action='store_true', help='keep 8-bit textures as fp32 texels instead of bytes')
    ap.add_argument('--repack', type=int, default=-1, help='repack threshold in percent of the pool (0 = never, -1 = library default)')
    ap.add_argument('--env-guides', type=int, default=1, help='environment CDF search: 1 all guide tables, 2 without the whole-range table, 0 plain bisections')
    ap.add_argument('--lanes', type=int, default=0, help='concurrent wavefront lanes (0 = library default)')
    ap.add_argument('--drain', type=int, default=-1, help='finish the last paths in one kernel once at most this many remain (-1 = library default, 0 = never)')
    ap.add_argument('--node-width', type=int, default=4, help='4-wide or binary (2) compact nodes in the traversal kernel')
    ap.add_argument('--tune', type=str, default='', help='refill_lanes,leaf_vote,blocks_per_cu for the traversal kernel')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import ptamd
    from ptamd import scenes
    import util

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node %d' % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the render path has no CPU fallback')
    device = local_rank if args.device < 0 else args.device
    torch.cuda.set_device(device)
    if world > 1:
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', device))
        else:
            dist.init_process_group(args.backend)

    W, H, spp = args.width, args.height, args.spp
    detail = args.detail if args.detail is not None else util.ATRIUM_DETAIL
    # every rank generates the same seeded scene in its own directory
    util.SCRATCH = os.path.join(ROOT, 'tests', '_scratch', 'rank%d' % rank)
    bundle = util.atrium(detail=detail)
    P = scenes.atrium_params(ptamd.default_params(W, H))
    r = bundle.renderer(P, device, pack_textures=not args.fp32_textures)
    r.set_max_bounces(args.bounces)
    r.set_mode(0 if args.mode == 'wavefront' else 1)
    if args.pool or args.chunk_frames:
        r.set_pool(args.pool or (1 << 24), args.chunk_frames or 512)
    r.set_node_width(args.node_width)
    if args.drain >= 0:
        r.set_drain(args.drain)
    if args.lanes:
        r.set_lanes(args.lanes)
    if args.repack >= 0:
        r.set_repack(args.repack)
    if args.env_guides != 1:
        r.set_env_guides(args.env_guides)
    if args.tune:
        r.set_extend_tuning(*[int(x) for x in args.tune.split(',')])
    # Everything of a step (clear, render, reduce, resolve) is ordered on one explicit stream that both torch
    # and the library use. torch's default stream has handle 0, which the library would read as "use the
    # context's own stream": the clear and the RCCL reduce would then not be ordered with the render.
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    r.set_stream(stream.cuda_stream)
    accum = torch.zeros((H, W, 3), dtype=torch.float32, device='cuda')
    r.bind_accum(accum.data_ptr())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    kernel_ms = []

    def step(k, timed):
        f0, f1 = ptamd.partition.step_frames(k, world, rank, spp)
        accum.zero_()
        r.render(P, f0, f1, sync=False)
        ptamd.partition.reduce_accum(accum, dst=0)
        if rank == 0:
            r.resolve_display(spp * world)
        else:
            torch.cuda.synchronize()
        if timed:
            r.sync()
            kernel_ms.append(r.counters()['last_ms'])

    for k in range(args.warmup):
        step(k, False)
    r.stage_timing(True)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k, True)
    barrier()
    elapsed = time.perf_counter() - t0
    extend_ms, extend_launches = r.stage_timing(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # untimed: count rays and tests of exactly the timed frames with the instrumented kernels
    r.enable_counters(True)
    r.reset_counters()
    for k in range(args.steps):
        f0, f1 = ptamd.partition.step_frames(args.warmup + k, world, rank, spp)
        r.render(P, f0, f1)
    c = r.counters()
    r.enable_counters(False)
    keys = ['rays', 'samples', 'box_tests', 'tri_tests', 'shade_vertices', 'env_samples', 'tex_taps', 'tex_taps_r']
    local = torch.tensor([float(c[k]) for k in keys], dtype=torch.float64, device='cuda')
    total = local.clone()
    if world > 1:
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
    tot = {k: float(v) for k, v in zip(keys, total.tolist())}
    loc = {k: float(v) for k, v in zip(keys, local.tolist())}

    result = None
    if rank == 0:
        mrays = tot['rays'] / elapsed / 1e6
        if args.mode == 'wavefront':
            # dominant kernel: extend (closest hit for every pending ray of the pool), one launch per iteration
            kname = 'wf_extend_kernel'
            kms = extend_ms / max(extend_launches, 1)
            alg_per_launch = (40 * loc['rays'] + 32 * loc['box_tests'] + 48 * loc['tri_tests']) / max(extend_launches, 1)
        else:
            kname = 'render_samples_kernel'
            kms = float(np.mean(kernel_ms))
            alg_per_launch = algorithmic_bytes(loc) / args.steps
        achieved = alg_per_launch / (kms * 1e-3) / 1e9
        node_b, tri_b = r.scene_bytes()
        traffic = None
        valu = {}
        pmc_path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.exists(pmc_path):
            try:
                with open(pmc_path) as f:
                    pm = json.load(f)
                if pm.get('workload') == [W, H, spp, args.bounces]:
                    traffic = pm.get('hbm_bytes_per_launch')
                    valu = {k: pm[k] for k in ('valu_busy', 'valu_lanes_per_instruction', 'valu_note') if k in pm}
            except Exception:
                traffic = None
        result = {
            'metric': 'Mrays/s', 'value': round(mrays, 3), 'unit': 'Mrays/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'samples_per_s': round(tot['samples'] / elapsed, 1),
            'rays_per_sample': round(tot['rays'] / tot['samples'], 3),
            'config': {'workload': 'atrium (Sponza-class, %d triangles, 25 PBR materials, 1024x512 HDR envmap) %dx%d, %d spp per GPU per step, bounce cap %d'
                                   % (len(bundle.tris), W, H, spp, args.bounces),
                       'partition': 'samples: rank r renders frames [(k*N+r)*spp, +spp), RCCL reduce to rank 0' if world > 1 else 'single GPU',
                       'scene_bytes': {'nodes': node_b, 'triangles': tri_b}},
            'roofline': {'bound': 'hbm', 'kernel': kname, 'launches': int(extend_launches) if args.mode == 'wavefront' else args.steps, 'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': traffic,
                         'kernel_ms': round(kms, 4),
                         'note': 'algorithmic bytes = 40 per ray + 32 per box test + 48 per triangle test of the reference-order traversal (counted); the scene is cache resident and the kernel walks 4-wide nodes, so fewer bytes reach HBM (traffic) and frac can exceed 1',
                         'whole_path_algorithmic_GBs': round(algorithmic_bytes(loc) / (elapsed * 1e9), 2), 'algorithmic_bytes_per_launch': int(alg_per_launch), **valu},
        }

    # CPU baseline: the oracle on this host's cores, rank 0, single-GPU runs only, bounded sample
    if rank == 0 and world == 1 and not args.no_cpu:
        import orc
        S = bundle.oracle(P)
        cpu_spp = args.cpu_spp
        t1 = time.perf_counter()
        _, st = orc.render(S, P, 0, cpu_spp, args.bounces)
        dt = time.perf_counter() - t1
        result['cpu_baseline'] = {'value': round(st['rays'] / dt / 1e6, 3), 'unit': 'Mrays/s', 'cores': orc.max_threads(), 'kind': 'port',
                                  'samples_per_s': round(st['samples'] / dt, 1),
                                  'sample': 'same scene and camera, %dx%d, frames [0,%d), %.1f s' % (W, H, cpu_spp, dt)}
    if rank == 0:
        print(json.dumps(result))
    r.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
