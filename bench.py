#!/usr/bin/env python3
"""Benchmark of the render hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

Workload (default): the Sponza-class atrium (about 262k triangles, 25 PBR materials
with texture maps, 1024x512 HDR environment), 1920x1080, 64 samples per pixel per GPU per step, bounce
cap 512 as in the reference.  One step = clear the accumulator, render this rank's 64 frames, sum the
per-GPU accumulators on rank 0 (RCCL reduce, N > 1 only), resolve to RGBA8.  Ranks render disjoint
frame ranges (sample partition), so work per GPU is fixed as N grows: weak scaling.
--total-spp S fixes the samples per pixel of a step over all GPUs instead (rank r renders frames
[r*S/N, (r+1)*S/N) of each step): strong scaling, the form of the 4096-spp and 1024-spp configurations.
--workload soup renders the 10-million-triangle soup (tree built on the GPU), whose 0.8 GB of nodes and
triangle records do not fit any cache: the case in which HBM bounds the traversal kernel.
--workload sunsky renders the atrium with glossy metal and glass materials under the sun/sky model.

With --gpus N > 1 and no launcher around it (WORLD_SIZE unset) the script starts itself N times under
torch.distributed.run as a child process and relays rank 0's line and the exit status.

Prints one JSON line: metric Mrays/s (a ray is one closest-hit query: camera, bounce or shadow-walk
segment), with samples/s, the roofline of the dominant kernel, a table of every stage kernel, and the CPU
oracle timed beside it.
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

N_SIMDS, CLOCK_GHZ = 1024, 2.4      # 256 CUs x 4 SIMDs; one wave64 vector instruction per SIMD every 4 cycles
VALU_PEAK_GINST = N_SIMDS * CLOCK_GHZ / 4
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E datasheet peak
STAGES = ('extend', 'path_hit', 'unused', 'miss', 'shadow', 'bsdf')      # the slots of pt_stage_times
KERNEL_NAMES = {'extend': 'wf_extend_kernel', 'path_hit': 'wf_pathhit_kernel', 'miss': 'wf_miss_kernel', 'shadow': 'wf_shadow_kernel', 'bsdf': 'wf_bsdf_kernel'}


def algorithmic_bytes(c):
    """Bytes the path must move per the layout-independent accounting: per ray 32 in + 8 out, 32 per
    ray/box test, 48 per ray/triangle test; per shaded vertex 64 (normals, uv, material index) + 64
    (material); 64 per 4-channel bilinear lookup, 16 per 1-channel lookup; 24 per sample (accumulator)."""
    return (40 * c['rays'] + 32 * c['box_tests'] + 48 * c['tri_tests'] + 128 * c['shade_vertices']
            + 64 * c['tex_taps'] + 16 * c['tex_taps_r'] + 24 * c['samples'])


def requested_bytes(c, packed_textures):
    """Bytes each stage kernel asks the memory system for over the counted frames, from the records it reads and
    writes (cuda-pathtracer_amd/csrc/pt_wavefront.hip) and the tallies of the instrumented run: c['path_rays'] path rays
    of which c['shade_vertices'] hit a surface, the rest leave the scene; c['rays'] - c['path_rays'] segments of
    visibility walks; one bsdf evaluation per vertex."""
    hits = c['shade_vertices']
    misses = max(c['path_rays'] - hits, 0.0)
    segments = max(c['rays'] - c['path_rays'], 0.0)
    tap4, tap1 = (16, 4) if packed_textures else (64, 16)       # four texels of a filtered lookup
    env = c['env_samples']
    material_taps = max(c['tex_taps'] - env - misses, 0.0)      # 4-channel lookups that are not environment lookups
    return {
        # both queues: origin + direction in (32), hit out (16); 64 per wide node visited, 48 per triangle tested
        'extend': 48 * c['rays'] + 64 * c['wide_nodes'] + 48 * c['wide_tris'],
        # hit and path record in (64), shading triangle (112), material (128), texels; light sample: guide, column table, per-texel values (48);
        # out: a whole shadow record for a vertex with a light sample (128), a path record for a ray that passed through (48); misses listed (4)
        'path_hit': 64 * c['path_rays'] + (112 + 128) * hits + tap4 * material_taps + tap1 * c['tex_taps_r'] * 0.5 + (48 + 128) * env + 48 * max(hits - env, 0.0) + 4 * misses,
        # list entry and path record in (52), four environment texels (64), a new camera ray out (48)
        'miss': (52 + 64 + 48) * misses,
        # direction, hit and state in (48), uv words of the shading triangle (32), material (128), alpha/metallic texels; state out (16), list entry (4)
        'shadow': (48 + 32 + 128 + 20) * segments + tap1 * c['tex_taps_r'] * 0.5,
        # list entry and whole record in (132), the material's constants (96, from the caches), continuing path out (48)
        'bsdf': (132 + 96 + 48) * hits,
    }


def self_launch(n):
    """Starts this script n times under torch.distributed.run as a child process; returns its exit status."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    for l in p.stdout.splitlines():
        if not l.startswith('{'):
            print(l, file=sys.stderr)
    if lines:
        print(lines[-1])
    return p.returncode if (lines or p.returncode) else 1


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
    ap.add_argument('--workload', choices=['atrium', 'soup', 'sunsky'], default='atrium',
                    help='atrium: the Sponza-class scene with its HDR environment; soup: 10 M random triangles; sunsky: the atrium with 30 %% glossy metal and 20 %% glass materials under the sun/sky model, bounce cap 16')
    ap.add_argument('--soup-tris', type=int, default=10000000)
    ap.add_argument('--total-spp', type=int, default=0, help='samples per pixel of one step over ALL GPUs (strong scaling); 0 = --spp per GPU (weak scaling)')
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU oracle baseline')
    ap.add_argument('--cpu-spp', type=int, default=0, help='frames the CPU oracle renders for the baseline (0: 12 for the atrium workloads, 1 for the soup: about 15 s on 16 cores)')
    ap.add_argument('--mode', choices=['wavefront', 'single_kernel'], default='wavefront')
    ap.add_argument('--pool', type=int, default=0, help='samples in flight (0 = library default)')
    ap.add_argument('--chunk-frames', type=int, default=0, help='frames per staging chunk (0 = library default)')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl is RCCL; gloo lets several ranks rehearse on one GPU)')
    ap.add_argument('--device', type=int, default=-1, help='GPU to use (default: LOCAL_RANK)')
    ap.add_argument('--fp32-textures', action='store_true', help='keep 8-bit textures as fp32 texels instead of bytes')
    ap.add_argument('--env-guides', type=int, default=1, help='environment CDF search: 1 all guide tables, 2 without the whole-range table, 0 plain bisections')
    ap.add_argument('--lanes', type=int, default=0, help='concurrent wavefront lanes (0 = library default)')
    ap.add_argument('--drain', type=int, default=-1, help='finish the last paths in one kernel once at most this many remain (-1 = library default, 0 = never)')
    ap.add_argument('--node-width', type=int, default=4, help='4-wide or binary (2) compact nodes in the traversal kernel')
    ap.add_argument('--leaf-merge', type=int, default=2, help='triangles per wide leaf at most')
    ap.add_argument('--tree', choices=['host', 'lbvh', 'ploc'], default='', help='tree builder: host (the reference builder; atrium default), lbvh (radix tree on the GPU; soup default), ploc (agglomerative on the GPU)')
    ap.add_argument('--tune', type=str, default='', help='refill_lanes,leaf_vote,blocks_per_cu for the traversal kernel')
    args = ap.parse_args()

    # launched plainly for several GPUs: become the launcher (a child process; this one has not touched the GPU)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist
    import ptamd
    from ptamd import scenes
    import util

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d does not match WORLD_SIZE %d' % (args.gpus, world))
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
    strong = args.total_spp > 0
    if strong and args.total_spp < world:
        raise SystemExit('--total-spp must be at least the number of GPUs')
    detail = args.detail if args.detail is not None else util.ATRIUM_DETAIL
    # every rank generates the same seeded scene in its own directory
    util.SCRATCH = os.path.join(ROOT, 'tests', '_scratch', 'rank%d' % rank)
    bundle = None
    soup_arrays = None
    if args.workload == 'soup':
        from ptamd import lib as ptlib
        tris, mats = scenes.soup(args.soup_tris, seed=2, extent=5000.0, edge=5.0)
        env = scenes.sky_envmap(256, 128)
        env4 = np.concatenate([env, np.zeros(env.shape[:2] + (1,), np.float32)], -1)
        cdf, env_total = ptlib.build_env_cdf(env4)
        P = ptamd.default_params(W, H)
        P.rayOrigin = ptamd.Float3(0.0, 0.0, 9000.0)
        P.isRendering = 1
        r = ptamd.Renderer(device)
        r.init(P)
        soup_flavour = ptlib.PT_BVH_GPU_PLOC if args.tree == 'ploc' else (ptlib.PT_BVH_REFERENCE if args.tree == 'host' else ptlib.PT_BVH_GPU_LBVH)
        r.upload_scene(tris, mats, None, None, flavour=soup_flavour)
        r.upload_envmap(env4, cdf, env_total, P)
        n_tris = len(tris)
        soup_arrays = (tris, mats, env4)
        workload_name = 'soup of %d random triangles (%s tree built on the GPU), 256x128 HDR envmap' % (n_tris, 'agglomerative' if args.tree == 'ploc' else 'radix')
    else:
        if args.workload == 'sunsky':
            bundle = util.atrium(detail=detail, glossy=0.3, glass=0.2, sky_mode=1)
            if args.bounces == 512:
                args.bounces = 16
        else:
            bundle = util.atrium(detail=detail)
        P = scenes.atrium_params(ptamd.default_params(W, H))
        r = bundle.renderer(P, device, pack_textures=not args.fp32_textures)
        n_tris = len(bundle.tris)
        workload_name = 'atrium (Sponza-class, %d triangles, 25 PBR materials, %s)' % (
            n_tris, '1024x512 HDR envmap' if args.workload == 'atrium' else '30 % glossy metal and 20 % glass materials, sun/sky model')
    r.set_max_bounces(args.bounces)
    r.set_mode(0 if args.mode == 'wavefront' else 1)
    if args.pool or args.chunk_frames:
        r.set_pool(args.pool or (1 << 27), args.chunk_frames or 512)
    if args.node_width != 4 or args.leaf_merge != 2:
        r.set_node_width(args.node_width, args.leaf_merge)
        if bundle is not None:
            r.upload_scene(bundle.tris, bundle.mats, bundle.idx, bundle.nodes)
        else:
            r.upload_scene(tris, mats, None, None, flavour=soup_flavour)
    if bundle is not None and args.tree in ('lbvh', 'ploc'):
        from ptamd import lib as ptlib
        r.upload_scene(bundle.tris, bundle.mats, None, None, flavour=ptlib.PT_BVH_GPU_PLOC if args.tree == 'ploc' else ptlib.PT_BVH_GPU_LBVH)
        workload_name += ', %s tree built on the GPU' % ('agglomerative' if args.tree == 'ploc' else 'radix')
    if args.drain >= 0:
        r.set_drain(args.drain)
    if args.lanes:
        r.set_lanes(args.lanes)
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

    def frames_of(k):
        """Frame range this rank renders in step k."""
        if strong:
            a, b = ptamd.partition.split_frames(args.total_spp, world, rank)
            return k * args.total_spp + a, k * args.total_spp + b
        return ptamd.partition.step_frames(k, world, rank, spp)

    step_spp = args.total_spp if strong else spp * world       # samples per pixel one step adds over all GPUs

    def step(k, timed):
        f0, f1 = frames_of(k)
        accum.zero_()
        r.render(P, f0, f1, sync=False)
        ptamd.partition.reduce_accum(accum, dst=0)
        if rank == 0:
            r.resolve_display(step_spp)
        else:
            torch.cuda.synchronize()
        if timed:
            r.sync()
            kernel_ms.append(r.counters()['last_ms'])

    for k in range(args.warmup):
        step(k, False)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k, True)
    barrier()
    elapsed = time.perf_counter() - t0
    image_mean = float(accum.mean().item()) / step_spp if rank == 0 else 0.0      # of the last step's reduced image (a checksum for N-GPU vs 1-GPU runs)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # untimed: the last timed step once more with HIP events around every stage kernel (on the launch streams), in the
    # timed configuration and on one lane, where a step's kernel times add up (the same thing unless --lanes 2 makes the
    # kernels of two lanes share the GPU)
    stage = {}
    if args.mode == 'wavefront':
        for tag in ('timed_config', 'one_lane'):
            if tag == 'one_lane':
                r.set_lanes(1)
            r.stage_timing(True)
            f0, f1 = frames_of(args.warmup + args.steps - 1)
            r.render(P, f0, f1)
            ms, iters = r.stage_times()
            r.stage_timing(False)
            stage[tag] = (ms, max(iters, 1))
        r.set_lanes(args.lanes or 1)
    # untimed: count rays and tests of exactly the timed frames with the instrumented kernels, twice: the
    # reference-order binary walk (its box and triangle tests are the reference's own) and the production walk
    # over the wide nodes (the node visits and triangle tests the timed kernel really made)
    c = {}
    for mode in (1, 2):
        r.enable_counters(mode)
        r.reset_counters()
        for k in range(args.steps):
            f0, f1 = frames_of(args.warmup + k)
            r.render(P, f0, f1)
        got = r.counters()
        if mode == 1:
            c.update(got)
        else:
            c['wide_nodes'], c['wide_tris'] = got['wide_nodes'], got['wide_tris']
    r.enable_counters(False)
    keys = ['rays', 'samples', 'box_tests', 'tri_tests', 'shade_vertices', 'env_samples', 'tex_taps', 'tex_taps_r', 'wide_nodes', 'wide_tris', 'path_rays']
    local = torch.tensor([float(c[k]) for k in keys], dtype=torch.float64, device='cuda')
    total = local.clone()
    if world > 1:
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
    tot = {k: float(v) for k, v in zip(keys, total.tolist())}
    loc = {k: float(v) for k, v in zip(keys, local.tolist())}

    result = None
    if rank == 0:
        mrays = tot['rays'] / elapsed / 1e6
        node_b, tri_b = r.scene_bytes()
        pm = {}
        pmc_path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.exists(pmc_path):
            try:
                with open(pmc_path) as f:
                    pm = json.load(f).get(args.workload, {})
                if pm.get('workload') != [W, H, spp, args.bounces]:
                    pm = {}
            except Exception:
                pm = {}
        pmk = pm.get('kernels', {})
        kernels = []
        if args.mode == 'wavefront':
            # dominant kernel: extend (closest hit for the pending ray of every record of both queues), one launch per iteration
            kname = 'wf_extend_kernel'
            ms_timed, launches = stage['timed_config']
            ms_one, launches_one = stage['one_lane']
            kms = ms_timed['extend'] / launches
            req = requested_bytes(loc, not args.fp32_textures)
            # bytes the timed kernel's own walk asks the memory system for: per ray 32 in + 16 out (the queue
            # arrays), 64 per wide node visited, 48 per triangle tested; over the counted steps / the launches of one step
            own_per_launch = req['extend'] / args.steps / launches
            # the layout-independent accounting on the reference's binary walk
            ref_per_launch = (40 * loc['rays'] + 32 * loc['box_tests'] + 48 * loc['tri_tests']) / args.steps / launches
            alg_per_launch = own_per_launch if loc['wide_nodes'] > 0 else ref_per_launch
            for name, slot in [(n, n) for n in STAGES if n in KERNEL_NAMES]:
                e = pmk.get(KERNEL_NAMES[name], {})
                k_ms = ms_one[slot] / launches_one
                req_launch = req[name] / args.steps / launches_one
                kernels.append({
                    'kernel': KERNEL_NAMES[name],
                    'ms_per_step': round(ms_one[slot], 3), 'ms_per_step_timed_config': round(ms_timed[slot], 3),
                    'launches_per_step': int(launches_one), 'ms_per_launch': round(k_ms, 4),
                    'requested_bytes_per_launch': int(req_launch),
                    'requested_GBs': round(req_launch / (k_ms * 1e-3) / 1e9, 1) if k_ms > 0 else None,
                    'pmc_bytes_per_launch': e.get('pmc_bytes_per_launch'),
                    'pmc_frac_of_hbm_peak': e.get('pmc_frac_of_hbm_peak'),
                    'valu_busy': e.get('valu_busy'), 'valu_lanes_per_instruction': e.get('valu_lanes_per_instruction'),
                    'valu_useful': (round(e['valu_busy'] * e['valu_lanes_per_instruction'], 3) if e.get('valu_busy') is not None else None),
                    'wave_wait_share': e.get('wave_wait_share'), 'waves_per_simd': e.get('waves_per_simd'),
                })
        else:
            kname = 'render_samples_kernel'
            launches = args.steps
            kms = float(np.mean(kernel_ms))
            alg_per_launch = ref_per_launch = own_per_launch = algorithmic_bytes(loc) / args.steps
        requested_gbs = alg_per_launch / (kms * 1e-3) / 1e9
        ext = pm.get('extend_timed_config') or {}
        traffic = ext.get('pmc_bytes_per_launch')
        traffic_frac = round(traffic / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic else None
        # What bounds the kernel, from the counters: memory when what reaches HBM is the larger share of its ceiling, else
        # vector instruction issue (the scene is cache resident then and the requested bytes are a gather rate, not HBM use).
        # achieved / peak / frac are stated against that bound: bytes that reached the memory side per second against
        # the HBM peak, or vector instructions issued per second (the instruction count of a launch from the committed
        # counter pass over the same workload, the launch time measured here) against one instruction per SIMD every
        # four cycles.
        valu = ext.get('valu_busy')
        insts = ext.get('valu_instructions_per_launch')
        if traffic_frac is None or valu is None or not insts:
            bound = 'unmeasured (no counters committed for this workload and size; achieved is requested bytes over time)'
            achieved, peak, unit, frac_of = requested_gbs, HBM_PEAK_GBS, 'GB/s', 'bytes the kernel requests / launch time / HBM peak: a gather rate, served mostly by the caches when the scene fits them'
        elif traffic_frac >= valu:
            bound = 'hbm'
            achieved, peak, unit, frac_of = traffic / (kms * 1e-3) / 1e9, HBM_PEAK_GBS, 'GB/s', 'PMC bytes that reached the memory side (FETCH_SIZE + WRITE_SIZE, profiles/pmc_traffic.json) / launch time / HBM peak'
        else:
            # (neither resource near its ceiling: the waves wait on dependent memory reads; the larger share is reported)
            bound = 'valu_issue' if valu >= 0.7 else 'latency (neither vector issue nor HBM bandwidth is saturated; achieved is vector issue, the larger share)'
            achieved, peak, unit = insts / (kms * 1e-3) / 1e9, VALU_PEAK_GINST, 'Ginst/s'
            frac_of = ('vector instructions of a launch (SQ_INSTS_VALU of the committed counter pass) / launch time measured here / (%d SIMDs x %.1f GHz / 4 cycles per wave64 instruction); '
                       'the same pass puts the vector ALUs at valu_busy of their cycles with valu_lanes_per_instruction of the lanes active' % (N_SIMDS, CLOCK_GHZ))
        ref_frac = ref_per_launch / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS
        result = {
            'metric': 'Mrays/s', 'value': round(mrays, 3), 'unit': 'Mrays/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'strong' if strong else 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'samples_per_s': round(tot['samples'] / elapsed, 1), 'image_mean': image_mean,
            'rays_per_sample': round(tot['rays'] / tot['samples'], 3),
            'config': {'workload': '%s %dx%d, %s, bounce cap %d'
                                   % (workload_name, W, H, ('%d spp per step over all GPUs' % args.total_spp) if strong else ('%d spp per GPU per step' % spp), args.bounces),
                       'partition': ('single GPU' if world == 1 else
                                     ('samples: rank r renders frames [k*S + r*S/N, k*S + (r+1)*S/N) of step k, RCCL reduce to rank 0' if strong else
                                      'samples: rank r renders frames [(k*N+r)*spp, +spp), RCCL reduce to rank 0')),
                       'scene_bytes': {'nodes': node_b, 'triangles': tri_b}},
            'roofline': {'bound': bound, 'kernel': kname, 'launches': int(launches), 'achieved': round(achieved, 2), 'peak': peak,
                         'unit': unit, 'frac': round(achieved / peak, 5), 'frac_of': frac_of,
                         'traffic': traffic, 'hbm_traffic_frac': traffic_frac,
                         'hbm_traffic_frac_of': 'PMC bytes that reached the memory side (FETCH_SIZE + WRITE_SIZE, profiles/pmc_traffic.json) / launch time / HBM peak',
                         'requested': {'GBs': round(requested_gbs, 2), 'over_hbm_peak': round(requested_gbs / HBM_PEAK_GBS, 5),
                                       'note': 'bytes the kernel asks the memory system for (48 per ray + 64 per wide node visited + 48 per triangle tested) / launch time: a gather rate, served mostly by L2 and Infinity Cache when the scene fits them, so it can exceed the HBM peak'},
                         'kernel_ms': round(kms, 4),
                         'algorithmic_bytes_per_launch': int(alg_per_launch),
                         'own_walk_per_ray': {'wide_nodes': round(loc['wide_nodes'] / max(loc['rays'], 1), 3), 'tri_tests': round(loc['wide_tris'] / max(loc['rays'], 1), 3),
                                              'bytes': round(own_per_launch * launches * args.steps / max(loc['rays'], 1), 1)},
                         'reference_walk': {'box_tests_per_ray': round(loc['box_tests'] / max(loc['rays'], 1), 3), 'tri_tests_per_ray': round(loc['tri_tests'] / max(loc['rays'], 1), 3),
                                            'bytes_per_launch': int(ref_per_launch), 'frac': round(ref_frac, 5),
                                            'valid': ref_frac <= 1.0,
                                            'note': 'the layout-independent accounting (40 + 32 per box test + 48 per triangle test of the reference\'s binary walk); above 1 it only says that the timed kernel does not do that work: its 4-wide walk visits a fifth as many nodes'},
                         'issue': ({'valu_busy': valu, 'valu_lanes_per_instruction': ext.get('valu_lanes_per_instruction'), 'wave_wait_share': ext.get('wave_wait_share')} if valu is not None else None),
                         'kernels': kernels,
                         'kernels_note': 'ms_per_step: HIP events around every launch of one untimed step on one lane (the times add up to the step); ms_per_step_timed_config: the same in the timed configuration (it differs only under --lanes 2, where the kernels of two lanes share the GPU); requested bytes from the record layouts and the tallies of the instrumented run; pmc_*, valu_*, wave_wait_share, waves_per_simd from the committed rocprofv3 counter passes (profiles/pmc_traffic.json), null when none matches this workload and size',
                         'whole_path_algorithmic_GBs': round(algorithmic_bytes(loc) / (elapsed * 1e9), 2)},
        }

    # CPU baseline: the oracle on this host's cores, rank 0, single-GPU runs only, bounded sample
    if rank == 0 and world == 1 and not args.no_cpu:
        import orc
        if bundle is None:
            # the oracle walks the reference's tree: built on the host here, outside the timed region
            tris, mats, env4 = soup_arrays
            t_build = time.perf_counter()
            bundle = util.Bundle(tris, mats, (), env4)
            build_s = time.perf_counter() - t_build
        else:
            build_s = None
        S = bundle.oracle(P)
        cpu_spp = args.cpu_spp or (1 if args.workload == 'soup' else 12)
        t1 = time.perf_counter()
        _, st = orc.render(S, P, 0, cpu_spp, args.bounces)
        dt = time.perf_counter() - t1
        result['cpu_baseline'] = {'value': round(st['rays'] / dt / 1e6, 3), 'unit': 'Mrays/s', 'cores': orc.max_threads(), 'kind': 'port',
                                  'samples_per_s': round(st['samples'] / dt, 1),
                                  'sample': 'same scene and camera, %dx%d, frames [0,%d), %.1f s%s' % (
                                      W, H, cpu_spp, dt, '' if build_s is None else ' (reference tree built on the host in %.1f s, not timed)' % build_s)}
    if rank == 0:
        print(json.dumps(result))
    r.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
