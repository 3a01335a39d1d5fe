#!/usr/bin/env python3
"""Builds profiles/pmc_traffic.json (read by bench.py for roofline.traffic and roofline.issue) from the per-kernel
counter summaries tools/profile_round.sh leaves: DIR/{sq,fetch,write}.txt for the atrium at the bench's default size and
DIR/soup_{sq,fetch,write}.txt for the 10-million-triangle soup at 16 samples a pixel.

usage: profiles/make_pmc_json.py DIR TAG     (TAG names the committed copies, profiles/TAG_*.txt)
"""
import json
import os
import sys

KERNEL = 'wf_extend_kernel<false, true>'


def counters(path):
    out = {}
    with open(path) as f:
        for line in f:
            if KERNEL not in line:
                continue
            parts = line.split()
            i = parts.index('dispatches')
            out[parts[i - 1]] = (int(parts[i + 1]), float(parts[i + 3]))
    return out


def rays_per_launch(d, prefix):
    """Rays one launch of the traversal kernel traces, from the bench line of the same run."""
    with open(os.path.join(d, prefix + 'bench.json')) as f:
        r = json.loads(f.read().strip().split('\n')[-1])['roofline']
    return r['algorithmic_bytes_per_launch'] / r['own_walk_per_ray']['bytes']


def entry(d, prefix, workload, tag, scene_note):
    sq = counters(os.path.join(d, prefix + 'sq.txt'))
    fetch = counters(os.path.join(d, prefix + 'fetch.txt'))['FETCH_SIZE']
    write = counters(os.path.join(d, prefix + 'write.txt'))['WRITE_SIZE']
    n = fetch[0]
    fetch_kib, write_kib = fetch[1] / n, write[1] / write[0]
    rays = rays_per_launch(d, prefix)
    inst = sq['SQ_ACTIVE_INST_VALU'][1]
    busy = inst * 4 / 1024 / (sq['GRBM_GUI_ACTIVE'][1] / 8)
    lanes = sq['SQ_THREAD_CYCLES_VALU'][1] / (inst * 64)
    return {
        'workload': workload,
        'kernel': KERNEL,
        'fetch_size_kib_per_launch': round(fetch_kib, 1),
        'write_size_kib_per_launch': round(write_kib, 1),
        # FETCH_SIZE counts a 128-byte request as 64 bytes: the coalesced reads of the ray arrays (32 B per ray) are counted
        # at half and corrected here; the random 64-byte node reads and the triangle records are counted as they are
        # (calibration: profiles/r2_fetch_size_calibration.txt)
        'hbm_bytes_per_launch': int((fetch_kib + write_kib) * 1024 + 16 * rays),
        'fetch_doubled_upper_bound': int((2 * fetch_kib + write_kib) * 1024),
        'note': 'FETCH_SIZE + 16 B per ray + WRITE_SIZE: gfx950 tallies 128-B read requests at 64 B, which halves the coalesced reads of the ray arrays '
                '(corrected) but not the random 64-byte node reads (profiles/r2_fetch_size_calibration.txt: 60 B counted per 64-B record); doubling everything '
                'gives the upper bound beside it. Separate --pmc passes (counter collection serialises the two lanes); source profiles/%s_%sfetch.txt, '
                '%s_%swrite.txt. %s' % (tag, prefix, tag, prefix, scene_note),
        'issue': {
            'valu_busy': round(busy, 3),
            'valu_lanes_per_instruction': round(lanes, 3),
            'valu_instructions_per_launch': int(sq['SQ_INSTS_VALU'][1] / sq['SQ_INSTS_VALU'][0]),
            'wave_wait_share': round(sq['SQ_WAIT_INST_ANY'][1] / sq['SQ_WAVE_CYCLES'][1], 3),
            'note': '%s from profiles/%s_%ssq.txt: valu_busy = SQ_ACTIVE_INST_VALU*4/1024/(GRBM_GUI_ACTIVE/8) (share of the 1024 SIMD issue '
                    'slots of the chip that issue a vector instruction), lanes = SQ_THREAD_CYCLES_VALU/(SQ_ACTIVE_INST_VALU*64), '
                    'wave_wait_share = SQ_WAIT_INST_ANY/SQ_WAVE_CYCLES' % (KERNEL, tag, prefix),
        },
    }


def main():
    d, tag = sys.argv[1], sys.argv[2]
    out = {
        'atrium': entry(d, '', [1920, 1080, 64, 512], tag,
                        'The scene (about 30 MB) is served from L2 / Infinity Cache, so memory-side traffic is far below the algorithmic byte count.'),
        'soup': entry(d, 'soup_', [1920, 1080, 16, 512], tag,
                      'The 0.8 GB of wide nodes and triangle records fit no cache, but the upper levels of the tree do: a little over half of the bytes asked for reach the memory side.'),
    }
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'pmc_traffic.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
