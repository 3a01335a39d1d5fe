#!/usr/bin/env python3
"""Builds profiles/pmc_traffic.json (read by bench.py for the counter-derived fields of its roofline) from the per-kernel
summaries tools/profile_round.sh leaves: DIR/{sq,fetch,write,stats1}.txt for the atrium at the bench's default size and
DIR/soup_{sq,fetch,write,stats1}.txt for the 10-million-triangle soup at 16 samples a pixel, DIR/sunsky_... for the sun/sky
configuration.

usage: profiles/make_pmc_json.py DIR TAG     (TAG names the committed copies, profiles/TAG_*.txt)
"""
import json
import os
import re
import sys

HBM_PEAK = 8.0e12
KERNELS = ['wf_extend_kernel', 'wf_pathhit_kernel', 'wf_miss_kernel', 'wf_shadow_kernel', 'wf_bsdf_kernel']


def timed_variant(line, kernel):
    """The uncounted instantiation of a stage kernel (template argument COUNT = false), which is what a timed render runs."""
    return ('ptamd::%s<false' % kernel) in line


def counters(path, kernel):
    out = {}
    with open(path) as f:
        for line in f:
            if not timed_variant(line, kernel):
                continue
            parts = line.split()
            i = parts.index('dispatches')
            out[parts[i - 1]] = (int(parts[i + 1]), float(parts[i + 3]))
    return out


def avg_ns(path, kernel):
    with open(path) as f:
        for line in f:
            if timed_variant(line, kernel):
                m = re.search(r'\s(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+[\d.]+%\s*$', line)
                if m:
                    return float(m.group(3))
    return None


def kernel_table(d, prefix, suffix, names):
    """suffix '': the passes in the bench's default configuration; '1': the passes with one lane forced."""
    kernels = {}
    for k in names:
        sq = counters(os.path.join(d, prefix + 'sq' + suffix + '.txt'), k)
        fetch = counters(os.path.join(d, prefix + 'fetch' + suffix + '.txt'), k).get('FETCH_SIZE')
        write = counters(os.path.join(d, prefix + 'write' + suffix + '.txt'), k).get('WRITE_SIZE')
        if not sq or not fetch or not write:
            continue
        fetch_kib, write_kib = fetch[1] / fetch[0], write[1] / write[0]
        inst = sq['SQ_ACTIVE_INST_VALU'][1]
        busy = inst * 4 / 1024 / (sq['GRBM_GUI_ACTIVE'][1] / 8)
        lanes = sq['SQ_THREAD_CYCLES_VALU'][1] / (inst * 64) if inst else 0.0
        ns = avg_ns(os.path.join(d, prefix + ('stats1.txt' if suffix else 'stats.txt')), k)
        pmc_bytes = (fetch_kib + write_kib) * 1024
        kernels[k] = {
            'fetch_size_kib_per_launch': round(fetch_kib, 1),
            'write_size_kib_per_launch': round(write_kib, 1),
            'pmc_bytes_per_launch': int(pmc_bytes),
            'pmc_bytes_upper_bound': int((2 * fetch_kib + write_kib) * 1024),
            'avg_launch_ns': ns,
            'pmc_frac_of_hbm_peak': round(pmc_bytes / (ns * 1e-9) / HBM_PEAK, 4) if ns else None,
            'pmc_upper_frac_of_hbm_peak': round((2 * fetch_kib + write_kib) * 1024 / (ns * 1e-9) / HBM_PEAK, 4) if ns else None,
            'valu_busy': round(busy, 3),
            'valu_lanes_per_instruction': round(lanes, 3),
            'valu_instructions_per_launch': int(sq['SQ_INSTS_VALU'][1] / sq['SQ_INSTS_VALU'][0]),
            'wave_wait_share': round(sq['SQ_WAIT_INST_ANY'][1] / sq['SQ_WAVE_CYCLES'][1], 3),
            'waves_per_simd': round(sq['SQ_WAVE_CYCLES'][1] / (sq['GRBM_GUI_ACTIVE'][1] / 8 * 1024 / 4), 2),
        }
    return kernels


def entry(d, prefix, workload, tag, scene_note):
    return {
        'workload': workload,
        'kernels': kernel_table(d, prefix, '1', KERNELS),
        'extend_timed_config': kernel_table(d, prefix, '', ['wf_extend_kernel']).get('wf_extend_kernel'),
        'note': 'kernels: one lane forced, where a launch covers the whole pool and runs alone; extend_timed_config: the traversal kernel in the bench\'s default configuration (one '
                'lane as well, since that became the default; its mean launch time is that of the default kernel trace, %s_%sstats.txt). '
                'Per launch of the uncounted kernels. pmc_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024 from separate --pmc passes (counter collection '
                'serialises concurrent kernels); gfx950 tallies a 128-byte read request at 64 bytes, which halves coalesced full-line reads but not random reads of 64-byte '
                'records (profiles/r2_fetch_size_calibration.txt), so the truth lies between pmc_bytes and pmc_bytes_upper_bound (2 * FETCH_SIZE + WRITE_SIZE): near the lower '
                'figure for the traversal kernel (random node reads), near the upper one for the stage kernels (streams over the queue arrays). Fractions of the HBM peak use the '
                'mean launch time of the one-lane kernel trace (%s_%sstats1.txt). valu_busy = SQ_ACTIVE_INST_VALU*4/1024/(GRBM_GUI_ACTIVE/8) (share of the 1024 SIMD issue slots '
                'of the chip that issue a vector instruction), lanes = SQ_THREAD_CYCLES_VALU/(SQ_ACTIVE_INST_VALU*64), wave_wait_share = SQ_WAIT_INST_ANY/SQ_WAVE_CYCLES, '
                'waves_per_simd = SQ_WAVE_CYCLES/(GRBM_GUI_ACTIVE/8*256). Sources profiles/%s_%s{sq,fetch,write,stats}{,1}.txt. %s' % (tag, prefix, tag, prefix, tag, prefix, scene_note),
    }


def main():
    d, tag = sys.argv[1], sys.argv[2]
    out = {
        'atrium': entry(d, '', [1920, 1080, 64, 512], tag,
                        'The scene (about 30 MB) is served from L2 / Infinity Cache, so the traversal kernel\'s memory-side traffic is far below its requested byte count.'),
        'soup': entry(d, 'soup_', [1920, 1080, 16, 512], tag,
                      'The 0.8 GB of wide nodes and triangle records fit no cache, but the upper levels of the tree do: a little over half of the bytes asked for reach the memory side.'),
        'sunsky': entry(d, 'sunsky_', [1920, 1080, 64, 16], tag,
                        'Sun/sky mode draws no light samples: there is no shadow stage, and the bsdf stage runs inside the path-hit kernel.'),
    }
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'pmc_traffic.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
