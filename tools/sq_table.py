"""Per-kernel issue figures from a pmc_summary.py listing of the SQ counter pass:
   VALU share = SQ_ACTIVE_INST_VALU * 4 / (SIMDs * GRBM_GUI_ACTIVE), lanes per instruction = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64),
   wait share = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, waves per SIMD = SQ_WAVE_CYCLES / (SQ_BUSY_CYCLES-derived busy time).
usage: python tools/sq_table.py gpurun_out/TAG/sq.txt"""
import re, sys, collections
rows = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    m = re.match(r'(.{60}) (\S+)\s+dispatches\s+(\d+)\s+sum\s+(\S+)', line)
    if not m:
        continue
    rows[m.group(1).strip()][m.group(2)] = (int(m.group(3)), float(m.group(4)))
print('%-62s %6s %9s %9s %7s %7s %7s %9s' % ('kernel', 'calls', 'Minstr', 'busy Mclk', 'VALU', 'lanes', 'wait', 'waves/SIMD'))
for k, r in sorted(rows.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', (0, 0))[1]):
    if 'SQ_INSTS_VALU' not in r or not k.startswith(('void ptamd::wf', 'ptamd::wf')):
        continue
    n = r['SQ_INSTS_VALU'][0]
    gui = r['GRBM_GUI_ACTIVE'][1]
    act = r['SQ_ACTIVE_INST_VALU'][1]
    thr = r['SQ_THREAD_CYCLES_VALU'][1]
    wc = r['SQ_WAVE_CYCLES'][1]
    wait = r['SQ_WAIT_INST_ANY'][1]
    # the SQ counters of this pass are sums over the shader engines sampled; ACTIVE_INST_VALU counts 4-cycle issue slots per SIMD
    valu = act * 4 / (1024.0 * gui) if gui else 0
    print('%-62s %6d %9.1f %9.2f %7.3f %7.3f %7.3f %9.2f' % (k[:62], n, r['SQ_INSTS_VALU'][1] / 1e6, gui / 1e6, valu, thr / (act * 64) if act else 0, wait / wc if wc else 0, wc / (gui * 1024 / 4.0) if gui else 0))
