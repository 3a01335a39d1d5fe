"""Kernel timeline of one 1-spp 1080p render from a rocprofv3 kernel-trace database (tools/one_frame.py TRACE=1 under
rocprofv3 --kernel-trace): per stage iteration the duration of every stage kernel, and the time no kernel was running."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if 'kernel_dispatch' in t and 'rocpd' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t or t.startswith('rocpd_info_kernel_symbol')][0]
rows = db.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)).fetchall()
# the last generate kernel starts the last render
gi = max(i for i, r in enumerate(rows) if 'wf_generate_kernel' in r[0])
rows = rows[gi:]
t0 = rows[0][1]
def short(n):
    m = re.search(r'(wf_[a-z_]+?_kernel|[a-z_]+_kernel|rocclr_\w+)', n)
    return m.group(1) if m else n
busy = sum(e - s for _, s, e in rows)
span = rows[-1][2] - t0
print('kernels %d  span %.3f ms  kernel time %.3f ms  idle %.3f ms' % (len(rows), span / 1e6, busy / 1e6, (span - busy) / 1e6))
it = -1
line = ''
for n, s, e in rows:
    k = short(n)
    if k == 'wf_plan_kernel':
        if line: print(line)
        it += 1; line = 'it %2d @%.2f ms:' % (it, (s - t0) / 1e6)
    line += ' %s %.0f' % (k.replace('wf_', '').replace('_kernel', ''), (e - s) / 1e3)
print(line)
