"""From a rocprofv3 kernel-trace database: the share of the wall time between the first and the last stage kernel during
which no kernel was running on the GPU at all (host-side gaps of the stage loop), and the longest such gaps.
usage: stream_gaps.py results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end from kernels order by start").fetchall()
rows = [r for r in rows if 'wf_' in r[0]]
t0, t1 = rows[0][1], max(r[2] for r in rows)
gaps = []
busy_end = rows[0][2]
for name, s, e in rows[1:]:
    if s > busy_end:
        gaps.append((s - busy_end, busy_end - t0, name))
    busy_end = max(busy_end, e)
idle = sum(g[0] for g in gaps)
print('stage kernels %d, span %.1f ms, no kernel running for %.2f ms (%.2f %%) in %d gaps' % (len(rows), (t1 - t0) / 1e6, idle / 1e6, 100.0 * idle / (t1 - t0), len(gaps)))
for g, at, name in sorted(gaps, reverse=True)[:8]:
    print('  gap %.3f ms at %.1f ms before %s' % (g / 1e6, at / 1e6, name.split('(')[0][:60]))
