"""Timeline of the last N kernels of a rocprofv3 kernel-trace database: start offset, duration, name."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rows = db.execute("select name,start,end from kernels order by start").fetchall()[-n:]
t0 = rows[0][1]
for name, s, e in rows:
    short = name.split('(')[0].replace('void ptamd::', '').replace('ptamd::', '')
    print('%9.3f ms  %8.3f ms  %s' % ((s - t0) / 1e6, (e - s) / 1e6, short))
