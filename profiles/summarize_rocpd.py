#!/usr/bin/env python3
"""Prints the per-kernel statistics of a rocprofv3 results database (the `--kernel-trace --stats` view)."""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print('%-72s %8s %14s %14s %14s %14s %7s' % ('kernel', 'calls', 'total_ns', 'avg_ns', 'min_ns', 'max_ns', 'pct'))
    for name, n, tot, avg, mn, mx in rows:
        print('%-72s %8d %14d %14.0f %14d %14d %6.2f%%' % (name[:72], n, tot, avg, mn, mx, 100.0 * tot / total))


if __name__ == '__main__':
    main(sys.argv[1])
