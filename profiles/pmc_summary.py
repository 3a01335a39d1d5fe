#!/usr/bin/env python3
"""Sums the hardware counters of a rocprofv3 --pmc results database per kernel name."""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [c[1] for c in cur.execute('pragma table_info(counters_collection)')]
    name_col = 'kernel_name' if 'kernel_name' in cols else 'name'
    rows = cur.execute('select %s, counter_name, count(*), sum(value) from counters_collection group by %s, counter_name' % (name_col, name_col)).fetchall()
    for k, c, n, v in rows:
        print('%-60s %-36s dispatches %6d  sum %.6g  per-dispatch %.6g' % (k[:60], c, n, v, v / n))


if __name__ == '__main__':
    main(sys.argv[1])
