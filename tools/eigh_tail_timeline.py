#!/usr/bin/env python3
"""Launch-by-launch timeline of what follows the tridiagonalisation inside the LAST eigh of a rocprofv3 kernel trace
(divide & conquer, compact-WY factors, back-transformation, outputs): start offset, duration and the idle gap in front of
every launch.  Usage: tools/eigh_tail_timeline.py <results.db>"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'^void ', '', name).replace('sella::', '').replace('(anonymous namespace)::', '')
    return re.sub(r'\(.*$', '', name)


cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute('select name, start, end from kernels order by start').fetchall()
last = max(i for i, r in enumerate(rows) if 'trd_tail_lds_kernel' in r[0] or 'tridiag_tail_kernel' in r[0])
t0 = rows[last][1]
prev_end = rows[last][2]
busy = gaps = 0
print('offset_us  gap_us  dur_us  kernel')
for name, st, en in rows[last:]:
    gap = (st - prev_end) / 1e3
    print(f'{(st - t0) / 1e3:9.1f} {gap:7.1f} {(en - st) / 1e3:7.1f}  {short(name)}')
    busy += en - st
    gaps += max(st - prev_end, 0)
    prev_end = max(prev_end, en)
print(f'span {(prev_end - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, idle {gaps / 1e3:.1f} us, {len(rows) - last} launches')
