#!/usr/bin/env python3
"""Launch-by-launch timeline of ONE block-Davidson iteration out of a rocprofv3 kernel trace of tools/block_iter.py
(iterations are split at the panel product that applies the operator to the new block).  Usage: <trace dir>"""
import glob
import re
import sqlite3
import sys

paths = glob.glob(sys.argv[1] + '/*.db') + glob.glob(sys.argv[1] + '/*/*.db')
rows = sqlite3.connect(paths[0]).execute('select name, start, end from kernels order by start').fetchall()


def short(n):
    n = re.sub(r'^void ', '', n).replace('sella::', '').replace('(anonymous namespace)::', '')
    return re.sub(r'\(.*$', '', n)[:40]


its, cur = [], []
for r in rows:
    if 'panel16_mfma_kernel' in r[0] and cur:
        its.append(cur)
        cur = []
    cur.append(r)
its.append(cur)
# one operator pass per iteration: take an iteration from the last run (skip the bare panel passes of the warm-up)
cands = [it for it in its if len(it) > 6]
mid = cands[len(cands) * 3 // 4]
t0 = mid[0][1]
busy = sum(e - s for _, s, e in mid)
print('iteration with %d launches, span %.1f us, busy %.1f us' % (len(mid), (mid[-1][2] - t0) / 1e3, busy / 1e3))
prev = t0
for n, s, e in mid:
    print('  +%7.1f us gap %6.1f  %-40s %7.2f us' % ((s - t0) / 1e3, (s - prev) / 1e3, short(n), (e - s) / 1e3))
    prev = e
spans = sorted((it[-1][2] - it[0][1]) / 1e3 for it in cands)
print('iterations', len(cands), 'median span us %.1f' % spans[len(spans) // 2])
