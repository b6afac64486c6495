#!/usr/bin/env python3
"""GPU busy fraction of a cohort run out of a rocprofv3 kernel trace of tools/emt_ensemble.py <members> c<width>: the timed
pass is the LAST run of the process, so the last N launches (N = launches issued in that pass, printed by the tool) are its
kernels.  busy = sum of kernel durations / (last end - first start).   usage: cohort_busy.py <results.db> <N>"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2])
rows = db.execute('select name, start, end from kernels order by start').fetchall()[-n:]
span = rows[-1][2] - rows[0][1]
busy = sum(e - s for _, s, e in rows)
gaps = sorted(((rows[i + 1][1] - rows[i][2]) for i in range(len(rows) - 1)), reverse=True)
big = [g for g in gaps if g > 10000]
print('%d launches, span %.2f ms, kernels %.2f ms, gpu_busy %.3f; %d gaps > 10 us (%.2f ms in them), mean kernel %.2f us'
      % (len(rows), span / 1e6, busy / 1e6, busy / span, len(big), sum(big) / 1e6, busy / len(rows) / 1e3))
by = {}
for nm, s, e in rows:
    k = re.sub(r'^void ', '', nm).replace('sella::', '')
    k = re.sub(r'batched_kernel<&(?:\(anonymous namespace\)::)?([A-Za-z0-9_]+).*', r'batched<\1>', k)[:48]
    c = by.setdefault(k, [0, 0])
    c[0] += 1
    c[1] += e - s
for k, (cnt, tot) in sorted(by.items(), key=lambda kv: -kv[1][1])[:14]:
    print('  %-48s %5d launches %8.1f us total %6.2f us mean' % (k, cnt, tot / 1e3, tot / cnt / 1e3))
