"""Kernel timeline of ONE optimizer step from a rocprofv3 --kernel-trace database of tools/opt_profile.py (or of the
EMT slab script): steps are delimited by the fused rank-2k update of B (`sym_rank2k_kernel`, once per quasi-Newton
update)."""
import glob
import sqlite3
import sys

paths = glob.glob(sys.argv[1] + '/*.db') + glob.glob(sys.argv[1] + '/*/*.db')
db = sqlite3.connect(paths[0])
rows = db.execute("select name,start,end,duration from kernels order by start").fetchall()
which = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5


def short(n):
    return n.replace('sella::', '').replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:40]


steps, cur = [], []
for r in rows:
    cur.append(r)
    if 'sym_rank2k_kernel' in r[0]:
        steps.append(cur)
        cur = []
mid = steps[int(len(steps) * which)]
t0 = mid[0][1]
print('step with %d launches, span %.1f us, busy %.1f us' % (len(mid), (mid[-1][2] - t0) / 1e3, sum(r[3] for r in mid) / 1e3))
prev_end = t0
for n_, s, e, d in mid:
    print('  +%7.1f us gap %6.1f  %-40s %6.2f us' % ((s - t0) / 1e3, (s - prev_end) / 1e3, short(n_), d / 1e3))
    prev_end = e
spans = sorted((st[-1][2] - st[0][1]) / 1e3 for st in steps[3:])
print('steps', len(steps), 'median span us (update to update)', spans[len(spans) // 2] if spans else None)
