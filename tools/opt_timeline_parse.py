"""Kernel timeline of ONE optimizer step from a rocprofv3 --kernel-trace database of tools/opt_profile.py (or of the
EMT slab script): steps are delimited by a kernel that runs once per quasi-Newton update — `lr_pre_kernel` of the
one-call step (default), or the name given as third argument (`sym_rank2k_kernel` for the general path)."""
import glob
import sqlite3
import sys

paths = glob.glob(sys.argv[1] + '/*.db') + glob.glob(sys.argv[1] + '/*/*.db')
db = sqlite3.connect(paths[0])
rows = db.execute("select name,start,end,duration from kernels order by start").fetchall()
which = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
mark = sys.argv[3] if len(sys.argv) > 3 else 'lr_pre_kernel'
nseg = int(sys.argv[4]) if len(sys.argv) > 4 else 1          # consecutive segments printed (2 for a step with a view)


def short(n):
    return n.replace('sella::', '').replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:40]


steps, cur = [], []
for r in rows:
    cur.append(r)
    if mark in r[0]:
        steps.append(cur)
        cur = []
k0 = int(len(steps) * which)
mid = [r for st in steps[k0:k0 + nseg] for r in st]
t0 = mid[0][1]
print('step with %d launches, span %.1f us, busy %.1f us' % (len(mid), (mid[-1][2] - t0) / 1e3, sum(r[3] for r in mid) / 1e3))
prev_end = t0
for n_, s, e, d in mid:
    print('  +%7.1f us gap %6.1f  %-40s %6.2f us' % ((s - t0) / 1e3, (s - prev_end) / 1e3, short(n_), d / 1e3))
    prev_end = e
spans = sorted((st[-1][2] - st[0][1]) / 1e3 for st in steps[3:])
print('steps', len(steps), 'median span us (update to update)', spans[len(spans) // 2] if spans else None)
