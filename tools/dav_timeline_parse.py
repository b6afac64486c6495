import glob, sqlite3, sys
paths = glob.glob(sys.argv[1] + '/*.db') + glob.glob(sys.argv[1] + '/*/*.db')
db = sqlite3.connect(paths[0])
rows = db.execute("select name,start,end,duration from kernels order by start").fetchall()
def short(n):
    return n.replace('sella::', '').replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:34]
# last davidson call: everything after the last launch of the eigensolver (its final transpose)
last = max(i for i, r in enumerate(rows) if 'wy_apply' in r[0])
rows = rows[last + 4:]
third = len(rows) * 2 // 3
rows = rows[third:]
# split into iterations at the residual kernel (first launch of the fused chain of an iteration)
its, cur = [], []
for r in rows:
    if 'dav_resid_kernel' in r[0] and cur:
        its.append(cur); cur = []
    cur.append(r)
its.append(cur)
# the iteration printed is the one with the MEDIAN span (what the rate of the loop corresponds to), not whichever sits in
# the middle of the list
inner = its[2:-1] if len(its) > 4 else its
mid = sorted(inner, key=lambda it: it[-1][2] - it[0][1])[len(inner) // 2]
t0 = mid[0][1]
print('iteration with %d launches, span %.1f us, busy %.1f us' % (len(mid), (mid[-1][2] - t0) / 1e3, sum(r[3] for r in mid) / 1e3))
prev_end = t0
for n_, s, e, d in mid:
    print('  +%7.1f us gap %6.1f  %-34s %6.2f us' % ((s - t0) / 1e3, (s - prev_end) / 1e3, short(n_), d / 1e3))
    prev_end = e
spans = [(it[-1][2] - it[0][1]) / 1e3 for it in its[2:-1]]
print('iterations', len(its), 'median span us', sorted(spans)[len(spans) // 2] if spans else None)
