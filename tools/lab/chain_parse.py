import collections, glob, sqlite3, sys
paths = glob.glob(sys.argv[1] + '/*.db') + glob.glob(sys.argv[1] + '/*/*.db')
db = sqlite3.connect(paths[0])
rows = db.execute("select name,start,duration from kernels order by start").fetchall()
agg = collections.defaultdict(list)
prev = None
for n, s, d in rows:
    if 'stream_read' in n:
        prev = d
    elif 'tiny' in n:
        agg[-1 if prev is None else round(prev / 1000)].append(d)
        prev = None if prev is None else prev
for k in sorted(agg)[:16]:
    print('prev stream ~%d us: tiny avg %.2f us over %d' % (k, sum(agg[k]) / len(agg[k]) / 1e3, len(agg[k])))
