for f in 0 2816 2560 2304 2048; do echo "== 3072 fold_min=$f"; EIGH_FOLD_MIN=$f timeout 300 python tools/eigh_only.py 3072 5 2>&1 | tail -2; done
mkdir -p gpurun_out/symv; cd /tmp && export TMPDIR=/tmp
EIGH_FOLD_MIN=1536 timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/symv/prof3072 -o p -- python /root/repo/tools/eigh_only.py 3072 2 > /root/repo/gpurun_out/symv/prof3072.log 2>&1
cd /root/repo
db=$(find gpurun_out/symv/prof3072 -name "*.db" | head -1)
python3 - $db <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, duration from kernels order by start").fetchall()
seq = []
for name, st, du in rows:
    if 'tridiag_tail' in name or 'trd_tail_lds' in name: break
    k = 'symv' if 'trd_symv_kernel' in name else 'gemv' if 'trd_gemv_kernel' in name else 'rowF' if ('trd_row_kernel' in name and 'true' in name) else 'row' if 'trd_row_kernel' in name else None
    if k: seq.append((k, du))
import collections
col = 0; bins = collections.defaultdict(lambda: collections.defaultdict(list))
for k, du in seq:
    if k in ('symv', 'gemv'): col += 1
    m = 3072 - 1 - max(col, 1)
    bins[m // 256][k].append(du)
for b in sorted(bins, reverse=True):
    print(b * 256, {k: round(sum(v) / len(v) / 1e3, 2) for k, v in bins[b].items()})
PY
rm -rf gpurun_out/symv/prof3072
