#!/bin/bash
# GPU visit for the one-launch-per-column chain of the tridiagonalisation (trd_upd_kernel): correctness at 3N = 3072, timing
# by switch-over size / rows per workgroup, per-m kernel times from traces.   Usage (repo root, via gpurun): bash tools/col_sweep.sh <tag>
TAG=${1:-col}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
say() { echo "$@" | tee -a $OUT/session.log; }
: > $OUT/session.log
say "== check (default)"; EIGH_CHECK=1 timeout 300 python tools/eigh_only.py 3072 3 2>&1 | tail -4 | tee -a $OUT/session.log
say "== old path"; EIGH_OPTS=eigh_upd_max=0 timeout 300 python tools/eigh_only.py 3072 4 2>&1 | tail -2 | tee -a $OUT/session.log
for MX in 512 1024 1536 1792 2048 2560 3072; do
  say "== upd_max $MX"; EIGH_OPTS=eigh_upd_max=$MX timeout 300 python tools/eigh_only.py 3072 4 2>&1 | tail -1 | tee -a $OUT/session.log
done
for RR in 2 4 8; do
  for NT in 512 256; do
  say "== rows $RR nt $NT"; EIGH_OPTS=eigh_upd_rows=$RR,eigh_upd_nt=$NT,eigh_upd_max=3072 timeout 300 python tools/eigh_only.py 3072 3 2>&1 | tail -1 | tee -a $OUT/session.log
  done
done
for RR in 2 4 8; do
  (cd /tmp && rm -rf /tmp/tr$RR && EIGH_OPTS=eigh_upd_rows=$RR,eigh_upd_max=3072 timeout 300 rocprofv3 --kernel-trace -d /tmp/tr$RR -o tr -- python $R/tools/eigh_only.py 3072 2 > /dev/null 2>&1)
  db=$(find /tmp/tr$RR -name "*.db" | head -1)
  say "== per-m, rows $RR"; python tools/col_by_m.py $db 3072 | tee -a $OUT/session.log
done
(cd /tmp && rm -rf /tmp/tr9 && EIGH_OPTS=eigh_upd_rows=4,eigh_upd_nt=256,eigh_upd_max=3072 timeout 300 rocprofv3 --kernel-trace -d /tmp/tr9 -o tr -- python $R/tools/eigh_only.py 3072 2 > /dev/null 2>&1)
db=$(find /tmp/tr9 -name "*.db" | head -1)
say "== per-m, rows 4 nt 256"; python tools/col_by_m.py $db 3072 | tee -a $OUT/session.log
(cd /tmp && rm -rf /tmp/trs && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/trs -o tr -- python $R/tools/eigh_only.py 3072 4 > /dev/null 2>&1)
db=$(find /tmp/trs -name "*.db" | head -1)
python tools/rocprof_summary.py $db $OUT/eigh_kernel_stats.md "tools/eigh_only.py 3072 4 (rocprofv3 --kernel-trace --stats)" | head -30 | tee -a $OUT/session.log
say "== gpu tests (eigh, big)"; timeout 900 python -m pytest tests/test_eigh.py tests/test_big_gpu.py -m gpu -q -x 2>&1 | tail -5 | tee -a $OUT/session.log
