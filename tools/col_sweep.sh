#!/bin/bash
# GPU visit for the one-launch-per-column tridiagonalisation: correctness at 3N = 3072, timing by rows per workgroup /
# panel width, per-m kernel times from traces.   Usage (repo root, via gpurun): bash tools/col_sweep.sh <tag>
TAG=${1:-col}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
say() { echo "$@" | tee -a $OUT/session.log; }
: > $OUT/session.log
say "== check (fused, auto)"; EIGH_CHECK=1 timeout 300 python tools/eigh_only.py 3072 3 2>&1 | tail -4 | tee -a $OUT/session.log
say "== old path"; EIGH_OPTS=eigh_fused=0 timeout 300 python tools/eigh_only.py 3072 4 2>&1 | tail -2 | tee -a $OUT/session.log
for RR in 2 4 8 16; do
  say "== fused, rows $RR"; EIGH_OPTS=eigh_col_rows=$RR timeout 300 python tools/eigh_only.py 3072 4 2>&1 | tail -2 | tee -a $OUT/session.log
done
for NB in 8 12 16 24 32; do
  say "== fused auto, nb $NB"; EIGH_NB=$NB timeout 300 python tools/eigh_only.py 3072 4 2>&1 | tail -2 | tee -a $OUT/session.log
done
say "== fused auto, nt 256"; EIGH_OPTS=eigh_col_nt=256 timeout 300 python tools/eigh_only.py 3072 4 2>&1 | tail -2 | tee -a $OUT/session.log
for RR in 2 4 8 16; do
  (cd /tmp && rm -rf /tmp/tr$RR && EIGH_OPTS=eigh_col_rows=$RR timeout 300 rocprofv3 --kernel-trace -d /tmp/tr$RR -o tr -- python $R/tools/eigh_only.py 3072 2 > /dev/null 2>&1)
  db=$(find /tmp/tr$RR -name "*.db" | head -1)
  say "== per-m, rows $RR"; python tools/col_by_m.py $db 3072 | tee -a $OUT/session.log
done
(cd /tmp && rm -rf /tmp/tr0 && EIGH_OPTS=eigh_fused=0 timeout 300 rocprofv3 --kernel-trace -d /tmp/tr0 -o tr -- python $R/tools/eigh_only.py 3072 2 > /dev/null 2>&1)
db=$(find /tmp/tr0 -name "*.db" | head -1)
say "== per-m, old path"; python tools/trd_by_m.py $db 256 3072 | tee -a $OUT/session.log
(cd /tmp && rm -rf /tmp/trs && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/trs -o tr -- python $R/tools/eigh_only.py 3072 4 > /dev/null 2>&1)
db=$(find /tmp/trs -name "*.db" | head -1)
python tools/rocprof_summary.py $db $OUT/eigh_kernel_stats.md "tools/eigh_only.py 3072 4, one launch per column (rocprofv3 --kernel-trace --stats)" | head -30 | tee -a $OUT/session.log
say "== gpu tests (eigh, big)"; timeout 900 python -m pytest tests/test_eigh.py tests/test_big_gpu.py -m gpu -q -x 2>&1 | tail -5 | tee -a $OUT/session.log
