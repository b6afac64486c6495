#!/bin/bash
# Lean GPU visit: smoke, gpu tests, optional extras given as further arguments (each a quoted command whose output goes
# to gpurun_out/<tag>/extra_<k>.log).  Usage (repo root, via gpurun):  bash tools/gpu_quick.sh <tag> [notests] [cmd ...]
TAG=${1:-quick}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== smoke" | tee $OUT/session.log
timeout 600 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/session.log
tail -2 $OUT/smoke.log | tee -a $OUT/session.log
if [ "$1" == "notests" ]; then shift; else
echo "== pytest -m gpu" | tee -a $OUT/session.log
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/session.log
tail -15 $OUT/pytest_gpu.log | tee -a $OUT/session.log
fi
K=0
for CMD in "$@"; do
  K=$((K+1))
  echo "== extra $K: $CMD" | tee -a $OUT/session.log
  timeout 900 bash -c "$CMD" > $OUT/extra_$K.log 2>&1; echo "exit $?" | tee -a $OUT/session.log
  tail -40 $OUT/extra_$K.log | cut -c1-600 | tee -a $OUT/session.log
done
