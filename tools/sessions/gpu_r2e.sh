#!/bin/bash
OUT=gpurun_out/r2e
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== smoke" | tee $OUT/session.log
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/session.log; tail -1 $OUT/smoke.log | tee -a $OUT/session.log
echo "== all gpu tests" | tee -a $OUT/session.log
timeout 1500 python -m pytest tests -m gpu -q --durations=6 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/session.log; tail -12 $OUT/pytest_gpu.log | tee -a $OUT/session.log
echo "== ensemble member profile" | tee -a $OUT/session.log
SELLA_DEBUG_TIMING=1 timeout 300 python tools/ens_profile.py > $OUT/ens_profile.log 2>&1; grep -v "^rank-one\|^update_H\|^stepper\|^eigh" $OUT/ens_profile.log | tail -30 | tee -a $OUT/session.log
echo "== bench" | tee -a $OUT/session.log
timeout 600 python bench.py > $OUT/bench.log 2>&1; echo "bench exit $?" | tee -a $OUT/session.log; tail -1 $OUT/bench.log | cut -c1-3000 | tee -a $OUT/session.log
