#!/bin/bash
# One GPU visit: smoke, gpu tests, micro-benchmarks, bench line, rocprof kernel trace.
# Usage (from the repo root, via gpurun):  bash tools/gpu_session.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== smoke" | tee $OUT/session.log
timeout 600 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/session.log
tail -3 $OUT/smoke.log | tee -a $OUT/session.log
echo "== pytest -m gpu" | tee -a $OUT/session.log
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/session.log
tail -15 $OUT/pytest_gpu.log | tee -a $OUT/session.log
echo "== microbench" | tee -a $OUT/session.log
MB_SIZES=${MB_SIZES:-768,3072} timeout 900 python tools/microbench.py > $OUT/microbench.log 2>&1; echo "microbench exit $?" | tee -a $OUT/session.log
tail -5 $OUT/microbench.log | tee -a $OUT/session.log
echo "== bench" | tee -a $OUT/session.log
timeout 900 python bench.py --steps 10 --warmup 2 > $OUT/bench.log 2>&1; echo "bench exit $?" | tee -a $OUT/session.log
tail -2 $OUT/bench.log | tee -a $OUT/session.log
echo "== rocprof" | tee -a $OUT/session.log
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1); echo "rocprof exit $?" | tee -a $OUT/session.log
find $OUT/prof -name "*stats*" | head | tee -a $OUT/session.log
