#!/bin/bash
# round-2 visit A: new tests first, then the whole gpu suite, rccl smoke, bench
OUT=gpurun_out/r2a
mkdir -p $OUT
export TMPDIR=/tmp
echo "== rccl smoke" | tee $OUT/session.log
timeout 120 python tools/rccl_smoke.py > $OUT/rccl.log 2>&1; echo "rccl exit $?" | tee -a $OUT/session.log; tail -3 $OUT/rccl.log | tee -a $OUT/session.log
echo "== smoke" | tee -a $OUT/session.log
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/session.log; tail -2 $OUT/smoke.log | tee -a $OUT/session.log
echo "== new tests" | tee -a $OUT/session.log
timeout 900 python -m pytest tests/test_big_gpu.py tests/test_block_davidson.py -m gpu -q -x --durations=8 > $OUT/pytest_new.log 2>&1; echo "pytest new exit $?" | tee -a $OUT/session.log; tail -25 $OUT/pytest_new.log | tee -a $OUT/session.log
echo "== bench" | tee -a $OUT/session.log
timeout 600 python bench.py > $OUT/bench.log 2>&1; echo "bench exit $?" | tee -a $OUT/session.log; tail -1 $OUT/bench.log | tee -a $OUT/session.log
echo "== all gpu tests" | tee -a $OUT/session.log
timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_big_gpu.py --deselect tests/test_block_davidson.py > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/session.log; tail -8 $OUT/pytest_gpu.log | tee -a $OUT/session.log
