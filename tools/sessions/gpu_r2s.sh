#!/bin/bash
OUT=gpurun_out/r2s; mkdir -p $OUT
for i in 1 2 3; do timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $OUT/pytest_$i.log 2>&1; tail -1 $OUT/pytest_$i.log | tee -a $OUT/session.log; done
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1 | tee -a $OUT/session.log
