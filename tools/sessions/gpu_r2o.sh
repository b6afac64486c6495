#!/bin/bash
OUT=gpurun_out/r2o; mkdir -p $OUT
grep -m1 "model name" /proc/cpuinfo | tee -a $OUT/session.log
grep -m1 -o "avx2\|avx512f" /proc/cpuinfo | sort -u | tr '\n' ' ' | tee -a $OUT/session.log; echo | tee -a $OUT/session.log
SELLA_DEBUG_TIMING=1 timeout 300 python tools/dav_time.py > $OUT/dav_time.log 2>&1; grep -v "^davidson\|^eigh" $OUT/dav_time.log | tail -4 | tee -a $OUT/session.log; grep "^davidson" $OUT/dav_time.log | tail -1 | tee -a $OUT/session.log
timeout 300 python tools/opt_profile.py 3072 20 2>/dev/null | head -3 | tee -a $OUT/session.log
timeout 300 python tools/emt_slab_opt.py 2>/dev/null | grep "per optimizer step" | tee -a $OUT/session.log
timeout 1200 python -m pytest tests -q -m gpu -x > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log | tee -a $OUT/session.log
