#!/bin/bash
OUT=gpurun_out/r2i; mkdir -p $OUT
export TMPDIR=/tmp
echo "== ensemble probe, GPU_MAX_HW_QUEUES=1 in the workers" | tee -a $OUT/session.log
timeout 300 python tools/ensemble_probe.py 16 4 6 8 > $OUT/probe1.log 2>&1; grep pool $OUT/probe1.log | tee -a $OUT/session.log
echo "== the same with 2 queues" | tee -a $OUT/session.log
SELLA_POOL_HW_QUEUES=2 timeout 300 python tools/ensemble_probe.py 16 4 6 8 > $OUT/probe2.log 2>&1; grep pool $OUT/probe2.log | tee -a $OUT/session.log
