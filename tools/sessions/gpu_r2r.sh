#!/bin/bash
OUT=gpurun_out/r2r; mkdir -p $OUT
for LEAF in 16 32; do
for N in 768 1536 3072; do
echo "n=$N leaf=$LEAF: $(EIGH_LEAF=$LEAF SELLA_DEBUG_TIMING=1 timeout 120 python tools/eigh_only.py $N 4 2>&1 | grep tridiag | tail -1)" | tee -a $OUT/session.log
done
echo "n=12288 leaf=$LEAF: $(EIGH_LEAF=$LEAF SELLA_DEBUG_TIMING=1 timeout 300 python tools/eigh_only.py 12288 2 2>&1 | grep tridiag | tail -1)" | tee -a $OUT/session.log
done
