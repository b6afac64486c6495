#!/bin/bash
OUT=gpurun_out/r2n; mkdir -p $OUT
for NB in 8 16 24 32; do for LEAF in 16 32 64; do
echo "nb=$NB leaf=$LEAF: $(EIGH_NB=$NB EIGH_LEAF=$LEAF SELLA_DEBUG_TIMING=1 timeout 120 python tools/eigh_only.py 3072 4 2>&1 | grep tridiag | tail -1)" | tee -a $OUT/session.log
done; done
