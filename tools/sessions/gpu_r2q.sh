#!/bin/bash
OUT=gpurun_out/r2q; mkdir -p $OUT
cd examples && timeout 600 python cu111_adatom.py > ../$OUT/example.log 2>&1; cd ..; tail -6 $OUT/example.log | tee -a $OUT/session.log
