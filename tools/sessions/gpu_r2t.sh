#!/bin/bash
OUT=gpurun_out/r2t; mkdir -p $OUT
timeout 600 python bench.py --no-cpu-baseline --no-optimizer --steps 3 --warmup 1 > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['vectors_per_call'], d['block_davidson'])" | tee -a $OUT/session.log
timeout 600 python -m pytest tests/test_big_gpu.py tests/test_block_davidson.py -q -m gpu -x -k "block" 2>&1 | tail -1 | tee -a $OUT/session.log
