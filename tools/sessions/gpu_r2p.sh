#!/bin/bash
OUT=gpurun_out/r2p; mkdir -p $OUT
timeout 600 python bench.py --no-cpu-baseline --block-n 0 > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['vectors_per_call'], d['davidson_loop_only_iter_per_s'], d['optimizer']['optimizer_steps_per_s'], d['optimizer']['ensemble']['searches_per_s'], d['optimizer']['emt_slab']['ms_per_step'])" | tee -a $OUT/session.log
