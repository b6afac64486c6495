#!/bin/bash
OUT=gpurun_out/r2g; mkdir -p $OUT
export TMPDIR=/tmp
for P in 2 4 6 7 8 12; do
  echo "== ensemble procs $P (1 BLAS thread per worker)" | tee -a $OUT/session.log
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --emt-steps 0 --block-n 0 --converged-n 0 --ensemble-procs $P > $OUT/bench_p$P.log 2>&1
  tail -1 $OUT/bench_p$P.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['optimizer']['ensemble'])" | tee -a $OUT/session.log
done
echo "== 16 members per GPU, 8 procs" | tee -a $OUT/session.log
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --emt-steps 0 --block-n 0 --converged-n 0 --ensemble-procs 8 --ensemble-per-gpu 16 > $OUT/bench_m16.log 2>&1
tail -1 $OUT/bench_m16.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['optimizer']['ensemble'])" | tee -a $OUT/session.log
nproc | tee -a $OUT/session.log; cat /sys/fs/cgroup/cpu.max | tee -a $OUT/session.log
