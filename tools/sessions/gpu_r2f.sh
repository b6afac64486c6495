#!/bin/bash
# ensemble worker-process sweep + EMT slab step breakdown
OUT=gpurun_out/r2f; mkdir -p $OUT
export TMPDIR=/tmp
for P in 0 4 8 16; do
  echo "== ensemble procs $P" | tee -a $OUT/session.log
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --emt-steps 0 --block-n 0 --converged-n 0 --ensemble-procs $P > $OUT/bench_p$P.log 2>&1
  tail -1 $OUT/bench_p$P.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['optimizer']['ensemble'])" | tee -a $OUT/session.log
done
echo "== emt slab" | tee -a $OUT/session.log
SELLA_DEBUG_TIMING=1 timeout 300 python tools/emt_slab_opt.py > $OUT/emt.log 2> $OUT/emt_timing.log
grep -v "^IRC\|^Sella" $OUT/emt.log | head -40 | tee -a $OUT/session.log
python - <<'PY' | tee -a $OUT/session.log
import re,collections
agg=collections.defaultdict(lambda:[0,0.0])
for ln in open('gpurun_out/r2f/emt_timing.log'):
    k=ln.split(':')[0][:40]
    agg[k][0]+=1
print({k:v[0] for k,v in agg.items()})
PY
tail -30 $OUT/emt_timing.log | tee -a $OUT/session.log
