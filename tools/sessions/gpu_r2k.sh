#!/bin/bash
OUT=gpurun_out/r2k; mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu" | tee -a $OUT/session.log
timeout 1200 python -m pytest tests -q -m gpu -x > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log | tee -a $OUT/session.log
for n in 3072 768; do
echo "== optimizer profile n=$n" | tee -a $OUT/session.log
SELLA_DEBUG_TIMING=1 timeout 300 python tools/opt_profile.py $n 20 > $OUT/opt_$n.log 2> $OUT/opt_${n}_timing.log; head -12 $OUT/opt_$n.log | tee -a $OUT/session.log
grep "stepper\|update_H" $OUT/opt_${n}_timing.log | tail -2 | tee -a $OUT/session.log
done
echo "== emt slab" | tee -a $OUT/session.log
SELLA_DEBUG_TIMING=1 timeout 300 python tools/emt_slab_opt.py > $OUT/emt.log 2> $OUT/emt_timing.log; grep "per optimizer step" -A10 $OUT/emt.log | tee -a $OUT/session.log
grep "update_H\|rank-one" $OUT/emt_timing.log | tail -5 | tee -a $OUT/session.log
echo "== bench" | tee -a $OUT/session.log
timeout 900 python bench.py > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | tee -a $OUT/session.log
