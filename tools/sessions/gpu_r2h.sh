#!/bin/bash
OUT=gpurun_out/r2h; mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest subset" | tee -a $OUT/session.log
timeout 900 python -m pytest tests/test_step_solve.py tests/test_trajectory.py tests/test_pes_sella.py tests/test_irc.py -q -m gpu -x > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log | tee -a $OUT/session.log
echo "== emt slab (batched bisection)" | tee -a $OUT/session.log
timeout 300 python tools/emt_slab_opt.py > $OUT/emt.log 2>&1; grep "per optimizer step" -A8 $OUT/emt.log | tee -a $OUT/session.log
echo "== ensemble probe" | tee -a $OUT/session.log
timeout 600 python tools/ensemble_probe.py 16 4 3 > $OUT/probe.log 2>&1; cat $OUT/probe.log | tee -a $OUT/session.log
