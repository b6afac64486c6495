#!/bin/bash
OUT=gpurun_out/r2l; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== ensemble test with worker processes + eigh tests" | tee -a $OUT/session.log
timeout 900 python -m pytest tests/test_big_gpu.py tests/test_eigh.py -q -m gpu -x -k "ensemble or eigh" > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log | tee -a $OUT/session.log
echo "== eigh alone" | tee -a $OUT/session.log
SELLA_DEBUG_TIMING=1 timeout 300 python tools/eigh_only.py 3072 6 > $OUT/eigh.log 2>&1; tail -4 $OUT/eigh.log | tee -a $OUT/session.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_eigh -o eigh -- python $R/tools/eigh_only.py 3072 4 > $R/$OUT/rocprof_eigh.log 2>&1)
DBE=$(find $OUT/prof_eigh -name "*.db" | head -1)
python tools/rocprof_summary.py $DBE $OUT/eigh_kernel_stats.md "tools/eigh_only.py 3072 4 (rocprofv3 --kernel-trace --stats)" > /dev/null
head -14 $OUT/eigh_kernel_stats.md | tee -a $OUT/session.log
rm -rf $OUT/prof_eigh
