#!/bin/bash
OUT=gpurun_out/r2c
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== dav_time fast" | tee $OUT/session.log
SELLA_DEBUG_TIMING=1 timeout 300 python tools/dav_time.py > $OUT/dav_fast.log 2>&1; grep -v "^davidson" $OUT/dav_fast.log | tee -a $OUT/session.log; grep "^davidson" $OUT/dav_fast.log | tail -2 | tee -a $OUT/session.log
echo "== rocprof dav_time" | tee -a $OUT/session.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_dav -o dav -- python $R/tools/dav_time.py > $R/$OUT/rocprof_dav.log 2>&1); echo "rocprof exit $?" | tee -a $OUT/session.log
DB=$(find $OUT/prof_dav -name "*.db" | head -1)
python tools/rocprof_summary.py $DB $OUT/dav_kernel_stats.md "tools/dav_time.py (rocprofv3 --kernel-trace --stats)" > /dev/null
head -22 $OUT/dav_kernel_stats.md | tee -a $OUT/session.log
rm -rf $OUT/prof_dav
echo "== emt slab profile" | tee -a $OUT/session.log
SELLA_DEBUG_TIMING=1 timeout 300 python tools/emt_slab_opt.py > $OUT/emt_slab.log 2>&1; tail -12 $OUT/emt_slab.log | tee -a $OUT/session.log
echo "== tests" | tee -a $OUT/session.log
timeout 1200 python -m pytest tests -m gpu -q -x -k "not converged_eigenpair and not 12288" --durations=8 > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/session.log; tail -15 $OUT/pytest.log | tee -a $OUT/session.log
echo "== bench" | tee -a $OUT/session.log
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.log 2>&1; echo "bench exit $?" | tee -a $OUT/session.log; tail -1 $OUT/bench.log | cut -c1-2500 | tee -a $OUT/session.log
