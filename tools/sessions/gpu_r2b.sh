#!/bin/bash
OUT=gpurun_out/r2b
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== rccl smoke" | tee $OUT/session.log
NCCL_DEBUG=WARN timeout 120 python tools/rccl_smoke.py > $OUT/rccl.log 2>&1; echo "rccl exit $?" | tee -a $OUT/session.log; tail -3 $OUT/rccl.log | tee -a $OUT/session.log
echo "== rccl smoke 1 rank via torchrun" | tee -a $OUT/session.log
timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/rccl_smoke.py > $OUT/rccl2.log 2>&1; echo "rccl2 exit $?" | tee -a $OUT/session.log; tail -2 $OUT/rccl2.log | tee -a $OUT/session.log
echo "== dav_time fast" | tee -a $OUT/session.log
SELLA_DEBUG_TIMING=1 timeout 300 python tools/dav_time.py > $OUT/dav_fast.log 2>&1; tail -6 $OUT/dav_fast.log | tee -a $OUT/session.log
echo "== dav_time sync" | tee -a $OUT/session.log
SELLA_DAV_SYNC=1 timeout 300 python tools/dav_time.py > $OUT/dav_sync.log 2>&1; tail -4 $OUT/dav_sync.log | tee -a $OUT/session.log
echo "== rocprof dav_time" | tee -a $OUT/session.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_dav -o dav -- python $R/tools/dav_time.py > $R/$OUT/rocprof_dav.log 2>&1); echo "rocprof exit $?" | tee -a $OUT/session.log
DB=$(find $OUT/prof_dav -name "*.db" | head -1)
python tools/rocprof_summary.py $DB $OUT/dav_kernel_stats.md "tools/dav_time.py (rocprofv3 --kernel-trace --stats)" > /dev/null
head -30 $OUT/dav_kernel_stats.md | tee -a $OUT/session.log
rm -rf $OUT/prof_dav
echo "== tests" | tee -a $OUT/session.log
timeout 900 python -m pytest tests/test_big_gpu.py tests/test_eigensolvers.py tests/test_oracle_golden.py tests/test_pes_sella.py -m gpu -q -x -k "not converged_eigenpair" --durations=5 > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/session.log; tail -15 $OUT/pytest.log | tee -a $OUT/session.log
echo "== bench" | tee -a $OUT/session.log
timeout 600 python bench.py --no-cpu-baseline --block-iters 0 > $OUT/bench.log 2>&1; echo "bench exit $?" | tee -a $OUT/session.log; tail -1 $OUT/bench.log | cut -c1-1500 | tee -a $OUT/session.log
