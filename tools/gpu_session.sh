#!/bin/bash
# One GPU visit: smoke, gpu tests, bench line, rocprof kernel trace of the bench command, PMC passes.
# Usage (from the repo root, via gpurun):  bash tools/gpu_session.sh [tag] [fast]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== smoke" | tee $OUT/session.log
timeout 600 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/session.log
tail -3 $OUT/smoke.log | tee -a $OUT/session.log
if [ "$2" != "fast" ]; then
echo "== pytest -m gpu" | tee -a $OUT/session.log
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/session.log
tail -6 $OUT/pytest_gpu.log | tee -a $OUT/session.log
fi
echo "== bench" | tee -a $OUT/session.log
timeout 900 python bench.py > $OUT/bench.log 2>&1; echo "bench exit $?" | tee -a $OUT/session.log
tail -1 $OUT/bench.log | tee -a $OUT/session.log
echo "== rocprof kernel trace of the bench command" | tee -a $OUT/session.log
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --ensemble-procs 0 > $R/$OUT/rocprof.log 2>&1); echo "rocprof exit $?" | tee -a $OUT/session.log
DB=$(find $OUT/prof -name "*.db" | head -1)
python tools/rocprof_summary.py $DB $OUT/bench_kernel_stats.md "bench.py --steps 5 --warmup 1 --no-cpu-baseline --ensemble-procs 0 (rocprofv3 --kernel-trace --stats)" > /dev/null
head -20 $OUT/bench_kernel_stats.md | tee -a $OUT/session.log
echo "== rocprof kernel trace of the eigensolver alone (4 calls at n = 3072)" | tee -a $OUT/session.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_eigh -o eigh -- python $R/tools/eigh_only.py 3072 4 > $R/$OUT/rocprof_eigh.log 2>&1); echo "rocprof eigh exit $?" | tee -a $OUT/session.log
DBE=$(find $OUT/prof_eigh -name "*.db" | head -1)
python tools/rocprof_summary.py $DBE $OUT/eigh_kernel_stats.md "tools/eigh_only.py 3072 4 (rocprofv3 --kernel-trace --stats)" > /dev/null
head -16 $OUT/eigh_kernel_stats.md | tee -a $OUT/session.log
rm -rf $OUT/prof_eigh
echo "== PMC passes (own runs, no tracing domains besides kernel dispatch)" | tee -a $OUT/session.log
for CNT in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $CNT --kernel-trace -d $R/$OUT/pmc_$CNT -o pmc -- python $R/tools/eigh_only.py 3072 1 > $R/$OUT/pmc_$CNT.log 2>&1); echo "pmc $CNT exit $?" | tee -a $OUT/session.log
  DBP=$(find $OUT/pmc_$CNT -name "*.db" | head -1)
  python tools/pmc_dump.py $DBP trd_gemv > $OUT/pmc_$CNT.txt 2>&1
  head -12 $OUT/pmc_$CNT.txt | tee -a $OUT/session.log
  rm -rf $OUT/pmc_$CNT
done
echo "== MFMA utilisation (PMC, own runs): block product H.V at 3N = 12288, eigensolver GEMMs / back-transformation" | tee -a $OUT/session.log
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/$OUT/pmc_mfma_panel -o pmc -- python $R/tools/panel_bench.py 12288 16 > $R/$OUT/pmc_mfma_panel.log 2>&1); echo "pmc mfma panel exit $?" | tee -a $OUT/session.log
DBM=$(find $OUT/pmc_mfma_panel -name "*.db" | head -1)
python tools/mfma_util.py $DBM panel16_mfma_kernel 4831838208 > $OUT/pmc_mfma.txt 2>&1
rm -rf $OUT/pmc_mfma_panel
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/$OUT/pmc_mfma_eigh -o pmc -- python $R/tools/eigh_only.py 3072 1 > $R/$OUT/pmc_mfma_eigh.log 2>&1); echo "pmc mfma eigh exit $?" | tee -a $OUT/session.log
DBM=$(find $OUT/pmc_mfma_eigh -name "*.db" | head -1)
python tools/mfma_util.py $DBM wy_apply_mfma_kernel 57982058496 >> $OUT/pmc_mfma.txt 2>&1
python tools/mfma_util.py $DBM gemm128_merge_batched_kernel >> $OUT/pmc_mfma.txt 2>&1
python tools/mfma_util.py $DBM rank2k_stream_kernel >> $OUT/pmc_mfma.txt 2>&1
rm -rf $OUT/pmc_mfma_eigh
cat $OUT/pmc_mfma.txt | tee -a $OUT/session.log
rm -f $OUT/prof/*/*.db.tmp
echo "== configs[2]: internal coordinates / geodesic at 1024 atoms" | tee -a $OUT/session.log
timeout 600 python tools/geodesic_bench.py --steps 3 --sella-steps 2 > $OUT/geodesic.log 2>&1; echo "geodesic exit $?" | tee -a $OUT/session.log
grep "^{" $OUT/geodesic.log | cut -c1-400 | tee -a $OUT/session.log
EXACT_GEODESIC=1 timeout 600 python tools/geodesic_bench.py --steps 1 --sella-steps 2 > $OUT/geodesic_exact.log 2>&1; grep "Sella(internal)" $OUT/geodesic_exact.log | cut -c1-300 | tee -a $OUT/session.log
echo "== Davidson loop alone" | tee -a $OUT/session.log
SELLA_DEBUG_TIMING=1 timeout 300 python tools/dav_time.py > $OUT/dav_time.log 2>&1; grep -v "^davidson\|^eigh" $OUT/dav_time.log | tail -4 | tee -a $OUT/session.log; grep "^davidson" $OUT/dav_time.log | tail -1 | tee -a $OUT/session.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_dav -o dav -- python $R/tools/dav_time.py > $R/$OUT/rocprof_dav.log 2>&1); echo "rocprof dav exit $?" | tee -a $OUT/session.log
DBD=$(find $OUT/prof_dav -name "*.db" | head -1)
python tools/rocprof_summary.py $DBD $OUT/dav_kernel_stats.md "tools/dav_time.py (rocprofv3 --kernel-trace --stats)" > /dev/null
rm -rf $OUT/prof_dav
echo "== EMT slab, optimizer profile" | tee -a $OUT/session.log
SELLA_DEBUG_TIMING=1 timeout 300 python tools/emt_slab_opt.py > $OUT/emt.log 2> $OUT/emt_timing.log; grep "per optimizer step" -A8 $OUT/emt.log | tee -a $OUT/session.log
grep "update_H\|rank-one" $OUT/emt_timing.log | tail -5 | tee -a $OUT/session.log
SELLA_DEBUG_TIMING=1 timeout 300 python tools/opt_profile.py 3072 20 > $OUT/opt_3072.log 2> $OUT/opt_3072_timing.log; head -10 $OUT/opt_3072.log | tee -a $OUT/session.log
grep "update_H" $OUT/opt_3072_timing.log | tail -1 | tee -a $OUT/session.log
echo "== ensemble worker processes" | tee -a $OUT/session.log
timeout 300 python tools/ensemble_probe.py 16 2 4 > $OUT/probe.log 2>&1; grep pool $OUT/probe.log | tee -a $OUT/session.log
echo "== rccl (1 rank)" | tee -a $OUT/session.log
timeout 120 python tools/rccl_smoke.py > $OUT/rccl.log 2>&1; tail -1 $OUT/rccl.log | tee -a $OUT/session.log
