#!/bin/bash
# One GPU visit for the evidence kept under profiles/: smoke, gpu tests, bench line, rocprofv3 kernel statistics of the
# bench / the eigensolver / the Davidson loop / the block iteration / the optimizer step, PMC passes (own runs, kernel
# dispatch tracing only).   Usage (repo root, via gpurun):  bash tools/gpu_session.sh [tag] [fast]
TAG=${1:-r04}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
say() { echo "$@" | tee -a $OUT/session.log; }
prof() {   # prof <name> <title> <cmd...>: kernel trace + stats of a command, summary -> $OUT/<name>_kernel_stats.md
  local name=$1 title=$2; shift 2
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_$name -o $name -- "$@" > $R/$OUT/rocprof_$name.log 2>&1); say "rocprof $name exit $?"
  local db=$(find $OUT/prof_$name -name "*.db" | head -1)
  python tools/rocprof_summary.py $db $OUT/${name}_kernel_stats.md "$title" > /dev/null
  head -18 $OUT/${name}_kernel_stats.md | tee -a $OUT/session.log
  rm -rf $OUT/prof_$name
}
pmc() {    # pmc <name> <kernel substring> <counters...> -- <cmd...>
  local name=$1 needle=$2; shift 2
  local ctr=()
  while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  (cd /tmp && timeout 600 rocprofv3 --pmc "${ctr[@]}" --kernel-trace -d $R/$OUT/pmc_$name -o pmc -- "$@" > $R/$OUT/pmc_$name.log 2>&1); say "pmc $name (${ctr[*]}) exit $?"
  local db=$(find $OUT/pmc_$name -name "*.db" | head -1)
  python tools/pmc_dump.py $db "$needle" 2>&1 | grep -v "^tables\|columns:" > $OUT/pmc_$name.txt
  head -8 $OUT/pmc_$name.txt | cut -c1-220 | tee -a $OUT/session.log
  rm -rf $OUT/pmc_$name
}
: > $OUT/session.log
say "== smoke"
timeout 600 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; say "smoke exit $?"
tail -2 $OUT/smoke.log | tee -a $OUT/session.log
if [ "$2" != "fast" ]; then
say "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; say "pytest exit $?"
tail -5 $OUT/pytest_gpu.log | tee -a $OUT/session.log
fi
say "== bench"
timeout 900 python bench.py > $OUT/bench.log 2>&1; say "bench exit $?"
tail -1 $OUT/bench.log > $OUT/bench_line.json; cat $OUT/bench_line.json | tee -a $OUT/session.log
say "== kernel statistics"
# (the threaded legs — ensemble threads, cohorts' issuing threads, concurrent problems — have crashed rocprofv3's kernel tracing
#  inside a launch issued from a worker thread, r03D / r05G: the statistics are taken with those legs off)
prof bench "bench.py --steps 5 --warmup 1 --no-cpu-baseline --ensemble-procs 0 --ensemble-threads 0 --concurrent 1 (rocprofv3 --kernel-trace --stats)" python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --ensemble-procs 0 --ensemble-threads 0 --concurrent 1
prof eigh "tools/eigh_only.py 3072 4 (rocprofv3 --kernel-trace --stats)" python $R/tools/eigh_only.py 3072 4
prof davidson_loop "tools/dav_time.py (rocprofv3 --kernel-trace --stats)" python $R/tools/dav_time.py
prof block_iter "tools/block_iter.py 12288 12 (rocprofv3 --kernel-trace --stats)" python $R/tools/block_iter.py 12288 12
prof optimizer_step "tools/opt_profile.py 3072 20 (rocprofv3 --kernel-trace --stats)" python $R/tools/opt_profile.py 3072 20
# kernel timelines of ONE optimizer step (model PES: one job; EMT slab: full-space job + view job)
(cd /tmp && rm -rf /tmp/optl && timeout 600 rocprofv3 --kernel-trace -d /tmp/optl -o optl -- python $R/tools/opt_profile.py 3072 10 > /dev/null 2>&1)
python tools/opt_timeline_parse.py /tmp/optl 0.6 lr_pre_plan > $OUT/opt_step_timeline.txt 2>&1; head -1 $OUT/opt_step_timeline.txt | tee -a $OUT/session.log
(cd /tmp && rm -rf /tmp/emtl && timeout 600 rocprofv3 --kernel-trace -d /tmp/emtl -o emtl -- python $R/tools/emt_slab_lib.py 12 > /dev/null 2>&1)
python tools/opt_timeline_parse.py /tmp/emtl 0.5 lr_pre_plan 2 > $OUT/emt_step_timeline.txt 2>&1; head -1 $OUT/emt_step_timeline.txt | tee -a $OUT/session.log
say "== PMC passes"
for CNT in FETCH_SIZE WRITE_SIZE; do
  pmc eigh_$CNT trd_gemv $CNT -- python $R/tools/eigh_only.py 3072 1
done
# the same two databases hold the write-bound question of the streaming rank-2k update
pmc rank2k_FETCH rank2k_stream FETCH_SIZE -- python $R/tools/eigh_only.py 3072 1
pmc rank2k_WRITE rank2k_stream WRITE_SIZE -- python $R/tools/eigh_only.py 3072 1
# L2 re-read traffic of the back-transformation (requests from the CUs into L2, hits / misses there)
pmc wy_l2req wy_apply_ TCP_TCC_READ_REQ_sum -- python $R/tools/eigh_only.py 3072 1
pmc wy_l2hit wy_apply_ TCC_HIT_sum TCC_MISS_sum -- python $R/tools/eigh_only.py 3072 1
pmc wy_fetch wy_apply_ FETCH_SIZE -- python $R/tools/eigh_only.py 3072 1
say "== MFMA utilisation (PMC): block product, eigensolver kernels"
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/$OUT/pmc_mfma_panel -o pmc -- python $R/tools/panel_bench.py 12288 16 > $R/$OUT/pmc_mfma_panel.log 2>&1); say "pmc mfma panel exit $?"
DBM=$(find $OUT/pmc_mfma_panel -name "*.db" | head -1)
python tools/mfma_util.py $DBM panel16_mfma_kernel 4831838208 > $OUT/pmc_mfma.txt 2>&1
rm -rf $OUT/pmc_mfma_panel
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/$OUT/pmc_mfma_eigh -o pmc -- python $R/tools/eigh_only.py 3072 1 > $R/$OUT/pmc_mfma_eigh.log 2>&1); say "pmc mfma eigh exit $?"
DBM=$(find $OUT/pmc_mfma_eigh -name "*.db" | head -1)
python tools/mfma_util.py $DBM wy_apply_ 57982058496 >> $OUT/pmc_mfma.txt 2>&1
python tools/mfma_util.py $DBM gemm128_merge_batched_kernel >> $OUT/pmc_mfma.txt 2>&1
python tools/mfma_util.py $DBM rank2k_stream_fixed_kernel >> $OUT/pmc_mfma.txt 2>&1
rm -rf $OUT/pmc_mfma_eigh
cat $OUT/pmc_mfma.txt | tee -a $OUT/session.log
say "== timings"
SELLA_DEBUG_TIMING=1 timeout 300 python tools/dav_time.py > $OUT/dav_time.log 2>&1; grep -v "^davidson\|^eigh" $OUT/dav_time.log | tail -4 | tee -a $OUT/session.log; grep "^davidson" $OUT/dav_time.log | tail -1 | tee -a $OUT/session.log
BLOCK_ITER_CONVERGE=1 timeout 300 python tools/block_iter.py > $OUT/block_iter.log 2>&1; cat $OUT/block_iter.log | tee -a $OUT/session.log
SELLA_BD_TIMING=1 timeout 300 python tools/block_iter.py 12288 12 1 2>&1 | grep "^block davidson" | tail -4 | cut -c1-260 >> $OUT/block_iter.log
(cd /tmp && rm -rf /tmp/bt && timeout 300 rocprofv3 --kernel-trace -d /tmp/bt -o bt -- python $R/tools/block_iter.py 12288 12 1 > /dev/null 2>&1)
python tools/block_timeline_parse.py /tmp/bt > $OUT/block_iter_timeline.txt 2>&1; cat $OUT/block_iter_timeline.txt | tee -a $OUT/session.log
SELLA_DEBUG_TIMING=1 timeout 300 python tools/dav_fixed.py 2>&1 | grep "allocation\|maxiter" | tail -4 | tee -a $OUT/session.log
SELLA_DEBUG_TIMING=1 timeout 300 python tools/opt_profile.py 3072 20 > $OUT/opt_3072.log 2> $OUT/opt_3072_timing.log; head -12 $OUT/opt_3072.log | tee -a $OUT/session.log
grep "update_H\|rank-one" $OUT/opt_3072_timing.log | tail -3 | tee -a $OUT/session.log
SELLA_DEBUG_TIMING=1 timeout 300 python tools/emt_slab_opt.py > $OUT/emt.log 2> $OUT/emt_timing.log; grep "per optimizer step" -A8 $OUT/emt.log | tee -a $OUT/session.log
grep "update_H" $OUT/emt_timing.log | tail -2 | tee -a $OUT/session.log
say "== one-call optimizer step, A/B by option (lr_chain: fused launch chain of round 4; lr_pipe: force call queued in front of the update)"
timeout 600 python tools/opt_ab.py 3072 30 > $OUT/opt_ab.log 2>&1; cat $OUT/opt_ab.log | tee -a $OUT/session.log
{ python tools/emt_slab_lib.py 20; python tools/emt_slab_lib.py 20 lr_chain=0 lr_pipe=0; python tools/emt_slab_lib.py 20; } > $OUT/emt_lib_ab.log 2>&1; cat $OUT/emt_lib_ab.log | tee -a $OUT/session.log
say "== configs[3] as named: 256-atom EMT members on host threads"
timeout 300 python tools/emt_ensemble.py 64 t1 t4 t8 t16 > $OUT/emt_ensemble.log 2>&1; cat $OUT/emt_ensemble.log | tee -a $OUT/session.log
say "== configs[3] in lockstep cohorts (csrc/cohort.hip): one issuing thread per cohort, one batched launch per kernel for its members"
timeout 300 python tools/emt_ensemble.py 8 t8 c8 c4x2 c3x3 c2x4 > $OUT/emt_cohorts.log 2>&1; timeout 300 python tools/emt_ensemble.py 64 t12 c16x4 c8x8 >> $OUT/emt_cohorts.log 2>&1; grep -v "^  cohort" $OUT/emt_cohorts.log | tee -a $OUT/session.log
grep "^  cohort" $OUT/emt_cohorts.log | cut -c1-330 >> $OUT/session.log
SELLA_COHORT_TRACE=2 timeout 300 python tools/emt_ensemble.py 8 c8 > /dev/null 2> $OUT/cohort_by_kernel.log
KEEP_DB=1 bash tools/prof_cmd.sh $TAG cohort8 python $R/tools/emt_ensemble.py 8 c8 > /dev/null
NCO=$(grep -o "launches_issued.: [0-9]*" $OUT/rocprof_cohort8.log | tail -1 | grep -o "[0-9]*$")
{ grep "members in" $OUT/rocprof_cohort8.log; python tools/cohort_busy.py $OUT/cohort8.db $NCO; } > $OUT/cohort_busy.txt 2>&1; rm -f $OUT/cohort8.db; cat $OUT/cohort_busy.txt | cut -c1-200 | tee -a $OUT/session.log
KEEP_DB=1 bash tools/prof_cmd.sh $TAG member1 python $R/tools/emt_member_time.py > /dev/null; rm -f $OUT/member1.db
say "== eigensolver at 3N = 6144 / 8192 / 12288: defaults, then symmetric-aware matvec and 64-reflector blocks off"
{ for n in 6144 8192 12288; do
    echo "n = $n, defaults (eigh_symv_min 5120, eigh_wy_nb64_min 2560)"; timeout 300 python tools/eigh_only.py $n 3 2>&1 | tail -2
    echo "n = $n, eigh_symv_min 0, eigh_wy_nb64_min 0 (streaming matvec over the full block, 32-reflector blocks)"; EIGH_SYMV_MIN=0 EIGH_WY64_MIN=0 timeout 300 python tools/eigh_only.py $n 3 2>&1 | tail -2
  done; } > $OUT/eigh_large.log 2>&1; cat $OUT/eigh_large.log | tee -a $OUT/session.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_eigh12288 -o eigh12288 -- python $R/tools/eigh_only.py 12288 1 > $R/$OUT/rocprof_eigh12288.log 2>&1); say "rocprof eigh 12288 exit $?"
db=$(find $OUT/prof_eigh12288 -name "*.db" | head -1)
python tools/rocprof_summary.py $db $OUT/eigh12288_kernel_stats.md "tools/eigh_only.py 12288 1 (rocprofv3 --kernel-trace --stats)" > /dev/null
python tools/trd_by_m.py $db 1024 12288 > $OUT/eigh12288_by_m.txt 2>&1; head -14 $OUT/eigh12288_kernel_stats.md | tee -a $OUT/session.log; cat $OUT/eigh12288_by_m.txt | tee -a $OUT/session.log
rm -rf $OUT/prof_eigh12288
say "== tridiagonalisation by trailing size: blocked chain (row kernel + matvec), one-launch chain below eigh_upd_max; switch-over sweep"
(cd /tmp && rm -rf /tmp/trm && timeout 300 rocprofv3 --kernel-trace -d /tmp/trm -o tr -- python $R/tools/eigh_only.py 3072 2 > /dev/null 2>&1)
db=$(find /tmp/trm -name "*.db" | head -1)
{ python tools/trd_by_m.py $db 256 3072; python tools/col_by_m.py $db 3072; python tools/eigh_tail_timeline.py $db | tail -1; } > $OUT/eigh_by_m.txt 2>&1; cat $OUT/eigh_by_m.txt | tee -a $OUT/session.log
{ for MX in 0 512 1024 1536 2048; do echo -n "eigh_upd_max $MX: "; EIGH_OPTS=eigh_upd_max=$MX timeout 300 python tools/eigh_only.py 3072 4 2>&1 | tail -1; done;
  echo -n "eigh_wy_overlap 0: "; EIGH_OPTS=eigh_wy_overlap=0 timeout 300 python tools/eigh_only.py 3072 4 2>&1 | tail -1;
  for O in eigh_gemv_flat=0 rank2k_fixed=0 eigh_dc_pipeline=0 eigh_gemv_flat=1; do echo -n "$O: "; EIGH_OPTS=$O timeout 300 python tools/eigh_only.py 3072 4 2>&1 | tail -1; done;
  echo "divide & conquer, host side per level (SELLA_DC_TIMING):"; SELLA_DC_TIMING=1 timeout 300 python tools/eigh_only.py 3072 2 2>&1 | grep "dc level" | tail -8; } > $OUT/eigh_switch.log 2>&1; cat $OUT/eigh_switch.log | tee -a $OUT/session.log
say "== Davidson iteration, launch by launch"
(cd /tmp && rm -rf /tmp/dt && timeout 300 rocprofv3 --kernel-trace -d /tmp/dt -o dt -- python $R/tools/dav_timeline.py > /dev/null 2>&1)
python tools/dav_timeline_parse.py /tmp/dt > $OUT/dav_iter_timeline.txt 2>&1; cat $OUT/dav_iter_timeline.txt | tee -a $OUT/session.log
say "== configs[2]: internal coordinates / geodesic at 1024 atoms"
timeout 600 python tools/geodesic_bench.py --steps 3 --sella-steps 2 > $OUT/geodesic.log 2>&1; say "geodesic exit $?"
grep "^{" $OUT/geodesic.log | cut -c1-400 | tee -a $OUT/session.log
say "== ensemble: worker processes, host threads"
timeout 300 python tools/ensemble_probe.py 16 1 2 4 > $OUT/probe.log 2>&1; grep "pool\|cpus" $OUT/probe.log | tee -a $OUT/session.log
timeout 300 python tools/ensemble_threads.py 32 1 2 4 8 12 16 > $OUT/threads.log 2>&1; grep threads $OUT/threads.log | tee -a $OUT/session.log
SELLA_LIBRARY_SEARCH=0 timeout 300 python tools/ensemble_threads.py 32 1 4 8 > $OUT/threads_general.log 2>&1; grep threads $OUT/threads_general.log | sed "s/^/general driver: /" | tee -a $OUT/session.log
SELLA_DEBUG_TIMING=1 timeout 300 python tools/ens_profile.py > $OUT/ens_profile.log 2>&1; grep "seconds per member\|update_H n=768 k=[23]\|structured eigen" $OUT/ens_profile.log | tail -5 | tee -a $OUT/session.log
timeout 300 python tools/thread_scaling.py 768 40 > $OUT/thread_scaling.log 2>&1; cat $OUT/thread_scaling.log | tee -a $OUT/session.log
say "== rccl: 1 rank, then 2 ranks on the one GPU"
timeout 120 python tools/rccl_smoke.py > $OUT/rccl.log 2>&1; tail -1 $OUT/rccl.log | tee -a $OUT/session.log
NCCL_DEBUG=WARN timeout 300 python tools/rccl_two_ranks_one_gpu.py > $OUT/rccl2.log 2>&1; grep -i "duplicate\|two RCCL" $OUT/rccl2.log | cut -c1-400 | tee -a $OUT/session.log
