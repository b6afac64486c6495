timeout 600 python -m pytest tests/test_step_solve.py tests/test_configs_gpu.py tests/test_fused_step.py -m gpu -x -q 2>&1 | tail -2
mkdir -p gpurun_out/symv; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/symv/profemt -o p -- python /root/repo/tools/emt_slab_opt.py > /root/repo/gpurun_out/symv/profemt.log 2>&1
cd /root/repo
db=$(find gpurun_out/symv/profemt -name "*.db" | head -1); python tools/rocprof_summary.py $db gpurun_out/symv/profemt_stats.md "emt" > /dev/null; head -22 gpurun_out/symv/profemt_stats.md | cut -c1-120
rm -rf gpurun_out/symv/profemt
grep "per optimizer step" gpurun_out/symv/profemt.log
