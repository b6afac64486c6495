timeout 900 python -m pytest tests/test_eigensolvers.py tests/test_big_gpu.py tests/test_fused_step.py tests/test_device_kernels.py -m gpu -x -q 2>&1 | tail -2
mkdir -p gpurun_out/symv; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/symv/profopt -o p -- python /root/repo/tools/opt_profile.py 3072 20 > /root/repo/gpurun_out/symv/profopt.log 2>&1
cd /root/repo
db=$(find gpurun_out/symv/profopt -name "*.db" | head -1); python tools/rocprof_summary.py $db gpurun_out/symv/profopt_stats.md "opt" > /dev/null; grep "lincomb\|lr_plan\|gemm_mfma" gpurun_out/symv/profopt_stats.md | cut -c1-120
rm -rf gpurun_out/symv/profopt
grep "ms per step" gpurun_out/symv/profopt.log | tail -2
