cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python tools/emt_slab_lib.py 40 2>&1 | tail -1
python tools/emt_slab_lib.py 40 2>&1 | tail -1
timeout 300 python tools/emt_ensemble.py 64 t8 2>&1 | tail -2
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_emt -o emt -- python $R/tools/emt_slab_lib.py 12 > /tmp/rocprof_emt.log 2>&1)
db=$(find /tmp/prof_emt -name "*.db" | head -1)
python tools/rocprof_summary.py $db /tmp/emt_stats.md "emt" > /dev/null
grep "emt_" /tmp/emt_stats.md
timeout 600 python -m pytest tests/test_library_calculator.py tests/test_pes_oracle.py tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -2
