cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 3072 12288; do
(cd /tmp && EIGH_TWO_STAGE=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$n -o ts -- python $R/tools/eigh_only.py $n 2 > /tmp/rocprof_$n.log 2>&1)
db=$(find /tmp/prof_$n -name "*.db" | head -1)
python tools/rocprof_summary.py $db gpurun_out/ts_v2_stats_$n.md "two-stage eigh n=$n, 2 calls" > /dev/null
head -24 gpurun_out/ts_v2_stats_$n.md
done
