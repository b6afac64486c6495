set -x
mkdir -p gpurun_out/symv
O=gpurun_out/symv
timeout 600 python -m pytest tests/test_eigh.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
for thr in 0 4096 2048 6144; do echo "== 12288 symv_min=$thr"; EIGH_SYMV_MIN=$thr timeout 300 python tools/eigh_only.py 12288 3 2>&1 | tail -2; done
for thr in 0 2048 1536 1024; do echo "== 3072 symv_min=$thr"; EIGH_SYMV_MIN=$thr timeout 300 python tools/eigh_only.py 3072 5 2>&1 | tail -2; done
cd /tmp && export TMPDIR=/tmp
for thr in 1024; do
 EIGH_SYMV_MIN=$thr timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof3072 -o p -- python /root/repo/tools/eigh_only.py 3072 2 > /root/repo/$O/prof3072.log 2>&1
done
EIGH_SYMV_MIN=4096 timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof12288 -o p -- python /root/repo/tools/eigh_only.py 12288 1 > /root/repo/$O/prof12288.log 2>&1
cd /root/repo
for d in prof3072 prof12288; do db=$(find $O/$d -name "*.db" | head -1); python tools/rocprof_summary.py $db $O/${d}_stats.md "$d" > /dev/null; head -14 $O/${d}_stats.md; python tools/trd_by_m.py $db $([ $d = prof12288 ] && echo 1024 || echo 256) > $O/${d}_by_m.txt 2>&1; cat $O/${d}_by_m.txt; rm -rf $O/$d; done
