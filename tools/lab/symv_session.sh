set -x
mkdir -p gpurun_out/symv
O=gpurun_out/symv
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
for n in 12288 8192 6144 4096; do echo "== $n defaults"; timeout 300 python tools/eigh_only.py $n 3 2>&1 | tail -2; done
