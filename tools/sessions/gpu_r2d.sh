#!/bin/bash
OUT=gpurun_out/r2d
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== eigh timing: rank2k stream on / off" | tee $OUT/session.log
cat > /tmp/eigh_ab.py <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from sella_amd.device import Context
from bench import hessian_like
ctx = Context(0)
for n in (3072,):
    A, P, g = hessian_like(n, 0)
    dP = ctx.upload(P)
    ref = None
    for rep in range(3):
        for opt in (1, 0):
            ctx.set_option('rank2k_stream', opt)
            w, V, Vt = ctx.eigh(dP); V.free(); Vt.free()
            ctx.sync(); t0 = time.perf_counter()
            for _ in range(3):
                w, V, Vt = ctx.eigh(dP)
                if _ < 2: V.free(); Vt.free()
            ctx.sync(); dt = (time.perf_counter() - t0) / 3
            Vn = V.numpy(); V.free(); Vt.free()
            res = np.abs(P @ Vn - Vn * w).max(); orth = np.abs(Vn.T @ Vn - np.eye(n)).max()
            print(f'n={n} rank2k_stream={opt}: {1e3*dt:.2f} ms  residual {res:.2e} orth {orth:.2e} evdiff {np.abs(w-np.linalg.eigvalsh(P)).max():.2e}', flush=True)
PY
timeout 600 python /tmp/eigh_ab.py > $OUT/eigh_ab.log 2>&1; tail -8 $OUT/eigh_ab.log | tee -a $OUT/session.log
echo "== rocprof eigh" | tee -a $OUT/session.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_eigh -o eigh -- python $R/tools/eigh_only.py 3072 4 > $R/$OUT/rocprof_eigh.log 2>&1); echo "rocprof exit $?" | tee -a $OUT/session.log
DBE=$(find $OUT/prof_eigh -name "*.db" | head -1)
python tools/rocprof_summary.py $DBE $OUT/eigh_kernel_stats.md "tools/eigh_only.py 3072 4 (rocprofv3 --kernel-trace --stats)" > /dev/null
head -16 $OUT/eigh_kernel_stats.md | tee -a $OUT/session.log
rm -rf $OUT/prof_eigh
echo "== eigh 12288 timing" | tee -a $OUT/session.log
timeout 300 python tools/eigh_big_check.py 12288 > $OUT/eigh_12288.log 2>&1; tail -2 $OUT/eigh_12288.log | tee -a $OUT/session.log
echo "== converged tests" | tee -a $OUT/session.log
SELLA_DEBUG_TIMING=1 timeout 900 python -m pytest tests/test_big_gpu.py -m gpu -q -x -k "converged_eigenpair" --durations=3 -s > $OUT/pytest_conv.log 2>&1; echo "pytest exit $?" | tee -a $OUT/session.log; grep "^davidson\|passed\|failed" $OUT/pytest_conv.log | tail -6 | tee -a $OUT/session.log
echo "== ensemble threads" | tee -a $OUT/session.log
for T in 1 2 4 8; do timeout 300 python bench.py --no-cpu-baseline --block-iters 0 --emt-steps 0 --steps 2 --warmup 1 --converged-n 0 --ensemble-threads $T > $OUT/bench_t$T.log 2>&1; grep -o '"ensemble": {[^}]*}' $OUT/bench_t$T.log | tee -a $OUT/session.log; done
