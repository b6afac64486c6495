"""Davidson loop time at 3N = 3072 with P's eigendecomposition resident (options on/off comparison)."""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.environ['SELLA_AB_ROOT']) if os.environ.get('SELLA_AB_ROOT')      # tools/ab_build.sh
                else os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(1, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import hessian_like  # noqa: E402
from sella_amd.device import Context  # noqa: E402

ctx = Context()
n = 3072
A, P, g = hessian_like(n, 0)
dA, dP = ctx.upload(A), ctx.upload(P)
w, V, Vt = ctx.eigh(dP)
for opt, zc, poll in ((0, 0, 0), (1, 0, 0), (1, 1, 0), (1, 0, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)):
    ctx.set_option('host_scalars', opt)
    ctx.set_option('dav_zero_copy', zc)
    ctx.set_option('dav_poll', poll)
    for mi in (40,):
        ctx.davidson(dA, n, g, 1e-32, method='jd0', maxiter=mi, Pvecs=V, PvecsT=Vt, pevals=w)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            out = ctx.davidson(dA, n, g, 1e-32, method='jd0', maxiter=mi, Pvecs=V, PvecsT=Vt, pevals=w)
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps
        print(f'host_scalars={opt} dav_zero_copy={zc} dav_poll={poll} maxiter={mi}: {out[1].shape[1]} vectors, {1e3 * dt:.2f} ms per call, '
              f'{1e6 * dt / out[1].shape[1]:.1f} us per vector', flush=True)
