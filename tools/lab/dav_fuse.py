"""Davidson loop at 3N = 3072, P's eigendecomposition resident: chain options on / off (dav_fuse_scale, dav_zero_copy)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import hessian_like  # noqa: E402
from sella_amd.device import Context  # noqa: E402

ctx = Context()
n = 3072
A, P, g = hessian_like(n, 0)
dA, dP = ctx.upload(A), ctx.upload(P)
w, V, Vt = ctx.eigh(dP)
ref = None
for fuse, zc in ((0, 0), (1, 0), (0, 1), (1, 1), (0, 0), (1, 1)):
    ctx.set_option('dav_fuse_scale', fuse)
    ctx.set_option('dav_zero_copy', zc)
    ctx.davidson(dA, n, g, 1e-32, method='jd0', maxiter=40, Pvecs=V, PvecsT=Vt, pevals=w)
    t0 = time.perf_counter()
    reps = 8
    for _ in range(reps):
        out = ctx.davidson(dA, n, g, 1e-32, method='jd0', maxiter=40, Pvecs=V, PvecsT=Vt, pevals=w)
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    lam = float(out[0][0])
    ref = lam if ref is None else ref
    print(f'fuse={fuse} zero_copy={zc}: {out[1].shape[1]} vectors, {1e3 * dt:.3f} ms per call, '
          f'{1e6 * dt / out[1].shape[1]:.1f} us per vector, lam0 - first = {lam - ref:.2e}', flush=True)
