import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import hessian_like
from sella_amd.device import Context
ctx = Context()
n = 3072
A, P, g = hessian_like(n, 0)
dA, dP = ctx.upload(A), ctx.upload(P)
w, V, Vt = ctx.eigh(dP)
for _ in range(2):
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        out = ctx.davidson(dA, n, g, 0.1, method='jd0', maxiter=40, Pvecs=V, PvecsT=Vt, pevals=w)
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    print(f'gamma 0.1: {out[1].shape[1]} vectors, {1e3 * dt:.3f} ms per call, {1e6 * dt / out[1].shape[1]:.1f} us per vector', flush=True)
