import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import hessian_like
from sella_amd.device import Context
ctx = Context()
n = 3072
A, P, g = hessian_like(n, 0)
dA, dP = ctx.upload(A), ctx.upload(P)
w, V, Vt = ctx.eigh(dP)
def run(reps):
    for _ in range(reps):
        out = ctx.davidson(dA, n, g, 0.1, method='jd0', maxiter=40, Pvecs=V, PvecsT=Vt, pevals=w)
    ctx.sync()
run(5)
pr = cProfile.Profile(); pr.enable(); run(100); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(12)
