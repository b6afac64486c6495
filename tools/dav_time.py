import sys, os, numpy as np, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sella_amd.device import Context
from bench import hessian_like
ctx = Context(0); n=3072
A,P,g = hessian_like(n,0); dA=ctx.upload(A); dP=ctx.upload(P)
w,V,Vt = ctx.eigh(dP)
for gamma, mi in ((1e-32,40),(1e-32,40),(1e-32,20),(1e-32,80)):
    t0=time.perf_counter(); out = ctx.davidson(dA,n,g,gamma,method='jd0',maxiter=mi,Pvecs=V,PvecsT=Vt,pevals=w); ctx.sync(); print(mi, out[1].shape[1], 1e3*(time.perf_counter()-t0),'ms')
