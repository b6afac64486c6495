import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import hessian_like
from sella_amd.device import Context
import numpy as np
ctx = Context()
for n in (3072, 6144):
    A, P, g = hessian_like(n, 0)
    dA = ctx.upload(A)
    x = np.random.RandomState(0).normal(size=(n, 2))
    for rw in (1, 2, 4, 2):
        ctx.set_option('gemv_rw', rw)
        ctx.symm_mm(dA, x)
        ctx.prof_reset(); ctx.prof_enable(True)
        for _ in range(20):
            ctx.symm_mm(dA, x)
        ctx.prof_enable(False)
        p = ctx.prof_get(0)
        us = 1e3 * p['ms'] / max(1, p['launches'])
        print(f'n={n} gemv_rw={rw}: {us:.2f} us per pass (2 rhs), {8.0*n*n/us/1e3:.0f} GB/s, launches {p["launches"]}', flush=True)
