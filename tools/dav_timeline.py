"""Kernel timeline of Davidson iterations (run under rocprofv3 --kernel-trace): per-iteration kernel list,
busy time and gaps."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import hessian_like  # noqa: E402
from sella_amd.device import Context  # noqa: E402

n = 3072
ctx = Context()
A, P, g = hessian_like(n, 0)
dA, dP = ctx.upload(A), ctx.upload(P)
w, V, Vt = ctx.eigh(dP)
for _ in range(3):
    ctx.davidson(dA, n, g, 0.1, method='jd0', maxiter=40, Pvecs=V, PvecsT=Vt, pevals=w)
