import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
from sella_amd.device import Context
ctx = Context(0)
n = 5137
rng = np.random.RandomState(5)
A = rng.normal(size=(n, n)); A = A + A.T
wr = np.linalg.eigvalsh(A)
for opts in ({}, {'rank2k_pair': 0}, {'eigh_gemv_flat': 0}, {'eigh_dc_pipeline': 0}):
    for k, v in opts.items(): ctx.set_option(k, v)
    w, V, Vt = ctx.eigh(ctx.upload(A))
    Vn = V.numpy()
    print(opts, 'eig err %.2e (tol %.2e) resid %.2e orth %.2e' % (np.abs(w - wr).max(), 5e-13 * n ** 0.5 * np.abs(wr).max(), np.abs(A @ Vn - Vn * w).max(), np.abs(Vn.T @ Vn - np.eye(n)).max()), flush=True)
    for k, v in opts.items(): ctx.set_option(k, 1)
