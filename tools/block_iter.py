#!/usr/bin/env python3
"""Block Davidson iterations at BASELINE configs[4] size on one GPU (dense symmetric pseudo-random operator, diagonal
preconditioner, fixed iteration count): ms per block iteration of the pipelined driver (option bd_pipeline 1) and of
the general loop (0).   usage: block_iter.py [n] [iterations] [flags ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sella_amd import device as _dev  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ctx = _dev.get_context()
rng = np.random.RandomState(0)
H = rng.standard_normal((n, n)) * 0.01
H = H + H.T
H[np.arange(n), np.arange(n)] += 0.5 + 50.0 * (np.arange(n) / n) ** 2
dH = ctx.upload(H)
diag = np.ascontiguousarray(H.diagonal())
del H
flags = [int(a) for a in sys.argv[3:]] or [1, 0, 1]
for flag in flags:
    ctx.set_option('bd_pipeline', flag)
    ctx.davidson_block(dH, n, 16, block=16, tol=1e-10, maxiter=2, maxvec=48, diag=diag)
    ctx.sync()
    t = time.perf_counter()
    out = ctx.davidson_block(dH, n, 16, block=16, tol=1e-10, maxiter=iters, maxvec=48, diag=diag)
    ctx.sync()
    dt = time.perf_counter() - t
    print('bd_pipeline=%d: %d iterations, %.3f ms per block iteration, lowest Ritz %.12f'
          % (flag, out['niter'], 1e3 * dt / max(1, out['niter']), out['lams'][0]), flush=True)
if os.environ.get('BLOCK_ITER_CONVERGE'):
    for flag in (1, 0):
        ctx.set_option('bd_pipeline', flag)
        t = time.perf_counter()
        out = ctx.davidson_block(dH, n, 16, block=16, tol=1e-9, maxiter=300, diag=diag)
        dt = time.perf_counter() - t
        print('bd_pipeline=%d: converged run, tol 1e-9: %d of 16 pairs in %d iterations, %d matrix-vector products, %.1f ms, lowest %.12f'
              % (flag, out['nconv'], out['niter'], out['nmatvec'], 1e3 * dt, out['lams'][0]), flush=True)
