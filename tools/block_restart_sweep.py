#!/usr/bin/env python3
"""Block Davidson at 3N = 12288 (operator of tools/block_iter.py, diagonal preconditioner, nev = block = 16, tol 1e-9): iterations
and time to convergence by basis limit (maxvec) — and, through SELLA_BD_KEEP, by the number of Ritz vectors a thick restart keeps."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sella_amd import device as _dev  # noqa: E402

n = 12288
ctx = _dev.get_context()
rng = np.random.RandomState(0)
H = rng.standard_normal((n, n)) * 0.01
H = H + H.T
H[np.arange(n), np.arange(n)] += 0.5 + 50.0 * (np.arange(n) / n) ** 2
dH = ctx.upload(H)
diag = np.ascontiguousarray(H.diagonal())
del H
for maxvec in [int(a) for a in sys.argv[1:]] or [48, 64, 96]:
    ctx.davidson_block(dH, n, 16, block=16, tol=1e-9, maxiter=2, maxvec=maxvec, diag=diag)
    ctx.sync()
    t = time.perf_counter()
    out = ctx.davidson_block(dH, n, 16, block=16, tol=1e-9, maxiter=600, maxvec=maxvec, diag=diag)
    ctx.sync()
    dt = time.perf_counter() - t
    print('maxvec %d (keep %s): %d of 16 pairs in %d iterations, %d products, %.1f ms (%.3f ms per iteration), lowest %.10f'
          % (maxvec, os.environ.get('SELLA_BD_KEEP', 'default'), out['nconv'], out['niter'], out['nmatvec'], 1e3 * dt,
             1e3 * dt / max(1, out['niter']), out['lams'][0]), flush=True)
