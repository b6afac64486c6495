#!/usr/bin/env python3
"""The block-Davidson leg of bench.py alone (its integer-hash operator at 3N = 12288, diagonal preconditioner, 12 iterations)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sella_amd import device as _dev  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ctx = _dev.get_context()
ii = np.arange(n, dtype=np.int64)[:, None]
jj = np.arange(n, dtype=np.int64)[None, :]
a_, b_ = np.minimum(ii, jj), np.maximum(ii, jj)
hsh = ((a_ * 73856093) ^ (b_ * 19349663) ^ 0x5bd1e995) & 0xFFFFF
H = (hsh.astype(np.float64) / 0xFFFFF - 0.5) * 0.02
H[np.arange(n), np.arange(n)] += 0.5 + 50.0 * (np.arange(n) / n) ** 2
dH = ctx.upload(H)
diag = np.ascontiguousarray(H.diagonal())
del H, hsh, a_, b_
for flag in ((1,) if os.environ.get("SELLA_BD_CHECK") else (1, 0, 1)):
    ctx.set_option('bd_pipeline', flag)
    ctx.davidson_block(dH, n, 16, block=16, tol=1e-10, maxiter=2, maxvec=48, diag=diag)
    ctx.sync()
    t = time.perf_counter()
    out = ctx.davidson_block(dH, n, 16, block=16, tol=1e-10, maxiter=iters, maxvec=48, diag=diag)
    ctx.sync()
    dt = time.perf_counter() - t
    print('bd_pipeline=%d: %d iterations, %.3f ms per block iteration, %d products, lowest Ritz %.12f, residuals %s'
          % (flag, out['niter'], 1e3 * dt / max(1, out['niter']), out['nmatvec'], out['lams'][0],
             np.array2string(np.asarray(out['res']), precision=1)), flush=True)
