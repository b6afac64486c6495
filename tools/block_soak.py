#!/usr/bin/env python3
"""Soak of sella_davidson_block on the device: random operators, sizes, nev / block, start blocks and preconditioners; every run
must converge to the eigenvalues of numpy.linalg.eigvalsh within the iteration limit.  usage: block_soak.py [runs] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sella_amd import device as _dev  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = _dev.get_context()
bad, its, t0 = [], [], time.perf_counter()
for r in range(runs):
    n = int(rng.choice([200, 300, 517, 1024, 2100]))
    nev = int(rng.choice([1, 3, 5, 8, 16, 20]))
    block = int(os.environ.get('SOAK_BLOCK', 0)) or int(rng.choice([4, 8, 16]))
    kind = rng.choice(['diag_unit', 'diag_random', 'eigen', 'none_random'])
    N = rng.normal(size=(n, n))
    A = np.diag(np.sort(rng.uniform(0.5, 0.5 * n, size=n))) + rng.choice([0.005, 0.02, 0.05]) * (N + N.T)
    w = np.linalg.eigvalsh(A)
    only = os.environ.get('SOAK_ONLY')
    if only is not None and int(only) != r:
        # (draw what the run would have drawn, so that the generator stays in step)
        if kind.endswith('random'):
            rng.normal(size=(n, int(rng.choice([min(16, max(nev, 2)), 16]))))
        if kind == 'eigen':
            rng.normal(size=n)
        rng.choice([1e-7, 1e-9]); rng.randint(2)
        continue
    dA = ctx.upload(A)
    kw = {}
    if kind.startswith('diag'):
        kw['diag'] = np.diag(A).copy()
    if kind.endswith('random'):
        kw['V0'] = rng.normal(size=(n, int(rng.choice([min(16, max(nev, 2)), 16]))))
    if kind == 'eigen':
        P = A + 0.02 * np.diag(rng.normal(size=n))
        wq, Q, Qt = ctx.eigh(ctx.upload(P))
        kw.update(Pvecs=Q, PvecsT=Qt, pevals=wq)
    tol = float(rng.choice([1e-7, 1e-9]))
    early = int(rng.randint(2))
    ctx.set_option('bd_early_matvec', early)
    out = ctx.davidson_block(dA, n, nev, block=block, tol=tol, maxiter=1500, **kw)
    err = np.abs(out['lams'] - w[:nev]).max()
    ok = out['nconv'] == nev and err < 1e-6
    its.append(out['niter'])
    if ok and out['niter'] > 400:
        print('slow', (r, n, nev, block, str(kind), tol, early, out['niter']), flush=True)
    if not ok:
        bad.append((r, n, nev, block, kind, tol, early, out['niter'], out['nconv'], err))
        print('FAILED', bad[-1], flush=True)
ctx.set_option('bd_early_matvec', 1)
print('%d runs, %d failed, iterations median %d max %d, %.1f s' % (runs, len(bad), int(np.median(its)), max(its), time.perf_counter() - t0))
