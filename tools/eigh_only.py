#!/usr/bin/env python3
"""Profiling target: a few device eigh calls at n = 3072 (run under rocprofv3 --kernel-trace)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.abspath(os.environ['SELLA_AB_ROOT']) if os.environ.get('SELLA_AB_ROOT')      # tools/ab_build.sh
                else os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sella_amd.device import Context  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ctx = Context(0)
if os.environ.get('EIGH_NB'):
    ctx.set_option('eigh_nb', int(os.environ['EIGH_NB']))
if os.environ.get('EIGH_WY_ROWS'):
    ctx.set_option('eigh_wy_rows', int(os.environ['EIGH_WY_ROWS']))
if os.environ.get('EIGH_WY_WAVES'):
    ctx.set_option('eigh_wy_waves', int(os.environ['EIGH_WY_WAVES']))
if os.environ.get('EIGH_SYMV_MIN'):
    ctx.set_option('eigh_symv_min', int(os.environ['EIGH_SYMV_MIN']))
if os.environ.get('EIGH_SYMV_TR'):
    ctx.set_option('eigh_symv_tr', int(os.environ['EIGH_SYMV_TR']))
if os.environ.get('EIGH_WY64_MIN'):
    ctx.set_option('eigh_wy_nb64_min', int(os.environ['EIGH_WY64_MIN']))
if os.environ.get('EIGH_TAIL_LDS'):
    ctx.set_option('eigh_tail_lds', int(os.environ['EIGH_TAIL_LDS']))
if os.environ.get('EIGH_LEAF'):
    ctx.set_option('eigh_leaf', int(os.environ['EIGH_LEAF']))
for kv in filter(None, os.environ.get('EIGH_OPTS', '').split(',')):     # generic: EIGH_OPTS=key=value,key=value
    key, value = kv.split('=')
    ctx.set_option(key, int(value))
rng = np.random.RandomState(0)
kind = sys.argv[3] if len(sys.argv) > 3 else 'dense'
if kind == 'dense':
    A = rng.normal(size=(n, n))
    A = A + A.T
else:                                   # what an approximate Hessian looks like early on: lam0*I + low rank
    u = rng.normal(size=(n, 6))
    A = 1.7 * np.eye(n) + u[:, :4] @ u[:, :4].T - u[:, 4:] @ u[:, 4:].T
dA = ctx.upload(A)
for r in range(reps):
    t0 = time.perf_counter()
    w, V, Vt = ctx.eigh(dA)
    ctx.sync()
    print(f'eigh n={n}: {1e3 * (time.perf_counter() - t0):.2f} ms', flush=True)
    if r == reps - 1 and os.environ.get('EIGH_CHECK'):
        Vn = V.numpy()
        wr = np.linalg.eigvalsh(A)
        print('check: eig err %.2e, resid %.2e, orth %.2e' % (np.abs(w - wr).max(), np.abs(A @ Vn - Vn * w).max(), np.abs(Vn.T @ Vn - np.eye(n)).max()), flush=True)
    V.free()
    Vt.free()
