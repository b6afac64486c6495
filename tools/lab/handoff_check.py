#!/usr/bin/env python3
"""Hand-over launches of the tridiagonalisation (option eigh_handoff) on the device: timing against the two-launch chain,
LAPACK check, and run-to-run bit identity (a stale read through a non-coherent L2 would show as a difference).
Usage: tools/handoff_check.py [n] [repetitions]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sella_amd.device import Context  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = Context(0)
rng = np.random.RandomState(0)
A = rng.normal(size=(n, n))
A = A + A.T
dA = ctx.upload(A)
wr = np.linalg.eigvalsh(A)
for label, opts in (('two launches per column', dict(eigh_handoff=0)), ('hand-over, auto rows', dict(eigh_handoff=1)),
                    ('hand-over, 4 rows', dict(eigh_handoff=1, eigh_handoff_rows=4)),
                    ('hand-over, 8 rows', dict(eigh_handoff=1, eigh_handoff_rows=8)),
                    ('hand-over everywhere (no one-launch chain below 1024)', dict(eigh_handoff=1, eigh_handoff_rows=0, eigh_upd_max=0)),
                    ('two launches everywhere', dict(eigh_handoff=0, eigh_upd_max=0))):
    for k, v in {**dict(eigh_handoff_rows=0, eigh_upd_max=1024), **opts}.items():
        ctx.set_option(k, v)
    first = None
    best = 1e9
    same = True
    for r in range(reps):
        t0 = time.perf_counter()
        w, V, Vt = ctx.eigh(dA)
        ctx.sync()
        best = min(best, time.perf_counter() - t0)
        Vn = V.numpy()
        if first is None:
            first = (w.copy(), Vn.copy())
        else:
            same = same and np.array_equal(w, first[0]) and np.array_equal(Vn, first[1])
        V.free()
        Vt.free()
    w, Vn = first
    print('%-56s: %.2f ms, eig err %.1e resid %.1e orth %.1e, %d runs bit-identical: %s' % (
        label, 1e3 * best, np.abs(w - wr).max(), np.abs(A @ Vn - Vn * w).max(), np.abs(Vn.T @ Vn - np.eye(n)).max(), reps, same), flush=True)
