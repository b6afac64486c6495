#!/usr/bin/env python3
"""eigh timing vs panel width / leaf size (tuning aid)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sella_amd.device import Context  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
ctx = Context(0)
rng = np.random.RandomState(0)
A = rng.normal(size=(n, n))
A = A + A.T
dA = ctx.upload(A)
for nb in (8, 16, 24, 32, 48, 64):
    ctx.set_option('eigh_nb', nb)
    best = 1e9
    for r in range(3):
        t0 = time.perf_counter()
        w, V, Vt = ctx.eigh(dA)
        ctx.sync()
        best = min(best, time.perf_counter() - t0)
        V.free()
        Vt.free()
    print(f'n={n} nb={nb}: {1e3 * best:.2f} ms', flush=True)
