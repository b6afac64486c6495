#!/usr/bin/env python3
"""fp64 GEMM rates of the library's MFMA kernels at the shapes of a blocked back-transformation
(X (n x m) times Y^T (m x K), (n x K) times Y (K x m))."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sella_amd.device import Context  # noqa: E402

ctx = Context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
rng = np.random.RandomState(0)
X = ctx.upload(rng.normal(size=(n, n)))
for K in (32, 64, 128, 256, 512):
    Y = ctx.upload(rng.normal(size=(K, n)))
    Yt = ctx.upload(rng.normal(size=(n, K)))
    M = ctx.zeros(n, K)
    for name, fn, flops in (
            ('M = X Yt      (NN, N=K)', lambda: ctx.gemm(X, Yt, M), 2.0 * n * n * K),
            ('M = X Y^T     (NT, N=K)', lambda: ctx.gemm(X, Y, M, transB=True), 2.0 * n * n * K),
            ('X -= M Y      (NN, K=K)', lambda: ctx.gemm(M, Y, X, alpha=-1e-9, beta=1.0), 2.0 * n * n * K)):
        fn()
        ctx.sync()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            fn()
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps
        print(f'n={n} K={K:4d} {name}: {1e3 * dt:7.3f} ms  {flops / dt / 1e12:6.2f} TFLOP/s', flush=True)
