"""Eigensolver at sizes around its thresholds (64-reflector blocks from 2560, symmetric-aware matvec from 5120 trailing rows,
LDS tail, ragged panels): residual, orthogonality and eigenvalues against LAPACK."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sella_amd.device import Context  # noqa: E402

ctx = Context()
rng = np.random.RandomState(5)
worst = 0.0
for n in [int(a) for a in sys.argv[1:]] or [129, 2559, 2560, 2577, 4097, 5119, 5121, 5137, 5200, 6001]:
    A = rng.normal(size=(n, n))
    A = A + A.T
    t0 = time.perf_counter()
    w, V, Vt = ctx.eigh(ctx.upload(A))
    ctx.sync()
    dt = time.perf_counter() - t0
    Vn = V.numpy()
    wr = np.linalg.eigvalsh(A)
    scale = np.abs(wr).max()
    e_val = np.abs(w - wr).max() / scale
    e_res = np.abs(A @ Vn - Vn * w).max() / scale
    e_ort = np.abs(Vn.T @ Vn - np.eye(n)).max()
    ok = e_val <= 5e-13 * n ** 0.5 and e_res <= 5e-13 * n and e_ort <= 5e-13 * n
    worst = max(worst, e_res / n, e_ort / n)
    print(f'n={n}: {1e3 * dt:.1f} ms  eigenvalues {e_val:.2e}  residual {e_res:.2e}  orthogonality {e_ort:.2e}  {"ok" if ok else "FAILED"}', flush=True)
    V.free(); Vt.free()
    assert ok
print('all ok')
