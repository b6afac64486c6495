"""One eigh at a large size (default 12288 = BASELINE configs[4]) with a sampled residual / orthogonality check."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sella_amd.device import Context  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
ctx = Context()
rng = np.random.RandomState(0)
A = rng.normal(size=(n, n))
A = A + A.T
dA = ctx.upload(A)
t0 = time.perf_counter()
w, V, Vt = ctx.eigh(dA)
ctx.sync()
dt = time.perf_counter() - t0
Vt_n = Vt.numpy()
idx = np.r_[0:4, n // 2:n // 2 + 4, n - 4:n]
Vs = Vt_n[idx].T
res = np.abs(A @ Vs - Vs * w[idx]).max()
orth = np.abs(Vt_n[idx] @ Vt_n.T - np.eye(n)[idx]).max()
print(f'eigh n={n}: {dt:.3f} s; sampled residual {res:.2e} (|A| ~ {np.abs(w).max():.1f}), orthogonality {orth:.2e}, '
      f'ascending {bool(np.all(np.diff(w) >= 0))}')
