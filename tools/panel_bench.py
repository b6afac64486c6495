#!/usr/bin/env python3
"""Block product H·V (BASELINE configs[4]: n = 12288, k = 16) on one GPU: kernel time of the MFMA panel
kernel (matrix streamed once) against two passes of the 8-RHS row-panel matvec."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sella_amd.device import Context  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ctx = Context()
rng = np.random.RandomState(0)
H = rng.normal(size=(n, n))
X = rng.normal(size=(n, k))
dH = ctx.upload(H)
ref = H[:64] @ X
for mode, prow in ((1, 0), (1, 16), (1, 32), (1, 48), (1, 64), (0, 0)):
    ctx.set_option('panel_mfma', mode)
    ctx.set_option('panel_rows', prow)
    Y = ctx.symm_mm(dH, X)
    err = float(np.abs(Y[:64] - ref).max())
    ctx.prof_reset()
    ctx.prof_enable(True)
    for _ in range(10):
        ctx.symm_mm(dH, X)
    ctx.prof_enable(False)
    p = ctx.prof_get(0)
    us = 1e3 * p['ms'] / 10
    print(json.dumps(dict(op='H.V block product', n=n, k=k, kernel=(f'panel16_mfma<{prow} rows>' if prow else 'panel16_mfma<rows by size>') if mode else 'gemv_rows x2',
                          launches_per_product=p['launches'] / 10, us_per_product=round(us, 1),
                          matrix_GBps=round(8.0 * n * n / (us * 1e-6) / 1e9, 1),
                          tflops=round(2.0 * n * n * k / (us * 1e-6) / 1e12, 2), max_err=err)), flush=True)
