"""Soak of the pipelined divide & conquer (one wait per level, uploads from staging arrays that the host refills): many
sizes, every result against the two-wait loop bit for bit and against LAPACK, with other host threads launching in between."""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sella_amd.device import Context  # noqa: E402

ctx = Context()
rng = np.random.RandomState(7)
stop = False


def noise():
    c2 = Context()
    B = c2.upload(np.random.RandomState(1).normal(size=(1500, 1500)))
    while not stop:
        c2.eigh(B, vectors=False)


th = [threading.Thread(target=noise) for _ in range(2)] if '--noise' in sys.argv else []
for t in th:
    t.start()
bad = 0
sizes = [97, 200, 515, 1030, 1536, 2049, 2577, 3072, 3500, 4097, 5137, 6001]
for rep in range(2):
    for n in sizes:
        A = rng.normal(size=(n, n))
        A = A + A.T
        dA = ctx.upload(A)
        out = []
        for pipe in (1, 0, 1):
            ctx.set_option('eigh_dc_pipeline', pipe)
            w, V, Vt = ctx.eigh(dA)
            out.append((np.array(w), V.numpy()))
            V.free(); Vt.free()
        same = all(np.array_equal(out[0][k], out[j][k]) for j in (1, 2) for k in (0, 1))
        wr = np.linalg.eigvalsh(A)
        err = np.abs(out[0][0] - wr).max() / np.abs(wr).max()
        ok = same and err < 5e-13 * n ** 0.5
        bad += not ok
        print(f'n={n}: pipelined == two-wait loop: {same}, eigenvalue error {err:.1e} {"ok" if ok else "FAILED"}', flush=True)
        dA.free()
stop = True
for t in th:
    t.join()
print('all ok' if not bad else f'{bad} FAILED')
