"""What serialises host threads?  T threads, each with a device context (stream) of its own on the same GPU, each
running the same loop of C-ABI calls with NO Python between them but the loop itself:
   mode 'update': sella_update_h_eig at n (quasi-Newton update + carried eigendecomposition, ~10 host syncs per call)
   mode 'gemv'  : sella_symm_mm (one n x n matvec, one sync)
   mode 'sync'  : sella_ctx_sync on an idle stream (pure runtime call)
Prints calls/s in total for T = 1, 2, 4, 8.   usage: thread_scaling.py [n] [reps]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sella_amd.device import Context  # noqa: E402


def worker(mode, n, reps, barrier, out, slot):
    ctx = Context()
    rng = np.random.RandomState(slot)
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    lam = np.exp(rng.uniform(np.log(0.05), np.log(50.0), n))
    lam[0] = -1.0
    A = (Q * lam) @ Q.T
    A = 0.5 * (A + A.T)
    dB = ctx.upload(A)
    w, V, Vt = ctx.eigh(dB)
    S = rng.normal(size=(reps + 4, n, 1)) * 0.05
    x = rng.normal(size=n)

    def call(i):
        nonlocal w
        if mode == 'update':
            w, _ = ctx.update_h_eig(dB, S[i], A @ S[i] + 0.01 * S[i], w, V, Vt, method='TS-BFGS', symm=2, max_rank=8)
        elif mode == 'gemv':
            ctx.symm_mm(dB, x)
        else:
            ctx.sync()
    for i in range(3):
        call(i)
    barrier.wait()
    t = time.perf_counter()
    for i in range(reps):
        call(3 + i if mode == 'update' else 0)
    out[slot] = time.perf_counter() - t
    barrier.wait()
    ctx.close()


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 768
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    for mode in ('update', 'gemv', 'sync'):
        for T in (1, 2, 4, 8):
            out = [0.0] * T
            bar = threading.Barrier(T)
            ths = [threading.Thread(target=worker, args=(mode, n, reps * (1 if mode == 'update' else 10), bar, out, k))
                   for k in range(T)]
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            r = reps * (1 if mode == 'update' else 10)
            print('%-6s n=%d threads=%d: %8.0f calls/s in total (%.1f us per call per thread)'
                  % (mode, n, T, T * r / max(out), 1e6 * max(out) / r), flush=True)
