#!/usr/bin/env python3
"""Does multi-threaded use of the runtime slow down later single-threaded calls of the same process?  (bench.py: the legs
behind the threaded 'concurrent problems' leg ran ~0.16 ms per call slower than the same calls in a fresh process.)"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import hessian_like  # noqa: E402
from sella_amd import device as _devm  # noqa: E402
from sella_amd.device import Context  # noqa: E402

ctx = Context()
n = 3072
A, P, g = hessian_like(n, 0)
dA, dP = ctx.upload(A), ctx.upload(P)
w, V, Vt = ctx.eigh(dP)


def measure(tag):
    for _ in range(3):
        ctx.davidson(dA, n, g, 0.1, method='jd0', maxiter=40, Pvecs=V, PvecsT=Vt, pevals=w)
    ctx.sync()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        out = ctx.davidson(dA, n, g, 0.1, method='jd0', maxiter=40, Pvecs=V, PvecsT=Vt, pevals=w)
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    print(f'{tag}: {out[1].shape[1]} vectors, {1e3 * dt:.3f} ms per call, {1e6 * dt / out[1].shape[1]:.1f} us per vector', flush=True)


measure('fresh process')
mode = sys.argv[1] if len(sys.argv) > 1 else 'contexts'


def work():
    if mode == 'idle':
        time.sleep(0.05)
        return
    cx = Context()
    _devm.use_context(cx)
    try:
        if mode == 'contexts':
            a_, p_ = cx.upload(A), cx.upload(P)
            w_, V_, Vt_ = cx.eigh(p_)
            cx.davidson(a_, n, g, 0.1, method='jd0', maxiter=40, Pvecs=V_, PvecsT=Vt_, pevals=w_)
            cx.sync()
    finally:
        _devm.use_context(None)
        cx.close()


ths = [threading.Thread(target=work) for _ in range(4)]
for t in ths:
    t.start()
for t in ths:
    t.join()
measure('after 4 threads (%s)' % mode)
time.sleep(1.0)
measure('one second later')
