#!/usr/bin/env python3
"""Kernel-level timings on the MI355X (run through gpurun): row-panel matvec variants, GEMM
variants, eigh, update, Davidson.  Prints one JSON object per measurement."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sella_amd.device import Context  # noqa: E402

ctx = Context(0)
print(json.dumps(dict(device=ctx.name)), flush=True)
rng = np.random.RandomState(0)


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps


def prof_kind(fn, kind, reps=20):
    fn()
    ctx.prof_reset()
    ctx.prof_enable(True)
    for _ in range(reps):
        fn()
    ctx.prof_enable(False)
    return ctx.prof_get(kind)


sizes = [int(s) for s in os.environ.get('MB_SIZES', '768,3072,12288').split(',')]
for n in sizes:
    A = rng.normal(size=(n, n))
    dA = ctx.upload(A)
    for k in (1, 2, 4, 8):
        X = rng.normal(size=(n, k))
        for rw in (1, 2, 4):
            ctx.set_option('gemv_rw', rw)
            p = prof_kind(lambda: ctx.symm_mm(dA, X), 0)
            us = 1e3 * p['ms'] / p['launches']
            print(json.dumps(dict(op='gemv_rows', n=n, nrhs=k, rw=rw, us=round(us, 2),
                                  GBs=round(8.0 * n * n / us / 1e3, 1))), flush=True)
    ctx.set_option('gemv_rw', 0)
    X = rng.normal(size=(n, 2))
    t = timeit(lambda: ctx.tmatmul(dA, X), reps=10)
    print(json.dumps(dict(op='gemv_cols(host-inclusive)', n=n, nrhs=2, us=round(1e6 * t, 1))), flush=True)
    dA.free()

for n in [s for s in sizes if s <= 4096]:
    a = ctx.upload(rng.normal(size=(n, n)))
    b = ctx.upload(rng.normal(size=(n, n)))
    c = ctx.zeros(n, n)
    for mf in (1, 0):
        ctx.set_option('gemm_mfma', mf)
        for tA, tB in ((0, 0), (0, 1), (1, 0)):
            p = prof_kind(lambda: ctx.gemm(a, b, c, tA, tB), 1, reps=5)
            us = 1e3 * p['ms'] / p['launches']
            print(json.dumps(dict(op='gemm', n=n, mfma=mf, tA=tA, tB=tB, us=round(us, 1),
                                  TFLOPs=round(2.0 * n ** 3 / us / 1e6, 2))), flush=True)
    ctx.set_option('gemm_mfma', 1)
    # skinny shapes of the eigensolver: (n x n)(n x 32) and rank-32 update
    y = ctx.upload(rng.normal(size=(32, n)))
    m = ctx.zeros(n, 32)
    p = prof_kind(lambda: ctx.gemm(a, y, m, 0, 1), 1, reps=5)
    print(json.dumps(dict(op='gemm n x n x 32 (NT)', n=n, us=round(1e3 * p['ms'] / p['launches'], 1))), flush=True)
    p = prof_kind(lambda: ctx.gemm(m, y, c, 0, 0, -1.0, 1.0), 1, reps=5)
    print(json.dumps(dict(op='gemm rank-32 update', n=n, us=round(1e3 * p['ms'] / p['launches'], 1))), flush=True)
    for h in (a, b, c, y, m):
        h.free()

for n in [s for s in sizes if s <= 4096]:
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    lam = np.exp(rng.uniform(np.log(0.05), np.log(50.0), n))
    lam[0] = -1
    A = (Q * lam) @ Q.T
    A = 0.5 * (A + A.T)
    N = rng.normal(size=(n, n))
    P = A + 5e-3 * 0.5 * (N + N.T)
    g = rng.normal(size=n)
    dA, dP = ctx.upload(A), ctx.upload(P)

    def eig():
        w, V, Vt = ctx.eigh(dP)
        V.free()
        Vt.free()
    for leaf in (16, 32, 64):
        ctx.set_option('eigh_leaf', leaf)
        t = timeit(eig, reps=3, warm=1)
        print(json.dumps(dict(op='eigh', n=n, leaf=leaf, ms=round(1e3 * t, 2))), flush=True)
    ctx.set_option('eigh_leaf', 32)
    ctx.prof_reset()
    ctx.prof_enable(True)
    eig()
    ctx.prof_enable(False)
    print(json.dumps(dict(op='eigh breakdown', n=n, gemv=ctx.prof_get(0), gemm=ctx.prof_get(1))), flush=True)
    t0 = time.perf_counter()
    np.linalg.eigh(P)
    print(json.dumps(dict(op='host LAPACK eigh', n=n, ms=round(1e3 * (time.perf_counter() - t0), 1))), flush=True)

    w, V, Vt = ctx.eigh(dP)
    res = {}

    def dav():
        res['out'] = ctx.davidson(dA, n, g, 0.1, method='jd0', maxiter=40, Pvecs=V, PvecsT=Vt, pevals=w)
    t = timeit(dav, reps=5, warm=1)
    k = res['out'][1].shape[1]
    print(json.dumps(dict(op='davidson jd0', n=n, k=k, ms=round(1e3 * t, 3), iter_per_s=round(k / t, 1))), flush=True)

    def dav1():
        res['out'] = ctx.davidson(dA, n, g, 1e-32, method='jd0', maxiter=40, Pvecs=V, PvecsT=Vt, pevals=w)
    t = timeit(dav1, reps=3, warm=1)
    k = res['out'][1].shape[1]
    print(json.dumps(dict(op='davidson jd0 fixed 40 vectors', n=n, k=k, ms=round(1e3 * t, 3),
                          iter_per_s=round(k / t, 1), us_per_iter=round(1e6 * t / k, 1))), flush=True)
    S = rng.normal(size=(n, 3))
    Y = A @ S
    for kk in (1, 3, 8):
        Sk, Yk = rng.normal(size=(n, kk)), None
        Yk = A @ Sk
        dB = ctx.upload(P)
        t = timeit(lambda: ctx.update_h(dB, Sk, Yk, 'TS-BFGS', 2, evals=w, evecs=V, evecsT=Vt), reps=5, warm=1)
        p = prof_kind(lambda: ctx.update_h(dB, Sk, Yk, 'TS-BFGS', 2, evals=w, evecs=V, evecsT=Vt), 2, reps=3)
        us = 1e3 * p['ms'] / max(1, p['launches'])
        print(json.dumps(dict(op='update_h TS-BFGS', n=n, k=kk, ms_total=round(1e3 * t, 3), fused_pass_us=round(us, 1),
                              fused_GBs=round(16.0 * n * n / us / 1e3, 1))), flush=True)
        dB.free()
