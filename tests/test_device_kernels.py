"""Streaming kernels through the C ABI against numpy (the oracle for a matvec is the matvec):
row-panel matvec for every right-hand-side count and rows-per-wave variant, transposed product,
fp64 GEMM (MFMA and VALU tiles, all transpose combinations, ragged shapes), projection."""
import numpy as np
import pytest


def sizes(ctx):
    return [(37, 1), (64, 2), (130, 3), (200, 8), (70, 11)] if ctx.backend == 'emu' else \
        [(37, 1), (64, 2), (130, 3), (1000, 8), (3072, 1), (3072, 2), (777, 11), (4099, 2)]


def test_symm_mm(ctx):
    rng = np.random.RandomState(0)
    for n, k in sizes(ctx):
        A = rng.normal(size=(n, n))
        X = rng.normal(size=(n, k))
        dA = ctx.upload(A)
        ref = A @ X
        for rw in (1, 2, 4):
            ctx.set_option('gemv_rw', rw)
            np.testing.assert_allclose(ctx.symm_mm(dA, X), ref, atol=1e-13 * n * np.abs(ref).max())
        np.testing.assert_allclose(ctx.symm_mm(dA, X[:, 0]), ref[:, 0], atol=1e-13 * n * np.abs(ref).max())
        np.testing.assert_array_equal(dA.numpy(), A)
        np.testing.assert_array_equal(dA.transpose().numpy(), A.T)
        dA.free()
    ctx.set_option('gemv_rw', 0)


def test_rectangular_and_transposed_products(ctx):
    rng = np.random.RandomState(1)
    for rows, cols, k in [(50, 83, 2), (83, 50, 3), (1, 40, 1), (300, 7, 5)]:
        A = rng.normal(size=(rows, cols))
        dA = ctx.upload(A)
        X = rng.normal(size=(cols, k))
        Z = rng.normal(size=(rows, k))
        np.testing.assert_allclose(ctx.symm_mm(dA, X), A @ X, atol=1e-12)
        np.testing.assert_allclose(ctx.tmatmul(dA, Z), A.T @ Z, atol=1e-12)


@pytest.mark.parametrize('mfma', [0, 1])
def test_gemm(ctx, mfma):
    rng = np.random.RandomState(2)
    ctx.set_option('gemm_mfma', mfma)
    shapes = [(64, 64, 16), (70, 45, 33), (130, 64, 100), (5, 3, 2), (200, 193, 37)]   # last: 128x128 tiles
    if ctx.backend == 'hip':
        shapes += [(512, 384, 256), (1000, 1000, 64), (33, 2000, 1500)]
    for M, N, K in shapes:
        for tA in (0, 1):
            for tB in (0, 1):
                a = rng.normal(size=(K, M) if tA else (M, K))
                b = rng.normal(size=(N, K) if tB else (K, N))
                c0 = rng.normal(size=(M, N))
                dC = ctx.upload(c0)
                ctx.gemm(ctx.upload(a), ctx.upload(b), dC, tA, tB, 0.7, 0.3)
                ref = 0.7 * (a.T if tA else a) @ (b.T if tB else b) + 0.3 * c0
                np.testing.assert_allclose(dC.numpy(), ref, atol=1e-13 * K * max(1, np.abs(ref).max()))
    ctx.set_option('gemm_mfma', 1)


def test_gemm_identity_with_asymmetric_operand(ctx):
    """A = I against an asymmetric B catches a transposed accumulator layout."""
    n = 48
    B = np.arange(n * n, dtype=float).reshape(n, n)
    dC = ctx.zeros(n, n)
    ctx.gemm(ctx.upload(np.eye(n)), ctx.upload(B), dC)
    np.testing.assert_array_equal(dC.numpy(), B)


def test_project(ctx):
    rng = np.random.RandomState(3)
    n, m = (90, 17) if ctx.backend == 'emu' else (1500, 400)
    H = rng.normal(size=(n, n))
    H = H + H.T
    U = np.linalg.qr(rng.normal(size=(n, m)))[0]
    ref = U.T @ H @ U
    np.testing.assert_allclose(ctx.project(ctx.upload(H), U), ref, atol=1e-12 * n)


def test_error_reporting(ctx):
    from sella_amd._lib import SellaHipError
    dA = ctx.upload(np.eye(4))
    with pytest.raises(ValueError):
        ctx.symm_mm(dA, np.zeros(5))
    dA.free()
    with pytest.raises(SellaHipError):
        dA.numpy()
    with pytest.raises(SellaHipError):
        ctx.set_option('no_such_option', 1)


def test_profiling_hooks(ctx):
    """sella_prof_*: launches between enable/disable are counted per kind with their algorithmic
    bytes; the timed launch is the kernel's own dispatch (events attached to the packet)."""
    rng = np.random.RandomState(9)
    n = 200
    A = rng.normal(size=(n, n))
    dA = ctx.upload(A + A.T)
    ctx.prof_reset()
    ctx.prof_enable(True)
    ctx.symm_mm(dA, rng.normal(size=n))
    ctx.set_option('eigh_upd_max', 0)            # the blocked chain at this size too (slot 5 counts ITS matvec launches only)
    try:
        ctx.eigh(dA)
    finally:
        ctx.set_option('eigh_upd_max', 1024)
    ctx.prof_enable(False)
    small, trd = ctx.prof_get(4), ctx.prof_get(5)
    assert small['launches'] >= 1 and small['bytes'] >= 8.0 * n * n and small['ms'] >= 0.0
    # every 4th column is sampled; from the first panel boundary (panels of 16) with at most 128 trailing rows on, one
    # workgroup finishes the factorisation in LDS (eigh_tail_lds) and there is no matvec launch any more
    tail = next(j0 for j0 in range(0, n, 16) if n - j0 <= 128)
    cols = [j for j in range(tail) if j % 4 == 0]
    assert trd['launches'] == len(cols)
    assert abs(trd['bytes'] - 8.0 * sum((n - 1 - j) ** 2 for j in cols)) < 1e-6
    ctx.symm_mm(dA, rng.normal(size=n))          # not counted once disabled
    assert ctx.prof_get(4)['launches'] == small['launches']
    # default options: a trailing block this small goes through the one-launch-per-column chain, which is not slot 5
    ctx.prof_reset()
    ctx.prof_enable(True)
    ctx.eigh(dA)
    ctx.prof_enable(False)
    assert ctx.prof_get(5)['launches'] == 0 and ctx.prof_get(3)['launches'] > 0


def test_block_panel_product(ctx):
    """More than 8 right-hand sides go through the MFMA panel kernel (matrix streamed once per 16
    vectors); same results as the row-panel matvec path and as NumPy, for ragged shapes."""
    rng = np.random.RandomState(21)
    shapes = [(40, 40, 9), (70, 53, 16), (33, 90, 17), (5, 12, 33), (130, 64, 12),
              (100, 300, 16)]
    if ctx.backend == 'hip':
        shapes += [(3072, 3072, 16), (1537, 2049, 24)]
    for rows, cols, k in shapes:
        A = rng.normal(size=(rows, cols))
        X = rng.normal(size=(cols, k))
        dA = ctx.upload(A)
        ref = A @ X
        tol = 1e-13 * cols * max(1.0, np.abs(ref).max())
        ctx.set_option('panel_mfma', 1)
        Y1 = ctx.symm_mm(dA, X)
        ctx.set_option('panel_mfma', 0)
        Y0 = ctx.symm_mm(dA, X)
        ctx.set_option('panel_mfma', 1)
        np.testing.assert_allclose(Y1, ref, atol=tol, rtol=0)
        np.testing.assert_allclose(Y0, ref, atol=tol, rtol=0)
        for rows_per_wg in (32, 48, 64):                          # the launcher picks by size; every variant here
            ctx.set_option('panel_rows', rows_per_wg)
            np.testing.assert_allclose(ctx.symm_mm(dA, X), ref, atol=tol, rtol=0)
        ctx.set_option('panel_rows', 0)
        dA.free()
