"""Device symmetric eigensolver (tridiagonalisation + divide & conquer + compact-WY) against
LAPACK: eigenvalues, residuals and orthogonality, on Hessian-like spectra and on the degenerate
spectra an approximate Hessian really has (scaled identity + low rank)."""
import numpy as np
import pytest


def check(ctx, A, tol=5e-13):
    n = A.shape[0]
    w, V, Vt = ctx.eigh(ctx.upload(A))
    Vn, Vtn = V.numpy(), Vt.numpy()
    wr = np.linalg.eigvalsh(A)
    scale = max(1.0, np.abs(wr).max())
    assert np.all(np.diff(w) >= 0)
    np.testing.assert_allclose(w, wr, atol=tol * scale * n ** 0.5)
    assert np.abs(A @ Vn - Vn * w).max() <= tol * scale * n
    assert np.abs(Vn.T @ Vn - np.eye(n)).max() <= tol * n
    np.testing.assert_array_equal(Vtn.T, Vn)
    return w


def cases(n, rng):
    Q = np.linalg.qr(rng.normal(size=(n, n)))[0]
    lam = np.exp(rng.uniform(np.log(.05), np.log(50), n))
    lam[0] = -1
    u = rng.normal(size=(n, 3))
    A = rng.normal(size=(n, n))
    yield 'random', A + A.T
    yield 'hessian-like', (Q * lam) @ Q.T
    yield 'scaled identity', 3.0 * np.eye(n)
    yield 'identity + low rank', 2.5 * np.eye(n) + u @ u.T - 0.3 * np.outer(u[:, 0] + 1, u[:, 0] + 1)
    yield 'clusters', (Q * np.repeat(np.arange(1, n // 8 + 2), 8)[:n].astype(float)) @ Q.T
    yield 'diagonal', np.diag(rng.normal(size=n))
    T = np.diag(rng.normal(size=n)) + np.diag(rng.normal(size=n - 1), 1)
    yield 'tridiagonal', T + T.T
    yield 'zero', np.zeros((n, n))


def test_small_sizes(ctx):
    rng = np.random.RandomState(0)
    for leaf in (4, 32):
        ctx.set_option('eigh_leaf', leaf)
        for n in (1, 2, 3, 5, 17, 33, 70):
            A = rng.normal(size=(n, n))
            check(ctx, A + A.T)
    ctx.set_option('eigh_leaf', 32)


def test_spectra(ctx):
    rng = np.random.RandomState(1)
    n = 96 if ctx.backend == 'emu' else 700
    ctx.set_option('eigh_leaf', 8 if ctx.backend == 'emu' else 32)
    for name, A in cases(n, rng):
        check(ctx, A)
    ctx.set_option('eigh_leaf', 32)


@pytest.mark.gpu
def test_benchmark_size(ctx):
    if ctx.backend != 'hip':
        pytest.skip('hardware only')
    from conftest import hessian_like
    A, P, g = hessian_like(3072, 0)
    check(ctx, P, tol=2e-12)
