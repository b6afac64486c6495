"""Full-size checks on the MI355X (BASELINE.json sizes): parity against the scalar digests the
real reference produced at n = 300 / 768 / 3072 (tests/golden/big_digests.json, matrices are
regenerated from the seeded recipe), plus size-independent properties at 3N = 3072."""
import json
import os

import numpy as np
import pytest

from conftest import GOLD, hessian_like

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def digests():
    with open(os.path.join(GOLD, 'big_digests.json')) as f:
        return json.load(f)


def ritz_of_span(A, V):
    Q, _ = np.linalg.qr(V)
    return np.linalg.eigvalsh(Q.T @ A @ Q)


@pytest.mark.parametrize('n', [300, 768, 3072])
def test_davidson_trajectory_digest(ctx, digests, n):
    """Step-for-step parity with the reference at benchmark sizes.  The reference's own trajectory
    is not reproducible across BLAS builds beyond the first iterations (its (P - theta)^-1 correction
    amplifies roundoff ~3x per iteration: the same NumPy code stops at k = 16 in the build container
    and at k = 31 on this box's CPU for n = 3072), so the trajectory is pinned where it is well
    defined: the lowest Ritz value after j = 2..8 vectors must equal the reference's to 1e-10."""
    if ctx.backend != 'hip':
        pytest.skip('hardware only')
    from sella_amd.eigensolvers import rayleigh_ritz
    d = digests[str(n)]
    A, P, g = hessian_like(n, 0, eps=5e-3)
    dA = ctx.upload(A)
    dP = ctx.upload(P)
    w, Q, Qt = ctx.eigh(dP)
    # tolerance per vector count = 100 x the reference algorithm's own response to a 1-ulp
    # perturbation of P, measured with tools/krylov_sensitivity.py (n = 768: 6e-14, 6e-12, 2e-9,
    # 1e-7, 6e-6, 6e-2 at j = 2, 4, 6, 8, 12, 16)
    for j, tol in ((2, 1e-10), (3, 1e-10), (4, 1e-9), (6, 2e-7), (8, 1e-5)):
        lams, V, AV, nmv = ctx.davidson(dA, n, g, 0.1, method='jd0', maxiter=j, Pvecs=Q, PvecsT=Qt, pevals=w)
        assert V.shape[1] == j
        assert abs(lams[0] - d['ritz'][j - 1]) < tol * max(1.0, abs(d['ritz'][j - 1])), (j, lams[0], d['ritz'][j - 1])
        np.testing.assert_allclose(AV, A @ V, atol=1e-10)
        np.testing.assert_allclose(V.T @ V, np.eye(j), atol=1e-12)
    # whole default call through the product API: structural properties only
    lams, V, AV = rayleigh_ritz(A, 0.1, P, v0=g, method='jd0', maxiter=40)
    k = V.shape[1]
    np.testing.assert_allclose(AV, A @ V, atol=1e-10)
    np.testing.assert_allclose(V.T @ V, np.eye(k), atol=1e-12)
    np.testing.assert_allclose(np.diag(V.T @ AV), lams, atol=1e-10)
    if k < 40:       # converged by the reference criterion (eigensolvers.py:80-89)
        assert np.linalg.norm(AV[:, 0] - lams[0] * V[:, 0]) < 0.1 * abs(lams[0]) * (1 + 1e-8)


@pytest.mark.parametrize('n', [768, 3072])
def test_davidson_converged_eigenpair(ctx, n):
    """north_star: the converged lowest eigenpair within 1e-10 of the exact (LAPACK) one."""
    if ctx.backend != 'hip':
        pytest.skip('hardware only')
    from sella_amd.eigensolvers import rayleigh_ritz
    A, P, g = hessian_like(n, 0, eps=5e-3)
    lams, V, AV = rayleigh_ritz(A, 1e-7, P, v0=g, method='jd0', maxiter=900)
    assert V.shape[1] < 900
    assert abs(lams[0] - (-1.0)) < 1e-10
    assert np.linalg.norm(A @ V[:, 0] - lams[0] * V[:, 0]) < 1e-6


@pytest.mark.parametrize('n', [300, 768, 3072])
def test_update_digest(ctx, digests, n):
    if ctx.backend != 'hip':
        pytest.skip('hardware only')
    from sella_amd.linalg import ApproximateHessian
    d = digests[str(n)]
    A, P, g = hessian_like(n, 0, eps=5e-3)
    H = ApproximateHessian(n, n, P)
    S = np.random.RandomState(1).normal(size=(n, 3))
    Y = A @ S
    H.update(S, Y)
    B = H.B
    assert abs(np.linalg.norm(B) - d['update_fro']) < 1e-9 * d['update_fro']
    assert abs(np.trace(B) - d['update_trace']) < 1e-9 * abs(d['update_trace'])
    # secant condition (tests/test_hessian_update.py:33-37) and exact symmetry
    np.testing.assert_allclose(B @ S, Y, atol=1e-8 * np.abs(Y).max())
    np.testing.assert_array_equal(B, B.T)


@pytest.mark.parametrize('n', [300, 768])
def test_prfo_digest(ctx, digests, n):
    if ctx.backend != 'hip':
        pytest.skip('hardware only')
    from helpers import FakePES
    from sella_amd.linalg import ApproximateHessian
    from sella_amd.optimize.restricted_step import get_restricted_step
    d = digests[str(n)]
    A, P, g = hessian_like(n, 0, eps=5e-3)
    s, smag = get_restricted_step('tr')(FakePES(ApproximateHessian, P, g, 0, seed=0), 1, 0.1, 'prfo').get_s()
    probe = np.cos(np.arange(n) * 0.37)
    assert abs(np.linalg.norm(s) - d['prfo_tr_s_norm']) < 1e-10
    assert abs(probe @ s - d['prfo_tr_s_probe']) < 1e-9


def test_eigh_properties_3072(ctx):
    if ctx.backend != 'hip':
        pytest.skip('hardware only')
    n = 3072
    A, P, g = hessian_like(n, 1)
    dP = ctx.upload(P)
    w, V, Vt = ctx.eigh(dP)
    Vn = V.numpy()
    assert np.all(np.diff(w) >= 0)
    np.testing.assert_allclose(w, np.linalg.eigvalsh(P), atol=1e-10)
    assert np.abs(P @ Vn - Vn * w).max() < 1e-10
    assert np.abs(Vn.T @ Vn - np.eye(n)).max() < 1e-11
    # trace / Frobenius invariants (size-independent checks)
    assert abs(w.sum() - np.trace(P)) < 1e-8
    assert abs(np.sqrt((w ** 2).sum()) - np.linalg.norm(P)) < 1e-8


def test_matvec_linearity_3072(ctx):
    if ctx.backend != 'hip':
        pytest.skip('hardware only')
    n = 3072
    rng = np.random.RandomState(5)
    A = rng.normal(size=(n, n))
    dA = ctx.upload(A)
    x, y = rng.normal(size=n), rng.normal(size=n)
    lhs = ctx.symm_mm(dA, 2.0 * x - 3.0 * y)
    rhs = 2.0 * ctx.symm_mm(dA, x) - 3.0 * ctx.symm_mm(dA, y)
    np.testing.assert_allclose(lhs, rhs, atol=1e-9)
    np.testing.assert_allclose(ctx.symm_mm(dA, x), A @ x, atol=1e-9)
