"""NumericalHessian / MatrixSum / ApproximateHessian of sella_amd.linalg: golden parity for update sequences from the
uninitialised state (g6) and for the finite-difference operator including its sign-canonicalisation branches (g9), plus
own property tests on a model with closed-form derivatives."""
import numpy as np
import pytest

from conftest import load_golden
from helpers import SmoothModel, random_matrix


@pytest.mark.parametrize('sub,threepoint,quadratic', [(0, False, True), (0, True, False), (4, True, False), (6, False, False)])
def test_numerical_hessian_products_and_sums(ctx, sub, threepoint, quadratic):
    """The finite-difference operator (linalg.py:14-118) on a model with closed-form Hessian: `dot` and `T.dot` agree with
    the Hessian (projected through `Uproj` when there is one) to the difference error, also for columns that are parallel
    to the gradient or to an earlier column (the branches that reuse or canonicalise a direction), and `op + op + matrix`
    is a `MatrixSum` whose products are the sums of the products."""
    from sella_amd.linalg import MatrixSum, NumericalHessian
    rng = np.random.RandomState(31)
    n = 9
    m1, m2 = SmoothModel(n, rng, quadratic_only=quadratic), SmoothModel(n, rng, quadratic_only=quadratic)
    x = 0.5 * rng.standard_normal(n)
    U = np.linalg.qr(rng.standard_normal((n, sub)))[0] if sub else None
    k = sub if sub else n
    proj = (lambda H: U.T @ H @ U) if sub else (lambda H: H)
    kw = dict(x0=x, eta=1e-6, threepoint=threepoint, Uproj=U)
    g1, g2 = m1.energy_gradient(x)[1], m2.energy_gradient(x)[1]
    op1 = NumericalHessian(m1.energy_gradient, g0=g1, **kw)
    extra = random_matrix(rng, k, symmetric=True)
    total = op1 + NumericalHessian(m2.energy_gradient, g0=g2, **kw) + extra
    assert isinstance(total, MatrixSum)
    Href = proj(m1.hessian(x)) + proj(m2.hessian(x)) + extra
    gk = U.T @ g1 if sub else g1
    M = rng.standard_normal((k, 4))
    M[:, 1] = gk / np.linalg.norm(gk)                      # parallel to the gradient
    M[:, 2] = -2.0 * M[:, 0]                               # parallel to an earlier column, opposite sign
    tol = 1e-9 if quadratic else (2e-7 if threepoint else 5e-5)
    np.testing.assert_allclose(total.dot(M), Href @ M, atol=tol * max(1.0, np.abs(Href).max()))
    np.testing.assert_allclose(total.T.dot(M), Href.T @ M, atol=tol * max(1.0, np.abs(Href).max()))
    np.testing.assert_allclose(op1.dot(M[:, 3]), proj(m1.hessian(x)) @ M[:, 3], atol=tol * max(1.0, np.abs(Href).max()))


def test_golden_numerical_hessian(ctx, manifest):
    from sella_amd.linalg import NumericalHessian
    g = load_golden('g9_numhess')
    for case in manifest['g9_numhess']:
        i = case['id']
        A, U4, c3, c4 = g[f'c{i}_A'], g[f'c{i}_U4'], case['c3'], case['c4']

        def f(x, A=A, U4=U4, c3=c3, c4=c4):
            p = U4 @ x
            return (0.5 * x @ A @ x + c3 / 3 * np.sum(p ** 3) + c4 / 4 * np.sum(p ** 4),
                    A @ x + U4.T @ (c3 * p ** 2 + c4 * p ** 3))
        U = g[f'c{i}_Uproj'] if case['sub'] > 0 else None
        H = NumericalHessian(f, g[f'c{i}_x'], g[f'c{i}_g'], 1e-6, case['threepoint'], U)
        np.testing.assert_allclose(H.dot(g[f'c{i}_M']), g[f'c{i}_out'], atol=1e-12)
        np.testing.assert_allclose(H.Vs, g[f'c{i}_Vs'], atol=1e-12)
        np.testing.assert_allclose(H.AVs, g[f'c{i}_AVs'], atol=1e-12)


def test_golden_approximate_hessian(ctx, manifest):
    from sella_amd.linalg import ApproximateHessian
    g = load_golden('g6_approx_hessian')
    case = manifest['g6_approx_hessian'][0]
    n = case['n']
    H = ApproximateHessian(n, n, None)
    assert H.B is None and H.evals is None and not H.initialized
    np.testing.assert_array_equal(H.asarray(), np.eye(n))
    v = np.arange(n, dtype=float)
    np.testing.assert_array_equal(H @ v, v)
    for step in range(len(case['seq'])):
        H.update(g[f's{step}_dx'], g[f's{step}_dg'])
        ref = g[f's{step}_B']
        np.testing.assert_allclose(H.B, ref, atol=1e-10 * np.abs(ref).max())
    np.testing.assert_allclose(H.project(g['proj_U']).B, g['proj_B'], atol=1e-11)
    np.testing.assert_allclose(H.evals, g['evals'], atol=1e-10)
    np.testing.assert_allclose((H + g['add_M']).B, g['add_B'], atol=1e-11)
    np.testing.assert_allclose(H @ v, H.B @ v, atol=1e-11)
    np.testing.assert_allclose(H.evecs @ np.diag(H.evals) @ H.evecs.T, H.B, atol=1e-10)


def test_approximate_hessian_protocol(ctx):
    """tests/test_core_functionality.py:26-93 re-stated."""
    from sella_amd.linalg import ApproximateHessian
    rng = np.random.RandomState(3)
    n = 12
    B0 = rng.normal(size=(n, n))
    B0 = B0 + B0.T
    H = ApproximateHessian(n, n, B0)
    assert H.initialized
    np.testing.assert_allclose(H.evals, np.linalg.eigvalsh(B0), atol=1e-12)
    H.set_B(None)
    assert H.B is None and not H.initialized
    H.set_B(2.0)
    np.testing.assert_array_equal(H.B, 2.0 * np.eye(n))
    assert not H.initialized
    U = np.linalg.qr(rng.normal(size=(n, 4)))[0]
    assert ApproximateHessian(n, n, None).project(U).B is None
    assert (ApproximateHessian(n, n, None) + np.eye(n)).B is None
    X = rng.normal(size=(n, 3))
    H.set_B(B0)
    np.testing.assert_allclose(H @ X, B0 @ X, atol=1e-12)


def test_approximate_hessian_carries_eigenpairs(ctx):
    """ApproximateHessian.update keeps (evals, evecs) valid across quasi-Newton updates without a new
    factorisation, and agrees with the recompute-from-scratch behaviour of the reference
    (linalg.py:274-304 drops the cache, :174-231 recomputes)."""
    from sella_amd import linalg
    rng = np.random.RandomState(3)
    n = 24
    A = rng.normal(size=(n, n))
    H = A + A.T
    B0 = np.diag(np.linspace(0.5, 3.0, n))
    carried = linalg.ApproximateHessian(n, n, B0.copy())
    old = linalg.EIG_UPDATE_MAX_RANK
    try:
        linalg.EIG_UPDATE_MAX_RANK = 0
        fresh = linalg.ApproximateHessian(n, n, B0.copy())
        for step in range(5):
            dx = 0.1 * rng.normal(size=n)
            dg = H @ dx
            linalg.EIG_UPDATE_MAX_RANK = 8
            carried.evals                       # make sure a decomposition exists to be carried
            carried.update(dx, dg)
            assert carried._evals is not None and carried._eig_age > 0
            linalg.EIG_UPDATE_MAX_RANK = 0
            fresh.update(dx, dg)
            assert fresh._evals is None
            np.testing.assert_allclose(carried.B, fresh.B, atol=1e-12)
            np.testing.assert_allclose(carried.evals, fresh.evals, atol=1e-11)
            V = carried.evecs
            assert np.abs(carried.B @ V - V * carried.evals).max() < 1e-11
    finally:
        linalg.EIG_UPDATE_MAX_RANK = old


def test_principal_view_stays_in_step(ctx):
    """The projected Hessian of a pinned-coordinate constraint set (U = columns of the identity) is registered
    as a view of B: every quasi-Newton update reaches its matrix and its eigendecomposition too, so the step
    solve never re-diagonalises (the reference: get_HL_projected + eigh at every step, peswrapper.py:363-386)."""
    from sella_amd import linalg
    rng = np.random.RandomState(7)
    n = 30
    A = rng.normal(size=(n, n))
    Htrue = A + A.T + 4 * np.eye(n)
    free = np.sort(rng.choice(n, 17, replace=False))
    U = np.ascontiguousarray(np.eye(n)[:, free])
    H = linalg.ApproximateHessian(n, n, np.diag(np.linspace(0.5, 3.0, n)))
    sub = H.project(U)
    sub.evals                                   # the step solve diagonalised it once
    H.evals
    H.register_view(U, sub)
    assert H.principal_view(U) is sub and H.principal_view(U.copy()) is None
    for step in range(6):
        dx = 0.1 * rng.normal(size=n)
        H.update(dx, Htrue @ dx)
        view = H.principal_view(U)
        assert view is sub and sub._evals is not None and sub._eig_age > 0      # carried, not recomputed
        Bs = H.B[np.ix_(free, free)]
        np.testing.assert_allclose(sub.B, Bs, atol=1e-12)
        np.testing.assert_allclose(sub.evals, np.linalg.eigvalsh(Bs), atol=1e-11)
        V = sub.evecs
        assert np.abs(Bs @ V - V * sub.evals).max() < 1e-11
        assert np.abs(V.T @ V - np.eye(len(free))).max() < 1e-12
    # a view without eigenpairs still gets the matrix update; set_B drops the view
    sub._drop_eig()
    dx = 0.1 * rng.normal(size=n)
    H.update(dx, Htrue @ dx)
    np.testing.assert_allclose(H.principal_view(U).B, H.B[np.ix_(free, free)], atol=1e-12)
    H.set_B(np.eye(n))
    assert H.principal_view(U) is None


def test_host_thread_cap():
    """utilities/hostcpu.py: the BLAS pools are capped at the CPUs the process may use (cgroup quota).
    In a subprocess: the cap is process-wide."""
    import os
    import subprocess
    import sys
    code = (
        "import os\n"
        "from threadpoolctl import threadpool_info\n"
        "import sella_amd\n"
        "from sella_amd.utilities.hostcpu import effective_cpu_count, limit_blas_threads\n"
        "import numpy, scipy.linalg\n"
        "n = effective_cpu_count()\n"
        "assert 1 <= n <= (os.cpu_count() or 1)\n"
        "assert all(p['num_threads'] <= n for p in threadpool_info()), threadpool_info()\n"
        "assert limit_blas_threads(1) == 1\n"
        "assert all(p['num_threads'] == 1 for p in threadpool_info())\n"
        "limit_blas_threads(n)\n"
        "assert all(p['num_threads'] == 1 for p in threadpool_info())\n"
        "print('ok')\n")
    env = {k: v for k, v in os.environ.items() if k not in ('SELLA_HOST_THREADS', 'OPENBLAS_NUM_THREADS')}
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', code], cwd=repo, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and 'ok' in r.stdout, r.stderr
