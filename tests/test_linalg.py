"""NumericalHessian / MatrixSum / ApproximateHessian of sella_amd.linalg: the reference's own
tests (tests/test_linalg.py:11-58, tests/test_core_functionality.py:26-93) re-stated, plus golden
parity for update sequences from the uninitialised state (g6) and the finite-difference operator
including its sign-canonicalisation branches (g9)."""
import numpy as np
import pytest
from scipy.stats import ortho_group

from conftest import load_golden
from helpers import poly_factory


@pytest.mark.parametrize("dim,subdim,order,threepoint",
                         [(3, None, 1, False), (3, None, 1, True), (5, 3, 2, True),
                          (10, None, 4, True), (10, 6, 4, False)])
def test_NumericalHessian(ctx, dim, subdim, order, threepoint, eta=1e-6, atol=1e-4):
    from sella_amd.linalg import NumericalHessian
    rng = np.random.RandomState(2)
    tol = dict(rtol=atol, atol=eta ** 2)
    x = rng.normal(size=dim)
    poly1 = poly_factory(dim, order, rng)
    _, g1, h1 = poly1(x)
    poly2 = poly_factory(dim, order, rng)
    _, g2, h2 = poly2(x)
    if subdim is None:
        U, subdim, g1proj, xproj = None, dim, g1, x
    else:
        U = ortho_group.rvs(dim, random_state=rng)[:, :subdim]
        h1 = U.T @ h1 @ U
        h2 = U.T @ h2 @ U
        g1proj = U.T @ g1
        xproj = U.T @ x
    Hkwargs = dict(x0=x, eta=eta, threepoint=threepoint, Uproj=U)
    H1 = NumericalHessian(lambda x: poly1(x)[:2], g0=g1, **Hkwargs)
    M1 = rng.normal(size=(subdim, subdim))
    H2 = H1 + NumericalHessian(lambda x: poly2(x)[:2], g0=g2, **Hkwargs) + M1
    H3 = h1 + h2 + M1
    M1[:, 0] = xproj - g1proj * (xproj @ g1proj) / (g1proj @ g1proj)
    M1[:, 1] -= M1[:, 0] * (M1[:, 1] @ M1[:, 0]) / (M1[:, 0] @ M1[:, 0])
    M1[:, 1] -= g1proj * (M1[:, 1] @ g1proj) / (g1proj @ g1proj)
    np.testing.assert_allclose(H2.T.dot(M1), H3.T @ M1, **tol)


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
