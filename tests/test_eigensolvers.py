"""Davidson / Rayleigh-Ritz on the device.

* golden parity: every case of tests/golden/g1_davidson.npz (generated from the real reference)
  — subspace size, Ritz values, Ritz vectors up to sign, and AV = A V;
* the reference's own property tests (tests/test_eigensolvers.py:11-71) re-stated on
  sella_amd.eigensolvers with the finite-difference operator as a host callback."""
import numpy as np
import pytest

from conftest import load_golden
from helpers import SmoothModel, colsign, random_matrix

STRICT = ('jd0', 'jd0_alt', 'lanczos')


def run_case(ctx, g, case):
    i = case['id']
    A, P, n = g[f'c{i}_A'], g[f'c{i}_P'], case['n']
    maxiter = None if case['maxiter'] < 0 else case['maxiter']
    dA = ctx.upload(A)
    kw = {}
    w = Q = None
    if case['P'] != 'eye':
        w, Q = np.linalg.eigh(P)          # inputs of the kernel under test, not its result
        kw = dict(Pvecs=ctx.upload(Q), PvecsT=ctx.upload(Q.T.copy()), pevals=w)
    if case['start'] == 'v0':
        v0 = g[f'c{i}_v0']
    else:
        v0 = Q[:, :max(1, int((w < 0).sum()))]
    lams, V, AV, nmv = ctx.davidson(dA, n, v0, case['gamma'], method=case['method'],
                                    maxiter=maxiter, **kw)
    rl, rV = g[f'c{i}_lams'], g[f'c{i}_V']
    assert V.shape == rV.shape, (case, V.shape, rV.shape)
    strict = case['method'] in STRICT
    # A Krylov process amplifies roundoff roughly geometrically with the iteration count (the JD
    # correction applies a nearly singular (P - theta)^-1), so only the leading, converging Ritz
    # pair is pinned tightly; the trailing, unconverged Ritz values are trajectory-sensitive even
    # between two LAPACK builds.  Bounds scale with the oracle-vs-reference deviation recorded at
    # golden-generation time (manifest dev_*); floor = north_star's 1e-10 on the lowest eigenvalue.
    assert abs(lams[0] - rl[0]) <= max(1e-10 if strict else 1e-7, 100 * case['dev_lam0'])
    assert np.abs(lams - rl).max() <= max(1e-5 if strict else 1e-3, 1e4 * case['dev_lams'])
    v0d, v0r = V[:, 0], rV[:, 0]
    assert min(np.abs(v0d - v0r).max(), np.abs(v0d + v0r).max()) <= max(1e-5 if strict else 1e-3, 1e4 * case['dev_V'])
    np.testing.assert_allclose(AV, A @ V, atol=1e-11)
    np.testing.assert_allclose(V.T @ V, np.eye(V.shape[1]), atol=1e-12)
    assert nmv == V.shape[1]


def test_golden_davidson(ctx, manifest):
    g = load_golden('g1_davidson')
    cases = manifest['g1_davidson']
    if ctx.backend == 'emu':          # the emulator is slow: default method in full, one of each other
        cases = [c for c in cases if c['method'] == 'jd0' or (c['P'] == 'noisy' and c['gamma'] == 1e-32)
                 or (c['start'] == 'P')]
    for case in cases:
        run_case(ctx, g, case)


def test_golden_expand_through_one_iteration(ctx, manifest):
    """One correction vector per method (g2_expand): drive the device loop for exactly one
    expansion from the golden Ritz basis and compare span(V, t)."""
    g = load_golden('g2_expand')
    for case in manifest['g2_expand']:
        if case['seeking'] != 0:
            continue
        i = case['id']
        V, P, lams, t = g[f'c{i}_V'], g[f'c{i}_P'], g[f'c{i}_lams'], g[f'c{i}_t']
        n, k = V.shape
        # a symmetric operator whose action on span(V) is the golden Y
        Y = g[f'c{i}_Y']
        w, Q = np.linalg.eigh(P)
        A = Y @ V.T + V @ Y.T - V @ (V.T @ Y) @ V.T
        lam_dev, Vd, AVd, _ = ctx.davidson(ctx.upload(A), n, V, 1e-32, method=case['method'],
                                            maxiter=k + 1, Pvecs=ctx.upload(Q),
                                            PvecsT=ctx.upload(Q.T.copy()), pevals=w)
        assert Vd.shape[1] == k + 1
        # the new direction must be the golden t orthogonalised against V
        tperp = t - V @ (V.T @ t)
        tperp /= np.linalg.norm(tperp)
        resid = tperp - Vd @ (Vd.T @ tperp)
        assert np.linalg.norm(resid) < 1e-8, case


@pytest.mark.parametrize('threepoint', [False, True])
def test_exact_on_a_matrix_and_on_a_finite_difference_operator(ctx, threepoint):
    """`exact` (eigensolvers.py:9-28): a dense matrix goes through the device eigh, an operator is applied to the columns
    of the identity (or of P's eigenvectors) first.  Against LAPACK on the closed-form Hessian of a smooth model; the
    operator route within the finite-difference error."""
    from sella_amd.eigensolvers import exact
    from sella_amd.linalg import NumericalHessian
    rng = np.random.RandomState(23)
    n = 11
    model = SmoothModel(n, rng)
    x = 0.4 * rng.standard_normal(n)
    Hx = model.hessian(x)
    wref, Vref = np.linalg.eigh(Hx)
    lam, V, AV = exact(Hx)
    np.testing.assert_allclose(lam, wref, atol=1e-12)
    np.testing.assert_allclose(colsign(V, Vref), Vref, atol=1e-9)
    np.testing.assert_allclose(AV, Hx @ V, atol=1e-12)
    op = NumericalHessian(model.energy_gradient, g0=model.energy_gradient(x)[1], x0=x, eta=1e-6, threepoint=threepoint)
    fd = 1e-7 if threepoint else 2e-5                      # central / forward difference error at eta = 1e-6
    for P in (None, Hx + 1e-3 * random_matrix(rng, n, symmetric=True)):
        lam2, V2, AV2 = exact(op, P=P)
        np.testing.assert_allclose(lam2, wref, atol=fd)
        assert np.abs(np.abs(Vref.T @ V2) - np.eye(n)).max() < 50 * fd
        np.testing.assert_allclose(AV2, Hx @ V2, atol=10 * fd)


@pytest.mark.parametrize('method', ['jd0', 'jd0_alt', 'mjd0', 'mjd0_alt', 'gd', 'lanczos'])
def test_rayleigh_ritz_on_a_finite_difference_operator(ctx, method):
    """Every `method` of `expand` driving the loop on an OPERATOR (each product a pair of gradient calls): the returned
    Ritz values are those of V^T A V, V is orthonormal with AV = A V, a converged run (gamma = 0: until the space is
    exhausted) reproduces the lowest eigenpair, `maxiter` is respected, and `vref` only adds a log line."""
    from sella_amd.eigensolvers import rayleigh_ritz
    from sella_amd.linalg import NumericalHessian
    rng = np.random.RandomState(29)
    n = 10
    model = SmoothModel(n, rng)
    x = 0.3 * rng.standard_normal(n)
    Hx = model.hessian(x)
    wref, Vref = np.linalg.eigh(Hx)

    def operator():
        return NumericalHessian(model.energy_gradient, g0=model.energy_gradient(x)[1], x0=x, eta=1e-6, threepoint=True)
    lam, V, AV = rayleigh_ritz(operator(), 0.1, np.eye(n), method=method)
    np.testing.assert_allclose(V.T @ V, np.eye(V.shape[1]), atol=1e-9)
    np.testing.assert_allclose(AV, Hx @ V, atol=1e-6)
    np.testing.assert_allclose(lam, np.linalg.eigvalsh(V.T @ AV), atol=1e-6)
    assert lam[0] >= wref[0] - 1e-6                        # a Ritz value never undercuts the spectrum
    lam, V, AV = rayleigh_ritz(operator(), 0.0, np.eye(n), method=method, v0=rng.standard_normal(n))
    assert lam[0] == pytest.approx(wref[0], abs=1e-6)
    assert abs(V[:, 0] @ Vref[:, 0]) > 1 - 1e-6
    lam, V, AV = rayleigh_ritz(operator(), 1e-30, np.eye(n), method=method, v0=rng.standard_normal(n), maxiter=3,
                               vref=Vref[:, 0])
    assert V.shape[1] <= 4


def test_unknown_method_and_metric(ctx):
    from sella_amd.eigensolvers import rayleigh_ritz
    A = np.diag(np.arange(1., 9.))
    with pytest.raises(ValueError):
        rayleigh_ritz(A, 0.1, np.eye(8), v0=np.ones(8), method='nope')
    with pytest.raises(ValueError):                              # a metric must be positive definite
        rayleigh_ritz(A, 0.1, np.eye(8), B=-np.eye(8), v0=np.ones(8))


@pytest.mark.parametrize('as_operator', [False, True])
def test_generalised_metric(ctx, as_operator):
    """A x = theta B x (eigensolvers.py:35-36, 58, 70): converged pairs against scipy's generalised eigh; V comes back
    B-orthonormal and AV = A V, as the reference returns them."""
    from scipy.linalg import eigh as geigh
    from sella_amd.eigensolvers import rayleigh_ritz
    rng = np.random.RandomState(31)
    n = 24
    A = rng.normal(size=(n, n))
    A = 0.5 * (A + A.T)
    M = rng.normal(size=(n, n))
    B = M @ M.T / n + 0.5 * np.eye(n)
    wref, Xref = geigh(A, B)
    P = A + 0.05 * np.diag(rng.normal(size=n))

    class Op:
        shape = A.shape

        def dot(self, v):
            return A @ v

    lams, V, AV = rayleigh_ritz(Op() if as_operator else A, 1e-9, P, B=B, v0=rng.normal(size=n), method='jd0')
    k = V.shape[1]
    np.testing.assert_allclose(V.T @ B @ V, np.eye(k), atol=1e-9)
    np.testing.assert_allclose(AV, A @ V, atol=1e-9)
    assert abs(lams[0] - wref[0]) < 1e-8
    x = V[:, 0]
    assert np.linalg.norm(A @ x - lams[0] * (B @ x)) < 1e-6
    assert abs(abs(x @ B @ Xref[:, 0]) - 1.0) < 1e-6


def test_callback_failure_propagates(ctx):
    from sella_amd.eigensolvers import rayleigh_ritz

    class Boom:
        shape = (8, 8)

        def dot(self, v):
            raise RuntimeError('calculator failed')
    with pytest.raises(RuntimeError, match='calculator failed'):
        rayleigh_ritz(Boom(), 0.1, np.eye(8), v0=np.ones(8))


def test_size_limit_of_the_one_vector_solver_is_reported(ctx):
    """`sella_davidson` keeps its per-iteration partial sums in a fixed exchange layout (3N <= 16000: all BASELINE
    configurations, 12288 included); beyond it the call fails with a message naming the block solver instead of
    overrunning the layout."""
    from sella_amd._lib import SellaHipError
    n = 16001
    with pytest.raises(SellaHipError, match='sella_davidson_block'):
        ctx.davidson(lambda v: v, n, np.ones(n), 0.1, method='lanczos', maxiter=2)
