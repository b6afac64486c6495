"""Full-size checks on the MI355X (BASELINE.json sizes): parity against the scalar digests the
real reference produced at n = 300 / 768 / 3072 (tests/golden/big_digests.json, matrices are
regenerated from the seeded recipe), plus size-independent properties at 3N = 3072."""
import json
import os

import numpy as np
import pytest

from conftest import GOLD, hessian_like, make_context

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx(request):
    """Hardware only: this module never instantiates the emulator (the module-level gpu mark would otherwise
    select the 'emu' parameter of the shared fixture and build the host emulation on the GPU box)."""
    yield from make_context(request, 'hip')


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
    from sella_amd.eigensolvers import rayleigh_ritz
    A, P, g = hessian_like(n, 0, eps=5e-3)
    lams, V, AV = rayleigh_ritz(A, 1e-7, P, v0=g, method='jd0', maxiter=900)
    assert V.shape[1] < 900
    assert abs(lams[0] - (-1.0)) < 1e-10
    assert np.linalg.norm(A @ V[:, 0] - lams[0] * V[:, 0]) < 1e-6


@pytest.mark.parametrize('n', [300, 768, 3072])
def test_update_digest(ctx, digests, n):
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
    n = 3072
    rng = np.random.RandomState(5)
    A = rng.normal(size=(n, n))
    dA = ctx.upload(A)
    x, y = rng.normal(size=n), rng.normal(size=n)
    lhs = ctx.symm_mm(dA, 2.0 * x - 3.0 * y)
    rhs = 2.0 * ctx.symm_mm(dA, x) - 3.0 * ctx.symm_mm(dA, y)
    np.testing.assert_allclose(lhs, rhs, atol=1e-9)
    np.testing.assert_allclose(ctx.symm_mm(dA, x), A @ x, atol=1e-9)


# ---- BASELINE configs[3]: ensemble of independent searches on one device -------------------------------------
def _ensemble_member_factory(ctx, ne, host):
    from sella_amd.atoms import Atoms, QuadraticCubicModel

    def make_member(i):
        Ai, Ui, x0 = host[i]
        dAi = ctx.upload(Ai)
        at = Atoms(['X'] * (ne // 3), x0.copy(), pbc=True)
        at.calc = QuadraticCubicModel(lambda x, dAi=dAi: ctx.symm_mm(dAi, x), Ui, c=0.05)
        return at
    return make_member


def test_ensemble_single_gpu(ctx):
    """8 x (3N = 768) saddle searches through `run_ensemble` on the real device (configs[3], one GPU's share):
    per-replica summaries and final positions bit-identical to running each member alone, every lambda_min < 0."""
    from sella_amd.ensemble import run_ensemble, run_one
    ne, nrep, steps = 768, 8, 12
    host = {}
    for i in range(nrep):
        rng = np.random.RandomState(6000 + i)
        U = rng.normal(size=(8, ne))
        U /= np.linalg.norm(U, axis=1)[:, None]
        host[i] = (hessian_like(ne, seed=5000 + i)[0], U, 0.05 * rng.normal(size=(ne // 3, 3)))
    kw = dict(order=1, eta=1e-4, gamma=0.1, delta0=0.1, rs='tr', proj_trans=False)
    make_member = _ensemble_member_factory(ctx, ne, host)
    res = run_ensemble(make_member, nrep, fmax=0.0, steps=steps, sella_kwargs=kw)
    assert res['summary'].shape == (nrep, 5)
    assert np.all(res['summary'][:, 1] == steps)
    assert np.all(res['summary'][:, 4] < 0.0), res['summary'][:, 4]
    assert np.all(np.isfinite(res['summary']))
    for i in (0, 3, 7):
        sm, pos = run_one(make_member(i), 0.0, steps, kw)
        np.testing.assert_array_equal(sm, res['summary'][i])
        np.testing.assert_array_equal(pos, res['positions'][i])
    # host threads, one context per thread on the same device: same per-replica results
    from sella_amd import device

    def make_member_t(i):
        return _ensemble_member_factory(device.get_context(), ne, host)(i)
    res_t = run_ensemble(make_member_t, nrep, fmax=0.0, steps=steps, sella_kwargs=kw, threads=4)
    np.testing.assert_array_equal(res_t['summary'], res['summary'])
    # worker processes, own interpreter and device context each (`EnsemblePool`, what bench.py uses): the same again
    from conftest_shim import EnsembleFactory
    from sella_amd.ensemble import EnsemblePool
    with EnsemblePool(3) as pool:
        assert len(set(pool.pids)) == 3
        res_p = run_ensemble(EnsembleFactory(ne, host), nrep, fmax=0.0, steps=steps, sella_kwargs=kw, pool=pool)
    np.testing.assert_array_equal(res_p['summary'], res['summary'])
    for i in range(nrep):
        np.testing.assert_array_equal(res_p['positions'][i], res['positions'][i])


# ---- BASELINE configs[4]: n = 12288 --------------------------------------------------------------------------
N4 = 12288


@pytest.fixture(scope='module')
def big_matrix():
    """Dense symmetric 12288 x 12288 (1.2 GB): Wigner-type, no O(n^3) host work to build."""
    rng = np.random.RandomState(12288)
    G = rng.standard_normal((N4, N4))
    G += G.T
    return G


def test_panel16_12288(ctx, big_matrix):
    """Block product H V with k = 16 right-hand sides on the matrix cores (`panel16_mfma_kernel`) vs NumPy."""
    rng = np.random.RandomState(3)
    X = rng.standard_normal((N4, 16))
    dH = ctx.upload(big_matrix)
    Y = ctx.symm_mm(dH, X)
    ref = big_matrix @ X
    np.testing.assert_allclose(Y, ref, atol=1e-9 * np.abs(ref).max())
    # linearity through the same kernel (size-independent property)
    Y2 = ctx.symm_mm(dH, 2.0 * X[:, ::-1] - X)
    np.testing.assert_allclose(Y2, 2.0 * Y[:, ::-1] - Y, atol=1e-9 * np.abs(ref).max())
    dH.free()


def test_eigh_12288(ctx, big_matrix):
    """`sella_eigh` at configs[4] size: ordering, trace / Frobenius invariants, and residual / orthogonality of a
    sample of eigenpairs against the host (a full LAPACK eigh at this size takes minutes on the box's cores)."""
    A = big_matrix
    dA = ctx.upload(A)
    w, V, Vt = ctx.eigh(dA)
    assert np.all(np.diff(w) >= 0)
    scale = np.abs(w).max()
    assert abs(w.sum() - np.trace(A)) < 1e-9 * scale * N4
    assert abs(np.sqrt((w ** 2).sum()) - np.linalg.norm(A)) < 1e-10 * np.linalg.norm(A)
    Vt_n = Vt.numpy()
    idx = np.r_[0:8, N4 // 2 - 4:N4 // 2 + 4, N4 - 8:N4, np.random.RandomState(1).randint(0, N4, 40)]
    Vs = Vt_n[idx].T
    assert np.abs(A @ Vs - Vs * w[idx]).max() < 5e-13 * N4 * scale / 100
    G = Vt_n[idx] @ Vt_n.T
    G[np.arange(len(idx)), idx] -= 1.0
    assert np.abs(G).max() < 1e-10
    # known spectrum: three Householder reflections of a diagonal matrix (O(n^2) to build, eigenvalues exact)
    rng = np.random.RandomState(9)
    d = np.sort(rng.uniform(-3.0, 7.0, N4))
    M = np.diag(d)
    for _ in range(3):
        u = rng.standard_normal(N4)
        u /= np.linalg.norm(u)
        Mu = M @ u
        M = M - 2.0 * np.outer(u, Mu) - 2.0 * np.outer(Mu, u) + 4.0 * (u @ Mu) * np.outer(u, u)
    dA.set(M)
    w2, V2, Vt2 = ctx.eigh(dA)
    np.testing.assert_allclose(w2, d, atol=1e-10)
    for m in (dA, V, Vt, V2, Vt2):
        m.free()


def test_block_davidson_12288(ctx, big_matrix):
    """configs[4]: block Davidson (16 new vectors per iteration, operator streamed once per block on the matrix
    cores) for the 16 lowest eigenpairs at 3N = 12288, preconditioned through the eigenbasis of a perturbed operator.
    Parity target: exact() (sella/eigensolvers.py:9-28) — here the device eigensolver checked in test_eigh_12288 —
    to 1e-10 relative to |A|, plus host-side residuals; the caps of the one-vector driver do not apply."""
    A = big_matrix
    rng = np.random.RandomState(77)
    E = rng.standard_normal((N4, N4))
    E += E.T
    P = A + 5e-4 * E
    del E
    dA, dP = ctx.upload(A), ctx.upload(P)
    del P
    wA, _, _ = ctx.eigh(dA, vectors=False)
    w, Q, Qt = ctx.eigh(dP)
    nev = 16
    out = ctx.davidson_block(dA, N4, nev, block=16, tol=1e-9, maxiter=300, Pvecs=Q, PvecsT=Qt, pevals=w)
    assert out['nconv'] == nev, out
    scale = np.abs(wA).max()
    np.testing.assert_allclose(out['lams'], wA[:nev], atol=1e-10 * scale)
    V = out['V']
    np.testing.assert_allclose(V.T @ V, np.eye(nev), atol=1e-10)
    assert np.abs(A @ V - V * out['lams']).max() < 1e-6 * scale
    # the one-vector driver refuses what it cannot hold instead of overrunning its exchange buffers
    from sella_amd._lib import SellaHipError
    with pytest.raises(SellaHipError):
        ctx.davidson(dA, 20000, np.ones(20000), 0.1)
    for m in (dA, dP, Q, Qt):
        m.free()
