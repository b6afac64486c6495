"""Gram-Schmidt (sella/utilities/math.pyx semantics, tests/utilities/test_math.py:46-75) and
thin QR through the C ABI."""
import numpy as np
import pytest

from conftest import load_golden


def test_mgs_golden(ctx, manifest):
    g = load_golden('g3_mgs')
    for case in manifest['g3_mgs']:
        i = case['id']
        Y = g[f'c{i}_Y'] if case['hasY'] else None
        out = ctx.mgs(g[f'c{i}_X'], Y)
        ref = g[f'c{i}_out']
        assert out.shape == ref.shape, case
        np.testing.assert_allclose(out, ref, atol=1e-12)


def test_mgs_properties(ctx):
    """orthonormal output, orthogonal to Y, duplicate columns dropped (reference test_math.py)."""
    from sella_amd.utilities.math import modified_gram_schmidt
    rng = np.random.RandomState(4)
    n = 40 if ctx.backend == 'emu' else 600
    X = rng.normal(size=(n, 6))
    Y = rng.normal(size=(n, 5))
    X[:, 3] = X[:, 1]
    out = modified_gram_schmidt(X, Y)
    assert out.shape == (n, 5)
    np.testing.assert_allclose(out.T @ out, np.eye(5), atol=1e-13)
    np.testing.assert_allclose(out.T @ Y, 0, atol=1e-12)
    assert modified_gram_schmidt(X[:, :0]).shape == (n, 0)
    assert modified_gram_schmidt(np.zeros((n, 2))).shape == (n, 0)


def test_qr_thin(ctx):
    rng = np.random.RandomState(5)
    shapes = [(5, 5), (9, 4), (40, 17), (130, 33)]
    if ctx.backend == 'hip':
        shapes += [(3000, 300), (1024, 1024)]
    for m, n in shapes:
        A = rng.normal(size=(m, n))
        Q, R = ctx.qr_thin(A)
        assert np.abs(Q @ R - A).max() < 1e-12 * m
        assert np.abs(Q.T @ Q - np.eye(n)).max() < 1e-13 * m
        assert np.abs(np.tril(R, -1)).max() == 0
        Qr, Rr = np.linalg.qr(A)          # same Householder sign convention as LAPACK
        np.testing.assert_allclose(np.abs(R), np.abs(Rr), atol=1e-11 * m)


def test_accelerator_seam_by_name(ctx):
    """The five functions of the reference's accelerator seam (sella/_gpu.py:38-132), called BY THESE NAMES with the
    reference's numpy-in / numpy-out contract: `to_gpu`, `gpu_eigh(A, A_gpu=None)`, `gpu_eigh_t(A_gpu)`,
    `gpu_qr(A)`, `gpu_project(H, U, H_gpu=None)` — and `_gpu_ok`."""
    from sella_amd import _gpu
    from sella_amd._gpu import _gpu_ok, gpu_eigh, gpu_eigh_t, gpu_project, gpu_qr, to_gpu
    assert set(_gpu.__all__) >= {'to_gpu', 'gpu_eigh', 'gpu_eigh_t', 'gpu_qr', 'gpu_project'}
    rng = np.random.RandomState(9)
    n, m = 48, 11
    A = rng.normal(size=(n, n))
    A = 0.5 * (A + A.T)
    A0 = A.copy()
    assert _gpu_ok(n)
    h = to_gpu(A)
    np.testing.assert_array_equal(h.numpy(), A)                     # upload / download round trip
    w, V = gpu_eigh(A)
    wl = np.linalg.eigvalsh(A)
    np.testing.assert_allclose(w, wl, atol=1e-12 * n)
    np.testing.assert_allclose(A @ V, V * w, atol=1e-11 * n)
    np.testing.assert_allclose(V.T @ V, np.eye(n), atol=1e-12 * n)
    w2, V2 = gpu_eigh(None, A_gpu=h)                                # cached device copy (linalg.py:183-207)
    np.testing.assert_allclose(w2, w, atol=1e-13)
    w3, Vd, Vtd = gpu_eigh_t(h)
    np.testing.assert_allclose(w3, w, atol=1e-13)
    np.testing.assert_allclose(Vd.numpy(), Vtd.numpy().T, atol=0)
    B = rng.normal(size=(n, m))
    Q, R = gpu_qr(B)
    assert Q.shape == (n, m) and R.shape == (m, m)
    np.testing.assert_allclose(Q @ R, B, atol=1e-12 * n)
    np.testing.assert_allclose(Q.T @ Q, np.eye(m), atol=1e-13 * n)
    U = np.linalg.qr(rng.normal(size=(n, m)))[0]
    np.testing.assert_allclose(gpu_project(A, U), U.T @ A @ U, atol=1e-12 * n)
    np.testing.assert_allclose(gpu_project(None, U, H_gpu=h), U.T @ A @ U, atol=1e-12 * n)
    np.testing.assert_array_equal(A, A0)                            # inputs are never mutated (_gpu.py:64)


# ---- sella_amd.utilities.math on its own terms -----------------------------------------------------------------------------
@pytest.mark.parametrize('shape,rank', [((4, 4), 4), ((60, 5), 5), ((7, 30), 7), ((40, 6), 4)])
def test_pseudo_inverse_factors_and_rank(ctx, shape, rank):
    """`pseudo_inverse` (utilities/math.pyx:17-41): the factors reproduce the matrix on its numerical range, the inverse is
    the Moore-Penrose one (the four Penrose conditions), and singular values below `eps` are counted out."""
    from sella_amd.utilities.math import pseudo_inverse
    rng = np.random.RandomState(37)
    A = rng.standard_normal((shape[0], rank)) @ rng.standard_normal((rank, shape[1]))
    U, s, VT, Ainv, nsing = pseudo_inverse(A.copy(), eps=1e-9)
    assert nsing == rank and np.all(s[:nsing] > 1e-9)
    np.testing.assert_allclose((U[:, :nsing] * s[:nsing]) @ VT[:nsing], A, atol=1e-10)
    for lhs, rhs in ((A @ Ainv @ A, A), (Ainv @ A @ Ainv, Ainv), ((A @ Ainv).T, A @ Ainv), ((Ainv @ A).T, Ainv @ A)):
        np.testing.assert_allclose(lhs, rhs, atol=1e-9)


def test_gram_schmidt_spans_and_drops(ctx):
    """`modified_gram_schmidt` (utilities/math.pyx:74-159): orthonormal columns with the span of X (same projector as a
    QR of X), orthogonal to Y when a Y is given with the span of X's part outside Y, and a column that is a combination
    of earlier ones (or lies inside Y) is dropped instead of being normalised from roundoff."""
    from sella_amd.utilities.math import modified_gram_schmidt
    rng = np.random.RandomState(41)
    n = 80
    X = rng.standard_normal((n, 12))
    Q = modified_gram_schmidt(X)
    assert Q.shape == (n, 12)
    np.testing.assert_allclose(Q.T @ Q, np.eye(12), atol=1e-12)
    Qr = np.linalg.qr(X)[0]
    np.testing.assert_allclose(Q @ Q.T, Qr @ Qr.T, atol=1e-10)
    Y = rng.standard_normal((n, 5))
    Q2 = modified_gram_schmidt(X, Y)
    np.testing.assert_allclose(Q2.T @ Q2, np.eye(Q2.shape[1]), atol=1e-12)
    assert np.abs(Q2.T @ Y).max() < 1e-10
    Qy = np.linalg.qr(Y)[0]
    Xperp = X - Qy @ (Qy.T @ X)
    Qp = np.linalg.qr(Xperp)[0]
    np.testing.assert_allclose(Q2 @ Q2.T, Qp @ Qp.T, atol=1e-9)
    Xd = X.copy()
    Xd[:, 7] = 2.0 * Xd[:, 2] - 0.5 * Xd[:, 4]            # dependent on earlier columns
    Xd[:, 9] = Y @ rng.standard_normal(5)                  # inside Y
    assert modified_gram_schmidt(Xd).shape[1] == 11
    assert modified_gram_schmidt(Xd, Y).shape[1] == 10


def test_one_launch_gram_schmidt_agrees_with_the_sweep_by_sweep_launches(ctx):
    """`gs_small_kernel` (one workgroup: sweeps, norms and the accept / drop decisions of math.pyx:112-129 in one launch,
    default up to 2048 entries) against the launch-per-sweep form of the same routine: same columns kept, same vectors to
    roundoff — on well-conditioned blocks, duplicates, near-duplicates (a third sweep), zero columns and a block with Y."""
    from sella_amd.utilities.math import modified_gram_schmidt
    rng = np.random.RandomState(11)
    n = 70 if ctx.backend == 'emu' else 1500
    X = rng.normal(size=(n, 9))
    X[:, 2] = X[:, 0]                                        # dropped
    X[:, 4] = X[:, 1] + 1e-4 * rng.normal(size=n)            # survives after heavy cancellation: more than two sweeps
    X[:, 6] = 0.0
    Y = rng.normal(size=(n, 4))
    try:
        outs = []
        for opt in (0, 2048):
            ctx.set_option('gs_small', opt)
            outs.append((modified_gram_schmidt(X), modified_gram_schmidt(X, Y), modified_gram_schmidt(X[:, :1])))
        for a, b in zip(*outs):
            assert a.shape == b.shape
            np.testing.assert_allclose(a, b, atol=1e-10 if a.shape[1] > 1 else 1e-15)
            np.testing.assert_allclose(b.T @ b, np.eye(b.shape[1]), atol=1e-13)
        assert outs[1][0].shape == (n, 7) and outs[1][1].shape == (n, 7)
        np.testing.assert_allclose(outs[1][1].T @ Y, 0, atol=1e-12)
    finally:
        ctx.set_option('gs_small', 2048)
