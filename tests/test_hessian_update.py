"""Quasi-Newton updates on the device: golden parity against the real reference for every method / B kind / k / symm
(g4, g5) and the properties the formulas promise (secant condition, symmetry, no-op threshold)."""
import numpy as np
import pytest

from conftest import load_golden
from helpers import random_matrix


METHODS = ('TS-BFGS', 'BFGS', 'PSB', 'DFP', 'SR1', 'Greenstadt', 'BFGS_auto')


@pytest.mark.parametrize('method', METHODS)
def test_secant_condition_and_symmetry(ctx, method):
    """What every formula of hessian_update.py:40-203 promises, checked on secant pairs Y = H S of a random symmetric H:
    B+ S = Y after the update (with the pairs symmetrised first, symm = 2: S^T Y symmetric makes all seven formulas exact
    for several pairs at once), an exactly symmetric result, the guess-free form (B = None), a vector pair equal to its
    one-column block, and a pair below the length threshold leaving B untouched (the same object comes back)."""
    from sella_amd.hessian_update import update_H
    rng = np.random.RandomState(17)
    n = 12
    for k in (1, 3):
        for positive in ((False, True) if method == 'BFGS_auto' else (False,)):
            B = random_matrix(rng, n, positive=positive, symmetric=True)
            H = random_matrix(rng, n, positive=positive, symmetric=True)
            S = random_matrix(rng, n, k)
            Y = H @ S
            for symm in ((0, 1, 2) if method == 'TS-BFGS' and k == 3 else (2,)):
                Bn = update_H(B, S, Y, method=method, symm=symm)
                assert np.abs(Bn @ S - Y).max() < 1e-8 * max(1.0, np.abs(Y).max())
                np.testing.assert_array_equal(Bn, Bn.T)
                B0 = update_H(None, S, Y, method=method, symm=symm)
                assert np.abs(B0 @ S - Y).max() < 1e-8 * max(1.0, np.abs(Y).max())
            if k == 1:
                np.testing.assert_allclose(update_H(B, S[:, 0], Y[:, 0], method=method), Bn, atol=1e-10, rtol=1e-10)
                assert update_H(B, 1e-13 * S[:, 0], 1e-13 * Y[:, 0], method=method) is B


def test_unknown_method(ctx):
    from sella_amd.hessian_update import update_H
    with pytest.raises(ValueError):
        update_H(np.eye(4), np.ones((4, 1)), np.ones((4, 1)), method='nope')


def test_golden_symmetrize(ctx, manifest):
    from sella_amd.hessian_update import symmetrize_Y
    g = load_golden('g4_symmetrize')
    for case in manifest['g4_symmetrize']:
        i = case['id']
        symm = None if case['symm'] < 0 else case['symm']
        out = symmetrize_Y(g[f'c{i}_S'], g[f'c{i}_Y'], symm)
        np.testing.assert_allclose(out, g[f'c{i}_out'], atol=1e-11, rtol=1e-11)


def test_golden_update(ctx, manifest):
    from sella_amd.hessian_update import update_H
    g = load_golden('g5_update_h')
    cases = manifest['g5_update_h']
    if ctx.backend == 'emu':
        cases = [c for c in cases if c['k'] != 8 or c['method'] in ('TS-BFGS', 'SR1')]
    for case in cases:
        i = case['id']
        B = None if case['B'] == 'none' else g[f'c{i}_B']
        out = update_H(B, g[f'c{i}_S'], g[f'c{i}_Y'], method=case['method'], symm=case['symm'])
        ref = g[f'c{i}_out']
        np.testing.assert_allclose(out, ref, atol=1e-10 * np.abs(ref).max(), rtol=0, err_msg=str(case))
    out = update_H(g['oned_B'], g['oned_s'], g['oned_y'])
    np.testing.assert_allclose(out, g['oned_out'], atol=1e-11)


def test_device_resident_update_returns_handle(ctx):
    """B_gpu branch (hessian_update.py:70-75): in-place on the device, (numpy, handle) returned."""
    from sella_amd.hessian_update import update_H
    rng = np.random.RandomState(7)
    n = 24
    B = random_matrix(rng, n, symmetric=True)
    S = rng.normal(size=(n, 3))
    Y = random_matrix(rng, n, symmetric=True) @ S
    ref = update_H(B, S, Y)
    dB = ctx.upload(B)
    out, handle = update_H(B, S, Y, B_gpu=dB)
    assert handle is dB
    np.testing.assert_allclose(out, ref, atol=1e-12)
    np.testing.assert_allclose(dB.numpy(), ref, atol=1e-12)


def _eig_check(B, w, V, Vt, tol):
    n = B.shape[0]
    scale = max(1.0, np.abs(B).max())
    Vn, Vtn = V.numpy(), Vt.numpy()
    np.testing.assert_array_equal(Vtn.T, Vn)
    assert np.all(np.diff(w) >= 0)
    np.testing.assert_allclose(w, np.linalg.eigvalsh(B), atol=tol * scale * n ** 0.5, rtol=0)
    assert np.abs(B @ Vn - Vn * w).max() <= tol * scale * n
    assert np.abs(Vn.T @ Vn - np.eye(n)).max() <= tol * n


@pytest.mark.parametrize('kind', ['dense', 'scaled identity', 'identity + low rank'])
def test_update_carries_eigendecomposition(ctx, kind):
    """sella_update_h_eig: after every quasi-Newton update the carried (evals, evecs) are those of
    the updated B (checked against LAPACK on the downloaded matrix) — for every update formula,
    single and block secant pairs, positive and negative rank-one weights, and the heavily deflated
    spectra an approximate Hessian starts from."""
    rng = np.random.RandomState(5)
    n = 36 if ctx.backend == 'emu' else 500
    if kind == 'dense':
        A = rng.normal(size=(n, n))
        B = A + A.T
    elif kind == 'scaled identity':
        B = 2.5 * np.eye(n)
    else:
        u = rng.normal(size=(n, 3))
        B = 1.7 * np.eye(n) + u @ u.T - np.outer(u[:, 0] + 1, u[:, 0] + 1)
    H = rng.normal(size=(n, n))
    H = H + H.T
    dB = ctx.upload(B)
    w, V, Vt = ctx.eigh(dB)
    methods = ['TS-BFGS', 'PSB', 'SR1', 'BFGS_auto', 'DFP', 'Greenstadt', 'TS-BFGS', 'TS-BFGS']
    total = 0
    for step, method in enumerate(methods * (1 if ctx.backend == 'emu' else 3)):
        k = 2 if step % 4 == 3 else 1
        S = rng.normal(size=(n, k)) * 0.1
        Y = H @ S + 0.05 * rng.normal(size=(n, k))
        w, nr = ctx.update_h_eig(dB, S, Y, w, V, Vt, method=method, symm=2, max_rank=8)
        assert 0 <= nr <= 4 * k
        total += nr
        _eig_check(dB.numpy(), w, V, Vt, 2e-13 * (1 + total))
    assert total > 0


def test_update_eig_rank_limit(ctx):
    """An update whose rank exceeds max_rank still updates B and reports nrank1 = -1."""
    from sella_amd.hessian_update import update_H
    rng = np.random.RandomState(6)
    n, k = 30, 5
    A = rng.normal(size=(n, n))
    B = A + A.T
    S, Y = rng.normal(size=(n, k)), rng.normal(size=(n, k))
    dB = ctx.upload(B)
    w, V, Vt = ctx.eigh(dB)
    w2, nr = ctx.update_h_eig(dB, S, Y, w, V, Vt, method='TS-BFGS', symm=2, max_rank=8)
    assert nr == -1
    np.testing.assert_array_equal(w2, w)
    np.testing.assert_allclose(dB.numpy(), update_H(B, S, Y, method='TS-BFGS', symm=2), atol=1e-11)


@pytest.mark.parametrize('n', [256, 333])
def test_rank2k_pass_unsymmetric_input(ctx, n):
    """The fused symmetrise + rank-2k pass (update.hip) on a size that is not a multiple of the tile and a
    slightly unsymmetric input: exactly symmetric result, equal to the oracle's update_H."""
    rng = np.random.RandomState(n)
    B0 = rng.normal(size=(n, n))
    B0 = B0 + B0.T + 0.01 * rng.normal(size=(n, n))          # slightly unsymmetric input: the pass symmetrises
    S = rng.normal(size=(n, 3))
    Y = rng.normal(size=(n, 3))
    dB = ctx.upload(B0)
    ctx.update_h(dB, S, Y, method='SR1', symm=2)
    out = dB.numpy()
    dB.free()
    np.testing.assert_array_equal(out, out.T)
    import oracle.sella_oracle as orc
    ref = orc.update_H(B0, S, Y, method="SR1", symm=2)      # B as given; the final symmetrisation is part of update_H
    np.testing.assert_allclose(out, ref, atol=1e-9 * np.abs(ref).max())


