"""Re-entrancy of the host callbacks (include/sella_hip.h, "RE-ENTRANCY CONTRACT").

The Davidson operator of the reference is NumericalHessian._matvec (sella/linalg.py:39-95), which is
allowed to do anything; InternalPES's operator re-enters the library and creates device matrices on the
way.  Until round 3 the solver kept raw pointers into a handle table that could move (use after realloc)
and shared its scratch slots with whatever the callback called.  Here the callback deliberately does all
of that — uploads a hundred matrices, frees some, runs products, factorisations and a nested Davidson
solve on the same context — and the result must be BIT-IDENTICAL to the run with a quiet callback."""
import numpy as np
import pytest

from conftest import hessian_like


def _problem(n, seed=3):
    A, P, g = hessian_like(n, seed)
    w, Q = np.linalg.eigh(P)
    return A, P, g, w, Q


def _noisy_operator(ctx, A, log):
    """v -> A v, computed on the host, after hammering the library on the same context."""
    n = A.shape[0]
    rng = np.random.RandomState(11)
    M = rng.normal(size=(37, 37))
    M = 0.5 * (M + M.T)
    Bn, Pn, gn = hessian_like(24, 5)
    wn, Qn = np.linalg.eigh(Pn)
    keep = []

    def op(v):
        hs = [ctx.upload(np.eye(3) * (i + 1)) for i in range(100)]          # the handle table grows ...
        for h in hs[::2]:
            h.free()                                                          # ... and gets holes
        keep.extend(hs[1::2][:3])                                             # some survive the solve
        dM = ctx.upload(M)
        wm, Vm, Vmt = ctx.eigh(dM)                                               # scratch slots + scalar exchange
        np.testing.assert_allclose(wm, np.linalg.eigvalsh(M), atol=1e-11)
        X = rng.normal(size=(37, 5))
        np.testing.assert_allclose(ctx.symm_mm(dM, X), M @ X, atol=1e-11)
        Qm, Rm = ctx.qr_thin(rng.normal(size=(40, 7)))
        np.testing.assert_allclose(Qm.T @ Qm, np.eye(7), atol=1e-12)
        ctx.mgs(rng.normal(size=(n, 3)))
        # a nested solve of the same kind on the same context, resident operator and preconditioner
        ln, Vn, AVn, _ = ctx.davidson(ctx.upload(Bn), 24, gn, 0.1, method='jd0', maxiter=6, Pvecs=ctx.upload(Qn),
                                      PvecsT=ctx.upload(Qn.T.copy()), pevals=wn)
        np.testing.assert_allclose(AVn, Bn @ Vn, atol=1e-11)
        log.append(len(hs))
        return A @ v
    return op


@pytest.mark.parametrize('method', ['jd0', pytest.param('gd', marks=pytest.mark.emu_heavy),
                                    pytest.param('mjd0', marks=pytest.mark.emu_heavy)])
def test_davidson_callback_may_reenter_the_library(ctx, method):
    n = 96 if ctx.backend == 'emu' else 300
    A, P, g, w, Q = _problem(n)
    dQ, dQt = ctx.upload(Q), ctx.upload(Q.T.copy())
    kw = dict(method=method, maxiter=9, Pvecs=dQ, PvecsT=dQt, pevals=w)
    quiet = ctx.davidson(lambda v: A @ v, n, g, 0.1, **kw)
    log = []
    noisy = ctx.davidson(_noisy_operator(ctx, A, log), n, g, 0.1, **kw)
    assert len(log) == noisy[3] and noisy[3] == quiet[3]
    for a, b in zip(quiet[:3], noisy[:3]):
        assert a.shape == b.shape
        assert np.array_equal(a, b), np.abs(a - b).max()       # same arithmetic, same bits
    # and both are the operator's Ritz pairs
    lams, V, AV = noisy[:3]
    np.testing.assert_allclose(AV, A @ V, atol=1e-11)
    np.testing.assert_allclose(V.T @ V, np.eye(V.shape[1]), atol=1e-12)


def test_structured_preconditioner_survives_reentry(ctx):
    """r < n explicit eigenpairs (Pvecs n x r): the branch that dereferences Qt->ld after every callback — where the
    driver's GPU run of round 3 died with 'matrix must be 16-byte aligned with even leading dimension'."""
    n, r = (96, 5) if ctx.backend == 'emu' else (300, 9)
    A, P, g, w, Q = _problem(n, seed=4)
    W = np.ascontiguousarray(Q[:, :r])
    kw = dict(method='jd0', maxiter=8, Pvecs=ctx.upload(W), PvecsT=ctx.upload(W.T.copy()), pevals=w[:r], pscale=1.7)
    quiet = ctx.davidson(lambda v: A @ v, n, g, 0.1, **kw)
    log = []
    noisy = ctx.davidson(_noisy_operator(ctx, A, log), n, g, 0.1, **kw)
    for a, b in zip(quiet[:3], noisy[:3]):
        assert np.array_equal(a, b)


def test_callback_error_leaves_the_context_usable(ctx):
    n = 64
    A, P, g, w, Q = _problem(n, seed=6)
    calls = []

    def bad(v):
        calls.append(1)
        ctx.upload(np.eye(4))
        if len(calls) == 3:
            raise ValueError('calculator failed')
        return A @ v
    with pytest.raises(ValueError, match='calculator failed'):
        ctx.davidson(bad, n, g, 0.1, method='jd0', maxiter=8)
    # the call depth was unwound: the next solve runs on the outer working set and is correct
    lams, V, AV, _ = ctx.davidson(ctx.upload(A), n, g, 0.1, method='jd0', maxiter=8)
    np.testing.assert_allclose(AV, A @ V, atol=1e-11)
    ref = ctx.davidson(lambda v: A @ v, n, g, 0.1, method='jd0', maxiter=8)
    np.testing.assert_allclose(ref[0], lams, atol=1e-12)
