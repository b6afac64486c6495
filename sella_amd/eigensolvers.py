"""Davidson / Rayleigh-Ritz partial diagonalisation — drop-in for sella/eigensolvers.py
(`exact` :9-28, `rayleigh_ritz` :31-112; the correction methods of `expand` :115-153 are
selected by the same `method` strings).  The whole loop runs inside libsella_hip
(sella_amd/csrc/davidson.hip); only the operator A may call back into Python, because the
finite-difference Hessian lives behind the calculator boundary.
"""
import numpy as np

from .device import DeviceFdOperator, DeviceMatrix, get_context
from .linalg import ApproximateHessian


def _is_identity(P):
    n = P.shape[0]
    if P.shape != (n, n) or P[0, 0] != 1.0:
        return False
    return np.count_nonzero(P) == n and bool(np.all(P.diagonal() == 1.0))


def _dense_from_operator(A, probes):
    """B = sum_i p_i (A p_i)^T over the ROWS p_i of `probes`, symmetrised (eigensolvers.py:22-26)."""
    n = A.shape[0]
    B = np.zeros((n, n))
    for i in range(n):
        p = probes[i]
        B += np.outer(p, np.asarray(A.dot(p)).ravel())
    return 0.5 * (B + B.T)


def exact(A, gamma=None, P=None):
    """Full diagonalisation on the device; returns (lams, vecs, lams * vecs)."""
    ctx = get_context()
    if isinstance(A, DeviceMatrix):
        dA, own = A, False
    elif isinstance(A, np.ndarray):
        dA, own = ctx.upload(A), True
    else:
        probes = np.eye(A.shape[0]) if P is None else exact(P)[1]
        dA, own = ctx.upload(_dense_from_operator(A, probes)), True
    lams, V, Vt = ctx.eigh(dA)
    vecs = V.numpy()
    V.free()
    Vt.free()
    if own:
        dA.free()
    return lams, vecs, lams[np.newaxis, :] * vecs


def _preconditioner(P):
    """-> dict(Pvecs, PvecsT, pevals, pscale), owned handles to free afterwards."""
    ctx = get_context()
    if P is None:
        return dict(pscale=1.0), []
    if isinstance(P, ApproximateHessian):
        if P._is_none:
            return dict(pscale=1.0), []
        lr = P.device_eig_lr()
        if lr is not None:
            # structured P = lam0 (I - W^T W) + W^T diag(mu) W: r explicit pairs, lam0 on the complement
            if lr['r'] == 0:
                return dict(pscale=float(lr['lam0'])), []
            Wt = ctx.mat_rows(lr['Wt'], 0, lr['r'])
            W = Wt.transpose()
            return dict(Pvecs=W, PvecsT=Wt, pevals=lr['mu'][:lr['r']].copy(), pscale=float(lr['lam0'])), [W, Wt]
        w, V, Vt = P.device_eig()
        return dict(Pvecs=V, PvecsT=Vt, pevals=w), []
    P = np.asarray(P, dtype=np.float64)
    if _is_identity(P):
        return dict(pscale=1.0), []
    dP = ctx.upload(P)
    w, V, Vt = ctx.eigh(dP)
    dP.free()
    return dict(Pvecs=V, PvecsT=Vt, pevals=w), [V, Vt]


def rayleigh_ritz(A, gamma, P, B=None, v0=None, vref=None, vreftol=0.99,
                  method='jd0', maxiter=None):
    """Same contract as the reference: returns (lams, V, AV) with the columns of V rotated
    into the Ritz basis.  `A` may be a numpy array, a DeviceMatrix or any object with
    `.shape` and `.dot(v)` (e.g. NumericalHessian); `P` a numpy array or an
    ApproximateHessian (whose cached device eigendecomposition is then reused)."""
    n, _ = A.shape
    if B is not None and not _is_identity(np.asarray(B)):
        return _rayleigh_ritz_metric(A, gamma, P, np.asarray(B, dtype=np.float64), v0, vref, vreftol, method, maxiter)
    if maxiter is None:
        maxiter = 2 * n + 1
    if gamma <= 0:
        return exact(A, gamma, P)

    ctx = get_context()
    pre, owned = _preconditioner(P)
    try:
        if v0 is not None:
            start = np.asarray(v0, dtype=np.float64).reshape((n, -1))
        else:
            # leading eigenvectors of P span the start block (eigensolvers.py:46-50)
            if 'Pvecs' in pre:
                P_lams = pre['pevals']
                nneg = max(1, int(np.sum(P_lams < 0)))
                start = pre['PvecsT'].numpy()[:nneg].T
            else:
                start = np.eye(n)[:, :1]
        own_A = False
        if isinstance(A, np.ndarray):
            op, own_A = ctx.upload(A), True
        elif isinstance(A, (DeviceMatrix, DeviceFdOperator)):
            op = A
        else:
            def op(v, _A=A):
                return np.asarray(_A.dot(v)).ravel()
        lams, V, AV, _ = ctx.davidson(op, n, np.ascontiguousarray(start), gamma, method=method,
                                      maxiter=maxiter, vref=vref, vreftol=vreftol, **pre)
        if own_A:
            op.free()
    finally:
        for h in owned:
            h.free()
    return lams, V, AV


class _Congruent:
    """x -> S (A (S x)) for an operator A that only offers `.dot` (S symmetric, resident)."""

    def __init__(self, A, S):
        self.A, self.S, self.shape = A, S, A.shape

    def dot(self, v):
        ctx = get_context()
        inner = ctx.symm_mm(self.S, np.asarray(v, dtype=np.float64).ravel())
        return ctx.symm_mm(self.S, np.asarray(self.A.dot(inner), dtype=np.float64).ravel())


def _rayleigh_ritz_metric(A, gamma, P, B, v0, vref, vreftol, method, maxiter):
    """Generalised problem A x = theta B x (sella/eigensolvers.py:35-36, 58, 70; no caller of the reference passes a metric).
    Reduced to the standard one by the congruence with S = B^-1/2 from the device eigendecomposition of B:
    (S A S) y = theta y, x = S y, and the standard device path runs on (S A S, S P S).  Returned as the reference returns
    them: Ritz values, V with V^T B V = I, AV = A V.
    Deviation: the reference keeps a Euclidean-orthonormal basis and solves a generalised Rayleigh-Ritz problem in it, so its
    search space after k steps is not this one and its convergence test |A x - theta B x| is |S^-1 (residual here)|;
    converged pairs agree, trajectories do not."""
    ctx = get_context()
    n = B.shape[0]
    dB = ctx.upload(0.5 * (B + B.T))
    wB, VB, VBt = ctx.eigh(dB)
    dB.free()
    if not wB[0] > 0.0:
        VB.free()
        VBt.free()
        raise ValueError('rayleigh_ritz: the metric B must be positive definite')
    Q = VB.numpy()
    VB.free()
    VBt.free()
    # S = B^-1/2 resident, B^1/2 applied on the host to single vectors / narrow panels only (O(n^2 k))
    dQs, dQt, S = ctx.upload(Q * wB ** -0.5), ctx.upload(np.ascontiguousarray(Q.T)), ctx.zeros(n, n)
    ctx.gemm(dQs, dQt, S)
    dQs.free()
    dQt.free()

    def sqrtB(X):
        return Q @ ((wB ** 0.5)[:, None] * (Q.T @ X.reshape((n, -1))))

    owned = [S]
    try:
        def congruent(M):
            dM = M if isinstance(M, DeviceMatrix) else ctx.upload(np.asarray(M, dtype=np.float64))
            tmp, out = ctx.zeros(n, n), ctx.zeros(n, n)
            ctx.gemm(S, dM, tmp)
            ctx.gemm(tmp, S, out)
            tmp.free()
            if dM is not M:
                dM.free()
            return out
        if isinstance(A, (np.ndarray, DeviceMatrix)):
            At = congruent(A)
            owned.append(At)
        else:
            At = _Congruent(A, S)
        Pt = None
        if P is not None:
            Pn = P.B if isinstance(P, ApproximateHessian) else np.asarray(P, dtype=np.float64)
            if Pn is not None:
                dPt = congruent(Pn)
                Pt = dPt.numpy()
                Pt = 0.5 * (Pt + Pt.T)
                dPt.free()
        v0t = None if v0 is None else sqrtB(np.asarray(v0, dtype=np.float64)).ravel()
        vreft = None if vref is None else ctx.symm_mm(S, np.asarray(vref, dtype=np.float64).ravel())
        lams, Vt_, AVt = rayleigh_ritz(At, gamma, Pt, None, v0t, vreft, vreftol, method, maxiter)
        V = ctx.symm_mm(S, np.ascontiguousarray(Vt_))
        return lams, V, sqrtB(np.ascontiguousarray(AVt))
    finally:
        for h in owned:
            h.free()


def block_davidson(A, nev, P=None, tol=1e-8, block=16, maxiter=500, maxvec=0, v0=None):
    """Lowest `nev` eigenpairs of a dense symmetric A by block Davidson (`sella_davidson_block`: `block` <= 16 new
    vectors per iteration, A streamed once per block on the matrix cores — BASELINE configs[4]).  The reference has
    no block method (one vector per iteration, eigensolvers.py:111-112); the result matches `exact(A)` truncated.
    P: approximate operator whose eigenbasis preconditions the corrections ('gd', eigensolvers.py:119-121),
    an ApproximateHessian, or None (diagonal of A).  Returns (lams (nev,), V (n, nev), residual norms (nev,))."""
    ctx = get_context()
    own = []
    if isinstance(A, DeviceMatrix):
        dA, diag = A, None
    else:
        A = np.asarray(A, dtype=np.float64)
        dA, diag = ctx.upload(A), np.ascontiguousarray(A.diagonal())
        own.append(dA)
    n = dA.shape[0]
    kw = {}
    if P is not None:
        pk, owned = _preconditioner(P)
        own += owned
        if 'Pvecs' in pk:
            kw = dict(Pvecs=pk['Pvecs'], PvecsT=pk['PvecsT'], pevals=pk['pevals'])
    if not kw and diag is not None:
        kw = dict(diag=diag)
    out = ctx.davidson_block(dA, n, nev, block=block, tol=tol, maxiter=maxiter, maxvec=maxvec, V0=v0, **kw)
    for m in own:
        m.free()
    if out['nconv'] < nev:
        raise RuntimeError(f'block_davidson: {out["nconv"]} of {nev} pairs converged in {out["niter"]} iterations')
    return out['lams'], out['V'], out['res']
