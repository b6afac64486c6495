"""ORACLE (test infrastructure, never imported by the product).

Davidson / Rayleigh-Ritz partial diagonalisation, restating
sella/eigensolvers.py:9-153 with NumPy/SciPy.  The correction equation is
solved exactly as the reference does (dense LU of the bordered matrix per
iteration, eigensolvers.py:133-139) so this file is also what `bench.py`
times as the CPU baseline ("port").
"""
import numpy as np
from scipy.linalg import eigh, solve

from .mgs import modified_gram_schmidt
from .secant import symmetrize_Y


def exact(A, gamma=None, P=None):
    """Full diagonalisation, eigensolvers.py:9-28.

    For an operator A the dense matrix is rebuilt from n operator applications
    on the *rows* of P's eigenvector matrix (eigensolvers.py:23-25).
    """
    if isinstance(A, np.ndarray):
        lams, vecs = eigh(A)
    else:
        n = A.shape[0]
        if P is None:
            probes = np.eye(n)
        else:
            probes = exact(P)[1]
        dense = np.zeros((n, n))
        for i in range(n):
            p = probes[i]
            dense += np.outer(p, A.dot(p))
        dense = 0.5 * (dense + dense.T)
        lams, vecs = eigh(dense)
    return lams, vecs, lams[None, :] * vecs


def correction(V, Y, P, B, lams, vecs, shift, method='jd0', seeking=0):
    """``expand`` of the reference, eigensolvers.py:115-153."""
    n, k = V.shape
    R = Y @ vecs - B @ V @ vecs * lams[None, :]
    Psh = P - shift * B
    if method == 'lanczos':
        return R[:, seeking]
    if method == 'gd':
        return np.linalg.solve(Psh, R[:, seeking])
    if method == 'jd0_alt':
        v = V @ vecs[:, seeking]
        xr = solve(Psh, R[:, seeking])
        xv = solve(Psh, v)
        den = v.T @ xv
        if abs(den) < 1e-12:
            return xr
        return xv * (v.T @ xr / den) - xr
    if method == 'jd0':
        v = V @ vecs[:, seeking]
        aug = np.block([[Psh, v[:, None]], [v, 0]])
        rhs = np.zeros(n + 1)
        rhs[:n] = R[:, seeking]
        return solve(aug, -rhs)[:n]
    if method == 'mjd0_alt':
        Vr = V @ vecs
        xr = solve(Psh, R[:, seeking])
        xV = solve(Psh, Vr)
        alpha = solve(Vr.T @ xV, Vr.T @ xr)
        return solve(Psh, Vr @ alpha - R[:, seeking])
    if method == 'mjd0':
        Vr = V @ vecs
        aug = np.block([[Psh, Vr], [Vr.T, np.zeros((k, k))]])
        rhs = np.zeros(n + k)
        rhs[:n] = R[:, seeking]
        return solve(aug, -rhs)[:n]
    raise ValueError("Unknown diagonalization method {}".format(method))


def rayleigh_ritz(A, gamma, P, B=None, v0=None, vref=None, vreftol=0.99,
                  method='jd0', maxiter=None, trace=None, rng=None):
    """Davidson driver, eigensolvers.py:31-112.

    ``trace`` (optional list) receives one dict per outer iteration with the
    Ritz values and the expansion vector — used to pin step-for-step parity.
    ``rng`` seeds the otherwise unseeded random restart (eigensolvers.py:107).
    """
    n = A.shape[0]
    if B is None:
        B = np.eye(n)
    if maxiter is None:
        maxiter = 2 * n + 1
    if gamma <= 0:
        return exact(A, gamma, P)

    if v0 is not None:
        V = modified_gram_schmidt(v0.reshape((-1, 1)))
    else:
        Pl, Pv, _ = exact(P, 0)
        nneg = max(1, int(np.sum(Pl < 0)))
        V = modified_gram_schmidt(Pv[:, :nneg])
    AV = A.dot(V)

    while True:
        At = V.T @ symmetrize_Y(V, AV, symm=2)
        lams, vecs = eigh(At, V.T @ B @ V)               # generalized, lower
        nneg = max(1, int(np.sum(lams < 0)))
        AV = AV @ vecs                                   # rotate to Ritz basis
        V = V @ vecs
        k = V.shape[1]
        vecs = np.eye(k)
        if k >= min(n, maxiter):                         # :65-66
            return lams, V, AV

        Yt = symmetrize_Y(V, AV, symm=2)
        R = Yt[:, :nneg] - (B @ V)[:, :nneg] * lams[None, :nneg]
        Rnorm = np.linalg.norm(R, axis=0)

        if vref is not None:                             # :74-77
            if np.abs(V[:, 0] @ vref) > vreftol:
                return lams, V, AV

        seeking = None
        for i in range(len(Rnorm)):                      # :80-89
            if k == 1 or Rnorm[i] >= gamma * np.abs(lams[i]):
                seeking = i
                break
        if seeking is None:
            return lams, V, AV
        ri = R[:, seeking]
        theta = lams[seeking]

        t = correction(V, Yt, P, B, lams, vecs, theta, method, seeking)
        t = t / np.linalg.norm(t)
        if np.linalg.norm(t - V @ (V.T @ t)) < 1e-2:     # :93-95
            t = ri / np.linalg.norm(ri)
        t = modified_gram_schmidt(t[:, None], V)
        if t.shape[1] == 0:                              # :100-109
            for rj in R.T:
                t = modified_gram_schmidt(rj[:, None], V)
                if t.shape[1] == 1:
                    break
            else:
                draw = (np.random if rng is None else rng).normal(size=(n, 1))
                t = modified_gram_schmidt(draw, V)
                if t.shape[1] == 0:
                    return lams, V, AV
        if trace is not None:
            trace.append(dict(k=k, lams=lams.copy(), theta=theta,
                              seeking=seeking, rnorm=Rnorm.copy(),
                              t=t[:, 0].copy()))
        V = np.hstack([V, t])
        AV = np.hstack([AV, A.dot(t).reshape(n, -1)])
