"""ORACLE (test infrastructure, never imported by the product).

Secant-pair symmetrisation and multi-secant quasi-Newton Hessian updates,
restating sella/hessian_update.py:12-157 with NumPy/SciPy.
"""
import numpy as np
from scipy.linalg import eigh, lstsq, solve


# --------------------------------------------------------------------------
# symmetrisation of S^T Y                          (hessian_update.py:12-37)
# --------------------------------------------------------------------------
def _sequential_correction(S, Y):
    """``symmetrize_Y2`` (hessian_update.py:12-24): column-by-column minimal
    correction dY in span(S[:, :i]) making S^T (Y + dY) symmetric."""
    k = S.shape[1]
    dY = np.zeros_like(Y)
    YtS = Y.T @ S
    dYtS = np.zeros_like(YtS)
    StS = S.T @ S
    for i in range(1, k):
        rhs = YtS[i, :i].T - YtS[:i, i] - dYtS[:i, i]
        coef = np.linalg.lstsq(StS[:i, :i], rhs, rcond=None)[0]
        dY[:, i] = -S[:, :i] @ coef
        dYtS[i, :] = -StS[:, :i] @ coef
    return dY


def symmetrize_Y(S, Y, symm):
    """hessian_update.py:27-37."""
    if symm is None or S.shape[1] == 1:
        return Y
    if symm == 0:
        skew = np.tril(S.T @ Y - Y.T @ S, -1).T
        return Y + S @ lstsq(S.T @ S, skew)[0]
    if symm == 1:
        skew = np.tril(S.T @ Y - Y.T @ S, -1).T
        return Y + Y @ lstsq(S.T @ Y, skew)[0]
    if symm == 2:
        return Y + _sequential_correction(S, Y)
    raise ValueError("Unknown symmetrization method {}".format(symm))


# --------------------------------------------------------------------------
# update formulas: each returns Delta = B+ - B      (hessian_update.py:114-157)
# --------------------------------------------------------------------------
def _two_sided(U, J, S):
    """U J^T + J U^T - U (J^T S) U^T — common shape of the multi-secant family."""
    UJt = U @ J.T
    return (UJt + UJt.T) - U @ (J.T @ S) @ U.T


def delta_bfgs(B, S, Y):                             # :114-115
    BS = B @ S
    return Y @ solve(Y.T @ S, Y.T) - BS @ solve(S.T @ BS, S.T @ B)


def delta_ts_bfgs(B, S, Y, lams, vecs):              # :118-125
    J = Y - B @ S
    X1 = S.T @ Y @ Y.T
    absBS = vecs @ (np.abs(lams)[:, None] * (vecs.T @ S))
    X2 = S.T @ absBS @ absBS.T
    X = X1 + X2
    U = lstsq(X @ S, X)[0].T
    return _two_sided(U, J, S)


def delta_psb(B, S, Y):                              # :128-132
    J = Y - B @ S
    U = solve(S.T @ S, S.T).T
    return _two_sided(U, J, S)


def delta_dfp(B, S, Y):                              # :135-139
    J = Y - B @ S
    U = solve(S.T @ Y, Y.T).T
    return _two_sided(U, J, S)


def delta_sr1(B, S, Y):                              # :142-144
    J = Y - B @ S
    return J @ solve(J.T @ S, J.T)


def delta_greenstadt(B, S, Y):                       # :147-153
    J = Y - B @ S
    BS = B @ S
    U = solve(S.T @ BS, BS.T).T
    return _two_sided(U, J, S)


def update_H(B, S, Y, method='TS-BFGS', symm=2, lams=None, vecs=None):
    """Dispatcher, hessian_update.py:40-111 (CPU branch only)."""
    if S.ndim == 1:
        if np.linalg.norm(S) < 1e-8:
            return B                                  # same object, :49-52
        S = S[:, None]
    if Y.ndim == 1:
        Y = Y[:, None]

    Yt = symmetrize_Y(S, Y, symm)

    if B is None:                                     # :58-67
        thetas = np.maximum(np.abs(eigh(S.T @ Yt)[0]), 1e-12)
        lam0 = np.exp(np.average(np.log(thetas)))
        B = lam0 * np.eye(S.shape[0])

    if lams is None or vecs is None:
        lams, vecs = eigh(B)                          # :77-78

    if method == 'BFGS_auto':                         # :80-87
        method = 'TS-BFGS'
        if np.all(lams > 0):
            if np.all(eigh(S.T @ Yt, S.T @ S)[0] > 0):
                method = 'BFGS'

    if method == 'BFGS':
        D = delta_bfgs(B, S, Yt)
    elif method == 'TS-BFGS':
        D = delta_ts_bfgs(B, S, Yt, lams, vecs)
    elif method == 'PSB':
        D = delta_psb(B, S, Yt)
    elif method == 'DFP':
        D = delta_dfp(B, S, Yt)
    elif method == 'SR1':
        D = delta_sr1(B, S, Yt)
    elif method == 'Greenstadt':
        D = delta_greenstadt(B, S, Yt)
    else:
        raise ValueError('Unknown update method {}'.format(method))

    Bp = D + B                                        # :104
    return (Bp + Bp.T) * 0.5                          # :109
