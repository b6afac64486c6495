"""ORACLE (test infrastructure, never imported by the product).

Iterated modified Gram-Schmidt with drop tolerance, restating the reference's
Cython routine ``mgs`` (sella/utilities/math.pyx:74-140) and its Python wrapper
``modified_gram_schmidt`` (sella/utilities/math.pyx:143-159).

Semantics that matter:
  * a column is first normalised, then swept against every column of Y and
    every already accepted column of X, *renormalising after every single
    projection*; the product of those norms is ``normtot``;
  * the column is dropped as soon as ``normtot < eps2`` and accepted when
    ``0 <= 1 - normtot <= eps1``; otherwise the sweep repeats (<= maxiter);
  * Y is copied and orthonormalised *by itself* before X is swept against it.
"""
import numpy as np


def mgs_inplace(X, Y=None, eps1=1e-15, eps2=1e-6, maxiter=100):
    """In-place kernel; returns number of kept columns, -1 shape error, -2 no convergence.

    Follows sella/utilities/math.pyx:74-140 statement by statement.
    """
    n, nx = X.shape
    if Y is None:
        ny = 0
    else:
        if Y.shape[0] != n:
            return -1                                    # math.pyx:89-90
        ny = Y.shape[1]

    kept = 0
    for col in range(nx):                                # math.pyx:99
        if col != kept:
            X[:, kept] = X[:, col]                       # math.pyx:100-101
        X[:, kept] /= np.linalg.norm(X[:, kept])         # math.pyx:102-104
        converged = False
        dropped = False
        for _ in range(maxiter):                         # math.pyx:105
            normtot = 1.0
            for basis, nb in ((Y, ny), (X, kept)):       # math.pyx:107-126
                for j in range(nb):
                    bj = basis[:, j]
                    X[:, kept] -= (bj @ X[:, kept]) * bj
                    nrm = np.linalg.norm(X[:, kept])
                    normtot *= nrm
                    if normtot < eps2:
                        dropped = True
                        break
                    X[:, kept] /= nrm
                if dropped:
                    break
            if dropped:
                break                                    # math.pyx:116-117,127-128
            if 0.0 <= 1.0 - normtot <= eps1:             # math.pyx:129-131
                kept += 1
                converged = True
                break
        if not converged and not dropped:
            return -2                                    # math.pyx:132-133
    X[:, kept:] = 0.0                                    # math.pyx:136-138
    return kept


def modified_gram_schmidt(Xin, Yin=None, eps1=1e-15, eps2=1e-6, maxiter=100):
    """sella/utilities/math.pyx:143-159."""
    if Xin.shape[1] == 0:
        return Xin
    Yout = None
    if Yin is not None:
        Yout = np.array(Yin, dtype=float, order='C', copy=True)
        ny = mgs_inplace(Yout, None, eps1, eps2, maxiter)  # math.pyx:149-151
        Yout = Yout[:, :ny]
    Xout = np.array(Xin, dtype=float, order='C', copy=True)
    nx = mgs_inplace(Xout, Yout, eps1, eps2, maxiter)
    if nx < 0:
        raise RuntimeError("MGS failed.")                # math.pyx:157-158
    return Xout[:, :nx]
