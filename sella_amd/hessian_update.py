"""Quasi-Newton Hessian updates — drop-in for sella/hessian_update.py (same names, arguments
and numpy-in / numpy-out behaviour), computed by libsella_hip:

  * `symmetrize_Y(S, Y, symm)`                          hessian_update.py:27-37
  * `update_H(B, S, Y, method, symm, lams, vecs, ...)`   hessian_update.py:40-111

The n x n work (B S, |B| S through the eigenbasis, the fused symmetrise + rank-2k update) runs
on the device (sella_amd/csrc/update.hip); only k x k glue stays in Python.
"""
import numpy as np
from scipy.linalg import eigh as _small_eigh

from .device import DeviceMatrix, get_context

_METHODS = ('BFGS', 'TS-BFGS', 'PSB', 'DFP', 'SR1', 'Greenstadt', 'BFGS_auto')


def symmetrize_Y(S, Y, symm):
    if symm is None or S.shape[1] == 1:
        return Y
    if symm not in (0, 1, 2):
        raise ValueError("Unknown symmetrization method {}".format(symm))
    return get_context().symmetrize_y(S, Y, symm)


def update_H(B, S, Y, method='TS-BFGS', symm=2, lams=None, vecs=None,
             B_gpu=None, evals_gpu=None, evecs_gpu=None, evecsT_gpu=None, download=True):
    """Quasi-Newton update.  Returns B+ (numpy); when `B_gpu` (a DeviceMatrix holding B) is
    supplied it is updated in place and `(B+_numpy, B_gpu)` is returned, mirroring the
    reference's device-resident branch (hessian_update.py:70-75)."""
    if S.ndim == 1:
        if np.linalg.norm(S) < 1e-8:
            return B
        S = S[:, np.newaxis]
    if Y.ndim == 1:
        Y = Y[:, np.newaxis]
    if method not in _METHODS:
        raise ValueError('Unknown update method {}'.format(method))
    ctx = get_context()
    S = np.ascontiguousarray(S, dtype=np.float64)
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    n = S.shape[0]

    scaled_identity = None
    if B is None:
        # B := lam0 * I with lam0 the geometric mean |Ritz value| of S^T Ytilde   (:58-67)
        Yt = symmetrize_Y(S, Y, symm)
        thetas = np.maximum(np.abs(_small_eigh(S.T @ Yt)[0]), 1e-12)
        scaled_identity = float(np.exp(np.average(np.log(thetas))))
        dB = ctx.upload(scaled_identity * np.eye(n))
    elif B_gpu is not None:
        dB = B_gpu
    else:
        dB = ctx.upload(B)

    evals, dV, dVt = None, None, None
    own_eig = False
    if scaled_identity is not None:
        evals = np.array([scaled_identity])
    elif method in ('TS-BFGS', 'BFGS_auto'):
        if evals_gpu is not None and evecs_gpu is not None:
            evals, dV = np.asarray(evals_gpu), evecs_gpu
            dVt = evecsT_gpu if evecsT_gpu is not None else dV.transpose()
        elif lams is not None and vecs is not None:
            evals = np.asarray(lams, dtype=np.float64)
            dV = ctx.upload(vecs)
            dVt = dV.transpose()
            own_eig = True
        else:
            evals, dV, dVt = ctx.eigh(dB)                                   # (:77-78)
            own_eig = True

    ctx.update_h(dB, S, Y, method=method, symm=symm, evals=evals, evecs=dV, evecsT=dVt)
    Bplus = dB.numpy() if (download or B_gpu is None) else None
    if own_eig:
        dV.free()
        dVt.free()
    if B_gpu is not None:
        return Bplus, dB
    dB.free()
    return Bplus
