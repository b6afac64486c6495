"""`modified_gram_schmidt` — drop-in for the reference's Cython routine
(sella/utilities/math.pyx:143-159) running on the device (sella_amd/csrc/gs.hip)."""
import numpy as np

from ..device import get_context


def modified_gram_schmidt(Xin, Yin=None, eps1=1.e-15, eps2=1.e-6, maxiter=100):
    Xin = np.asarray(Xin, dtype=np.float64)
    if Xin.shape[1] == 0:
        return Xin
    if Yin is not None and Yin.shape[1] == 0:
        Yin = None
    return get_context().mgs(Xin, Yin, eps1=eps1, eps2=eps2, maxiter=maxiter)
