"""`modified_gram_schmidt` — drop-in for the reference's Cython routine
(sella/utilities/math.pyx:143-159) running on the device (sella_amd/csrc/gs.hip)."""
import weakref

import numpy as np

from ..device import get_context


def modified_gram_schmidt(Xin, Yin=None, eps1=1.e-15, eps2=1.e-6, maxiter=100):
    Xin = np.asarray(Xin, dtype=np.float64)
    if Xin.shape[1] == 0:
        return Xin
    if Yin is not None and Yin.shape[1] == 0:
        Yin = None
    return get_context().mgs(Xin, Yin, eps1=eps1, eps2=eps2, maxiter=maxiter)


# ---- identity bases -----------------------------------------------------------------------------
# The reduced / free bases of an unconstrained PES are identity matrices (peswrapper.py:334-339 in
# the reference builds np.eye(dim) at every geometry).  Here one read-only identity per dimension is
# shared, so that "is this basis the identity?" is an object comparison instead of an O(n^2) scan.
_IDENTITIES = {}


def pseudo_inverse(A, eps=1e-6):
    """Moore-Penrose pseudo-inverse through the singular value decomposition, contract of sella/utilities/math.pyx:219-236
    (`mppi`): -> (U, s, VT, Ainv, nsing) with A = U[:, :nsing] diag(s[:nsing]) VT[:nsing], singular values not above `eps`
    dropped.  The reference has no caller for it outside its tests (its InternalPES uses `_gpu_qr` + SVD directly); kept
    for API parity.  A tall matrix is first reduced by the device QR (`sella_qr_thin`), so the SVD is that of the small
    triangular factor; U is returned THIN (n x min(n, m)) where the reference allocates n x n and fills min(n, m) columns."""
    A = np.ascontiguousarray(np.asarray(A, dtype=np.float64))
    n, m = A.shape
    if n >= m and m > 0:
        Q, R = get_context().qr_thin(A)
        Ur, s, VT = np.linalg.svd(R)
        U = Q @ Ur
    else:
        U, s, VT = np.linalg.svd(A, full_matrices=False)
    nsing = int(np.sum(s > eps))
    Ainv = (VT[:nsing].T / s[:nsing]) @ U[:, :nsing].T
    return U, s[:nsing], VT, Ainv, nsing


def shared_identity(n):
    I = _IDENTITIES.get(n)
    if I is None:
        I = np.eye(n)
        I.flags.writeable = False
        _IDENTITIES[n] = I
    return I


_SELECTIONS = {}          # id(basis array) -> (weak reference, row index of every column)


def register_selection(U, idx):
    """Remember that U[:, k] = e_idx[k] (built by peswrapper._split_cons_subspace for constraints that pin single
    coordinates).  Products with such a basis are index operations; consumers ask `selection_of`.  The record lives
    exactly as long as the array (weak reference keyed on its identity): any number of replicas, driven from any
    number of host threads, keep their own bases without evicting each other."""
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    key = id(U)
    _SELECTIONS[key] = (weakref.ref(U, lambda _r, key=key: _SELECTIONS.pop(key, None)), idx)
    return U


def selection_of(U):
    """Row indices if U is a registered selection basis (same object), else None."""
    hit = _SELECTIONS.get(id(U))
    if hit is not None and hit[0]() is U:
        return hit[1]
    return None


def is_identity(U):
    """True if the 2-D array U is an identity matrix (fast for the shared instances)."""
    if U is None or np.ndim(U) != 2 or U.shape[0] != U.shape[1]:
        return False
    n = U.shape[0]
    if U is _IDENTITIES.get(n):
        return True
    if n == 0:
        return True
    if U[0, 0] != 1.0 or U[-1, -1] != 1.0:
        return False
    return bool(np.count_nonzero(U) == n and np.all(U.diagonal() == 1.0))
