"""Shared test helpers (recipes re-stated from the reference's tests/test_utils)."""
import math
from itertools import permutations

import numpy as np


def get_matrix(n, m, pd=False, symm=False, rng=None):
    """tests/test_utils/matrix_factory.py:3-15."""
    if rng is None:
        rng = np.random.RandomState(1)
    A = rng.normal(size=(n, m))
    if symm:
        A = 0.5 * (A + A.T)
    if pd:
        lams, vecs = np.linalg.eigh(A)
        A = vecs @ (np.abs(lams)[:, np.newaxis] * vecs.T)
    return A


def poly_factory(dim, order, rng=None):
    """Random multi-dimensional polynomial f, g, H (tests/test_utils/poly_factory.py:8-39)."""
    if rng is None:
        rng = np.random.RandomState(1)
    coeffs = []
    for i in range(order + 1):
        tmp = rng.normal(size=(dim,) * i)
        coeff = np.zeros_like(tmp)
        nperm = 0
        for permute in permutations(range(i)):
            coeff += np.transpose(tmp, permute)
            nperm += 1
        coeffs.append(coeff / (nperm * math.factorial(i)))

    def poly(x):
        res = 0
        grad = np.zeros_like(x)
        hess = np.zeros((dim, dim))
        for i, coeff in enumerate(coeffs):
            lastlast = last = None
            for _ in range(i):
                lastlast, last = last, coeff
                coeff = coeff @ x
            if last is not None:
                grad += i * last
            if lastlast is not None:
                hess += i * (i - 1) * lastlast
            res += coeff
        return res, grad, hess
    return poly


def colsign(V, Vref):
    s = np.sign(np.sum(V * Vref, axis=0))
    s[s == 0] = 1
    return V * s


class FakePES:
    """Duck-typed PES exposing exactly what restricted_step.py:28-62 reads
    (same construction as oracle/make_golden.py)."""
    int = None
    n_cell_dof = 0

    def __init__(self, Hcls, B, g, ncons=0, seed=0):
        n = len(g)
        rng = np.random.RandomState(seed)
        self.H = Hcls(n, n, B)
        self.g = g
        Q = np.linalg.qr(rng.normal(size=(n, n)))[0]
        self.Ucons, self.Ufree = Q[:, :ncons], Q[:, ncons:]
        self.scons = self.Ucons @ (1e-3 * rng.normal(size=ncons))
        self.Hcls = Hcls

    def get_g(self):
        return self.g.copy()

    def get_scons(self):
        return self.scons.copy()

    def get_H(self):
        return self.H

    def get_Unred(self):
        return np.eye(len(self.g))

    def get_Ufree(self):
        return self.Ufree

    def get_HL_projected(self, U):
        return self.Hcls(U.shape[1], 0, U.T @ self.H.B @ U)


class InternalCounts:
    """The block counts `MaxInternalStep._get_weights` reads from `pes.int` (restricted_step.py:217-243)."""

    def __init__(self, ntrans, nbonds, nangles, ndihedrals, nother, nrotations):
        self.ntrans, self.nbonds, self.nangles, self.ndihedrals = ntrans, nbonds, nangles, ndihedrals
        self.nother, self.nrotations = nother, nrotations
