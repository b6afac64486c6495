"""Shared test helpers: own input generators (nothing here is taken from the reference's test utilities)."""
import numpy as np


def random_matrix(rng, rows, cols=None, symmetric=False, positive=False):
    """Gaussian entries; `symmetric` (square only) mirrors the upper triangle, `positive` shifts the spectrum of the
    symmetric matrix above zero (a Gershgorin-free way: subtract the lowest eigenvalue, add one)."""
    cols = rows if cols is None else cols
    M = rng.standard_normal((rows, cols))
    if symmetric or positive:
        M = np.triu(M) + np.triu(M, 1).T
    if positive:
        M = M + (1.0 - np.linalg.eigvalsh(M)[0]) * np.eye(rows)
    return M


class SmoothModel:
    """f(x) = 1/2 x.A x + sum_k a_k sin(u_k.x) + b_k/4 (u_k.x)^4 with closed-form gradient and Hessian: a smooth,
    non-quadratic function for the finite-difference operator tests (third and fourth derivatives both present)."""

    def __init__(self, dim, rng, nterms=6, quadratic_only=False):
        self.A = random_matrix(rng, dim, symmetric=True)
        self.U = rng.standard_normal((nterms, dim)) / np.sqrt(dim)
        self.a = np.zeros(nterms) if quadratic_only else 0.7 * rng.standard_normal(nterms)
        self.b = np.zeros(nterms) if quadratic_only else 0.3 * rng.standard_normal(nterms)

    def energy_gradient(self, x):
        p = self.U @ x
        f = 0.5 * x @ self.A @ x + np.sum(self.a * np.sin(p) + 0.25 * self.b * p ** 4)
        return f, self.A @ x + self.U.T @ (self.a * np.cos(p) + self.b * p ** 3)

    def hessian(self, x):
        p = self.U @ x
        return self.A + (self.U.T * (-self.a * np.sin(p) + 3.0 * self.b * p ** 2)) @ self.U


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
