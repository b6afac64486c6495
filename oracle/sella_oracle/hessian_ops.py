"""ORACLE (test infrastructure, never imported by the product).

Operators of the hot path restated from sella/linalg.py:14-353:
finite-difference Hessian-vector operator, operator sum and the approximate
Hessian container (CPU semantics only: no device mirrors).
"""
import numpy as np
from scipy.linalg import eigh

from .secant import update_H


class _Op:
    """Minimal stand-in for scipy's LinearOperator protocol used by the path
    (``shape``, ``dot`` on 1-D / 2-D input, ``T``, ``+``, ``-``)."""
    dtype = np.dtype('float64')

    def __init__(self, shape):
        self.shape = shape

    def _apply(self, v):
        raise NotImplementedError

    def dot(self, x):
        x = np.asarray(x)
        if x.ndim == 1:
            return self._apply(x)
        cols = [self._apply(x[:, j]) for j in range(x.shape[1])]
        return np.stack(cols, axis=1) if cols else np.zeros((self.shape[0], 0))

    matvec = dot

    @property
    def T(self):
        return self._transposed()

    def _transposed(self):
        return self

    def __add__(self, other):
        return OperatorSum(self, other)

    def __sub__(self, other):
        return OperatorSum(self, -other)


class FiniteDifferenceHessian(_Op):
    """``NumericalHessian`` (linalg.py:14-101): H v ~ |v| (g(x+eta v^) - g0)/eta."""

    def __init__(self, func, x0, g0, eta, threepoint=False, Uproj=None):
        self.func = func
        self.x0 = x0.copy()
        self.g0 = g0.copy()
        self.eta = eta
        self.threepoint = threepoint
        self.calls = 0
        self.Uproj = Uproj
        self.ntrue = len(self.x0)
        n = self.ntrue if Uproj is None else Uproj.shape[1]
        super().__init__((n, n))
        self.Vs = np.empty((self.ntrue, 0))
        self.AVs = np.empty((self.ntrue, 0))

    def _apply(self, v):
        self.calls += 1
        v = np.asarray(v, dtype=float).ravel()
        if self.Uproj is not None:
            v = self.Uproj @ v
        # canonical displacement sign, linalg.py:59-73
        vg = v @ self.g0
        vx = v @ self.x0
        sign = 1.0
        if abs(vg) > 1e-4:
            sign = -1.0 if vg >= 0 else 1.0
        elif abs(vx) > 1e-4:
            sign = -1.0 if vx >= 0 else 1.0
        else:
            for vi in v:
                if vi > 1e-4:
                    sign = 1.0
                    break
                if vi < -1e-4:
                    sign = -1.0
                    break
        vnorm = np.linalg.norm(v)
        if vnorm < 1e-12:                                # linalg.py:76-80
            nout = self.shape[0]
            return np.zeros(nout)
        vnorm *= sign
        _, gp = self.func(self.x0 + self.eta * v / vnorm)
        if self.threepoint:
            _, gm = self.func(self.x0 - self.eta * v / vnorm)
            Av = vnorm * (gp - gm) / (2 * self.eta)
        else:
            Av = vnorm * (gp - self.g0) / self.eta
        self.Vs = np.hstack((self.Vs, v.reshape((self.ntrue, -1))))
        self.AVs = np.hstack((self.AVs, Av.reshape((self.ntrue, -1))))
        if self.Uproj is not None:
            Av = self.Uproj.T @ Av
        return Av


class OperatorSum(_Op):
    """``MatrixSum`` (linalg.py:104-140): dense terms are pre-summed."""

    def __init__(self, *terms):
        super().__init__(terms[0].shape)
        dense = None
        self.terms = []
        for t in terms:
            assert t.shape == self.shape, (t.shape, self.shape)
            if isinstance(t, np.ndarray):
                dense = t.astype(float) if dense is None else dense + t
            else:
                self.terms.append(t)
        if dense is not None:
            self.terms.append(dense)

    def _apply(self, v):
        w = np.zeros(self.shape[0])
        for t in self.terms:
            w = w + t.dot(v)
        return w

    def _transposed(self):
        return OperatorSum(*[t.T for t in self.terms])

    def __add__(self, other):
        return OperatorSum(*self.terms, other)


class QuasiNewtonHessian:
    """``ApproximateHessian`` (linalg.py:143-353), host semantics."""

    def __init__(self, dim, ncart, B0=None, update_method='TS-BFGS', symm=2,
                 initialized=False):
        self.dim = dim
        self.ncart = ncart
        self.shape = (dim, dim)
        self.update_method = update_method
        self.symm = symm
        self.initialized = initialized
        self._evals = None
        self._evecs = None
        self.set_B(B0)

    def set_B(self, target):                             # linalg.py:233-256
        self._evals = self._evecs = None
        if target is None:
            self.B = None
            self.initialized = False
            return
        if np.isscalar(target):
            target = target * np.eye(self.dim)
        else:
            self.initialized = True
        assert target.shape == self.shape
        self.B = target

    def _eig(self):
        if self._evals is None and self.B is not None:
            self._evals, self._evecs = eigh(self.B)

    @property
    def evals(self):
        self._eig()
        return self._evals

    @property
    def evecs(self):
        self._eig()
        return self._evecs

    def update(self, dx, dg):                            # linalg.py:274-304
        B = np.zeros(self.shape) if self.B is None else self.B.copy()
        if not self.initialized:
            self.initialized = True
            nc = self.ncart
            B[:nc, :nc] = update_H(None, dx[:nc], dg[:nc],
                                   method=self.update_method, symm=self.symm)
            self.set_B(B)
            return
        self.set_B(update_H(B, dx, dg, method=self.update_method,
                            symm=self.symm, lams=self.evals, vecs=self.evecs))

    def project(self, U):                                # linalg.py:306-317
        Bp = None if self.B is None else U.T @ self.B @ U
        return QuasiNewtonHessian(U.shape[1], 0, Bp, self.update_method,
                                  self.symm)

    def asarray(self):
        return np.eye(self.dim) if self.B is None else self.B

    def dot(self, X):
        return X if self.B is None else self.B @ X

    __matmul__ = dot

    def __add__(self, other):                            # linalg.py:340-353
        initialized = self.initialized
        if isinstance(other, QuasiNewtonHessian):
            initialized = initialized and other.initialized
            other = other.B
        if not self.initialized or other is None:
            tot, initialized = None, False
        else:
            tot = self.B + other
        return QuasiNewtonHessian(self.dim, self.ncart, tot,
                                  self.update_method, self.symm,
                                  initialized=initialized)
