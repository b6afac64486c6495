"""Restricted-step root find — drop-in for sella/optimize/restricted_step.py:11-253.

`RS(pes, order, delta, method=...).get_s() -> (s, smag)`: one-dimensional search over the
step-length parameter alpha of a step family (stepper.py) until the chosen measure of the
total step (Euclidean norm, largest per-atom displacement, largest weighted internal
displacement) equals the radius `delta`.  Same bracketing / Newton / bisection schedule and
tolerances as the reference (:64-66, :87-117); each trial alpha costs O(m) host arithmetic plus
one device matvec instead of the reference's dense eigh.
"""
import inspect
from typing import List, Optional

import numpy as np

from .stepper import BaseStepper, NaiveStepper, get_stepper


class BaseRestrictedStep:
    synonyms: List[str] = []

    def __init__(self, pes, order: int, delta: float, method: str = 'qn', tol: float = None,
                 maxiter: int = 1000, d1: Optional[np.ndarray] = None,
                 W: Optional[np.ndarray] = None):
        self.pes = pes
        self.delta = delta
        self.d1 = d1
        g0 = self.pes.get_g()
        self.scons = self.pes.get_scons()
        g = g0 + self.pes.get_H() @ self.scons                      # :35-37

        if inspect.isclass(method) and issubclass(method, BaseStepper):
            stepper = method
        else:
            stepper = get_stepper(method.lower())

        if self.cons(self.scons) - self.delta > 1e-8:               # infeasible correction, :44-48
            dx = self.pes.get_Unred().T @ self.scons
            self._lift = self.pes.get_Unred()
            self.stepper = NaiveStepper(dx)
            self.scons[:] *= 0
        else:
            Ufree = self.pes.get_Ufree()
            if W is not None:
                # P = Ufree^T W (restricted_step.py:50-53); P^T = W^T Ufree is the basis handed on.  A diagonal W
                # (the IRC's mass weighting, irc.py:174-175) is a row scaling, not an n^2 product.
                W = np.asarray(W, dtype=np.float64)
                if W.ndim == 1:
                    Ufree = W[:, None] * Ufree
                elif np.count_nonzero(W - np.diag(np.diagonal(W))) == 0:
                    Ufree = np.diagonal(W)[:, None] * Ufree
                else:
                    Ufree = W.T @ Ufree
            kw = {}
            if d1 is not None:
                kw['d1'] = np.linalg.lstsq(Ufree, d1, rcond=None)[0]          # :54-56
            # the stepper composes the basis with the eigenbasis on the device and returns
            # unprojected vectors, so eval() needs no further products
            self._lift = None
            self.stepper = stepper(g, self.pes.get_HL_projected(Ufree), order, U=Ufree, **kw)

        if tol is None:
            tol = 1e-10 if self.stepper.newton_safe else 1e-15
        self.tol = tol
        self.maxiter = maxiter
        self.alphas = []

    def cons(self, s, dsda=None):
        raise NotImplementedError

    def eval(self, alpha):
        s, dsda = self.stepper.get_s(alpha)
        if self._lift is not None:
            s, dsda = self._lift @ s, self._lift @ dsda
        stot = s + self.scons
        val, dval = self.cons(stot, dsda)
        self.alphas.append(alpha)
        return stot, val, dval

    def get_s(self):
        st = self.stepper
        alpha = st.alpha0
        s, val, dval = self.eval(alpha)
        if val < self.delta:
            assert val > 0.
            return s, val
        err = val - self.delta
        lower, upper = st.alphamin, st.alphamax

        for niter in range(self.maxiter):
            if abs(err) <= self.tol:
                break
            if np.nextafter(lower, upper) >= upper:
                break
            if err * st.slope > 0:
                upper = alpha
            else:
                lower = alpha
            newton = alpha - err / dval
            use_bisection = (np.isnan(newton) or newton <= lower or newton >= upper
                             or (niter > 4 and not st.newton_safe))
            if use_bisection:
                mid = (lower + upper) / 2.
                if np.isinf(mid):
                    alpha = alpha + max(1, 0.5 * alpha) * np.sign(mid)
                else:
                    alpha = mid
            else:
                alpha = newton
            s, val, dval = self.eval(alpha)
            err = val - self.delta
        else:
            raise RuntimeError("Restricted step failed to converge!")

        assert val > 0
        return s, self.delta

    @classmethod
    def match(cls, name):
        return name in cls.synonyms


class TrustRegion(BaseRestrictedStep):
    synonyms = ['tr', 'trust region', 'trust-region', 'trust radius', 'trust-radius']

    def cons(self, s, dsda=None):
        val = np.linalg.norm(s)
        if dsda is None:
            return val
        return val, dsda @ s / max(val, 1e-12)


class IRCTrustRegion(TrustRegion):
    """Trust sphere of the IRC inner loop: |(s + d1) * sqrt(m)| = dx (restricted_step.py:145-158)."""
    synonyms = []

    def __init__(self, *args, sqrtm=None, **kwargs):
        assert sqrtm is not None
        self.sqrtm = sqrtm
        self.d1 = kwargs.get('d1')
        TrustRegion.__init__(self, *args, **kwargs)
        assert self.d1 is not None

    def cons(self, s, dsda=None):
        s = (s + self.d1) * self.sqrtm
        if dsda is not None:
            dsda = dsda * self.sqrtm
        return TrustRegion.cons(self, s, dsda)


class RestrictedAtomicStep(BaseRestrictedStep):
    synonyms = ['ras', 'restricted atomic step']

    def __init__(self, pes, *args, **kwargs):
        if pes.int is not None:
            raise ValueError("Internal coordinates are not compatible with "
                             f"the {self.__class__.__name__} trust region method.")
        BaseRestrictedStep.__init__(self, pes, *args, **kwargs)

    def cons(self, s, dsda=None):
        s_mat = s.reshape((-1, 3))
        s_norms = np.linalg.norm(s_mat, axis=1)
        index = np.argmax(s_norms)
        val = s_norms[index]
        if dsda is None:
            return val
        return val, dsda.reshape((-1, 3))[index] @ s_mat[index] / max(val, 1e-12)


class MaxInternalStep(BaseRestrictedStep):
    synonyms = ['mis', 'max internal step']

    def __init__(self, pes, *args, wx=1., wb=1., wa=1., wd=1., wo=1., wc=1., **kwargs):
        if pes.int is None:
            raise ValueError("Internal coordinates are required for the "
                             f"{self.__class__.__name__} trust region method")
        self.wx, self.wb, self.wa, self.wd, self.wo, self.wc = wx, wb, wa, wd, wo, wc
        BaseRestrictedStep.__init__(self, pes, *args, **kwargs)

    def _get_weights(self):
        it = self.pes.int
        w = np.array([self.wx] * it.ntrans + [self.wb] * it.nbonds + [self.wa] * it.nangles
                     + [self.wd] * it.ndihedrals + [self.wo] * it.nother
                     + [self.wx] * it.nrotations)
        ncell = getattr(self.pes, 'n_cell_dof', 0)
        if ncell > 0:
            w = np.concatenate([w, [self.wc] * ncell])
        return w

    def cons(self, s, dsda=None):
        w = self._get_weights()
        assert len(w) == len(s)
        sw = np.abs(s * w)
        idx = np.argmax(sw)
        val = sw[idx]
        if dsda is None:
            return val
        return val, np.sign(s[idx]) * dsda[idx] * w[idx]


_all_restricted_step = [TrustRegion, RestrictedAtomicStep, MaxInternalStep]


def get_restricted_step(name):
    for rs in _all_restricted_step:
        if rs.match(name):
            return rs
    raise ValueError("Unknown restricted step name: {}".format(name))
