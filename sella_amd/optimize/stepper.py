"""Step families — drop-in for sella/optimize/stepper.py:20-199 (same class names, synonyms,
alpha ranges and `get_s(alpha) -> (s, dsda)` contract).

MI355X formulation: every family is evaluated in the eigenbasis of the (projected) approximate
Hessian that `ApproximateHessian` keeps on the device.  RFO / P-RFO therefore never form or
diagonalise the (m+1) x (m+1) augmented matrix the reference rebuilds for every trial alpha
(stepper.py:128-131): it is a bordered diagonal matrix there, solved by a secular equation in
O(m) (sella_amd/csrc/stepper.hip), followed by one 2-right-hand-side device matvec.

Extension: `U` (optional, numpy (n, m)) — an orthonormal basis the Hessian was projected with.
When given, `g` is the unprojected gradient and `get_s` returns vectors in the unprojected
space (U @ s, U @ dsda), saving the caller two n x m host products per trial alpha.
"""
from typing import List, Optional, Tuple, Type

import numpy as np

from ..device import DeviceStepper, get_context
from ..linalg import ApproximateHessian
from ..utilities.math import is_identity, selection_of


class BaseStepper:
    alpha0: Optional[float] = None
    alphamin: Optional[float] = None
    alphamax: Optional[float] = None
    slope: Optional[float] = None          # sign of d|s|/dalpha
    newton_safe: bool = True
    synonyms: List[str] = []
    _kind: Optional[str] = None

    def __init__(self, g: np.ndarray, H: ApproximateHessian, order: int = 0,
                 d1: Optional[np.ndarray] = None, U: Optional[np.ndarray] = None) -> None:
        self.g = g
        self.H = H
        self.order = order
        self.d1 = d1
        self.U = U
        self._stepper_init()

    @classmethod
    def match(cls, name: str) -> bool:
        return name in cls.synonyms

    def _device_eig(self):
        """(evals, V, Vt) of H on the device; an uninitialised H is the identity
        (stepper.py:62-64 falls back to eigh(H.asarray()))."""
        ctx = get_context()
        eig = self.H.device_eig()
        if eig is None:
            m = self.H.shape[0]
            V = ctx.upload(np.eye(m))
            return np.ones(m), V, V.transpose()
        return eig

    def _stepper_init(self) -> None:
        ctx = get_context()
        U = self.U
        if U is not None and is_identity(U):
            U = None                                  # unconstrained: the basis is the identity
        self._sel = None
        g = self.g
        # structured eigendecomposition (lam0 I + rank r, linalg.ApproximateHessian): the family is evaluated on r + 1
        # modes instead of dim (`sella_stepper_create_lr`); not for a general projection basis, whose projected
        # Hessian is a dense matrix of its own
        lr = self.H.device_eig_lr() if (self._kind != 'qn_irc' and hasattr(self.H, 'device_eig_lr')) else None
        sel = selection_of(U) if U is not None else None
        if lr is not None and (U is None or sel is not None):
            if sel is not None:
                self._sel, self._nfull = sel, U.shape[0]
                g = np.ascontiguousarray(np.asarray(g, dtype=np.float64)[sel])
            self._dev = DeviceStepper(ctx, self._kind, None, None, None, g, self.order, lr=lr)
            return
        evals, V, Vt = self._device_eig()
        if U is not None:
            sel = selection_of(U)
            if sel is not None:
                # columns of the identity (constraints that pin single coordinates): U^T g and U s are index
                # operations on the host, the device works in the projected space only
                self._sel, self._nfull = sel, U.shape[0]
                g = np.ascontiguousarray(np.asarray(g, dtype=np.float64)[sel])
            else:
                # compose the projection with the eigenbasis once: (U V) is n x m
                VU = ctx.zeros(self.U.shape[0], V.shape[1])
                ctx.gemm(ctx.resident(self.U), V, VU)
                V, Vt = VU, VU.transpose()
        self._dev = DeviceStepper(ctx, self._kind, V, Vt, evals, g, self.order)

    def get_s(self, alpha: float) -> Tuple[np.ndarray, np.ndarray]:
        s, dsda = self._dev.get_s(alpha)
        if self._sel is None:
            return s, dsda
        s_full, ds_full = np.zeros(self._nfull), np.zeros(self._nfull)
        s_full[self._sel] = s
        ds_full[self._sel] = dsda
        return s_full, ds_full

    def solve_radius(self, measure, delta, tol, maxiter, scons=None, orthonormal=True, w=None, d1=None):
        """The restricted-step root find over this family in one device call (`sella_restricted_step`):
        -> (total step, reported measure, trial alphas).  Same alpha schedule as restricted_step.py:78-120."""
        sel = self._sel
        return self._dev.restricted_step(measure, delta, self.alpha0, self.alphamin, self.alphamax, self.slope,
                                         self.newton_safe, tol, maxiter, scons=scons, w=w, d1=d1,
                                         orthonormal=orthonormal, sel=sel, nfull=self._nfull if sel is not None else 0)


class NaiveStepper(BaseStepper):
    synonyms = []
    alpha0 = 0.5
    alphamin = 0.
    alphamax = 1.
    slope = 1.

    def __init__(self, dx: np.ndarray) -> None:
        self.dx = dx

    def get_s(self, alpha: float) -> Tuple[np.ndarray, np.ndarray]:
        return alpha * self.dx, self.dx


class QuasiNewton(BaseStepper):
    alpha0 = 0.
    alphamin = 0.
    alphamax = np.inf
    slope = -1
    _kind = 'qn'
    synonyms = [
        'qn', 'quasi-newton', 'quasi newton', 'quasi-newton', 'newton', 'mmf',
        'minimum mode following', 'minimum-mode following', 'dimer',
    ]


class QuasiNewtonIRC(QuasiNewton):
    """Quasi-Newton step family of the IRC inner loop (stepper.py:99-111):
    s(alpha) = -V (V^T g + alpha V^T d1) / (|lam| + alpha), d1 given in the projected space.
    Evaluated by the same device family object as the others (kind `qn_irc`); only V^T d1 is formed here."""
    synonyms = []
    _kind = 'qn_irc'

    def _stepper_init(self) -> None:
        if self.d1 is None:
            raise ValueError('QuasiNewtonIRC needs the accumulated displacement d1 (projected space)')
        Veig = self._device_eig()[1]                                    # eigenvectors of the projected Hessian
        BaseStepper._stepper_init(self)
        self._dev.set_d1hat(get_context().tmatmul(Veig, np.asarray(self.d1, dtype=np.float64)))


class RationalFunctionOptimization(BaseStepper):
    alpha0 = 1.
    alphamin = 0.
    alphamax = 1.
    slope = 1.
    newton_safe = False
    _kind = 'rfo'
    synonyms = ['rfo', 'rational function optimization']


class PartitionedRationalFunctionOptimization(RationalFunctionOptimization):
    _kind = 'prfo'
    synonyms = ['prfo', 'p-rfo', 'partitioned rational function optimization']


_all_steppers = [
    QuasiNewton,
    RationalFunctionOptimization,
    PartitionedRationalFunctionOptimization,
]


def get_stepper(name: str) -> Type[BaseStepper]:
    for stepper in _all_steppers:
        if stepper.match(name):
            return stepper
    raise ValueError("Unknown stepper name: {}".format(name))
