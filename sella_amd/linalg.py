"""Operators of the hot path — drop-in for sella/linalg.py:14-353.

  NumericalHessian   finite-difference Hessian-vector products through the calculator
                     boundary (linalg.py:14-101); the projection products U v / U^T Av run on
                     the device when the basis is large.
  MatrixSum          operator sum (linalg.py:104-140)
  ApproximateHessian owner of the n x n approximate Hessian B (linalg.py:143-353).  B lives in
                     HBM (`_B_gpu`); its eigenvectors stay on the device both as columns and as
                     rows, ready for the Davidson preconditioner, the TS-BFGS |B| term and the
                     P-RFO step; numpy copies are produced lazily.
"""
import os

import numpy as np
from scipy.sparse.linalg import LinearOperator

from .device import DeviceMatrix, get_context
from .hessian_update import update_H
from .utilities.math import is_identity

_DEVICE_MIN = 64        # below this a projection basis is not worth a device round trip


class NumericalHessian(LinearOperator):
    """H v by finite differences of the gradient, `func(x) -> (f, g)`, around (x0, g0) with displacement eta;
    optionally seen through a basis Uproj (ntrue x n): v -> Uproj^T H Uproj v.  Every product is remembered — the
    displaced directions as columns of `Vs`, the differences as columns of `AVs` (both in the full space) — because
    `PES.diag` turns them into secant pairs for the Hessian update afterwards (peswrapper.py:545-553)."""
    dtype = np.dtype('float64')
    _SIGNIFICANT = 1e-4        # linalg.py:45-73: components / projections below this do not fix the orientation

    def __init__(self, func, x0, g0, eta, threepoint=False, Uproj=None):
        self.func, self.eta, self.threepoint, self.Uproj = func, eta, threepoint, Uproj
        self.x0, self.g0 = np.array(x0, dtype=np.float64), np.array(g0, dtype=np.float64)
        self.ntrue = self.x0.size
        self.calls = 0
        if Uproj is not None and Uproj.shape[0] != self.ntrue:
            raise ValueError('Uproj must have %d rows' % self.ntrue)
        n = self.ntrue if Uproj is None else Uproj.shape[1]
        super().__init__(self.dtype, (n, n))
        self._pairs = []                     # (direction, difference quotient), full space, in call order
        self._U_gpu = None
        # an identity basis (unconstrained Cartesian search, peswrapper.py:403) needs no products at all
        self._U_identity = Uproj is not None and is_identity(Uproj)
        if Uproj is not None and not self._U_identity and min(Uproj.shape) >= _DEVICE_MIN:
            self._U_gpu = get_context().upload(Uproj)

    # the reference keeps two growing (ntrue x k) arrays; callers only read them after the Davidson run
    def _stacked(self, which):
        if not self._pairs:
            return np.empty((self.ntrue, 0), dtype=self.dtype)
        return np.column_stack([p[which] for p in self._pairs])

    Vs = property(lambda self: self._stacked(0))
    AVs = property(lambda self: self._stacked(1))

    def _lift(self, v):
        if self.Uproj is None or self._U_identity:
            return v
        if self._U_gpu is not None:
            return get_context().symm_mm(self._U_gpu, v)
        return self.Uproj @ v

    def _restrict(self, w):
        if self.Uproj is None or self._U_identity:
            return w
        if self._U_gpu is not None:
            return get_context().tmatmul(self._U_gpu, w)
        return self.Uproj.T @ w

    def _orientation(self, v):
        """+1 / -1: which of +-v is displaced along.  Downhill if v has a gradient component, else towards the origin,
        else so that the first significant component is positive (a deterministic choice keeps H v reproducible when
        the same direction comes back with the other sign)."""
        for ref in (self.g0, self.x0):
            proj = v @ ref
            if abs(proj) > self._SIGNIFICANT:
                return -1.0 if proj > 0 else 1.0
        lead = np.flatnonzero(np.abs(v) > self._SIGNIFICANT)
        return -1.0 if lead.size and v[lead[0]] < 0 else 1.0

    def _matvec(self, v):
        self.calls += 1
        v = self._lift(np.asarray(v, dtype=np.float64).ravel())
        length = np.linalg.norm(v)
        if length < 1e-12:
            return np.zeros(self.shape[0])
        scale = self._orientation(v) * length            # v / scale is the unit displacement direction
        # operation order as in linalg.py:75-84: a last-bit change of the displaced point is amplified by 1 / eta
        ahead = self.func(self.x0 + self.eta * v / scale)[1]
        if self.threepoint:
            behind = self.func(self.x0 - self.eta * v / scale)[1]
            Av = scale * (ahead - behind) / (2 * self.eta)
        else:
            Av = scale * (ahead - self.g0) / self.eta
        self._pairs.append((v.copy(), Av))
        return self._restrict(Av)

    def __add__(self, other):
        return MatrixSum(self, other)

    def _transpose(self):
        return self


class MatrixSum(LinearOperator):
    """Sum of operators of one shape (linalg.py:104-140); dense terms are folded into a single array."""

    def __init__(self, *matrices):
        shape = matrices[0].shape
        for mat in matrices:
            if mat.shape != shape:
                raise ValueError('MatrixSum: shapes differ: %s vs %s' % (mat.shape, shape))
        super().__init__(np.result_type(*[mat.dtype for mat in matrices]), shape)
        arrays = [mat for mat in matrices if isinstance(mat, np.ndarray)]
        self.matrices = [mat for mat in matrices if not isinstance(mat, np.ndarray)]
        if arrays:
            self.matrices.append(np.sum(arrays, axis=0, dtype=self.dtype))

    def _matvec(self, v):
        return sum(np.asarray(mat.dot(v), dtype=self.dtype).ravel() for mat in self.matrices)

    def _transpose(self):
        return MatrixSum(*(mat.T for mat in self.matrices))

    def __add__(self, other):
        return MatrixSum(*self.matrices, other)


# Quasi-Newton updates of rank <= EIG_UPDATE_MAX_RANK carry the device eigendecomposition along by
# rank-one modifications; after EIG_UPDATE_REFRESH of them it is recomputed from scratch.  The accumulated
# rounding grows linearly and slowly — measured 1.5e-17 per modification in the orthogonality defect and
# 6e-17 in the residual (n = 48: 800 modifications, n = 320: 320 modifications, start lam0*I) — so the bound is
# a safety net (about 3e-14 at the limit), not a cost: at 64 it used to put a 55 ms eigh into every 32nd step
# of the 1024-atom search.  EIG_UPDATE_MAX_RANK = 0 restores "eigh after every update".
EIG_UPDATE_MAX_RANK = 8
EIG_UPDATE_REFRESH = 1024

# STRUCTURED eigendecomposition.  An approximate Hessian that starts uninitialised becomes lam0 * I + (its first
# update) (linalg.py:274-289) and stays "lam0 * I + rank r" through every later quasi-Newton update: r explicit eigenpairs
# and the eigenvalue lam0 on the complement of their span (csrc/eigh.hip, lr_lowrank_update).  Carrying THAT costs O(n r)
# per update instead of the O(n^2) passes of the dense form, the step families work on r + 1 modes instead of n
# (csrc/stepper.hip, sella_stepper_create_lr) and the Davidson preconditioner costs O(n r) per application.  Used from
# LR_MIN_DIM on while r <= LR_MAX_FRACTION * dim; then the decomposition goes dense (one device eigh) and stays so.
# Below a few hundred degrees of freedom the explicit rank reaches that limit within a handful of steps and the dense
# machinery is cheap anyway.  Measured at 3N = 768 (16 searches of 20 steps): 33 against 29 searches/s in one process,
# 103 against 85 with four worker processes (session r03p).  LR_MIN_DIM = None switches the structured form off; the
# tests lower it to exercise the structured path at emulation sizes.
LR_MIN_DIM = 256
if os.environ.get('SELLA_LR_MIN_DIM'):                 # measurement knob ('0' or 'none': off)
    _v = os.environ['SELLA_LR_MIN_DIM'].lower()
    LR_MIN_DIM = None if _v in ('0', 'none', 'off') else int(_v)
LR_MAX_FRACTION = 0.4


class ApproximateHessian(LinearOperator):
    def __init__(self, dim, ncart, B0=None, update_method='TS-BFGS', symm=2,
                 initialized=False):
        self.dim = dim
        self.ncart = ncart
        super().__init__(np.float64, (dim, dim))
        self.update_method = update_method
        self.symm = symm
        self.initialized = initialized
        self._B = None             # numpy copy (None while stale)
        self._B_gpu = None         # DeviceMatrix (None while not uploaded)
        self._B_stale = False      # the device matrix lags behind the structured decomposition (sella_opt_step)
        self._is_none = True
        self.version = 0           # bumped whenever B changes (cache key for projections)
        self._view = None          # (idx, basis, ApproximateHessian of B[idx][idx], version it is in step with)
        self._drop_eig()
        self.set_B(B0)

    # ---- storage -------------------------------------------------------------------------
    def _drop_eig(self):
        self._eig_age = 0
        self._evals = None
        self._evecs = None
        # (references only: a live DeviceStepper or a caller of device_eig() may still hold the matrices, and
        # handles are table indices — the DeviceMatrix finalizer frees them when the last holder lets go)
        self._evecs_gpu = None
        self._evecsT_gpu = None
        self._evals_gpu = None
        self._lr = None            # structured form: dict(Wt, r, mu, lam0)

    def _drop_dense_eig(self):
        lr = self._lr
        self._drop_eig()
        self._lr = lr

    def _drop_lr(self):
        """Leave the structured form for good: its eigenvector block is returned to the device pool now (not whenever the
        finalizer runs — it is capacity x dim doubles held exactly when the rank has grown to 0.4 dim)."""
        lr, self._lr = self._lr, None
        if lr is not None and lr.get('Wt') is not None:
            lr['Wt'].free()

    # ---- structured eigendecomposition ---------------------------------------------------------------------------
    def device_eig_lr(self):
        """dict(Wt DeviceMatrix (capacity x dim, leading r rows = explicit eigenvectors), r, mu (ascending), lam0) if
        the eigendecomposition is held in structured form, else None."""
        return None if self._is_none else self._lr

    def _lr_evals(self):
        lr = self._lr
        r = lr['r']
        return np.sort(np.concatenate((lr['mu'][:r], np.full(self.dim - r, lr['lam0']))))

    def lowest_evals(self, k):
        """The k lowest eigenvalues (what the re-diagonalisation schedule of `Sella.step` looks at, optimize.py:363-378)
        — from the structured form without expanding the (n - r)-fold eigenvalue; None for an unset Hessian."""
        if self._is_none:
            return None
        if self._lr is not None and self._evals is None:
            lr = self._lr
            r = lr['r']
            pool = np.concatenate((lr['mu'][:r], np.full(min(int(k), self.dim - r), lr['lam0'])))
            return np.sort(pool)[:k]
        return self.evals[:k]

    def _lr_reserve(self, extra):
        """Room for `extra` more explicit rows; False if the explicit rank would pass LR_MAX_FRACTION * dim (the caller
        then goes dense)."""
        lr = self._lr
        need = lr['r'] + extra
        if need > max(1, int(LR_MAX_FRACTION * self.dim)) and need > 8:
            return False
        cap = lr['Wt'].shape[0]
        if need > cap:
            ctx = get_context()
            newcap = min(self.dim, max(2 * cap, need + 64))
            Wt = ctx.zeros(newcap, self.dim)
            ctx.mat_copy_into(lr['Wt'], Wt, lr['r'])
            mu = np.zeros(newcap)
            mu[:lr['r']] = lr['mu'][:lr['r']]
            lr['Wt'].free()
            lr['Wt'], lr['mu'] = Wt, mu
        return True

    def _init_structured(self, S, Y):
        """First update of an uninitialised Hessian (linalg.py:274-289 with hessian_update.py:58-67): B = lam0 I + update,
        lam0 the geometric mean of |eig(S^T Ytilde)| — formed on the device together with its structured eigenpairs."""
        from .hessian_update import _small_eigh, symmetrize_Y
        ctx = get_context()
        n, k = self.dim, S.shape[1]
        Yt = symmetrize_Y(S, Y, self.symm)
        thetas = np.maximum(np.abs(_small_eigh(S.T @ Yt)[0]), 1e-12)
        lam0 = float(np.exp(np.average(np.log(thetas))))
        dB = ctx.zeros(n, n)
        ctx.mat_add_diag(dB, lam0)
        cap = min(n, 2 * k + 64)
        lr = dict(Wt=ctx.zeros(cap, n), r=0, mu=np.zeros(cap), lam0=lam0)
        ctx.update_h_lr(dB, S, Y, lr, method=self.update_method, symm=self.symm)
        self.set_B(dB)
        self._lr = lr

    @property
    def B(self):
        if self._is_none:
            return None
        if self._B is None:
            self._B = self._get_B_gpu().numpy()
        return self._B

    @B.setter
    def B(self, value):
        self.set_B(value)

    def _get_B_gpu(self):
        """Device mirror of B, uploaded lazily (linalg.py:197-207)."""
        if self._is_none:
            return None
        if self._B_gpu is None:
            self._B_gpu = get_context().upload(self._B)
        if self._B_stale:
            # the one-call optimizer step keeps (W, mu, lam0) only; the matrix is rebuilt for whoever asks for it
            lr = self._lr
            get_context().lr_materialize(self._B_gpu, lr['Wt'], lr['r'], lr['mu'], lr['lam0'])
            self._B_stale = False
        return self._B_gpu

    def set_B(self, target):
        self.version += 1
        self._view = None
        self._B_stale = False
        self._drop_eig()
        if self._B_gpu is not None:
            self._B_gpu.free()
        self._B_gpu = None
        if target is None:
            self._B = None
            self._is_none = True
            self.initialized = False
            return
        if isinstance(target, DeviceMatrix):
            assert target.shape == self.shape
            self._B, self._B_gpu = None, target
            self.initialized = True
        else:
            if np.isscalar(target):
                target = target * np.eye(self.dim)
            else:
                self.initialized = True
            target = np.asarray(target, dtype=np.float64)
            assert target.shape == self.shape
            self._B = target
        self._is_none = False

    # ---- eigendecomposition (lazy, device-resident) -------------------------------------------
    def _ensure_eigen_computed(self):
        if self._evals is not None or self._is_none:
            return
        w, V, Vt = get_context().eigh(self._get_B_gpu())
        self._evals = w
        self._evals_gpu = w
        self._evecs_gpu = V
        self._evecsT_gpu = Vt

    @property
    def evals(self):
        if self._lr is not None and self._evals is None and not self._is_none:
            return self._lr_evals()
        self._ensure_eigen_computed()
        return self._evals

    @evals.setter
    def evals(self, value):
        self._evals = value

    @property
    def evecs(self):
        self._ensure_eigen_computed()
        if self._evecs is None and self._evecs_gpu is not None:
            self._evecs = self._evecs_gpu.numpy()
        return self._evecs

    @evecs.setter
    def evecs(self, value):
        self._evecs = value

    def device_eig(self):
        """(evals numpy, evecs DeviceMatrix [columns], evecsT DeviceMatrix [rows]) or None."""
        self._ensure_eigen_computed()
        if self._evals is None:
            return None
        return self._evals, self._evecs_gpu, self._evecsT_gpu

    # ---- quasi-Newton update ----------------------------------------------------------------
    def update(self, dx, dg):
        """Perform a quasi-Newton update on B (linalg.py:274-304)."""
        if not self.initialized:
            self.initialized = True
            nc = self.ncart
            if (self._is_none and nc == self.dim and LR_MIN_DIM is not None and self.dim >= LR_MIN_DIM
                    and not (np.ndim(dx) == 1 and np.linalg.norm(dx) < 1e-8)):
                S0 = np.ascontiguousarray(dx[:, None] if np.ndim(dx) == 1 else dx, dtype=np.float64)
                Y0 = np.ascontiguousarray(dg[:, None] if np.ndim(dg) == 1 else dg, dtype=np.float64)
                if 2 * S0.shape[1] <= max(8, int(LR_MAX_FRACTION * self.dim)):
                    self._init_structured(S0, Y0)
                    return
            B = np.zeros(self.shape) if self._is_none else self.B.copy()
            B[:nc, :nc] = update_H(None, dx[:nc], dg[:nc], method=self.update_method,
                                   symm=self.symm)
            self.set_B(B)
            return
        if np.ndim(dx) == 1 and np.linalg.norm(dx) < 1e-8:
            return                                           # update_H returns B itself
        dB = self._get_B_gpu()
        need_eig = self.update_method in ('TS-BFGS', 'BFGS_auto')
        have_eig = self._evals is not None and self._evecs_gpu is not None
        S2 = np.ascontiguousarray(dx[:, None] if np.ndim(dx) == 1 else dx, dtype=np.float64)
        Y2 = np.ascontiguousarray(dg[:, None] if np.ndim(dg) == 1 else dg, dtype=np.float64)
        if self._lr is not None:
            if self._lr_reserve(2 * S2.shape[1]):
                view = self._view if (self._view is not None and self._view[3] == self.version) else None
                if view is not None:
                    idx, _, sub, _ = view
                    lrs = sub._lr
                    if lrs is not None and not sub._lr_reserve(2 * S2.shape[1]):
                        sub._get_B_gpu()                      # (its matrix brought up to date while the decomposition exists)
                        sub._drop_lr()                        # the view goes dense (eigh when next needed)
                        lrs = None
                    get_context().update_h_lr(dB, S2, Y2, self._lr, method=self.update_method, symm=self.symm,
                                              view=(sub._get_B_gpu(), idx, lrs))
                    sub._B = None
                    sub.version += 1
                    sub._drop_dense_eig()
                    self._view = (view[0], view[1], sub, self.version + 1)
                else:
                    self._view = None
                    get_context().update_h_lr(dB, S2, Y2, self._lr, method=self.update_method, symm=self.symm)
                self._B = None
                self.version += 1
                self.initialized = True
                self._drop_dense_eig()
                return
            self._drop_lr()                                   # rank no longer low: dense from here on
        if (need_eig or have_eig) and EIG_UPDATE_MAX_RANK > 0:
            # carry the eigendecomposition across the update (rank-one modifications on the device)
            # instead of paying a new eigh at the next step solve, linalg.py:174-231
            evals, V, Vt = self.device_eig()
            view = self._view if (self._view is not None and self._view[3] == self.version) else None
            if view is not None:
                # the projected Hessian of a pinned-coordinate constraint set (PES.get_HL_projected) receives
                # the same update restricted to its coordinates, and keeps its eigendecomposition the same way
                idx, _, sub, _ = view
                seig = sub.device_eig() if sub._evals is not None else None
                new_evals, nr, sub_evals, nrs = get_context().update_h_eig_view(
                    dB, S2, Y2, evals, V, Vt, sub._get_B_gpu(), idx,
                    *(seig if seig is not None else (None, None, None)),
                    method=self.update_method, symm=self.symm, max_rank=EIG_UPDATE_MAX_RANK)
                sub._B = None
                sub.version += 1
                if seig is None or nrs < 0 or sub._eig_age + nrs > EIG_UPDATE_REFRESH:
                    sub._drop_eig()
                else:
                    sub._eig_age += nrs
                    sub._evals = sub._evals_gpu = sub_evals
                    sub._evecs = None
                self._view = (view[0], view[1], sub, self.version + 1)
            else:
                self._view = None
                new_evals, nr = get_context().update_h_eig(dB, S2, Y2, evals, V, Vt, method=self.update_method,
                                                           symm=self.symm, max_rank=EIG_UPDATE_MAX_RANK)
            self._B = None
            self.version += 1
            self.initialized = True
            if nr < 0 or self._eig_age + nr > EIG_UPDATE_REFRESH:
                self._drop_eig()                # recomputed from scratch when next needed
            else:
                self._eig_age += nr
                self._evals = self._evals_gpu = new_evals
                self._evecs = None
            return
        eig = self.device_eig() if need_eig else None
        kw = {}
        if eig is not None:
            kw = dict(evals_gpu=eig[0], evecs_gpu=eig[1], evecsT_gpu=eig[2])
        # (B itself is only inspected for None-ness when a device mirror is supplied)
        update_H(False, S2, Y2, method=self.update_method, symm=self.symm, B_gpu=dB,
                 download=False, **kw)
        # the device matrix was updated in place: host copy and eigenpairs are stale
        self._B = None
        self.version += 1
        self._view = None
        self._drop_eig()
        self.initialized = True

    def principal_view(self, U):
        """U^T B U for a basis U made of columns of the identity, as an ApproximateHessian that `update` keeps
        in step with B (matrix and eigendecomposition): the projection of peswrapper.py:363-386 for constraints
        that pin single coordinates, without a fresh eigh at every step.  None if no view of U is in step."""
        v = self._view
        if v is not None and v[1] is U and v[3] == self.version:
            return v[2]
        return None

    def register_view(self, U, sub):
        from .utilities.math import selection_of
        sel = selection_of(U)
        idx = np.ascontiguousarray(U.argmax(axis=0) if sel is None else sel, dtype=np.int32)
        if len(idx) > 1 and not np.all(np.diff(idx) > 0):
            return                                   # not an ordered selection of coordinates
        self._view = (idx, U, sub, self.version)
        if self._lr is not None and sub._lr is None and sub._evals is None and self._lr['r'] <= 256:
            # structured eigendecomposition of the principal submatrix from that of B (`sella_lr_restrict`)
            m = len(idx)
            sub._lr = get_context().lr_restrict(self._lr, idx, min(m, self._lr['r'] + 64))

    def project(self, U):
        """Project B into the subspace spanned by the columns of U (linalg.py:306-317)."""
        m, n = U.shape
        assert m == self.dim
        if self._is_none:
            Bproj = None
        else:
            Bproj = get_context().project(self._get_B_gpu(), U)
        return ApproximateHessian(n, 0, Bproj, self.update_method, self.symm)

    def asarray(self):
        if not self._is_none:
            return self.B
        return np.eye(self.dim)

    def _matvec(self, v):
        if self._is_none:
            return v
        return get_context().symm_mm(self._get_B_gpu(), np.asarray(v, dtype=np.float64).ravel())

    def _rmatvec(self, v):
        return self.matvec(v)

    def _matmat(self, X):
        if self._is_none:
            return X
        return get_context().symm_mm(self._get_B_gpu(), X)

    def _rmatmat(self, X):
        return self.matmat(X)

    def __add__(self, other):
        initialized = self.initialized
        if isinstance(other, ApproximateHessian):
            initialized = initialized and other.initialized
            other = other.B
        if not self.initialized or other is None:
            tot = None
            initialized = False
        else:
            tot = self.B + other
        return ApproximateHessian(self.dim, self.ncart, tot, self.update_method, self.symm,
                                  initialized=initialized)
