"""ORACLE (test infrastructure; never imported by the product): dense NumPy restatement of the Cartesian `PES` glue
and of the `Sella.step` loop, on top of the pinned restatements of this package (davidson, secant, hessian_ops,
stepsolve).  Follows sella/peswrapper.py:214-607 and sella/optimize/optimize.py:317-440 step by step.

PARITY UNPINNED: the reference classes need ASE + JAX, which are not installable here, so this file cannot be run
against the reference itself.  It is an independent second implementation — plain dense linear algebra, none of the
product's device mirrors, selection bases, carried eigendecompositions or fused root finder — against which the
product's trajectory (energies, gradients, trust radii, steps, Hessians) is compared step by step in
tests/test_pes_oracle.py.

Constraints: translations only (single-atom pins and centroids), i.e. the constraint Jacobian is constant and the
constraint Hessian vanishes (sella/internal.py:466-493) — what the README slab and the Cartesian tests use.
"""
import numpy as np
from scipy.linalg import eigh, qr

from .davidson import rayleigh_ritz
from .hessian_ops import FiniteDifferenceHessian, QuasiNewtonHessian
from .secant import symmetrize_Y
from .stepsolve import get_restricted_step


class TranslationConstraints:
    """Rows (indices, dim, target): mean of positions[indices, dim] == target (internal.py:466-493, :2861-2893)."""

    def __init__(self, atoms):
        self.atoms = atoms
        self.rows = []

    def fix_translation(self, index=None, dim=None):
        idx = np.arange(len(self.atoms)) if index is None else np.atleast_1d(index)
        for d in (range(3) if dim is None else (dim,)):
            self.rows.append((idx.copy(), d, float(self.atoms.positions[idx, d].mean())))

    def residual(self):
        return np.array([self.atoms.positions[i, d].mean() - t for i, d, t in self.rows])

    def jacobian(self):
        J = np.zeros((len(self.rows), 3 * len(self.atoms)))
        for r, (i, d, _) in enumerate(self.rows):
            J[r, 3 * i + d] = 1.0 / len(i)
        return J


def split_cons_subspace(drdx, tol_factor=1e-6):
    """peswrapper.py:51-69: rank-revealing pivoted QR of drdx^T."""
    n = drdx.shape[1]
    if drdx.shape[0] == 0:
        return np.zeros((n, 0)), np.eye(n)
    Q, R, _ = qr(drdx.T, mode='full', pivoting=True)
    diag = np.abs(np.diag(R))
    ncons = int(np.sum(diag > tol_factor * diag[0])) if diag.size and diag[0] > 0 else 0
    return Q[:, :ncons], Q[:, ncons:]


class OraclePES:
    """peswrapper.py:214-607 for a Cartesian search with translation constraints."""

    def __init__(self, atoms, cons, eta=1e-4, eigensolver='jd0', v0=None):
        self.atoms, self.cons, self.eta, self.eigensolver, self.v0 = atoms, cons, eta, eigensolver, v0
        self.dim = self.ncart = 3 * len(atoms)
        self.int = None
        self.n_cell_dof = 0
        self.H = QuasiNewtonHessian(self.dim, self.ncart, None)                     # :282-283
        self.H.initialized = False
        self.curr = dict(x=None, f=None, g=None, L=None)
        self.last = dict(self.curr)
        self.first_diag = True
        self.neval = 0
        self._saved = None

    # -- geometry ------------------------------------------------------------------------------------------
    def get_x(self):
        return self.atoms.positions.ravel().copy()

    def set_x(self, target):                                                      # :332-335
        diff = target - self.get_x()
        self.atoms.positions = target.reshape((-1, 3))
        g = self.curr.get('g')
        return diff, diff, (np.zeros_like(diff) if g is None else g)

    def save(self):
        self._saved = self.atoms.positions.copy()

    def restore(self):
        self.atoms.positions = self._saved.copy()

    # -- evaluation and bases ------------------------------------------------------------------------------------
    def eval(self):                                                               # :413-418
        self.neval += 1
        return self.atoms.get_potential_energy(), -self.atoms.get_forces().ravel()

    def _calc_eg(self, x):                                                        # :420-427
        keep = self.atoms.positions.copy()
        self.atoms.positions = x.reshape((-1, 3))
        out = self.eval()
        self.atoms.positions = keep
        return out

    def _update(self, feval=True):                                                # :440-465
        key = self.atoms.positions.tobytes()
        fresh = True
        if self.curr['x'] is not None and key == self.curr.get('key'):
            if feval and self.curr['f'] is None:
                fresh = False
            else:
                return False
        x = self.get_x()
        f, g = self.eval() if feval else (None, None)
        if fresh:
            self.last = dict(self.curr)
        self.curr.update(x=x, key=key, f=f, g=g)
        self._update_basis()
        return True

    def _update_basis(self):                                                      # :395-407, :467-481
        drdx = self.cons.jacobian()
        Ucons, Ufree = split_cons_subspace(drdx)
        self.curr.update(drdx=drdx, Ucons=Ucons, Ufree=Ufree, Unred=np.eye(self.dim))
        g = self.curr['g']
        self.curr['L'] = None if g is None else np.linalg.lstsq(drdx.T, g, rcond=None)[0]

    def get_f(self):
        self._update()
        return self.curr['f']

    def get_g(self):
        self._update()
        return self.curr['g'].copy()

    def get_Ufree(self):
        self._update(False)
        return self.curr['Ufree']

    def get_Ucons(self):
        self._update(False)
        return self.curr['Ucons']

    def get_Unred(self):
        self._update(False)
        return self.curr['Unred']

    def get_H(self):
        return self.H

    def get_scons(self):                                                          # :429-438
        Ucons = self.get_Ucons()
        return -Ucons @ np.linalg.lstsq(self.cons.jacobian() @ Ucons, self.cons.residual(), rcond=None)[0]

    def get_HL_projected(self, U):                                                # :363-386 (Hc = 0)
        B = self.H.B
        return QuasiNewtonHessian(U.shape[1], 0, None if B is None else U.T @ B @ U)

    # -- Davidson through the calculator ---------------------------------------------------------------------------
    def diag(self, gamma=0.1, threepoint=False, maxiter=None):                    # :508-556
        if self.curr['f'] is None:
            self._update(True)
        Ufree = self.get_Ufree()
        nfree = Ufree.shape[1]
        if nfree == 0:
            return
        Pobj = self.get_HL_projected(Ufree)
        if Pobj.B is None or self.first_diag:
            v0 = self.v0 if self.v0 is not None else self.get_g() @ Ufree
            if np.linalg.norm(v0) < 1e-12:
                v0 = None
        else:
            v0 = None
        P = np.eye(nfree) if Pobj.B is None else Pobj.asarray()
        Hproj = FiniteDifferenceHessian(self._calc_eg, self.get_x(), self.get_g(), self.eta, threepoint, Ufree)
        rayleigh_ritz(Hproj, gamma, P, v0=v0, method=self.eigensolver, maxiter=maxiter)
        Vs, AVs = Hproj.Vs, Hproj.AVs
        _, X = eigh(Vs.T @ symmetrize_Y(Vs, AVs, symm=2))
        self.H.update(Vs @ X, AVs @ X)
        self.first_diag = False

    def get_projected_forces(self):                                               # :558-562
        Ufree = self.get_Ufree()
        return -(Ufree @ (Ufree.T @ self.get_g())).reshape((-1, 3))

    def converged(self, fmax, cmax=1e-5):                                         # :564-568
        f1 = np.linalg.norm(self.get_projected_forces(), axis=1).max()
        c1 = np.linalg.norm(self.cons.residual())
        return (f1 < fmax) and (c1 < cmax), f1, c1

    def kick(self, dx, diag=False, **diag_kwargs):                                # :578-602
        x0, f0, g0 = self.get_x(), self.get_f(), self.get_g()
        B0 = self.H.asarray()
        dx_i, dx_f, g_par = self.set_x(x0 + dx)
        df_pred = None if B0 is None else g0 @ dx_i + 0.5 * dx_i @ B0 @ dx_i
        dg = self.get_g() - g_par
        df = self.get_f() - f0
        ratio = None if (df_pred is None or abs(df_pred) < 1e-14) else df / df_pred
        if self.last['x'] is not None and self.last['g'] is not None:             # :483-486
            self.H.update(dx_f, dg)
        if diag:
            self.diag(**diag_kwargs)
        return ratio


class OracleSella:
    """optimize.py:42-440 for order-1 / order-0 Cartesian searches (defaults table :20-39)."""
    _defaults = dict(minimum=dict(delta0=1e-1, sigma_inc=1.15, sigma_dec=0.90, rho_inc=1.035, rho_dec=100, method='qn',
                                  eig=False),
                     saddle=dict(delta0=0.1, sigma_inc=1.15, sigma_dec=0.65, rho_inc=1.035, rho_dec=5.0, method='prfo',
                                 eig=True))

    def __init__(self, atoms, cons, order=1, eta=1e-4, gamma=0.1, rs='ras', nsteps_per_diag=3, delta0=None):
        d = self._defaults['minimum' if order == 0 else 'saddle']
        self.pes = OraclePES(atoms, cons, eta=eta)
        self.ord, self.eig, self.method = order, d['eig'], d['method']
        self.sigma_inc, self.sigma_dec, self.rho_inc, self.rho_dec = d['sigma_inc'], d['sigma_dec'], d['rho_inc'], d['rho_dec']
        self.rs = get_restricted_step(rs)
        delta0 = d['delta0'] if delta0 is None else delta0
        self.delta = delta0 if rs in ('mis', 'ras') else delta0 * self.pes.get_Ufree().shape[1]      # :183-186
        self.delta_min = eta
        self.diagkwargs = dict(gamma=gamma, threepoint=False)
        self.nsteps_per_diag, self.nsteps_since_diag = nsteps_per_diag, 0
        self.initialized = False
        self.rho = 1.
        self.trace = []

    def step(self):
        pes = self.pes
        if not self.initialized:                                                  # :318-326
            pes.get_g()
            if self.eig:
                pes.diag(**self.diagkwargs)
                self.nsteps_since_diag = -1
            self.initialized = True
        pes._update_basis()
        pes.save()
        s, smag = self.rs(pes, self.ord, self.delta, method=self.method).get_s()   # :352-355
        if self.eig and self.nsteps_since_diag >= self.nsteps_per_diag:           # :363-378
            ev = pes.H.evals is None or bool((pes.get_HL_projected(pes.get_Unred()).evals[:self.ord] > 0).any())
        else:
            ev = False
        self.nsteps_since_diag = 0 if ev else self.nsteps_since_diag + 1
        rho = pes.kick(s, ev, **self.diagkwargs)
        if rho is not None:                                                       # :413-434
            if rho < 1. / self.rho_dec or rho > self.rho_dec:
                self.delta = max(smag * self.sigma_dec, self.delta_min)
            elif 1. / self.rho_inc < rho < self.rho_inc:
                self.delta = max(self.sigma_inc * smag, self.delta)
            self.rho = rho
        else:
            self.rho = 1.
        self.trace.append(dict(s=s.copy(), smag=smag, rho=rho, delta=self.delta, f=pes.get_f(), g=pes.get_g(),
                               B=None if pes.H.B is None else pes.H.B.copy(), rediag=ev))
