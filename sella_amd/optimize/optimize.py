"""`Sella` — the optimizer driver, drop-in for sella/optimize/optimize.py:42-502 on the saddle-point
path: same constructor keywords and defaults table (:20-39), `run(fmax, steps)` inherited from the
ASE `Optimizer` (or the built-in equivalent when ASE is absent), the diagonalisation schedule and
trust-radius rules of `step()` (:359-434).

`internal=True` (or an `InternalCoordinates` object) selects `InternalPES` (geodesic steps in redundant
internal coordinates).  Options outside the saddle-point scope raise NotImplementedError:
`optimize_cell=True` (Cell*PES).
"""
import warnings
from time import localtime, strftime

import numpy as np

from ..atoms import Optimizer
from ..internal import Constraints
from ..peswrapper import PES
from .restricted_step import get_restricted_step

_default_kwargs = dict(
    minimum=dict(delta0=1e-1, sigma_inc=1.15, sigma_dec=0.90, rho_inc=1.035, rho_dec=100,
                 method='qn', eig=False),
    saddle=dict(delta0=0.1, sigma_inc=1.15, sigma_dec=0.65, rho_inc=1.035, rho_dec=5.0,
                method='prfo', eig=True),
)


class Sella(Optimizer):
    def __init__(self, atoms, restart=None, logfile='-', trajectory=None, master=None,
                 delta0=None, sigma_inc=None, sigma_dec=None, rho_dec=None, rho_inc=None,
                 order=1, eig=None, eta=1e-4, method=None, gamma=0.1, threepoint=False,
                 constraints=None, constraints_tol=1e-5, v0=None, internal=False,
                 append_trajectory=False, rs=None, nsteps_per_diag=3, diag_every_n=None,
                 hessian_function=None, optimize_cell=False, cell_mask=None, exp_cell_factor=None,
                 scalar_pressure=0.0, smax=None, allow_fragments=False, niggli=False,
                 refine_initial_hessian=False, save_hessian=None, exact_geodesic=None, **kwargs):
        # keyword set of the reference constructor (optimize.py:42-80).  The cell keywords (cell_mask,
        # exp_cell_factor, scalar_pressure, smax, niggli, refine_initial_hessian, save_hessian) only act with
        # optimize_cell=True there and are accepted and ignored here as there; the two that would change the
        # saddle-point path are refused.
        if optimize_cell:
            raise NotImplementedError('optimize_cell requires order=0 and is outside the saddle-point scope')
        if allow_fragments:
            raise NotImplementedError('allow_fragments needs the TRIC translation / rotation coordinates, which '
                                      'are outside the saddle-point scope (DESIGN.md section 7)')
        # the reference integrates the exact geodesic unless told otherwise (optimize.py:125)
        self.exact_geodesic = True if exact_geodesic is None else bool(exact_geodesic)
        default = _default_kwargs['minimum' if order == 0 else 'saddle']
        self.optimize_cell = False
        self.peskwargs = kwargs.copy()
        self.user_internal = internal
        self.initialize_pes(atoms, trajectory, order, eta, constraints, v0, internal,
                            hessian_function, **kwargs)
        if rs is None:
            rs = 'mis' if internal else 'ras'                                    # :178-179
        self.rs = get_restricted_step(rs)
        Optimizer.__init__(self, atoms, restart=restart, logfile=logfile, trajectory=None,
                           master=master)
        if delta0 is None:
            delta0 = default['delta0']
        if rs in ['mis', 'ras']:
            self.delta = delta0
        else:
            self.delta = delta0 * self.pes.get_Ufree().shape[1]                 # :183-186
        pick = lambda v, k: v if v is not None else default[k]                  # noqa: E731
        self.sigma_inc = pick(sigma_inc, 'sigma_inc')
        self.sigma_dec = pick(sigma_dec, 'sigma_dec')
        self.rho_inc = pick(rho_inc, 'rho_inc')
        self.rho_dec = pick(rho_dec, 'rho_dec')
        self.method = pick(method, 'method')
        self.eig = pick(eig, 'eig')
        self.ord = order
        self.eta = eta
        self.delta_min = self.eta
        self.constraints_tol = constraints_tol
        self.diagkwargs = dict(gamma=gamma, threepoint=threepoint)
        self.rho = 1.
        if self.ord != 0 and not self.eig:
            warnings.warn("Saddle point optimizations with eig=False will most likely fail!\n"
                          " Proceeding anyway, but you shouldn't be optimistic.")
        self.initialized = False
        self.xi = 1.
        self.nsteps_per_diag = nsteps_per_diag
        self.fmax = None
        self._last_converged = None
        self.nsteps_since_diag = 0
        self.diag_every_n = np.inf if diag_every_n is None else diag_every_n

    def initialize_pes(self, atoms, trajectory=None, order=1, eta=1e-4, constraints=None, v0=None,
                       internal=False, hessian_function=None, **kwargs):
        if internal:                                                              # :237-285
            from ..internal import InternalCoordinates
            from ..peswrapper import InternalPES
            if isinstance(internal, InternalCoordinates):
                if constraints is not None:
                    raise ValueError("Internals object and Constraint object cannot both be provided to Sella. "
                                     "Instead, you must pass the Constraints object to the constructor of the "
                                     "Internals object.")
                auto = False
            else:
                internal = InternalCoordinates.from_atoms(atoms, cons=constraints)
                auto = True
            self.internal = internal.copy()
            self.constraints = None
            self.pes = InternalPES(atoms, internals=internal, trajectory=trajectory, eta=eta, v0=v0,
                                   auto_find_internals=auto, hessian_function=hessian_function,
                                   exact_geodesic=getattr(self, 'exact_geodesic', True), **kwargs)
            self.trajectory = self.pes.traj
            return
        self.internal = None
        if constraints is None:
            constraints = Constraints(atoms)
        self.constraints = constraints
        self.pes = PES(atoms, constraints=constraints, trajectory=trajectory, eta=eta, v0=v0,
                       hessian_function=hessian_function, **kwargs)
        self.trajectory = self.pes.traj

    # ---- restartable state (the reference has no resume) --------------------------------
    def save_state(self, filename):
        """Everything the next step depends on besides the atoms: approximate Hessian, trust radius, schedule."""
        H = self.pes.H
        np.savez(filename, positions=self.pes.atoms.positions, B=(np.zeros((0, 0)) if H.B is None else H.B),
                 has_B=H.B is not None, H_initialized=H.initialized, delta=self.delta, rho=self.rho,
                 nsteps=self.nsteps, nsteps_since_diag=self.nsteps_since_diag, initialized=self.initialized,
                 first_diag=self.pes.first_diag)

    def load_state(self, filename):
        z = np.load(filename if str(filename).endswith('.npz') else str(filename) + '.npz')
        self.pes.atoms.positions = z['positions'].copy()
        self.pes.set_H(z['B'].copy() if bool(z['has_B']) else None, initialized=bool(z['H_initialized']))
        self.delta, self.rho = float(z['delta']), float(z['rho'])
        self.nsteps, self.nsteps_since_diag = int(z['nsteps']), int(z['nsteps_since_diag'])
        self.initialized = bool(z['initialized'])
        self.pes.first_diag = bool(z['first_diag'])

    def _predict_step(self):                                                     # :317-357
        if not self.initialized:
            self.pes.get_g()
            if self.eig:
                if self.pes.hessian_function is not None:
                    self.pes.calculate_hessian()
                else:
                    self.pes.diag(**self.diagkwargs)
                self.nsteps_since_diag = -1
            self.initialized = True
        self.pes.cons.disable_satisfied_inequalities()
        self.pes._update_basis()
        self.pes.save()
        x0 = self.pes.get_x()
        if self.pes.cons.has_inequalities():
            all_valid = False
            while not all_valid:
                s, smag = self.rs(self.pes, self.ord, self.delta, method=self.method).get_s()
                self.pes.set_x(x0 + s)
                all_valid = self.pes.cons.validate_inequalities()
                self.pes._update_basis()
                self.pes.restore()
            self.pes._update_basis()
        else:
            s, smag = self.rs(self.pes, self.ord, self.delta, method=self.method).get_s()
        return s, smag

    def step(self):                                                              # :359-434
        s, smag = self._predict_step()
        if self.nsteps_since_diag >= self.diag_every_n:
            ev = True
        elif self.eig and self.nsteps_since_diag >= self.nsteps_per_diag:
            if self.pes.H.evals is None:
                ev = True
            else:
                Unred = self.pes.get_Unred()
                ev = bool((self.pes.get_HL_projected(Unred).evals[:self.ord] > 0).any())
        else:
            ev = False
        if ev:
            self.nsteps_since_diag = 0
        else:
            self.nsteps_since_diag += 1
        rho = self.pes.kick(s, ev, **self.diagkwargs)
        if rho is not None:
            if rho < 1. / self.rho_dec or rho > self.rho_dec:
                self.delta = max(smag * self.sigma_dec, self.delta_min)
            elif 1. / self.rho_inc < rho < self.rho_inc:
                self.delta = max(self.sigma_inc * smag, self.delta)
            self.rho = rho
        else:
            self.rho = 1.

    def gradient_converged(self, gradient=None):
        return self.converged()

    def converged(self, forces=None):
        fmax = self.fmax if self.fmax is not None else 0.05
        result = self.pes.converged(fmax)
        self._last_converged = result
        return result[0]

    def log(self, forces=None):
        if self.logfile is None:
            return
        result = self._last_converged
        if result is None or len(result) != 3:
            result = self.pes.converged(self.fmax if self.fmax is not None else 0.05)
        _, fmax, cmax = result
        e = self.pes.get_f()
        T = strftime("%H:%M:%S", localtime())
        name = self.__class__.__name__
        if self.nsteps == 0:
            self.logfile.write(" " * len(name) + "{:>4s} {:>8s} {:>15s} {:>12s} {:>12s} {:>12s} {:>12s}\n"
                               .format("Step", "Time", "Energy", "fmax", "cmax", "rtrust", "rho"))
        self.logfile.write("{} {:>3d} {:>8s} {:>15.6f} {:>12.4f} {:>12.4f} {:>12.4f} {:>12.4f}\n"
                           .format(name, self.nsteps, T, e, fmax, cmax, self.delta, self.rho))
        try:
            self.logfile.flush()
        except (AttributeError, TypeError):
            pass
