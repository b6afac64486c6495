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
        self.exact_geodesic = exact_geodesic is None or bool(exact_geodesic)
        self.optimize_cell = False
        self.user_internal, self.peskwargs = internal, dict(kwargs)
        self._user_constraints = constraints
        # what `LibrarySearch` would be built from (`run` hands the whole search to the library when it is covered)
        self._lib, self._lib_authoritative = None, False
        self._lib_kw = None
        if (not internal and restart is None and v0 is None and hessian_function is None and trajectory is None
                and logfile is None and master is None and not (set(kwargs) - {'proj_trans', 'proj_rot'})):
            self._lib_kw = dict(order=order, eta=eta, gamma=gamma, delta0=delta0, sigma_inc=sigma_inc, sigma_dec=sigma_dec,
                                rho_inc=rho_inc, rho_dec=rho_dec, rs=rs, method=method, eig=eig, threepoint=threepoint,
                                nsteps_per_diag=nsteps_per_diag, diag_every_n=diag_every_n, constraints=constraints,
                                proj_trans=kwargs.get('proj_trans'), proj_rot=kwargs.get('proj_rot'))
        own_traj = isinstance(trajectory, str)
        if own_traj:                                                              # :144-150
            from ..peswrapper import open_trajectory
            trajectory = open_trajectory(trajectory, atoms, append=append_trajectory)
        self.initialize_pes(atoms, trajectory, order, eta, constraints, v0, internal, hessian_function, **kwargs)
        Optimizer.__init__(self, atoms, restart=restart, logfile=logfile, trajectory=None, master=master)
        if own_traj:
            self.closelater(trajectory)

        # tunables: explicit keyword > table for the kind of stationary point sought (optimize.py:20-39, 120-123)
        given = dict(delta0=delta0, sigma_inc=sigma_inc, sigma_dec=sigma_dec, rho_inc=rho_inc, rho_dec=rho_dec,
                     method=method, eig=eig)
        table = _default_kwargs['minimum' if order == 0 else 'saddle']
        chosen = {key: (table[key] if val is None else val) for key, val in given.items()}
        for key in ('sigma_inc', 'sigma_dec', 'rho_inc', 'rho_dec', 'method', 'eig'):
            setattr(self, key, chosen[key])
        self.ord, self.eta, self.constraints_tol = order, eta, constraints_tol
        if order != 0 and not self.eig:
            warnings.warn("Saddle point optimizations with eig=False will most likely fail!\n"
                          " Proceeding anyway, but you shouldn't be optimistic.")

        # trust region: measure (:174-179) and initial radius — per degree of freedom for the Euclidean measure,
        # per atom / per coordinate for the max-norm ones (:183-186); never below eta (:198)
        rs_name = rs if rs is not None else ('mis' if internal else 'ras')
        self.rs = get_restricted_step(rs_name)
        per_dof = rs_name not in ('mis', 'ras')
        self.delta = chosen['delta0'] * (self.pes.get_Ufree().shape[1] if per_dof else 1)
        self.delta_min = eta
        self.rho, self.xi = 1., 1.

        # curvature schedule (:187-197)
        self.diagkwargs = {'gamma': gamma, 'threepoint': threepoint}
        self.nsteps_per_diag = nsteps_per_diag
        self.diag_every_n = diag_every_n if diag_every_n is not None else np.inf
        self.nsteps_since_diag = 0
        self.initialized = False
        self.fmax = None
        self._last_converged = None

    # ---- the whole search inside the library (sella_amd/search.py) -----------------------------------------------
    # `run()` on a fresh optimizer whose configuration the library loop covers (Cartesian PES, pinned coordinates at
    # most, a calculator that lives in the library, no log / trajectory / observers) hands the search to
    # `sella_search_run`: no interpreter between the force calls.  The library then holds the truth — geometry, trust
    # radius, approximate Hessian — until somebody looks: the first access to `self.pes` (so `step()`, `irun()`,
    # `converged()`, `log()`, `save_state()` too) brings it back (`_adopt`), approximate Hessian with its structured
    # eigendecomposition and the view of the pinned coordinates included, and the general driver continues from there.
    use_library_loop = True

    @property
    def pes(self):
        if self.__dict__.get('_lib_authoritative'):
            self._adopt()
        return self._pes

    @pes.setter
    def pes(self, value):
        self._pes = value

    def _library_run_applies(self):
        if not self.use_library_loop or self._lib_kw is None or self.observers or self.logfile is not None:
            return False
        if self._lib is not None:
            return True
        pes = self._pes
        if self.initialized or self.nsteps != 0 or type(pes) is not PES or pes.traj is not None or not pes.H._is_none:
            return False
        from ..search import LibrarySearch
        try:
            return LibrarySearch.applies(self.atoms, **self._lib_kw)
        except Exception:                                        # noqa: BLE001 — anything odd: the general driver
            return False

    def run(self, fmax=0.05, steps=100000000):
        if self._library_run_applies():
            from ..search import LibrarySearch, SearchLeftLibrary
            if self._lib is None:
                self._lib = LibrarySearch(self.atoms, **self._lib_kw)
            ls = self._lib
            if not np.array_equal(np.asarray(self.atoms.positions, dtype=np.float64).ravel(), ls.positions_flat()):
                # somebody moved the atoms between two runs: the library's geometry is no longer the truth.  The state
                # comes back as the library's point (`_adopt` caches it under the library's geometry), the PES notices
                # the change through its state hash like the reference's, and the general driver continues.
                self._lib_authoritative = True
                self._adopt()
                return Optimizer.run(self, fmax=fmax, steps=steps)
            self.fmax = fmax
            self.max_steps = self.nsteps + steps
            before = (ls.nsteps, ls.one_call_steps)
            try:
                conv = ls.run(fmax, steps)
                left = False
            except SearchLeftLibrary:
                conv, left = False, True
            taken = ls.nsteps - before[0]
            self.nsteps += taken
            self.fused_steps += ls.one_call_steps - before[1]
            self.delta, self.rho = ls.delta, ls.rho
            self._lib_authoritative = True
            if not left:
                return conv
            steps -= taken                                       # the rest with the general driver, from where it stands
        return Optimizer.run(self, fmax=fmax, steps=steps)

    def _adopt(self):
        """Bring the state of the library search back into this object's PES and drop the library search."""
        from ..linalg import ApproximateHessian
        ls, self._lib, self._lib_authoritative = self._lib, None, False
        pes = self._pes
        pending = ls.pending_pairs()
        st = ls.release_hessian()
        pes.neval = ls.neval
        # Energy and gradient belong to the LIBRARY's geometry.  If somebody moved the atoms in the meantime (`run()`
        # after a perturbation), the cached point must still be the library's — geometry, state hash and basis — so that
        # the first look at the PES sees 'moved', makes its force call at the new positions and keeps the library's
        # point as `last` (what the reference's PES does after an external move, peswrapper.py:440-474).
        x_lib = ls.positions_flat().reshape(-1, 3)
        user_pos = np.array(self.atoms.positions, dtype=np.float64)
        moved = not np.array_equal(user_pos, x_lib)
        if moved:
            self.atoms.positions = x_lib
        try:
            pes.curr.update(x=pes.get_x(), state_hash=pes._state_hash(), f=ls.energy, g=ls.gradient.copy())
            pes._update_basis()
            pes.last = dict(pes.curr)
        finally:
            if moved:                                    # whatever happened above, the user's geometry is what stays
                self.atoms.positions = user_pos
        pes.first_diag = st['first_diag']
        self.initialized = ls.initialized
        self.nsteps_since_diag = st['nsteps_since_diag']
        Hd = st['hessian']
        if Hd is not None:
            H = pes.H
            H.set_B(Hd['B'])
            H._lr = dict(Wt=Hd['Wt'], r=Hd['r'], mu=Hd['mu'], lam0=Hd['lam0'])
            H._B_stale = Hd['stale']
            v = Hd['view']
            if v is not None:
                sub = ApproximateHessian(len(v['idx']), 0, v['B'], H.update_method, H.symm)
                sub._lr = dict(Wt=v['Wt'], r=v['r'], mu=v['mu'], lam0=Hd['lam0'])
                sub._B_stale = v['stale']
                H._view = (np.ascontiguousarray(v['idx'], dtype=np.int32), pes.get_Ufree(), sub, H.version)
        if pending is not None:
            # the library left in the middle of a diagonalisation (its block update exceeds the structured form): the
            # force calls are spent and counted, the pairs come with the hand-over — applied here, this object is where
            # the reference is after PES.diag (peswrapper.py:545-553)
            pes.H.update(*pending)
            self.pairs_adopted = pending[0].shape[1]
        ls.close()

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

    # ---- one optimizer step ---------------------------------------------------------------------------------------
    # optimize.py:317-440 as three separable decisions — which step, whether to re-diagonalise, how the trust
    # radius reacts — around the single device call that solves the restricted step.
    def _first_use(self):
        """Initial gradient and curvature information (optimize.py:318-326)."""
        self.pes.get_g()
        if self.eig:
            if self.pes.hessian_function is not None:
                self.pes.calculate_hessian()
            else:
                self.pes.diag(**self.diagkwargs)
            self.nsteps_since_diag = -1
        self.initialized = True

    def _solve_step(self):
        """(s, smag) of the restricted step at the current geometry — `sella_restricted_step` behind the rs class."""
        return self.rs(self.pes, self.ord, self.delta, method=self.method).get_s()

    def _predict_step(self):
        if not self.initialized:
            self._first_use()
        pes = self.pes
        pes.cons.disable_satisfied_inequalities()
        pes._update_basis()
        pes.save()
        if not pes.cons.has_inequalities():
            return self._solve_step()
        # inequality constraints: re-solve until the trial geometry violates none that was switched off (:339-350)
        origin = pes.get_x()
        while True:
            s, smag = self._solve_step()
            pes.set_x(origin + s)
            feasible = pes.cons.validate_inequalities()
            pes._update_basis()
            pes.restore()
            if feasible:
                break
        pes._update_basis()
        return s, smag

    def _wants_diagonalisation(self):
        """The re-diagonalisation schedule (optimize.py:363-378): always after `diag_every_n` steps; after
        `nsteps_per_diag` steps only if the approximate Hessian has lost the `order` negative modes."""
        if self.nsteps_since_diag >= self.diag_every_n:
            return True
        if not (self.eig and self.nsteps_since_diag >= self.nsteps_per_diag):
            return False
        if self.pes.H._is_none:
            return True
        lowest = self.pes.get_HL_projected(self.pes.get_Unred()).lowest_evals(self.ord)
        return bool(np.any(lowest > 0))

    def _adapt_radius(self, rho, smag):
        """Trust radius from the ratio of actual to predicted change (optimize.py:413-434)."""
        if rho is None:
            self.rho = 1.
            return
        if not (1. / self.rho_dec <= rho <= self.rho_dec):
            self.delta = max(smag * self.sigma_dec, self.delta_min)
        elif 1. / self.rho_inc < rho < self.rho_inc:
            self.delta = max(self.sigma_inc * smag, self.delta)
        self.rho = rho

    def _rebuild_if_internals_degraded(self):
        """optimize.py:384-410: a step that drives an internal coordinate into a singular region (an angle near
        0 / pi, `check_for_bad_internals`) invalidates the coordinate system: build a fresh PES — new internals
        from the current geometry unless the user supplied them, new approximate Hessian, new initial
        diagonalisation — and skip this step's trust-radius update."""
        if not self.internal:
            return False
        pes = self.pes
        # both are index arrays (np.flatnonzero) or None: an array has no truth value, and [0] would read as False
        stale = getattr(pes, 'bad_int', None) is not None or pes.int.check_for_bad_internals() is not None
        if not stale:
            return False
        # the user's Constraints object goes along when the internals are regenerated from the geometry (with a
        # user-supplied InternalCoordinates object the constraints live inside it)
        from ..internal import InternalCoordinates
        cons = None if isinstance(self.user_internal, InternalCoordinates) else self._user_constraints
        self.initialize_pes(pes.atoms, trajectory=pes.traj, order=self.ord, eta=pes.eta, constraints=cons,
                            v0=None, internal=self.user_internal, hessian_function=pes.hessian_function,
                            **self.peskwargs)
        self.initialized = False
        self.rho = 1
        return True

    # ---- the step as ONE library call (`sella_opt_step`, csrc/optstep.hip) ----------------------------------------
    # Between two force calls the reference runs kick's model prediction and quasi-Newton update, the trust-radius
    # rule and the next restricted step as a few hundred interpreter-level operations; for the common configuration —
    # Cartesian PES, no constraints or constraints that pin single coordinates (all satisfied), approximate Hessian in
    # structured form, built-in step family and measure — they are one call that hands back the next step.  Anything
    # else (and every step that re-diagonalises) takes the general path below; both produce the same numbers
    # (tests/test_fused_step.py).
    use_fused_step = True
    fused_steps = 0                    # steps that took the one-call route (diagnostics, bench.py)

    def _fused_block(self):
        """The argument block of `sella_opt_step` for the current state, or None if this step needs the general path."""
        from ..device import CONSTRAINT_KINDS, STEPPER_KINDS, UPDATE_METHODS, OptStep
        from .restricted_step import RestrictedAtomicStep, TrustRegion
        from .stepper import _all_steppers, get_stepper
        pes = self.pes
        if not self.use_fused_step or type(pes) is not PES or not self.initialized:
            return None
        if self.rs not in (TrustRegion, RestrictedAtomicStep):
            return None
        family = self.method if isinstance(self.method, type) else get_stepper(self.method.lower())
        if family not in _all_steppers:
            return None
        H = pes.H
        if H._is_none or H._lr is None or H.update_method not in UPDATE_METHODS:
            return None
        cons = pes.cons
        if cons.has_inequalities() or pes._has_curved_constraints():
            return None
        drdx = pes.curr.get('drdx')
        if drdx is None:
            return None
        view = None
        if drdx.shape[0] > 0:
            hit = getattr(pes, '_pinned_basis', None)
            if pes._pinned() is None or hit is None or pes.curr.get('Ufree') is not hit[2] or np.any(pes.get_res()):
                return None
            v = H._view
            if v is None or v[1] is not hit[2] or v[3] != H.version or v[2]._lr is None:
                return None
            if not v[2]._lr_reserve(4):                        # two new rows + the work rows of the coordinate update
                return None
            view = (self._mirror(v[2]), v[0], v[2]._lr)
        if not H._lr_reserve(4):
            return None
        blk = getattr(self, '_opt_block', None)
        if blk is None or blk.c.n != pes.dim:
            blk = self._opt_block = OptStep(pes.dim)
        blk.set_hessian(self._mirror(H), H._lr, H.update_method, H.symm, view, stale=H._B_stale,
                        stale_sub=view is not None and H._view[2]._B_stale)
        c = blk.c
        c.stepper_kind, c.order, c.cons = STEPPER_KINDS[family._kind], int(self.ord), CONSTRAINT_KINDS[self.rs.measure]
        c.tol, c.maxiter = (1e-10 if family.newton_safe else 1e-15), 1000
        c.delta_min, c.sigma_inc, c.sigma_dec = self.delta_min, self.sigma_inc, self.sigma_dec
        c.rho_inc, c.rho_dec = self.rho_inc, self.rho_dec
        return blk

    @staticmethod
    def _mirror(H):
        """The device matrix of H as it is — possibly lagging behind the decomposition (`_B_stale`): the library call
        rebuilds it itself if its general route needs it."""
        if H._B_gpu is None:
            return H._get_B_gpu()
        return H._B_gpu

    def _step_fused(self, blk):
        from ..device import get_context
        pes = self.pes
        ahead = self.__dict__.pop('_proposed', None)
        if (ahead is not None and ahead[0] == pes.curr.get('state_hash') and ahead[1] == self.delta
                and ahead[2] == (id(pes.H), pes.H.version)):
            s, smag = ahead[3], ahead[4]
            pes.save()
        else:
            s, smag = self._predict_step()
        rediag = self._wants_diagonalisation()
        self.nsteps_since_diag = 0 if rediag else self.nsteps_since_diag + 1
        # PES.kick: the move and the force call stay here (calculator boundary) ...
        origin, f_old, g_old = pes.get_x(), pes.get_f(), pes.get_g()
        dx = pes.set_x(origin + s)[0]
        g_new, f_new = pes.get_g(), pes.get_f()
        # ... everything after it is the library's
        c = blk.c
        c.flags = blk.LEARN | (0 if rediag else blk.PROPOSE)
        blk.point('dx', np.ascontiguousarray(dx, dtype=np.float64))
        blk.point('g_old', g_old)
        blk.point('g_new', g_new)
        c.f_old, c.f_new, c.smag = float(f_old), float(f_new), float(smag)
        c.delta, c.rho = float(self.delta), float(self.rho)
        get_context().opt_step(blk)
        self.fused_steps += 1
        H = pes.H
        H._B_stale = bool(c.B_stale)
        if c.m > 0:
            H._view[2]._B_stale = bool(c.Bsub_stale)
        if c.updated:
            H._lr['r'] = blk.r
            v = H._view
            if c.m > 0:
                sub = v[2]
                sub._lr['r'] = blk.r_sub
                sub._B = None
                sub.version += 1
                sub._drop_dense_eig()
                H._view = (v[0], v[1], sub, H.version + 1)
            else:
                H._view = None
            H._B = None
            H.version += 1
            H._drop_dense_eig()
        self.delta, self.rho = c.delta, c.rho
        if rediag:
            if pes.hessian_function is not None:
                pes.calculate_hessian()
            else:
                pes.diag(**self.diagkwargs)
        else:
            self._proposed = (pes.curr.get('state_hash'), self.delta, (id(pes.H), pes.H.version), blk.s.copy(), c.smag_out)

    def step(self):
        blk = self._fused_block()
        if blk is not None:
            return self._step_fused(blk)
        self.__dict__.pop('_proposed', None)
        s, smag = self._predict_step()
        rediag = self._wants_diagonalisation()
        self.nsteps_since_diag = 0 if rediag else self.nsteps_since_diag + 1
        rho = self.pes.kick(s, rediag, **self.diagkwargs)
        if self._rebuild_if_internals_degraded():
            return
        self._adapt_radius(rho, smag)

    # ---- ASE Optimizer protocol -------------------------------------------------------------------------------------
    def _verdict(self):
        """(converged, fmax, cmax) at the threshold of the current run (0.05 before `run` set one)."""
        self._last_converged = self.pes.converged(0.05 if self.fmax is None else self.fmax)
        return self._last_converged

    def converged(self, forces=None):
        return self._verdict()[0]

    gradient_converged = converged            # newer ASE releases ask under this name

    def log(self, forces=None):
        """One line per step: Step Time Energy fmax cmax rtrust rho (optimize.py:457-502)."""
        out = self.logfile
        if out is None:
            return
        verdict = self._last_converged
        if verdict is None or len(verdict) != 3:
            verdict = self._verdict()
        label = type(self).__name__
        if self.nsteps == 0:
            heads = ("Step", "Time", "Energy", "fmax", "cmax", "rtrust", "rho")
            out.write(" " * len(label) + "{:>4s} {:>8s} {:>15s} {:>12s} {:>12s} {:>12s} {:>12s}\n".format(*heads))
        out.write("{} {:>3d} {:>8s} {:>15.6f} {:>12.4f} {:>12.4f} {:>12.4f} {:>12.4f}\n".format(
            label, self.nsteps, strftime("%H:%M:%S", localtime()), self.pes.get_f(), verdict[1], verdict[2],
            self.delta, self.rho))
        flush = getattr(out, 'flush', None)
        if callable(flush):
            flush()
