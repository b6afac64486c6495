"""`LibrarySearch` — a whole `Sella(atoms, ...).run(fmax, steps)` as library calls (`sella_search_*`, csrc/search.hip).

The general driver (`sella_amd.optimize.optimize.Sella`) keeps the reference's module structure and crosses into the
library ~100 times per optimizer step and ~300 times per diagonalisation; for a small system that is where the time
goes, and because the interpreter lock is held in between, host threads cannot share a GPU.  For the configuration an
ensemble of independent searches consists of (BASELINE configs[3]) the same loop runs inside the library:

    Cartesian PES, no constraints or constraints that pin single Cartesian coordinates (all satisfied);
    a calculator that lives in the library (`atoms.calc.device_calculator()`: the model PES, device EMT);
    3N >= `linalg.LR_MIN_DIM` (structured approximate Hessian), TS-BFGS, `rs` in {'tr', 'ras'}, built-in step families,
    no trajectory, no log, no observers.

`LibrarySearch.applies(atoms, **kwargs)` says whether a set of `Sella` keywords is covered; `run_one`
(sella_amd/ensemble.py) uses it for every member it can.  Results agree with the general driver's to the amplification
of last-bit differences (tests/test_library_search.py).  If a search leaves the covered configuration on the way (its
explicit rank passes 0.4 n) `run` raises `SearchLeftLibrary`; the atoms then hold the last geometry reached.
"""
import weakref
from ctypes import byref, c_double, c_int, c_long, c_void_p

import numpy as np

from . import _lib, linalg
from ._lib import SellaHipError, check, ptr
from .device import CONSTRAINT_KINDS, DAVIDSON_METHODS, STEPPER_KINDS, UPDATE_METHODS, get_context
from .optimize.optimize import _default_kwargs
from .optimize.restricted_step import RestrictedAtomicStep, TrustRegion, get_restricted_step
from .optimize.stepper import _all_steppers, get_stepper


class SearchLeftLibrary(RuntimeError):
    pass


_COVERED = {'order', 'eta', 'gamma', 'delta0', 'sigma_inc', 'sigma_dec', 'rho_inc', 'rho_dec', 'rs', 'method', 'eig',
            'threepoint', 'nsteps_per_diag', 'diag_every_n', 'constraints', 'proj_trans', 'proj_rot', 'logfile',
            'trajectory', 'internal'}


def _pinned_free(atoms, constraints, proj_trans, proj_rot):
    """Free coordinates of a constraint set made of single-coordinate pins — None: unconstrained; False: not covered
    (the defaults of `PES.__init__`, peswrapper.py:236-253, add a global translation / rotation constraint, whose
    projection basis is not a selection of coordinates)."""
    from .peswrapper import _pinned_coordinates
    has_trans = constraints is not None and bool(constraints.internals['translations'])
    if ((not has_trans) if proj_trans is None else proj_trans) or ((not np.any(atoms.pbc)) if proj_rot is None else proj_rot):
        return False
    if constraints is None:
        return None
    c = constraints
    if (c.nbonds + c.nangles + c.ndihedrals) > 0 or c.has_inequalities() or c.internals['rotations']:
        return False
    # (asked three times per search — `applies` twice, the constructor once — with a dense constraint Jacobian each time:
    # 4 ms of interpreter time per 256-atom ensemble member, serial under the interpreter lock; remembered per constraint
    # set and geometry)
    key = (getattr(c, '_ver', None), len(atoms), hash(np.asarray(atoms.positions, dtype=np.float64).tobytes()))
    memo = c.__dict__.get('_pinned_free_memo')
    if memo is not None and memo[0] == key and key[0] is not None:
        return memo[1]
    tr = c.internals['translations']
    act = c._active['translations']
    if tr and all(act) and all(len(t.indices) == 1 for t in tr):
        # every constraint pins one Cartesian coordinate of one atom (the README pattern): read them off the list
        # instead of building and scanning the dense Jacobian
        pins = np.fromiter((3 * int(t.indices[0]) + t.kwargs['dim'] for t in tr), dtype=np.int64, count=len(tr))
        x = np.asarray(atoms.positions, dtype=np.float64).ravel()
        if len(np.unique(pins)) != len(pins) or np.any(x[pins] - np.asarray(c._targets['translations'], dtype=np.float64)):
            out = False
        else:
            out = np.setdiff1d(np.arange(3 * len(atoms)), pins).astype(np.int32)
        c.__dict__['_pinned_free_memo'] = (key, out)
        return out
    drdx = c.jacobian()
    if drdx.shape[0] == 0:
        out = None
    else:
        pinned = _pinned_coordinates(drdx)
        if pinned is None or np.any(c.residual()):
            out = False
        else:
            out = np.setdiff1d(np.arange(3 * len(atoms)), pinned[0]).astype(np.int32)
    c.__dict__['_pinned_free_memo'] = (key, out)
    return out


class LibrarySearch:
    @staticmethod
    def applies(atoms, **kw):
        if set(kw) - _COVERED or kw.get('trajectory') is not None or kw.get('logfile') is not None or kw.get('internal'):
            return False
        calc = getattr(atoms, 'calc', None)
        if not getattr(calc, 'library_form', False):
            return False
        n = 3 * len(atoms)
        if linalg.LR_MIN_DIM is None or n < linalg.LR_MIN_DIM:
            return False
        order = kw.get('order', 1)
        table = _default_kwargs['minimum' if order == 0 else 'saddle']
        try:
            rs = get_restricted_step(kw['rs'] if kw.get('rs') is not None else 'ras')
            method = kw.get('method') or table['method']
            family = method if isinstance(method, type) else get_stepper(method.lower())
        except ValueError:
            return False
        if rs not in (TrustRegion, RestrictedAtomicStep) or family not in _all_steppers:
            return False
        return _pinned_free(atoms, kw.get('constraints'), kw.get('proj_trans'), kw.get('proj_rot')) is not False

    def __init__(self, atoms, order=1, eta=1e-4, gamma=0.1, delta0=None, sigma_inc=None, sigma_dec=None, rho_inc=None,
                 rho_dec=None, rs=None, method=None, eig=None, threepoint=False, nsteps_per_diag=3, diag_every_n=None,
                 constraints=None, proj_trans=None, proj_rot=None, logfile=None, trajectory=None, internal=False):
        if not self.applies(atoms, order=order, rs=rs, method=method, eig=eig, constraints=constraints,
                            proj_trans=proj_trans, proj_rot=proj_rot, logfile=logfile, trajectory=trajectory,
                            internal=internal):
            raise ValueError('this search is not covered by the library loop: use sella_amd.Sella')
        self.atoms = atoms
        table = _default_kwargs['minimum' if order == 0 else 'saddle']
        pick = lambda v, key: table[key] if v is None else v          # noqa: E731
        rs_name = rs if rs is not None else 'ras'
        rs_cls = get_restricted_step(rs_name)
        method = pick(method, 'method')
        family = method if isinstance(method, type) else get_stepper(method.lower())
        free = _pinned_free(atoms, constraints, proj_trans, proj_rot)
        n = 3 * len(atoms)
        nfree = n if free is None else len(free)
        p = _lib.SearchParams()
        p.order, p.eig, p.threepoint = int(order), int(bool(pick(eig, 'eig'))), int(bool(threepoint))
        p.dav_method = DAVIDSON_METHODS['jd0']
        p.stepper_kind, p.cons = STEPPER_KINDS[family._kind], CONSTRAINT_KINDS[rs_cls.measure]
        p.update_method, p.symm = UPDATE_METHODS['TS-BFGS'], 2
        p.nsteps_per_diag = int(nsteps_per_diag)
        p.diag_every_n = -1 if diag_every_n is None or not np.isfinite(diag_every_n) else int(diag_every_n)
        p.eta, p.gamma = float(eta), float(gamma)
        p.delta0 = float(pick(delta0, 'delta0')) * (nfree if rs_cls.measure == 'tr' else 1)          # optimize.py:183-186
        p.delta_min = float(eta)
        p.sigma_inc, p.sigma_dec = float(pick(sigma_inc, 'sigma_inc')), float(pick(sigma_dec, 'sigma_dec'))
        p.rho_inc, p.rho_dec = float(pick(rho_inc, 'rho_inc')), float(pick(rho_dec, 'rho_dec'))
        # the calculator has to know the species / cell before it can be handed to the library: the first force call is
        # made here, through the host-language object, and handed over
        f0 = float(atoms.get_potential_energy())
        g0 = np.ascontiguousarray(-np.asarray(atoms.get_forces(), dtype=np.float64)).ravel()
        self._calc = atoms.calc.device_calculator()
        if self._calc is None:
            raise ValueError('the calculator has no library form')
        self._free = free
        x0 = np.ascontiguousarray(atoms.positions, dtype=np.float64).ravel()
        h = c_void_p()
        ctx = get_context()
        check(_lib.lib().sella_search_create(ctx._h, self._calc._h, n, ptr(x0),
                                             None if free is None else free.ctypes.data_as(c_void_p),
                                             0 if free is None else len(free), byref(p), byref(h)))
        self._h, self._ctx, self._n = h, ctx, n
        check(_lib.lib().sella_search_seed(h, f0, ptr(g0)))
        # (finalizers run in reverse order of creation at interpreter exit: before the calculator's and the context's)
        self._fin = ctx.child(weakref.finalize(self, _lib.lib().sella_search_destroy, h))
        self.nsteps = 0
        self._sync()

    def _sync(self):
        x, g = np.empty(self._n), np.empty(self._n)
        sc, cn = np.zeros(5), (c_long * 6)()
        check(_lib.lib().sella_search_state(self._h, ptr(x), ptr(g), ptr(sc), cn))
        self.atoms.positions = x.reshape(-1, 3)
        self.gradient = g
        self.energy, self.fmax_now, self.delta, self.rho, self.lambda_min = (float(v) for v in sc)
        self.nsteps, self.neval, self.one_call_steps, self.rank, self.rank_view = (int(v) for v in cn[:5])
        self.initialized = bool(cn[5])

    def positions_flat(self):
        """The library's own copy of the geometry (3N)."""
        x = np.empty(self._n)
        sc, cn = np.zeros(5), (c_long * 6)()
        check(_lib.lib().sella_search_state(self._h, ptr(x), None, ptr(sc), cn))
        return x

    def run(self, fmax=0.05, steps=100000000):
        conv = c_int(0)
        try:
            check(_lib.lib().sella_search_run(self._h, float(fmax), int(steps), byref(conv)))
        except SellaHipError as e:
            self._sync()
            if getattr(e, 'status', None) == -7 or 'structured form' in str(e):
                raise SearchLeftLibrary(str(e)) from None
            raise
        self._sync()
        return bool(conv.value)

    def pending_pairs(self):
        """(S, Y) of a diagonalisation whose block update no longer fitted the structured form (the search then left the
        library with `SearchLeftLibrary`), or None: the caller applies them to the Hessian it takes over."""
        k = c_int(0)
        check(_lib.lib().sella_search_pending_pairs(self._h, byref(k), None, None))
        if k.value == 0:
            return None
        S, Y = np.empty((self._n, k.value)), np.empty((self._n, k.value))
        check(_lib.lib().sella_search_pending_pairs(self._h, byref(k), ptr(S), ptr(Y)))
        return S, Y

    def release_hessian(self):
        """Hand the approximate Hessian over (`sella_search_release_hessian`): -> dict(B, Wt, r, mu, lam0, stale, view=dict
        or None, nsteps_since_diag, first_diag) with `DeviceMatrix` objects that now own the device memory, or None if
        the search has no Hessian yet.  The search cannot be run afterwards."""
        from .device import DeviceMatrix
        n = self._n
        m = n if self._free is None else len(self._free)
        mats = (c_int * 4)()
        ints = (c_long * 8)()
        cap = max(8, min(n, max(max(1, int(0.4 * n)), 8) + 72))
        cap_sub = max(8, min(m, max(max(1, int(0.4 * m)), 8) + 72))
        mu, mu_sub = np.zeros(cap), np.zeros(cap_sub)
        lam0 = c_double(0.0)
        check(_lib.lib().sella_search_release_hessian(self._h, mats, ints, ptr(mu), ptr(mu_sub), byref(lam0)))
        out = dict(nsteps_since_diag=int(ints[6]), first_diag=bool(ints[7]), hessian=None)
        if ints[0] >= 0:
            ctx = self._ctx
            H = dict(B=DeviceMatrix(ctx, mats[0], (n, n)), Wt=DeviceMatrix(ctx, mats[1], (int(ints[2]), n)), r=int(ints[0]),
                     mu=mu[:int(ints[2])].copy(), lam0=float(lam0.value), stale=bool(ints[4]), view=None)
            if ints[1] >= 0:
                H['view'] = dict(B=DeviceMatrix(ctx, mats[2], (m, m)), Wt=DeviceMatrix(ctx, mats[3], (int(ints[3]), m)),
                                 r=int(ints[1]), mu=mu_sub[:int(ints[3])].copy(), stale=bool(ints[5]), idx=self._free)
            out['hessian'] = H
        return out

    def close(self):
        self._fin()
