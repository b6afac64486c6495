"""`PES` — the potential-energy-surface wrapper of the saddle-point loop, drop-in for the Cartesian
class of sella/peswrapper.py:214-607 (same constructor, getters, `diag()`, `kick()`,
`converged()`), built on the device-resident `ApproximateHessian`.

The calculator boundary is untouched: `eval()` calls `atoms.get_potential_energy()` /
`atoms.get_forces()` exactly like peswrapper.py:413-418.  Everything n x n happens on the
MI355X: B·s products, U^T B U projections, the eigendecomposition shared by the Davidson
preconditioner / TS-BFGS / P-RFO, the Davidson loop and the quasi-Newton updates.
Host-side (as in the reference, outside its accelerator seam): the rank-revealing pivoted QR
of the (ncons x n) constraint Jacobian and O(n) vector algebra.

Not restated here (out of the saddle-point scope): InternalPES / Cell*PES.
"""
import numpy as np
from scipy.linalg import eigh, qr

from .atoms import Atoms  # noqa: F401  (re-export for users without ASE)
from .device import get_context
from .eigensolvers import rayleigh_ritz
from .hessian_update import symmetrize_Y
from .internal import Constraints, DuplicateInternalError
from .linalg import ApproximateHessian, NumericalHessian
from .utilities.math import register_selection, selection_of, is_identity, shared_identity


class _LRU2:
    """Two-entry cache keyed by the geometry hash (peswrapper.py:24-48)."""

    def __init__(self):
        self._entries = [None, None]
        self._next = 0

    def get(self, key):
        for e in self._entries:
            if e is not None and e[0] == key:
                return e[1]
        return None

    def put(self, key, value):
        if self.get(key) is not None:
            return
        self._entries[self._next] = (key, value)
        self._next = 1 - self._next


def _pinned_coordinates(drdx):
    """(column index, value) per row when every constraint pins exactly one distinct Cartesian coordinate
    (fix_translation on single atoms), else None.  For such constraints the least-squares solves of the glue
    code (multipliers, linear constraint correction, peswrapper.py:429-438, 475-479) are componentwise divisions."""
    if drdx.shape[0] == 0:
        return None
    memo = _pinned_memo[0]                           # ONE read of ONE slot: (drdx, out) is published atomically, so
    if memo is not None and memo[0] is drdx:         # threads driving different replicas never mix key and value
        return memo[1]
    nz = drdx != 0.0
    if not np.all(nz.sum(axis=1) == 1):
        out = None
    else:
        cidx = nz.argmax(axis=1)
        out = None if len(np.unique(cidx)) != len(cidx) else (cidx, drdx[np.arange(len(cidx)), cidx])
    if not drdx.flags.writeable:                     # read-only Jacobian of a translation-only constraint set
        _pinned_memo[0] = (drdx, out)
    return out


_pinned_memo = [None]


def _split_cons_subspace(drdx, tol_factor=1e-6):
    """(Ucons, Ufree): row space of the constraint Jacobian and its orthogonal complement by
    rank-revealing pivoted QR of drdx^T (peswrapper.py:51-69)."""
    n = drdx.shape[1]
    if drdx.shape[0] == 0:
        return np.zeros((n, 0)), shared_identity(n)
    # Constraints that each pin one Cartesian coordinate (fix_translation on single atoms, the README slab) span a
    # coordinate subspace: the bases are columns of the identity and no O(n^3) full QR is needed.  Any
    # orthonormal basis of the same subspaces gives the same projected problem.
    nz = drdx != 0.0
    if np.all(nz.sum(axis=1) == 1):
        cols = np.flatnonzero(nz.any(axis=0))
        if len(cols) == drdx.shape[0]:
            free = np.setdiff1d(np.arange(n), cols)
            eye = shared_identity(n)
            # (fancy indexing along axis 1 returns a Fortran-ordered array: make the C copy once, not at every upload)
            return (register_selection(np.ascontiguousarray(eye[:, cols]), cols),
                    register_selection(np.ascontiguousarray(eye[:, free]), free))
    Q, R, _ = qr(drdx.T, mode='full', pivoting=True, check_finite=False)
    diag = np.abs(np.diag(R))
    ncons = int(np.sum(diag > tol_factor * diag[0])) if diag.size and diag[0] > 0 else 0
    return Q[:, :ncons], Q[:, ncons:]


def open_trajectory(name, atoms, append=False):
    """A file NAME passed as `trajectory=` (peswrapper.py:257-261, optimize.py:144-150 of the reference): an ASE
    `.traj` file (sella_amd/trajectory.py) like the reference writes — or, for names ending in .xyz / .extxyz, the
    extended-XYZ text writer."""
    if name.lower().endswith(('.xyz', '.extxyz')):
        from .atoms import XYZTrajectory
        return XYZTrajectory(name, atoms, mode='a' if append else 'w')
    from .trajectory import Trajectory
    return Trajectory(name, 'a' if append else 'w', atoms)


class PES:
    n_cell_dof = 0

    def __init__(self, atoms, H0=None, constraints=None, eigensolver='jd0', trajectory=None,
                 eta=1e-4, v0=None, proj_trans=None, proj_rot=None, hessian_function=None):
        self.atoms = atoms
        self.cons = self._constraint_set(atoms, constraints, proj_trans, proj_rot)
        self.eigensolver, self.eta, self.v0 = eigensolver, eta, v0
        self.hessian_function = hessian_function
        self.traj = open_trajectory(trajectory, atoms) if isinstance(trajectory, str) else trajectory
        # coordinate system: Cartesian here (subclasses set `int` / `dummies` and the sizes themselves)
        self.int = self.dummies = None
        self.dim = self.ncart = 3 * len(atoms)
        # the cached point and its predecessor (plain dicts: the IRC driver swaps them in and out), counters
        self.curr = dict.fromkeys(('x', 'f', 'g'))
        self.last = dict(self.curr)
        self.savepoint = dict.fromkeys(('apos', 'dpos'))
        self.neval = 0
        self.first_diag = True
        self._basis_cache = _LRU2()
        self.set_H(H0, initialized=H0 is not None)

    @staticmethod
    def _constraint_set(atoms, constraints, proj_trans, proj_rot):
        """The constraints a search runs under (peswrapper.py:226-253): the user's, plus — unless they already say
        something about it or the caller decides otherwise — a fixed centre of the system and, for non-periodic
        systems, fixed global rotations.  The rotation constraint is the LINEARISED one (infinitesimal generators about
        the centroid, internal.py Constraints.fix_rotation): the three rotational soft modes leave the Davidson / P-RFO
        subspace exactly as in the reference."""
        cons = Constraints(atoms) if constraints is None else constraints
        want_trans = (not cons.internals['translations']) if proj_trans is None else proj_trans
        want_rot = (not np.any(atoms.pbc)) if proj_rot is None else proj_rot
        if want_trans:
            try:
                cons.fix_translation(replace_ok=False)
            except DuplicateInternalError:
                pass                                             # some component is pinned already
        if want_rot and not cons.internals['rotations']:
            cons.fix_rotation()
        return cons

    apos = property(lambda self: self.atoms.positions.copy())
    dpos = property(lambda self: None)

    def _state_hash(self):
        """Key of everything cached per geometry: the bytes of the positions and, if there is one, of the cell."""
        parts = [np.ascontiguousarray(self.atoms.positions).tobytes()]
        cell = np.asarray(self.atoms.cell, dtype=float)
        if cell.any():
            parts.append(cell.tobytes())
        return b''.join(parts)

    def save(self):
        self.savepoint = {'apos': self.apos, 'dpos': self.dpos}

    def restore(self):
        kept = self.savepoint['apos']
        if kept is None:
            raise AssertionError('PES.restore() without a saved geometry')
        self.atoms.positions = kept

    def close(self):
        if self.traj is not None:
            self.traj.close()
            self.traj = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    # ---- positions -------------------------------------------------------------------------
    def set_x(self, target):
        diff = target - self.get_x()
        self.atoms.positions = target.reshape((-1, 3))
        g = self.curr.get('g')
        return diff, diff, (np.zeros_like(diff) if g is None else g)

    def get_x(self):
        return self.apos.ravel().copy()

    # ---- Hessians ----------------------------------------------------------------------------
    def get_H(self):
        return self.H

    def set_H(self, target, *args, **kwargs):
        self.H = ApproximateHessian(self.dim, self.ncart, target, *args, **kwargs)

    def get_Hc(self):
        """Constraint curvature contracted with the multipliers, sum_k L_k d2 r_k / dx2."""
        multipliers = self.curr.get('L')
        if multipliers is None:
            raise RuntimeError("PES.get_Hc() called with L=None.")
        return self.cons.hessian().ldot(multipliers)

    def get_HL(self):
        """Hessian of the Lagrangian as a lazy sum (linalg.MatrixSum semantics)."""
        return self.H + (-self.get_Hc())

    def _has_curved_constraints(self):
        c = self.cons
        return (c.nbonds + c.nangles + c.ndihedrals) > 0

    def get_HL_projected(self, U):
        """ApproximateHessian(U^T (B - Hc) U) without forming HL (peswrapper.py:363-386).

        The result is cached per (Hessian version, basis): the optimizer asks for the same
        projection from the step solve and from the curvature test of one iteration.  Without
        curved constraints and with an identity basis the projection *is* B, and the object
        returned is the PES Hessian itself, so its device eigendecomposition is computed once and
        shared with the quasi-Newton update."""
        H = self.get_H()
        n = U.shape[1]
        if H._is_none:
            return ApproximateHessian(n, 0, None, H.update_method, H.symm)
        L = self.curr.get('L')
        curved = L is not None and L.size > 0 and self._has_curved_constraints()
        if is_identity(U) and not curved:
            return H
        selection = (not curved) and getattr(self, '_pinned_basis', None) is not None and U is self._pinned_basis[2]
        if selection:
            view = H.principal_view(U)
            if view is not None:
                return view
        key = (id(H), H.version, id(U), None if not curved else L.tobytes())
        hit = getattr(self, '_hlproj_cache', None)
        if hit is not None and hit[0] == key and hit[1] is U:
            return hit[2]
        ctx = get_context()
        if is_identity(U):
            Bproj = H._get_B_gpu().copy()
        else:
            Bproj = ctx.project_dev(H._get_B_gpu(), ctx.resident(U))
        if curved:
            Hc = self.get_Hc()          # zero for translation-only constraints: skipped above
            dHc = ctx.upload(U.T @ Hc @ U)
            Bnew = ctx.axpby(1.0, Bproj, -1.0, dHc)
            Bproj.free()
            dHc.free()
            Bproj = Bnew
        out = ApproximateHessian(n, 0, Bproj, H.update_method, H.symm)
        self._hlproj_cache = (key, U, out)
        if selection:
            H.register_view(U, out)          # from now on updated together with H (linalg.ApproximateHessian.update)
        return out

    # ---- constraints ------------------------------------------------------------------------
    def get_res(self):
        return self.cons.residual()

    def get_drdx(self):
        return self.cons.jacobian()

    def _calc_basis(self):
        key = self._state_hash()
        cached = self._basis_cache.get(key)
        if cached is not None:
            return cached
        drdx = self.get_drdx()
        pinned = _pinned_coordinates(drdx)
        self._pinned_cache = (key, pinned)
        if pinned is not None:
            # coordinate-pinning constraints: the bases depend on the constraint set only, so the SAME arrays are
            # handed out at every geometry (downstream caches and device uploads key on the object)
            sig = pinned[0].tobytes()
            hit = getattr(self, '_pinned_basis', None)
            if hit is None or hit[0] != sig:
                Ucons, Ufree = _split_cons_subspace(drdx)
                hit = (sig, Ucons, Ufree)
                self._pinned_basis = hit
            Ucons, Ufree = hit[1], hit[2]
        else:
            Ucons, Ufree = _split_cons_subspace(drdx)
        result = (drdx, Ucons, shared_identity(self.dim), Ufree)
        self._basis_cache.put(key, result)
        return result

    def write_traj(self):
        if self.traj is not None:
            self.traj.write()

    # ---- the calculator boundary (peswrapper.py:413-418) ----------------------------------------
    def eval(self):
        self.neval += 1
        f = self.atoms.get_potential_energy()
        g = -np.asarray(self.atoms.get_forces()).ravel()
        self.write_traj()
        return f, g

    def _calc_eg(self, x):
        self.save()
        self.set_x(x)
        f, g = self.eval()
        self.restore()
        return f, g

    def get_scons(self):
        """Minimum-norm linear correction towards the constraint manifold (peswrapper.py:429-438)."""
        Ucons = self.get_Ucons()
        if Ucons.shape[1] == 0:
            return np.zeros(self.dim)
        pinned = self._pinned()
        if pinned is not None and self.int is None:
            s = np.zeros(self.dim)
            s[pinned[0]] = -self.get_res() / pinned[1]
            return s
        return -Ucons @ np.linalg.lstsq(self.get_drdx() @ Ucons, self.get_res(), rcond=None)[0]

    def _pinned(self):
        """Cached `_pinned_coordinates` of the current constraint Jacobian (it only depends on the constraint
        set for translation constraints, but is keyed on the geometry like the basis)."""
        key = self._state_hash()
        hit = getattr(self, '_pinned_cache', None)
        if hit is None or hit[0] != key:
            hit = (key, _pinned_coordinates(PES.get_drdx(self)) if self.int is None else None)
            self._pinned_cache = hit
        return hit[1]

    # ---- the cached point: geometry -> (basis, energy, gradient, multipliers) ---------------------------------------
    # `curr` / `last` are plain dicts because callers swap them in and out (IRC, sella/optimize/irc.py:62-75); what
    # fills them is organised around ONE question — does the cache describe the geometry the atoms have now?
    def _cache_status(self, want_energy):
        """'fresh': nothing to do; 'energy': same geometry, energy / gradient still missing; 'moved': new geometry."""
        here = self._state_hash()
        if self.curr['x'] is None or here != self.curr.get('state_hash'):
            return 'moved', here
        if want_energy and self.curr['f'] is None:
            return 'energy', here
        return 'fresh', here

    def _update(self, feval=True):
        status, here = self._cache_status(feval)
        if status == 'fresh':
            return False
        basis = self._calc_basis()
        energy, gradient = self.eval() if feval else (None, None)
        if status == 'moved':
            self.last = self.curr.copy()          # the previous point becomes the reference of the next secant pair
        self.curr.update(x=self.get_x(), state_hash=here, f=energy, g=gradient)
        self._update_basis(basis)
        return True

    def _multipliers(self, drdx, g):
        """Lagrange multipliers of the constraint forces: least-squares solution of drdx^T L = g (peswrapper.py:475-479);
        a division per constraint when every constraint pins one coordinate."""
        if g is None:
            return None
        if drdx.shape[0] == 0:
            return np.zeros(0)
        pinned = self._pinned()
        if pinned is not None:
            return g[pinned[0]] / pinned[1]
        return np.linalg.lstsq(drdx.T, g, rcond=None)[0]

    def _update_basis(self, basis=None):
        drdx, Ucons, Unred, Ufree = self._calc_basis() if basis is None else basis
        self.curr.update(drdx=drdx, Ucons=Ucons, Unred=Unred, Ufree=Ufree, L=self._multipliers(drdx, self.curr['g']))

    def _update_H(self, dx, dg):
        if self.last['x'] is None or self.last['g'] is None:
            return
        self.H.update(dx, dg)

    def _point_getter(name, needs_energy, copy=False):      # noqa: N805 — class-body helper, not a method
        """Accessor of one entry of the cached point: bring the cache up to date (with or without a force call), hand
        the entry out (a copy where callers modify what they get)."""
        def getter(self):
            self._update(needs_energy)
            value = self.curr[name]
            return value.copy() if copy else value
        getter.__name__ = 'get_' + name
        return getter

    get_f = _point_getter('f', True)
    get_g = _point_getter('g', True, copy=True)
    get_Unred = _point_getter('Unred', False)
    get_Ufree = _point_getter('Ufree', False)
    get_Ucons = _point_getter('Ucons', False)
    del _point_getter

    # ---- iterative diagonalisation (peswrapper.py:508-556) ----------------------------------------
    def diag(self, gamma=0.1, threepoint=False, maxiter=None):
        self._update(True)                                  # energy and gradient of the point the operator is built at
        Ufree = self.get_Ufree()
        if Ufree.shape[1] == 0:
            return
        P = self.get_HL_projected(Ufree)
        P_is_none = P._is_none
        # start vector (peswrapper.py:521-529): only while there is no curvature information to start from — the user's
        # v0, else the projected gradient, unless that vanishes
        v0 = None
        if P_is_none or self.first_diag:
            v0 = self.v0
            if v0 is None:
                g = self.get_g()
                v0 = g if is_identity(Ufree) else g @ Ufree
            if np.linalg.norm(v0) < 1e-12:
                v0 = None
        Hproj = self._library_fd_operator(Ufree, threepoint)
        if Hproj is None:
            Hproj = NumericalHessian(self._calc_eg, self.get_x(), self.get_g(), self.eta, threepoint, Ufree)
        A = Hproj
        Hc = None
        if self._has_curved_constraints():
            Hc = self.get_Hc()
            A = Hproj + (-(Ufree.T @ Hc @ Ufree))
        # P is handed over as the ApproximateHessian itself: its device eigendecomposition is the
        # preconditioner of the JD correction (sella_amd/csrc/davidson.hip)
        rayleigh_ritz(A, gamma, None if P_is_none else P, v0=v0, method=self.eigensolver, maxiter=maxiter)

        Vs, AVs = Hproj.Vs, Hproj.AVs
        if not isinstance(Hproj, NumericalHessian):
            self.neval += Hproj.calls * (2 if threepoint else 1)     # force calls the library made itself
        # Ritz vectors of the collected full-space iterates (peswrapper.py:545-551)
        Atilde = Vs.T @ symmetrize_Y(Vs, AVs, symm=2)
        if Hc is not None:
            Atilde = Atilde - Vs.T @ Hc @ Vs
        _, X = eigh(Atilde)
        self.H.update(Vs @ X, AVs @ X)
        self.first_diag = False

    def _library_fd_operator(self, Ufree, threepoint):
        """The finite-difference Hessian as a library object (`sella_fd_*`) when the calculator itself lives in the
        library (`atoms.calc.device_calculator()`): the force calls of the Davidson run are then library calls, with no
        interpreter frame between them.  Needs a basis that is the identity or a selection of coordinates, no curved
        constraints, and nobody listening per force call (a trajectory writes one image per call, peswrapper.py:409-418)."""
        from .utilities.math import selection_of
        calc = getattr(self.atoms, 'calc', None)
        maker = getattr(calc, 'device_calculator', None)
        if maker is None or self.traj is not None or self._has_curved_constraints() or type(self) is not PES:
            return None
        if getattr(self, 'use_library_calculator', True) is False:
            return None
        free = None
        if not is_identity(Ufree):
            free = selection_of(Ufree)
            if free is None:
                return None
        dc = maker()
        if dc is None:
            return None
        from .device import DeviceFdOperator
        return DeviceFdOperator(dc, self.get_x(), self.get_g(), self.eta, threepoint, free)

    def get_projected_forces(self):
        g = self.get_g()
        Ufree = self.get_Ufree()
        if is_identity(Ufree):
            return -g.reshape((-1, 3))
        hit = getattr(self, '_pinned_basis', None)
        if hit is not None and Ufree is hit[2]:
            # columns of the identity: the projection keeps the free components and zeroes the pinned ones
            if len(hit) < 4:
                sel = selection_of(Ufree)
                hit = self._pinned_basis = hit + (Ufree.argmax(axis=0) if sel is None else sel,)
            out = np.zeros_like(g)
            out[hit[3]] = g[hit[3]]
            return -out.reshape((-1, 3))
        return -(Ufree @ (Ufree.T @ g)).reshape((-1, 3))

    def converged(self, fmax, cmax=1e-5):
        """(done, largest projected force on an atom, norm of the constraint residual)."""
        per_atom = self.get_projected_forces()
        worst_force = np.sqrt(np.einsum('ij,ij->i', per_atom, per_atom).max())
        violation = np.linalg.norm(self.get_res())
        return bool(worst_force < fmax and violation < cmax), worst_force, violation

    def wrap_dx(self, dx):
        return dx

    def get_df_pred(self, dx, g, H):
        """Energy change of the quadratic model along dx (one Hessian-vector product on the device)."""
        return None if H is None else float(g @ dx + 0.5 * (dx @ (H @ dx)))

    # ---- step + update (peswrapper.py:578-602) -------------------------------------------------------
    def kick(self, dx, diag=False, **diag_kwargs):
        """Take the step dx: returns the ratio of the actual to the predicted energy change (None when no prediction
        is possible), after teaching the approximate Hessian the new secant pair and — if asked — refreshing its
        lowest modes by iterative diagonalisation."""
        before = dict(x=self.get_x(), f=self.get_f(), g=self.get_g())
        dx_asked, dx_taken, g_transported = self.set_x(before['x'] + dx)
        # quadratic model of the energy change along the step that was asked for (B dx on the device)
        predicted = self.get_df_pred(dx_asked, before['g'], self.H)
        secant_dg = self.get_g() - g_transported
        actual = self.get_f() - before['f']
        ratio = actual / predicted if (predicted is not None and abs(predicted) >= 1e-14) else None
        self._update_H(dx_taken, secant_dg)
        if diag:
            if self.hessian_function is not None:
                self.calculate_hessian()
            else:
                self.diag(**diag_kwargs)
        return ratio

    def calculate_hessian(self):
        assert self.hessian_function is not None
        self.H.set_B(self.hessian_function(self.atoms))


# ------------------------------------------------------------------------------------------------
# InternalPES — optimisation in redundant internal coordinates with geodesic steps
# (sella/peswrapper.py:609-1288; Hermes et al., J. Chem. Phys. 155, 094105 (2021)).
#
# Scope of this build: bonds / angles / dihedrals given as an `InternalCoordinates` object (or found
# automatically from covalent radii), Cartesian constraints expressed through the usual `Constraints`
# object.  Not built: dummy atoms and the re-generation of internals when an angle becomes linear
# (`update_internals`, :1126-1172 — a RuntimeError is raised instead), the Newton "iterative stepper"
# shortcut (:742-838; the ODE path it falls back to is the one implemented), cell degrees of freedom.
# The heavy pieces run on the device: B-matrix rows and D(v) = H_i v (csrc/internals.hip), the range basis
# and pseudo-inverse of B through the device eigensolver (`_BFactor`, in place of the reference's
# `_gpu_qr` + SVD, :691-709), the Hessian algebra as for `PES`.
# Parity: the reference class needs ASE + JAX and cannot be imported in the build container, so this
# restatement is property-tested only (tests/test_internal_pes.py) — unpinned.
# ------------------------------------------------------------------------------------------------
from scipy.integrate import LSODA  # noqa: E402

from .internal import InternalCoordinates  # noqa: E402


class _BFactor:
    """B = dq/dx (nint x 3N) held sparse together with the spectral factor of its Gram matrix.

    The reference takes an economy QR of the dense B and, because B never has full column rank without
    TRIC coordinates (the rigid-body motions are in its null space), falls through to a dense SVD
    (peswrapper.py:674-709) to get the range basis Q and the pseudo-inverse.  Both follow from the
    eigendecomposition of the small Gram matrix — G = B^T B = V L V^T (3N x 3N) when nint >= 3N, else
    G = B B^T — which is assembled from the sparse rows and diagonalised by the device eigensolver:

        Q = B V_r L_r^-1/2   R = L_r^1/2 V_r^T   B^+ = M B^T,  M = V_r L_r^-1 V_r^T   (device resident)

    (singular values s_i = sqrt(l_i) > 1e-6 kept, like `Si > 1e-6` in the reference).  B^+ y is then one
    sparse product and one symmetric matrix-panel product on the device; nothing of size nint x 3N is
    formed unless a caller asks for the dense pseudo-inverse.  The pseudo-inverse is unique, Q is one of
    the (equally arbitrary) orthonormal bases of range(B)."""

    def __init__(self, Bs, tol=1e-6):
        ctx = get_context()
        self.Bs = Bs.tocsr()
        self.BsT = Bs.T.tocsr()
        nint, nx = Bs.shape
        self.shape = (nint, nx)
        self.left = nint < nx                       # Gram matrix on the short side
        G = ((self.Bs @ self.BsT) if self.left else (self.BsT @ self.Bs)).toarray()
        n = G.shape[0]
        if n == 0:
            self.rank, self.s, self.M, self._N0 = 0, np.zeros(0), None, None
            self._Q, self._R, self.BinvQ, self._dense = np.zeros((nint, 0)), np.zeros((0, nx)), np.zeros((nx, 0)), None
            return
        hG = ctx.upload(G)
        w, V, Vt = ctx.eigh(hG)
        hG.free()
        V.free()
        floor = max(tol * tol, 64 * n * np.finfo(float).eps * max(w[-1], 0.0))
        keep = np.flatnonzero(w > floor)[::-1]                    # descending singular values
        Vall = Vt.numpy()
        Vr = Vall[keep]                                           # (r, n): eigenvectors as rows
        # null space of the Gram matrix = null(B) on the Cartesian side (rigid-body motions): its basis lets
        # `pinv_dot_moved` return the MINIMUM-NORM solution at a nearby geometry
        self._N0 = None if self.left else np.ascontiguousarray(Vall[np.setdiff1d(np.arange(n), keep)].T)
        Vt.free()
        self.s = np.sqrt(w[keep])
        self.rank = len(keep)
        X = np.ascontiguousarray(Vr / self.s[:, None])            # rows v_i / s_i
        hX = ctx.upload(X)
        self.M = ctx.zeros(n, n)
        ctx.gemm(hX, hX, self.M, transA=True)                     # M = V_r L_r^-1 V_r^T
        hX.free()
        self._Vr = Vr
        self.BinvQ = None if self.left else np.ascontiguousarray(X.T)        # B^+ Q = V_r L_r^-1/2
        self._Q = self._R = self._dense = None
        if self.left:
            self._Q = np.ascontiguousarray(Vr.T)                  # left singular vectors
            self._R = np.ascontiguousarray((self.BsT @ self._Q).T)          # Q^T B
            self.BinvQ = np.ascontiguousarray(self._R.T / (self.s ** 2)[None, :])

    @property
    def Q(self):
        """Orthonormal basis of range(B), (nint, rank) — dense, formed on first use."""
        if self._Q is None:
            self._Q = np.ascontiguousarray(self.Bs @ self.BinvQ)
        return self._Q

    @property
    def R(self):
        if self._R is None:
            self._R = self.s[:, None] * self._Vr
        return self._R

    def pinv_dot(self, Y):
        """B^+ Y for Y (nint,) or (nint, k)."""
        if self.M is None or np.size(Y) == 0:
            return np.zeros((self.shape[1],) + np.shape(Y)[1:])
        ctx = get_context()
        if self.left:
            return self.BsT @ ctx.symm_mm(self.M, Y)
        return ctx.symm_mm(self.M, self.BsT @ Y)

    _NULL_MAX = 16          # more null directions than this: not a rigid-body null space, factorise anew

    def pinv_dot_moved(self, Bs_new, Y, tol=1e-11, maxit=30):
        """B_new^+ Y for the Jacobian of a NEARBY geometry, without a new factorisation: preconditioned conjugate
        gradients on the normal equations (B_new^T B_new) x = B_new^T Y with this factor's M = (B^T B)^+ as the
        preconditioner — M B_new^T B_new is the projector onto range(B^T) plus a perturbation of the size of the
        geometry change, so a handful of iterations (two sparse products and one device panel product each) reach
        1e-11.  The CG iterates stay in range(B^T) of the OLD geometry, so what converges is a least-squares solution
        that differs from the minimum-norm one by a null vector of B_new (the null space — rigid rotations of a
        molecule — has turned with the geometry).  The same solve therefore carries d extra columns B_new N0 (N0 = this
        factor's null basis, d = 3 ... 6): N0 - C spans null(B_new), and the solution is projected onto its
        orthogonal complement, which makes it B_new^+ Y exactly (the reference applies the pseudo-inverse of the
        current point, peswrapper.py:1200-1221).  Returns None if CG does not converge or the null space is not
        a small rigid-body one (the caller then factorises anew)."""
        if self.M is None or self.left or np.size(Y) == 0:
            return None
        ctx = get_context()
        Y2 = Y[:, None] if np.ndim(Y) == 1 else Y
        ny = Y2.shape[1]
        N0 = self._N0
        d = 0 if N0 is None else N0.shape[1]
        if d > self._NULL_MAX:
            return None
        if d:
            Y2 = np.column_stack((Y2, np.asarray(Bs_new @ N0)))
        BT_new = Bs_new.T.tocsr()
        R = np.asarray(BT_new @ Y2)
        X = np.zeros_like(R)
        Z = ctx.symm_mm(self.M, R)
        P = Z.copy()
        rz = np.einsum('ij,ij->j', R, Z)
        rz0 = np.where(rz > 0, rz, 1.0)
        # (null directions that B_new barely sees start with a residual at rounding level: relative to the scale
        # of the data columns they are converged from the start)
        floor = tol * tol * max(float(rz0[:ny].max()), 1e-300) * 1e-6
        for _ in range(maxit):
            AP = np.asarray(BT_new @ (Bs_new @ P))
            pap = np.einsum('ij,ij->j', P, AP)
            live = (rz > np.maximum(tol * tol * rz0, floor)) & (pap > 0)
            if not live.any():
                break
            alpha = np.where(live, rz / np.where(pap > 0, pap, 1.0), 0.0)
            X += alpha * P
            R -= alpha * AP
            Z = ctx.symm_mm(self.M, R)
            rz_new = np.einsum('ij,ij->j', R, Z)
            P = Z + np.where(live, rz_new / np.where(rz != 0, rz, 1.0), 0.0) * P
            rz = rz_new
        else:
            return None
        sol = X[:, :ny]
        if d:
            Nn, _ = np.linalg.qr(N0 - X[:, ny:])                 # orthonormal basis of null(B_new)
            sol = sol - Nn @ (Nn.T @ sol)
        return sol[:, 0] if np.ndim(Y) == 1 else sol

    def pinvT_dot(self, Y):
        """(B^+)^T Y for Y (3N,) or (3N, k) — e.g. the Cartesian gradient -> internal gradient."""
        if self.M is None or np.size(Y) == 0:
            return np.zeros((self.shape[0],) + np.shape(Y)[1:])
        ctx = get_context()
        if self.left:
            return ctx.symm_mm(self.M, self.Bs @ Y)
        return self.Bs @ ctx.symm_mm(self.M, Y)

    def pinv(self):
        """Dense B^+ (3N x nint)."""
        if self._dense is None:
            if self.M is None:
                self._dense = np.zeros(self.shape[::-1])
            else:
                M = self.M.numpy()
                self._dense = np.asarray(self.BsT @ M) if self.left else np.ascontiguousarray(np.asarray(self.Bs @ M).T)
        return self._dense

    def project_diag(self, h):
        """P diag(h) P with P = Q Q^T the projector onto range(B) (`_range_space_projector`,
        peswrapper.py:72-82, applied as in :644-650): three device GEMMs."""
        nint = self.shape[0]
        if self.rank == 0:
            return np.zeros((nint, nint))
        ctx = get_context()
        hQ = ctx.upload(self.Q)
        hQh = ctx.upload(self.Q * np.asarray(h)[:, None])
        hT = ctx.zeros(self.rank, self.rank)
        ctx.gemm(hQ, hQh, hT, transA=True)
        hQh.free()
        hZ = ctx.zeros(nint, self.rank)
        ctx.gemm(hQ, hT, hZ)
        hT.free()
        hH = ctx.zeros(nint, nint)
        ctx.gemm(hZ, hQ, hH, transB=True)
        H = hH.numpy()
        for hnd in (hQ, hZ, hH):
            hnd.free()
        return 0.5 * (H + H.T)


class InternalPES(PES):
    def __init__(self, atoms, internals, *args, H0=None, iterative_stepper=0, auto_find_internals=True,
                 exact_geodesic=False, **kwargs):
        if internals is None or internals is True:
            internals = InternalCoordinates.from_atoms(atoms, cons=kwargs.pop('constraints', None))
        self.int_orig = internals
        new_int = internals.copy()
        if new_int.cons is None:
            new_int.cons = Constraints(atoms)
        kwargs.pop('constraints', None)
        # global translations / rotations are never projected out in internal space (peswrapper.py:633-641): a
        # caller's `Sella(internal=True, proj_rot=...)` must not collide with the explicit keywords below
        kwargs.pop('proj_trans', None)
        kwargs.pop('proj_rot', None)
        self._factor_cache = _LRU2()
        self._Hc_cache = _LRU2()
        PES.__init__(self, atoms, *args, constraints=new_int.cons, H0=None, proj_trans=False, proj_rot=False,
                     **kwargs)
        self.int = new_int
        self.dim = len(self.get_x())
        self.ncart = self.int.ndof
        if H0 is None:
            # guess Hessian with the components in the infeasible (redundant) subspace zeroed, :644-650
            self.set_H(self._get_factor().project_diag(self.int.guess_hessian(diagonal_only=True)), initialized=False)
        else:
            self.set_H(H0, initialized=True)
        self.bad_int = None
        self.exact_geodesic = exact_geodesic
        self.iterative_stepper = iterative_stepper

    # ---- B = dq/dx: range basis and pseudo-inverse from one factorisation (:674-736) -------------------
    def _get_factor(self):
        key = self._state_hash()
        cached = self._factor_cache.get(key)
        if cached is None:
            cached = _BFactor(self.int.jacobian_csr())
            self._factor_cache.put(key, cached)
        return cached

    def _get_jacobian_qr(self):
        fac = self._get_factor()
        return fac.Q, fac.R

    def _get_Binv(self):
        return self._get_factor().pinv()

    # ---- geodesic position update (:840-880, :1200-1221) ----------------------------------------------
    def _q_ode(self, t, y):
        nx = 3 * len(self.atoms)
        x, dxdt, g = y.reshape((3, nx))
        dydt = np.zeros((3, nx))
        dydt[0] = dxdt
        self.atoms.positions = x.reshape((-1, 3)).copy()
        # rows of D(dxdt) are H_i dxdt (one device launch per kind); only D @ [dxdt, g] is needed
        rhs = self.int.hessian_rdot_mult(dxdt, np.column_stack((dxdt, g)))
        fac = self._ode_factor
        if self.exact_geodesic:
            # pseudo-inverse AT THE CURRENT POINT of the path (peswrapper.py:1200-1221 re-evaluates it at every
            # right-hand side): carried from the factor of the starting point by a few preconditioned CG
            # iterations on the sparse Jacobian instead of a new spectral factorisation per right-hand side
            # (130 ms each at 1024 atoms); a new factor is taken — and becomes the carrier — only if that stalls
            sol = fac.pinv_dot_moved(self.int.jacobian_csr(), rhs)
            if sol is None:
                fac = self._ode_factor = self._get_factor()
                sol = fac.pinv_dot(rhs)
            out = -sol
        else:
            out = -fac.pinv_dot(rhs)                                                          # (nx, 2)
        dydt[1] = out[:, 0]
        dydt[2] = out[:, 1]
        return dydt.ravel()

    def _set_x_ode(self, target):
        dx = self.wrap_dx(target - self.get_x())
        fac = self._get_factor()
        self._ode_factor = fac
        g_int = self.curr.get('g')
        if g_int is None:
            g_int = np.zeros_like(dx)
        y0 = np.hstack((self.apos.ravel(), fac.pinv_dot(np.column_stack((dx, g_int))).T.ravel()))
        ode = LSODA(self._q_ode, 0.0, y0, t_bound=1.0, atol=1e-6)
        t0, y = 0.0, y0
        while ode.status == 'running':
            ode.step()
            y, t0 = ode.y, ode.t
            self.bad_int = self.int.check_for_bad_internals()
            if self.bad_int is not None:
                break
            if ode.nfev > 1000:
                raise RuntimeError("Geometry update ODE is taking too long to converge!")
        if ode.status == 'failed':
            raise RuntimeError("Geometry update ODE failed to converge!")
        nx = 3 * len(self.atoms)
        y = y.reshape((3, nx))
        self.atoms.positions = y[0].reshape((-1, 3))
        B = self.int.jacobian_csr()
        return t0 * dx, t0 * (B @ y[1]), B @ y[2]

    def _set_x_iterative(self, target, max_iter=20):
        """Newton back-transformation q(x) = target (peswrapper.py:749-839): faster than the geodesic for small
        feasible steps; returns None — positions restored — when it does not settle, and `set_x` integrates the
        geodesic instead.  Acceptance rules as in the reference: root-mean-square residual below 1e-8, or a
        stagnated iteration that at least halved it and ends below 1e-6; a residual that doubles, a stagnation
        above half the initial residual or an internal coordinate turning degenerate abort.  The Newton correction
        is B^+ (target - q) with the pseudo-inverse of the CURRENT geometry (spectral factor, cached per geometry)."""
        start = self.atoms.positions.copy()
        q0 = self.get_x()
        dq_wanted = target - q0
        g_int = self.curr.get('g')
        g_cart = self._get_factor().pinv_dot(g_int if g_int is not None else np.zeros_like(dq_wanted))

        def give_up():
            self.atoms.positions = start
            return None

        first = previous = None
        stalled = 0
        for sweep in range(max_iter):
            miss = self.wrap_dx(target - self.get_x())
            rms = float(np.sqrt(miss @ miss / len(miss)))
            first = rms if first is None else first
            if rms < 1e-8:
                break
            if rms > 2.0 * first:
                return give_up()
            if sweep > 3 and rms > 0.95 * previous:
                stalled += 1
                if stalled >= 3:
                    if rms > 0.5 * first:
                        return give_up()
                    break
            elif sweep > 3:
                stalled = 0
            previous = rms
            self.atoms.positions = self.atoms.positions + self._get_factor().pinv_dot(miss).reshape((-1, 3))
            if self.int.check_for_bad_internals() is not None:
                return give_up()
        miss = self.wrap_dx(target - self.get_x())
        if np.sqrt(miss @ miss / len(dq_wanted)) > 1e-6:
            return give_up()
        return dq_wanted, self.get_x() - q0, self.int.jacobian_csr() @ g_cart

    def set_x(self, target):
        res = self._set_x_iterative(target) if self.iterative_stepper else None
        dx_initial, dx_final_ode, g_final = res if res is not None else self._set_x_ode(target)
        q_after = self.int.calc().copy()
        moved = self._project_to_constraints()
        return dx_initial, self._add_proj_delta(dx_final_ode, q_after, moved), g_final

    def _add_proj_delta(self, dx_int_final, q_after_ode, proj_moved):                 # :903-926
        if not proj_moved:
            return dx_int_final
        return dx_int_final + self.int.wrap(self.int.calc() - q_after_ode)

    def _project_to_constraints(self, target_tol=1e-7, max_iter=8, safety_limit=0.05):  # :928-994
        if self.cons.residual().size == 0:
            return False
        moved = False
        for _ in range(max_iter):
            r = self.cons.residual()
            if np.linalg.norm(r, ord=np.inf) < target_tol:
                return moved
            drdx, Ucons, _, _ = self._compute_basis_int()
            if Ucons.shape[1] == 0:
                return moved
            s = np.linalg.lstsq(drdx @ Ucons, -r, rcond=None)[0]
            dx = self._get_factor().pinv_dot(Ucons @ s)
            if np.linalg.norm(dx, ord=np.inf) > safety_limit:
                return moved
            self.atoms.positions = self.atoms.positions + dx.reshape(-1, 3)
            moved = True
        return moved

    def get_x(self):
        x = self.int.calc()
        nd = self.int.ndihedrals
        if self.curr['x'] is not None and nd:
            # keep dihedrals continuous with the previous point instead of jumping by 2 pi (:996-1008)
            dx = x[-nd:] - self.curr['x'][-nd:]
            x[-nd:] = self.curr['x'][-nd:] + (dx + np.pi) % (2 * np.pi) - np.pi
        return x

    def wrap_dx(self, dx):
        return self.int.wrap(dx)

    # ---- constraints in internal space (:1011-1122) ------------------------------------------------------
    def _compute_Hc_int(self):
        if self.curr['L'] is None:
            raise RuntimeError("InternalPES.get_Hc() called with L=None.")
        fac = self._get_factor()
        n_dof = fac.shape[0]
        if self.curr['L'].size == 0:
            return np.zeros((n_dof, n_dof))
        D_cons = self.cons.hessian().ldot(self.curr['L'])
        L_int = fac.pinvT_dot(self.curr['L'] @ self.cons.jacobian())
        D_int = self.int.hessian().ldot(L_int)
        # Binv^T D Binv with D symmetric (3N x 3N)
        return fac.pinvT_dot(fac.pinvT_dot(D_cons - D_int).T)

    def get_Hc(self):
        key = self._state_hash()
        cached = self._Hc_cache.get(key)
        if cached is None:
            cached = self._compute_Hc_int()
            self._Hc_cache.put(key, cached)
        return cached

    def _has_curved_constraints(self):
        return True                      # in internal space even a fixed translation has a curvature term

    def get_drdx(self):
        return self._get_factor().pinvT_dot(PES.get_drdx(self).T).T     # dr/dq = dr/dx dx/dq

    def _compute_basis_int(self):
        fac = self._get_factor()
        Unred = Q = fac.Q
        n_int = Q.shape[0]
        cons_jac = self.cons.jacobian()
        if cons_jac.shape[0] == 0:
            return np.zeros((0, n_int)), np.zeros((n_int, 0)), Unred, Unred
        drdxnred = cons_jac @ fac.BinvQ                    # dr/dx B^+ Q
        Vcons, Vfree = _split_cons_subspace(drdxnred)
        return drdxnred @ Q.T, Unred @ Vcons, Unred, Unred @ Vfree

    def _calc_basis(self):
        key = self._state_hash()
        cached = self._basis_cache.get(key)
        if cached is None:
            cached = self._compute_basis_int()
            self._basis_cache.put(key, cached)
        return cached

    # ---- calculator boundary: Cartesian gradient -> internal (:1124-1127) ---------------------------------
    def eval(self):
        f, g_cart = PES.eval(self)
        return f, self._get_factor().pinvT_dot(g_cart)

    def get_df_pred(self, dx, g, H):                                                    # :1174-1181
        if H is None:
            return None
        # dx_r . (Unred^T H Unred) . dx_r = p . H p with p = Unred Unred^T dx: two thin products and ONE
        # Hessian-vector product instead of the reference's projected matrix
        Unred = self.get_Unred()
        dx_r, g_r = dx @ Unred, g @ Unred
        p = Unred @ dx_r
        return g_r @ dx_r + (p @ (H @ p)) / 2.

    def get_projected_forces(self):                                                     # :1183-1192
        g = self.get_g()
        Ufree = self.get_Ufree()
        B = self.curr.get('B')
        if B is None:
            B = self.int.jacobian_csr()
        return -np.asarray((Ufree @ (Ufree.T @ g)) @ B).reshape((-1, 3))

    def _update(self, feval=True):
        if not PES._update(self, feval=feval):
            return
        self.curr.update(B=self._get_factor().Bs)
        return True

    def kick(self, dx, diag=False, **diag_kwargs):
        # A geodesic that runs into a degenerate internal coordinate (an angle close to 0 or pi) stops there:
        # `bad_int` stays set and the caller — `Sella.step`, optimize.py:384-410 — rebuilds the coordinate system
        # from the geometry reached.
        return PES.kick(self, dx, diag=diag, **diag_kwargs)
