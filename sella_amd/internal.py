"""Internal-coordinate primitives and the `Constraints` container — the subset of
sella/internal.py the saddle-point path needs:

  * value / gradient / Hessian of translations, bonds, angles and dihedrals with periodic shift
    vectors, same definitions as internal.py:58-80 (`_bond_value`, `_angle_value`,
    `_dihedral_value`) and internal.py:466-470 (`_translation`);
  * `Constraints`: fix_translation / fix_bond / fix_angle / fix_dihedral with eq / lt / gt kinds,
    `residual()`, `jacobian()` (dense, internal.py:1780-1902), `hessian().ldot(L)`
    (internal.py:2189-2305, linalg.py:601-618), inequality bookkeeping (internal.py:2788-2823).

The reference differentiates these functions with JAX (CPU).  Here the derivatives are exact
too: second-order forward-mode (hyper-dual) arithmetic inside one HIP kernel per coordinate kind
(csrc/internals.hip), one thread per (coordinate, component) — no JAX.  Out of scope here: TRIC rotations,
cell derivatives, dummy atoms, automatic topology search.
"""
from functools import partialmethod

import numpy as np

from .device import get_context


class DuplicateInternalError(ValueError):
    pass


class DuplicateConstraintError(DuplicateInternalError):
    pass


# ------------------------------------------------------------------------------------------
# batched primitives: one device launch per kind (csrc/internals.hip, sella_internals_eval)
# ------------------------------------------------------------------------------------------
_NATOMS = {'bonds': 2, 'angles': 3, 'dihedrals': 4}


def evaluate_kind(kind, pos, tvec, tangent=None, hessian=True):
    """pos (nc, natoms, 3), tvec (nc, natoms-1, 3) -> value (nc,), grad (nc, natoms, 3),
    hess (nc, natoms, 3, natoms, 3) [, hvp (nc, natoms, 3) when `tangent` is given]
    — internal.py:85-97 / :106-135 batched over the coordinates of one kind."""
    na = _NATOMS[kind]
    pos = np.asarray(pos, dtype=np.float64).reshape(-1, na, 3)
    nc = pos.shape[0]
    if nc == 0:
        out = (np.zeros(0), np.zeros((0, na, 3)), np.zeros((0, na, 3, na, 3)))
        return out + (np.zeros((0, na, 3)),) if tangent is not None else out
    q, g, hv, H = get_context().internals_eval(pos, tvec, tangent, hessian=hessian)
    return (q, g, H, hv) if tangent is not None else (q, g, H)


# ------------------------------------------------------------------------------------------
# coordinate objects (identity / bookkeeping only; numerics are batched per kind)
# ------------------------------------------------------------------------------------------
class Coordinate:
    nindices = None
    kind = None

    def __init__(self, indices, ncvecs=None, **kwargs):
        self.indices = np.array(indices, dtype=np.int64)
        if self.nindices is not None and len(self.indices) != self.nindices:
            raise ValueError(f'{self.__class__.__name__} needs {self.nindices} atom indices')
        n = max(len(self.indices) - 1, 0)
        self.ncvecs = np.zeros((n, 3), dtype=np.int64) if ncvecs is None else np.array(ncvecs, dtype=np.int64).reshape((n, 3))
        self.kwargs = kwargs

    def reverse(self):
        return self.__class__(self.indices[::-1], -self.ncvecs[::-1], **self.kwargs)

    def __eq__(self, other):
        if not isinstance(other, self.__class__):
            return NotImplemented
        for cand in (other, other.reverse()):
            if np.array_equal(self.indices, cand.indices) and np.array_equal(self.ncvecs, cand.ncvecs):
                return True
        return False

    def __repr__(self):
        return f'{self.__class__.__name__}({self.indices.tolist()})'

    def calc(self, atoms):
        pos = atoms.positions[self.indices][None]
        tv = (self.ncvecs @ np.asarray(atoms.cell, dtype=float))[None]
        return float(evaluate_kind(self.kind, pos, tv)[0][0])


class Bond(Coordinate):
    nindices, kind = 2, 'bonds'


class Angle(Coordinate):
    nindices, kind = 3, 'angles'


class Dihedral(Coordinate):
    nindices, kind = 4, 'dihedrals'


class Translation(Coordinate):
    """Mean position of a set of atoms along one Cartesian axis (internal.py:466-493)."""
    kind = 'translations'

    def __init__(self, indices, dim, ncvecs=None):
        Coordinate.__init__(self, np.atleast_1d(indices))
        self.kwargs['dim'] = int(dim)
        self.ncvecs = np.zeros((0, 3), dtype=np.int64)

    def reverse(self):
        return self

    def __eq__(self, other):
        if not isinstance(other, self.__class__):
            return NotImplemented
        return self.kwargs['dim'] == other.kwargs['dim'] and set(self.indices.tolist()) == set(other.indices.tolist())

    def _dupkey(self):
        """What `__eq__` compares, hashable (Constraints._add)."""
        return (self.kwargs['dim'], frozenset(self.indices.tolist()))

    def calc(self, atoms):
        return float(atoms.positions[self.indices, self.kwargs['dim']].mean())


class RotationGenerator(Coordinate):
    """Infinitesimal rigid rotation of a group of atoms about one Cartesian axis through their centroid."""
    natoms = 0

    def __init__(self, indices, axis):
        self.indices = np.array(indices, dtype=np.int64).ravel()
        self.ncvecs = np.zeros((0, 3), dtype=np.int64)
        self.kwargs = dict(axis=int(axis))

    def reverse(self):
        return self

    def __eq__(self, other):
        return (isinstance(other, RotationGenerator) and self.kwargs['axis'] == other.kwargs['axis']
                and np.array_equal(np.sort(self.indices), np.sort(other.indices)))

    def __repr__(self):
        return f'RotationGenerator(axis={self.kwargs["axis"]}, natoms={len(self.indices)})'

    def calc(self, atoms):
        return 0.0

    def generator(self, atoms):
        """(1, 3N) row: d(theta_axis)/dx for a rigid rotation, unit norm (zero row for a degenerate group)."""
        pos = np.asarray(atoms.positions, dtype=np.float64)[self.indices]
        rel = pos - pos.mean(axis=0)
        e = np.zeros(3)
        e[self.kwargs['axis']] = 1.0
        g = np.cross(e, rel)
        row = np.zeros((1, 3 * len(atoms)))
        row[0, (3 * self.indices[:, None] + np.arange(3)[None, :]).ravel()] = g.ravel()
        nrm = np.linalg.norm(row)
        return row / nrm if nrm > 1e-12 else row


class _HessianStack:
    """Per-coordinate Hessians held as blocks — (dof index array (nc, m), hess (nc, m, m)) per group of
    coordinates that touch the same number of atoms, rows numbered through the groups in order — with the
    contractions of SparseInternalHessians (sella/linalg.py:520-646): `ldot(v) = sum_i v_i H_i` as a dense
    (ndof, ndof) matrix (:601-618), `rdot(x)_i = H_i x` (:620-640), `ddot(u, x)_i = u . H_i x` (:642-646),
    `asarray()` the dense stack (:595-596).  An atom that occurs twice in one coordinate (a periodic image of
    itself) accumulates, as the reference's np.add.at / bincount scatter does."""

    def __init__(self, ndof, blocks):
        self.ndof = ndof
        self.blocks = blocks
        self.shape = (sum(len(d) for d, _ in blocks), ndof, ndof)

    def _groups(self):
        off = 0
        for dofs, H in self.blocks:
            nc = len(dofs)
            if nc and dofs.shape[1]:
                yield off, dofs, H
            off += nc

    def ldot(self, v):
        out = np.zeros((self.ndof, self.ndof))
        for off, dofs, H in self._groups():
            nc, m = dofs.shape
            w = np.asarray(v[off:off + nc])
            rows = np.repeat(dofs[:, :, None], m, axis=2)
            cols = np.repeat(dofs[:, None, :], m, axis=1)
            np.add.at(out, (rows.ravel(), cols.ravel()), (w[:, None, None] * H).ravel())
        return out

    def rdot(self, x):
        x = np.asarray(x, dtype=np.float64).ravel()
        out = np.zeros((self.shape[0], self.ndof))
        for off, dofs, H in self._groups():
            nc = len(dofs)
            hv = np.einsum('iab,ib->ia', H, x[dofs])
            np.add.at(out, (np.arange(off, off + nc)[:, None], dofs), hv)
        return out

    def ddot(self, u, x):
        u, x = np.asarray(u, dtype=np.float64).ravel(), np.asarray(x, dtype=np.float64).ravel()
        out = np.zeros(self.shape[0])
        for off, dofs, H in self._groups():
            out[off:off + len(dofs)] = np.einsum('ia,iab,ib->i', u[dofs], H, x[dofs])
        return out

    def asarray(self):
        out = np.zeros(self.shape)
        for off, dofs, H in self._groups():
            nc, m = dofs.shape
            i = np.repeat(np.arange(off, off + nc), m * m)
            rows = np.repeat(dofs[:, :, None], m, axis=2).ravel()
            cols = np.repeat(dofs[:, None, :], m, axis=1).ravel()
            np.add.at(out, (i, rows, cols), H.ravel())
        return out

    def __array__(self, dtype=None, copy=None):
        """`np.asarray(stack)`: the dense (ncoords, ndof, ndof) array, like the reference's class."""
        out = self.asarray()
        return out if dtype is None else out.astype(dtype, copy=False)


class _JacobianStack:
    """Per-coordinate gradients as blocks (dof index array (nc, m), grad (nc, m)): the scatter and the two
    products of SparseInternalJacobian (sella/linalg.py:362-401)."""

    def __init__(self, ndof, blocks):
        self.ndof = ndof
        self.blocks = blocks
        self.shape = (sum(len(d) for d, _ in blocks), ndof)

    def _groups(self):
        off = 0
        for dofs, G in self.blocks:
            nc = len(dofs)
            if nc and dofs.shape[1]:
                yield off, dofs, G
            off += nc

    def asarray(self):
        B = np.zeros(self.shape)
        for off, dofs, G in self._groups():
            np.add.at(B, (np.arange(off, off + len(dofs))[:, None], dofs), G)
        return B

    def matvec(self, x):
        x = np.asarray(x, dtype=np.float64).ravel()
        out = np.zeros(self.shape[0])
        for off, dofs, G in self._groups():
            out[off:off + len(dofs)] = np.einsum('ia,ia->i', G, x[dofs])
        return out

    def rmatvec(self, y):
        y = np.asarray(y, dtype=np.float64).ravel()
        out = np.zeros(self.ndof)
        for off, dofs, G in self._groups():
            np.add.at(out, dofs, y[off:off + len(dofs), None] * G)
        return out


class Constraints:
    _names = ('translations', 'bonds', 'angles', 'dihedrals', 'other', 'rotations')

    def __init__(self, atoms, dummies=None, dinds=None, ignore_rotation=True):
        self.atoms = atoms
        self.internals = {k: [] for k in self._names}
        self._targets = {k: [] for k in self._names}
        self._active = {k: [] for k in self._names}
        self._kind = {k: [] for k in self._names}
        self.ignore_rotation = ignore_rotation
        self._ver = 0                 # bumped whenever the set of (active) constraints or a target changes
        self._memo_store = {}
        for cons in getattr(atoms, 'constraints', None) or []:
            self.merge_ase_constraint(cons)

    def _memo(self, key, fn):
        """Per-structure cache: a slab search carries thousands of single-atom pins, and the optimizer asks for
        counts, active lists and targets several times per step."""
        hit = self._memo_store.get(key)
        if hit is None or hit[0] != self._ver:
            hit = (self._ver, fn())
            self._memo_store[key] = hit
        return hit[1]

    # ---- bookkeeping ---------------------------------------------------------------------
    @property
    def all_atoms(self):
        return self.atoms

    @property
    def natoms(self):
        return len(self.atoms)

    @property
    def ndof(self):
        return 3 * self.natoms

    def _count(self, name):
        return self._memo(('count', name), lambda: int(sum(self._active[name])))

    ntrans = property(lambda self: self._count('translations'))
    nbonds = property(lambda self: self._count('bonds'))
    nangles = property(lambda self: self._count('angles'))
    ndihedrals = property(lambda self: self._count('dihedrals'))
    nother = property(lambda self: 0)
    nrotations = property(lambda self: self._count('rotations'))

    @property
    def nint(self):
        return self.ntrans + self.nbonds + self.nangles + self.ndihedrals + self.nrotations

    def copy(self):
        new = self.__class__(self.atoms, ignore_rotation=self.ignore_rotation)
        for name in self._names:
            new.internals[name] = list(self.internals[name])
            new._targets[name] = list(self._targets[name])
            new._active[name] = list(self._active[name])
            new._kind[name] = list(self._kind[name])
        return new

    def _active_list(self, name):
        return self._memo(('active', name),
                          lambda: [c for c, a in zip(self.internals[name], self._active[name]) if a])

    @property
    def targets(self):
        def build():
            vec = []
            for name in self._names:
                vec += [t for t, a in zip(self._targets[name], self._active[name]) if a]
            return np.array(vec, dtype=np.float64)
        return self._memo('targets', build).copy()

    # ---- numerics (batched per kind) --------------------------------------------------------
    def _gather(self, name):
        coords = self._active_list(name)
        na = _NATOMS[name]
        idx = np.array([c.indices for c in coords], dtype=np.int64).reshape((len(coords), na))
        ncv = np.array([c.ncvecs for c in coords], dtype=np.float64).reshape((len(coords), na - 1, 3))
        pos = self.atoms.positions[idx] if len(coords) else np.zeros((0, na, 3))
        tvec = ncv @ np.asarray(self.atoms.cell, dtype=float)
        return idx, pos, tvec

    def _translation_arrays(self):
        """Translations as a sparse averaging operator: (row, dof, weight) triplets, rebuilt only when the set of
        active translation constraints changes (thousands of single-atom pins on a slab: no Python loop per call)."""
        trans = self._active_list('translations')          # memoised: the same list object while the set is unchanged
        key = (self._ver, id(trans))
        hit = getattr(self, '_trans_cache', None)
        if hit is None or hit[0] != key:
            rows, dofs, wts = [], [], []
            for r, c in enumerate(trans):
                k = len(c.indices)
                rows += [r] * k
                dofs += (3 * c.indices + c.kwargs['dim']).tolist()
                wts += [1.0 / k] * k
            hit = (key, np.array(rows, dtype=np.int64), np.array(dofs, dtype=np.int64), np.array(wts), len(trans))
            self._trans_cache = hit
        return hit[1:]

    def calc(self):
        rows, dofs, wts, nt = self._translation_arrays()
        x = self.atoms.positions.ravel()
        vals = [np.bincount(rows, weights=wts * x[dofs], minlength=nt) if nt else np.zeros(0)]
        for name in ('bonds', 'angles', 'dihedrals'):
            idx, pos, tvec = self._gather(name)
            vals.append(evaluate_kind(name, pos, tvec)[0])
        vals.append(np.zeros(self.nrotations))              # linearised rotations: value 0 at every geometry
        return np.concatenate(vals) if vals else np.zeros(0)

    def wrap(self, vec):
        """Dihedral differences live on the circle (internal.py:2577-2587)."""
        nd = self.ndihedrals
        if nd:
            lo = self.ntrans + self.nbonds + self.nangles
            vec[lo:lo + nd] = (vec[lo:lo + nd] + np.pi) % (2 * np.pi) - np.pi
        return vec

    def residual(self):
        return self.wrap(self.calc() - self.targets)

    def jacobian(self):
        """Dense (nactive, 3N) constraint Jacobian."""
        n3 = self.ndof
        rows, dofs, wts, nt = self._translation_arrays()
        only_trans = (not any(len(self._gather(name)[0]) for name in ('bonds', 'angles', 'dihedrals'))
                      and self.nrotations == 0)
        if only_trans:
            # translation constraints do not depend on the geometry: the SAME (read-only) array is handed out as
            # long as the constraint set is unchanged, so everything downstream that keys on the object
            # (pinned-coordinate analysis, bases, device copies) is computed once per search, not once per step
            hit = getattr(self, '_jac_cache', None)
            if hit is not None and hit[0] is rows and hit[1].shape == (nt, n3):      # `rows` is itself cached
                return hit[1]
        J = np.zeros((nt, n3))
        J[rows, dofs] = wts
        if only_trans:
            J.setflags(write=False)
            self._jac_cache = (rows, J)
            return J
        for name in ('bonds', 'angles', 'dihedrals'):
            idx, pos, tvec = self._gather(name)
            if len(idx) == 0:
                continue
            grad = evaluate_kind(name, pos, tvec)[1]            # (nc, na, 3)
            block = np.zeros((len(idx), n3))
            dofs = (3 * idx[:, :, None] + np.arange(3)[None, None, :]).reshape(len(idx), -1)
            np.add.at(block, (np.arange(len(idx))[:, None], dofs), grad.reshape(len(idx), -1))
            J = np.vstack([J, block])
        rot = self._active_list('rotations')
        if rot:
            J = np.vstack([J] + [r.generator(self.atoms) for r in rot])
        return J

    def hessian(self):
        blocks = [(np.zeros((self.ntrans, 0), dtype=np.int64), np.zeros((self.ntrans, 0, 0)))]
        for name in ('bonds', 'angles', 'dihedrals'):
            idx, pos, tvec = self._gather(name)
            nc = len(idx)
            na = idx.shape[1] if nc else 0
            H = evaluate_kind(name, pos, tvec)[2].reshape(nc, 3 * na, 3 * na) if nc else np.zeros((0, 0, 0))
            dofs = (3 * idx[:, :, None] + np.arange(3)[None, None, :]).reshape(nc, -1) if nc else np.zeros((0, 0), dtype=np.int64)
            blocks.append((dofs, H))
        nrot = self.nrotations                              # linear in x at fixed generators: no curvature term
        blocks.append((np.zeros((nrot, 0), dtype=np.int64), np.zeros((nrot, 0, 0))))
        return _HessianStack(self.ndof, blocks)

    # ---- inequality bookkeeping (internal.py:2788-2823) ----------------------------------------
    def has_inequalities(self):
        return self._memo('has_ineq', lambda: any(k in ('lt', 'gt') for name in self._names for k in self._kind[name]))

    def _set_active(self, name, i, value):
        if self._active[name][i] != value:
            self._active[name][i] = value
            self._ver += 1

    def disable_satisfied_inequalities(self):
        if not self.has_inequalities():
            return
        for name in self._names:
            for i, (coord, kind, target) in enumerate(zip(self.internals[name], self._kind[name], self._targets[name])):
                if kind == 'lt' and coord.calc(self.atoms) <= target:
                    self._set_active(name, i, False)
                elif kind == 'gt' and coord.calc(self.atoms) >= target:
                    self._set_active(name, i, False)
                else:
                    self._set_active(name, i, True)

    def validate_inequalities(self):
        if not self.has_inequalities():
            return True
        all_valid = True
        for name in self._names:
            for i, (coord, kind, target) in enumerate(zip(self.internals[name], self._kind[name], self._targets[name])):
                if self._active[name][i]:
                    continue
                val = coord.calc(self.atoms)
                if (kind == 'lt' and val > target) or (kind == 'gt' and val < target):
                    self._set_active(name, i, True)
                    all_valid = False
        return all_valid

    # ---- adding constraints (internal.py:2861-2955) ----------------------------------------------
    def _add(self, name, new, target, kind, replace_ok):
        # duplicate look-up through a dictionary where the coordinate names itself (`_dupkey`): pinning a slab atom by
        # atom is hundreds of additions, and a linear search with `__eq__` made that quadratic — 20 ms of interpreter
        # time per 256-atom ensemble member, serial under the interpreter lock however many host threads run members
        keyf = getattr(new, '_dupkey', None)
        try:
            if keyf is None:
                idx = self.internals[name].index(new)
            else:
                table = self.__dict__.setdefault('_dup', {}).setdefault(name, None)
                lst = self.internals[name]
                if table is None or table[0] != len(lst):
                    table = [len(lst), {c._dupkey(): i for i, c in enumerate(lst) if hasattr(c, '_dupkey')}]
                    if len(table[1]) != len(lst):
                        table = None                       # mixed list: the linear search
                if table is None:
                    idx = lst.index(new)
                else:
                    idx = table[1].get(keyf())
                    if idx is None:
                        table[1][keyf()] = len(lst)
                        table[0] = len(lst) + 1
                        self._dup[name] = table
                        raise ValueError
                    self._dup[name] = table
        except ValueError:
            self.internals[name].append(new)
            self._targets[name].append(target)
            self._active[name].append(True)
            self._kind[name].append(kind)
            self._ver += 1
            return
        if replace_ok:
            self._targets[name][idx] = target
            self._kind[name][idx] = kind
            self._ver += 1
            return
        raise DuplicateConstraintError(f'Coordinate {new} is already fixed to target {self._targets[name][idx]}')

    def fix_translation(self, index=None, dim=None, target=None, replace_ok=True):
        if isinstance(index, Translation):
            if dim is not None:
                raise ValueError('"dim" keyword cannot be used with explicit Translation')
            new = index
        else:
            if index is None:
                index = np.arange(self.natoms)
            if np.isscalar(index):
                index = np.array((index,))
            if dim is None:
                if target is not None:
                    raise ValueError('"target" keyword requires explicit "dim"!')
                for d in range(3):
                    self.fix_translation(index, dim=d, replace_ok=replace_ok)
                return
            new = Translation(index, dim)
        if target is None:
            target = new.calc(self.atoms)
        self._add('translations', new, target, 'eq', replace_ok)

    def fix_rotation(self, indices=None, axis=None):
        """Remove the rigid rotation of `indices` (default: all atoms) about `axis` (0, 1, 2; default: all three)
        from the free subspace (internal.py:2871-2893 adds a TRIC `Rotation` coordinate there; peswrapper.py:244-253
        does so for every non-periodic system).  Here the constraint is the LINEARISED rotation: its Jacobian row is
        the infinitesimal generator e_axis x (r_i - centroid), normalised, re-evaluated at every geometry, with
        residual and curvature identically zero — which is what the saddle search needs from it (the rotational
        soft modes leave the Davidson / P-RFO subspace); finite rotation targets (TRIC proper) are out of scope."""
        if indices is None:
            indices = np.arange(self.natoms)
        axes = range(3) if axis is None else (int(axis),)
        for ax in axes:
            self._add('rotations', RotationGenerator(indices, ax), 0.0, 'eq', True)

    def _fix_internal(self, cls, name, conv, indices, ncvecs=None, mic=None, target=None,
                      comparator='eq', replace_ok=True):
        new = indices if isinstance(indices, cls) else cls(indices, ncvecs=ncvecs)
        target = new.calc(self.atoms) if target is None else target * conv
        self._add(name, new, target, comparator, replace_ok)

    fix_bond = partialmethod(_fix_internal, Bond, 'bonds', 1.)
    fix_angle = partialmethod(_fix_internal, Angle, 'angles', np.pi / 180.)
    fix_dihedral = partialmethod(_fix_internal, Dihedral, 'dihedrals', np.pi / 180.)

    def fix_other(self, *args, **kwargs):
        raise NotImplementedError('user-defined constraint functions are differentiated with JAX in the reference '
                                  '(internal.py:2932-2955); this build has closed-form kernels for translations, '
                                  'bonds, angles and dihedrals only')

    def merge_ase_constraint(self, cons):
        """FixAtoms / FixCom equivalents by duck typing (internal.py:2981-3030)."""
        name = cons.__class__.__name__
        if name == 'FixAtoms':
            for index in np.atleast_1d(cons.index):
                self.fix_translation(int(index))
        elif name == 'FixCom':
            self.fix_translation()
        else:
            raise NotImplementedError(f'ASE constraint {name} is not supported by this path')


# ------------------------------------------------------------------------------------------
# array-based redundant internal coordinates (the numerical core of BaseInternals,
# sella/internal.py:1362-1902, 2189-2587: calc / jacobian / hessian ldot / hessian_rdot / wrap)
# ------------------------------------------------------------------------------------------
class InternalCoordinates:
    """Bonds, angles and dihedrals of one structure as index arrays, evaluated in batches on the device.

    The reference keeps a Python object per coordinate and builds padded batch arrays from them
    (`_build_batched_arrays`, internal.py:1362-1529); here the index arrays ARE the representation:
    `bonds (nb, 2)`, `angles (na, 3)`, `dihedrals (nd, 4)` atom indices plus integer cell offsets
    `*_ncvecs (n, natoms-1, 3)` for periodic images.  Order of the coordinates: bonds, angles, dihedrals.
    Automatic topology search, dummy atoms and TRIC rotations are not part of this class.
    """
    _order = ('bonds', 'angles', 'dihedrals')

    def __init__(self, atoms, bonds=None, angles=None, dihedrals=None, bond_ncvecs=None, angle_ncvecs=None,
                 dihedral_ncvecs=None):
        self.atoms = atoms
        self.idx, self.ncv = {}, {}
        for name, arr, ncv in (('bonds', bonds, bond_ncvecs), ('angles', angles, angle_ncvecs),
                               ('dihedrals', dihedrals, dihedral_ncvecs)):
            na = _NATOMS[name]
            a = np.zeros((0, na), dtype=np.int64) if arr is None else np.asarray(arr, dtype=np.int64).reshape(-1, na)
            v = np.zeros((len(a), na - 1, 3)) if ncv is None else np.asarray(ncv, dtype=np.float64).reshape(len(a), na - 1, 3)
            self.idx[name], self.ncv[name] = a, v

    ndof = property(lambda self: 3 * len(self.atoms))
    nint = property(lambda self: sum(len(self.idx[k]) for k in self._order))

    def _batch(self, name):
        idx = self.idx[name]
        pos = self.atoms.positions[idx]
        tvec = self.ncv[name] @ np.asarray(self.atoms.cell, dtype=np.float64)
        dofs = (3 * idx[:, :, None] + np.arange(3)[None, None, :]).reshape(len(idx), 3 * _NATOMS[name])
        return pos, tvec, dofs

    def calc(self):
        """q(x) (internal.py:1735-1778)."""
        return np.concatenate([evaluate_kind(k, *self._batch(k)[:2], hessian=False)[0] for k in self._order])

    def wrap(self, vec):
        """Map dihedral differences into (-pi, pi] (internal.py:2577-2587)."""
        nd = len(self.idx['dihedrals'])
        if nd:
            vec = np.array(vec, dtype=np.float64)
            vec[-nd:] = (vec[-nd:] + np.pi) % (2 * np.pi) - np.pi
        return vec

    def jacobian_blocks(self):
        """The B-matrix as per-kind gradient blocks (`_JacobianStack`)."""
        blocks = []
        for k in self._order:
            pos, tvec, dofs = self._batch(k)
            nc = len(pos)
            g = evaluate_kind(k, pos, tvec, hessian=False)[1].reshape(nc, -1) if nc else np.zeros((0, dofs.shape[1]))
            blocks.append((dofs, g))
        return _JacobianStack(self.ndof, blocks)

    def jacobian(self):
        """Dense Wilson B-matrix dq/dx, (nint, 3N) (internal.py:1780-1902)."""
        return self.jacobian_blocks().asarray()

    def jacobian_csr(self):
        """The same B-matrix in CSR form, 6 / 9 / 12 stored entries per row (the sparsity the reference keeps
        for D(v) only, internal.py:1481-1527) — what `InternalPES` works with, so that nothing of size
        nint x 3N is ever dense on the way to the pseudo-inverse."""
        from scipy.sparse import csr_matrix
        data, cols, counts = [], [], []
        for k in self._order:
            pos, tvec, dofs = self._batch(k)
            nc = len(pos)
            if nc:
                data.append(evaluate_kind(k, pos, tvec, hessian=False)[1].reshape(-1))
                cols.append(dofs.reshape(-1))
                counts.append(np.full(nc, dofs.shape[1], dtype=np.int64))
        if not data:
            return csr_matrix((0, self.ndof))
        indptr = np.concatenate([[0], np.cumsum(np.concatenate(counts))])
        return csr_matrix((np.concatenate(data), np.concatenate(cols), indptr), shape=(self.nint, self.ndof))

    def hessian_rdot_mult(self, v, W):
        """D(v) @ W for W (3N, k) without forming D(v): (nint, k)."""
        v = np.asarray(v, dtype=np.float64).ravel()
        W = np.asarray(W, dtype=np.float64).reshape(self.ndof, -1)
        out = np.zeros((self.nint, W.shape[1]))
        row = 0
        for k in self._order:
            pos, tvec, dofs = self._batch(k)
            nc = len(pos)
            if nc:
                tan = v[dofs].reshape(pos.shape)
                hv = evaluate_kind(k, pos, tvec, tangent=tan, hessian=False)[3].reshape(nc, -1)
                out[row:row + nc] = np.einsum('ia,iak->ik', hv, W[dofs])
            row += nc
        return out

    def hessian_rdot(self, v):
        """D(v)_i = H_i v as a dense (nint, 3N) matrix (internal.py:2307-2575: one HVP per coordinate, from the
        device's Hessian-vector kernel; the scatter is `_JacobianStack`'s)."""
        v = np.asarray(v, dtype=np.float64).ravel()
        blocks = []
        for k in self._order:
            pos, tvec, dofs = self._batch(k)
            nc = len(pos)
            if nc:
                tan = v[dofs].reshape(pos.shape)
                hv = evaluate_kind(k, pos, tvec, tangent=tan, hessian=False)[3].reshape(nc, -1)
            else:
                hv = np.zeros((0, dofs.shape[1]))
            blocks.append((dofs, hv))
        return _JacobianStack(self.ndof, blocks).asarray()

    def hessian(self):
        """Per-coordinate Hessian blocks with the `ldot` contraction (internal.py:2189-2305)."""
        blocks = []
        for k in self._order:
            pos, tvec, dofs = self._batch(k)
            nc, m = len(pos), 3 * _NATOMS[k]
            H = evaluate_kind(k, pos, tvec)[2].reshape(nc, m, m) if nc else np.zeros((0, m, m))
            blocks.append((dofs, H))
        nrot = self.nrotations                              # linear in x at fixed generators: no curvature term
        blocks.append((np.zeros((nrot, 0), dtype=np.int64), np.zeros((nrot, 0, 0))))
        return _HessianStack(self.ndof, blocks)


def neighbour_bonds(atoms, rcut):
    """All pairs closer than rcut under the minimum-image convention of the periodic directions:
    (bonds (nb, 2), ncvecs (nb, 1, 3)) with i < j.  A vectorised stand-in for `find_all_bonds`
    (internal.py:3260-3400) on regular lattices; O(N^2) memory."""
    pos = atoms.positions
    n = len(pos)
    iu = np.triu_indices(n, 1)
    d = pos[iu[1]] - pos[iu[0]]
    cell = np.asarray(atoms.cell, dtype=np.float64)
    shift = np.zeros((len(d), 3))
    per = np.where(atoms.pbc)[0]
    if len(per):
        C = cell[per]
        s = -np.round(d @ np.linalg.pinv(C))
        shift[:, per] = s
        d = d + s @ C
    m = np.linalg.norm(d, axis=1) < rcut
    return np.stack([iu[0][m], iu[1][m]], axis=1), shift[m][:, None, :]


def angles_from_bonds(bonds, ncvecs):
    """Every pair of bonds sharing an atom gives an angle with that atom at the vertex
    (internal.py:3402-3470): (angles (na, 3), ncvecs (na, 2, 3))."""
    nb = len(bonds)
    # directed half-bonds centre -> neighbour with the image offset seen from the centre
    cen = np.concatenate([bonds[:, 0], bonds[:, 1]])
    nei = np.concatenate([bonds[:, 1], bonds[:, 0]])
    off = np.concatenate([ncvecs[:, 0], -ncvecs[:, 0]])
    order = np.argsort(cen, kind='stable')
    cen, nei, off = cen[order], nei[order], off[order]
    starts = np.flatnonzero(np.r_[True, cen[1:] != cen[:-1], True])
    ang, ncv = [], []
    for a, b in zip(starts[:-1], starts[1:]):
        k = b - a
        if k < 2:
            continue
        i, j = np.triu_indices(k, 1)
        # angle (n_i, centre, n_j): first vector centre - n_i image = -off_i, second n_j - centre = off_j
        ang.append(np.stack([nei[a + i], np.full(len(i), cen[a]), nei[a + j]], axis=1))
        ncv.append(np.stack([-off[a + i], off[a + j]], axis=1))
    if not ang:
        return np.zeros((0, 3), dtype=np.int64), np.zeros((0, 2, 3))
    return np.concatenate(ang), np.concatenate(ncv)


# ------------------------------------------------------------------------------------------
# what InternalPES needs on top of the numerical core: counts, constraints, guess Hessian, topology
# ------------------------------------------------------------------------------------------
_HARTREE, _BOHR = 27.211386245988, 0.529177210903        # eV, Angstrom
# covalent radii (Cordero et al. 2008, the table ase.data.covalent_radii carries), Angstrom
_RCOV = dict(H=0.31, He=0.28, Li=1.28, Be=0.96, B=0.84, C=0.76, N=0.71, O=0.66, F=0.57, Ne=0.58, Na=1.66, Mg=1.41,
             Al=1.21, Si=1.11, P=1.07, S=1.05, Cl=1.02, Ar=1.06, K=2.03, Ca=1.76, Ti=1.60, Fe=1.32, Co=1.26, Ni=1.24,
             Cu=1.32, Zn=1.22, Br=1.20, Pd=1.39, Ag=1.45, Pt=1.36, Au=1.36)


def covalent_radius(symbol, default=1.0):
    return _RCOV.get(symbol, default)


def _ic_counts(self):
    return dict(ntrans=0, nbonds=len(self.idx['bonds']), nangles=len(self.idx['angles']),
                ndihedrals=len(self.idx['dihedrals']), nother=0, nrotations=0)


for _name in ('ntrans', 'nbonds', 'nangles', 'ndihedrals', 'nother', 'nrotations'):
    setattr(InternalCoordinates, _name, property(lambda self, _n=_name: _ic_counts(self)[_n]))


def _ic_copy(self):
    new = InternalCoordinates(self.atoms, self.idx['bonds'], self.idx['angles'], self.idx['dihedrals'],
                              self.ncv['bonds'], self.ncv['angles'], self.ncv['dihedrals'])
    new.cons = self.cons.copy() if getattr(self, 'cons', None) is not None else None
    return new


def _ic_radii(self):
    return np.array([covalent_radius(s) for s in self.atoms.symbols])


def _ic_guess_hessian(self, diagonal_only=False):
    """Diagonal model Hessian in the internal coordinates (internal.py:3738-3820: the Schlegel-type
    exponential formulas of `_h0_bond`, `_h0_angle`, `_h0_dihedral`)."""
    rc = _ic_radii(self)
    q = self.calc()
    nb, na, nd = self.nbonds, self.nangles, self.ndihedrals
    h0 = np.zeros(self.nint)
    b = self.idx['bonds']
    rcov = rc[b].sum(axis=1) if nb else np.zeros(0)
    h0[:nb] = 0.3601 * np.exp(-1.944 * (q[:nb] - rcov) / _BOHR) * _HARTREE / _BOHR ** 2
    nbonds_of = np.bincount(b.ravel(), minlength=len(self.atoms)) if nb else np.zeros(len(self.atoms), dtype=int)
    if na:
        a = self.idx['angles']
        pa, ta, _ = self._batch('angles')
        rab = np.linalg.norm(pa[:, 1] - pa[:, 0] + ta[:, 0], axis=1)
        rbc = np.linalg.norm(pa[:, 2] - pa[:, 1] + ta[:, 1], axis=1)
        cab, cbc = rc[a[:, 0]] + rc[a[:, 1]], rc[a[:, 1]] + rc[a[:, 2]]
        h0[nb:nb + na] = (0.089 + 0.11 * np.exp(-0.44 * (rab + rbc - cab - cbc) / _BOHR)
                          / (cab * cbc / _BOHR ** 2) ** (-0.42)) * _HARTREE
    if nd:
        d = self.idx['dihedrals']
        pd, td, _ = self._batch('dihedrals')
        rbc = np.linalg.norm(pd[:, 2] - pd[:, 1] + td[:, 1], axis=1)
        cbc = rc[d[:, 1]] + rc[d[:, 2]]
        L = nbonds_of[d[:, 1]] + nbonds_of[d[:, 2]] - 2
        h0[nb + na:] = (0.0015 + 14.0 * np.maximum(L, 0) ** 0.57 * np.exp(-2.85 * (rbc - cbc) / _BOHR)
                        / (rbc * cbc / _BOHR ** 2) ** 4.00) * _HARTREE
    return np.abs(h0) if diagonal_only else np.diag(np.abs(h0))


def _ic_check_bad(self, tol=np.pi / 12):
    """Angles within `tol` (15 degrees, `Internals.atol`) of 0 or pi make the coordinate system degenerate
    (internal.py:3704-3736); the caller stops the step there and `Sella.step` rebuilds the internals."""
    if not self.nangles:
        return None
    pos, tvec, _ = self._batch('angles')
    ang = evaluate_kind('angles', pos, tvec, hessian=False)[0]
    bad = np.flatnonzero((ang > np.pi - tol) | (ang < tol))
    return bad if len(bad) else None


def _fragments(natoms, bonds):
    """Connected components of the bond graph: label per atom (union-find, path halving)."""
    parent = np.arange(natoms)

    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i
    for i, j in bonds:
        ri, rj = find(int(i)), find(int(j))
        if ri != rj:
            parent[max(ri, rj)] = min(ri, rj)
    return np.array([find(i) for i in range(natoms)])


def _ic_from_atoms(cls, atoms, cons=None, scale=1.25, dihedrals=True, atol=15.):
    """Automatic redundant internals, the topology search of sella/internal.py:3366-3671 as array code:

    * bonds: pairs closer than scale * (r_cov,i + r_cov,j); while the bond graph is disconnected the scale grows by
      5 % and bonds BETWEEN fragments are added (`find_all_bonds`, :3366-3423; minimum-image convention);
    * angles: every pair of bonds at an atom whose angle lies in (atol, pi - atol), atol = 15 degrees; a (nearly)
      linear one at an atom with a third neighbour is replaced by the improper dihedral through that neighbour
      (`find_all_angles`, :3458-3573; the two-neighbour case needs a dummy atom there — out of scope, the angle
      is simply left out);
    * dihedrals: every chain a-b-c-d of two kept angles sharing the bond b-c (`find_all_dihedrals`, :3575-3600), plus
      one improper n0-c-n1-n2 for centres with 3 or 4 neighbours that no proper dihedral passes through
      (:3602-3660: keeps the Jacobian well conditioned at planar geometries)."""
    atol = atol * np.pi / 180.
    natoms = len(atoms)
    rc = np.array([covalent_radius(s) for s in atoms.symbols])
    cell = np.asarray(atoms.cell, dtype=np.float64)
    pos = atoms.positions
    allp, allv = neighbour_bonds(atoms, np.inf)                          # every pair once, minimum image
    if len(allp):
        dist = np.linalg.norm(pos[allp[:, 1]] - pos[allp[:, 0]] + allv[:, 0] @ cell, axis=1)
        reach = rc[allp[:, 0]] + rc[allp[:, 1]]
        have = dist <= scale * reach
        for _ in range(200):
            labels = _fragments(natoms, allp[have])
            if len(np.unique(labels)) == 1:
                break
            scale *= 1.05
            have |= (labels[allp[:, 0]] != labels[allp[:, 1]]) & (dist <= scale * reach)
        bonds, bncv = allp[have], allv[have]
    else:
        bonds, bncv = np.zeros((0, 2), dtype=np.int64), np.zeros((0, 1, 3))
    angles, ancv = angles_from_bonds(bonds, bncv)
    dl, dv, seen = [], [], set()

    def add_dihedral(idx4, ncv3):
        a, b2, c, d = (int(v) for v in idx4)
        key = (a, b2, c, d) if (a, b2) < (d, c) else (d, c, b2, a)
        if key not in seen:
            seen.add(key)
            dl.append([a, b2, c, d])
            dv.append([np.asarray(v, dtype=np.float64) for v in ncv3])

    if len(angles):
        aval = cls(atoms, angles=angles, angle_ncvecs=ancv).calc()
        ok = (aval > atol) & (aval < np.pi - atol)
        # neighbour lists (neighbour, image offset seen from the centre) in bond order, both directions
        nbrs = [[] for _ in range(natoms)]
        for (i, j), v in zip(bonds, bncv[:, 0]):
            nbrs[int(i)].append((int(j), v))
            nbrs[int(j)].append((int(i), -v))
        if dihedrals:
            for (n1, c, n2), (v1, v2) in zip(angles[~ok], ancv[~ok]):
                # linear n1-c-n2 with a third neighbour n3: improper (n1, c, n3, n2), :3556-3573
                for n3, v3 in nbrs[int(c)]:
                    if (n3 == n1 and np.array_equal(v3, -v1)) or (n3 == n2 and np.array_equal(v3, v2)):
                        continue
                    add_dihedral((n1, c, n3, n2), (v1, v3, v2 - v3))
                    break
        angles, ancv = angles[ok], ancv[ok]
    if dihedrals and len(angles):
        # proper dihedrals: join angles (a, b, c) and (b, c, d) over the shared bond b-c with consistent images
        by_bond = {}
        for k, (a, b2, c) in enumerate(angles):
            by_bond.setdefault((b2, c, tuple(ancv[k, 1])), []).append((a, ancv[k, 0]))           # ... a-b-c, bond b->c
            by_bond.setdefault((b2, a, tuple(-ancv[k, 0])), []).append((c, -ancv[k, 1]))         # reversed: c-b-a, bond b->a
        for (b2, c, off), lefts in by_bond.items():
            rights = by_bond.get((c, b2, tuple(-np.array(off))), [])
            for a, oa in lefts:
                for dd, od in rights:
                    if a == c or dd == b2:
                        continue
                    if a == dd and not np.any(np.asarray(oa) + np.array(off) - np.asarray(od)):
                        continue                                   # the same atom (same image) at both ends, :3593-3599
                    add_dihedral((a, b2, c, dd), (oa, np.array(off), -od))
        centres = set()
        for a, b2, c, d in dl:
            centres.update((b2, c))
        for centre in range(natoms):
            if len(nbrs[centre]) in (3, 4) and centre not in centres:
                (n0, v0), (n1, v1), (n2, v2) = nbrs[centre][:3]
                add_dihedral((n0, centre, n1, n2), (-v0, v1, v2 - v1))
    dih = np.array(dl, dtype=np.int64) if dl else np.zeros((0, 4), dtype=np.int64)
    dncv = np.array(dv, dtype=np.float64) if dl else np.zeros((0, 3, 3))
    ic = cls(atoms, bonds=bonds, angles=angles, dihedrals=dih, bond_ncvecs=bncv, angle_ncvecs=ancv,
             dihedral_ncvecs=dncv)
    ic.cons = cons if cons is not None else Constraints(atoms)
    return ic


InternalCoordinates.copy = _ic_copy
InternalCoordinates.guess_hessian = _ic_guess_hessian
InternalCoordinates.check_for_bad_internals = _ic_check_bad
InternalCoordinates.from_atoms = classmethod(_ic_from_atoms)
InternalCoordinates.cons = None
