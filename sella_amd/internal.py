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

    def calc(self, atoms):
        return float(atoms.positions[self.indices, self.kwargs['dim']].mean())


class _HessianStack:
    """Per-coordinate Hessians with the `ldot` contraction of SparseInternalHessians
    (sella/linalg.py:601-618): ldot(v) = sum_i v_i H_i as a dense (ndof, ndof) matrix."""

    def __init__(self, ndof, blocks):
        self.ndof = ndof
        self.blocks = blocks        # list of (dof index array (nc, m), hess (nc, m, m))

    def ldot(self, v):
        out = np.zeros((self.ndof, self.ndof))
        off = 0
        for dofs, H in self.blocks:
            nc = len(dofs)
            if nc:
                w = np.asarray(v[off:off + nc])
                rows = np.repeat(dofs[:, :, None], dofs.shape[1], axis=2)
                cols = np.repeat(dofs[:, None, :], dofs.shape[1], axis=1)
                np.add.at(out, (rows.ravel(), cols.ravel()), (w[:, None, None] * H).ravel())
            off += nc
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
        for cons in getattr(atoms, 'constraints', None) or []:
            self.merge_ase_constraint(cons)

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
        return int(sum(self._active[name]))

    ntrans = property(lambda self: self._count('translations'))
    nbonds = property(lambda self: self._count('bonds'))
    nangles = property(lambda self: self._count('angles'))
    ndihedrals = property(lambda self: self._count('dihedrals'))
    nother = property(lambda self: 0)
    nrotations = property(lambda self: 0)

    @property
    def nint(self):
        return self.ntrans + self.nbonds + self.nangles + self.ndihedrals

    def copy(self):
        new = self.__class__(self.atoms, ignore_rotation=self.ignore_rotation)
        for name in self._names:
            new.internals[name] = list(self.internals[name])
            new._targets[name] = list(self._targets[name])
            new._active[name] = list(self._active[name])
            new._kind[name] = list(self._kind[name])
        return new

    def _active_list(self, name):
        return [c for c, a in zip(self.internals[name], self._active[name]) if a]

    @property
    def targets(self):
        vec = []
        for name in self._names:
            vec += [t for t, a in zip(self._targets[name], self._active[name]) if a]
        return np.array(vec, dtype=np.float64)

    # ---- numerics (batched per kind) --------------------------------------------------------
    def _gather(self, name):
        coords = self._active_list(name)
        na = _NATOMS[name]
        idx = np.array([c.indices for c in coords], dtype=np.int64).reshape((len(coords), na))
        ncv = np.array([c.ncvecs for c in coords], dtype=np.float64).reshape((len(coords), na - 1, 3))
        pos = self.atoms.positions[idx] if len(coords) else np.zeros((0, na, 3))
        tvec = ncv @ np.asarray(self.atoms.cell, dtype=float)
        return idx, pos, tvec

    def calc(self):
        vals = [np.array([c.calc(self.atoms) for c in self._active_list('translations')])]
        for name in ('bonds', 'angles', 'dihedrals'):
            idx, pos, tvec = self._gather(name)
            vals.append(evaluate_kind(name, pos, tvec)[0])
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
        rows = []
        n3 = self.ndof
        for c in self._active_list('translations'):
            r = np.zeros(n3)
            r[3 * c.indices + c.kwargs['dim']] = 1.0 / len(c.indices)
            rows.append(r)
        J = np.array(rows).reshape((len(rows), n3))
        for name in ('bonds', 'angles', 'dihedrals'):
            idx, pos, tvec = self._gather(name)
            if len(idx) == 0:
                continue
            grad = evaluate_kind(name, pos, tvec)[1]            # (nc, na, 3)
            block = np.zeros((len(idx), n3))
            dofs = (3 * idx[:, :, None] + np.arange(3)[None, None, :]).reshape(len(idx), -1)
            np.add.at(block, (np.arange(len(idx))[:, None], dofs), grad.reshape(len(idx), -1))
            J = np.vstack([J, block])
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
        return _HessianStack(self.ndof, blocks)

    # ---- inequality bookkeeping (internal.py:2788-2823) ----------------------------------------
    def has_inequalities(self):
        return any(k in ('lt', 'gt') for name in self._names for k in self._kind[name])

    def disable_satisfied_inequalities(self):
        for name in self._names:
            for i, (coord, kind, target) in enumerate(zip(self.internals[name], self._kind[name], self._targets[name])):
                if kind == 'lt' and coord.calc(self.atoms) <= target:
                    self._active[name][i] = False
                elif kind == 'gt' and coord.calc(self.atoms) >= target:
                    self._active[name][i] = False
                else:
                    self._active[name][i] = True

    def validate_inequalities(self):
        all_valid = True
        for name in self._names:
            for i, (coord, kind, target) in enumerate(zip(self.internals[name], self._kind[name], self._targets[name])):
                if self._active[name][i]:
                    continue
                val = coord.calc(self.atoms)
                if (kind == 'lt' and val > target) or (kind == 'gt' and val < target):
                    self._active[name][i] = True
                    all_valid = False
        return all_valid

    # ---- adding constraints (internal.py:2861-2955) ----------------------------------------------
    def _add(self, name, new, target, kind, replace_ok):
        try:
            idx = self.internals[name].index(new)
        except ValueError:
            self.internals[name].append(new)
            self._targets[name].append(target)
            self._active[name].append(True)
            self._kind[name].append(kind)
            return
        if replace_ok:
            self._targets[name][idx] = target
            self._kind[name][idx] = kind
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
        raise NotImplementedError('rotation constraints (TRIC) are outside the saddle-search scope '
                                  '(periodic slabs never add them: peswrapper.py:244-253)')

    def _fix_internal(self, cls, name, conv, indices, ncvecs=None, mic=None, target=None,
                      comparator='eq', replace_ok=True):
        new = indices if isinstance(indices, cls) else cls(indices, ncvecs=ncvecs)
        target = new.calc(self.atoms) if target is None else target * conv
        self._add(name, new, target, comparator, replace_ok)

    fix_bond = partialmethod(_fix_internal, Bond, 'bonds', 1.)
    fix_angle = partialmethod(_fix_internal, Angle, 'angles', np.pi / 180.)
    fix_dihedral = partialmethod(_fix_internal, Dihedral, 'dihedrals', np.pi / 180.)

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

    def jacobian(self):
        """Dense Wilson B-matrix dq/dx, (nint, 3N) (internal.py:1780-1902)."""
        B = np.zeros((self.nint, self.ndof))
        row = 0
        for k in self._order:
            pos, tvec, dofs = self._batch(k)
            nc = len(pos)
            if nc:
                g = evaluate_kind(k, pos, tvec, hessian=False)[1].reshape(nc, -1)
                np.add.at(B, (np.arange(row, row + nc)[:, None], dofs), g)
            row += nc
        return B

    def hessian_rdot(self, v):
        """D(v)_i = H_i v as a dense (nint, 3N) matrix (internal.py:2307-2575: one HVP per coordinate)."""
        v = np.asarray(v, dtype=np.float64).ravel()
        D = np.zeros((self.nint, self.ndof))
        row = 0
        for k in self._order:
            pos, tvec, dofs = self._batch(k)
            nc = len(pos)
            if nc:
                tan = v[dofs].reshape(pos.shape)
                hv = evaluate_kind(k, pos, tvec, tangent=tan, hessian=False)[3].reshape(nc, -1)
                np.add.at(D, (np.arange(row, row + nc)[:, None], dofs), hv)
            row += nc
        return D

    def hessian(self):
        """Per-coordinate Hessian blocks with the `ldot` contraction (internal.py:2189-2305)."""
        blocks = []
        for k in self._order:
            pos, tvec, dofs = self._batch(k)
            nc, m = len(pos), 3 * _NATOMS[k]
            H = evaluate_kind(k, pos, tvec)[2].reshape(nc, m, m) if nc else np.zeros((0, m, m))
            blocks.append((dofs, H))
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
