"""CPU restatement of the internal-coordinate primitives of sella/internal.py — TEST INFRASTRUCTURE.

Value functions exactly as internal.py:58-80 (`_bond_value`, `_angle_value` with its clip,
`_dihedral_value` through arctan2); the reference differentiates them with JAX (`grad` :85-87,
`jacfwd(grad)` :95-97, `jvp(grad)` :106-135).  JAX is not installable in the build container, so this
restatement carries exact first and second derivatives by hyper-dual-number arithmetic vectorised over
the coordinates of one kind, and is itself checked against central finite differences of the plain
value functions (the reference's own test of these derivatives, tests/internal/test_get_internal.py:26-57).
**Parity unpinned** against the reference's JAX output (not importable here).

Only tests/ may import this module (oracle/README.md).
"""
import numpy as np


# ------------------------------------------------------------------------------------------
# hyper-dual numbers: value v (nc,), gradient g (nc, m), Hessian H (nc, m, m)
# ------------------------------------------------------------------------------------------
class HD:
    __slots__ = ('v', 'g', 'H')

    def __init__(self, v, g, H):
        self.v, self.g, self.H = v, g, H

    @staticmethod
    def variables(x):
        """x (nc, m) -> list of m independent variables."""
        nc, m = x.shape
        out = []
        for i in range(m):
            g = np.zeros((nc, m))
            g[:, i] = 1.0
            out.append(HD(x[:, i].copy(), g, np.zeros((nc, m, m))))
        return out

    @staticmethod
    def _lift(o, like):
        if isinstance(o, HD):
            return o
        v = np.broadcast_to(np.asarray(o, dtype=float), like.v.shape)
        return HD(v, np.zeros_like(like.g), np.zeros_like(like.H))

    def __add__(self, o):
        o = HD._lift(o, self)
        return HD(self.v + o.v, self.g + o.g, self.H + o.H)

    __radd__ = __add__

    def __neg__(self):
        return HD(-self.v, -self.g, -self.H)

    def __sub__(self, o):
        return self + (-HD._lift(o, self))

    def __rsub__(self, o):
        return HD._lift(o, self) - self

    def __mul__(self, o):
        o = HD._lift(o, self)
        gg = self.g[:, :, None] * o.g[:, None, :]
        return HD(self.v * o.v, self.v[:, None] * o.g + o.v[:, None] * self.g,
                  self.v[:, None, None] * o.H + o.v[:, None, None] * self.H + gg + gg.transpose(0, 2, 1))

    __rmul__ = __mul__

    def apply(self, f, df, d2f):
        """Elementwise function with first and second derivative values."""
        return HD(f, df[:, None] * self.g,
                  df[:, None, None] * self.H + d2f[:, None, None] * (self.g[:, :, None] * self.g[:, None, :]))

    def recip(self):
        return self.apply(1.0 / self.v, -1.0 / self.v ** 2, 2.0 / self.v ** 3)

    def __truediv__(self, o):
        return self * HD._lift(o, self).recip()

    def sqrt(self):
        s = np.sqrt(self.v)
        return self.apply(s, 0.5 / s, -0.25 / s ** 3)

    def arccos(self):
        c = np.clip(self.v, -1.0, 1.0)
        om = np.maximum(1.0 - c * c, 1e-300)
        return self.apply(np.arccos(c), -1.0 / np.sqrt(om), -c / om ** 1.5)


def hd_arctan2(y, x):
    r2 = x * x + y * y
    U = x / r2
    W = -(y / r2)
    g = U.v[:, None] * y.g + W.v[:, None] * x.g
    H = (U.v[:, None, None] * y.H + W.v[:, None, None] * x.H
         + U.g[:, :, None] * y.g[:, None, :] + W.g[:, :, None] * x.g[:, None, :])
    return HD(np.arctan2(y.v, x.v), g, 0.5 * (H + H.transpose(0, 2, 1)))


def _dot(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def _cross(a, b):
    return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]


def _sub(a, b, t=None):
    out = [a[i] - b[i] for i in range(3)]
    if t is not None:
        out = [out[i] + t[:, i] for i in range(3)]
    return out


def _norm(a):
    return _dot(a, a).sqrt()


# value functions on hyper-dual inputs; p = list of atoms, each a list of 3 HD; t (nc, nvec, 3)
def _bond_hd(p, t):                                           # internal.py:58-60
    return _norm(_sub(p[1], p[0], t[:, 0]))


def _angle_hd(p, t):                                          # internal.py:63-70
    dx1 = [-c for c in _sub(p[1], p[0], t[:, 0])]
    dx2 = _sub(p[2], p[1], t[:, 1])
    return (_dot(dx1, dx2) / (_norm(dx1) * _norm(dx2))).arccos()


def _dihedral_hd(p, t):                                       # internal.py:73-80
    dx1 = _sub(p[1], p[0], t[:, 0])
    dx2 = _sub(p[2], p[1], t[:, 1])
    dx3 = _sub(p[3], p[2], t[:, 2])
    c12, c23 = _cross(dx1, dx2), _cross(dx2, dx3)
    numer = _dot(dx2, _cross(c12, c23))
    denom = _norm(dx2) * _dot(c12, c23)
    return hd_arctan2(numer, denom)


_KINDS = {'bonds': (2, _bond_hd), 'angles': (3, _angle_hd), 'dihedrals': (4, _dihedral_hd)}


def value_only(kind, pos, tvec):
    """Plain NumPy value functions, internal.py:58-80 (for finite-difference checks)."""
    pos = np.asarray(pos, dtype=float)
    t = np.zeros((pos.shape[0], pos.shape[1] - 1, 3)) if tvec is None else np.asarray(tvec, dtype=float)
    if kind == 'bonds':
        return np.linalg.norm(pos[:, 1] - pos[:, 0] + t[:, 0], axis=1)
    if kind == 'angles':
        dx1 = -(pos[:, 1] - pos[:, 0] + t[:, 0])
        dx2 = pos[:, 2] - pos[:, 1] + t[:, 1]
        c = (dx1 * dx2).sum(1) / (np.linalg.norm(dx1, axis=1) * np.linalg.norm(dx2, axis=1))
        return np.arccos(np.clip(c, -1.0, 1.0))
    dx1 = pos[:, 1] - pos[:, 0] + t[:, 0]
    dx2 = pos[:, 2] - pos[:, 1] + t[:, 1]
    dx3 = pos[:, 3] - pos[:, 2] + t[:, 2]
    c12, c23 = np.cross(dx1, dx2), np.cross(dx2, dx3)
    numer = (dx2 * np.cross(c12, c23)).sum(1)
    denom = np.linalg.norm(dx2, axis=1) * (c12 * c23).sum(1)
    return np.arctan2(numer, denom)


def evaluate_kind(kind, pos, tvec):
    """pos (nc, natoms, 3), tvec (nc, natoms-1, 3) -> value (nc,), grad (nc, natoms, 3),
    hess (nc, natoms, 3, natoms, 3)."""
    na, fn = _KINDS[kind]
    nc = pos.shape[0]
    if nc == 0:
        return np.zeros(0), np.zeros((0, na, 3)), np.zeros((0, na, 3, na, 3))
    vars_ = HD.variables(pos.reshape(nc, 3 * na))
    p = [[vars_[3 * a + d] for d in range(3)] for a in range(na)]
    out = fn(p, tvec)
    return out.v, out.g.reshape(nc, na, 3), out.H.reshape(nc, na, 3, na, 3)


