"""CPU restatement of the geodesic position update of sella/peswrapper.py (`InternalPES`) — TEST INFRASTRUCTURE.

Dense NumPy/SciPy, in the reference's own formulation: economy QR of the dense B-matrix with the SVD
fall-through for a rank-deficient Jacobian (peswrapper.py:674-709), `Binv = R^-1 Q^T` or the truncated-SVD
pseudo-inverse (:711-736), the ODE state `[x, xdot, g]` with `xddot = -Binv (D(xdot) xdot)` and the gradient
transported the same way (:1200-1221), integrated by LSODA over t in [0, 1] with atol = 1e-6 (:840-880).
The coordinate derivatives come from `internals.py` of this package (hyper-dual restatement of internal.py).
**Parity unpinned** against the reference class itself (needs ASE + JAX, not importable in the build
container); the product's sparse / spectral-factor formulation (`sella_amd.peswrapper._BFactor`) is tested
against THIS dense restatement.

Only tests/ may import this module (oracle/README.md).
"""
import numpy as np
from scipy.integrate import LSODA
from scipy.linalg import solve_triangular

from . import internals as _ic

_NATOMS = {'bonds': 2, 'angles': 3, 'dihedrals': 4}
_ORDER = ('bonds', 'angles', 'dihedrals')


class DenseInternals:
    """Index-array description of a coordinate set (same conventions as sella_amd.internal.InternalCoordinates:
    `idx[kind]` (n, natoms) atom indices, `ncv[kind]` (n, natoms-1, 3) integer cell offsets) evaluated densely."""

    def __init__(self, idx, ncv, cell):
        self.idx = {k: np.asarray(idx.get(k, np.zeros((0, _NATOMS[k]))), dtype=np.int64).reshape(-1, _NATOMS[k])
                    for k in _ORDER}
        self.ncv = {k: np.asarray(ncv.get(k, np.zeros((len(self.idx[k]), _NATOMS[k] - 1, 3))), dtype=float)
                    .reshape(len(self.idx[k]), _NATOMS[k] - 1, 3) for k in _ORDER}
        self.cell = np.asarray(cell, dtype=float).reshape(3, 3)

    def _batch(self, kind, pos):
        idx = self.idx[kind]
        dofs = (3 * idx[:, :, None] + np.arange(3)[None, None, :]).reshape(len(idx), -1)
        return pos[idx], self.ncv[kind] @ self.cell, dofs

    def calc(self, pos):                                                        # internal.py:1735-1778
        return np.concatenate([_ic.value_only(k, *self._batch(k, pos)[:2]) for k in _ORDER])

    def jacobian(self, pos):                                                    # internal.py:1780-1902
        nint = sum(len(self.idx[k]) for k in _ORDER)
        B = np.zeros((nint, pos.size))
        row = 0
        for k in _ORDER:
            p, t, dofs = self._batch(k, pos)
            if len(p):
                g = _ic.evaluate_kind(k, p, t)[1].reshape(len(p), -1)
                B[np.arange(row, row + len(p))[:, None], dofs] = g
            row += len(p)
        return B

    def hessian_rdot(self, pos, v):                                             # internal.py:2307-2575
        nint = sum(len(self.idx[k]) for k in _ORDER)
        D = np.zeros((nint, pos.size))
        row = 0
        for k in _ORDER:
            p, t, dofs = self._batch(k, pos)
            if len(p):
                H = _ic.evaluate_kind(k, p, t)[2].reshape(len(p), dofs.shape[1], dofs.shape[1])
                D[np.arange(row, row + len(p))[:, None], dofs] = np.einsum('iab,ib->ia', H, v[dofs])
            row += len(p)
        return D

    def wrap(self, vec):                                                        # internal.py:2577-2587
        nd = len(self.idx['dihedrals'])
        vec = np.array(vec, dtype=float)
        if nd:
            vec[-nd:] = (vec[-nd:] + np.pi) % (2 * np.pi) - np.pi
        return vec


def jacobian_qr(B):
    """(Q, R, Binv or None) — peswrapper.py:674-709."""
    if B.shape[0] >= B.shape[1]:
        Q, R = np.linalg.qr(B, mode='reduced')
    else:
        Q, R = np.linalg.qr(B, mode='reduced')
    rdiag = np.abs(np.diag(R))
    if len(rdiag) > 0 and rdiag.min() < 1e-6 * rdiag.max():
        Ui, Si, VTi = np.linalg.svd(B, full_matrices=False)
        nnred = int(np.sum(Si > 1e-6))
        Q = Ui[:, :nnred]
        R = np.diag(Si[:nnred]) @ VTi[:nnred]
        return Q, R, VTi[:nnred].T @ np.diag(1.0 / Si[:nnred]) @ Ui[:, :nnred].T
    return Q, R, None


def pseudo_inverse(B):
    """peswrapper.py:711-736."""
    Q, R, Binv = jacobian_qr(B)
    if Binv is not None:
        return Binv
    if R.size == 0:
        return np.empty((B.shape[1], 0))
    if R.shape[0] == R.shape[1]:
        return solve_triangular(R, Q.T, check_finite=False)
    return np.linalg.pinv(B)


def geodesic_update(ints, pos, target_dq, g_int=None, exact_geodesic=False, atol=1e-6):
    """Positions after the geodesic step towards q(pos) + target_dq, with the transported quantities:
    (positions (N, 3), dx_initial, dx_final, g_final, number of right-hand sides) — peswrapper.py:840-880."""
    pos = np.asarray(pos, dtype=float)
    nx = pos.size
    dx = ints.wrap(target_dq)
    Binv0 = pseudo_inverse(ints.jacobian(pos))
    if g_int is None:
        g_int = np.zeros_like(dx)

    def rhs(t, y):                                                              # :1200-1221
        x, dxdt, g = y.reshape((3, nx))
        out = np.zeros((3, nx))
        out[0] = dxdt
        p = x.reshape(-1, 3)
        D = ints.hessian_rdot(p, dxdt)
        Binv = pseudo_inverse(ints.jacobian(p)) if exact_geodesic else Binv0
        o = -Binv @ (D @ np.column_stack((dxdt, g)))
        out[1], out[2] = o[:, 0], o[:, 1]
        return out.ravel()

    y0 = np.hstack((pos.ravel(), Binv0 @ dx, Binv0 @ g_int))
    ode = LSODA(rhs, 0.0, y0, t_bound=1.0, atol=atol)
    t0, y = 0.0, y0
    while ode.status == 'running':
        ode.step()
        y, t0 = ode.y, ode.t
        if ode.nfev > 1000:
            raise RuntimeError('Geometry update ODE is taking too long to converge!')
    if ode.status == 'failed':
        raise RuntimeError('Geometry update ODE failed to converge!')
    y = y.reshape((3, nx))
    newpos = y[0].reshape(-1, 3)
    B = ints.jacobian(newpos)
    return newpos, t0 * dx, t0 * (B @ y[1]), B @ y[2], ode.nfev
