"""`hessian_like` and a picklable ensemble-member factory for worker processes that must not import pytest's conftest
machinery."""
import numpy as np


def hessian_like(n, seed, eps=5e-3, nneg=1):
    rng = np.random.RandomState(seed)
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    lam = np.exp(rng.uniform(np.log(0.05), np.log(50.0), n))
    lam[:nneg] = -np.linspace(1.0, 0.5, nneg)
    A = (Q * lam) @ Q.T
    A = 0.5 * (A + A.T)
    N = rng.normal(size=(n, n))
    P = A + eps * 0.5 * (N + N.T)
    g = rng.normal(size=n)
    return A, P, g


class EnsembleFactory:
    """Member i of the configs[3] ensemble tests from host data shipped with the factory (so that every process
    builds bit-identical members): model PES f(x) = x^T A x / 2 + c / 3 sum (u_j . x)^3 on the calling process's
    device context."""

    def __init__(self, ne, host):
        self.ne, self.host = ne, host

    def __call__(self, i):
        from sella_amd import device
        from sella_amd.atoms import Atoms, QuadraticCubicModel
        A, U, x0 = self.host[i]
        ctx = device.get_context()
        dA = ctx.upload(A)
        at = Atoms(['X'] * (self.ne // 3), x0.copy(), pbc=True)
        at.calc = QuadraticCubicModel(lambda x, ctx=ctx, dA=dA: ctx.symm_mm(dA, x), U, c=0.05)
        return at
