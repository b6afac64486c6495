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


def emt_slab(size, lift=1.9, seed=None, jitter=0.0, calculator=None):
    """Cu(111) slab with one top-layer atom lifted onto the neighbouring bridge site (adatom + vacancy), lower half
    frozen atom by atom — the README pattern (README.md:18-25).  Returns (atoms, Constraints, pinned atom indices)."""
    from sella_amd import Constraints
    from sella_amd.atoms import EMT, fcc111
    slab = fcc111('Cu', size, vacuum=7.5)
    if seed is not None:
        slab.positions += jitter * np.random.RandomState(seed).normal(size=slab.positions.shape)
    top = int(np.argmax(slab.positions[:, 2]))
    site = slab.info['adsorbate_sites']['bridge']
    slab.positions[top] += np.array([site[0], site[1], lift])
    cons = Constraints(slab)
    pinned = [a.index for a in slab if a.position[2] < slab.cell[2, 2] / 2.]
    for i in pinned:
        cons.fix_translation(i)
    slab.calc = EMT() if calculator is None else calculator
    return slab, cons, np.array(pinned)


class EmtMember:
    """Picklable factory of configs[3] members as BASELINE names them: member i = 256-atom Cu(111) EMT slab
    (8 x 8 x 4) with a lifted surface atom, lower half pinned, thermal jitter from seed i; returns
    (atoms, the member's own `Sella` keywords)."""

    def __init__(self, size=(8, 8, 4)):
        self.size = size

    def __call__(self, i):
        slab, cons, _ = emt_slab(self.size, seed=100 + i, jitter=0.02)
        return slab, dict(constraints=cons)
