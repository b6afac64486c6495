"""A/B of the one-call optimizer step: model PES at 3N = n ('tr', P-RFO) and the 1024-atom EMT slab ('ras', pins -> view),
with the fused launch chain of csrc/lrstep.hip (lr_chain 1) and with the chain of round 3 (lr_chain 0)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import hessian_like  # noqa: E402
from sella_amd import Constraints, Sella, device as _dev  # noqa: E402
from sella_amd.atoms import EMT, Atoms, QuadraticCubicModel  # noqa: E402
from tools.emt_slab_opt import make_slab  # noqa: E402


def model(ctx, n):
    A = hessian_like(n, seed=0)[0]
    dA = ctx.upload(A)
    rng = np.random.RandomState(100)
    U = rng.normal(size=(8, n))
    U /= np.linalg.norm(U, axis=1)[:, None]
    atoms = Atoms(['X'] * (n // 3), 0.05 * rng.normal(size=(n // 3, 3)), pbc=True)
    atoms.calc = QuadraticCubicModel(lambda x: ctx.symm_mm(dA, x), U, c=0.05, device_matrix=dA)
    return Sella(atoms, order=1, eta=1e-4, gamma=0.1, delta0=0.1, rs='tr', logfile=None, constraints=Constraints(atoms),
                 proj_trans=False)


def slab():
    s = make_slab()
    cons = Constraints(s)
    for atom in s:
        if atom.position[2] < s.cell[2, 2] / 2.:
            cons.fix_translation(atom.index)
    s.calc = EMT()
    return Sella(s, constraints=cons, logfile=None)


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    ctx = _dev.get_context()
    for name, make in (('model PES 3N=%d' % n, lambda: model(ctx, n)), ('EMT slab 1024 atoms', slab)):
        for chain, pipe in ((1, 1), (1, 0), (0, 0), (1, 1), (1, 0), (0, 0)):
            ctx.set_option('lr_chain', chain)
            ctx.set_option('lr_pipe', pipe)
            opt = make()
            opt.run(fmax=0.0, steps=3)
            ctx.sync()
            t = time.perf_counter()
            opt.run(fmax=0.0, steps=steps)
            ctx.sync()
            dt = time.perf_counter() - t
            print('%-22s lr_chain %d lr_pipe %d: %.3f ms per step, x[0..2] %s' % (name, chain, pipe, 1e3 * dt / steps,
                  np.array2string(opt.atoms.positions.ravel()[:3], precision=12)), flush=True)
