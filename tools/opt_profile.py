"""cProfile of the bench's optimizer leg (model PES, 3N = n, rs='tr', P-RFO): where a Sella step spends its time."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import hessian_like  # noqa: E402
from sella_amd import device as _dev  # noqa: E402
from sella_amd.atoms import Atoms, QuadraticCubicModel  # noqa: E402
from sella_amd.internal import Constraints  # noqa: E402
from sella_amd.optimize.optimize import Sella  # noqa: E402

if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    ctx = _dev.get_context()
    A = hessian_like(n, seed=0)[0]
    dA = ctx.upload(A)
    rng = np.random.RandomState(100)
    U = rng.normal(size=(8, n))
    U /= np.linalg.norm(U, axis=1)[:, None]
    atoms = Atoms(['X'] * (n // 3), 0.05 * rng.normal(size=(n // 3, 3)), pbc=True)
    atoms.calc = QuadraticCubicModel(lambda x: ctx.symm_mm(dA, x), U, c=0.05, device_matrix=dA)
    opt = Sella(atoms, order=1, eta=1e-4, gamma=0.1, delta0=0.1, rs='tr', logfile=None,
                constraints=Constraints(atoms), proj_trans=False)
    opt.run(fmax=0.0, steps=2)
    ctx.sync()
    t = time.perf_counter()
    opt.run(fmax=0.0, steps=steps)
    ctx.sync()
    print('n = %d: %.3f ms per step (unprofiled)' % (n, 1e3 * (time.perf_counter() - t) / steps))
    pr = cProfile.Profile()
    pr.enable()
    opt.run(fmax=0.0, steps=steps)
    ctx.sync()
    pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(18)
