"""cProfile of one ensemble member (3N = 768 model PES, 20 optimizer steps incl. the initial diagonalisation)."""
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
from sella_amd.device import Context  # noqa: E402
from sella_amd.ensemble import run_one  # noqa: E402

ne = int(os.environ.get('N', '768'))
ctx = Context()
_dev._default = ctx


def member(i):
    Ai = hessian_like(ne, seed=5000 + i)[0]
    dAi = ctx.upload(Ai)
    rngi = np.random.RandomState(6000 + i)
    Ui = rngi.normal(size=(8, ne))
    Ui /= np.linalg.norm(Ui, axis=1)[:, None]
    at = Atoms(['X'] * (ne // 3), 0.05 * rngi.normal(size=(ne // 3, 3)), pbc=True)
    at.calc = QuadraticCubicModel(lambda x, dAi=dAi: ctx.symm_mm(dAi, x), Ui, c=0.05, device_matrix=dAi)
    return at


kw = dict(order=1, eta=1e-4, gamma=0.1, delta0=0.1, rs='tr', proj_trans=False)
for w in range(3):
    run_one(member(w), 0.0, 20, kw)                  # warm-up (scratch allocation)
at = member(3)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
run_one(at, 0.0, 20, kw)
pr.disable()
print('seconds per member', time.perf_counter() - t0)
pstats.Stats(pr).sort_stats('tottime').print_stats(14)

# the bench leg itself: 8 members through run_ensemble on this context
from sella_amd.ensemble import run_ensemble  # noqa: E402

ats = {i: member(10 + i) for i in range(8)}
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
res = run_ensemble(lambda i: ats[i], 8, fmax=0.0, steps=20, sella_kwargs=kw, threads=1)
pr.disable()
print('run_ensemble: seconds per member', (time.perf_counter() - t0) / 8)
pstats.Stats(pr).sort_stats('tottime').print_stats(10)
