"""TEST INFRASTRUCTURE.  Trajectory of BASELINE configs[0] — the reference's README example (README.md:18-25:
Cu fcc111(5,5,6) + adatom at the bridge site, every atom of the lower half held by a translation constraint, default
Sella settings, EMT) — computed by the dense CPU oracle (oracle/sella_oracle/pes.py driving oracle/sella_oracle/emt.py)
and written to tests/golden/g13_config0_trace.npz: per step the step vector, energy, gradient, trust radius, rho and
the number of force calls.  About a minute of NumPy; the GPU test (tests/test_configs_gpu.py::test_config0_*) replays
the product against it step by step.  Both restatements are unpinned against ASE (absent here).

    python oracle/make_config0_trace.py [nsteps]
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle.sella_oracle.emt import EMTOracle                      # noqa: E402
from oracle.sella_oracle.pes import OracleSella, TranslationConstraints   # noqa: E402
from sella_amd.atoms import add_adsorbate, fcc111                  # noqa: E402  (geometry builder only: no device code)


def readme_slab():
    slab = fcc111('Cu', (5, 5, 6), vacuum=7.5)
    add_adsorbate(slab, 'Cu', 2.0, 'bridge')
    pinned = [a.index for a in slab if a.position[2] < slab.cell[2, 2] / 2.]
    return slab, pinned


def main():
    nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    slab, pinned = readme_slab()
    cons = TranslationConstraints(slab)
    for i in pinned:
        cons.fix_translation(int(i))
    slab.calc = EMTOracle()
    x_start = slab.positions.copy()
    ora = OracleSella(slab, cons, order=1, rs='ras')
    out = dict(x_start=x_start, pinned=np.array(pinned), delta0=ora.delta)
    for i in range(nsteps):
        ora.step()
        t = ora.trace[-1]
        out[f's{i}'] = t['s']
        out[f'g{i}'] = t['g']
        out[f'scal{i}'] = np.array([t['f'], t['delta'], np.nan if t['rho'] is None else t['rho'], ora.pes.neval])
        print(i, t['f'], np.linalg.norm(t['s']), t['delta'], t['rho'], ora.pes.neval, flush=True)
    out['nsteps'] = nsteps
    np.savez_compressed(os.path.join(REPO, 'tests', 'golden', 'g13_config0_trace.npz'), **out)


if __name__ == '__main__':
    main()
