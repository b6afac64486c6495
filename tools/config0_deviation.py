import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden
from sella_amd import Constraints, Sella
from sella_amd.atoms import EMT, add_adsorbate, fcc111
t = load_golden('g13_config0_trace')
slab = fcc111('Cu', (5, 5, 6), vacuum=7.5)
add_adsorbate(slab, 'Cu', 2.0, 'bridge')
cons = Constraints(slab)
pinned = [a.index for a in slab if a.position[2] < slab.cell[2, 2] / 2.]
for i in pinned:
    cons.fix_translation(i)
slab.calc = EMT()
dyn = Sella(slab, constraints=cons, logfile=None)
for i in range(int(t['nsteps'])):
    x_before = dyn.pes.get_x().copy()
    dyn.step()
    f, delta, rho, neval = t[f'scal{i}']
    s = dyn.pes.get_x() - x_before
    print(i, 'max|s| %.3e  dev s %.3e  dev f %.3e  dev g %.3e  |g| %.3e delta %.4e/%.4e rho %.4f/%.4f neval %d/%d' % (
        np.abs(t[f's{i}']).max(), np.abs(s - t[f's{i}']).max(), abs(dyn.pes.get_f() - f), np.abs(dyn.pes.get_g() - t[f'g{i}']).max(),
        np.abs(t[f'g{i}']).max(), dyn.delta, delta, dyn.rho if dyn.rho is not None else np.nan, rho, dyn.pes.neval, int(neval)))
