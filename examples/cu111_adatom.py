#!/usr/bin/env python3
"""The example script of the reference's README (README.md:10-40) on the MI355X path: adatom hop on
Cu fcc(111), bottom half of the slab frozen with translation constraints.

Differences from the reference script: the two `ase` imports (ASE is not in this image; the builders and
an EMT restatement come from sella_amd.atoms — with ASE installed the original imports work unchanged, the
calculator boundary is untouched).  The trajectory file is ASE's own `.traj` format (sella_amd/trajectory.py),
like the README's `trajectory='test_emt.traj'`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a source checkout

from sella_amd import Constraints, Sella  # noqa: E402
from sella_amd.atoms import EMT, add_adsorbate, fcc111  # noqa: E402

size = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (5, 5, 6)

# Set up your system as an atoms object
slab = fcc111('Cu', size, vacuum=7.5)
add_adsorbate(slab, 'Cu', 2.0, 'bridge')

# Optionally, create and populate a Constraints object.
cons = Constraints(slab)
for atom in slab:
    if atom.position[2] < slab.cell[2, 2] / 2.:
        cons.fix_translation(atom.index)

# Set up your calculator
slab.calc = EMT()

# Set up a Sella Dynamics object
dyn = Sella(
    slab,
    constraints=cons,
    trajectory='test_emt.traj',
)

dyn.run(1e-3, 1000)

from sella_amd.trajectory import Trajectory  # noqa: E402

with Trajectory('test_emt.traj') as traj:
    print('%d images in test_emt.traj, final energy %.6f eV' % (len(traj), traj[-1].get_potential_energy()))
