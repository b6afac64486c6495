"""BASELINE configs[1] as named: 1024-atom Cu(111) EMT slab (one surface atom lifted onto a bridge site), lower
half frozen, Sella order-1 search; optimizer steps/s with the device EMT calculator."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sella_amd import Constraints, Sella  # noqa: E402
from sella_amd.atoms import EMT, fcc111  # noqa: E402


def make_slab(size=(8, 8, 16)):
    slab = fcc111('Cu', size, vacuum=7.5)
    top = np.argmax(slab.positions[:, 2])
    site = slab.info['adsorbate_sites']['bridge']
    # lift one top-layer atom out of its site onto the neighbouring bridge: adatom + vacancy, N unchanged
    slab.positions[top] += np.array([site[0], site[1], 1.9])
    return slab


if __name__ == '__main__':
    slab = make_slab()
    cons = Constraints(slab)
    for atom in slab:
        if atom.position[2] < slab.cell[2, 2] / 2.:
            cons.fix_translation(atom.index)
    slab.calc = EMT()
    dyn = Sella(slab, constraints=cons, logfile='-')
    t0 = time.perf_counter()
    dyn.run(1e-3, 2)
    print('first 2 steps (incl. initial diagonalisation): %.2f s, force calls %d' % (time.perf_counter() - t0, slab.calc.ncalls))
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    dyn.run(1e-3, 10)
    pr.disable()
    dt = time.perf_counter() - t0
    print('n = %d, nfree = %d: %.1f ms per optimizer step' % (3 * len(slab), dyn.pes.get_Ufree().shape[1], 1e3 * dt / 10))
    pstats.Stats(pr).sort_stats('tottime').print_stats(12)
