"""A/B of the per-atom root search on the 1024-atom EMT slab (default `Sella`: 'ras', P-RFO, pins -> view): the whole
search in one device round trip (rs_dev_root 1) against the round-trip search (0), in the library loop."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sella_amd import Constraints, Sella, device as _dev  # noqa: E402
from sella_amd.atoms import EMT  # noqa: E402
from tools.emt_slab_opt import make_slab  # noqa: E402


def slab():
    s = make_slab()
    cons = Constraints(s)
    for atom in s:
        if atom.position[2] < s.cell[2, 2] / 2.:
            cons.fix_translation(atom.index)
    s.calc = EMT()
    return Sella(s, constraints=cons, logfile=None)


if __name__ == '__main__':
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    modes = [int(a) for a in sys.argv[2:]] or [1, 0, 1, 0]
    ctx = _dev.get_context()
    for dev in modes:
        ctx.set_option('rs_dev_root', dev)
        opt = slab()
        opt.run(fmax=0.0, steps=3)
        ctx.sync()
        t = time.perf_counter()
        opt.run(fmax=0.0, steps=steps)
        ctx.sync()
        dt = time.perf_counter() - t
        print('EMT slab 1024 atoms, rs_dev_root %d: %.3f ms per step, x[0..2] %s' % (dev, 1e3 * dt / steps,
              np.array2string(opt.atoms.positions.ravel()[-3:], precision=12)), flush=True)
