"""BASELINE configs[1] as named — 1024-atom Cu(111) EMT slab, lower half pinned, default `Sella` — through the library
loop (`Sella.run` with no log: sella_search_run): ms per optimizer step; under rocprofv3 --kernel-trace the timeline of a
step (tools/opt_timeline_parse.py <dir> 0.5 lr_pre_plan 2)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sella_amd import Constraints, Sella, device as _dev  # noqa: E402
from sella_amd.atoms import EMT  # noqa: E402
from tools.emt_slab_opt import make_slab  # noqa: E402

if __name__ == '__main__':
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    ctx = _dev.get_context()
    for key, val in (kv.split('=') for kv in sys.argv[2:]):
        ctx.set_option(key, int(val))
    slab = make_slab()
    cons = Constraints(slab)
    for atom in slab:
        if atom.position[2] < slab.cell[2, 2] / 2.:
            cons.fix_translation(atom.index)
    slab.calc = EMT()
    dyn = Sella(slab, constraints=cons, logfile=None)
    dyn.run(0.0, 3)
    ctx.sync()
    t = time.perf_counter()
    dyn.run(0.0, steps)
    ctx.sync()
    dt = time.perf_counter() - t
    print('EMT slab, library loop: %.3f ms per step (%d steps, in the library: %s, one-call steps %d)'
          % (1e3 * dt / steps, steps, dyn._lib is not None, dyn.fused_steps))
