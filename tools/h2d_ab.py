"""A/B of the upload path of the optimizer step on one box: payloads of at least h2d_kernel_min bytes by a kernel that reads
the pinned ring (default 16384) against the runtime's copy for everything (0): model PES at 3N = 3072 and 768, the 1024-atom
EMT slab; same geometries bit for bit."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sella_amd import device as _dev  # noqa: E402
from tools.opt_ab import model, slab  # noqa: E402

if __name__ == '__main__':
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    ctx = _dev.get_context()
    for name, make in (('model PES 3N=3072', lambda: model(ctx, 3072)), ('model PES 3N=768', lambda: model(ctx, 768)), ('EMT slab 1024 atoms', slab)):
        for kmin in (16384, 0, 16384, 0):
            ctx.set_option('h2d_kernel_min', kmin)
            opt = make()
            opt.run(fmax=0.0, steps=3)
            ctx.sync()
            t = time.perf_counter()
            opt.run(fmax=0.0, steps=steps)
            ctx.sync()
            dt = time.perf_counter() - t
            print('%-22s h2d_kernel_min %5d: %.3f ms per step, x[0..2] %s' % (name, kmin, 1e3 * dt / steps,
                  np.array2string(opt.atoms.positions.ravel()[:3], precision=12)), flush=True)
    ctx.set_option('h2d_kernel_min', 16384)
