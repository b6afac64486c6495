import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from sella_amd import device as _dev
from tools.opt_ab import model, slab
ctx = _dev.get_context()
for name, make in (('model', lambda: model(ctx, 3072)), ('slab', slab)):
    opt = make()
    opt.run(fmax=0.0, steps=3)
    ctx.sync()
    print('==', name, flush=True)
    sys.stderr.flush()
    t = time.perf_counter(); opt.run(fmax=0.0, steps=8); ctx.sync(); print('%.3f ms per step' % (1e3 * (time.perf_counter() - t) / 8), flush=True)
