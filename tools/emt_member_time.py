"""One configs[3] member as named (256-atom Cu(111) EMT slab, lower half pinned, default keywords) through the library
loop: seconds per search, force calls, where the time goes (SELLA_DEBUG_TIMING=1 for the stage timings)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import EmtSlabMember  # noqa: E402
from sella_amd.device import get_context  # noqa: E402
from sella_amd.search import LibrarySearch  # noqa: E402

fac = EmtSlabMember()
if os.environ.get('SELLA_GS_SMALL') is not None:
    get_context().set_option('gs_small', int(os.environ['SELLA_GS_SMALL']))
fac.warmup()
for i in range(3):
    atoms, own = fac(i)
    kw = dict(EmtSlabMember.SELLA_KW, **own)
    t0 = time.perf_counter()
    ls = LibrarySearch(atoms, **kw)
    t1 = time.perf_counter()
    ls.run(0.0, 20)
    get_context().sync()
    t2 = time.perf_counter()
    print('member %d: set-up %.1f ms, 20 steps %.1f ms, force calls %d, one-call steps %d, rank %d / view %d, fmax %.3g'
          % (i, 1e3 * (t1 - t0), 1e3 * (t2 - t1), ls.neval, ls.one_call_steps, ls.rank, ls.rank_view, ls.fmax_now), flush=True)
    ls.close()
