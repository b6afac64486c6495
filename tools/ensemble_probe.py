"""Per-member wall time of the bench's ensemble leg (serial), then the same members through EnsemblePool(P)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import EnsembleMember  # noqa: E402
from sella_amd.ensemble import EnsemblePool, run_ensemble, run_one  # noqa: E402

if __name__ == '__main__':
    nmem = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    print('host cpus: os.cpu_count() = %s, usable = %d' % (os.cpu_count(), len(os.sched_getaffinity(0))), flush=True)
    procs = [int(a) for a in sys.argv[2:]] or [4]
    fac = EnsembleMember(768)
    for i in range(nmem):
        fac.prepare(i)
    fac.warmup()
    for i in range(nmem):
        t = time.perf_counter()
        sm, _ = run_one(fac(i), 0.0, 20, EnsembleMember.SELLA_KW)
        print('member %2d: %.3f s, steps %d, lambda_min %.4f' % (i, time.perf_counter() - t, sm[1], sm[4]), flush=True)
    for P in procs:
        with EnsemblePool(P) as pool:
            pool.prepare(fac, list(range(nmem)))
            t = time.perf_counter()
            run_ensemble(fac, nmem, fmax=0.0, steps=20, sella_kwargs=EnsembleMember.SELLA_KW, pool=pool, prepared=True)
            dt = time.perf_counter() - t
        print('pool %d: %d members in %.3f s = %.1f searches/s' % (P, nmem, dt, nmem / dt), flush=True)
