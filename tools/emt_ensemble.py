"""configs[3] as named — 256-atom EMT slab members — on host threads (EnsembleThreads) and on worker processes
(EnsemblePool) and in lockstep cohorts (EnsembleCohort): searches/s.
usage: emt_ensemble.py [members] [t<threads> | p<processes> | c<cohort width> | c<width>x<issuing threads> ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import EmtSlabMember  # noqa: E402
from sella_amd.ensemble import EnsembleCohort, EnsembleCohorts, EnsemblePool, EnsembleThreads, run_ensemble  # noqa: E402

if __name__ == '__main__':
    nmem = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    modes = sys.argv[2:] or ['t1', 't4', 't8', 'p4']
    fac = EmtSlabMember()
    ref = None
    for mode in modes:
        mt = mode.endswith('m')                                  # c<W>[x<T>]m: the members' host code on worker threads
        mode = mode.rstrip('m')
        k = int(mode[1:].split('x')[0])
        nthr = int(mode.split('x')[1]) if 'x' in mode else 1
        if mode[0] == 't':
            with EnsembleThreads(k) as pool:
                pool.prepare(fac)
                for i in range(nmem):
                    fac.prepare(i)                      # host-side description outside the clock (bench.py does the same)
                t = time.perf_counter()
                res = run_ensemble(fac, nmem, fmax=0.0, steps=20, sella_kwargs=EmtSlabMember.SELLA_KW, threads=pool)
                dt = time.perf_counter() - t
        elif mode[0] == 'c':
            with (EnsembleCohort(k, member_threads=mt) if nthr == 1 else EnsembleCohorts(k, nthr, member_threads=mt)) as pool:
                pool.prepare(fac)
                for i in range(nmem):
                    fac.prepare(i)
                run_ensemble(lambda i: fac(-1 - i), min(nmem, k * nthr), fmax=0.0, steps=2, sella_kwargs=EmtSlabMember.SELLA_KW, cohort=pool)   # warm-up (members of its own)
                before = pool.stats()
                t = time.perf_counter()
                res = run_ensemble(fac, nmem, fmax=0.0, steps=20, sella_kwargs=EmtSlabMember.SELLA_KW, cohort=pool)
                dt = time.perf_counter() - t
                after = pool.stats()
                print('  cohort:', {k2: after[k2] - before[k2] for k2 in after}, {k2: round(1e3 * v, 2) for k2, v in getattr(pool, 'last_timing', {}).items()}, 'ms', flush=True)
        else:
            with EnsemblePool(k) as pool:
                pool.prepare(fac, [])
                t = time.perf_counter()
                res = run_ensemble(fac, nmem, fmax=0.0, steps=20, sella_kwargs=EmtSlabMember.SELLA_KW, pool=pool, prepared=True)
                dt = time.perf_counter() - t
        same = True if ref is None else bool((res['summary'] == ref).all())
        ref = res['summary'] if ref is None else ref
        print('%s %d: %d members in %.3f s = %.1f searches/s (bit-identical to the first run: %s)'
              % ({'t': 'threads', 'p': 'processes', 'c': '%d thread(s) x %scohort of' % (nthr, 'member-thread ' if mt else '')}[mode[0]], k, nmem, dt, nmem / dt, same), flush=True)
