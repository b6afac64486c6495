"""The bench's ensemble leg (3N = 768 model-PES members, 20 steps each) on HOST THREADS of one process — one device
context (HIP stream) per thread — for T = 1, 2, 4, 8, ...: searches/s.   usage: ensemble_threads.py [members] [T ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import EnsembleMember  # noqa: E402
from sella_amd.ensemble import EnsembleThreads, run_ensemble, run_one  # noqa: E402

if __name__ == '__main__':
    nmem = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    threads = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8]
    fac = EnsembleMember(768)
    for i in range(-1, nmem):
        fac.prepare(i)
    run_one(fac(-1), 0.0, 3, EnsembleMember.SELLA_KW)
    ref = None
    for T in threads:
        with EnsembleThreads(T) as pool:
            pool.prepare(fac)
            run_ensemble(fac, min(nmem, 2 * T), fmax=0.0, steps=3, sella_kwargs=EnsembleMember.SELLA_KW, threads=pool)
            t = time.perf_counter()
            res = run_ensemble(fac, nmem, fmax=0.0, steps=20, sella_kwargs=EnsembleMember.SELLA_KW, threads=pool)
            dt = time.perf_counter() - t
        same = True if ref is None else bool((res['summary'] == ref).all())
        ref = res['summary'] if ref is None else ref
        print('threads %d: %d members in %.3f s = %.1f searches/s (bit-identical to the first run: %s)'
              % (T, nmem, dt, nmem / dt, same), flush=True)
