"""Host CPU budget of this process and the BLAS thread pools.

The host layer's NumPy work is glue — vectors and thin panels — but NumPy's BLAS sizes its thread pool
from the number of CPUs it can *see*.  In a container with a CFS quota (the GPU boxes here: 256 visible
CPUs, quota 16) the pool's idle workers spin after every call, the cgroup exhausts its quota and the whole
process, including the thread that feeds the GPU, is throttled for the rest of the scheduling period:
measured 100 ms instead of 38 ms per optimizer step on the 1024-atom EMT slab, with 1.2 ms stalls landing
in pure host code or in stream synchronisation alike.  So the pools are capped at the CPUs the process may
actually use.  `SELLA_HOST_THREADS=<n>` overrides the cap, `SELLA_HOST_THREADS=0` leaves the pools alone.
"""
import math
import os

_applied = None


def effective_cpu_count():
    """CPUs this process can use: scheduler affinity capped by the cgroup CPU quota (v2, then v1)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            q, p = f.read().split()[:2]
        if q != 'max':
            quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f:
                q = float(f.read())
            with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
                p = float(f.read())
            if q > 0 and p > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(math.floor(quota))))
    return max(1, n)


def limit_blas_threads(limit=None):
    """Cap the BLAS / OpenMP pools of this process (never raises them).  Returns the cap applied, or None
    when nothing was done (threadpoolctl missing or disabled by SELLA_HOST_THREADS=0)."""
    global _applied
    env = os.environ.get('SELLA_HOST_THREADS')
    if limit is None and env is not None:
        try:
            limit = int(env)
        except ValueError:
            limit = None
        if limit == 0:
            return None
    if limit is None:
        limit = effective_cpu_count()
    # libraries loaded from now on read their pool size from the environment ...
    for var in ('OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
        os.environ.setdefault(var, str(limit))
    # ... and the ones this package uses are loaded now, so that the cap below reaches them
    import numpy  # noqa: F401
    import scipy.linalg  # noqa: F401
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
    except ImportError:
        return None
    current = [p.get('num_threads', 1) for p in threadpool_info()]
    if current and max(current) > limit:
        threadpool_limits(limits=limit)
    _applied = limit
    return limit
