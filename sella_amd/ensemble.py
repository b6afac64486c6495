"""Ensembles of independent saddle searches sharded over the GPUs of one node (SURVEY.md §8e,
BASELINE.json configs[3]).

The reference has no multi-GPU code: every `Sella` object is independent, so an ensemble shards
with zero coupling.  Replica r runs on rank r mod world (one process and one device context per
GPU); the only exchange is ONE all-gather of the per-replica summaries `[converged, nsteps, energy, fmax,
lambda_min]` and final positions at the end — a few hundred KB in total, latency bound, so any xGMI link
suffices and there is nothing to overlap.  The collective is `ncclAllGather` on a device buffer of this
library's context, bound with ctypes (`sella_amd/comm.py`); PyTorch is not imported on that path.  (In the CPU
tests an initialised `torch.distributed` gloo group takes its place.)  The data path itself has no collective.

    results = run_ensemble(make_replica, 64, fmax=1e-3, steps=200, sella_kwargs=dict(order=1))

`make_replica(i) -> atoms` (or `(atoms, sella_keywords_of_this_member)`) must build replica i (with its
calculator) deterministically from i, so every rank can construct exactly its own members.

Within one GPU a small search (3N = 768) is bound by the host: Python between sub-millisecond kernels leaves the
device idle two thirds of the time, and host threads do not help (the interpreter lock).  `EnsemblePool(P)` therefore
runs the rank's members in P worker PROCESSES, each with its own interpreter, device context and HIP stream on the same
GPU; kernels of different workers overlap on the device.  The members stay independent — the sharding rule of the
ranks, one level further down — and the results are bit-identical to the serial run.

    with EnsemblePool(8) as pool:
        results = run_ensemble(make_replica, 64, ..., pool=pool)       # make_replica must be picklable
"""
import os
import sys

import numpy as np

SUMMARY_FIELDS = ('converged', 'nsteps', 'energy', 'fmax', 'lambda_min')


def rank_and_world():
    from .comm import get_communicator
    if int(os.environ.get('WORLD_SIZE', '1')) == 1 and 'torch' not in sys.modules:
        return 0, 1                      # plain single-process use: no communicator, no device context needed yet
    comm = get_communicator()
    return comm.rank, comm.world


# members whose configuration the library loop covers (sella_amd/search.py) run there; False: always the general driver
USE_LIBRARY_SEARCH = os.environ.get('SELLA_LIBRARY_SEARCH', '1') != '0'


def local_members(n_replicas, rank, world):
    """Round-robin assignment: replica r belongs to rank r mod world."""
    return list(range(rank, n_replicas, world))


def run_one(atoms, fmax, steps, sella_kwargs):
    """One saddle search; returns (summary[5], positions (N, 3)).  `atoms` may be a pair (atoms, keywords): what a
    replica factory returns when a member needs `Sella` keywords of its own (its Constraints object, say)."""
    from .optimize.optimize import Sella
    kw = dict(logfile=None)
    kw.update(sella_kwargs or {})
    if isinstance(atoms, tuple):
        atoms, own = atoms
        kw.update(own)
    if USE_LIBRARY_SEARCH:
        # the whole search inside the library when it is covered (sella_amd/search.py): no interpreter between the
        # force calls, so that members on host threads overlap on the GPU
        from .search import LibrarySearch, SearchLeftLibrary
        if LibrarySearch.applies(atoms, **kw):
            start = np.asarray(atoms.positions, dtype=np.float64).copy()
            ls = LibrarySearch(atoms, **kw)
            try:
                conv = ls.run(fmax=fmax, steps=steps)
                summary = np.array([1.0 if conv else 0.0, float(ls.nsteps), ls.energy, ls.fmax_now, ls.lambda_min])
                return summary, np.asarray(atoms.positions, dtype=np.float64).copy()
            except SearchLeftLibrary:
                atoms.positions = start                      # from the beginning with the general driver
            finally:
                ls.close()
    # the general driver: with the switch off, or for a search the library handed back (from the beginning).  Sella.run
    # would otherwise hand a covered search to the library itself — so the two routes stay two routes.
    opt = Sella(atoms, **kw)
    opt.use_library_loop = False
    conv = opt.run(fmax=fmax, steps=steps)
    pes = opt.pes
    f = pes.get_projected_forces()
    fm = float(np.sqrt((f ** 2).sum(axis=1).max()))
    evals = pes.H.evals
    lam = float(evals[0]) if evals is not None else float('nan')
    summary = np.array([1.0 if conv else 0.0, float(opt.nsteps), float(pes.get_f()), fm, lam])
    return summary, np.asarray(atoms.positions, dtype=np.float64).copy()


def _run_members(members, make_replica, fmax, steps, sella_kwargs, own_context):
    """Worker: runs its members one after the other on a context of its own."""
    from . import device
    out = {}
    ctx = None
    if own_context:
        ctx = device.Context()
        device.use_context(ctx)
    try:
        for i in members:
            out[i] = run_one(make_replica(i), fmax, steps, sella_kwargs)
    finally:
        if own_context:
            device.use_context(None)
            import gc
            gc.collect()
            ctx.close()
    return out


def _pool_worker(conn, initializer, initargs):
    """Main of one worker process: own interpreter, own default device context."""
    try:
        if initializer is not None:
            initializer(*initargs)
        from . import device
        ctx = device.get_context()
        conn.send(('ready', os.getpid()))
    except BaseException as e:                                  # noqa: BLE001 - reported to the parent, which raises
        conn.send(('error', repr(e)))
        return
    factory = None
    while True:
        try:
            msg = conn.recv()
        except EOFError:
            break
        try:
            if msg[0] == 'stop':
                break
            if msg[0] == 'prepare':
                factory = msg[1]
                prep = getattr(factory, 'prepare', None)
                if prep is not None:
                    for i in msg[2]:
                        prep(i)
                warm = getattr(factory, 'warmup', None)
                if warm is not None:
                    warm()
                conn.send(('ok', None))
            elif msg[0] == 'run':
                _, fac, members, fmax, steps, kw = msg
                if fac is not None:
                    factory = fac
                out = {i: run_one(factory(i), fmax, steps, kw) for i in members}
                ctx.sync()
                conn.send(('ok', out))
        except BaseException as e:                              # noqa: BLE001
            import traceback
            conn.send(('error', traceback.format_exc() + repr(e)))


class EnsemblePool:
    """P worker processes on this rank's GPU, started once and reused.

    `initializer(*initargs)` runs first in every worker (the CPU tests use it to select the host emulation).  The
    workers are started with the `spawn` method — the HIP runtime of the parent must not be forked — and inherit the
    environment, so they open the same device as the parent (`LOCAL_RANK` / `SELLA_HIP_DEVICE`); their BLAS pools are
    capped at one thread each (`SELLA_POOL_HOST_THREADS` overrides).

    Measured on one MI355X with 3N = 768 members (tools/ensemble_probe.py, profiles/r02_ensemble_pool.md): 24 searches/s
    in one process, 45 with P = 2, 79 with P = 4, then DOWN again — 59 (P = 6), 41 (P = 8), 26 (P = 12): beyond four
    client processes the device time-slices their queues instead of overlapping them (fewer hardware queues per
    process, `GPU_MAX_HW_QUEUES`, does not help: 41 / 33 / 28 searches/s at P = 4 / 6 / 8 with one queue each).  Use
    P <= 4 per GPU."""

    BEST_PER_GPU = 4

    def __init__(self, processes, initializer=None, initargs=()):
        import multiprocessing as mp
        self.processes = int(processes)
        if self.processes < 1:
            raise ValueError('EnsemblePool needs at least one process')
        mpc = mp.get_context('spawn')
        keep = os.environ.get('SELLA_HOST_THREADS')
        # one BLAS thread per worker unless told otherwise: the members are small, and idle pool threads spinning in
        # P processes at once exhaust a container's CPU quota (utilities/hostcpu.py)
        os.environ['SELLA_HOST_THREADS'] = os.environ.get('SELLA_POOL_HOST_THREADS', '1')
        # Small transfers of several client processes queue up behind each other on the device's few SDMA engines:
        # with four workers the same members take 0.35 s through the copy engines and 0.19 s with the copies issued
        # as shader blits (46 against 84 searches/s, session r03n) — while ONE process is faster with the engines
        # (eigh 43.7 against 49.0 ms, optimizer step 1.01 against 1.35 ms).  So the workers, and only they, run with
        # HSA_ENABLE_SDMA=0 (read by the runtime when it starts; an explicit setting of the caller wins).
        keep_sdma = os.environ.get('HSA_ENABLE_SDMA')
        if keep_sdma is None:
            os.environ['HSA_ENABLE_SDMA'] = os.environ.get('SELLA_POOL_SDMA', '0')
        self._workers = []
        try:
            for _ in range(self.processes):
                parent, child = mpc.Pipe()
                p = mpc.Process(target=_pool_worker, args=(child, initializer, initargs), daemon=True)
                p.start()
                child.close()
                self._workers.append((p, parent))
        finally:
            if keep is None:
                os.environ.pop('SELLA_HOST_THREADS', None)
            else:
                os.environ['SELLA_HOST_THREADS'] = keep
            if keep_sdma is None:
                os.environ.pop('HSA_ENABLE_SDMA', None)
        try:
            self.pids = [self._expect(conn) for _, conn in self._workers]
        except Exception:
            self.close()
            raise

    @staticmethod
    def _expect(conn):
        try:
            tag, val = conn.recv()
        except EOFError:
            raise RuntimeError('ensemble worker died') from None
        if tag == 'error':
            raise RuntimeError('ensemble worker failed: ' + str(val))
        return val

    def deal(self, members):
        return [list(members[p::self.processes]) for p in range(self.processes)]

    def prepare(self, factory, members):
        """Ship the factory to the workers and let each call `factory.prepare(i)` for its members (host-side data
        built ahead of the searches, e.g. outside a timed region) and then `factory.warmup()`, if it has them."""
        for (_, conn), mine in zip(self._workers, self.deal(members)):
            conn.send(('prepare', factory, mine))
        for _, conn in self._workers:
            self._expect(conn)
        self._prepared = True

    def run(self, factory, members, fmax, steps, sella_kwargs):
        """All members, dealt round-robin to the workers; returns {member: (summary, positions)}.  factory=None
        re-uses the one shipped by `prepare`."""
        for (_, conn), mine in zip(self._workers, self.deal(members)):
            conn.send(('run', factory, mine, fmax, steps, sella_kwargs))
        out = {}
        for _, conn in self._workers:
            out.update(self._expect(conn))
        return out

    def close(self):
        for p, conn in self._workers:
            try:
                conn.send(('stop',))
            except (OSError, BrokenPipeError):
                pass
        for p, conn in self._workers:
            p.join(timeout=30)
            if p.is_alive():
                p.terminate()
            conn.close()
        self._workers = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class EnsembleThreads:
    """T host threads on this rank's GPU, started once and reused, each with a device context (HIP stream, scratch,
    pinned rings) of its own that outlives the runs — creating a context costs tens of milliseconds (its first arena
    slab is a `hipMalloc`), which is the whole budget of a small search.

    Worth having since the searches themselves run inside the library (`LibrarySearch`): the interpreter lock is
    released for the duration of a member, and the threads overlap on the device like the launch-and-wait loops of
    tools/lab/wait_lab.hip.  Members are handed out one by one (whoever is free takes the next); results do not depend
    on which thread ran a member."""

    def __init__(self, threads):
        import threading
        from concurrent.futures import ThreadPoolExecutor
        self.threads = int(threads)
        if self.threads < 1:
            raise ValueError('EnsembleThreads needs at least one thread')
        self._ctxs = []
        self._lock = threading.Lock()
        self._ex = ThreadPoolExecutor(max_workers=self.threads, initializer=self._enter)
        self._all(lambda: None)                                  # every thread up, every context created

    def _enter(self):
        from . import device
        ctx = device.Context()
        device.use_context(ctx)
        with self._lock:
            self._ctxs.append(ctx)

    def _all(self, fn):
        """Run fn once on EVERY thread (a barrier keeps a fast thread from taking two)."""
        import threading
        gate = threading.Barrier(self.threads)

        def task():
            gate.wait()
            return fn()
        for f in [self._ex.submit(task) for _ in range(self.threads)]:
            f.result()

    def prepare(self, factory, members=()):
        """`factory.prepare(i)` for the members (host-side data), then `factory.warmup()` on every thread."""
        prep = getattr(factory, 'prepare', None)
        if prep is not None:
            for i in members:
                prep(i)
        warm = getattr(factory, 'warmup', None)
        if warm is not None:
            self._all(warm)

    def run(self, factory, members, fmax, steps, sella_kwargs):
        futs = {i: self._ex.submit(lambda i=i: run_one(factory(i), fmax, steps, sella_kwargs)) for i in members}
        return {i: f.result() for i, f in futs.items()}

    def close(self):
        if self._ex is None:
            return
        from . import device

        def leave():
            import gc
            ctx = device.get_context()
            device.use_context(None)
            gc.collect()
            ctx.close()
        try:
            self._all(leave)
        finally:
            self._ex.shutdown()
            self._ex = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class EnsembleCohort:
    """W members of this rank's GPU advanced in LOCKSTEP by one host thread (`sella_cohort_*`, csrc/cohort.hip): every
    member keeps a device context of its own, but while the cohort runs they share one stream, every kernel of the step
    is launched once for all members that have reached it (the member is the grid's z index) and every wait is one
    stream synchronisation for all of them — the replica dimension of SURVEY.md 8(e).  Host threads with a stream each
    sit at the runtime's launch rate (~63 launches per member step); a cohort divides the launches by its width.

    Members are taken in waves of `width` (<= 16).  The searches run inside the library (`LibrarySearch`); a member the
    library loop does not cover, or that leaves it on the way, is run by the general driver afterwards, from its start
    geometry, exactly as `run_one` does.  Per-member results are bit-identical to `run_one` (independent `Sella` objects,
    sella/optimize/optimize.py:42-81)."""

    MAX_WIDTH = 16

    def __init__(self, width=8, member_threads=False):
        """member_threads: the members' host code between their launches runs on a worker thread each instead of on fibers
        of the issuing thread — parallel host code, merged launches; width + 1 cores spin while the cohort runs."""
        from . import device
        self.width = int(width)
        if not 1 <= self.width <= self.MAX_WIDTH:
            raise ValueError('EnsembleCohort: width must be in [1, %d]' % self.MAX_WIDTH)
        self._ctxs = [device.Context() for _ in range(self.width)]
        if os.environ.get('SELLA_COHORT_HOST_SCALARS', '1') != '0':
            # scalars only the host consumes are written by their kernels straight into the pinned mirror: in a cohort
            # the read-back copy would be one more (batched) launch per wait
            for c in self._ctxs:
                c.set_option('host_scalars', 1)
        from ctypes import byref, c_void_p
        from . import _lib
        arr = (c_void_p * self.width)(*[c._h for c in self._ctxs])
        h = c_void_p()
        _lib.check(_lib.lib().sella_cohort_create(arr, self.width, byref(h)))
        self._h = h
        self.member_threads = bool(member_threads)
        if self.member_threads:
            _lib.check(_lib.lib().sella_cohort_member_threads(h, 1))

    def prepare(self, factory, members=()):
        """`factory.prepare(i)` for the members (host-side data), then `factory.warmup()` on every member context."""
        from . import device
        prep = getattr(factory, 'prepare', None)
        if prep is not None:
            for i in members:
                prep(i)
        warm = getattr(factory, 'warmup', None)
        if warm is not None:
            keep = getattr(device._tls, 'ctx', None)
            try:
                for c in self._ctxs:
                    device.use_context(c)
                    warm()
            finally:
                device.use_context(keep)

    def stats(self):
        from ctypes import c_long
        from . import _lib
        cn = (c_long * 8)()
        _lib.check(_lib.lib().sella_cohort_stats(self._h, cn))
        return dict(zip(('rounds', 'launches_asked', 'launches_issued', 'waits_asked', 'stream_syncs', 'barrier_arrivals',
                         'us_in_members', 'us_issuing'), (int(v) for v in cn[:8])))

    def run(self, factory, members, fmax, steps, sella_kwargs):
        from ctypes import c_int, c_void_p
        from . import _lib, device
        from .search import LibrarySearch
        import time
        members = list(members)
        out = {}
        keep = getattr(device._tls, 'ctx', None)
        self.last_timing = dict(setup=0.0, run=0.0, collect=0.0)                  # seconds: member set-up / lockstep run / results
        try:
            for lo in range(0, len(members), self.width):
                t_a = time.perf_counter()
                wave = members[lo:lo + self.width]
                searches, leftovers = [], []
                for slot, i in enumerate(wave):
                    device.use_context(self._ctxs[slot])
                    atoms = factory(i)
                    kw = dict(logfile=None)
                    kw.update(sella_kwargs or {})
                    if isinstance(atoms, tuple):
                        atoms, own = atoms
                        kw.update(own)
                    if USE_LIBRARY_SEARCH and LibrarySearch.applies(atoms, **kw):
                        start = np.asarray(atoms.positions, dtype=np.float64).copy()
                        searches.append((slot, i, atoms, kw, start, LibrarySearch(atoms, **kw)))
                    else:
                        leftovers.append((slot, i, atoms, kw))
                t_b = time.perf_counter()
                self.last_timing['setup'] += t_b - t_a
                if searches:
                    n = max(s[0] for s in searches) + 1
                    handles = (c_void_p * n)()
                    for slot, _, _, _, _, ls in searches:
                        handles[slot] = ls._h
                    conv, status = (c_int * n)(), (c_int * n)()
                    _lib.check(_lib.lib().sella_cohort_run_searches(self._h, handles, n, float(fmax), int(steps), conv, status))
                    t_c = time.perf_counter()
                    self.last_timing['run'] += t_c - t_b
                    for slot, i, atoms, kw, start, ls in searches:
                        device.use_context(self._ctxs[slot])
                        try:
                            ls._sync()
                            if status[slot] == 0:
                                out[i] = (np.array([1.0 if conv[slot] else 0.0, float(ls.nsteps), ls.energy, ls.fmax_now,
                                                    ls.lambda_min]), np.asarray(atoms.positions, dtype=np.float64).copy())
                            elif status[slot] == -7:                   # SELLA_E_UNSUPPORTED: left the covered configuration
                                atoms.positions = start
                                leftovers.append((slot, i, atoms, kw))
                            else:
                                msg = _lib.lib().sella_cohort_error(self._h, slot)
                                raise _lib.SellaHipError(status[slot], (msg or b'').decode())
                        finally:
                            ls.close()
                    self.last_timing['collect'] += time.perf_counter() - t_c
                for slot, i, atoms, kw in leftovers:                   # the general driver, one after the other
                    from .optimize.optimize import Sella
                    device.use_context(self._ctxs[slot])
                    opt = Sella(atoms, **kw)
                    opt.use_library_loop = False
                    cv = opt.run(fmax=fmax, steps=steps)
                    pes = opt.pes
                    f = pes.get_projected_forces()
                    evals = pes.H.evals
                    out[i] = (np.array([1.0 if cv else 0.0, float(opt.nsteps), float(pes.get_f()),
                                        float(np.sqrt((f ** 2).sum(axis=1).max())),
                                        float(evals[0]) if evals is not None else float('nan')]),
                              np.asarray(atoms.positions, dtype=np.float64).copy())
        finally:
            device.use_context(keep)
        return out

    def close(self):
        if self._h is None:
            return
        from . import _lib
        import gc
        gc.collect()
        _lib.lib().sella_cohort_destroy(self._h)
        self._h = None
        for c in self._ctxs:
            c.close()
        self._ctxs = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class EnsembleCohorts:
    """T issuing threads on this rank's GPU, each advancing an `EnsembleCohort` of `width` members of its own (own
    contexts, own stream): the members' host code between the launches — serial inside one cohort — runs on T cores, and
    T streams overlap on the device.  The rank's members are dealt to the threads in contiguous blocks of `width`, so a
    member's cohort-mates — which never influence its results — are fixed by its index.  `run_ensemble(..., cohort=...)`
    takes either class."""

    def __init__(self, width=8, threads=2, member_threads=False):
        from concurrent.futures import ThreadPoolExecutor
        self.member_threads = bool(member_threads)
        import threading
        self.width, self.threads = int(width), int(threads)
        if self.threads < 1:
            raise ValueError('EnsembleCohorts needs at least one thread')
        self._local = threading.local()
        self._all_cohorts = []
        self._lock = threading.Lock()
        self._ex = ThreadPoolExecutor(max_workers=self.threads, initializer=self._enter)
        self._each(lambda: None)

    def _enter(self):
        co = EnsembleCohort(self.width, member_threads=self.member_threads)
        self._local.cohort = co
        with self._lock:
            self._all_cohorts.append(co)

    def _each(self, fn):
        import threading
        gate = threading.Barrier(self.threads)

        def task():
            gate.wait()
            return fn()
        return [f.result() for f in [self._ex.submit(task) for _ in range(self.threads)]]

    def prepare(self, factory, members=()):
        prep = getattr(factory, 'prepare', None)
        if prep is not None:
            for i in members:
                prep(i)
        if getattr(factory, 'warmup', None) is not None:
            self._each(lambda: self._local.cohort.prepare(_WarmupOnly(factory)))

    def stats(self):
        out = {}
        for co in self._all_cohorts:
            for k, v in co.stats().items():
                out[k] = out.get(k, 0) + v
        return out

    def run(self, factory, members, fmax, steps, sella_kwargs):
        members = list(members)
        blocks = [members[lo:lo + self.width] for lo in range(0, len(members), self.width)]
        futs = [self._ex.submit(lambda b=b: self._local.cohort.run(factory, b, fmax, steps, sella_kwargs)) for b in blocks]
        out = {}
        for f in futs:
            out.update(f.result())
        return out

    def close(self):
        if self._ex is None:
            return
        try:
            self._each(lambda: self._local.cohort.close())
        finally:
            self._ex.shutdown()
            self._ex = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class _WarmupOnly:
    def __init__(self, factory):
        self.warmup = factory.warmup


def run_ensemble(make_replica, n_replicas, fmax=0.05, steps=1000, sella_kwargs=None, threads=1, pool=None,
                 prepared=False, cohort=None):
    """Run `n_replicas` independent searches, sharded over the initialised process group (or all in
    this process when there is none).  Every rank returns the same
    `dict(summary=(n_replicas, 5) array, positions=list of (N_i, 3) arrays, owner=(n_replicas,))`.

    threads: an `EnsembleThreads` (persistent threads and contexts), or a number > 1: the rank's members are dealt to
    that many host threads created for this call, each with its own device
    context (own HIP stream): a small search is host-latency bound (Python between sub-millisecond
    kernels), so several of them keep one GPU busy.  `make_replica(i)` is then called inside the
    worker thread and must build its calculator on `sella_amd.device.get_context()`.

    cohort: an `EnsembleCohort`; the rank's members advance in lockstep through batched launches, `width` at a time.

    pool: an `EnsemblePool`; the rank's members run in its worker processes (`make_replica` is pickled to them,
    or — prepared=True — the factory already shipped by `pool.prepare` is used)."""
    rank, world = rank_and_world()
    mine = local_members(n_replicas, rank, world)
    summaries, positions = {}, {}
    if cohort is not None:
        for i, (sm, ps) in cohort.run(make_replica, mine, fmax, steps, sella_kwargs).items():
            summaries[i], positions[i] = sm, ps
    elif pool is not None:
        done = pool.run(None if prepared else make_replica, mine, fmax, steps, sella_kwargs)
        for i, (sm, ps) in done.items():
            summaries[i], positions[i] = sm, ps
    elif isinstance(threads, EnsembleThreads):
        for i, (sm, ps) in threads.run(make_replica, mine, fmax, steps, sella_kwargs).items():
            summaries[i], positions[i] = sm, ps
    elif threads > 1 and len(mine) > 1:
        from concurrent.futures import ThreadPoolExecutor
        nt = min(threads, len(mine))
        with ThreadPoolExecutor(max_workers=nt) as pool:
            futs = [pool.submit(_run_members, mine[t::nt], make_replica, fmax, steps, sella_kwargs, True)
                    for t in range(nt)]
            for f in futs:
                for i, (sm, ps) in f.result().items():
                    summaries[i], positions[i] = sm, ps
    else:
        for i, (sm, ps) in _run_members(mine, make_replica, fmax, steps, sella_kwargs, False).items():
            summaries[i], positions[i] = sm, ps
    owner = np.arange(n_replicas) % world
    if world == 1:
        return dict(summary=np.array([summaries[i] for i in range(n_replicas)]),
                    positions=[positions[i] for i in range(n_replicas)], owner=owner)

    from .comm import get_communicator
    comm = get_communicator()
    # fixed-size payload per rank: max members per rank x (5 + 1 + 3 * max atoms)
    per_rank = (n_replicas + world - 1) // world
    natoms_local = max([positions[i].shape[0] for i in mine], default=0)
    natoms_max = int(comm.max_host(natoms_local))
    width = len(SUMMARY_FIELDS) + 1 + 3 * natoms_max
    payload = np.zeros((per_rank, width))
    for slot, i in enumerate(mine):
        payload[slot, :5] = summaries[i]
        payload[slot, 5] = positions[i].shape[0]
        payload[slot, 6:6 + positions[i].size] = positions[i].ravel()
    gathered = comm.allgather_host(payload.ravel())      # the one collective of the ensemble (RCCL over xGMI)
    out_s = np.zeros((n_replicas, len(SUMMARY_FIELDS)))
    out_p = [None] * n_replicas
    for r in range(world):
        block = gathered[r].reshape(per_rank, width)
        for slot, i in enumerate(local_members(n_replicas, r, world)):
            out_s[i] = block[slot, :5]
            na = int(block[slot, 5])
            out_p[i] = block[slot, 6:6 + 3 * na].reshape(na, 3).copy()
    return dict(summary=out_s, positions=out_p, owner=owner)
