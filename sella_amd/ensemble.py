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

`make_replica(i) -> atoms` must build replica i (with its calculator) deterministically from i, so
every rank can construct exactly its own members.
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


def local_members(n_replicas, rank, world):
    """Round-robin assignment: replica r belongs to rank r mod world."""
    return list(range(rank, n_replicas, world))


def run_one(atoms, fmax, steps, sella_kwargs):
    """One saddle search; returns (summary[5], positions (N, 3))."""
    from .optimize.optimize import Sella
    kw = dict(logfile=None)
    kw.update(sella_kwargs or {})
    opt = Sella(atoms, **kw)
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


def run_ensemble(make_replica, n_replicas, fmax=0.05, steps=1000, sella_kwargs=None, threads=1):
    """Run `n_replicas` independent searches, sharded over the initialised process group (or all in
    this process when there is none).  Every rank returns the same
    `dict(summary=(n_replicas, 5) array, positions=list of (N_i, 3) arrays, owner=(n_replicas,))`.

    threads > 1: the rank's members are dealt to that many host threads, each with its own device
    context (own HIP stream): a small search is host-latency bound (Python between sub-millisecond
    kernels), so several of them keep one GPU busy.  `make_replica(i)` is then called inside the
    worker thread and must build its calculator on `sella_amd.device.get_context()`."""
    rank, world = rank_and_world()
    mine = local_members(n_replicas, rank, world)
    summaries, positions = {}, {}
    if threads > 1 and len(mine) > 1:
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
