"""Ensembles of independent saddle searches sharded over the GPUs of one node (SURVEY.md §8e,
BASELINE.json configs[3]).

The reference has no multi-GPU code: every `Sella` object is independent, so an ensemble shards
with zero coupling.  Replica r runs on rank r mod world (one process and one device context per
GPU, launched with `python -m torch.distributed.run`); the only exchange is ONE all-gather of the
per-replica summaries `[converged, nsteps, energy, fmax, lambda_min]` and final positions at the
end — a few hundred KB in total, latency bound, so any xGMI link suffices and there is nothing to
overlap.  `torch.distributed` is used for exactly that (backend "nccl" = RCCL on a GPU box, "gloo"
in the CPU tests); the data path itself has no collective.

    results = run_ensemble(make_replica, 64, fmax=1e-3, steps=200, sella_kwargs=dict(order=1))

`make_replica(i) -> atoms` must build replica i (with its calculator) deterministically from i, so
every rank can construct exactly its own members.
"""
import os
import sys

import numpy as np

SUMMARY_FIELDS = ('converged', 'nsteps', 'energy', 'fmax', 'lambda_min')


def _dist():
    # A process group can only have been initialised by code that imported torch already: single-process use
    # neither needs torch nor pays its import (0.9 s, which used to land inside the first ensemble call).
    if 'torch' not in sys.modules:
        return None
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def rank_and_world():
    dist = _dist()
    if dist is None:
        return 0, 1
    return dist.get_rank(), dist.get_world_size()


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

    import torch
    dist = _dist()
    on_gpu = dist.get_backend() == 'nccl'
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0'))) if on_gpu else torch.device('cpu')
    # fixed-size payload per rank: max members per rank x (5 + 1 + 3 * max atoms)
    per_rank = (n_replicas + world - 1) // world
    natoms_local = max([positions[i].shape[0] for i in mine], default=0)
    t = torch.tensor([natoms_local], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    natoms_max = int(t.item())
    width = len(SUMMARY_FIELDS) + 1 + 3 * natoms_max
    payload = torch.zeros((per_rank, width), dtype=torch.float64)
    for slot, i in enumerate(mine):
        payload[slot, :5] = torch.from_numpy(summaries[i])
        payload[slot, 5] = positions[i].shape[0]
        payload[slot, 6:6 + positions[i].size] = torch.from_numpy(positions[i].ravel())
    payload = payload.to(dev)
    gathered = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(gathered, payload)             # the one collective of the ensemble (RCCL over xGMI)
    out_s = np.zeros((n_replicas, len(SUMMARY_FIELDS)))
    out_p = [None] * n_replicas
    for r in range(world):
        block = gathered[r].cpu().numpy()
        for slot, i in enumerate(local_members(n_replicas, r, world)):
            out_s[i] = block[slot, :5]
            na = int(block[slot, 5])
            out_p[i] = block[slot, 6:6 + 3 * na].reshape(na, 3).copy()
    return dict(summary=out_s, positions=out_p, owner=owner)
