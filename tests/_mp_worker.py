"""Worker of tests/test_multi.py: one rank of a torch.distributed (gloo) job on the CPU.
The kernel sources run through the host emulation (tests/hostemu) — same code path as on a GPU box
with RCCL, only the backend and the device library differ."""
import ctypes
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests', 'hostemu'))
sys.path.insert(0, os.path.join(REPO, 'tests'))


def use_emulator():
    import build_emu
    from sella_amd import _lib
    _lib._set_library_for_tests(ctypes.CDLL(build_emu.build()))


def make_replica(i):
    """Replica i: 4-atom model PES with one negative mode, deterministic in i."""
    from sella_amd.atoms import Atoms, QuadraticCubicModel
    rng = np.random.RandomState(1000 + i)
    n = 12
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    lam = np.linspace(0.5, 3.0, n)
    lam[0] = -0.8
    A = (Q * lam) @ Q.T
    U = rng.normal(size=(2, n))
    U /= np.linalg.norm(U, axis=1)[:, None]
    atoms = Atoms(['X'] * (n // 3), 0.05 * rng.normal(size=(n // 3, 3)), pbc=True)
    atoms.calc = QuadraticCubicModel(lambda x: A @ x, U, c=0.02)
    return atoms


def make_library_replica(i):
    """Replica i in a form the library loop covers (structured Hessian from 3N = 96, calculator with a library form): what
    the lockstep cohorts advance.  Deterministic in i; built on the calling thread's device context."""
    from conftest_shim import hessian_like
    from sella_amd import device, linalg
    from sella_amd.atoms import Atoms, QuadraticCubicModel
    from sella_amd.internal import Constraints
    linalg.LR_MIN_DIM = 96
    n = 120
    ctx = device.get_context()
    A = hessian_like(n, 41 + 3 * i, nneg=1)[0]
    dA = ctx.upload(A)
    rng = np.random.RandomState(42 + 3 * i)
    U = rng.normal(size=(8, n))
    U /= np.linalg.norm(U, axis=1)[:, None]
    at = Atoms(['X'] * (n // 3), 0.05 * rng.normal(size=(n // 3, 3)), pbc=True)
    at.calc = QuadraticCubicModel(lambda x: ctx.symm_mm(dA, x), U, c=0.05, device_matrix=dA)
    return at, dict(constraints=Constraints(at))


COHORT_KW = dict(order=1, eta=1e-4, gamma=0.1, delta0=0.1, proj_trans=False, rs='tr', nsteps_per_diag=3)


def ensemble_cohort(out_path, n_replicas, sharded):
    """configs[3] at toy size with the replica dimension in the kernels: every rank advances its members in lockstep
    cohorts (width 2), one all-gather at the end; `sharded` False: one process, members one after the other (run_one)."""
    from sella_amd import linalg
    from sella_amd.ensemble import EnsembleCohort, run_ensemble
    linalg.LR_MIN_DIM = 96
    if sharded:
        import torch.distributed as dist
        dist.init_process_group(backend='gloo')
        with EnsembleCohort(2) as cohort:
            res = run_ensemble(make_library_replica, n_replicas, fmax=0.0, steps=3, sella_kwargs=COHORT_KW, cohort=cohort)
            st = cohort.stats()
        assert st['launches_issued'] < st['launches_asked'] or len(res['summary']) < 2 * dist.get_world_size()
        rank0 = dist.get_rank() == 0
    else:
        res = run_ensemble(make_library_replica, n_replicas, fmax=0.0, steps=3, sella_kwargs=COHORT_KW)
        rank0 = True
    if rank0:
        np.savez(out_path, summary=res['summary'], owner=res['owner'], **{f'pos{i}': p for i, p in enumerate(res['positions'])})
    if sharded:
        dist.destroy_process_group()


def ensemble(out_path, n_replicas):
    import torch.distributed as dist
    from sella_amd.ensemble import run_ensemble
    from sella_amd.internal import Constraints
    dist.init_process_group(backend='gloo')
    res = run_ensemble(make_replica, n_replicas, fmax=1e-6, steps=60,
                       sella_kwargs=dict(order=1, eta=1e-5, gamma=0.0, rs='tr', proj_trans=False))
    if dist.get_rank() == 0:
        np.savez(out_path, summary=res['summary'], owner=res['owner'],
                 **{f'pos{i}': p for i, p in enumerate(res['positions'])})
    dist.destroy_process_group()


def bench(out_path):
    import io
    from contextlib import redirect_stdout
    import bench as bench_mod
    os.environ['SELLA_BENCH_COMM'] = 'gloo'
    sys.argv = ['bench.py', '--gpus', os.environ['WORLD_SIZE'], '--steps', '2', '--warmup', '1', '--n', '36', '--option', 'eigh_tail_lds=0',
                '--maxiter', '8', '--seeds', '2', '--converged-n', '0', '--emt-steps', '0', '--no-cpu-baseline', '--opt-steps', '2', '--ensemble-per-gpu', '2', '--ensemble-n', '12', '--ensemble-steps', '2', '--block-n', '70', '--block-iters', '3', '--block-eigh', '0']
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench_mod.main()
    if os.environ.get('RANK', '0') == '0':
        with open(out_path, 'w') as f:
            f.write(buf.getvalue())


def panel(out_path):
    import torch.distributed as dist
    from sella_amd.parallel import RowShardedOperator
    dist.init_process_group(backend='gloo')
    rng = np.random.RandomState(77)
    n, k = 37, 16                                # ragged: 19 + 18 rows over two ranks
    H = rng.normal(size=(n, n))
    H = H + H.T
    X = rng.normal(size=(n, k))
    op = RowShardedOperator.from_full(H)
    Y = op.matmat(X)
    y1 = op.matmat(X[:, 0])
    # block Davidson on the sharded operator (configs[4] at toy size): one all-gather per block iteration
    from conftest_shim import hessian_like
    A, P, g = hessian_like(n, 5, nneg=2)
    opA = RowShardedOperator.from_full(A)
    out = opA.block_davidson(4, block=4, tol=1e-9, maxiter=300, diag=np.diag(A).copy())
    np.savez(out_path + f'.rank{dist.get_rank()}.npz', Y=Y, ref=H @ X, y1=y1, m_local=op.m_local, row0=op.row0,
             lams=out['lams'], V=out['V'], nconv=out['nconv'], exact=np.linalg.eigvalsh(A)[:4])
    dist.destroy_process_group()


if __name__ == '__main__':
    use_emulator()
    if sys.argv[1] == 'ensemble':
        ensemble(sys.argv[2], int(sys.argv[3]))
    elif sys.argv[1] in ('ensemble-cohort', 'ensemble-cohort-serial'):
        ensemble_cohort(sys.argv[2], int(sys.argv[3]), sys.argv[1] == 'ensemble-cohort')
    elif sys.argv[1] == 'bench':
        bench(sys.argv[2])
    elif sys.argv[1] == 'panel':
        panel(sys.argv[2])
    elif sys.argv[1] in ('ensemble-serial', 'ensemble-threads'):
        from sella_amd.ensemble import run_ensemble
        res = run_ensemble(make_replica, int(sys.argv[3]), fmax=1e-6, steps=60,
                           sella_kwargs=dict(order=1, eta=1e-5, gamma=0.0, rs='tr', proj_trans=False),
                           threads=3 if sys.argv[1] == 'ensemble-threads' else 1)
        np.savez(sys.argv[2], summary=res['summary'], owner=res['owner'],
                 **{f'pos{i}': p for i, p in enumerate(res['positions'])})
    elif sys.argv[1] == 'ensemble-pool':
        from sella_amd.ensemble import EnsemblePool, run_ensemble
        with EnsemblePool(2, initializer=use_emulator) as pool:
            assert len(set(pool.pids)) == 2 and os.getpid() not in pool.pids
            res = run_ensemble(make_replica, int(sys.argv[3]), fmax=1e-6, steps=60, pool=pool,
                               sella_kwargs=dict(order=1, eta=1e-5, gamma=0.0, rs='tr', proj_trans=False))
        np.savez(sys.argv[2], summary=res['summary'], owner=res['owner'],
                 **{f'pos{i}': p for i, p in enumerate(res['positions'])})
    else:
        raise SystemExit('unknown mode')
    json.dump({'ok': True}, sys.stderr)
