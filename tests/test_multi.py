"""N > 1 paths on the CPU: two gloo ranks (torch.distributed.run, 127.0.0.1) drive the same code that
runs one-process-per-GPU over RCCL on the GPU box — the ensemble driver with its single all-gather
and bench.py's replica mode with its barrier / max-over-ranks timing."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, '_mp_worker.py')


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch(nproc, *args, timeout=600):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    if nproc == 1:
        cmd = [sys.executable, WORKER, *args]
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={nproc}',
               '--master-addr', '127.0.0.1', '--master-port', str(free_port()), WORKER, *args]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r


def test_ensemble_two_ranks_matches_single_process(tmp_path):
    n_rep = 5                                   # odd: the ranks hold 3 and 2 members
    two = str(tmp_path / 'two.npz')
    one = str(tmp_path / 'one.npz')
    launch(2, 'ensemble', two, str(n_rep))
    launch(1, 'ensemble-serial', one, str(n_rep))
    a, b = np.load(two), np.load(one)
    np.testing.assert_array_equal(a['owner'], np.arange(n_rep) % 2)
    # identical per-replica results whether or not the ensemble was sharded
    np.testing.assert_array_equal(a['summary'], b['summary'])
    for i in range(n_rep):
        np.testing.assert_array_equal(a[f'pos{i}'], b[f'pos{i}'])
    s = a['summary']
    assert np.all(s[:, 0] == 1.0), s            # every search converged ...
    assert np.all(s[:, 3] < 1e-6)               # ... to a stationary point ...
    assert np.all(s[:, 4] < 0.0)                # ... with a negative lowest Hessian eigenvalue


def test_ensemble_cohorts_two_ranks_match_members_run_alone(tmp_path):
    """The replica dimension under sharding: two gloo ranks, each advancing its members in lockstep
    cohorts — batched launches, merged waits (csrc/cohort.hip) — give, bit for bit, what one process gives that runs
    the members one after the other."""
    n_rep = 5                                   # ranks hold 3 and 2 members: a full cohort + a ragged one, and a full one
    two, one = str(tmp_path / 'two.npz'), str(tmp_path / 'one.npz')
    launch(2, 'ensemble-cohort', two, str(n_rep))
    launch(1, 'ensemble-cohort-serial', one, str(n_rep))
    a, b = np.load(two), np.load(one)
    np.testing.assert_array_equal(a['owner'], np.arange(n_rep) % 2)
    np.testing.assert_array_equal(a['summary'], b['summary'])
    for i in range(n_rep):
        np.testing.assert_array_equal(a[f'pos{i}'], b[f'pos{i}'])
    assert np.all(a['summary'][:, 1] == 3) and len({round(e, 9) for e in a['summary'][:, 2]}) == n_rep


def test_ensemble_host_threads_match_serial(tmp_path):
    """Several host threads, one device context each, drive one device: same per-replica results."""
    thr, one = str(tmp_path / 'thr.npz'), str(tmp_path / 'one.npz')
    launch(1, 'ensemble-threads', thr, '5')
    launch(1, 'ensemble-serial', one, '5')
    a, b = np.load(thr), np.load(one)
    np.testing.assert_array_equal(a['summary'], b['summary'])
    for i in range(5):
        np.testing.assert_array_equal(a[f'pos{i}'], b[f'pos{i}'])


def test_ensemble_process_pool_matches_serial(tmp_path):
    """Worker processes (own interpreter, own device context each) on one device: same per-replica results."""
    pool, one = str(tmp_path / 'pool.npz'), str(tmp_path / 'one.npz')
    launch(1, 'ensemble-pool', pool, '5')
    launch(1, 'ensemble-serial', one, '5')
    a, b = np.load(pool), np.load(one)
    np.testing.assert_array_equal(a['summary'], b['summary'])
    for i in range(5):
        np.testing.assert_array_equal(a[f'pos{i}'], b[f'pos{i}'])


def test_ensemble_pool_reports_worker_failures():
    """A worker that cannot start (here: its initializer raises) surfaces as an exception in the parent, with the
    worker's message, and leaves no process behind."""
    import operator

    from sella_amd.ensemble import EnsemblePool
    with pytest.raises(RuntimeError, match='ZeroDivisionError'):
        EnsemblePool(2, initializer=operator.truediv, initargs=(1, 0))


def test_bench_two_ranks(tmp_path):
    out = str(tmp_path / 'bench.json')
    launch(2, 'bench', out)
    lines = [ln for ln in open(out).read().splitlines() if ln.startswith('{')]
    assert len(lines) == 1                      # rank 0 prints exactly one JSON line
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['warmup'] == 1
    assert d['scaling'] == 'weak' and d['unit'] == 'davidson_iter/s' and d['higher_is_better'] is True
    assert d['value'] > 0 and d['ms_per_step'] > 0
    assert d['cpu_baseline'] is None            # rank 0 at N = 1 only
    assert d['roofline']['launches'] > 0
    blk = d['block_davidson']
    assert blk['n'] == 70 and blk['rows_per_rank'] == 35 and blk['iterations'] >= 1 and blk['block_iter_per_s'] > 0
    assert d['collective']['kind'] == 'torch.distributed' and d['collective']['nranks'] == 2
    ens = d['optimizer']['ensemble']
    assert ens['replicas'] == 4 and ens['per_gpu'] == 2 and ens['optimizer_steps_per_s'] > 0


def test_row_sharded_block_product(tmp_path):
    """configs[4] exchange step: each rank multiplies its row panel, one all-gather assembles H V."""
    out = str(tmp_path / 'panel')
    launch(2, 'panel', out)
    r0, r1 = np.load(out + '.rank0.npz'), np.load(out + '.rank1.npz')
    assert int(r0['m_local']) == 19 and int(r1['m_local']) == 18 and int(r1['row0']) == 19
    for r in (r0, r1):
        np.testing.assert_allclose(r['Y'], r['ref'], atol=1e-11)
        np.testing.assert_allclose(r['y1'], r['ref'][:, 0], atol=1e-11)
    np.testing.assert_array_equal(r0['Y'], r1['Y'])      # every rank holds the same assembled block
    # block Davidson over the sharded rows: converged, equal to LAPACK, identical on both ranks
    for r in (r0, r1):
        assert int(r['nconv']) == 4
        np.testing.assert_allclose(r['lams'], r['exact'], atol=1e-10)
    np.testing.assert_array_equal(r0['lams'], r1['lams'])
    np.testing.assert_array_equal(r0['V'], r1['V'])


def test_tcp_rendezvous_broadcast(tmp_path):
    """The unique-id hand-over of sella_amd.comm (what ncclCommInitRank needs before any collective exists):
    rank 0 -> ranks 1, 2 over TCP on MASTER_ADDR, no torch involved."""
    port = free_port()
    code = ("import os, sys; sys.path.insert(0, %r); from sella_amd.comm import broadcast_from_root; "
            "r = int(os.environ['RANK']); p = broadcast_from_root(bytes(range(128)) if r == 0 else b'', r, 3); "
            "assert p == bytes(range(128)), p; assert 'torch' not in sys.modules; print('ok', r)" % os.path.dirname(HERE))
    procs = []
    for r in (1, 2, 0):                       # the root comes up last: the others must retry
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE='3')
        procs.append(subprocess.Popen([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      text=True))
    for p in procs:
        out = p.communicate(timeout=120)[0]
        assert p.returncode == 0, out
