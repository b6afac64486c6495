#!/usr/bin/env python3
"""Two RCCL ranks — two processes — on the ONE GPU of a single-GPU lease.

Purpose: execute `ncclCommInitRank` with nranks = 2, the TCP unique-id rendezvous and an
all-gather on real device pointers before the driver's 8-GPU scaling run is the first to try; or record precisely
why a one-GPU box cannot (RCCL normally rejects two ranks on one device: "Duplicate GPU detected").  Both ranks open
device 0 (SELLA_HIP_DEVICE=0).  Writes one JSON line per rank to stdout; the parent prints a summary.

    python tools/rccl_two_ranks_one_gpu.py            # parent: launches the two ranks, collects the outcome
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def child():
    sys.path.insert(0, REPO)
    out = dict(rank=int(os.environ['RANK']), world=int(os.environ['WORLD_SIZE']), stage='import')
    try:
        import numpy as np
        from sella_amd.comm import RcclCommunicator
        from sella_amd.device import Context
        out['stage'] = 'context'
        ctx = Context()
        out['stage'] = 'ncclCommInitRank'
        comm = RcclCommunicator(ctx)
        out.update(stage='allgather_host', nranks=comm.nranks)
        x = np.arange(5, dtype=np.float64) + 10 * comm.rank
        g = comm.allgather_host(x)
        assert g.shape == (comm.world, 5) and all((g[r] == np.arange(5) + 10 * r).all() for r in range(comm.world))
        out['stage'] = 'allgather_device'
        M = ctx.zeros(1, 64 * (comm.world + 1))
        p, _ = ctx.device_pointer(M)
        ctx.host_to_device(p, np.full(64, float(comm.rank)))
        comm.allgather_device(ctx, p, p + 8 * 64, 8 * 64, ctx.stream)
        ctx.sync()
        got = ctx.device_to_host(p + 8 * 64, 8 * 64 * comm.world).reshape(comm.world, 64)
        assert (got == np.arange(comm.world)[:, None]).all()
        out['stage'] = 'max_host'
        assert comm.max_host(comm.rank + 0.5) == comm.world - 0.5
        comm.barrier()
        comm.close()
        out.update(stage='done', ok=True, torch_imported='torch' in sys.modules)
    except BaseException as e:                                   # noqa: BLE001 — the outcome IS the result
        out.update(ok=False, error=repr(e)[:600])
    print('RCCL2 ' + json.dumps(out), flush=True)


def main():
    if os.environ.get('SELLA_RCCL2_CHILD') == '1':
        return child()
    env = dict(os.environ, SELLA_RCCL2_CHILD='1', WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='29631',
               SELLA_HIP_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0', NCCL_DEBUG=os.environ.get('NCCL_DEBUG', 'WARN'))
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    results, logs = [], []
    for p in procs:
        try:
            txt = p.communicate(timeout=180)[0]
        except subprocess.TimeoutExpired:
            p.kill()
            txt = (p.communicate()[0] or '') + '\n[timeout after 180 s: killed]'
        logs.append(txt)
        for line in txt.splitlines():
            if line.startswith('RCCL2 '):
                results.append(json.loads(line[6:]))
    for r, txt in enumerate(logs):
        tail = [l for l in txt.splitlines() if not l.startswith('RCCL2 ')][-12:]
        if tail:
            print(f'--- rank {r} output (tail) ---')
            print('\n'.join(tail))
    ok = len(results) == 2 and all(r.get('ok') for r in results)
    print('two RCCL ranks on one GPU: ' + ('OK — ' if ok else 'NOT possible here — ') + json.dumps(results))
    return 0


if __name__ == '__main__':
    sys.exit(main())
