#!/usr/bin/env python3
"""librccl through ctypes on one GPU (world size 1): communicator, all-gather on the context's buffers, all-reduce.
Multi-rank: python -m torch.distributed.run --nproc-per-node N tools/rccl_smoke.py (torch only launches)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sella_amd.comm import RcclCommunicator  # noqa: E402
from sella_amd.device import Context  # noqa: E402

ctx = Context()
comm = RcclCommunicator(ctx)
x = np.arange(5, dtype=np.float64) + 10 * comm.rank
g = comm.allgather_host(x)
assert g.shape == (comm.world, 5)
for r in range(comm.world):
    np.testing.assert_array_equal(g[r], np.arange(5) + 10 * r)
assert comm.max_host(comm.rank + 0.5) == comm.world - 0.5
M = ctx.zeros(1, 64 * (comm.world + 1))
p, _ = ctx.device_pointer(M)
ctx.host_to_device(p, np.full(64, float(comm.rank)))
comm.allgather_device(ctx, p, p + 8 * 64, 8 * 64, ctx.stream)
ctx.sync()
out = ctx.device_to_host(p + 8 * 64, 8 * 64 * comm.world).reshape(comm.world, 64)
assert np.all(out == np.arange(comm.world)[:, None])
comm.barrier()
print(f'rccl smoke ok: rank {comm.rank} of {comm.world}, nranks {comm.nranks}, torch imported: {"torch" in sys.modules}')
comm.close()
