"""Row-sharded dense operator for block products H·V across the GPUs of one node
(BASELINE.json configs[4]: n = 12288, k = 16).

GPU p holds the row panel H[p·m : (p+1)·m, :] with m = ceil(n / P) (151 MB at n = 12288, P = 8); the block V
(n × k, 1.6 MB) is replicated.  One block product is: the local slice `H_p V` on the matrix cores
(`panel16_mfma_kernel`, the matrix streamed once for 16 vectors), then ONE all-gather of the P slices (196 KB each)
so that every rank holds the full `H V` — the only exchange step of a block iteration; Gram matrices `Vᵀ(HV)`
(k × k) are computed redundantly on every rank, no all-reduce.  The all-gather is `ncclAllGather` on the library's
own device buffers and stream (`sella_amd/comm.py`, ctypes, no PyTorch); slices this small are latency bound, and a
single collective over xGMI's all-to-all links costs about one hop.  `block_davidson` runs the whole block-Davidson
iteration (`sella_davidson_block`) on top of it.  The reference has no counterpart (no collective anywhere in
zadorlab/sella; its Davidson adds one vector per iteration, sella/eigensolvers.py:111-112).
"""
import numpy as np

from .comm import get_communicator
from .device import get_context


class RowShardedOperator:
    """H V for a symmetric (or general square) H whose rows are split over the process group."""

    def __init__(self, rows_local, row0, n):
        """rows_local: this rank's (m_local × n) panel, row0: its first global row."""
        self.ctx = get_context()
        self.comm = get_communicator(self.ctx)
        self.n = int(n)
        self.row0 = int(row0)
        rows_local = np.ascontiguousarray(rows_local, dtype=np.float64)
        assert rows_local.shape[1] == self.n
        self.m_local = rows_local.shape[0]
        self.world, self.rank = self.comm.world, self.comm.rank
        self.m_max = -(-self.n // self.world)            # every rank contributes a slice of this height
        self.dH = self.ctx.upload(rows_local) if self.m_local else self.ctx.zeros(1, self.n)

    @classmethod
    def from_full(cls, H):
        """Convenience for tests / benchmarks: every rank slices the same full matrix."""
        comm = get_communicator()
        n = H.shape[0]
        m = -(-n // comm.world)
        lo, hi = min(comm.rank * m, n), min((comm.rank + 1) * m, n)
        return cls(H[lo:hi], lo, n)

    def local_matmat(self, X):
        """(m_local × k) slice H_p X — no communication."""
        return self.ctx.symm_mm(self.dH, X)

    def matmat(self, X):
        """Full H X (n × k) on every rank (host in / host out)."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        one_d = X.ndim == 1
        X2 = X[:, None] if one_d else X
        k = X2.shape[1]
        local = self.local_matmat(X2) if self.m_local else np.zeros((0, k))
        if self.world == 1:
            out = local
        else:
            send = np.zeros((self.m_max, k))
            send[:self.m_local] = local
            out = self.comm.allgather_host(send.ravel()).reshape(self.world * self.m_max, k)[:self.n]
        return out[:, 0] if one_d else out

    def block_davidson(self, nev, block=16, tol=1e-8, maxiter=500, maxvec=0, V0=None, Pvecs=None, PvecsT=None,
                       pevals=None, diag=None):
        """Lowest `nev` eigenpairs of the sharded operator: every rank runs the same (deterministic) iteration on
        replicated panels, multiplies its own rows, and one all-gather per block assembles H V."""
        comm, ctx = self.comm, self.ctx

        def gather(send, recv, nbytes, stream):
            comm.allgather_device(ctx, send, recv, nbytes, stream)

        return ctx.davidson_block(self.dH, self.n, nev, block=block, tol=tol, maxiter=maxiter, maxvec=maxvec, V0=V0,
                                  Pvecs=Pvecs, PvecsT=PvecsT, pevals=pevals, diag=diag, row0=self.row0,
                                  world=self.world, allgather=gather if self.world > 1 else None) \
            if self.world > 1 or self.m_local == self.n else None
