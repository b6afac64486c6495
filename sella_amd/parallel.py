"""Row-sharded dense operator for block products H·V across the GPUs of one node
(BASELINE.json configs[4]: n = 12288, k = 16).

GPU p holds the row panel H[p·n/P : (p+1)·n/P, :] (151 MB at n = 12288, P = 8); the block V (n × k,
1.6 MB) is replicated.  One block product is: the local slice `H_p V` on the matrix cores
(`panel16_mfma_kernel`, the matrix streamed once for 16 vectors), then ONE all-gather of the P slices
(196 KB each) so that every rank holds the full `H V` — the only exchange step of a block iteration;
Gram matrices `Vᵀ(HV)` (k × k) are computed redundantly on every rank, no all-reduce.  With RCCL the
all-gather of such small slices is latency bound; it is issued as a single collective (not a ring of
point-to-point copies), which on xGMI's all-to-all links costs about one hop.
The reference has no counterpart (no collective anywhere in zadorlab/sella).
"""
import os

import numpy as np

from .device import get_context


def _dist():
    try:
        import torch.distributed as dist
    except ImportError:
        return None
    return dist if dist.is_available() and dist.is_initialized() else None


class RowShardedOperator:
    """H V for a symmetric (or general square) H whose rows are split over the process group."""

    def __init__(self, rows_local, row0, n):
        """rows_local: this rank's (m_local × n) panel, row0: its first global row."""
        self.ctx = get_context()
        self.n = int(n)
        self.row0 = int(row0)
        rows_local = np.ascontiguousarray(rows_local, dtype=np.float64)
        assert rows_local.shape[1] == self.n
        self.m_local = rows_local.shape[0]
        self.dH = self.ctx.upload(rows_local)
        dist = _dist()
        self.world = dist.get_world_size() if dist else 1
        self.rank = dist.get_rank() if dist else 0
        self.m_max = -(-self.n // self.world)            # every rank contributes a slice of this height

    @classmethod
    def from_full(cls, H):
        """Convenience for tests / benchmarks: every rank slices the same full matrix."""
        dist = _dist()
        world = dist.get_world_size() if dist else 1
        rank = dist.get_rank() if dist else 0
        n = H.shape[0]
        m = -(-n // world)
        lo, hi = min(rank * m, n), min((rank + 1) * m, n)
        return cls(H[lo:hi], lo, n)

    def local_matmat(self, X):
        """(m_local × k) slice H_p X — no communication."""
        return self.ctx.symm_mm(self.dH, X)

    def matmat(self, X):
        """Full H X (n × k) on every rank."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        one_d = X.ndim == 1
        X2 = X[:, None] if one_d else X
        local = self.local_matmat(X2) if self.m_local else np.zeros((0, X2.shape[1]))
        dist = _dist()
        if dist is None or self.world == 1:
            out = local
        else:
            import torch
            on_gpu = dist.get_backend() == 'nccl'
            dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0'))) if on_gpu else torch.device('cpu')
            k = X2.shape[1]
            send = torch.zeros((self.m_max, k), dtype=torch.float64)
            send[:self.m_local] = torch.from_numpy(local)
            send = send.to(dev)
            recv = torch.empty((self.world * self.m_max, k), dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(recv, send)          # the one exchange step of a block iteration
            out = recv.cpu().numpy()[:self.n]
        return out[:, 0] if one_d else out
