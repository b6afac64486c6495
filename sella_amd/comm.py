"""Process-group plumbing for the two sharded configurations — without PyTorch on the GPU path.

One process per GPU.  The reference has no collective anywhere; what is needed here is tiny:

* configs[3] (ensemble): ONE all-gather of the per-replica summaries at the end;
* configs[4] (row-sharded block product): ONE all-gather of the 16 x n/P slices per block iteration, on the
  library's own device buffers and stream.

`RcclCommunicator` binds librccl directly with ctypes (`ncclCommInitRank`, `ncclAllGather`, `ncclAllReduce`); the
128-byte unique id travels over a small TCP rendezvous derived from MASTER_ADDR / MASTER_PORT (the variables
`python -m torch.distributed.run` exports), so no torch import is needed.  `GlooCommunicator` wraps an already
initialised `torch.distributed` group and exists for the CPU tests only (world_size-2 gloo jobs on the host
emulation).  `get_communicator()` picks: an initialised torch group if torch was imported by the caller (tests),
else RCCL when WORLD_SIZE > 1, else the single-process stand-in.
"""
import ctypes
import hashlib
import os
import socket
import struct
import sys
import time

import numpy as np


class SingleProcess:
    rank, world, kind = 0, 1, 'single'

    def barrier(self):
        pass

    def allgather_host(self, arr):
        return np.asarray(arr, dtype=np.float64)[None].copy()

    def allgather_device(self, ctx, send, recv, nbytes, stream):
        ctx.copy_device(recv, send, nbytes)

    def max_host(self, value):
        return float(value)

    def close(self):
        pass


# ---- TCP rendezvous (unique-id broadcast, nothing else) -------------------------------------------------------------
def _token(world):
    key = '|'.join([os.environ.get('MASTER_ADDR', '127.0.0.1'), os.environ.get('MASTER_PORT', '29500'), str(world),
                    os.environ.get('TORCHELASTIC_RUN_ID', ''), os.environ.get('SELLA_COMM_SALT', '')])
    return hashlib.sha256(key.encode()).digest()[:16]


def _candidate_ports():
    base = int(os.environ.get('MASTER_PORT', '29500'))
    start = 20000 + (base * 7 + 13) % 20000
    return [start + 31 * k for k in range(8)]


def _recv_exact(sock, n):
    buf = b''
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError('rendezvous peer closed the connection')
        buf += chunk
    return buf


def broadcast_from_root(payload, rank, world, timeout=120.0):
    """Rank 0 hands `payload` (bytes) to every other rank over TCP on MASTER_ADDR; returns the payload."""
    if world == 1:
        return payload
    addr = os.environ.get('MASTER_ADDR', '127.0.0.1')
    tok = _token(world)
    if rank == 0:
        srv = None
        for port in _candidate_ports():
            try:
                srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                srv.bind((addr if addr not in ('localhost',) else '127.0.0.1', port))
                break
            except OSError:
                srv.close()
                srv = None
        if srv is None:
            raise RuntimeError('sella_amd.comm: no rendezvous port could be bound')
        srv.listen(world)
        srv.settimeout(timeout)
        served = set()
        try:
            while len(served) < world - 1:
                conn, _ = srv.accept()
                try:
                    conn.settimeout(10.0)
                    hello = _recv_exact(conn, 20)
                    if hello[:16] != tok:
                        continue
                    peer = struct.unpack('<i', hello[16:])[0]
                    conn.sendall(struct.pack('<q', len(payload)) + payload)
                    served.add(peer)
                except (OSError, ConnectionError):
                    pass
                finally:
                    conn.close()
        finally:
            srv.close()
        return payload
    deadline = time.time() + timeout
    last = None
    while time.time() < deadline:
        for port in _candidate_ports():
            try:
                with socket.create_connection((addr, port), timeout=2.0) as s:
                    s.sendall(tok + struct.pack('<i', rank))
                    n = struct.unpack('<q', _recv_exact(s, 8))[0]
                    return _recv_exact(s, n)
            except (OSError, ConnectionError) as e:
                last = e
        time.sleep(0.1)
    raise RuntimeError(f'sella_amd.comm: rendezvous with rank 0 failed: {last}')


# ---- RCCL through ctypes ---------------------------------------------------------------------------------------------
class _UniqueId(ctypes.Structure):
    _fields_ = [('internal', ctypes.c_ubyte * 128)]      # (c_char arrays read back truncated at the first NUL)


NCCL_DOUBLE, NCCL_SUM, NCCL_MAX = 8, 0, 2


def _load_rccl():
    names = [os.environ.get('SELLA_RCCL_LIB', ''), '/opt/rocm/lib/librccl.so.1', '/opt/rocm/lib/librccl.so',
             'librccl.so.1', 'librccl.so']
    err = None
    for name in names:
        if not name:
            continue
        try:
            lib = ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)
            break
        except OSError as e:
            err = e
    else:
        raise RuntimeError(f'librccl not found: {err}')
    vp = ctypes.c_void_p
    lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(vp), ctypes.c_int, _UniqueId, ctypes.c_int]
    lib.ncclAllGather.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_int, vp, vp]
    lib.ncclAllReduce.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, vp, vp]
    lib.ncclCommDestroy.argtypes = [vp]
    lib.ncclCommCount.argtypes = [vp, ctypes.POINTER(ctypes.c_int)]
    lib.ncclGetErrorString.restype = ctypes.c_char_p
    lib.ncclGetErrorString.argtypes = [ctypes.c_int]
    for f in ('ncclGetUniqueId', 'ncclCommInitRank', 'ncclAllGather', 'ncclAllReduce', 'ncclCommDestroy', 'ncclCommCount'):
        getattr(lib, f).restype = ctypes.c_int
    return lib


class RcclCommunicator:
    """ncclComm over the GPUs of one node; collectives run on a sella_amd Context's stream and buffers."""
    kind = 'rccl'

    def __init__(self, ctx, rank=None, world=None):
        self.rank = int(os.environ.get('RANK', '0')) if rank is None else int(rank)
        self.world = int(os.environ.get('WORLD_SIZE', '1')) if world is None else int(world)
        self.ctx = ctx                                    # created first: it made its device current
        self.lib = _load_rccl()
        uid = _UniqueId()
        if self.rank == 0:
            self._chk(self.lib.ncclGetUniqueId(ctypes.byref(uid)), 'ncclGetUniqueId')
        raw = broadcast_from_root(ctypes.string_at(ctypes.addressof(uid), 128) if self.rank == 0 else b'',
                                  self.rank, self.world)
        if len(raw) != 128:
            raise RuntimeError(f'sella_amd.comm: unique id of {len(raw)} bytes from the rendezvous')
        ctypes.memmove(ctypes.addressof(uid), raw, 128)
        self.comm = ctypes.c_void_p()
        self._chk(self.lib.ncclCommInitRank(ctypes.byref(self.comm), self.world, uid, self.rank), 'ncclCommInitRank')
        n = ctypes.c_int(0)
        self._chk(self.lib.ncclCommCount(self.comm, ctypes.byref(n)), 'ncclCommCount')
        self.nranks = n.value
        self._buf = None

    def _chk(self, code, what):
        if code != 0:
            raise RuntimeError(f'{what} failed: {self.lib.ncclGetErrorString(code).decode()}')

    def _buffers(self, count):
        """Device staging for host-side payloads: send (count) + recv (world * count) doubles."""
        need = count * (self.world + 1)
        if self._buf is None or self._buf[1] < need:
            M = self.ctx.zeros(1, max(need, 1024))
            self._buf = (M, M.shape[1], self.ctx.device_pointer(M)[0])
        return self._buf[2], self._buf[2] + 8 * count

    def allgather_device(self, ctx, send, recv, nbytes, stream):
        """recv[r * nbytes : (r + 1) * nbytes] = rank r's send — device pointers, the context's stream."""
        self._chk(self.lib.ncclAllGather(send, recv, nbytes // 8, NCCL_DOUBLE, self.comm, stream), 'ncclAllGather')

    def allgather_host(self, arr):
        """(world, len) array of every rank's fixed-size float64 payload, through RCCL."""
        arr = np.ascontiguousarray(arr, dtype=np.float64).ravel()
        send, recv = self._buffers(arr.size)
        self.ctx.host_to_device(send, arr)
        self._chk(self.lib.ncclAllGather(send, recv, arr.size, NCCL_DOUBLE, self.comm, self.ctx.stream), 'ncclAllGather')
        return self.ctx.device_to_host(recv, 8 * arr.size * self.world).reshape(self.world, arr.size)

    def max_host(self, value):
        send, recv = self._buffers(1)
        self.ctx.host_to_device(send, np.array([float(value)]))
        self._chk(self.lib.ncclAllReduce(send, recv, 1, NCCL_DOUBLE, NCCL_MAX, self.comm, self.ctx.stream), 'ncclAllReduce')
        return float(self.ctx.device_to_host(recv, 8)[0])

    def barrier(self):
        self.ctx.sync()
        self.max_host(0.0)

    def close(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None


class GlooCommunicator:
    """CPU-test shim over an initialised torch.distributed group (gloo); device buffers are staged through the host."""
    kind = 'torch.distributed'

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def _gather(self, arr):
        import torch
        arr = np.ascontiguousarray(arr, dtype=np.float64).ravel()
        on_gpu = self.dist.get_backend() == 'nccl'
        dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0'))) if on_gpu else torch.device('cpu')
        send = torch.from_numpy(arr.copy()).to(dev)
        recv = torch.empty(self.world * arr.size, dtype=torch.float64, device=dev)
        self.dist.all_gather_into_tensor(recv, send)
        return recv.cpu().numpy().reshape(self.world, arr.size)

    def allgather_host(self, arr):
        return self._gather(arr)

    def allgather_device(self, ctx, send, recv, nbytes, stream):
        ctx.sync()
        ctx.host_to_device(recv, self._gather(ctx.device_to_host(send, nbytes)))

    def max_host(self, value):
        return float(self._gather(np.array([float(value)])).max())

    def barrier(self):
        self.dist.barrier()

    def close(self):
        pass


_comm = None


def torch_group():
    if 'torch' not in sys.modules:
        return None
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def get_communicator(ctx=None):
    """The process-wide communicator (created on first use; see the module docstring for the choice)."""
    global _comm
    if _comm is not None:
        return _comm
    if torch_group() is not None:
        _comm = GlooCommunicator()
    elif int(os.environ.get('WORLD_SIZE', '1')) > 1:
        if ctx is None:
            from .device import get_context
            ctx = get_context()
        _comm = RcclCommunicator(ctx)
    else:
        _comm = SingleProcess()
    return _comm


def reset_communicator():
    global _comm
    if _comm is not None:
        _comm.close()
    _comm = None
