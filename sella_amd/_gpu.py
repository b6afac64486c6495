"""Accelerator seam — same function names and numpy-in / numpy-out contract as the
reference's sella/_gpu.py:38-132, backed by libsella_hip (hand-written HIP for gfx950)
instead of torch.cuda.

Differences, on purpose:
  * there is no CPU fallback and no size gate: every call runs on the MI355X; a failure
    raises `SellaHipError` instead of silently switching to LAPACK (reference: _gpu.py:82-84);
  * device handles are `DeviceMatrix` objects instead of torch tensors.
"""
import numpy as np

from .device import DeviceMatrix, get_context


def _gpu_ok(n):
    """The reference gates on SELLA_GPU_MIN_DIM / OOM history (_gpu.py:38-41); here the device
    path is the only path."""
    return True


def to_gpu(A):
    """Upload a numpy array as a contiguous fp64 device matrix (_gpu.py:55-67)."""
    return get_context().upload(np.ascontiguousarray(A, dtype=np.float64))


def gpu_eigh(A, A_gpu=None):
    """Eigendecomposition, numpy out (_gpu.py:70-84)."""
    ctx = get_context()
    At = A_gpu if A_gpu is not None else ctx.upload(A)
    w, V, Vt = ctx.eigh(At)
    out = V.numpy()
    V.free()
    Vt.free()
    return w, out


def gpu_eigh_t(A_gpu):
    """Eigendecomposition that keeps the eigenvectors on the device (_gpu.py:87-97).
    Returns (evals numpy, evecs DeviceMatrix [columns], evecsT DeviceMatrix [rows])."""
    return get_context().eigh(A_gpu)


def gpu_qr(A):
    """Economy QR (_gpu.py:100-111)."""
    return get_context().qr_thin(A)


def gpu_project(H, U, H_gpu=None):
    """U.T @ H @ U as a numpy array (_gpu.py:114-132)."""
    ctx = get_context()
    Ht = H_gpu if H_gpu is not None else ctx.upload(H)
    return ctx.project(Ht, U)


__all__ = ['to_gpu', 'gpu_eigh', 'gpu_eigh_t', 'gpu_qr', 'gpu_project', 'DeviceMatrix']
