"""Python face of the C ABI: a `Context` (one HIP device + stream) and `DeviceMatrix`
handles.  numpy in / numpy out, fp64, like the reference's accelerator seam
(sella/_gpu.py:55-132) — but the device side is libsella_hip, not torch.
"""
import ctypes
import os
import threading
import weakref
from ctypes import byref, c_char_p, c_double, c_int, c_long, c_void_p, create_string_buffer, pointer

import sys

import numpy as np

from . import _lib
from ._lib import SELLA_NO_MAT, as_f64, check, ptr

DAVIDSON_METHODS = {'lanczos': 0, 'gd': 1, 'jd0': 2, 'jd0_alt': 3, 'mjd0': 4, 'mjd0_alt': 5}
UPDATE_METHODS = {'TS-BFGS': 0, 'BFGS': 1, 'PSB': 2, 'DFP': 3, 'SR1': 4, 'Greenstadt': 5,
                  'BFGS_auto': 6}


class DeviceMatrix:
    """Row-major fp64 matrix resident in HBM (opaque handle owned by a Context)."""

    def __init__(self, ctx, handle, shape):
        self.ctx = ctx
        self.handle = handle
        self.shape = tuple(shape)
        self._fin = weakref.finalize(self, DeviceMatrix._release, ctx, handle)

    @staticmethod
    def _release(ctx, handle):
        if ctx._h is None:
            return
        if threading.get_ident() == ctx._owner:
            _lib.lib().sella_mat_free(ctx._h, handle)
        else:
            # the garbage collector may run this in another thread: a context is single-threaded, so
            # the handle is parked and freed by the owning thread at its next allocation
            ctx._parked.append(handle)

    def free(self):
        self._fin()

    def numpy(self):
        out = np.empty(self.shape, dtype=np.float64)
        check(_lib.lib().sella_mat_download(self.ctx._h, self.handle, ptr(out)))
        return out

    def set(self, A):
        A = as_f64(A, self.shape)
        check(_lib.lib().sella_mat_set(self.ctx._h, self.handle, ptr(A)))

    def copy(self):
        h = c_int(-1)
        check(_lib.lib().sella_mat_copy(self.ctx._h, self.handle, byref(h)))
        return DeviceMatrix(self.ctx, h.value, self.shape)

    def transpose(self):
        h = c_int(-1)
        check(_lib.lib().sella_mat_transpose(self.ctx._h, self.handle, byref(h)))
        return DeviceMatrix(self.ctx, h.value, self.shape[::-1])


class Context:
    def __init__(self, device=None):
        L = _lib.lib()
        if device is None:
            device = int(os.environ.get('SELLA_HIP_DEVICE', os.environ.get('LOCAL_RANK', '0')))
            n = c_int(0)
            check(L.sella_device_count(byref(n)))
            if n.value > 0:
                device %= n.value
        h = c_void_p()
        check(L.sella_ctx_create(int(device), byref(h)))
        self._h = h
        self.device = int(device)
        self._owner = threading.get_ident()
        self._parked = []
        # finalizers of the library objects created on this context (calculators, finite-difference operators, step
        # families, searches): they hold a pointer to the context, so `close()` runs them BEFORE the context goes — an
        # object that outlives its context (a test failure keeps frames alive until interpreter exit) then has nothing
        # left to destroy instead of a dangling pointer
        self._children = []
        # the finalizer holds only the raw handle: passing `self` would keep the context alive for ever
        self._fin = weakref.finalize(self, Context._destroy, self._children, h)

    @staticmethod
    def _destroy(children, h):
        while children:
            children.pop()()                    # newest first; weakref.finalize objects are idempotent
        _lib.lib().sella_ctx_destroy(h)

    def child(self, fin):
        """Register the finalizer of an object that lives on this context; returns it."""
        if len(self._children) > 256:
            self._children[:] = [f for f in self._children if f.alive]
        self._children.append(fin)
        return fin

    def _drain(self):
        while self._parked:
            _lib.lib().sella_mat_free(self._h, self._parked.pop())

    def close(self):
        if self._h is not None:
            self._fin()
            self._h = None

    # ---- misc -------------------------------------------------------------------------
    @property
    def name(self):
        buf = create_string_buffer(256)
        check(_lib.lib().sella_ctx_device_name(self._h, buf, 256))
        return buf.value.decode()

    def sync(self):
        check(_lib.lib().sella_ctx_sync(self._h))

    def set_option(self, key, value):
        check(_lib.lib().sella_ctx_set_option(self._h, key.encode(), int(value)))

    # ---- matrices ---------------------------------------------------------------------
    def upload(self, A):
        self._drain()
        A = as_f64(A)
        if A.ndim == 1:
            A = A[:, None]
        h = c_int(-1)
        check(_lib.lib().sella_mat_upload(self._h, ptr(A), A.shape[0], A.shape[1], byref(h)))
        return DeviceMatrix(self, h.value, A.shape)

    def resident(self, A):
        """Device copy of a host array that is handed out as the SAME object over and over (projection bases
        shared across geometries, peswrapper.py): uploaded once and kept, keyed on the object.  The array must not
        be modified afterwards; at most four copies are kept, least recently used dropped."""
        cache = self.__dict__.setdefault('_resident', [])
        for k, (host, dev) in enumerate(cache):
            if host is A:
                cache.append(cache.pop(k))
                return dev
        dev = self.upload(A)
        cache.append((A, dev))
        if len(cache) > 4:
            cache.pop(0)[1].free()
        return dev

    def zeros(self, rows, cols):
        self._drain()
        h = c_int(-1)
        check(_lib.lib().sella_mat_alloc(self._h, rows, cols, byref(h)))
        return DeviceMatrix(self, h.value, (rows, cols))

    def axpby(self, alpha, A, beta=0.0, B=None):
        h = c_int(-1)
        check(_lib.lib().sella_mat_axpby(self._h, float(alpha), A.handle, float(beta),
                                         SELLA_NO_MAT if B is None else B.handle, byref(h)))
        return DeviceMatrix(self, h.value, A.shape)

    # ---- products -----------------------------------------------------------------------
    def symm_mm(self, A, X):
        """A @ X with A resident; X (n,) or (n, k) host."""
        X = as_f64(X)
        one_d = X.ndim == 1
        X2 = X[:, None] if one_d else X
        X2 = np.ascontiguousarray(X2)
        if X2.shape[0] != A.shape[1]:
            raise ValueError(f'dimension mismatch: {A.shape} @ {X.shape}')
        Y = np.empty((A.shape[0], X2.shape[1]))
        check(_lib.lib().sella_symm_mm(self._h, A.handle, ptr(X2), X2.shape[1], ptr(Y)))
        return Y[:, 0] if one_d else Y

    def tmatmul(self, A, X):
        """A.T @ X with A resident."""
        X = as_f64(X)
        one_d = X.ndim == 1
        X2 = np.ascontiguousarray(X[:, None] if one_d else X)
        if X2.shape[0] != A.shape[0]:
            raise ValueError(f'dimension mismatch: {A.shape}.T @ {X.shape}')
        Y = np.empty((A.shape[1], X2.shape[1]))
        check(_lib.lib().sella_gemm_tn_host(self._h, A.handle, ptr(X2), X2.shape[1], ptr(Y)))
        return Y[:, 0] if one_d else Y

    def project(self, H, U):
        """U.T @ H @ U (numpy result), H resident, U host (n, m)."""
        U = as_f64(U)
        out = np.empty((U.shape[1], U.shape[1]))
        check(_lib.lib().sella_project(self._h, H.handle, ptr(U), U.shape[1], ptr(out)))
        return out

    def project_dev(self, H, U):
        h = c_int(-1)
        check(_lib.lib().sella_project_dev(self._h, H.handle, U.handle, byref(h)))
        return DeviceMatrix(self, h.value, (U.shape[1], U.shape[1]))

    def gemm(self, A, B, C, transA=False, transB=False, alpha=1.0, beta=0.0):
        check(_lib.lib().sella_gemm(self._h, int(transA), int(transB), float(alpha), A.handle,
                                    B.handle, float(beta), C.handle))
        return C

    # ---- factorizations -------------------------------------------------------------------
    def eigh(self, A, vectors=True):
        """(w, V, Vt): eigenvalues ascending (numpy), eigenvectors as columns / rows (device)."""
        self._drain()
        n = A.shape[0]
        w = np.empty(n)
        hv, hvt = c_int(-1), c_int(-1)
        check(_lib.lib().sella_eigh(self._h, A.handle, ptr(w), byref(hv) if vectors else None,
                                    byref(hvt) if vectors else None))
        if not vectors:
            return w, None, None
        return w, DeviceMatrix(self, hv.value, (n, n)), DeviceMatrix(self, hvt.value, (n, n))

    def rank1_eig(self, D, w, rho, vectors=True):
        """Eigenpairs of diag(D) + rho w w^T (D strictly ascending, w nonzero): (lam, Ut rows)."""
        D = as_f64(D)
        w = as_f64(w)
        K = D.shape[0]
        lam = np.empty(K)
        Ut = np.empty((K, K)) if vectors else None
        check(_lib.lib().sella_rank1_eig(self._h, K, ptr(D), ptr(w), float(rho), ptr(lam),
                                         ptr(Ut) if vectors else None))
        return lam, Ut

    def qr_thin(self, A):
        A = as_f64(A)
        m, n = A.shape
        Q = np.empty((m, n))
        R = np.empty((n, n))
        check(_lib.lib().sella_qr_thin(self._h, ptr(A), m, n, ptr(Q), ptr(R)))
        return Q, R

    def mgs(self, X, Y=None, eps1=1e-15, eps2=1e-6, maxiter=100):
        X = as_f64(X)
        n, nx = X.shape
        ny = 0
        if Y is not None:
            Y = as_f64(Y)
            ny = Y.shape[1]
        out = np.zeros((n, nx))
        kept = c_int(0)
        check(_lib.lib().sella_mgs(self._h, ptr(X), n, nx, ptr(Y) if ny else None, ny, eps1, eps2,
                                   maxiter, ptr(out), byref(kept)))
        return np.ascontiguousarray(out[:, :kept.value])

    # ---- Davidson -------------------------------------------------------------------------
    def davidson(self, A, n, v0, gamma, method='jd0', maxiter=None, Pvecs=None, PvecsT=None,
                 pevals=None, pscale=1.0, vref=None, vreftol=0.99):
        """A: DeviceMatrix or a python callable v -> A v.  Returns (lams, V, AV, nmatvec)."""
        if method not in DAVIDSON_METHODS:
            raise ValueError("Unknown diagonalization method {}".format(method))
        v0 = as_f64(v0)
        if v0.ndim == 1:
            v0 = v0[:, None]
        v0 = np.ascontiguousarray(v0)
        nv0 = v0.shape[1]
        if maxiter is None:
            maxiter = 2 * n + 1
        kmax = min(n, max(int(maxiter), nv0))
        lams = np.zeros(kmax + 1)
        # (the library writes n x k of these; fresh zeroed megabytes per call were 0.16 ms of page faults in a 1.3 ms call:
        # the context keeps one pair of scratch arrays per size and the results are copied out of it)
        # Results are VIEWS of the output arrays; a pair of arrays is reused only when nobody holds such a view any more (its
        # reference count is back to the pool's own), so a caller may keep results of earlier calls as long as it likes.
        key = n * (kmax + 1)
        bufs = self.__dict__.setdefault('_dav_out', {})
        if key not in bufs and len(bufs) > 4:
            bufs.clear()
        pool = bufs.setdefault(key, [])
        V = AV = None
        for pair in pool:
            if sys.getrefcount(pair[0]) == 2 and sys.getrefcount(pair[1]) == 2:      # (the tuple's reference + the call's argument)
                V, AV = pair
                break
        if V is None:
            V, AV = np.empty(key), np.empty(key)
            if len(pool) < 4:
                pool.append((V, AV))
        k = c_int(0)
        nmv = c_int(0)
        err = []
        user = None
        if isinstance(A, DeviceMatrix):
            hA, cb = A.handle, _lib.MATVEC_FN()
        elif isinstance(A, DeviceFdOperator):
            # the finite-difference Hessian of a calculator that lives in the library: its products are library calls
            hA, cb, user = SELLA_NO_MAT, A.callback(), A._h
        else:
            hA = SELLA_NO_MAT

            def _cb(user, vp, avp, nn):
                try:
                    v = np.ctypeslib.as_array(vp, shape=(nn,)).copy()
                    out = np.asarray(A(v), dtype=np.float64).ravel()
                    np.ctypeslib.as_array(avp, shape=(nn,))[:] = out
                    return 0
                except BaseException as e:      # noqa: B902 — must not unwind through C
                    err.append(e)
                    return 1
            cb = _lib.MATVEC_FN(_cb)
        pe = as_f64(pevals) if pevals is not None else None
        vr = as_f64(vref) if vref is not None else None
        st = _lib.lib().sella_davidson(
            self._h, hA, cb, user,
            SELLA_NO_MAT if Pvecs is None else Pvecs.handle,
            SELLA_NO_MAT if PvecsT is None else PvecsT.handle,
            ptr(pe), float(pscale), int(n), ptr(v0), nv0, float(gamma),
            DAVIDSON_METHODS[method], int(maxiter), ptr(vr), float(vreftol),
            ptr(lams), ptr(V), ptr(AV), byref(k), byref(nmv))
        if err:
            raise err[0]
        check(st)
        kk = k.value
        return (lams[:kk], V[:n * kk].reshape(n, kk), AV[:n * kk].reshape(n, kk), nmv.value)

    def davidson_block(self, A, n, nev, block=16, tol=1e-8, maxiter=500, maxvec=0, V0=None, Pvecs=None,
                       PvecsT=None, pevals=None, diag=None, row0=0, world=1, allgather=None):
        """Lowest `nev` eigenpairs by block Davidson (csrc/davidson_block.hip).  A: DeviceMatrix (n x n), or this
        rank's row panel with `allgather(send_ptr, recv_ptr, nbytes, stream_ptr) -> None` assembling the block
        product over the ranks.  Returns dict(lams, V (n x nev), res, niter, nmatvec, nconv)."""
        lams = np.zeros(nev)
        V = np.zeros((n, nev))
        res = np.zeros(nev)
        niter, nmv, nconv = c_int(0), c_int(0), c_int(0)
        err = []
        if allgather is None:
            cb = _lib.ALLGATHER_FN()
        else:
            def _cb(user, send, recv, nbytes, stream):
                try:
                    allgather(send, recv, nbytes, stream)
                    return 0
                except BaseException as e:      # noqa: B902 — must not unwind through C
                    err.append(e)
                    return 1
            cb = _lib.ALLGATHER_FN(_cb)
        v0 = None
        nv0 = 0
        if V0 is not None:
            v0 = as_f64(V0)
            if v0.ndim == 1:
                v0 = v0[:, None]
            v0 = np.ascontiguousarray(v0)
            nv0 = v0.shape[1]
        pe = as_f64(pevals) if pevals is not None else None
        dg = as_f64(diag) if diag is not None else None
        st = _lib.lib().sella_davidson_block(
            self._h, A.handle, int(n), int(row0), int(world), cb, None,
            SELLA_NO_MAT if Pvecs is None else Pvecs.handle, SELLA_NO_MAT if PvecsT is None else PvecsT.handle,
            ptr(pe), ptr(dg), ptr(v0), nv0, int(nev), int(block), int(maxvec), float(tol), int(maxiter),
            ptr(lams), ptr(V), ptr(res), byref(niter), byref(nmv), byref(nconv))
        if err:
            raise err[0]
        check(st)
        return dict(lams=lams, V=V, res=res, niter=niter.value, nmatvec=nmv.value, nconv=nconv.value)

    @property
    def stream(self):
        """Address of the context's hipStream_t (for collectives issued on the library's own stream)."""
        st = c_void_p()
        check(_lib.lib().sella_ctx_stream(self._h, byref(st)))
        return st.value or 0

    def device_pointer(self, M):
        """(device address, leading dimension) of a resident matrix."""
        p, ld = c_void_p(), c_int(0)
        check(_lib.lib().sella_mat_ptr(self._h, M.handle, byref(p), byref(ld)))
        return p.value, ld.value

    def copy_device(self, dst, src, nbytes):
        """Device-to-device copy between raw device addresses (buffers handed to an all-gather callback)."""
        check(_lib.lib().sella_dev_copy(self._h, c_void_p(dst), c_void_p(src), int(nbytes), 0))

    def device_to_host(self, src, nbytes):
        out = np.empty(int(nbytes) // 8, dtype=np.float64)
        check(_lib.lib().sella_dev_copy(self._h, ptr(out), c_void_p(src), int(nbytes), 1))
        return out

    def host_to_device(self, dst, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        check(_lib.lib().sella_dev_copy(self._h, c_void_p(dst), ptr(arr), arr.nbytes, 2))

    # ---- quasi-Newton -----------------------------------------------------------------------
    def update_h(self, B, S, Y, method='TS-BFGS', symm=2, evals=None, evecs=None, evecsT=None):
        """In-place update of the resident B (n x n)."""
        if method not in UPDATE_METHODS:
            raise ValueError('Unknown update method {}'.format(method))
        S = as_f64(S)
        Y = as_f64(Y)
        n, k = S.shape
        ev = as_f64(evals) if evals is not None else None
        check(_lib.lib().sella_update_h(
            self._h, B.handle, SELLA_NO_MAT if evecs is None else evecs.handle,
            SELLA_NO_MAT if evecsT is None else evecsT.handle, ptr(ev), ptr(S), ptr(Y), n, k,
            UPDATE_METHODS[method], -1 if symm is None else int(symm)))

    def update_h_eig(self, B, S, Y, evals, evecs, evecsT, method='TS-BFGS', symm=2, max_rank=8):
        """update_h that also updates (evals, evecs, evecsT) of B by rank-one modifications.

        Returns (evals_new, nrank1); nrank1 == -1 means the eigendecomposition was NOT carried over
        (rank of the update above max_rank) and must be recomputed by the caller."""
        if method not in UPDATE_METHODS:
            raise ValueError('Unknown update method {}'.format(method))
        S = as_f64(S)
        Y = as_f64(Y)
        n, k = S.shape
        ev = np.array(evals, dtype=np.float64, copy=True)
        nr = c_int(-1)
        check(_lib.lib().sella_update_h_eig(
            self._h, B.handle, evecs.handle, evecsT.handle, ptr(ev), ptr(S), ptr(Y), n, k,
            UPDATE_METHODS[method], -1 if symm is None else int(symm), int(max_rank), byref(nr)))
        return ev, nr.value

    def update_h_eig_view(self, B, S, Y, evals, evecs, evecsT, Bsub, idx, evals_sub=None, evecs_sub=None,
                          evecsT_sub=None, method='TS-BFGS', symm=2, max_rank=8):
        """`update_h_eig` that keeps the principal submatrix Bsub = B[idx][idx] (and, when given, its
        eigendecomposition) in step.  Returns (evals_new, nrank1, evals_sub_new or None, nrank1_sub)."""
        if method not in UPDATE_METHODS:
            raise ValueError('Unknown update method {}'.format(method))
        S = as_f64(S)
        Y = as_f64(Y)
        n, k = S.shape
        ev = np.array(evals, dtype=np.float64, copy=True)
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        have = evals_sub is not None and evecs_sub is not None and evecsT_sub is not None
        evs = np.array(evals_sub, dtype=np.float64, copy=True) if have else None
        nr, nrs = c_int(-1), c_int(-1)
        check(_lib.lib().sella_update_h_eig_view(
            self._h, B.handle, evecs.handle, evecsT.handle, ptr(ev), ptr(S), ptr(Y), n, k,
            UPDATE_METHODS[method], -1 if symm is None else int(symm), int(max_rank), byref(nr),
            Bsub.handle, evecs_sub.handle if have else SELLA_NO_MAT, evecsT_sub.handle if have else SELLA_NO_MAT,
            ptr(evs) if have else None, idx.ctypes.data_as(c_void_p), len(idx), byref(nrs)))
        return ev, nr.value, evs, nrs.value

    # ---- structured eigendecompositions (lam0 * I + rank r; csrc/eigh.hip lr_lowrank_update) -----------------------
    def mat_rows(self, M, row0, nrows):
        h = c_int(-1)
        check(_lib.lib().sella_mat_rows(self._h, M.handle, int(row0), int(nrows), byref(h)))
        return DeviceMatrix(self, h.value, (int(nrows), M.shape[1]))

    def mat_copy_into(self, src, dst, nrows):
        check(_lib.lib().sella_mat_copy_into(self._h, src.handle, dst.handle, int(nrows)))

    def mat_add_diag(self, M, alpha):
        check(_lib.lib().sella_mat_add_diag(self._h, M.handle, float(alpha)))

    def update_h_lr(self, B, S, Y, lr, method='TS-BFGS', symm=2, view=None):
        """`sella_update_h_lr`: quasi-Newton update of the dense B together with its STRUCTURED eigendecomposition
        `lr` = dict(Wt=DeviceMatrix (capacity x n), r, mu (capacity,), lam0), updated in place; `view` =
        (Bsub DeviceMatrix, idx, lr_sub or None) keeps a principal submatrix in step.  Returns (nrank1, nrank1_sub)."""
        if method not in UPDATE_METHODS:
            raise ValueError('Unknown update method {}'.format(method))
        S = as_f64(S)
        Y = as_f64(Y)
        n, k = S.shape
        r = c_int(int(lr['r']))
        nr, nrs = c_int(-1), c_int(-1)
        if view is None:
            check(_lib.lib().sella_update_h_lr(
                self._h, B.handle, lr['Wt'].handle, byref(r), ptr(lr['mu']), float(lr['lam0']), ptr(S), ptr(Y), n, k,
                UPDATE_METHODS[method], -1 if symm is None else int(symm), byref(nr), SELLA_NO_MAT, SELLA_NO_MAT, None,
                None, None, 0, None))
        else:
            Bsub, idx, lrs = view
            idx = np.ascontiguousarray(idx, dtype=np.int32)
            rs = c_int(int(lrs['r']) if lrs is not None else 0)
            check(_lib.lib().sella_update_h_lr(
                self._h, B.handle, lr['Wt'].handle, byref(r), ptr(lr['mu']), float(lr['lam0']), ptr(S), ptr(Y), n, k,
                UPDATE_METHODS[method], -1 if symm is None else int(symm), byref(nr), Bsub.handle,
                lrs['Wt'].handle if lrs is not None else SELLA_NO_MAT, byref(rs) if lrs is not None else None,
                ptr(lrs['mu']) if lrs is not None else None, idx.ctypes.data_as(c_void_p), len(idx), byref(nrs)))
            if lrs is not None:
                lrs['r'] = rs.value
        lr['r'] = r.value
        return nr.value, nrs.value

    def lr_restrict(self, lr, idx, capacity):
        """Structured eigendecomposition of the principal submatrix [idx][idx] (`sella_lr_restrict`)."""
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        m = len(idx)
        Wts = self.zeros(int(capacity), m)
        mus = np.zeros(int(capacity))
        rs = c_int(0)
        check(_lib.lib().sella_lr_restrict(self._h, lr['Wt'].handle, int(lr['r']), ptr(lr['mu']), float(lr['lam0']),
                                           idx.ctypes.data_as(c_void_p), m, Wts.handle, byref(rs), ptr(mus)))
        return dict(Wt=Wts, r=rs.value, mu=mus, lam0=float(lr['lam0']))

    def lr_materialize(self, B, Wt, r, mu, lam0):
        """`sella_lr_materialize`: B <- lam0 I + W^T diag(mu - lam0) W (dense mirror of a structured decomposition)."""
        check(_lib.lib().sella_lr_materialize(self._h, B.handle, Wt.handle, int(r), ptr(mu), float(lam0)))

    def opt_step(self, args):
        """`sella_opt_step`: one optimizer step (learn + adapt + propose) in one call; `args` is an `OptStep` holder."""
        check(_lib.lib().sella_opt_step(self._h, byref(args.c)))

    def symmetrize_y(self, S, Y, symm):
        S = as_f64(S)
        Y = as_f64(Y)
        n, k = S.shape
        out = np.empty((n, k))
        check(_lib.lib().sella_symmetrize_y(self._h, ptr(S), ptr(Y), n, k,
                                            -1 if symm is None else int(symm), ptr(out)))
        return out

    # ---- internal-coordinate primitives -----------------------------------------------------------
    def internals_eval(self, pos, tvec=None, tangent=None, hessian=False):
        """Batched q, dq/dx[, H t][, H] of bonds / angles / dihedrals (pos: (nc, 2|3|4, 3))."""
        pos = as_f64(pos)
        nc, na = pos.shape[0], pos.shape[1]
        tv = as_f64(tvec) if tvec is not None else None
        tan = as_f64(tangent) if tangent is not None else None
        q = np.empty(nc)
        grad = np.empty((nc, na, 3))
        hvp = np.empty((nc, na, 3)) if tan is not None else None
        hess = np.empty((nc, na, 3, na, 3)) if hessian else None
        check(_lib.lib().sella_internals_eval(self._h, na, nc, ptr(pos), ptr(tv), ptr(tan), ptr(q), ptr(grad),
                                              ptr(hvp), ptr(hess)))
        return q, grad, hvp, hess

    # ---- EMT calculator -------------------------------------------------------------------------
    def emt_eval(self, pos, par, shifts, rc, acut, cutoff, beta):
        """(energy, gradient (n, 3)) of the EMT potential; par (9, n), shifts (nshift, 3)."""
        pos = as_f64(pos)
        par = as_f64(par)
        shifts = as_f64(shifts)
        n = pos.shape[0]
        e = c_double(0.0)
        grad = np.empty((n, 3))
        check(_lib.lib().sella_emt_eval(self._h, n, ptr(pos), ptr(par), shifts.shape[0], ptr(shifts), float(rc),
                                        float(acut), float(cutoff), float(beta), byref(e), ptr(grad)))
        return e.value, grad

    # ---- profiling ---------------------------------------------------------------------------
    def prof_enable(self, on=True):
        check(_lib.lib().sella_prof_enable(self._h, int(bool(on))))

    def prof_reset(self):
        check(_lib.lib().sella_prof_reset(self._h))

    def prof_get(self, kind):
        n, ms, b, f = c_long(0), c_double(0), c_double(0), c_double(0)
        check(_lib.lib().sella_prof_get(self._h, int(kind), byref(n), byref(ms), byref(b), byref(f)))
        return dict(launches=n.value, ms=ms.value, bytes=b.value, flops=f.value)


class DeviceCalculator:
    """A calculator that lives in the library (`sella_calc_*`, csrc/calc.hip): energy and gradient without a
    host-language frame, so that the finite-difference products of an iterative diagonalisation are library calls."""

    def __init__(self, ctx, handle, keep=()):
        self.ctx, self._h, self._keep = ctx, handle, keep
        self._fin = ctx.child(weakref.finalize(self, _lib.lib().sella_calc_destroy, handle))

    @classmethod
    def model(cls, ctx, A, U, c):
        U = as_f64(U)
        h = c_void_p()
        check(_lib.lib().sella_calc_model_create(ctx._h, A.handle, ptr(U), U.shape[0], U.shape[1], float(c), byref(h)))
        return cls(ctx, h, (A,))

    @classmethod
    def emt(cls, ctx, natoms, par, shifts, rc, acut, cutoff, beta):
        par, shifts = as_f64(par), as_f64(shifts)
        h = c_void_p()
        check(_lib.lib().sella_calc_emt_create(ctx._h, int(natoms), ptr(par), shifts.shape[0], ptr(shifts), float(rc),
                                               float(acut), float(cutoff), float(beta), byref(h)))
        return cls(ctx, h)

    def eval(self, x):
        x = as_f64(x).ravel()
        e = c_double(0.0)
        g = np.empty(x.size)
        check(_lib.lib().sella_calc_eval(self._h, ptr(x), byref(e), ptr(g)))
        return e.value, g

    ncalls = property(lambda self: int(_lib.lib().sella_calc_ncalls(self._h)))


class DeviceFdOperator:
    """`NumericalHessian` (sella/linalg.py:14-101) over a `DeviceCalculator`: `sella_fd_*`.  Passed to `Context.davidson`
    as the operator; `Vs` / `AVs` afterwards hold the recorded secant pairs (full space, one column per product)."""

    def __init__(self, calc, x0, g0, eta, threepoint=False, free=None):
        x0, g0 = as_f64(x0).ravel(), as_f64(g0).ravel()
        self.calc, self.ntrue = calc, x0.size
        self._free = None if free is None else np.ascontiguousarray(free, dtype=np.int32)
        n = self.ntrue if self._free is None else len(self._free)
        self.shape = (n, n)
        h = c_void_p()
        check(_lib.lib().sella_fd_create(calc._h, self.ntrue, ptr(x0), ptr(g0), float(eta), int(bool(threepoint)),
                                         None if self._free is None else self._free.ctypes.data_as(c_void_p),
                                         0 if self._free is None else len(self._free), byref(h)))
        self._h = h
        self._fin = calc.ctx.child(weakref.finalize(self, _lib.lib().sella_fd_destroy, h))

    def callback(self):
        """`sella_fd_matvec` as the `sella_matvec_fn` of `sella_davidson` (a function of the library itself)."""
        return ctypes.cast(_lib.lib().sella_fd_matvec, _lib.MATVEC_FN)

    calls = property(lambda self: int(_lib.lib().sella_fd_calls(self._h)))

    def _pairs(self):
        k = int(_lib.lib().sella_fd_npairs(self._h))
        Vs, AVs = np.empty((self.ntrue, k)), np.empty((self.ntrue, k))
        if k:
            check(_lib.lib().sella_fd_pairs(self._h, ptr(Vs), ptr(AVs)))
        return Vs, AVs

    Vs = property(lambda self: self._pairs()[0])
    AVs = property(lambda self: self._pairs()[1])


class OptStep:
    """Argument block of `sella_opt_step` (include/sella_hip.h) together with the arrays it points into — the block
    holds raw addresses, so the arrays live here for as long as the block does."""
    LEARN, PROPOSE = 1, 2

    def __init__(self, n):
        self.c = _lib.OptStepArgs()
        self.c.n = int(n)
        self.s = np.zeros(int(n))
        self.c.s_out = self.s.ctypes.data
        self._r, self._r_sub = c_int(0), c_int(0)
        self._keep = {}

    def point(self, field, array):
        """Store the address of a float64 / int32 array in `field` and keep the array alive."""
        self._keep[field] = array
        setattr(self.c, field, None if array is None else array.ctypes.data)

    def set_hessian(self, B, lr, method, symm, view=None, stale=False, stale_sub=False):
        c = self.c
        c.B, c.Wt, c.lam0 = B.handle, lr['Wt'].handle, float(lr['lam0'])
        c.B_stale, c.Bsub_stale = int(bool(stale)), int(bool(stale_sub))
        self._r.value = int(lr['r'])
        c.r = pointer(self._r)
        self.point('mu', lr['mu'])
        c.update_method, c.symm = UPDATE_METHODS[method], -1 if symm is None else int(symm)
        if view is None:
            c.Bsub, c.Wt_sub, c.m = SELLA_NO_MAT, SELLA_NO_MAT, 0
            c.r_sub = None
            self.point('mu_sub', None)
            self.point('idx', None)
        else:
            Bsub, idx, lrs = view
            c.Bsub, c.Wt_sub, c.m = Bsub.handle, lrs['Wt'].handle, len(idx)
            self._r_sub.value = int(lrs['r'])
            c.r_sub = pointer(self._r_sub)
            self.point('mu_sub', lrs['mu'])
            self.point('idx', idx)

    r = property(lambda self: self._r.value)
    r_sub = property(lambda self: self._r_sub.value)


STEPPER_KINDS = {'qn': 0, 'rfo': 1, 'prfo': 2, 'qn_irc': 3}
CONSTRAINT_KINDS = {'tr': 0, 'ras': 1, 'mis': 2, 'sphere': 3}


class DeviceStepper:
    """One (g, H, order) step-family instance evaluated in the eigenbasis of H
    (sella_amd/csrc/stepper.hip).  V is (nout x m) with the eigenvectors (optionally already
    multiplied by a projection basis) as columns, Vt its transpose; g has Vt.shape[1] entries."""

    def __init__(self, ctx, kind, V, Vt, evals, g, order, lr=None):
        self.ctx = ctx
        g = as_f64(g)
        h = c_void_p()
        if lr is not None:
            # structured eigendecomposition (`sella_stepper_create_lr`): r explicit pairs + lam0 on the complement
            self.nout = lr['Wt'].shape[1]
            self._keep = (lr['Wt'],)
            check(_lib.lib().sella_stepper_create_lr(ctx._h, STEPPER_KINDS[kind], lr['Wt'].handle, int(lr['r']),
                                                     ptr(lr['mu']), float(lr['lam0']), ptr(g), self.nout, int(order),
                                                     byref(h)))
        else:
            self.nout = V.shape[0]
            self._keep = (V, Vt)
            evals = as_f64(evals)
            check(_lib.lib().sella_stepper_create(ctx._h, STEPPER_KINDS[kind], V.handle, Vt.handle,
                                                  ptr(evals), ptr(g), len(evals), int(order), byref(h)))
        self._h = h
        self._fin = ctx.child(weakref.finalize(self, _lib.lib().sella_stepper_destroy, h))

    def get_s(self, alpha):
        s = np.empty(self.nout)
        dsda = np.empty(self.nout)
        check(_lib.lib().sella_stepper_get_s(self._h, float(alpha), ptr(s), ptr(dsda)))
        return s, dsda

    def set_d1hat(self, d1hat):
        """V^T d1 of the IRC quasi-Newton family (m entries, eigenbasis of the projected Hessian)."""
        d1hat = as_f64(d1hat)
        check(_lib.lib().sella_stepper_set_d1hat(self._h, ptr(d1hat), len(d1hat)))

    def restricted_step(self, cons, delta, alpha0, alphamin, alphamax, slope, newton_safe, tol, maxiter=1000,
                        scons=None, w=None, d1=None, orthonormal=False, sel=None, nfull=0):
        """The whole trust-radius root find in one call (`sella_restricted_step`):
        returns (s_total (nout,), val, alphas tried)."""
        sc = as_f64(scons) if scons is not None else None
        ww = as_f64(w) if w is not None else None
        dd = as_f64(d1) if d1 is not None else None
        isel = np.ascontiguousarray(sel, dtype=np.int32) if sel is not None else None
        s = np.empty(int(nfull) if sel is not None else self.nout)
        val = c_double(0.0)
        alphas = np.empty(int(maxiter) + 1)
        na = c_int(0)
        big = 1.7976931348623157e308
        check(_lib.lib().sella_restricted_step(
            self._h, CONSTRAINT_KINDS[cons], float(delta), ptr(sc), ptr(ww), ptr(dd), float(alpha0),
            float(max(alphamin, -big)), float(alphamax), float(slope), int(bool(newton_safe)), int(bool(orthonormal)),
            float(tol), int(maxiter), isel.ctypes.data_as(c_void_p) if isel is not None else None, int(nfull),
            ptr(s), byref(val), ptr(alphas), byref(na)))
        return s, val.value, alphas[:min(na.value, len(alphas))].copy()


_default = None


_tls = threading.local()


def get_context():
    """The context of the calling thread if one was installed with `use_context`, otherwise the
    process-wide default (one process per GPU).  A context is single-threaded (include/sella_hip.h);
    several host threads may drive the same GPU through one context each."""
    global _default
    ctx = getattr(_tls, 'ctx', None)
    if ctx is not None:
        return ctx
    if _default is None:
        _default = Context()
    return _default


def use_context(ctx):
    """Install `ctx` as the calling thread's context (None removes it)."""
    _tls.ctx = ctx


def _reset_default_context():
    global _default
    if _default is not None:
        _default.close()
    _default = None
