"""ctypes binding of libsella_hip.so (include/sella_hip.h).

There is deliberately no CPU fallback here: if the shared object is missing or no HIP
device is visible, importing the accelerated path fails loudly.
"""
import ctypes
import os
from ctypes import (POINTER, byref, c_char_p, c_double, c_int, c_long,  # noqa: F401
                    c_void_p)

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libsella_hip.so')

_lib = None

c_double_p = POINTER(c_double)
c_int_p = POINTER(c_int)
MATVEC_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_double_p, c_double_p, c_int)
ALLGATHER_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_void_p, ctypes.c_size_t, c_void_p)

class OptStepArgs(ctypes.Structure):
    """`sella_opt_step_t` of include/sella_hip.h, field for field."""
    _fields_ = [
        ('flags', c_int), ('n', c_int),
        ('B', c_int), ('Wt', c_int), ('r', c_int_p), ('mu', c_void_p), ('lam0', c_double),
        ('update_method', c_int), ('symm', c_int), ('B_stale', c_int), ('Bsub_stale', c_int),
        ('Bsub', c_int), ('Wt_sub', c_int), ('r_sub', c_int_p), ('mu_sub', c_void_p), ('idx', c_void_p), ('m', c_int),
        ('dx', c_void_p), ('g_old', c_void_p), ('g_new', c_void_p),
        ('f_old', c_double), ('f_new', c_double), ('smag', c_double),
        ('delta', c_double), ('rho', c_double),
        ('delta_min', c_double), ('sigma_inc', c_double), ('sigma_dec', c_double), ('rho_inc', c_double),
        ('rho_dec', c_double),
        ('df_pred', c_double), ('ratio', c_double),
        ('ratio_valid', c_int), ('updated', c_int), ('nrank1', c_int), ('nrank1_sub', c_int),
        ('stepper_kind', c_int), ('order', c_int), ('cons', c_int), ('maxiter', c_int),
        ('tol', c_double),
        ('s_out', c_void_p),
        ('smag_out', c_double),
        ('nalpha', c_int),
        ('alpha_hint', c_double),
    ]


class SearchParams(ctypes.Structure):
    """`sella_search_params_t` of include/sella_hip.h."""
    _fields_ = [('order', c_int), ('eig', c_int), ('threepoint', c_int), ('dav_method', c_int), ('stepper_kind', c_int),
                ('cons', c_int), ('update_method', c_int), ('symm', c_int), ('nsteps_per_diag', c_int),
                ('diag_every_n', c_long),
                ('eta', c_double), ('gamma', c_double), ('delta0', c_double), ('delta_min', c_double),
                ('sigma_inc', c_double), ('sigma_dec', c_double), ('rho_inc', c_double), ('rho_dec', c_double)]


# name -> (restype, argtypes); mirrors include/sella_hip.h one to one
SIGNATURES = {
    'sella_last_error': (c_char_p, []),
    'sella_version': (c_char_p, []),
    'sella_device_count': (c_int, [c_int_p]),
    'sella_ctx_create': (c_int, [c_int, POINTER(c_void_p)]),
    'sella_ctx_destroy': (c_int, [c_void_p]),
    'sella_ctx_sync': (c_int, [c_void_p]),
    'sella_ctx_device_name': (c_int, [c_void_p, c_char_p, c_int]),
    'sella_ctx_set_option': (c_int, [c_void_p, c_char_p, c_long]),
    'sella_mat_upload': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int_p]),
    'sella_mat_alloc': (c_int, [c_void_p, c_int, c_int, c_int_p]),
    'sella_mat_set': (c_int, [c_void_p, c_int, c_void_p]),
    'sella_mat_download': (c_int, [c_void_p, c_int, c_void_p]),
    'sella_mat_shape': (c_int, [c_void_p, c_int, c_int_p, c_int_p]),
    'sella_mat_copy': (c_int, [c_void_p, c_int, c_int_p]),
    'sella_mat_transpose': (c_int, [c_void_p, c_int, c_int_p]),
    'sella_mat_rows': (c_int, [c_void_p, c_int, c_int, c_int, c_int_p]),
    'sella_mat_copy_into': (c_int, [c_void_p, c_int, c_int, c_int]),
    'sella_mat_add_diag': (c_int, [c_void_p, c_int, c_double]),
    'sella_mat_free': (c_int, [c_void_p, c_int]),
    'sella_mat_axpby': (c_int, [c_void_p, c_double, c_int, c_double, c_int, c_int_p]),
    'sella_symm_mm': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p]),
    'sella_gemm_tn_host': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p]),
    'sella_project': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p]),
    'sella_project_dev': (c_int, [c_void_p, c_int, c_int, c_int_p]),
    'sella_gemm': (c_int, [c_void_p, c_int, c_int, c_double, c_int, c_int, c_double, c_int]),
    'sella_eigh': (c_int, [c_void_p, c_int, c_void_p, c_int_p, c_int_p]),
    'sella_rank1_eig': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_double, c_void_p, c_void_p]),
    'sella_qr_thin': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'sella_mgs': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_double, c_double,
                          c_int, c_void_p, c_int_p]),
    'sella_davidson': (c_int, [c_void_p, c_int, MATVEC_FN, c_void_p, c_int, c_int, c_void_p,
                               c_double, c_int, c_void_p, c_int, c_double, c_int, c_int, c_void_p,
                               c_double, c_void_p, c_void_p, c_void_p, c_int_p, c_int_p]),
    'sella_davidson_block': (c_int, [c_void_p, c_int, c_int, c_int, c_int, ALLGATHER_FN, c_void_p, c_int, c_int,
                                     c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_double, c_int,
                                     c_void_p, c_void_p, c_void_p, c_int_p, c_int_p, c_int_p]),
    'sella_ctx_stream': (c_int, [c_void_p, POINTER(c_void_p)]),
    'sella_mat_ptr': (c_int, [c_void_p, c_int, POINTER(c_void_p), c_int_p]),
    'sella_dev_copy': (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_size_t, c_int]),
    'sella_update_h': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                               c_int, c_int, c_int]),
    'sella_update_h_eig': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                   c_int, c_int, c_int, c_int, c_int_p]),
    'sella_update_h_eig_view': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                        c_int, c_int, c_int, c_int, c_int_p, c_int, c_int, c_int, c_void_p,
                                        c_void_p, c_int, c_int_p]),
    'sella_update_h_lr': (c_int, [c_void_p, c_int, c_int, c_int_p, c_void_p, c_double, c_void_p, c_void_p, c_int, c_int,
                                  c_int, c_int, c_int_p, c_int, c_int, c_int_p, c_void_p, c_void_p, c_int, c_int_p]),
    'sella_lr_restrict': (c_int, [c_void_p, c_int, c_int, c_void_p, c_double, c_void_p, c_int, c_int, c_int_p,
                                  c_void_p]),
    'sella_symmetrize_y': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'sella_stepper_create': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                                     POINTER(c_void_p)]),
    'sella_stepper_create_lr': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_double, c_void_p, c_int, c_int,
                                        POINTER(c_void_p)]),
    'sella_calc_model_create': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_double, POINTER(c_void_p)]),
    'sella_calc_emt_create': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_double, c_double, c_double, c_double,
                                      POINTER(c_void_p)]),
    'sella_calc_eval': (c_int, [c_void_p, c_void_p, POINTER(c_double), c_void_p]),
    'sella_calc_ncalls': (c_long, [c_void_p]),
    'sella_calc_dim': (c_int, [c_void_p]),
    'sella_calc_destroy': (c_int, [c_void_p]),
    'sella_fd_create': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_double, c_int, c_void_p, c_int, POINTER(c_void_p)]),
    'sella_fd_matvec': (c_int, [c_void_p, c_void_p, c_void_p, c_int]),
    'sella_fd_npairs': (c_int, [c_void_p]),
    'sella_fd_calls': (c_long, [c_void_p]),
    'sella_fd_pairs': (c_int, [c_void_p, c_void_p, c_void_p]),
    'sella_fd_destroy': (c_int, [c_void_p]),
    'sella_search_create': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, POINTER(SearchParams),
                                    POINTER(c_void_p)]),
    'sella_search_seed': (c_int, [c_void_p, c_double, c_void_p]),
    'sella_search_run': (c_int, [c_void_p, c_double, c_long, c_int_p]),
    'sella_search_pending_pairs': (c_int, [c_void_p, c_int_p, c_void_p, c_void_p]),
    'sella_search_state': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'sella_search_release_hessian': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_double)]),
    'sella_search_destroy': (c_int, [c_void_p]),
    'sella_search_ctx': (c_void_p, [c_void_p]),
    'sella_cohort_create': (c_int, [c_void_p, c_int, POINTER(c_void_p)]),
    'sella_cohort_size': (c_int, [c_void_p]),
    'sella_cohort_run_searches': (c_int, [c_void_p, c_void_p, c_int, c_double, c_long, c_void_p, c_void_p]),
    'sella_cohort_member_threads': (c_int, [c_void_p, c_int]),
    'sella_cohort_stats': (c_int, [c_void_p, c_void_p]),
    'sella_cohort_error': (c_char_p, [c_void_p, c_int]),
    'sella_cohort_destroy': (c_int, [c_void_p]),
    'sella_opt_step': (c_int, [c_void_p, POINTER(OptStepArgs)]),
    'sella_lr_materialize': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_double]),
    'sella_stepper_get_s': (c_int, [c_void_p, c_double, c_void_p, c_void_p]),
    'sella_stepper_destroy': (c_int, [c_void_p]),
    'sella_stepper_set_d1hat': (c_int, [c_void_p, c_void_p, c_int]),
    'sella_restricted_step': (c_int, [c_void_p, c_int, c_double, c_void_p, c_void_p, c_void_p, c_double, c_double,
                                      c_double, c_double, c_int, c_int, c_double, c_int, c_void_p, c_int, c_void_p,
                                      c_double_p, c_void_p, c_int_p]),
    'sella_internals_eval': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p]),
    'sella_emt_eval': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_double, c_double, c_double,
                               c_double, c_double_p, c_void_p]),
    'sella_prof_enable': (c_int, [c_void_p, c_int]),
    'sella_prof_reset': (c_int, [c_void_p]),
    'sella_prof_get': (c_int, [c_void_p, c_int, POINTER(c_long), c_double_p, c_double_p, c_double_p]),
}

SELLA_NO_MAT = -1


class SellaHipError(RuntimeError):
    """Raised for any non-zero status of the C ABI (carries sella_last_error())."""

    def __init__(self, status, message):
        super().__init__(f'libsella_hip status {status}: {message}')
        self.status = status


missing_symbols = []


def _declare(lib):
    del missing_symbols[:]
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:           # reported by tests/test_abi.py; using it raises
            missing_symbols.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    return lib


def lib():
    """The loaded library.  Raises if libsella_hip.so has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: build it with `python -m sella_amd.build` '
                '(hipcc --offload-arch=gfx950). sella_amd has no CPU fallback.')
        _lib = _declare(ctypes.CDLL(LIB_PATH))
    return _lib


def _set_library_for_tests(cdll):
    """TEST HOOK: bind a different build of the same sources (tests/hostemu)."""
    global _lib
    _lib = _declare(cdll) if cdll is not None else None


def check(status):
    if status != 0:
        raise SellaHipError(status, lib().sella_last_error().decode(errors='replace'))


def as_f64(a, shape=None):
    """C-contiguous float64 view/copy (inputs are never mutated, cf. _gpu.py:64)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != shape:
        raise ValueError(f'expected shape {shape}, got {a.shape}')
    return a


def ptr(a):
    return a.ctypes.data_as(c_void_p) if a is not None else None
