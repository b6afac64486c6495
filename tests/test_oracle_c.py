"""oracle/mgs_oracle.c (C restatement of sella/utilities/math.pyx:mgs) against the golden
vectors generated from the reference's compiled Cython routine, including its return codes
(tests/utilities/test_math.py:144-165)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO, load_golden


@pytest.fixture(scope='module')
def cmgs():
    subprocess.check_call(['make', '-s', '-C', os.path.join(REPO, 'oracle')])
    lib = ctypes.CDLL(os.path.join(REPO, 'oracle', '_build', 'libmgs_oracle.so'))
    lib.mgs_oracle.restype = ctypes.c_int
    lib.mgs_oracle.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                               ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int]

    def run(X, Y=None, eps1=1e-15, eps2=1e-6, maxiter=100, yrows=None):
        X = np.array(X, dtype=float, order='C')
        n, nx = X.shape
        if Y is not None:
            Y = np.array(Y, dtype=float, order='C')
            ny = lib.mgs_oracle(Y.shape[0], Y.ctypes.data, Y.shape[1], None, 0, 0, eps1, eps2, maxiter)
            Y = np.ascontiguousarray(Y[:, :ny])
            m = lib.mgs_oracle(n, X.ctypes.data, nx, Y.ctypes.data, Y.shape[1],
                               Y.shape[0] if yrows is None else yrows, eps1, eps2, maxiter)
        else:
            m = lib.mgs_oracle(n, X.ctypes.data, nx, None, 0, 0, eps1, eps2, maxiter)
        return m, X
    return run


def test_golden(cmgs, manifest):
    g = load_golden('g3_mgs')
    for case in manifest['g3_mgs']:
        i = case['id']
        Y = g[f'c{i}_Y'] if case['hasY'] else None
        m, X = cmgs(g[f'c{i}_X'], Y)
        ref = g[f'c{i}_out']
        assert m == ref.shape[1], case
        np.testing.assert_allclose(X[:, :m], ref, atol=1e-12)
        assert np.all(X[:, m:] == 0)


def test_return_codes(cmgs):
    rng = np.random.RandomState(0)
    X = rng.normal(size=(8, 3))
    assert cmgs(X, Y=rng.normal(size=(8, 2)), yrows=7)[0] == -1          # shape mismatch
    assert cmgs(X, maxiter=0)[0] == -2                                   # no sweep allowed
    Y = np.linalg.qr(rng.normal(size=(8, 8)))[0]
    assert cmgs(X, Y=Y)[0] == 0                                          # everything dropped
