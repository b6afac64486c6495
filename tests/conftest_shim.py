"""`hessian_like` for worker processes that must not import pytest's conftest machinery."""
import numpy as np


def hessian_like(n, seed, eps=5e-3, nneg=1):
    rng = np.random.RandomState(seed)
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    lam = np.exp(rng.uniform(np.log(0.05), np.log(50.0), n))
    lam[:nneg] = -np.linspace(1.0, 0.5, nneg)
    A = (Q * lam) @ Q.T
    A = 0.5 * (A + A.T)
    N = rng.normal(size=(n, n))
    P = A + eps * 0.5 * (N + N.T)
    g = rng.normal(size=n)
    return A, P, g
