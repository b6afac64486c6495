"""CPU restatement of the sparse per-coordinate Jacobian / Hessian containers of sella/linalg.py:362-646 —
TEST INFRASTRUCTURE (only tests/ and oracle/make_golden.py import this).

A coordinate touches 2, 3 or 4 atoms: its gradient is a (natoms_i, 3) block, its Hessian a
(natoms_i, 3, natoms_i, 3) block.  The containers scatter them into (nint, 3N) / (3N, 3N) arrays and contract
them with vectors.  Written as plain loops over the coordinates (the reference vectorises with
np.add.at / bincount over size groups: same sums, different order).  PINNED: oracle/make_golden.py runs the
real classes on the same random blocks and asserts agreement (fixture g11_sparse_internal).
"""
import numpy as np


def jacobian_dense(natoms, indices, vals):                        # SparseInternalJacobian.asarray :377-384
    B = np.zeros((len(indices), natoms, 3))
    for i, (idx, v) in enumerate(zip(indices, vals)):
        for a, atom in enumerate(idx):
            B[i, atom] += v[a]                                    # a repeated atom accumulates (np.add.at)
    return B.reshape(len(indices), 3 * natoms)


def jacobian_matvec(natoms, indices, vals, x):                    # _matvec :386-393
    xi = np.asarray(x).reshape(natoms, 3)
    return np.array([sum(xi[atom] @ v[a] for a, atom in enumerate(idx)) for idx, v in zip(indices, vals)])


def jacobian_rmatvec(natoms, indices, vals, y):                   # _rmatvec :395-401
    out = np.zeros((natoms, 3))
    for yi, idx, v in zip(y, indices, vals):
        for a, atom in enumerate(idx):
            out[atom] += yi * v[a]
    return out.ravel()


def hessian_dense(natoms, idx, vals):                             # SparseInternalHessian.asarray :424-446
    H = np.zeros((natoms, 3, natoms, 3))
    for a, ia in enumerate(idx):
        for b, ib in enumerate(idx):
            H[ia, :, ib, :] += vals[a, :, b, :]
    return H.reshape(3 * natoms, 3 * natoms)


def hessian_matvec(natoms, idx, vals, x):                         # _matvec :448-460
    xi = np.asarray(x).reshape(natoms, 3)
    out = np.zeros((natoms, 3))
    for a, ia in enumerate(idx):
        for b, ib in enumerate(idx):
            out[ia] += vals[a, :, b, :] @ xi[ib]
    return out.ravel()


def hessians_ldot(natoms, indices, vals, v):                      # SparseInternalHessians.ldot :601-618
    out = np.zeros((3 * natoms, 3 * natoms))
    for vi, idx, val in zip(v, indices, vals):
        out += vi * hessian_dense(natoms, idx, val)
    return out


def hessians_rdot(natoms, indices, vals, x):                      # rdot :620-640
    return np.array([hessian_matvec(natoms, idx, val, x) for idx, val in zip(indices, vals)])


def hessians_ddot(natoms, indices, vals, u, x):                   # ddot :642-646
    return np.array([np.asarray(u) @ hessian_matvec(natoms, idx, val, x) for idx, val in zip(indices, vals)])
