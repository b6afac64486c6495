"""ORACLE (test infrastructure; never imported by the product): restatement of the automatic internal-coordinate
search of the reference — `Internals.find_all_bonds / find_all_angles / find_all_dihedrals`,
sella/internal.py:3366-3671 — with the reference's own control flow (per-atom loops, flood fill, itertools
combinations), for non-periodic molecules and minimum-image periodic cells, without dummy atoms and fragments
(`allow_fragments=False`).  PARITY UNPINNED: sella/internal.py imports ASE and JAX at module level and cannot be
imported here; this file follows its text.  Returns plain Python sets so that tests can compare coordinate SETS."""
from itertools import combinations

import numpy as np


def _flood(i, adj, labels, label):                                         # :3248-3258 (iterative, no recursion limit)
    stack = [i]
    while stack:
        a = stack.pop()
        for b in adj[a]:
            if labels[b] != label:
                labels[b] = label
                stack.append(b)


def _min_image(atoms, i, j):
    d = atoms.positions[j] - atoms.positions[i]
    shift = np.zeros(3)
    per = np.where(atoms.pbc)[0]
    cell = np.asarray(atoms.cell, dtype=float)
    if len(per):
        s = -np.round(d @ np.linalg.pinv(cell[per]))
        shift[per] = s
        d = d + s @ cell[per]
    return d, shift


def find_all_bonds(atoms, rcov, scale=1.25):                              # :3366-3423
    n = len(atoms)
    bonds = {}                                       # (i, j) with i < j -> ncvec
    first = True
    while True:
        adj = [[] for _ in range(n)]
        for (i, j) in bonds:
            adj[i].append(j)
            adj[j].append(i)
        labels = -np.ones(n, dtype=int)
        nlabels = 0
        for i in range(n):
            if labels[i] == -1:
                labels[i] = nlabels
                _flood(i, adj, labels, nlabels)
                nlabels += 1
        if nlabels == 1:
            break
        for i in range(n):
            for j in range(i + 1, n):
                if labels[i] == labels[j] and not first:
                    continue
                if labels[i] == labels[j] and first and bonds:
                    continue
                d, shift = _min_image(atoms, i, j)
                if np.linalg.norm(d) <= scale * (rcov[i] + rcov[j]) and (i, j) not in bonds:
                    bonds[(i, j)] = shift
        first = False
        scale *= 1.05
        if scale > 1e3:
            raise RuntimeError('atoms cannot be connected')
    return bonds


def _angle(atoms, i, j, k, v1, v2):
    cell = np.asarray(atoms.cell, dtype=float)
    a = atoms.positions[i] - (atoms.positions[j] + np.asarray(v1) @ cell) if False else None
    # vectors from the vertex j: to i (image offset of j relative to i is v1) and to k (offset of k relative to j: v2)
    r1 = atoms.positions[i] - (atoms.positions[j] + np.asarray(v1) @ cell)
    r2 = atoms.positions[k] + np.asarray(v2) @ cell - atoms.positions[j]
    c = r1 @ r2 / np.linalg.norm(r1) / np.linalg.norm(r2)
    return np.arccos(np.clip(c, -1.0, 1.0))


def find_all_angles(atoms, bonds, atol=15.0):                             # :3458-3573
    atol = atol * np.pi / 180.0
    n = len(atoms)
    at = [[] for _ in range(n)]                      # bonds from each centre: (neighbour, offset seen from the centre)
    for (i, j), v in bonds.items():
        at[i].append((j, np.asarray(v)))
        at[j].append((i, -np.asarray(v)))
    angles, impropers = set(), set()
    for j, jb in enumerate(at):
        linear = []
        for (n1, o1), (n2, o2) in combinations(jb, 2):
            val = _angle(atoms, n1, j, n2, -o1, o2)
            if atol < val < np.pi - atol:
                angles.add((min(n1, n2), j, max(n1, n2)) if n1 != n2 else (n1, j, n2))
            else:
                linear.append(((n1, o1), (n2, o2)))
        if linear and len(jb) > 2:
            for (n1, o1), (n2, o2) in linear:
                for (n3, o3) in jb:
                    if (n3 == n1 and np.array_equal(o3, o1)) or (n3 == n2 and np.array_equal(o3, o2)):
                        continue
                    impropers.add((n1, j, n3, n2))
                    break
    return angles, impropers


def find_all_dihedrals(atoms, bonds, angles, impropers):                  # :3575-3660 (non-periodic index logic)
    n = len(atoms)
    dihedrals = set()

    def canon(a, b, c, d):
        return (a, b, c, d) if (a, b) < (d, c) else (d, c, b, a)
    for quad in impropers:
        dihedrals.add(canon(*quad))
    edge = {}
    for (i, j, k) in angles:
        for e in ((min(i, j), max(i, j)), (min(j, k), max(j, k))):
            edge.setdefault(e, []).append((i, j, k))
    for lst in edge.values():
        for a1, a2 in combinations(lst, 2):
            for p in (a1, a1[::-1]):
                for q in (a2, a2[::-1]):
                    if p[1] == q[0] and p[2] == q[1] and p[0] != q[2] and p[0] != q[1] and q[2] != p[1]:
                        dihedrals.add(canon(p[0], p[1], p[2], q[2]))
    centres = set()
    for (a, b, c, d) in dihedrals:
        centres.update((b, c))
    neigh = [[] for _ in range(n)]
    for (i, j) in bonds:
        neigh[i].append(j)
        neigh[j].append(i)
    for c in range(n):
        if len(neigh[c]) in (3, 4) and c not in centres:
            dihedrals.add(canon(neigh[c][0], c, neigh[c][1], neigh[c][2]))
    return dihedrals


def find_internals(atoms, rcov, scale=1.25, atol=15.0):
    bonds = find_all_bonds(atoms, rcov, scale)
    angles, impropers = find_all_angles(atoms, bonds, atol)
    return set(bonds), angles, find_all_dihedrals(atoms, bonds, angles, impropers)
