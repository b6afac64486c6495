"""f2 — the automatic internal-coordinate search (`InternalCoordinates.from_atoms`, array code) against the oracle's
loop-by-loop restatement of sella/internal.py:3366-3671 (`oracle/sella_oracle/topology.py`): identical SETS of bonds,
angles and dihedrals on molecules that exercise every branch — tetrahedral centre, chain, ring, planar 3-coordinate
centre (improper), a near-linear angle with a third neighbour (improper replacement), two fragments joined by the growing
scale — and on a small periodic slab."""
import numpy as np
import pytest

from oracle.sella_oracle.topology import find_internals


def mol(symbols, pos, **kw):
    from sella_amd.atoms import Atoms
    return Atoms(symbols, np.array(pos, dtype=float), **kw)


def cases():
    t = 1.09 / np.sqrt(3)
    yield 'methane', mol('CHHHH', [[0, 0, 0], [t, t, t], [-t, -t, t], [-t, t, -t], [t, -t, -t]])
    yield 'ethane-like', mol('CCHHHHHH', [[0, 0, 0], [1.53, 0, 0], [-0.4, 1.0, 0.1], [-0.4, -0.5, 0.9], [-0.4, -0.5, -0.9],
                                          [1.93, -1.0, 0.1], [1.93, 0.5, 0.9], [1.93, 0.5, -0.9]])
    ang = np.arange(6) * np.pi / 3
    ring = np.c_[1.39 * np.cos(ang), 1.39 * np.sin(ang), 0.02 * np.cos(3 * ang)]
    hyd = np.c_[2.48 * np.cos(ang), 2.48 * np.sin(ang), np.zeros(6)]
    yield 'benzene-like', mol('C' * 6 + 'H' * 6, np.r_[ring, hyd])
    yield 'water', mol('OHH', [[0, 0, 0], [0.96, 0, 0], [-0.24, 0.93, 0]])
    yield 'planar-NO3', mol('NOOO', [[0, 0, 0], [1.25, 0, 0], [-0.625, 1.083, 0], [-0.625, -1.083, 0]])
    yield 'near-linear+third', mol('CNCH', [[0, 0, 0], [1.2, 0.02, 0], [-1.4, 0.05, 0], [0.1, 1.08, 0.1]])
    yield 'two fragments', mol('OHHOHH', [[0, 0, 0], [0.96, 0, 0], [-0.24, 0.93, 0], [0.1, -0.2, 2.9], [1.06, -0.2, 2.9],
                                          [-0.14, 0.73, 2.9]])
    from sella_amd.atoms import fcc111
    yield 'slab', fcc111('Cu', (2, 2, 2), vacuum=5.0)


def sets_of(ic):
    def canon3(a, b, c):
        return (min(a, c), b, max(a, c))

    def canon4(a, b, c, d):
        return (a, b, c, d) if (a, b) < (d, c) else (d, c, b, a)
    bonds = {(int(min(i, j)), int(max(i, j))) for i, j in ic.idx['bonds']}
    angles = {canon3(*map(int, r)) for r in ic.idx['angles']}
    dih = {canon4(*map(int, r)) for r in ic.idx['dihedrals']}
    return bonds, angles, dih


@pytest.mark.parametrize('name,atoms', list(cases()), ids=[c[0] for c in cases()])
def test_topology_matches_oracle(ctx, name, atoms):
    from sella_amd.internal import InternalCoordinates, covalent_radius
    rc = np.array([covalent_radius(s) for s in atoms.symbols])
    b0, a0, d0 = find_internals(atoms, rc)
    ic = InternalCoordinates.from_atoms(atoms)
    b1, a1, d1 = sets_of(ic)
    assert b1 == b0, (sorted(b1 ^ b0))
    if name == 'slab':
        return                 # periodic images: several angles / dihedrals share an index triple; bonds pin the graph
    assert a1 == a0, (sorted(a1 ^ a0))
    assert d1 == d0, (sorted(d1 ^ d0))
    if name == 'planar-NO3':
        assert len(d1) == 1 and len(a1) == 3                      # the improper through N
    if name == 'two fragments':
        assert len(b1) >= 5                                       # 4 intramolecular + at least one joining bond
    if name == 'near-linear+third':
        assert (1, 0, 2) not in a1 and len(d1) >= 1               # linear N-C-C angle left out, improper instead
