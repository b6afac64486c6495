"""Small molecules end to end on the backend under test: internal coordinates, the two PES wrappers and whole
searches, each checked against a number computed independently on the host (closed forms, central differences of
the calculator, dense LAPACK) — not against "is not None".

ASE is not available here, so the geometries are written out (Angstrom) and the calculator is a pairwise Morse
potential; the checks are this package's own.  (The reference's smoke-level counterparts live in its
tests/test_core_functionality.py and tests/test_peswrapper.py; the semantic rules asserted below are
sella/peswrapper.py:51-69 (constraint / free split), :508-556 (diag), :578-602 (kick) and
sella/internal.py:58-80 (coordinate definitions).)"""
import numpy as np
import pytest

from sella_amd import Sella
from sella_amd.atoms import Atoms, MorseCluster
from sella_amd.internal import InternalCoordinates
from sella_amd.peswrapper import PES, InternalPES

MOLECULES = {
    'H2O': (['O', 'H', 'H'], [[0.0, 0.0, 0.1193], [0.0, 0.7632, -0.4770], [0.0, -0.7632, -0.4770]]),
    'CH4': (['C', 'H', 'H', 'H', 'H'], [[0.0, 0.0, 0.0], [0.6291, 0.6291, 0.6291], [-0.6291, -0.6291, 0.6291],
                                        [0.6291, -0.6291, -0.6291], [-0.6291, 0.6291, -0.6291]]),
    'N2': (['N', 'N'], [[0.0, 0.0, 0.5649], [0.0, 0.0, -0.5649]]),
    'C6H6': (['C'] * 6 + ['H'] * 6,
             [[1.397 * np.cos(k * np.pi / 3), 1.397 * np.sin(k * np.pi / 3), 0.0] for k in range(6)] +
             [[2.481 * np.cos(k * np.pi / 3), 2.481 * np.sin(k * np.pi / 3), 0.0] for k in range(6)]),
}
MORSE = dict(D=1.2, a=1.6, r0=1.05)


def molecule(name, calc=True, jiggle=0.0, seed=0):
    symbols, positions = MOLECULES[name]
    pos = np.array(positions, dtype=float)
    if jiggle:
        pos = pos + jiggle * np.random.RandomState(seed).normal(size=pos.shape)
    atoms = Atoms(symbols, pos, pbc=False)
    if calc:
        atoms.calc = MorseCluster(**MORSE)
    return atoms


def morse_energy(x):
    """The calculator again, from its definition: sum over pairs of D (1 - exp(-a (r - r0)))^2 - D."""
    pos = x.reshape(-1, 3)
    e = 0.0
    for i in range(len(pos)):
        for j in range(i + 1, len(pos)):
            r = np.linalg.norm(pos[i] - pos[j])
            e += MORSE['D'] * ((1.0 - np.exp(-MORSE['a'] * (r - MORSE['r0']))) ** 2 - 1.0)
    return e


def fd_gradient(x, h=1e-6):
    g = np.zeros_like(x)
    for i in range(x.size):
        d = np.zeros_like(x)
        d[i] = h
        g[i] = (morse_energy(x + d) - morse_energy(x - d)) / (2 * h)
    return g


def fd_hessian(x, h=1e-5):
    H = np.zeros((x.size, x.size))
    for i in range(x.size):
        d = np.zeros_like(x)
        d[i] = h
        H[:, i] = (fd_gradient(x + d, 1e-6) - fd_gradient(x - d, 1e-6)) / (2 * h)
    return 0.5 * (H + H.T)


@pytest.fixture(autouse=True)
def _device(ctx):
    """Every test here runs on the backend under test (the host emulation, or the GPU under -m gpu)."""
    yield


# ---- internal coordinates ---------------------------------------------------------------------------------------
def test_water_internal_values_are_the_textbook_ones():
    atoms = molecule('H2O', calc=False)
    ic = InternalCoordinates.from_atoms(atoms, dihedrals=False)
    q = np.sort(np.asarray(ic.calc()))
    p = atoms.positions
    r1, r2 = np.linalg.norm(p[1] - p[0]), np.linalg.norm(p[2] - p[0])
    ang = np.arccos(np.dot(p[1] - p[0], p[2] - p[0]) / (r1 * r2))
    np.testing.assert_allclose(q, np.sort([r1, r2, ang]), atol=1e-13)


@pytest.mark.parametrize('name', ['H2O', 'CH4', 'C6H6'])
def test_jacobian_and_hessians_against_central_differences(name):
    atoms = molecule(name, calc=False, jiggle=0.03, seed=2)
    ic = InternalCoordinates.from_atoms(atoms, dihedrals=(name == 'C6H6'))
    q0 = np.asarray(ic.calc())
    B = np.asarray(ic.jacobian())
    n = 3 * len(atoms)
    assert B.shape == (len(q0), n) and np.all(np.isfinite(B))
    Hs = np.asarray(ic.hessian())
    assert Hs.shape == (len(q0), n, n)
    x0 = atoms.positions.copy()
    h = 1e-5
    Bfd = np.zeros_like(B)
    Hfd = np.zeros_like(Hs)
    for e in range(n):
        d = np.zeros(n)
        d[e] = h
        atoms.positions = x0 + d.reshape(-1, 3)
        qp, Bp = np.asarray(ic.calc()), np.asarray(ic.jacobian())
        atoms.positions = x0 - d.reshape(-1, 3)
        qm, Bm = np.asarray(ic.calc()), np.asarray(ic.jacobian())
        dq = (qp - qm + np.pi) % (2 * np.pi) - np.pi if name == 'C6H6' else qp - qm
        Bfd[:, e] = dq / (2 * h)
        Hfd[:, :, e] = (Bp - Bm) / (2 * h)
    atoms.positions = x0
    np.testing.assert_allclose(B, Bfd, atol=1e-7, rtol=1e-7)          # the reference's own tolerance for this property
    np.testing.assert_allclose(Hs, Hfd, atol=2e-6, rtol=1e-6)
    np.testing.assert_allclose(Hs, Hs.transpose(0, 2, 1), atol=1e-12)
    # bonds, angles and dihedrals do not change under a rigid translation: B annihilates the three translations
    T = np.tile(np.eye(3), (len(atoms), 1))
    np.testing.assert_allclose(B @ T, 0.0, atol=1e-12)


# ---- PES wrappers ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', ['H2O', 'CH4'])
def test_cartesian_pes_gradient_and_converged_curvature(name):
    atoms = molecule(name, jiggle=0.05, seed=1)
    x = atoms.positions.ravel().copy()
    pes = PES(atoms)
    pes.kick(0., diag=True, gamma=1e-8)                 # tight: Davidson runs to the lowest eigenpair
    np.testing.assert_allclose(pes.get_g(), fd_gradient(x), atol=1e-7)
    assert abs(pes.get_f() - morse_energy(x)) < 1e-12
    U = pes.get_Ufree()
    Uc = pes.get_Ucons()
    np.testing.assert_allclose(U.T @ Uc, 0.0, atol=1e-10)             # peswrapper.py:51-69
    np.testing.assert_allclose(U.T @ U, np.eye(U.shape[1]), atol=1e-10)
    # the approximate Hessian reproduces the lowest curvature of the true (finite-difference) Hessian in the free space
    Hfree = U.T @ fd_hessian(x) @ U
    w = np.linalg.eigvalsh(Hfree)
    Bfree = U.T @ pes.get_H().asarray() @ U
    wb = np.linalg.eigvalsh(0.5 * (Bfree + Bfree.T))
    # (one-sided differences with eta = 1e-4 behind the operator: O(eta) x third derivative)
    assert abs(wb[0] - w[0]) < 5e-3 * max(1.0, abs(w[0])), (wb[0], w[0])


def test_internal_pes_after_diag_and_a_step():
    """The call sequence that ended round 3's GPU run: InternalPES.kick(diag=True) drives sella_davidson through a host
    callback that re-enters the library.  Checked here by value: basis split, gradient transformation, geodesic step."""
    atoms = molecule('H2O', jiggle=0.04, seed=5)
    x = atoms.positions.ravel().copy()
    ic = InternalCoordinates.from_atoms(atoms, dihedrals=False)
    pes = InternalPES(atoms, ic)
    pes.kick(0., diag=True, gamma=0.1)
    np.testing.assert_allclose(pes.get_Ufree().T @ pes.get_Ucons(), 0.0, atol=1e-10)
    # g_int = B^+T g_cart (peswrapper.py:1124-1127): B^T g_int is the Cartesian gradient projected on the row space of B
    B = np.asarray(ic.jacobian())
    gc = fd_gradient(x)
    Prow = np.linalg.pinv(B) @ B
    np.testing.assert_allclose(B.T @ pes.get_g(), Prow @ gc, atol=1e-6)
    # a short step in internal coordinates changes q by exactly that step (the geodesic ends where it was told to)
    q0 = np.asarray(ic.calc()).copy()
    dq = -0.02 * pes.get_g()
    pes.kick(dq)
    np.testing.assert_allclose(np.asarray(ic.calc()) - q0, dq, atol=1e-6)
    assert np.all(np.isfinite(pes.get_g()))


@pytest.mark.parametrize('name', ['CH4', 'C6H6'])
def test_both_wrappers_track_energy_and_count_force_calls(name, tmp_path):
    atoms = molecule(name, jiggle=0.02, seed=3)
    traj = str(tmp_path / (name + '.traj'))
    pes = PES(atoms, trajectory=traj)
    pes.kick(0., diag=True, gamma=0.1)
    e0 = pes.get_f()
    for _ in range(2):
        pes.kick(-pes.get_g() * 0.01)
    assert pes.get_f() < e0                                           # two short steepest-descent steps go downhill
    assert abs(pes.get_f() - morse_energy(atoms.positions.ravel())) < 1e-12
    assert not pes.converged(0.)[0] and pes.converged(1e100)[0]
    pes.close()
    from sella_amd.trajectory import Trajectory
    with Trajectory(traj) as images:                                   # one image per force call
        assert len(images) == pes.neval
        np.testing.assert_allclose(images[-1].positions, atoms.positions, atol=1e-12)


# ---- whole searches -------------------------------------------------------------------------------------------
@pytest.mark.parametrize('internal', [False, True])
def test_diatomic_relaxes_to_the_morse_minimum(internal):
    """A linear molecule: one of the three rotation generators vanishes identically; no NaN may come out of the
    rotation constraints, and the bond must end at r0."""
    atoms = molecule('N2')
    opt = Sella(atoms, order=0, internal=internal, logfile=None)
    opt.run(fmax=1e-3, steps=100)
    assert opt.converged()
    r = np.linalg.norm(atoms.positions[0] - atoms.positions[1])
    assert abs(r - MORSE['r0']) < 1e-3
    assert abs(atoms.get_potential_energy() + MORSE['D']) < 1e-5


def test_water_saddle_search_ends_on_a_stationary_point_with_negative_curvature():
    atoms = molecule('H2O', jiggle=0.05, seed=7)
    opt = Sella(atoms, order=1, logfile=None, gamma=1e-3)
    opt.run(fmax=2e-3, steps=200)
    assert opt.converged()
    x = atoms.positions.ravel()
    assert np.abs(fd_gradient(x)).max() < 5e-3
    w = np.linalg.eigvalsh(fd_hessian(x))
    # a saddle, not a minimum (three Morse-bound atoms: the search may end on the collinear arrangement, whose bending
    # pair is degenerate, so the index is not asserted — the approximate Hessian only has to follow ONE mode uphill)
    assert (w < -1e-3).sum() >= 1, w
    Bf = opt.pes.get_H().asarray()
    assert np.linalg.eigvalsh(0.5 * (Bf + Bf.T))[0] < 0
