"""The reference's general tests (tests/test_core_functionality.py of zadorlab/sella), restated on this package with the
same class and test names: Hessian arithmetic, internal coordinates, the PES wrappers, the eigensolvers, the
finite-difference Hessian, a linear molecule.  ASE is not available here, so the three molecules are written out
(experimental geometries, Angstrom) and the calculator is a pairwise Morse potential instead of ASE's EMT; the
assertions are the reference's."""
import numpy as np
import pytest

from sella_amd import Sella
from sella_amd.atoms import Atoms, MorseCluster
from sella_amd.eigensolvers import exact, rayleigh_ritz
from sella_amd.internal import InternalCoordinates
from sella_amd.linalg import ApproximateHessian, NumericalHessian
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


def molecule(name, calc=True):
    symbols, positions = MOLECULES[name]
    atoms = Atoms(symbols, np.array(positions, dtype=float), pbc=False)
    if calc:
        atoms.calc = MorseCluster(D=1.2, a=1.6, r0=1.05)
    return atoms


@pytest.fixture(autouse=True)
def _device(ctx):
    """Every test here runs on the backend under test (the host emulation, or the GPU under -m gpu)."""
    yield


class TestApproximateHessian:
    def test_hessian_arithmetic(self):
        dim = ncart = 5
        rng = np.random.RandomState(42)
        H1 = ApproximateHessian(dim, ncart, update_method='BFGS')
        s, y = rng.normal(size=dim), rng.normal(size=dim)
        s /= np.linalg.norm(s)
        y /= np.linalg.norm(y)
        H1.update(s, y)
        assert H1.initialized
        M = rng.normal(size=(dim, dim))
        M = 0.5 * (M + M.T)
        result = H1 + M
        np.testing.assert_allclose(result.B, H1.B + M, atol=1e-10)

    def test_hessian_addition_with_uninitialized(self):
        dim = ncart = 5
        rng = np.random.RandomState(42)
        H1 = ApproximateHessian(dim, ncart, update_method='BFGS')
        H2 = ApproximateHessian(dim, ncart, update_method='BFGS')
        H1.update(rng.normal(size=dim), rng.normal(size=dim))
        assert (H1 + H2) is not None

    def test_eigendecomposition(self):
        dim = ncart = 6
        rng = np.random.RandomState(42)
        H = ApproximateHessian(dim, ncart, update_method='BFGS')
        for _ in range(3):
            H.update(rng.normal(size=dim), rng.normal(size=dim))
        evals, evecs = H.evals, H.evecs
        assert evals is not None and evecs is not None and len(evals) == dim
        np.testing.assert_allclose(H.B, evecs @ np.diag(evals) @ evecs.T, atol=1e-10)


class TestSparseInternalHessians:
    def test_numpy_array_conversion(self):
        atoms = molecule('H2O', calc=False)
        internal = InternalCoordinates.from_atoms(atoms, dihedrals=False)
        arr = np.asarray(internal.hessian())
        assert isinstance(arr, np.ndarray)
        assert arr.shape == (len(internal.calc()), 3 * len(atoms), 3 * len(atoms))


class TestInternals:
    def test_basic_internal_coords(self):
        atoms = molecule('CH4', calc=False)
        internal = InternalCoordinates.from_atoms(atoms, dihedrals=False)
        q = internal.calc()
        assert len(q) > 0 and not np.any(np.isnan(q))
        jac = internal.jacobian()
        assert jac.shape == (len(q), 3 * len(atoms))

    def test_water_molecule(self):
        atoms = molecule('H2O', calc=False)
        internal = InternalCoordinates.from_atoms(atoms, dihedrals=False)
        q, jac = internal.calc(), internal.jacobian()
        assert len(q) >= 3                                   # two bonds and the angle
        assert np.all(np.isfinite(q)) and np.all(np.isfinite(np.asarray(jac)))


class TestPES:
    def test_pes_basic_operations(self):
        atoms = molecule('H2O')
        pes = PES(atoms)
        pes.kick(0., diag=True, gamma=0.1)
        g = pes.get_g()
        assert g is not None and len(g) == 3 * len(atoms)
        assert pes.get_H() is not None

    def test_internal_pes_operations(self):
        atoms = molecule('H2O')
        internal = InternalCoordinates.from_atoms(atoms, dihedrals=False)
        pes = InternalPES(atoms, internal)
        pes.kick(0., diag=True, gamma=0.1)
        np.testing.assert_allclose(pes.get_Ufree().T @ pes.get_Ucons(), 0, atol=1e-10)


class TestEigensolvers:
    def test_exact_eigensolver(self):
        dim = 5
        rng = np.random.RandomState(42)
        A = rng.normal(size=(dim, dim))
        A = 0.5 * (A + A.T)
        lams, vecs, Avecs = exact(A)
        for i in range(dim):
            np.testing.assert_allclose(A @ vecs[:, i], lams[i] * vecs[:, i], atol=1e-10)
        np.testing.assert_allclose(Avecs, lams[np.newaxis, :] * vecs, atol=1e-10)

    def test_rayleigh_ritz_small_gamma(self):
        dim = 6
        rng = np.random.RandomState(42)
        A = rng.normal(size=(dim, dim))
        A = A.T @ A + 0.1 * np.eye(dim)
        lams, vecs, Avecs = rayleigh_ritz(A, 0.1, np.eye(dim), maxiter=20)
        assert len(lams) > 0


class TestNumericalHessian:
    def test_matvec_with_zero_vector(self):
        x0 = np.array([1.0, 2.0, 3.0])
        H = NumericalHessian(lambda x: (0.5 * np.sum(x ** 2), x), x0, x0.copy(), eta=1e-5)
        np.testing.assert_allclose(H @ np.zeros(3), np.zeros(3), atol=1e-14)

    def test_matvec_symmetry(self):
        A = np.array([[2.0, 0.5], [0.5, 3.0]])

        def quadratic_func(x):
            return 0.5 * x @ A @ x, A @ x

        x0 = np.array([1.0, 1.0])
        H = NumericalHessian(quadratic_func, x0, quadratic_func(x0)[1], eta=1e-5, threepoint=True)
        H12 = (H @ np.array([1.0, 0.0]))[1]
        H21 = (H @ np.array([0.0, 1.0]))[0]
        np.testing.assert_allclose(H12, H21, rtol=1e-5)


class TestLinearMolecule:
    """A linear molecule must not produce NaN in the rotation constraints (the reference's regression test for its
    quaternion parameterisation; here the rotation constraint is the linearised one, and one of its three generators
    vanishes identically for a diatomic)."""

    def test_n2_cartesian(self):
        atoms = molecule('N2')
        opt = Sella(atoms, order=0, logfile=None)
        opt.run(fmax=0.01, steps=100)
        assert opt.converged()
        assert abs(np.linalg.norm(atoms.positions[0] - atoms.positions[1]) - 1.05) < 1e-2

    def test_n2_internal(self):
        atoms = molecule('N2')
        opt = Sella(atoms, order=0, internal=True, logfile=None)
        opt.run(fmax=0.01, steps=100)
        assert opt.converged()


@pytest.mark.parametrize("name,traj,cons", [("CH4", "CH4.traj", None),
                                            ("CH4", None, dict(bonds=((0, 1)))),
                                            ("C6H6", None, None)])
def test_PES(name, traj, cons, tmp_path):
    """tests/test_peswrapper.py of the reference, same parameters (its `cons` is never handed to the class either; the
    trajectory name makes the class write an ASE-format file, one image per force call)."""
    tol = dict(atol=1e-6, rtol=1e-6)
    atoms = molecule(name)
    for MyPES in [PES, InternalPES]:
        pes = PES(atoms, trajectory=None if traj is None else str(tmp_path / traj))
        pes.kick(0., diag=True, gamma=0.1)
        for i in range(2):
            pes.kick(-pes.get_g() * 0.01)
        assert pes.H is not None
        assert not pes.converged(0.)[0]
        assert pes.converged(1e100)
        np.testing.assert_allclose(pes.get_Ufree().T @ pes.get_Ucons(), 0, **tol)
        pes.kick(-pes.get_g() * 0.001, diag=True, gamma=0.1)
        pes.close()
    if traj is not None:
        from sella_amd.trajectory import Trajectory
        with Trajectory(str(tmp_path / traj)) as images:               # the second pass overwrote the first one's file
            assert len(images) == pes.neval
            assert images[-1].positions.shape == (len(atoms), 3)
