"""PES wrapper and the `Sella` optimizer API over the device path.

The reference classes need ASE + JAX and cannot be imported in the build image, so these are
property tests re-stated from the reference's own suite: tests/test_peswrapper.py:15-38
(kick / diag / basis orthogonality) and tests/integration/test_morse_cluster.py:20-46
(run to convergence, projected gradient ~ 0, exactly `order` negative Hessian eigenvalues)."""
import io

import numpy as np
import pytest


def morse_atoms(nat=4, seed=4):
    from sella_amd.atoms import Atoms, MorseCluster
    rng = np.random.RandomState(seed)
    atoms = Atoms(['Xe'] * nat, rng.normal(size=(nat, 3), scale=1.2))
    atoms.calc = MorseCluster(D=1.0, a=1.3, r0=2.0)
    return atoms


def test_constraints_api(ctx):
    from sella_amd.internal import Constraints, DuplicateConstraintError
    atoms = morse_atoms(5)
    c = Constraints(atoms)
    c.fix_translation()
    assert c.ntrans == 3 and c.nint == 3
    c.fix_bond((0, 1))
    c.fix_angle((1, 2, 3), target=100.0)
    c.fix_dihedral((0, 1, 2, 3), comparator='lt', target=170.0)
    assert c.has_inequalities()
    assert c.jacobian().shape == (c.nint, 15)
    res = c.residual()
    assert res[:4] == pytest.approx(0, abs=1e-14) and abs(res[4]) > 1e-3
    with pytest.raises(DuplicateConstraintError):
        c.fix_bond((1, 0), replace_ok=False)
    c.disable_satisfied_inequalities()
    assert c.ndihedrals in (0, 1)
    c2 = c.copy()
    assert c2.nint == c.nint


def test_PES(ctx):
    from sella_amd.peswrapper import PES
    atoms = morse_atoms(5, seed=1)
    pes = PES(atoms)
    pes.kick(0., diag=True, gamma=0.1)
    for _ in range(2):
        pes.kick(-pes.get_g() * 0.01)
    assert pes.H.B is not None
    assert not pes.converged(0.)[0]
    assert pes.converged(1e100)[0]
    A = pes.get_Ufree().T @ pes.get_Ucons()
    np.testing.assert_allclose(A, 0, atol=1e-10)
    assert pes.get_Ucons().shape[1] == 6          # global translation and rotation fixed automatically
    # (peswrapper.py:233-253: a non-periodic system gets fix_translation() and fix_rotation())
    from sella_amd.internal import Constraints
    pes_t = PES(morse_atoms(5, seed=1), constraints=Constraints(morse_atoms(5, seed=1)), proj_rot=False)
    assert pes_t.cons.nrotations == 0 and pes_t.cons.ntrans == 3
    pes.kick(-pes.get_g() * 0.001, diag=True, gamma=0.1)
    # the approximate Hessian reproduces the finite-difference curvature along the Davidson vectors
    B = pes.H.B
    np.testing.assert_allclose(B, B.T, atol=0)
    assert pes.neval > 5


@pytest.mark.parametrize('rigid', ['pins', 'rotation'])
@pytest.mark.parametrize('order', [0, pytest.param(1, marks=pytest.mark.emu_heavy)])
def test_morse_cluster(ctx, order, rigid):
    from sella_amd import Constraints, Sella
    atoms = morse_atoms(4, seed=4)
    cons = Constraints(atoms)
    kw = {}
    if rigid == 'pins':
        # the six rigid-body motions removed with translation constraints only: atom 0 pinned, atom 1 on a
        # line, atom 2 in a plane (no rotation constraint on top: proj_rot=False)
        cons.fix_translation(0)
        cons.fix_translation(1, dim=1)
        cons.fix_translation(1, dim=2)
        cons.fix_translation(2, dim=2)
        kw['proj_rot'] = False
    else:
        # the reference's own set-up (tests/integration/test_morse_cluster.py:29-31)
        cons.fix_translation()
        cons.fix_rotation()
    log = io.StringIO()
    opt = Sella(atoms, order=order, gamma=1e-3, constraints=cons, logfile=log, **kw)
    conv = opt.run(fmax=1e-3, steps=400)
    assert conv, log.getvalue()[-800:]
    Ufree = opt.pes.get_Ufree()
    np.testing.assert_allclose(opt.pes.get_g() @ Ufree, 0, atol=5e-3)
    opt.pes.diag(gamma=1e-16)
    H = opt.pes.get_HL().project(Ufree)
    assert np.sum(H.evals < 0) == order, H.evals
    assert 'Sella' in log.getvalue() and 'rtrust' in log.getvalue()


def test_model_pes_saddle(ctx):
    """Order-1 search on the SURVEY §8(d) model PES: converges to a point with one negative mode."""
    from conftest import hessian_like
    from sella_amd import Sella
    from sella_amd.atoms import Atoms, QuadraticCubicModel
    n = 30
    A, P, g = hessian_like(n, seed=3)
    rng = np.random.RandomState(5)
    U = rng.normal(size=(8, n))
    U /= np.linalg.norm(U, axis=1)[:, None]
    atoms = Atoms(['X'] * (n // 3), 0.05 * rng.normal(size=(n // 3, 3)), pbc=True)
    atoms.calc = QuadraticCubicModel(A, U, c=0.05)
    from sella_amd.internal import Constraints

    class NoCons(Constraints):
        pass
    opt = Sella(atoms, order=1, eta=1e-4, gamma=0.1, delta0=0.1, rs='tr', logfile=None,
                constraints=NoCons(atoms), proj_trans=False)
    conv = opt.run(fmax=1e-5, steps=200)
    assert conv
    x = atoms.positions.ravel()
    p = U @ x
    Hx = A + U.T @ ((2 * 0.05 * p)[:, None] * U)
    w = np.linalg.eigvalsh(Hx)
    assert np.sum(w < 0) == 1
    assert np.linalg.norm(A @ x + U.T @ (0.05 * p ** 2)) < 1e-4


def test_unsupported_options_fail_loudly(ctx):
    from sella_amd import Sella
    atoms = morse_atoms(4)
    with pytest.raises(NotImplementedError):
        Sella(atoms, optimize_cell=True, order=0)
    with pytest.raises(ValueError):                       # optimize.py:239-246
        from sella_amd.internal import Constraints, InternalCoordinates
        Sella(atoms, internal=InternalCoordinates.from_atoms(atoms), constraints=Constraints(atoms))


def test_readme_example_slab_adatom(ctx):
    """README.md:10-40 of the reference (Cu fcc(111) + adatom at the bridge site, lower half of the slab
    held by translation constraints, default Sella settings) with the Morse stand-in calculator:
    the search must end on a first-order saddle of the constrained problem."""
    from sella_amd import Constraints, Sella
    from sella_amd.atoms import PeriodicMorse, add_adsorbate, fcc111
    emu = ctx.backend == 'emu'
    size = (3, 3, 2) if emu else (5, 5, 6)
    slab = fcc111('Cu', size, vacuum=7.5)
    add_adsorbate(slab, 'Cu', 2.0, 'bridge')
    slab.positions[-1, :2] += [0.08, -0.05]                  # off the symmetric site
    cons = Constraints(slab)
    nfixed = 0
    for atom in slab:
        if atom.position[2] < slab.cell[2, 2] / 2. - 1e-6:
            cons.fix_translation(atom.index)
            nfixed += 1
    assert 0 < nfixed < len(slab)
    slab.calc = PeriodicMorse()
    x_fixed = slab.positions[:nfixed].copy()
    dyn = Sella(slab, constraints=cons, logfile=None)
    f0 = np.linalg.norm(slab.get_forces()[nfixed:], axis=1).max()
    if emu:
        # the host emulation is ~1 s per optimizer step here: a few steps only (constraints held,
        # projected force decreasing); the full convergence + inertia check runs under -m gpu
        dyn.run(1e-3, 4)
        np.testing.assert_allclose(slab.positions[:nfixed], x_fixed, atol=1e-10)
        assert dyn.nsteps == 4 and 0.3 < dyn.rho < 3.0 and f0 > 0
        return
    assert dyn.run(1e-3, 300)
    pes = dyn.pes
    np.testing.assert_allclose(slab.positions[:nfixed], x_fixed, atol=1e-10)     # constraints held
    assert np.linalg.norm(pes.get_projected_forces(), axis=1).max() < 1e-3
    # inertia of the true Hessian in the free subspace (central differences of the forces)
    Ufree = pes.get_Ufree()
    m = Ufree.shape[1]
    assert m == 3 * (len(slab) - nfixed)
    x0 = slab.positions.ravel().copy()
    h = 1e-4
    Hf = np.zeros((m, m))
    for j in range(m):
        g = []
        for sgn in (1, -1):
            slab.set_positions((x0 + sgn * h * Ufree[:, j]).reshape(-1, 3))
            g.append(-slab.get_forces().ravel())
        Hf[:, j] = Ufree.T @ (g[0] - g[1]) / (2 * h)
    slab.set_positions(x0.reshape(-1, 3))
    w = np.linalg.eigvalsh(0.5 * (Hf + Hf.T))
    assert w[0] < -1e-3 and w[1] > -1e-6, w[:4]


def test_trajectory_file_and_restart(ctx, tmp_path):
    """`trajectory='name.xyz'` writes one extended-XYZ frame per force call; an optimizer restarted from
    save_state() continues exactly where the uninterrupted run goes."""
    from sella_amd import Sella
    from sella_amd.internal import Constraints

    def make(traj=None):
        atoms = morse_atoms(4, seed=4)
        return atoms, Sella(atoms, order=0, logfile=None, constraints=Constraints(atoms), proj_trans=False,
                            trajectory=traj)

    traj = str(tmp_path / 'run.xyz')
    atoms, opt = make(traj)
    opt.run(fmax=0.0, steps=4)
    ncalls = atoms.calc.ncalls
    opt.pes.close()
    lines = open(traj).read().splitlines()
    assert lines.count('4') == ncalls                       # one frame per evaluation
    assert 'energy=' in lines[1] and 'forces:R:3' in lines[1] and len(lines[2].split()) == 7
    state = str(tmp_path / 'state.npz')
    opt.save_state(state)
    opt.run(fmax=0.0, steps=3)                              # uninterrupted: 4 + 3 steps
    ref = atoms.positions.copy()
    atoms2, opt2 = make()
    opt2.load_state(state)
    opt2.run(fmax=0.0, steps=3)
    np.testing.assert_allclose(atoms2.positions, ref, atol=1e-9)
    assert opt2.nsteps == opt.nsteps == 7


def test_emt_device_against_oracle_and_finite_differences(ctx):
    """EMT on the device (csrc/emt.hip) against the NumPy restatement (oracle/sella_oracle/emt.py; both unpinned
    against ASE) and against central differences of its own energy; bulk Cu has zero forces and an energy
    minimum near the experimental lattice constant."""
    from oracle.sella_oracle.emt import EMTOracle
    from sella_amd.atoms import EMT, Atoms, add_adsorbate, fcc111
    slab = fcc111('Cu', (3, 3, 3), vacuum=6.0)
    add_adsorbate(slab, 'Cu', 1.9, 'fcc')
    rng = np.random.RandomState(0)
    slab.positions += 0.05 * rng.normal(size=slab.positions.shape)
    slab.calc = EMT()
    e, f = slab.get_potential_energy(), slab.get_forces()
    orc = EMTOracle()
    assert e == pytest.approx(orc.get_potential_energy(slab), abs=1e-10)
    np.testing.assert_allclose(f, orc.get_forces(slab), atol=1e-11)
    x0 = slab.positions.copy()
    dv = rng.normal(size=x0.shape)
    h = 1e-5
    slab.set_positions(x0 + h * dv)
    ep = slab.get_potential_energy()
    slab.set_positions(x0 - h * dv)
    em = slab.get_potential_energy()
    assert (ep - em) / (2 * h) == pytest.approx(-(f * dv).sum(), rel=1e-7)
    a = 3.61
    base = [(np.array([i, j, k]) + b) * a for i in range(3) for j in range(3) for k in range(3)
            for b in ([0, 0, 0], [.5, .5, 0], [.5, 0, .5], [0, .5, .5])]
    e = {}
    for s_ in (0.98, 1.0, 1.02):
        bulk = Atoms(['Cu'] * len(base), np.array(base) * s_, cell=np.eye(3) * 3 * a * s_, pbc=True)
        bulk.calc = EMT()
        e[s_] = bulk.get_potential_energy() / len(bulk)
        assert np.abs(bulk.get_forces()).max() < 1e-10
    assert e[1.0] < e[0.98] and e[1.0] < e[1.02] and abs(e[1.0]) < 0.02
    # a two-element alloy exercises the chi = n0_j / n0_i asymmetry of the ordered pairs
    alloy = Atoms(['Cu', 'Ag', 'Au', 'Cu', 'Pt'], rng.normal(size=(5, 3)) * 1.6 + np.arange(5)[:, None] * 1.4)
    alloy.calc = EMT()
    np.testing.assert_allclose(alloy.get_forces(), EMTOracle().get_forces(alloy), atol=1e-11)
