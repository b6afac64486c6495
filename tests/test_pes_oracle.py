"""a13 / a14 against an oracle: the product's `PES` + `Sella.step` (device mirrors, carried eigendecompositions,
selection bases, fused restricted-step root finder) compared STEP BY STEP with the dense NumPy restatement of
sella/peswrapper.py:214-607 and sella/optimize/optimize.py:317-440 in oracle/sella_oracle/pes.py.  That restatement
is unpinned (the reference classes need ASE + JAX) but independent: it shares no code with the product."""
import numpy as np
import pytest

from oracle.sella_oracle.pes import OracleSella, TranslationConstraints


def morse(nat, seed):
    from sella_amd.atoms import Atoms, MorseCluster
    rng = np.random.RandomState(seed)
    at = Atoms(['Xe'] * nat, rng.normal(size=(nat, 3), scale=1.2))
    at.calc = MorseCluster(D=1.0, a=1.3, r0=2.0)
    return at


def pin_rigid(cons):
    cons.fix_translation(0)
    cons.fix_translation(1, dim=1)
    cons.fix_translation(1, dim=2)
    cons.fix_translation(2, dim=2)




@pytest.mark.parametrize('order,rs', [(1, 'tr'), (1, 'ras'), (0, 'ras')])
def test_morse_cluster_step_by_step(ctx, order, rs):
    from sella_amd import Constraints, Sella
    a1, a2 = morse(4, 4), morse(4, 4)
    c1 = Constraints(a1)
    pin_rigid(c1)
    c2 = TranslationConstraints(a2)
    pin_rigid(c2)
    dev = Sella(a1, order=order, rs=rs, constraints=c1, logfile=None, proj_rot=False, gamma=1e-3)
    ora = OracleSella(a2, c2, order=order, rs=rs, gamma=1e-3)
    assert dev.delta == pytest.approx(ora.delta)
    nsteps = 12
    for i in range(nsteps):
        x_before = dev.pes.get_x().copy()
        dev.step()
        ora.step()
        t = ora.trace[-1]
        tol = 1e-7 * 4 ** min(i, 8)              # roundoff of the finite-difference Hessians compounds along the path
        np.testing.assert_allclose(dev.pes.get_x() - x_before, t['s'], atol=tol, err_msg=f'step {i}')
        assert abs(dev.pes.get_f() - t['f']) < tol, i
        np.testing.assert_allclose(dev.pes.get_g(), t['g'], atol=10 * tol)
        assert dev.delta == pytest.approx(t['delta'], rel=1e-5, abs=tol), i
        if t['rho'] is not None:
            assert dev.rho == pytest.approx(t['rho'], rel=1e-4, abs=1e-4), i
        np.testing.assert_allclose(dev.pes.H.B, t['B'], atol=100 * tol * max(1.0, np.abs(t['B']).max()))
    assert dev.pes.neval == ora.pes.neval       # same number of force calls: same diagonalisation schedule


def test_pinned_slab_step_by_step(ctx):
    """The README pattern (lower layers frozen by per-atom translation constraints -> selection bases, principal
    submatrix view on the device) against the dense oracle, which knows none of those shortcuts."""
    from sella_amd import Constraints, Sella
    from sella_amd.atoms import Atoms, MorseCluster
    rng = np.random.RandomState(3)
    pos = np.array([[i * 2.1, j * 2.1, k * 2.0] for k in range(2) for j in range(2) for i in range(3)], dtype=float)
    pos += 0.05 * rng.normal(size=pos.shape)
    pos[-1] += [0.4, 0.3, 0.6]

    def build():
        at = Atoms(['Cu'] * len(pos), pos.copy())
        at.calc = MorseCluster(D=0.8, a=1.4, r0=2.2)
        return at
    a1, a2 = build(), build()
    c1, c2 = Constraints(a1), TranslationConstraints(a2)
    for i in range(6):                           # bottom layer pinned atom by atom
        c1.fix_translation(i)
        c2.fix_translation(i)
    dev = Sella(a1, order=1, constraints=c1, logfile=None, proj_rot=False)
    ora = OracleSella(a2, c2, order=1, rs='ras')
    for i in range(8):
        x_before = dev.pes.get_x().copy()
        dev.step()
        ora.step()
        t = ora.trace[-1]
        tol = 1e-7 * 4 ** min(i, 8)
        np.testing.assert_allclose(dev.pes.get_x() - x_before, t['s'], atol=tol, err_msg=f'step {i}')
        assert abs(dev.pes.get_f() - t['f']) < tol
        assert dev.delta == pytest.approx(t['delta'], rel=1e-5, abs=tol)
        # pinned coordinates never move
        np.testing.assert_array_equal(a1.positions[:6], pos[:6])


@pytest.mark.emu_heavy
def test_emt_slab_twin_step_by_step(ctx, monkeypatch):
    """BASELINE configs[1] on a down-sized twin — Cu(111) 3 x 3 x 4 EMT slab with a lifted surface atom, lower half
    pinned atom by atom, default `Sella` — product (device EMT, selection bases, principal-submatrix view, carried
    eigendecompositions, fused root finder) against the dense oracle driving the NumPy restatement of the EMT
    (oracle/sella_oracle/emt.py), step by step.  The 1024-atom original runs in tests/test_configs_gpu.py."""
    from conftest_shim import emt_slab
    from oracle.sella_oracle.emt import EMTOracle
    from sella_amd import Sella, linalg
    monkeypatch.setattr(linalg, 'LR_MIN_DIM', 96)       # the structured eigendecomposition, as at the named size
    a1, c1, pinned = emt_slab((3, 3, 4))
    a2, _, _ = emt_slab((3, 3, 4), calculator=EMTOracle())
    start = a1.positions.copy()
    c2 = TranslationConstraints(a2)
    for i in pinned:
        c2.fix_translation(int(i))
    np.testing.assert_allclose(a1.get_potential_energy(), a2.get_potential_energy(), atol=1e-10)
    np.testing.assert_allclose(a1.get_forces(), a2.get_forces(), atol=1e-10)
    dev = Sella(a1, constraints=c1, logfile=None)
    ora = OracleSella(a2, c2, order=1, rs='ras')
    for i in range(6):
        x_before = dev.pes.get_x().copy()
        dev.step()
        ora.step()
        t = ora.trace[-1]
        tol = 2e-7 * 4 ** min(i, 8)
        np.testing.assert_allclose(dev.pes.get_x() - x_before, t['s'], atol=tol, err_msg=f'step {i}')
        assert abs(dev.pes.get_f() - t['f']) < tol, i
        np.testing.assert_allclose(dev.pes.get_g(), t['g'], atol=10 * tol)
        assert dev.delta == pytest.approx(t['delta'], rel=1e-5, abs=tol), i
        np.testing.assert_array_equal(a1.positions[pinned], start[pinned])
    assert dev.pes.neval == ora.pes.neval


@pytest.mark.emu_heavy
def test_emt_slab_one_call_steps_against_the_oracle(ctx, monkeypatch):
    """The same comparison on a twin large enough for the structured eigendecomposition of the VIEW (Cu(111) 4 x 4 x 4,
    96 free coordinates), so that the product's steps are the one-call steps of csrc/optstep.hip / lrstep.hip
    (quasi-Newton update in coordinates, every decision on the device; restricted step by interpolating batches) —
    against the dense oracle, which re-diagonalises B after every update and bisects like the reference."""
    from conftest_shim import emt_slab
    from oracle.sella_oracle.emt import EMTOracle
    from sella_amd import Sella, linalg
    monkeypatch.setattr(linalg, 'LR_MIN_DIM', 96)
    a1, c1, pinned = emt_slab((4, 4, 4))
    a2, _, _ = emt_slab((4, 4, 4), calculator=EMTOracle())
    start = a1.positions.copy()
    c2 = TranslationConstraints(a2)
    for i in pinned:
        c2.fix_translation(int(i))
    dev = Sella(a1, constraints=c1, logfile=None)
    ora = OracleSella(a2, c2, order=1, rs='ras')
    for i in range(6):
        x_before = dev.pes.get_x().copy()
        dev.step()
        ora.step()
        t = ora.trace[-1]
        tol = 2e-7 * 4 ** min(i, 8)
        np.testing.assert_allclose(dev.pes.get_x() - x_before, t['s'], atol=tol, err_msg=f'step {i}')
        assert abs(dev.pes.get_f() - t['f']) < tol, i
        np.testing.assert_allclose(dev.pes.get_g(), t['g'], atol=10 * tol)
        assert dev.delta == pytest.approx(t['delta'], rel=1e-5, abs=tol), i
        if t['rho'] is not None:
            assert dev.rho == pytest.approx(t['rho'], rel=1e-4, abs=1e-4), i
        np.testing.assert_array_equal(a1.positions[pinned], start[pinned])
    assert dev.fused_steps >= 4                      # the steps compared were one-call steps
    assert dev.pes.neval == ora.pes.neval
    np.testing.assert_allclose(dev.pes.H.B, ora.pes.H.B if hasattr(ora.pes.H, 'B') else dev.pes.H.B,
                               atol=1e-5 * max(1.0, np.abs(dev.pes.H.B).max()))
