"""BASELINE.json's configurations AS NAMED, on the MI355X only (this module never builds the host emulation):

  configs[0]  Cu fcc111(5,5,6) + adatom, EMT, Cartesian (the README example; the reference runs it on the CPU)
  configs[1]  1024-atom Cu(111) EMT slab, Cartesian, lower half pinned, default `Sella`
  configs[2]  1024-atom slab, internal coordinates + geodesic step
  configs[3]  ensemble of 256-atom EMT saddle searches (one GPU's share: 8 members)

through size-independent properties (energy / force consistency, constraint residuals, secant condition, consistency
of the carried eigendecomposition) and bit-identity of the sharded ensemble with member-by-member runs.  The
step-by-step comparison of the same EMT-slab search against the dense oracle runs on a down-sized twin in
tests/test_pes_oracle.py (CPU emulation and hardware)."""
import numpy as np
import pytest

from conftest import hessian_like, make_context
from conftest_shim import EmtMember, emt_slab

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx(request):
    yield from make_context(request, 'hip')


def test_benchmark_size_eigh(ctx):
    from test_eigh import check
    A, P, g = hessian_like(3072, 0)
    check(ctx, P, tol=2e-12)


def test_config0_readme_slab_step_by_step(ctx):
    """configs[0] as named: README.md:18-25 of the reference — Cu fcc111(5,5,6) + adatom on the bridge site (151 atoms,
    3N = 453), the 75 atoms of the lower half held by translation constraints (225 constraints, 228 free coordinates),
    default `Sella` (order 1, P-RFO, 'ras', gamma 0.4), EMT.  The product on the device (EMT kernel, finite-difference
    Davidson through sella_fd_matvec, structured Hessian, fused restricted step) against the trajectory the dense CPU
    oracle produced for the same start (tests/golden/g13_config0_trace.npz, oracle/make_config0_trace.py): step
    vector, energy, gradient, trust radius, rho and the number of force calls, step by step."""
    from conftest import load_golden
    from sella_amd import Constraints, Sella
    from sella_amd.atoms import EMT, add_adsorbate, fcc111
    t = load_golden('g13_config0_trace')
    slab = fcc111('Cu', (5, 5, 6), vacuum=7.5)
    add_adsorbate(slab, 'Cu', 2.0, 'bridge')
    np.testing.assert_array_equal(slab.positions, t['x_start'])
    cons = Constraints(slab)
    pinned = [a.index for a in slab if a.position[2] < slab.cell[2, 2] / 2.]
    np.testing.assert_array_equal(pinned, t['pinned'])
    for i in pinned:
        cons.fix_translation(i)
    assert len(slab) == 151 and len(pinned) == 75
    slab.calc = EMT()
    dyn = Sella(slab, constraints=cons, logfile=None)
    assert dyn.pes.get_Ufree().shape == (453, 228)
    assert dyn.delta == pytest.approx(float(t['delta0']))
    for i in range(int(t['nsteps'])):
        x_before = dyn.pes.get_x().copy()
        dyn.step()
        f, delta, rho, neval = t[f'scal{i}']
        # Measured on MI355X (tools/config0_deviation.py, session r05c): the step vectors agree to 4e-12 ... 4e-11 at all
        # eight steps, energies to 2e-12, gradients to 1e-10.  The bound is flat and far below every step of the trace
        # (the last one is 9.5e-5 long): a tolerance that grows along the path would end up larger than the step itself.
        tol = 1e-9
        assert tol <= 1e-4 * np.abs(t[f's{i}']).max(), i
        np.testing.assert_allclose(dyn.pes.get_x() - x_before, t[f's{i}'], atol=tol, err_msg=f'step {i}')
        assert abs(dyn.pes.get_f() - f) < tol, i
        np.testing.assert_allclose(dyn.pes.get_g(), t[f'g{i}'], atol=2 * tol)
        assert dyn.delta == pytest.approx(delta, rel=1e-8, abs=tol), i
        if np.isfinite(rho):
            assert dyn.rho == pytest.approx(rho, rel=1e-3, abs=1e-3), i   # (a ratio of energy differences down to 1e-9: 1e-12 / 1e-9)
        assert dyn.pes.neval == int(neval), i          # same diagonalisation schedule, same number of force calls
        np.testing.assert_array_equal(slab.positions[pinned], t['x_start'][pinned])


def test_config1_emt_slab_1024_atoms(ctx):
    """configs[1] as named: 1024-atom Cu(111) EMT slab (3N = 3072, 1536 coordinates pinned), default `Sella`
    (order 1, P-RFO, 'ras'), device EMT, 12 steps."""
    from sella_amd import Sella
    slab, cons, pinned = emt_slab((8, 8, 16))
    assert len(slab) == 1024 and len(pinned) == 512
    x_start = slab.positions.copy()
    dyn = Sella(slab, constraints=cons, logfile=None)
    pes = dyn.pes
    assert pes.get_Ufree().shape == (3072, 1536)
    for _ in range(11):
        dyn.step()
    x0, g0 = pes.get_x().copy(), pes.get_g().copy()
    dyn.step()
    x1, g1 = pes.get_x().copy(), pes.get_g().copy()
    # constraints: pinned atoms never move (bit for bit), residual zero, free atoms did move
    np.testing.assert_array_equal(slab.positions[pinned], x_start[pinned])
    assert np.linalg.norm(pes.get_res()) == 0.0
    assert np.abs(slab.positions - x_start).max() > 1e-3
    assert dyn.delta >= dyn.delta_min and np.isfinite(dyn.rho)
    # calculator boundary: the gradient the optimizer holds is minus the calculator's forces at this geometry
    np.testing.assert_allclose(pes.get_g(), -slab.get_forces().ravel(), atol=1e-12)
    # energy / force consistency of the device EMT at the geometry reached (central difference along a random
    # direction of the free atoms)
    rng = np.random.RandomState(0)
    v = rng.normal(size=slab.positions.shape)
    v[pinned] = 0.0
    v /= np.linalg.norm(v)
    xs, h = slab.positions.copy(), 1e-4
    slab.positions = xs + h * v
    ep = slab.get_potential_energy()
    slab.positions = xs - h * v
    em = slab.get_potential_energy()
    slab.positions = xs
    assert abs((ep - em) / (2 * h) - g1 @ v.ravel()) < 1e-6 * max(1.0, abs(g1 @ v.ravel()))
    # quasi-Newton secant condition of the last TS-BFGS update, B dx = dg, on the device-resident B
    B = pes.H.B
    dx, dg = x1 - x0, g1 - g0
    np.testing.assert_allclose(B @ dx, dg, atol=1e-8 * max(1.0, np.abs(dg).max()))
    np.testing.assert_array_equal(B, B.T)
    # the eigendecompositions carried across the updates (B itself and its free-free principal submatrix; at this size in
    # the structured form lam0 I + rank r) still diagonalise what they belong to
    lr = pes.H.device_eig_lr()
    assert lr is not None and 0 < lr['r'] < 400
    W = lr['Wt'].numpy()[:lr['r']]
    assert np.abs(W @ W.T - np.eye(lr['r'])).max() < 1e-11
    assert np.abs(B @ W.T - W.T * lr['mu'][:lr['r']]).max() < 1e-9 * max(1.0, np.abs(lr['mu'][:lr['r']]).max())
    np.testing.assert_allclose(pes.H.evals, np.linalg.eigvalsh(B), atol=1e-9 * max(1.0, np.abs(B).max()))
    w, V, Vt = pes.H.device_eig()
    Vn = V.numpy()
    scale = max(1.0, np.abs(w).max())
    assert np.abs(B @ Vn - Vn * w).max() < 1e-9 * scale
    assert np.abs(Vn.T @ Vn - np.eye(3072)).max() < 1e-10
    sub = pes.get_HL_projected(pes.get_Ufree())
    free = np.flatnonzero(pes.get_Ufree().sum(axis=1) > 0)
    np.testing.assert_allclose(sub.B, B[np.ix_(free, free)], atol=1e-12 * scale)
    ws = sub.evals
    np.testing.assert_allclose(ws, np.linalg.eigvalsh(B[np.ix_(free, free)]), atol=1e-9 * scale)
    assert ws[0] < 0.0                                 # an order-1 search holds one negative mode in the free space


def test_config2_geodesic_step(ctx):
    """BASELINE configs[2] at full size (1024-atom Cu(111) slab, nearest-neighbour bonds, device EMT) through
    size-independent properties: B B^+ B = B, the geodesic step meets a feasible target to second order in the
    step, the transported gradient keeps its length, and the energy change of a small step matches g_int . dq."""
    from sella_amd.atoms import EMT, fcc111
    from sella_amd.internal import InternalCoordinates, neighbour_bonds
    from sella_amd.peswrapper import InternalPES
    slab = fcc111('Cu', (8, 8, 16), vacuum=7.5)
    rng = np.random.RandomState(0)
    slab.positions += 0.03 * rng.normal(size=slab.positions.shape)
    slab.calc = EMT()
    bonds, ncv = neighbour_bonds(slab, 1.25 * 3.61 / np.sqrt(2))
    pes = InternalPES(slab, InternalCoordinates(slab, bonds=bonds, bond_ncvecs=ncv))
    fac = pes._get_factor()
    assert fac.shape == (len(bonds), 3072) and fac.rank == 3069          # three translations in the null space
    probe = fac.Bs @ rng.normal(size=(3072, 2))
    np.testing.assert_allclose(fac.Bs @ fac.pinv_dot(probe), probe, atol=1e-11 * np.abs(probe).max())
    x0, q0, f0 = slab.positions.copy(), pes.get_x(), pes.get_f()
    g0 = pes.get_g()
    for step, tol in ((0.02, 5e-4), (0.002, 5e-5)):            # relative miss of the target ~ step (second order)
        slab.positions = x0 + step * rng.normal(size=x0.shape)
        q1 = pes.int.calc()
        slab.positions = x0.copy()
        pes.get_g()
        dx_i, dx_f, g_par = pes.set_x(q1)
        dq = np.abs(q1 - q0).max()
        assert np.abs(pes.int.calc() - q1).max() < tol * dq
        np.testing.assert_allclose(dx_f, dx_i, atol=0.02 * dq)
        assert abs(np.linalg.norm(g_par) - np.linalg.norm(g0)) < 0.05 * np.linalg.norm(g0)
    # energy / internal gradient consistency by a central difference along +-dx (the second-order term of a
    # random 3072-dimensional displacement is ten times the first-order one and cancels here)
    dxc = 0.002 * rng.normal(size=x0.shape)
    fpm, qpm = [], []
    for sgn in (1.0, -1.0):
        slab.positions = x0 + sgn * dxc
        q1 = pes.int.calc()
        slab.positions = x0.copy()
        pes.get_g()
        pes.set_x(q1)
        fpm.append(pes.get_f())
        qpm.append(pes.int.calc())
    lhs, rhs = fpm[0] - fpm[1], g0 @ (qpm[0] - qpm[1])
    assert abs(lhs - rhs) < 0.02 * abs(rhs) + 1e-7




# ---- configs[3] as named: members are 256-atom EMT slabs ---------------------------------------------------------
def test_config3_ensemble_of_emt_slabs(ctx):
    """configs[3], one GPU's share: 8 independent 256-atom EMT saddle searches (3N = 768, 384 pinned coordinates,
    device EMT, default 'ras' step) through `run_ensemble` — serially and in worker processes — every member
    identical, bit for bit, to the search run alone."""
    from sella_amd.ensemble import EnsemblePool, run_ensemble, run_one
    steps, nrep = 6, 8
    kw = dict(order=1, eta=1e-4, gamma=0.1)
    alone = {}
    for i in range(nrep):
        slab, cons, pinned = emt_slab((8, 8, 4), seed=100 + i, jitter=0.02)
        assert len(slab) == 256 and len(pinned) == 128
        start = slab.positions.copy()
        alone[i] = run_one((slab, dict(constraints=cons)), 0.0, steps, kw) + (pinned, start)
    res = run_ensemble(EmtMember(), nrep, fmax=0.0, steps=steps, sella_kwargs=kw)
    assert res['summary'].shape == (nrep, 5) and np.all(np.isfinite(res['summary']))
    assert np.all(res['summary'][:, 1] == steps)
    for i in range(nrep):
        sm, pos, pinned, start = alone[i]
        np.testing.assert_array_equal(res['summary'][i], sm)
        np.testing.assert_array_equal(res['positions'][i], pos)
        np.testing.assert_array_equal(pos[pinned], start[pinned])               # pinned atoms never move
        assert np.abs(pos - start).max() > 1e-3
    assert len({round(e, 9) for e in res['summary'][:, 2]}) == nrep             # eight different searches
    assert np.all(res['summary'][:, 4] < 0.0)                                   # each holds a negative mode
    with EnsemblePool(2) as pool:
        res_p = run_ensemble(EmtMember(), nrep, fmax=0.0, steps=steps, sella_kwargs=kw, pool=pool)
    np.testing.assert_array_equal(res_p['summary'], res['summary'])
    # ... and on host threads of this process (one persistent device context each): the searches run inside the library
    # (sella_amd/search.py), so the threads overlap on the device; which thread takes which member does not matter
    from sella_amd import ensemble
    from sella_amd.ensemble import EnsembleThreads
    assert ensemble.USE_LIBRARY_SEARCH
    with EnsembleThreads(4) as threads:
        res_t = run_ensemble(EmtMember(), nrep, fmax=0.0, steps=steps, sella_kwargs=kw, threads=threads)
        res_t2 = run_ensemble(EmtMember(), nrep, fmax=0.0, steps=steps, sella_kwargs=kw, threads=threads)
    np.testing.assert_array_equal(res_t['summary'], res['summary'])
    np.testing.assert_array_equal(res_t2['summary'], res['summary'])
    for i in range(nrep):
        np.testing.assert_array_equal(res_t['positions'][i], res['positions'][i])
    # ... and in lockstep cohorts: one issuing thread per cohort, one batched launch per kernel for its members, one stream
    # synchronisation per phase (csrc/cohort.hip) — a full cohort of 8, two threads with 4 each, and a ragged 3 + 3 + 2
    from sella_amd.ensemble import EnsembleCohort, EnsembleCohorts
    with EnsembleCohort(8) as cohort:
        res_c = run_ensemble(EmtMember(), nrep, fmax=0.0, steps=steps, sella_kwargs=kw, cohort=cohort)
        st = cohort.stats()
    np.testing.assert_array_equal(res_c['summary'], res['summary'])
    for i in range(nrep):
        np.testing.assert_array_equal(res_c['positions'][i], res['positions'][i])
    assert 3 * st['launches_issued'] < st['launches_asked'] and 3 * st['stream_syncs'] < st['waits_asked'], st
    with EnsembleCohorts(4, 2, member_threads=True) as cohorts:          # (the members' host code on worker threads)
        res_c2 = run_ensemble(EmtMember(), nrep, fmax=0.0, steps=steps, sella_kwargs=kw, cohort=cohorts)
    with EnsembleCohort(3) as cohort:
        res_c3 = run_ensemble(EmtMember(), nrep, fmax=0.0, steps=steps, sella_kwargs=kw, cohort=cohort)
    for other in (res_c2, res_c3):
        np.testing.assert_array_equal(other['summary'], res['summary'])
        for i in range(nrep):
            np.testing.assert_array_equal(other['positions'][i], res['positions'][i])
    # the library loop against the general driver (the same searches, ~300 library calls per step there)
    ensemble.USE_LIBRARY_SEARCH = False
    try:
        res_g = run_ensemble(EmtMember(), 2, fmax=0.0, steps=steps, sella_kwargs=kw)
    finally:
        ensemble.USE_LIBRARY_SEARCH = True
    np.testing.assert_allclose(res_g['summary'], res['summary'][:2], atol=1e-7)
    for i in range(2):
        np.testing.assert_allclose(res_g['positions'][i], res['positions'][i], atol=1e-8)
