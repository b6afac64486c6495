"""`LibrarySearch` (sella_amd/search.py -> `sella_search_*`, csrc/search.hip): a whole `Sella(...).run()` inside the
library — first-use diagonalisation through the library's own calculator, one-call optimizer steps, the reference's
re-diagonalisation schedule — against the general driver on the same searches: same geometries, energies, radii,
numbers of steps and force calls.  The general driver is `Sella` with `use_library_loop = False` (`Sella.run` would
otherwise hand a covered search to the library itself and the comparison would be the library against itself)."""
import numpy as np
import pytest

from conftest import hessian_like


@pytest.fixture(autouse=True)
def structured_from_96():
    from sella_amd import linalg
    old = linalg.LR_MIN_DIM
    linalg.LR_MIN_DIM = 96
    yield
    linalg.LR_MIN_DIM = old


def _model(ctx, n=120, seed=41, nneg=1):
    from sella_amd.atoms import Atoms, QuadraticCubicModel
    A = hessian_like(n, seed, nneg=nneg)[0]
    dA = ctx.upload(A)
    rng = np.random.RandomState(seed + 1)
    U = rng.normal(size=(8, n))
    U /= np.linalg.norm(U, axis=1)[:, None]
    at = Atoms(['X'] * (n // 3), 0.05 * rng.normal(size=(n // 3, 3)), pbc=True)
    at.calc = QuadraticCubicModel(lambda x: ctx.symm_mm(dA, x), U, c=0.05, device_matrix=dA)
    return at


KW = dict(order=1, eta=1e-4, gamma=0.1, delta0=0.1, proj_trans=False)


def general_driver(atoms, **kw):
    from sella_amd import Sella
    opt = Sella(atoms, logfile=None, **kw)
    opt.use_library_loop = False
    return opt


@pytest.mark.parametrize('rs,nsteps_per_diag', [('tr', 3), pytest.param('ras', 2, marks=pytest.mark.emu_heavy)])
def test_model_search_in_the_library(ctx, rs, nsteps_per_diag):
    from sella_amd import Sella
    from sella_amd.internal import Constraints
    from sella_amd.search import LibrarySearch
    a1, a2 = _model(ctx), _model(ctx)
    kw = dict(KW, rs=rs, nsteps_per_diag=nsteps_per_diag)
    assert LibrarySearch.applies(a1, constraints=Constraints(a1), **kw)
    ls = LibrarySearch(a1, constraints=Constraints(a1), **kw)
    assert not ls.run(0.0, 5)
    assert ls.run(0.0, 4) is False and ls.nsteps == 9          # a search can be continued
    opt = general_driver(a2, constraints=Constraints(a2), **kw)
    opt.run(0.0, 9)
    assert opt._lib is None
    assert (ls.nsteps, ls.neval) == (opt.nsteps, opt.pes.neval)
    assert a1.calc.ncalls == a2.calc.ncalls
    np.testing.assert_allclose(a1.positions, a2.positions, atol=1e-8)
    assert ls.energy == pytest.approx(opt.pes.get_f(), abs=1e-9)
    assert ls.delta == pytest.approx(opt.delta, rel=1e-8) and ls.rho == pytest.approx(opt.rho, rel=1e-5)
    assert ls.lambda_min == pytest.approx(opt.pes.H.evals[0], abs=1e-7)
    assert ls.fmax_now == pytest.approx(opt.pes.converged(0.0)[1], abs=1e-8)
    assert ls.one_call_steps == 9
    # converges like the general driver does
    ls.close()


@pytest.mark.emu_heavy
def test_pinned_slab_search_in_the_library(ctx):
    """BASELINE configs[1] on a down-sized twin: EMT (library calculator), lower half pinned, default keywords."""
    from conftest_shim import emt_slab
    from sella_amd import Sella
    from sella_amd.search import LibrarySearch
    a1, c1, pinned = emt_slab((4, 4, 4))
    a2, c2, _ = emt_slab((4, 4, 4))
    start = a1.positions.copy()
    ls = LibrarySearch(a1, constraints=c1, nsteps_per_diag=2)
    ls.run(0.0, 7)
    opt = general_driver(a2, constraints=c2, nsteps_per_diag=2)
    opt.run(0.0, 7)
    assert opt._lib is None
    assert (ls.nsteps, ls.neval) == (opt.nsteps, opt.pes.neval)
    np.testing.assert_array_equal(a1.positions[pinned], start[pinned])
    np.testing.assert_allclose(a1.positions, a2.positions, atol=1e-9)
    assert ls.energy == pytest.approx(opt.pes.get_f(), abs=1e-9)
    assert ls.delta == pytest.approx(opt.delta, rel=1e-8)
    assert ls.rank_view > 0


def test_minimisation_without_curvature_information(ctx):
    """order = 0 with the defaults of a minimisation (eig = False, quasi-Newton family, optimize.py:20-39): the search
    starts from the identity, the first secant pair initialises the Hessian, no diagonalisation ever."""
    from sella_amd import Sella
    from sella_amd.internal import Constraints
    from sella_amd.search import LibrarySearch
    a1, a2 = _model(ctx, nneg=0), _model(ctx, nneg=0)
    kw = dict(order=0, proj_trans=False, rs='tr')
    assert LibrarySearch.applies(a1, constraints=Constraints(a1), **kw)
    ls = LibrarySearch(a1, constraints=Constraints(a1), **kw)
    ls.run(0.0, 8)
    opt = general_driver(a2, constraints=Constraints(a2), **kw)
    opt.run(0.0, 8)
    assert opt._lib is None
    assert (ls.nsteps, ls.neval) == (opt.nsteps, opt.pes.neval) == (8, 9)
    np.testing.assert_allclose(a1.positions, a2.positions, atol=1e-9)
    assert ls.energy == pytest.approx(opt.pes.get_f(), abs=1e-10) and ls.energy < 0.5 * float(a1.calc.energy_and_gradient(
        _model(ctx, nneg=0).positions)[0])
    assert ls.delta == pytest.approx(opt.delta, rel=1e-8) and ls.rho == pytest.approx(opt.rho, rel=1e-6)


@pytest.mark.parametrize('kw,nneg', [
    (dict(order=0, proj_trans=False, rs='ras'), 0),
    pytest.param(dict(KW, rs='ras', nsteps_per_diag=2), 1, marks=pytest.mark.emu_heavy),
    pytest.param(dict(KW, rs='tr', diag_every_n=3), 1, marks=pytest.mark.emu_heavy),
    pytest.param(dict(KW, rs='tr', method='rfo'), 1, marks=pytest.mark.emu_heavy),
    pytest.param(dict(KW, rs='tr', threepoint=True), 1, marks=pytest.mark.emu_heavy),
], ids=['order0-ras', 'saddle-ras', 'diag_every_n', 'rfo', 'threepoint'])
def test_library_branches_against_the_general_driver(ctx, kw, nneg):
    """The branches of csrc/search.hip that no other comparison reaches, each against the general driver."""
    from sella_amd.internal import Constraints
    from sella_amd.search import LibrarySearch
    a1, a2 = _model(ctx, nneg=nneg), _model(ctx, nneg=nneg)
    assert LibrarySearch.applies(a1, constraints=Constraints(a1), **kw)
    ls = LibrarySearch(a1, constraints=Constraints(a1), **kw)
    ls.run(0.0, 7)
    opt = general_driver(a2, constraints=Constraints(a2), **kw)
    opt.run(0.0, 7)
    assert opt._lib is None
    assert (ls.nsteps, ls.neval) == (opt.nsteps, opt.pes.neval)
    np.testing.assert_allclose(a1.positions, a2.positions, atol=1e-7)
    assert ls.energy == pytest.approx(opt.pes.get_f(), abs=1e-8)
    assert ls.delta == pytest.approx(opt.delta, rel=1e-7) and ls.rho == pytest.approx(opt.rho, rel=1e-4, abs=1e-6)
    ls.close()


@pytest.mark.emu_heavy
def test_pinned_model_search_against_the_general_driver(ctx):
    """Pinned coordinates (selection bases, principal-submatrix view) in the library loop against the general driver —
    on the model PES, so that the comparison also runs in the emulation."""
    from sella_amd.internal import Constraints
    from sella_amd.search import LibrarySearch
    a1, a2 = _model(ctx, n=180), _model(ctx, n=180)            # 162 free coordinates: room for the first update's rank
    c1, c2 = Constraints(a1), Constraints(a2)
    for i in range(6):
        c1.fix_translation(i)
        c2.fix_translation(i)
    kw = dict(KW, rs='ras', nsteps_per_diag=2)
    start = a1.positions.copy()
    assert LibrarySearch.applies(a1, constraints=c1, **kw)
    ls = LibrarySearch(a1, constraints=c1, **kw)
    ls.run(0.0, 6)
    opt = general_driver(a2, constraints=c2, **kw)
    opt.run(0.0, 6)
    assert opt._lib is None
    assert (ls.nsteps, ls.neval) == (opt.nsteps, opt.pes.neval)
    np.testing.assert_array_equal(a1.positions[:6], start[:6])
    np.testing.assert_allclose(a1.positions, a2.positions, atol=1e-7)
    assert ls.delta == pytest.approx(opt.delta, rel=1e-7)
    ls.close()


def test_library_search_says_what_it_covers(ctx):
    from conftest_shim import emt_slab
    from sella_amd.atoms import Atoms, MorseCluster
    from sella_amd.internal import Constraints
    from sella_amd.search import LibrarySearch
    at = _model(ctx)
    assert not LibrarySearch.applies(at, constraints=Constraints(at))                 # default: global translation constraint
    assert not LibrarySearch.applies(at, constraints=Constraints(at), proj_trans=False, trajectory='x.traj')
    assert not LibrarySearch.applies(at, constraints=Constraints(at), proj_trans=False, rs='mis')
    mc = Atoms(['Xe'] * 40, np.random.RandomState(0).normal(size=(40, 3)), pbc=True)
    mc.calc = MorseCluster()
    assert not LibrarySearch.applies(mc, constraints=Constraints(mc), proj_trans=False)            # host-language calculator
    slab, cons, _ = emt_slab((4, 4, 4))
    cons.fix_bond((0, 1))
    assert not LibrarySearch.applies(slab, constraints=cons)
    with pytest.raises(ValueError):
        LibrarySearch(at, constraints=Constraints(at))


def test_run_one_takes_the_library_route(ctx, monkeypatch):
    from sella_amd import ensemble
    from sella_amd.internal import Constraints
    kw = dict(KW, rs='tr')
    a1, a2 = _model(ctx), _model(ctx)
    monkeypatch.setattr(ensemble, 'USE_LIBRARY_SEARCH', True)
    s1, p1 = ensemble.run_one((a1, dict(constraints=Constraints(a1))), 0.0, 6, kw)
    monkeypatch.setattr(ensemble, 'USE_LIBRARY_SEARCH', False)
    s2, p2 = ensemble.run_one((a2, dict(constraints=Constraints(a2))), 0.0, 6, kw)
    np.testing.assert_allclose(p1, p2, atol=1e-8)
    np.testing.assert_allclose(s1, s2, atol=1e-7)
    assert s1[1] == 6 and s1[4] < 0


@pytest.mark.parametrize('pinned', [pytest.param(False, marks=pytest.mark.emu_heavy),
                                    pytest.param(True, marks=pytest.mark.emu_heavy)])
def test_sella_run_hands_the_search_to_the_library_and_takes_it_back(ctx, pinned):
    """`Sella(...).run()` on a covered configuration runs inside the library; the first look at `opt.pes` brings the
    state back — geometry, energy, gradient, counters, approximate Hessian with its structured eigendecomposition (and
    the view of the pinned coordinates) — and the general driver continues from there exactly like an optimizer that
    never left it."""
    from conftest_shim import emt_slab
    from sella_amd import Sella
    from sella_amd.internal import Constraints

    def make(lib):
        if pinned:
            atoms, cons, _ = emt_slab((4, 4, 4))
            opt = Sella(atoms, constraints=cons, logfile=None, nsteps_per_diag=2)
        else:
            atoms = _model(ctx)
            opt = Sella(atoms, constraints=Constraints(atoms), logfile=None, rs='tr', **KW)
        opt.use_library_loop = lib
        return atoms, opt
    a1, o1 = make(True)
    a2, o2 = make(False)
    assert o1.run(0.0, 4) is False and o1._lib is not None and o1._lib_authoritative
    o1.run(0.0, 2)                                            # continues inside the library
    assert o1._lib is not None and o1.nsteps == 6 and o1.fused_steps >= 5
    o2.run(0.0, 6)
    assert o2._lib is None
    np.testing.assert_allclose(a1.positions, a2.positions, atol=1e-8)
    pes = o1.pes                                              # <- the state comes back here
    assert o1._lib is None and not o1._lib_authoritative
    assert (o1.nsteps, pes.neval, o1.initialized) == (o2.nsteps, o2.pes.neval, True)
    assert a1.calc.ncalls == a2.calc.ncalls
    assert pes.get_f() == pytest.approx(o2.pes.get_f(), abs=1e-9)
    np.testing.assert_allclose(pes.get_g(), o2.pes.get_g(), atol=1e-7)
    assert pes.neval == o2.pes.neval                          # (looking did not cost a force call)
    assert o1.delta == pytest.approx(o2.delta, rel=1e-8) and o1.nsteps_since_diag == o2.nsteps_since_diag
    assert pes.H.device_eig_lr() is not None
    np.testing.assert_allclose(pes.H.B, o2.pes.H.B, atol=1e-6 * max(1.0, np.abs(o2.pes.H.B).max()))
    if pinned:
        v1, v2 = pes.get_HL_projected(pes.get_Ufree()), o2.pes.get_HL_projected(o2.pes.get_Ufree())
        assert v1 is not pes.H and v1.device_eig_lr() is not None
        np.testing.assert_allclose(v1.B, v2.B, atol=1e-6 * max(1.0, np.abs(v2.B).max()))
    for _ in range(3):                                        # both in the general driver now
        o1.step()
        o2.step()
    np.testing.assert_allclose(a1.positions, a2.positions, atol=1e-7)
    assert o1.pes.neval == o2.pes.neval
    # a logging optimizer is never handed over
    a3, o3 = make(True)
    o3.logfile = open('/dev/null', 'w')
    o3.run(0.0, 1)
    assert o3._lib is None
    o3.logfile.close()


def test_atoms_moved_between_two_runs(ctx):
    """A run inside the library, then somebody moves the atoms, then `run()` again: the energy and gradient the library
    holds belong to ITS geometry, so the second run must make a force call at the new positions (not carry the stale
    gradient over as if it belonged there) and then walk exactly like the general driver put through the same sequence."""
    from sella_amd import Sella
    from sella_amd.internal import Constraints

    def make(lib):
        atoms = _model(ctx)
        opt = Sella(atoms, constraints=Constraints(atoms), logfile=None, rs='tr', **KW)
        opt.use_library_loop = lib
        return atoms, opt
    a1, o1 = make(True)
    a2, o2 = make(False)
    o1.run(0.0, 6)
    o2.run(0.0, 6)
    assert o1._lib is not None and o1._lib_authoritative and o1.nsteps == o2.nsteps == 6
    kick = 0.02 * np.random.RandomState(7).normal(size=a1.positions.shape)
    a1.positions = a1.positions + kick
    a2.positions = a2.positions + kick
    calls = a1.calc.ncalls
    o1.run(0.0, 3)
    o2.run(0.0, 3)
    assert o1._lib is None                                    # the state came back, the general driver continued
    assert a1.calc.ncalls > calls + 3                         # ... with a force call at the perturbed geometry on top of the steps'
    assert o1.nsteps == o2.nsteps and o1.pes.neval == o2.pes.neval and a1.calc.ncalls == a2.calc.ncalls
    np.testing.assert_allclose(a1.positions, a2.positions, atol=1e-7)
    assert o1.pes.get_f() == pytest.approx(o2.pes.get_f(), abs=1e-9)
    np.testing.assert_allclose(o1.pes.get_g(), o2.pes.get_g(), atol=1e-7)


@pytest.mark.parametrize('n,nsteps', [pytest.param(120, 26, marks=pytest.mark.emu_heavy), (96, 6)],
                         ids=['before-a-step', 'inside-a-diagonalisation'])
def test_hand_over_at_the_rank_limit_keeps_the_reference_semantics(ctx, n, nsteps):
    """The explicit rank of the structured Hessian reaches 0.4 n and the library hands the search back.  Nothing may be
    lost on the way.  n = 120: the limit is reached by the quasi-Newton pairs of the steps — the hand-over happens
    BEFORE a step, geometry and force-call count untouched.  n = 96: already the block of the first-use diagonalisation
    (2 x 29 vectors) exceeds it — the force calls are spent, so its secant pairs come with the hand-over
    (sella_search_pending_pairs) and the caller applies them.  Either way the run as a whole is the general driver's run:
    same steps, force calls, geometry, radius."""
    from sella_amd import Sella
    from sella_amd.internal import Constraints
    kw = dict(KW, rs='tr', nsteps_per_diag=3)
    a1, a2 = _model(ctx, n=n), _model(ctx, n=n)
    o1 = Sella(a1, constraints=Constraints(a1), logfile=None, **kw)
    o2 = general_driver(a2, constraints=Constraints(a2), **kw)
    o1.run(0.0, nsteps)
    assert o1._lib is None                                   # the library did hand the search back before the end
    if n == 120:
        assert o1.fused_steps > 3 and getattr(o1, 'pairs_adopted', 0) == 0
    else:
        assert o1.pairs_adopted >= 24                        # the first diagonalisation's block, applied on the dense route
    o2.run(0.0, nsteps)
    assert (o1.nsteps, o1.pes.neval) == (o2.nsteps, o2.pes.neval) == (nsteps, o2.pes.neval)
    assert a1.calc.ncalls == a2.calc.ncalls
    np.testing.assert_allclose(a1.positions, a2.positions, atol=1e-6)
    assert o1.delta == pytest.approx(o2.delta, rel=1e-6)
    assert o1.nsteps_since_diag == o2.nsteps_since_diag


def test_upload_paths_agree(ctx):
    """Host-to-device payloads by the copy kernel that reads the pinned ring (`h2d_kernel_min`: from 16 KB on by default)
    against the runtime's copy for everything (0) and against the kernel for everything including the one-double tail (8):
    a transport change only — the same search, bit for bit."""
    from sella_amd.internal import Constraints
    from sella_amd.search import LibrarySearch
    out = []
    try:
        for kmin in (16384, 0, 8):
            ctx.set_option('h2d_kernel_min', kmin)
            at = _model(ctx, n=126, seed=43)                   # 126: an odd number of doubles in some payloads
            ls = LibrarySearch(at, constraints=Constraints(at), **dict(KW, rs='tr', nsteps_per_diag=3))
            assert not ls.run(0.0, 6)
            out.append((at.positions.copy(), ls.energy, ls.delta))
            ls.close()
    finally:
        ctx.set_option('h2d_kernel_min', 16384)
    for o in out[1:]:
        np.testing.assert_array_equal(o[0], out[0][0])
        assert o[1] == out[0][1] and o[2] == out[0][2]


def _cohort_member(i):
    from sella_amd import device
    from sella_amd.internal import Constraints
    at = _model(device.get_context(), seed=41 + 3 * i)
    return at, dict(constraints=Constraints(at))


@pytest.mark.parametrize('rs', ['tr', pytest.param('ras', marks=pytest.mark.emu_heavy)])
def test_cohort_members_are_bit_identical_to_searches_run_alone(ctx, rs):
    """`EnsembleCohort`: members advanced in lockstep from one thread, one batched launch per kernel for all of them
    (csrc/cohort.hip) — every member's summary and geometry equal, bit for bit, what `run_one` gives for it alone
    (independent `Sella` objects, sella/optimize/optimize.py:42-81).  Width 3 with 5 members (4 on the emulator): a full wave and a ragged
    one; the members differ, so their ranks, root-search rounds and deflations do."""
    from sella_amd import device, ensemble
    kw = dict(KW, rs=rs, nsteps_per_diag=3)
    nrep, steps = (4, 4) if ctx.backend == 'emu' else (5, 7)
    alone = [ensemble.run_one(_cohort_member(i), 0.0, steps, kw) for i in range(nrep)]
    with ensemble.EnsembleCohort(3) as cohort:
        res = ensemble.run_ensemble(_cohort_member, nrep, fmax=0.0, steps=steps, sella_kwargs=kw, cohort=cohort)
        again = res if ctx.backend == 'emu' else ensemble.run_ensemble(_cohort_member, nrep, fmax=0.0, steps=steps, sella_kwargs=kw, cohort=cohort)
        st = cohort.stats()
    # the members' host code on worker threads instead of fibers (parallel host code, the same merged launches)
    with ensemble.EnsembleCohort(3, member_threads=True) as cohort:
        res_mt = ensemble.run_ensemble(_cohort_member, nrep, fmax=0.0, steps=steps, sella_kwargs=kw, cohort=cohort)
        st_mt = cohort.stats()
    assert st_mt['launches_issued'] < st_mt['launches_asked']
    assert device.get_context() is ctx
    for i in range(nrep):
        np.testing.assert_array_equal(res['summary'][i], alone[i][0])
        np.testing.assert_array_equal(res['positions'][i], alone[i][1])
        np.testing.assert_array_equal(again['summary'][i], alone[i][0])
        np.testing.assert_array_equal(res_mt['summary'][i], alone[i][0])
        np.testing.assert_array_equal(res_mt['positions'][i], alone[i][1])
    assert len({round(e, 9) for e in res['summary'][:, 2]}) == nrep
    assert st['launches_issued'] < st['launches_asked']                        # launches were merged across the members
    assert st['stream_syncs'] < st['waits_asked']
