"""`Sella.step` as one library call (`sella_opt_step`, csrc/optstep.hip: model prediction + quasi-Newton update with the
structured eigendecomposition + trust-radius rule + next restricted step) against the general path, which performs the
same operations one library call at a time (sella/optimize/optimize.py:359-440, peswrapper.py:578-602): the same
trajectories, radii, ratios, force-call counts and Hessians."""
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


def _model_search(fused, rs, method, order, nsteps=10, n=120):
    from sella_amd import Sella
    from sella_amd.atoms import Atoms, QuadraticCubicModel
    from sella_amd.internal import Constraints
    A = hessian_like(n, 41)[0]
    rng = np.random.RandomState(42)
    Uc = rng.normal(size=(8, n))
    Uc /= np.linalg.norm(Uc, axis=1)[:, None]
    at = Atoms(['X'] * (n // 3), 0.05 * rng.normal(size=(n // 3, 3)), pbc=True)
    at.calc = QuadraticCubicModel(lambda x: A @ x, Uc, c=0.05)
    opt = Sella(at, order=order, eta=1e-4, gamma=0.1, delta0=0.1, rs=rs, method=method, logfile=None,
                constraints=Constraints(at), proj_trans=False)
    opt.use_fused_step = fused
    return opt


def _run(opt, nsteps):
    rows, calls = [], 0
    real = type(opt)._step_fused

    def counting(self, blk):
        nonlocal calls
        calls += 1
        return real(self, blk)
    type(opt)._step_fused = counting
    try:
        for _ in range(nsteps):
            opt.step()
            rows.append((opt.pes.get_x().copy(), opt.pes.get_f(), opt.delta, opt.rho, opt.nsteps_since_diag))
    finally:
        type(opt)._step_fused = real
    return rows, calls


def _same_row(ra, rb, i):
    """Same search: the one-call step updates the eigendecomposition in coordinates (csrc/lrstep.hip) where the general
    path runs host-planned rank-one merges on the eigenvector panel — two evaluations of the same matrices, equal to
    roundoff at each step, the difference then growing with the search like between any two runs whose arithmetic
    differs in the last bit (tests/test_pes_oracle.py uses the same schedule of tolerances)."""
    tol = 1e-11 * 4 ** min(i, 10)
    np.testing.assert_allclose(ra[0], rb[0], atol=tol, rtol=0, err_msg=f'step {i}')
    assert ra[1] == pytest.approx(rb[1], abs=tol), i
    assert ra[2] == pytest.approx(rb[2], rel=1e-9, abs=tol), i
    assert ra[3] == pytest.approx(rb[3], rel=1e-6, abs=1e-6), i
    assert ra[4] == rb[4], i


@pytest.mark.parametrize('rs,method,order', [('tr', 'prfo', 1), pytest.param('ras', 'rfo', 0, marks=pytest.mark.emu_heavy),
                                             pytest.param('ras', 'qn', 0, marks=pytest.mark.emu_heavy)])
def test_fused_step_equals_the_general_path(ctx, rs, method, order):
    a, na = _run(_model_search(True, rs, method, order), 8)
    b, nb = _run(_model_search(False, rs, method, order), 8)
    assert na >= 6 and nb == 0                      # the first step initialises; the rest are single calls
    for i, (ra, rb) in enumerate(zip(a, b)):
        _same_row(ra, rb, i)


@pytest.mark.emu_heavy
def test_fused_step_with_pinned_coordinates(ctx):
    """BASELINE configs[1] on a down-sized twin (Cu(111) 4 x 4 x 4 EMT slab, lower half pinned atom by atom): pins -> selection bases and the
    principal-submatrix view, default `Sella` (`ras`, P-RFO)."""
    from conftest_shim import emt_slab
    from sella_amd import Sella
    out = {}
    for fused in (True, False):
        atoms, cons, pinned = emt_slab((4, 4, 4))        # (3, 3, 4) leaves the view too small for the structured form
        opt = Sella(atoms, constraints=cons, logfile=None)
        opt.use_fused_step = fused
        rows, calls = _run(opt, 6)
        out[fused] = (rows, calls, opt.pes.neval, opt.pes.H.B.copy(), atoms.positions[pinned].copy())
    assert out[True][1] >= 4 and out[False][1] == 0
    assert out[True][2] == out[False][2]
    for i, (ra, rb) in enumerate(zip(out[True][0], out[False][0])):
        _same_row(ra, rb, i)
    np.testing.assert_allclose(out[True][3], out[False][3], atol=1e-9)
    np.testing.assert_array_equal(out[True][4], out[False][4])


@pytest.mark.emu_heavy
def test_fused_step_steps_aside(ctx):
    """Configurations the one-call step does not cover take the general path: dense eigendecomposition (structured form
    off), a rotation constraint (general projection basis), a user-defined restricted step."""
    from sella_amd import linalg
    from sella_amd.optimize.restricted_step import TrustRegion
    linalg.LR_MIN_DIM = None
    rows, calls = _run(_model_search(True, 'tr', 'prfo', 1), 4)
    assert calls == 0
    linalg.LR_MIN_DIM = 96

    class MyRegion(TrustRegion):
        pass
    opt = _model_search(True, 'tr', 'prfo', 1)
    opt.rs = MyRegion
    assert _run(opt, 3)[1] == 0


def test_general_route_inside_the_call_and_stale_mirrors(ctx):
    """Inside `sella_opt_step` the coordinate update can be switched off (option `lr_dev`): the call then runs the
    one-phase entry points in sequence, after rebuilding the dense mirrors the coordinate steps before it left stale
    (`sella_lr_materialize`).  Switching back and forth in the middle of a search changes nothing but the last bits;
    `H.B` read in between equals the matrix the general path carries."""
    ref_rows, _ = _run(_model_search(False, 'tr', 'prfo', 1), 9)
    opt = _model_search(True, 'tr', 'prfo', 1)
    rows = []
    try:
        for leg, flag in enumerate((1, 0, 1)):
            ctx.set_option('lr_dev', flag)
            got, calls = _run(opt, 3)
            assert calls >= 2
            rows += got
            assert opt.pes.H._B_stale == bool(flag)          # coordinate steps leave the mirror behind, the others do not
            B = opt.pes.H.B                                    # (rebuilt on demand)
            assert not opt.pes.H._B_stale
            np.testing.assert_array_equal(B, B.T)
    finally:
        ctx.set_option('lr_dev', 1)
    for i, (ra, rb) in enumerate(zip(rows, ref_rows)):
        _same_row(ra, rb, i)


@pytest.mark.emu_heavy
def test_restart_drops_the_proposed_step(ctx, tmp_path):
    """`load_state` replaces the approximate Hessian: a step proposed by the previous call must not survive it."""
    opt = _model_search(True, 'tr', 'prfo', 1)
    _run(opt, 4)
    opt.save_state(str(tmp_path / 'state'))
    x_saved = opt.pes.get_x().copy()
    _run(opt, 2)
    opt.load_state(str(tmp_path / 'state'))
    np.testing.assert_array_equal(opt.pes.get_x(), x_saved)
    assert opt._fused_block() is None                        # dense B from the file: general route
    a, _ = _run(opt, 2)
    fresh = _model_search(False, 'tr', 'prfo', 1)
    _run(fresh, 4)
    fresh.save_state(str(tmp_path / 'state2'))
    fresh.load_state(str(tmp_path / 'state2'))
    b, _ = _run(fresh, 2)
    for i, (ra, rb) in enumerate(zip(a, b)):
        _same_row(ra, rb, i + 4)


def test_batched_trial_steps_on_the_panel_rows(ctx):
    """The per-atom measure's batched root search (`rs_batch`, 15 trial alphas per round trip) with the step family
    reading its modes from the update's panel in place (`rs_panel_apply_kernel`) against the one-at-a-time search: the
    same steps.  (The emulation's default switches the batches off — minutes of fibre switching per whole run — so this
    short run is the CPU suite's only pass through that kernel with 15 right-hand sides.)"""
    out = {}
    for batch in (1, 0):
        ctx.set_option('rs_batch', batch)
        try:
            out[batch], _ = _run(_model_search(True, 'ras', 'rfo', 0), 4)
        finally:
            ctx.set_option('rs_batch', 0 if ctx.backend == 'emu' else 1)
    for i, (ra, rb) in enumerate(zip(out[1], out[0])):
        np.testing.assert_allclose(ra[0], rb[0], atol=1e-9 * 4 ** i, rtol=0, err_msg=f'step {i}')
        assert ra[2] == pytest.approx(rb[2], rel=1e-9)
