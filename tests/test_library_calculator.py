"""Calculators that live in the library (`sella_calc_*`, csrc/calc.hip) and the finite-difference Hessian operator on
top of one (`sella_fd_*`: sella/linalg.py:14-101 restated on the far side of the calculator boundary, so that the force
calls of an iterative diagonalisation are library calls) against the host-language objects they stand in for."""
import ctypes

import numpy as np
import pytest

from conftest import hessian_like


def _model(ctx, n=120, seed=41, lib=True):
    from sella_amd.atoms import Atoms, QuadraticCubicModel
    A = hessian_like(n, seed)[0]
    dA = ctx.upload(A)
    rng = np.random.RandomState(seed + 1)
    U = rng.normal(size=(8, n))
    U /= np.linalg.norm(U, axis=1)[:, None]
    at = Atoms(['X'] * (n // 3), 0.05 * rng.normal(size=(n // 3, 3)), pbc=True)
    at.calc = QuadraticCubicModel(lambda x: ctx.symm_mm(dA, x), U, c=0.05, device_matrix=dA if lib else None)
    return at


def test_model_calculator_in_the_library(ctx):
    at = _model(ctx)
    e, g = at.calc.energy_and_gradient(at.positions)
    dc = at.calc.device_calculator()
    e2, g2 = dc.eval(at.positions)
    assert e2 == pytest.approx(e, rel=1e-13, abs=1e-14)
    np.testing.assert_allclose(g2, g.ravel(), atol=1e-13)
    assert dc.ncalls == 1
    assert _model(ctx, lib=False).calc.device_calculator() is None


def test_emt_calculator_in_the_library(ctx):
    from conftest_shim import emt_slab
    slab, _, _ = emt_slab((3, 3, 4))
    assert slab.calc.device_calculator() is None              # not set up for a geometry yet
    e = slab.get_potential_energy()
    g = -slab.get_forces()
    dc = slab.calc.device_calculator()
    e2, g2 = dc.eval(slab.positions)
    assert e2 == e
    np.testing.assert_array_equal(g2.reshape(g.shape), g)
    before = slab.calc.ncalls
    dc.eval(slab.positions + 0.01)
    assert slab.calc.ncalls == before + 1                    # force calls the library makes are counted with the others


def test_emt_neighbour_lists_that_overflow(ctx):
    """The EMT kernels note each thread's neighbours first (eight slots per thread, far more than EMT's cutoff ever fills)
    and hand the lists from the density pass to the force pass.  With one slot per thread (option `emt_hcap`) and the
    cutoff opened up, threads find more than their list holds: it is worked off in batches, marked incomplete, and the
    force pass sweeps again — same energy and forces as the oracle's all-pairs sums with the same cutoff."""
    from conftest_shim import emt_slab
    from oracle.sella_oracle.emt import EMTOracle
    slab, _, _ = emt_slab((4, 4, 5), seed=5, jitter=0.03, calculator=EMTOracle())
    slab.get_potential_energy()                                  # sets the oracle up for this cell
    S = slab.calc._setup[1]
    big = S['rc'] + 9.0
    S['cutoff'] = big
    e_ref, g_ref = slab.calc.energy_and_gradient(slab.positions)
    par = np.array([S['arr'][k] for k in ('E0', 's0', 'V0', 'eta2', 'kappa', 'lam', 'n0', 'gamma1', 'gamma2')])
    try:
        ctx.set_option('emt_hcap', 1)
        e, g = ctx.emt_eval(slab.positions, par, S['shifts'], S['rc'], S['acut'], big, slab.calc._BETA)
    finally:
        ctx.set_option('emt_hcap', 8)
    assert len(slab) * len(S['shifts']) > 2 * 256               # more pairs than one slot per thread holds
    assert abs(e - e_ref) <= 1e-11 * abs(e_ref)
    np.testing.assert_allclose(g, g_ref, atol=1e-11 * np.abs(g_ref).max())
    e8, g8 = ctx.emt_eval(slab.positions, par, S['shifts'], S['rc'], S['acut'], big, slab.calc._BETA)
    assert e8 == e and np.array_equal(g8, g)                    # the batches keep the order of the terms: same bits
    # and at EMT's own cutoff on the same geometry (complete lists, the usual path)
    S['cutoff'] = S['rc'] + 0.5
    e_ref, g_ref = slab.calc.energy_and_gradient(slab.positions)
    e, g = ctx.emt_eval(slab.positions, par, S['shifts'], S['rc'], S['acut'], S['cutoff'], slab.calc._BETA)
    assert abs(e - e_ref) <= 1e-12 * abs(e_ref)
    np.testing.assert_allclose(g, g_ref, atol=1e-12 * np.abs(g_ref).max())


@pytest.mark.parametrize('threepoint', [False, True])
@pytest.mark.parametrize('pinned', [False, True])
def test_fd_operator_equals_numerical_hessian(ctx, pinned, threepoint):
    from sella_amd import _lib
    from sella_amd.device import DeviceFdOperator
    from sella_amd.linalg import NumericalHessian
    from sella_amd.utilities.math import register_selection
    at = _model(ctx)
    n = 120
    x0 = at.positions.ravel().copy()
    dc = at.calc.device_calculator()
    _, g0 = dc.eval(x0)

    def func(x):
        return at.calc.energy_and_gradient(x.reshape(-1, 3))[0], at.calc.energy_and_gradient(x.reshape(-1, 3))[1].ravel()
    free, U = None, None
    if pinned:
        free = np.setdiff1d(np.arange(n), np.arange(0, n, 7)).astype(np.int32)
        U = register_selection(np.ascontiguousarray(np.eye(n)[:, free]), free)
    ref = NumericalHessian(func, x0, g0, 1e-4, threepoint, U)
    op = DeviceFdOperator(dc, x0, g0, 1e-4, threepoint, free)
    assert op.shape == ref.shape
    rng = np.random.RandomState(3)
    m = op.shape[0]
    for trial in range(4):
        v = rng.normal(size=m) * (1e-3 if trial == 2 else 1.0)
        if trial == 3:
            v = np.zeros(m)                                   # a vanishing direction: zero product, not remembered
        out = np.empty(m)
        assert _lib.lib().sella_fd_matvec(op._h, v.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), m) == 0
        np.testing.assert_allclose(out, ref._matvec(v), atol=2e-9 * max(1.0, np.abs(out).max()))
    assert op.calls == ref.calls == 4
    np.testing.assert_array_equal(op.Vs, ref.Vs)
    np.testing.assert_allclose(op.AVs, ref.AVs, atol=2e-9 * max(1.0, np.abs(ref.AVs).max()))
    assert op.Vs.shape == (n, 3)


@pytest.mark.emu_heavy
def test_search_through_the_library_calculator(ctx, monkeypatch):
    """The same `Sella` search with the diagonalisations' force calls made by the library and by the interpreter:
    same trajectory (to the amplification of last-bit differences of the model's cubic term by 1 / eta), same number
    of force calls on both counters."""
    from sella_amd import Sella, linalg
    from sella_amd.internal import Constraints
    monkeypatch.setattr(linalg, 'LR_MIN_DIM', 96)
    runs = {}
    for lib in (True, False):
        at = _model(ctx, lib=lib)
        opt = Sella(at, order=1, eta=1e-4, gamma=0.1, delta0=0.1, rs='tr', logfile=None, constraints=Constraints(at),
                    proj_trans=False, nsteps_per_diag=2)
        xs = []
        for _ in range(7):
            opt.step()
            xs.append(opt.pes.get_x().copy())
        runs[lib] = (np.array(xs), opt.pes.neval, at.calc.ncalls)
    assert runs[True][1] == runs[False][1] and runs[True][2] == runs[False][2]
    assert runs[True][1] > 20
    np.testing.assert_allclose(runs[True][0], runs[False][0], atol=1e-8)


def test_fd_operator_against_the_reference(ctx, manifest):
    """`sella_fd_matvec` over `sella_calc_model_*` against golden vectors of the REFERENCE's `NumericalHessian`
    (sella/linalg.py:14-101) on the same model PES: products — through every branch of the orientation rule, a short
    vector, the zero vector — and the remembered secant pairs, in the full space and through a selection of
    coordinates, forward and central differences (`g12_numhess_model`, oracle/make_golden.py)."""
    from conftest import load_golden
    from sella_amd import _lib
    from sella_amd.device import DeviceCalculator, DeviceFdOperator
    g = load_golden('g12_numhess_model')
    assert len(manifest['g12_numhess_model']) == 4
    for case in manifest['g12_numhess_model']:
        i, n = case['id'], case['n']
        A, U, x, g0, M = (g[f'c{i}_{k}'] for k in ('A', 'U', 'x', 'g', 'M'))
        free = g[f'c{i}_free'].astype(np.int32) if case['nfree'] > 0 else None
        dA = ctx.upload(A)
        dc = DeviceCalculator.model(ctx, dA, U, case['c'])
        f0, gl = dc.eval(x)
        np.testing.assert_allclose(gl, g0, atol=1e-13 * max(1.0, np.abs(g0).max()))
        op = DeviceFdOperator(dc, x, g0, case['eta'], case['threepoint'], free)
        m = op.shape[0]
        assert m == M.shape[0]
        out = np.empty_like(M)
        for col in range(M.shape[1]):
            v = np.ascontiguousarray(M[:, col])
            o = np.empty(m)
            assert _lib.lib().sella_fd_matvec(op._h, v.ctypes.data_as(ctypes.c_void_p), o.ctypes.data_as(ctypes.c_void_p), m) == 0
            out[:, col] = o
        scale = max(1.0, np.abs(g[f'c{i}_out']).max())
        # (a last-bit difference of the gradient — device matvec against NumPy's — is amplified by 1 / eta)
        tol = 4e-16 * np.abs(A).sum(axis=1).max() * max(1.0, np.abs(x).max()) / case['eta'] * 50
        np.testing.assert_allclose(out, g[f'c{i}_out'], atol=tol * scale, err_msg=str(case))
        np.testing.assert_array_equal(op.Vs, g[f'c{i}_Vs'])
        np.testing.assert_allclose(op.AVs, g[f'c{i}_AVs'], atol=tol * scale)
        assert op.calls == M.shape[1] and op.Vs.shape[1] == M.shape[1] - 1        # the zero vector is not remembered
