"""Step solve: step families (stepper.py) and the restricted-step root find (restricted_step.py)
on the device against goldens generated from the real reference (g7, g8) and against the oracle
(per-alpha traces)."""
import numpy as np
import pytest

import oracle.sella_oracle as orc
from conftest import load_golden
from helpers import FakePES, InternalCounts


def test_golden_steppers(ctx, manifest):
    from sella_amd.linalg import ApproximateHessian
    from sella_amd.optimize.stepper import get_stepper
    g = load_golden('g7_steppers')
    for case in manifest['g7_steppers']:
        i = case['id']
        gg = g[f'c{i}_g']
        n = len(gg)
        st = get_stepper(case['name'])(gg, ApproximateHessian(n, 0, g[f'c{i}_H']), case['order'])
        for a in range(case['nalpha']):
            s, ds = st.get_s(float(g[f'c{i}_a{a}_alpha']))
            rs, rd = g[f'c{i}_a{a}_s'], g[f'c{i}_a{a}_dsda']
            np.testing.assert_allclose(s, rs, atol=1e-10 * max(1, np.abs(rs).max()), err_msg=str(case))
            np.testing.assert_allclose(ds, rd, atol=1e-9 * max(1, np.abs(rd).max()), err_msg=str(case))


def test_stepper_registry():
    from sella_amd.optimize.stepper import (PartitionedRationalFunctionOptimization, QuasiNewton,
                                            RationalFunctionOptimization, get_stepper)
    assert get_stepper('mmf') is QuasiNewton
    assert get_stepper('rfo') is RationalFunctionOptimization
    assert get_stepper('p-rfo') is PartitionedRationalFunctionOptimization
    with pytest.raises(ValueError):
        get_stepper('nope')
    assert RationalFunctionOptimization.alpha0 == 1. and not RationalFunctionOptimization.newton_safe
    assert QuasiNewton.alphamax == np.inf and QuasiNewton.slope == -1


def test_golden_restricted_step(ctx, manifest):
    from sella_amd.linalg import ApproximateHessian
    from sella_amd.optimize.restricted_step import get_restricted_step
    g = load_golden('g8_restricted_step')
    for case in manifest['g8_restricted_step']:
        i = case['id']
        if ctx.backend == 'emu' and case['rs'] == 'ras' and case['method'] != 'prfo':
            continue          # CPU emulation: all of 'tr' + the P-RFO 'ras' cases (every branch); everything on the GPU
        pes = FakePES(ApproximateHessian, g[f'c{i}_H'], g[f'c{i}_g'], case['ncons'], seed=i)
        np.testing.assert_array_equal(pes.Ufree, g[f'c{i}_Ufree'])
        rs = get_restricted_step(case['rs'])(pes, case['order'], case['delta'], case['method'])
        s, smag = rs.get_s()
        ref = g[f'c{i}_s']
        np.testing.assert_allclose(smag, float(g[f'c{i}_smag']), rtol=1e-12)
        if case['method'] == 'rfo' and case['order'] >= 1 and len(rs.alphas) > 60:
            # plain RFO following an INTERIOR root only reaches a small radius at alpha ~ 1e-13,
            # where the selected eigenvector of the scaled augmented matrix sits in a cluster of
            # O(alpha^2) eigenvalues: the reference's own direction is roundoff-defined there
            # (Sella never uses this combination: saddles default to P-RFO, optimize.py:30-38)
            cos = s @ ref / np.linalg.norm(s) / np.linalg.norm(ref)
            assert cos > 0.9, case
            continue
        np.testing.assert_allclose(s, ref, atol=1e-10 * max(1, np.abs(ref).max()), err_msg=str(case))
        # same number of trial alphas as the reference algorithm (per-iteration parity)
        assert abs(len(rs.alphas) - int(g[f'c{i}_nalpha'])) <= 1, case


def _mis_pes(Hcls, g, case, i):
    pes = FakePES(Hcls, g[f'c{i}_H'], g[f'c{i}_g'], case['ncons'], seed=case['seed'])
    pes.int = InternalCounts(*case['blocks'])
    return pes


@pytest.mark.emu_heavy
def test_golden_restricted_mis(ctx, manifest):
    """`mis` — MaxInternalStep (restricted_step.py:186-243), the trust measure of every internal-coordinate search —
    against fixtures from the real reference class (oracle/make_golden.py::gen_restricted_mis): weights per block of
    coordinates, step, reported size and the trial-alpha sequence of the search."""
    from sella_amd.linalg import ApproximateHessian
    from sella_amd.optimize.restricted_step import get_restricted_step
    g = load_golden('g8_mis')
    for case in manifest['g8_mis']:
        i = case['id']
        pes = _mis_pes(ApproximateHessian, g, case, i)
        np.testing.assert_array_equal(pes.Ufree, g[f'c{i}_Ufree'])
        rs = get_restricted_step('mis')(pes, case['order'], case['delta'], case['method'], **case['weights'])
        np.testing.assert_array_equal(rs._get_weights(), g[f'c{i}_w'])
        s, smag = rs.get_s()
        ref, ref_alphas = g[f'c{i}_s'], g[f'c{i}_alphas']
        np.testing.assert_allclose(smag, float(g[f'c{i}_smag']), rtol=1e-12)
        m = min(len(rs.alphas), len(ref_alphas))
        # the first trials are pinned tightly; deep in the bisection the bracket decisions sit in the rounding noise of
        # the measure, so later alphas are compared through the step they lead to
        head = min(m, 12)
        np.testing.assert_allclose(rs.alphas[:head], ref_alphas[:head], rtol=1e-7, atol=1e-13, err_msg=str(case))
        if case['method'] == 'rfo' and case['order'] >= 1 and len(ref_alphas) > 60:
            cos = s @ ref / np.linalg.norm(s) / np.linalg.norm(ref)          # see test_golden_restricted_step
            assert cos > 0.9, case
            continue
        assert abs(len(rs.alphas) - len(ref_alphas)) <= 1, case
        np.testing.assert_allclose(s, ref, atol=1e-10 * max(1, np.abs(ref).max()), err_msg=str(case))


def test_oracle_restricted_mis_golden(manifest):
    g = load_golden('g8_mis')
    for case in manifest['g8_mis']:
        i = case['id']
        ro = orc.get_restricted_step('mis')(_mis_pes(orc.QuasiNewtonHessian, g, case, i), case['order'], case['delta'],
                                            case['method'], **case['weights'])
        s, smag = ro.get_s()
        np.testing.assert_allclose(s, g[f'c{i}_s'], atol=1e-9)
        np.testing.assert_allclose(smag, float(g[f'c{i}_smag']), rtol=1e-12)
        np.testing.assert_array_equal(ro.weights(), g[f'c{i}_w'])


def test_user_supplied_families_are_searched_on_the_host(ctx):
    """The one-call device search evaluates the built-in families and measures only: a stepper subclass with its own
    `get_s`, or a restricted-step subclass with its own `cons`, goes through the host search with the same schedule."""
    from conftest import hessian_like
    from sella_amd.linalg import ApproximateHessian
    from sella_amd.optimize.restricted_step import TrustRegion
    from sella_amd.optimize.stepper import QuasiNewton
    n = 24
    A, P, gvec = hessian_like(n, seed=3)
    calls = {'get_s': 0, 'cons': 0}

    class HalfQuasiNewton(QuasiNewton):
        def get_s(self, alpha):
            calls['get_s'] += 1
            s, ds = QuasiNewton.get_s(self, alpha)
            return 0.5 * s, 0.5 * ds

    class InfinityNorm(TrustRegion):
        measure = None

        def cons(self, s, dsda=None):
            calls['cons'] += 1
            i = int(np.abs(s).argmax())
            return abs(s[i]) if dsda is None else (abs(s[i]), np.sign(s[i]) * dsda[i])

    ref_s, ref_mag = TrustRegion(FakePES(ApproximateHessian, P, gvec, 0, 1), 1, 0.05, 'qn').get_s()
    rs = TrustRegion(FakePES(ApproximateHessian, P, gvec, 0, 1), 1, 0.05, HalfQuasiNewton)
    assert not rs._device_search_applies()
    s, mag = rs.get_s()
    assert calls['get_s'] > 1 and abs(np.linalg.norm(s) - 0.05) < 1e-9
    rs2 = InfinityNorm(FakePES(ApproximateHessian, P, gvec, 0, 1), 1, 0.01, 'qn')
    assert not rs2._device_search_applies()
    s2, mag2 = rs2.get_s()
    assert calls['cons'] > 1 and abs(np.abs(s2).max() - 0.01) < 1e-9
    assert TrustRegion(FakePES(ApproximateHessian, P, gvec, 0, 1), 1, 0.05, 'qn')._device_search_applies()
    assert abs(ref_mag - 0.05) < 1e-12 and abs(np.linalg.norm(ref_s) - 0.05) < 1e-9


def test_alpha_trace_matches_oracle(ctx):
    """alpha iterates of the root find, step for step, against the CPU oracle."""
    from conftest import hessian_like
    from sella_amd.linalg import ApproximateHessian
    from sella_amd.optimize.restricted_step import get_restricted_step
    n = 30
    A, P, gvec = hessian_like(n, seed=42)
    for rsname, method in (('tr', 'prfo'), ('ras', 'qn'), ('ras', 'prfo')):
        dev = get_restricted_step(rsname)(FakePES(ApproximateHessian, P, gvec, 0, 1), 1, 0.05, method)
        ref = orc.get_restricted_step(rsname)(FakePES(orc.QuasiNewtonHessian, P, gvec, 0, 1), 1, 0.05, method)
        s, _ = dev.get_s()
        sr, _ = ref.get_s()
        np.testing.assert_allclose(s, sr, atol=1e-10)
        m = min(len(dev.alphas), len(ref.alpha_trace))
        np.testing.assert_allclose(dev.alphas[:m], ref.alpha_trace[:m], rtol=1e-6, atol=1e-12)


@pytest.mark.emu_heavy
def test_batched_bisection_matches_sequential(ctx):
    """The bisection phase evaluated 15 trial alphas per round trip (`rs_batch`, csrc/stepper.hip) visits the same
    alphas and returns the same step as one evaluation per round trip."""
    from conftest import hessian_like
    from sella_amd.linalg import ApproximateHessian
    from sella_amd.optimize.restricted_step import get_restricted_step
    for n, ncons, order, method in ((30, 0, 1, 'prfo'), (45, 3, 1, 'prfo'), (33, 0, 0, 'rfo'), (36, 2, 2, 'prfo')):
        A, P, gvec = hessian_like(n, seed=7 + n, nneg=max(order, 1))
        out = {}
        for flag in (0, 1):
            ctx.set_option('rs_batch', flag)
            rs = get_restricted_step('ras')(FakePES(ApproximateHessian, P, gvec, ncons, 1), order, 0.03, method)
            s, smag = rs.get_s()
            out[flag] = (s, smag, np.array(rs.alphas))
        ctx.set_option('rs_batch', 0 if ctx.backend == 'emu' else 1)          # conftest.make_context's setting
        (s0, m0, a0), (s1, m1, a1) = out[0], out[1]
        assert len(a0) > 20                                      # the schedule did reach its bisection phase
        assert abs(len(a0) - len(a1)) <= 1
        k = min(len(a0), len(a1)) - 2                            # the last levels sit in the rounding noise of val
        np.testing.assert_allclose(a1[:k], a0[:k], rtol=1e-12, atol=0)
        np.testing.assert_allclose(s1, s0, atol=1e-12 * max(1.0, np.abs(s0).max()))
        assert m0 == m1


@pytest.mark.emu_heavy
def test_batched_bisection_all_measures(ctx):
    """The same for every measure of `sella_restricted_step` (component-wise 'mis', Euclidean 'tr' in the output space,
    mass-weighted 'sphere') and a selection basis, straight through the stepper binding."""
    from sella_amd.device import DeviceStepper
    rng = np.random.RandomState(11)
    m = 40
    Q, _ = np.linalg.qr(rng.normal(size=(m, m)))
    lam = np.sort(np.exp(rng.uniform(np.log(0.05), np.log(20.0), m)))
    lam[0] = -0.6
    g = 0.3 * rng.normal(size=m)
    V, Vt = ctx.upload(Q), ctx.upload(Q.T.copy())
    w = rng.uniform(0.5, 2.0, size=m)
    d1 = 1e-3 * rng.normal(size=m)           # |d1 w| below the radius: the sphere is reachable as alpha -> 0
    sel = np.sort(rng.choice(60, size=m, replace=False)).astype(np.int32)
    wf, sc = rng.uniform(0.5, 2.0, size=60), np.zeros(60)
    cases = [dict(cons='mis', w=w), dict(cons='tr'), dict(cons='sphere', w=w, d1=d1),
             dict(cons='ras', sel=sel, nfull=60), dict(cons='mis', w=wf, sel=sel, nfull=60, scons=sc)]
    for kw in cases:
        out = {}
        for flag in (0, 1):
            ctx.set_option('rs_batch', flag)
            st = DeviceStepper(ctx, 'prfo', V, Vt, lam, g, 1)
            out[flag] = st.restricted_step(delta=0.02, alpha0=1.0, alphamin=0.0, alphamax=1.0, slope=1.0,
                                           newton_safe=False, tol=1e-15, **kw)
        ctx.set_option('rs_batch', 0 if ctx.backend == 'emu' else 1)
        (s0, v0, a0), (s1, v1, a1) = out[0], out[1]
        assert len(a0) > 20 and abs(len(a0) - len(a1)) <= 1, kw['cons']
        k = min(len(a0), len(a1)) - 2
        np.testing.assert_allclose(a1[:k], a0[:k], rtol=1e-12, atol=0, err_msg=kw['cons'])
        np.testing.assert_allclose(s1, s0, atol=1e-12 * max(1.0, np.abs(s0).max()), err_msg=kw['cons'])
        assert v0 == v1 == 0.02


def test_registry_and_errors(ctx):
    from sella_amd.optimize.restricted_step import (MaxInternalStep, RestrictedAtomicStep, TrustRegion,
                                                    get_restricted_step)
    assert get_restricted_step('trust-radius') is TrustRegion
    assert get_restricted_step('ras') is RestrictedAtomicStep
    assert get_restricted_step('mis') is MaxInternalStep
    with pytest.raises(ValueError):
        get_restricted_step('nope')
    with pytest.raises(ValueError):
        from sella_amd.linalg import ApproximateHessian
        MaxInternalStep(FakePES(ApproximateHessian, np.eye(6), np.ones(6)), 0, 0.1)


def test_oracle_restricted_step_golden(manifest):
    """The oracle's own restricted step against the reference goldens (pins the oracle, CPU)."""
    g = load_golden('g8_restricted_step')
    for case in manifest['g8_restricted_step']:
        i = case['id']
        pes = FakePES(orc.QuasiNewtonHessian, g[f'c{i}_H'], g[f'c{i}_g'], case['ncons'], seed=i)
        s, smag = orc.get_restricted_step(case['rs'])(pes, case['order'], case['delta'], case['method']).get_s()
        np.testing.assert_allclose(s, g[f'c{i}_s'], atol=1e-9)
