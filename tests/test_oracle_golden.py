"""The CPU oracle (oracle/sella_oracle) against every golden vector generated from the real
reference (oracle/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest

import oracle.sella_oracle as orc
from conftest import load_golden


def colsign(V, Vref):
    s = np.sign(np.sum(V * Vref, axis=0))
    s[s == 0] = 1
    return V * s


def test_mgs(manifest):
    g = load_golden('g3_mgs')
    for case in manifest['g3_mgs']:
        i = case['id']
        Y = g[f'c{i}_Y'] if case['hasY'] else None
        out = orc.modified_gram_schmidt(g[f'c{i}_X'], Y)
        np.testing.assert_allclose(out, g[f'c{i}_out'], atol=1e-12, rtol=0)


def test_symmetrize(manifest):
    g = load_golden('g4_symmetrize')
    for case in manifest['g4_symmetrize']:
        i = case['id']
        symm = None if case['symm'] < 0 else case['symm']
        out = orc.symmetrize_Y(g[f'c{i}_S'], g[f'c{i}_Y'], symm)
        np.testing.assert_allclose(out, g[f'c{i}_out'], atol=1e-11, rtol=1e-11)


def test_update_H(manifest):
    g = load_golden('g5_update_h')
    for case in manifest['g5_update_h']:
        i = case['id']
        B = None if case['B'] == 'none' else g[f'c{i}_B']
        out = orc.update_H(B, g[f'c{i}_S'], g[f'c{i}_Y'], method=case['method'], symm=case['symm'])
        ref = g[f'c{i}_out']
        np.testing.assert_allclose(out, ref, atol=1e-9 * np.abs(ref).max(), rtol=0)
    B, s, y = g['oned_B'], g['oned_s'], g['oned_y']
    np.testing.assert_allclose(orc.update_H(B, s, y), g['oned_out'], atol=1e-10)
    assert orc.update_H(B, s / 1e12, y / 1e12) is B


def test_expand(manifest):
    g = load_golden('g2_expand')
    for case in manifest['g2_expand']:
        i = case['id']
        V, lams = g[f'c{i}_V'], g[f'c{i}_lams']
        k = V.shape[1]
        t = orc.correction(V, g[f'c{i}_Y'], g[f'c{i}_P'], np.eye(V.shape[0]), lams, np.eye(k),
                           lams[case['seeking']], case['method'], case['seeking'])
        ref = g[f'c{i}_t']
        np.testing.assert_allclose(t, ref, atol=1e-9 * np.abs(ref).max(), rtol=0)


def test_davidson(manifest):
    g = load_golden('g1_davidson')
    for case in manifest['g1_davidson']:
        i = case['id']
        v0 = g[f'c{i}_v0'] if case['start'] == 'v0' else None
        maxiter = None if case['maxiter'] < 0 else case['maxiter']
        lams, V, AV = orc.rayleigh_ritz(g[f'c{i}_A'], case['gamma'], g[f'c{i}_P'], v0=v0,
                                        method=case['method'], maxiter=maxiter)
        assert V.shape[1] == case['k']
        strict = case['method'] in ('jd0', 'jd0_alt', 'lanczos')
        np.testing.assert_allclose(lams[0], g[f'c{i}_lams'][0], atol=1e-9 if strict else 1e-5)
        np.testing.assert_allclose(colsign(V, g[f'c{i}_V']), g[f'c{i}_V'],
                                   atol=1e-6 if strict else 1e-2)


def test_approx_hessian(manifest):
    g = load_golden('g6_approx_hessian')
    case = manifest['g6_approx_hessian'][0]
    n = case['n']
    H = orc.QuasiNewtonHessian(n, n, None)
    for step in range(len(case['seq'])):
        H.update(g[f's{step}_dx'], g[f's{step}_dg'])
        np.testing.assert_allclose(H.B, g[f's{step}_B'], atol=1e-10)
    np.testing.assert_allclose(H.project(g['proj_U']).B, g['proj_B'], atol=1e-12)
    np.testing.assert_allclose(H.evals, g['evals'], atol=1e-10)
    np.testing.assert_allclose((H + g['add_M']).B, g['add_B'], atol=1e-12)


def test_steppers(manifest):
    g = load_golden('g7_steppers')
    for case in manifest['g7_steppers']:
        i = case['id']
        n = len(g[f'c{i}_g'])
        st = orc.get_stepper(case['name'])(g[f'c{i}_g'], orc.QuasiNewtonHessian(n, 0, g[f'c{i}_H']),
                                           case['order'])
        for a in range(case['nalpha']):
            s, ds = st.get_s(float(g[f'c{i}_a{a}_alpha']))
            np.testing.assert_allclose(s, g[f'c{i}_a{a}_s'], atol=1e-9 * max(1, np.abs(s).max()))
            np.testing.assert_allclose(ds, g[f'c{i}_a{a}_dsda'], atol=1e-7 * max(1, np.abs(ds).max()))


def test_numerical_hessian(manifest):
    g = load_golden('g9_numhess')
    for case in manifest['g9_numhess']:
        i = case['id']
        A, U4 = g[f'c{i}_A'], g[f'c{i}_U4']
        c3, c4 = case['c3'], case['c4']

        def f(x, A=A, U4=U4):
            p = U4 @ x
            return (0.5 * x @ A @ x + c3 / 3 * np.sum(p ** 3) + c4 / 4 * np.sum(p ** 4),
                    A @ x + U4.T @ (c3 * p ** 2 + c4 * p ** 3))
        U = g[f'c{i}_Uproj'] if case['sub'] > 0 else None
        H = orc.FiniteDifferenceHessian(f, g[f'c{i}_x'], g[f'c{i}_g'], 1e-6, case['threepoint'], U)
        np.testing.assert_allclose(H.dot(g[f'c{i}_M']), g[f'c{i}_out'], atol=1e-12)
        np.testing.assert_allclose(H.Vs, g[f'c{i}_Vs'], atol=1e-12)
        np.testing.assert_allclose(H.AVs, g[f'c{i}_AVs'], atol=1e-12)


def test_numerical_hessian_on_the_model_pes(manifest):
    """The oracle's finite-difference operator on the model PES the library's calculator implements, incl. a selection
    basis (`g12_numhess_model`): the vectors the library-side operator is pinned with (tests/test_library_calculator.py)."""
    g = load_golden('g12_numhess_model')
    for case in manifest['g12_numhess_model']:
        i, n = case['id'], case['n']
        A, Uc, c = g[f'c{i}_A'], g[f'c{i}_U'], case['c']

        def f(x, A=A, Uc=Uc, c=c):
            p = Uc @ x
            return 0.5 * x @ (A @ x) + c / 3.0 * np.sum(p ** 3), A @ x + Uc.T @ (c * p ** 2)
        U = np.eye(n)[:, g[f'c{i}_free']] if case['nfree'] > 0 else None
        H = orc.FiniteDifferenceHessian(f, g[f'c{i}_x'], g[f'c{i}_g'], case['eta'], case['threepoint'], U)
        np.testing.assert_allclose(H.dot(g[f'c{i}_M']), g[f'c{i}_out'], atol=1e-10)
        np.testing.assert_allclose(H.Vs, g[f'c{i}_Vs'], atol=1e-12)
        np.testing.assert_allclose(H.AVs, g[f'c{i}_AVs'], atol=1e-10)


class _OraclePES:
    int = None
    n_cell_dof = 0

    def __init__(self, H, g, Ufree, scons):
        n = len(g)
        self.H = orc.QuasiNewtonHessian(n, n, H)
        self.g, self.Ufree, self.scons = g, Ufree, scons

    def get_g(self):
        return self.g.copy()

    def get_scons(self):
        return self.scons.copy()

    def get_H(self):
        return self.H

    def get_Unred(self):
        return np.eye(len(self.g))

    def get_Ufree(self):
        return self.Ufree

    def get_HL_projected(self, U):
        return orc.QuasiNewtonHessian(U.shape[1], 0, U.T @ self.H.B @ U)


def test_restricted_step(manifest):
    g = load_golden('g8_restricted_step')
    for case in manifest['g8_restricted_step']:
        i = case['id']
        pes = _OraclePES(g[f'c{i}_H'], g[f'c{i}_g'], g[f'c{i}_Ufree'], g[f'c{i}_scons'])
        rs = orc.get_restricted_step(case['rs'])(pes, case['order'], case['delta'], case['method'])
        s, smag = rs.get_s()
        assert smag == pytest.approx(float(g[f'c{i}_smag']), abs=1e-12)
        np.testing.assert_allclose(s, g[f'c{i}_s'], atol=1e-9 * max(1, np.abs(s).max()))
        assert len(rs.alpha_trace) == int(g[f'c{i}_nalpha'])


def test_irc_steps(manifest):
    g = load_golden('g10_irc')
    for case in manifest['g10_irc']:
        i = case['id']
        H, gr, d1, sqrtm = g[f'c{i}_H'], g[f'c{i}_g'], g[f'c{i}_d1'], g[f'c{i}_sqrtm']
        st = orc.QuasiNewtonIRCStep(gr, orc.QuasiNewtonHessian(len(gr), 0, H), 0, d1=d1)
        for k, alpha in enumerate(case['alphas']):
            s, ds = st.get_s(alpha)
            np.testing.assert_allclose(s, g[f'c{i}_a{k}_s'], atol=1e-10 * max(1, np.abs(s).max()))
            np.testing.assert_allclose(ds, g[f'c{i}_a{k}_dsda'], atol=1e-10 * max(1, np.abs(ds).max()))
        pes = _OraclePES(H, gr, g[f'c{i}_Ufree'], g[f'c{i}_scons'])
        s, smag = orc.IRCTrustRegionStep(pes, 0, case['delta'], method=orc.QuasiNewtonIRCStep, sqrtm=sqrtm,
                                         d1=d1.copy(), W=np.diag(1.0 / sqrtm)).get_s()
        assert smag == pytest.approx(float(g[f'c{i}_smag']), abs=1e-12)
        np.testing.assert_allclose(s, g[f'c{i}_s'], atol=1e-9 * max(1, np.abs(s).max()))


def test_sparse_internal(manifest):
    """oracle/sella_oracle/sparse_internal.py against the fixtures from the reference's sparse containers
    (sella/linalg.py:362-646)."""
    from oracle.sella_oracle import sparse_internal as spo
    g = load_golden('g11_sparse_internal')
    for case in manifest['g11_sparse_internal']:
        i, natoms, sizes = case['id'], case['natoms'], case['sizes']
        idx = [g[f'c{i}_idx{k}'] for k in range(len(sizes))]
        gv = [g[f'c{i}_g{k}'] for k in range(len(sizes))]
        hv = [g[f'c{i}_h{k}'] for k in range(len(sizes))]
        x, u, y = g[f'c{i}_x'], g[f'c{i}_u'], g[f'c{i}_y']
        np.testing.assert_allclose(spo.jacobian_dense(natoms, idx, gv), g[f'c{i}_J'], atol=1e-14)
        np.testing.assert_allclose(spo.jacobian_matvec(natoms, idx, gv, x), g[f'c{i}_Jx'], atol=1e-13)
        np.testing.assert_allclose(spo.jacobian_rmatvec(natoms, idx, gv, y), g[f'c{i}_JTy'], atol=1e-13)
        np.testing.assert_allclose(spo.hessians_ldot(natoms, idx, hv, y), g[f'c{i}_ldot'], atol=1e-13)
        np.testing.assert_allclose(spo.hessians_rdot(natoms, idx, hv, x), g[f'c{i}_rdot'], atol=1e-13)
        np.testing.assert_allclose(spo.hessians_ddot(natoms, idx, hv, u, x), g[f'c{i}_ddot'], atol=1e-13)
