"""IRC pieces: the step family and trust sphere against golden vectors of the real reference
(stepper.py:99-111, restricted_step.py:145-158 — importable, so pinned), and the driver
(sella/optimize/irc.py, needs ASE: property test) walking downhill from a saddle to both minima."""
import numpy as np
import pytest

from conftest import load_golden


class FakePES:
    int = None
    n_cell_dof = 0

    def __init__(self, H, g, Ufree, scons):
        from sella_amd.linalg import ApproximateHessian
        n = len(g)
        self.H = ApproximateHessian(n, n, H)
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
        from sella_amd.linalg import ApproximateHessian
        return ApproximateHessian(U.shape[1], 0, U.T @ self.H.B @ U)


def test_golden_irc_steps(ctx, manifest):
    from sella_amd.linalg import ApproximateHessian
    from sella_amd.optimize.restricted_step import IRCTrustRegion
    from sella_amd.optimize.stepper import QuasiNewtonIRC
    g10 = load_golden('g10_irc')
    for case in manifest['g10_irc']:
        i = case['id']
        H, g, d1, sqrtm = g10[f'c{i}_H'], g10[f'c{i}_g'], g10[f'c{i}_d1'], g10[f'c{i}_sqrtm']
        n = len(g)
        st = QuasiNewtonIRC(g, ApproximateHessian(n, 0, H), 0, d1=d1)
        for k, alpha in enumerate(case['alphas']):
            s, dsda = st.get_s(alpha)
            scale = max(1.0, np.abs(g10[f'c{i}_a{k}_s']).max())
            np.testing.assert_allclose(s, g10[f'c{i}_a{k}_s'], atol=1e-10 * scale, rtol=0)
            np.testing.assert_allclose(dsda, g10[f'c{i}_a{k}_dsda'], atol=1e-9 * max(1.0, np.abs(g10[f'c{i}_a{k}_dsda']).max()),
                                       rtol=0)
        pes = FakePES(H, g, g10[f'c{i}_Ufree'], g10[f'c{i}_scons'])
        s, smag = IRCTrustRegion(pes, 0, case['delta'], method=QuasiNewtonIRC, sqrtm=sqrtm, d1=d1.copy(),
                                 W=np.diag(1.0 / sqrtm)).get_s()
        assert smag == pytest.approx(float(g10[f'c{i}_smag']), abs=1e-12)
        np.testing.assert_allclose(s, g10[f'c{i}_s'], atol=1e-8 * max(1.0, np.abs(g10[f'c{i}_s']).max()), rtol=0)
        # the defining property: the accumulated displacement sits on the mass-weighted sphere
        assert np.linalg.norm((s + d1) * sqrtm) == pytest.approx(case['delta'], abs=1e-9)


@pytest.mark.emu_heavy
def test_irc_from_rhombus_saddle(ctx):
    """4-atom Morse cluster: the planar rhombus is the first-order saddle between two tetrahedra; the
    IRC must run downhill from it in both directions and end on minima of the same energy."""
    from sella_amd import IRC, Sella
    from sella_amd.atoms import Atoms, MorseCluster
    from sella_amd.internal import Constraints
    r0 = 1.45
    pos = np.array([[0, 0, 0], [r0, 0, 0], [0.5 * r0, 0.866 * r0, 0.05], [0.5 * r0, -0.866 * r0, 0.05]])
    at = Atoms(['C'] * 4, pos)
    at.calc = MorseCluster(D=1.0, a=1.2, r0=r0)
    ts = Sella(at, order=1, logfile=None, eta=1e-5, gamma=1e-3, constraints=Constraints(at), proj_trans=False)
    assert ts.run(fmax=1e-4, steps=200)
    e_ts = at.get_potential_energy()
    x_ts = at.positions.copy()
    ends = []
    for direction in ('forward', 'reverse'):
        at.positions = x_ts.copy()
        irc = IRC(at, logfile=None, dx=0.1, eta=1e-5, gamma=1e-3, keep_going=True)
        energies = []
        irc.attach(lambda: energies.append(at.get_potential_energy()))
        irc.run(fmax=5e-3, steps=25 if ctx.backend == 'emu' else 60, direction=direction)
        assert len(energies) > 3
        assert energies[-1] < e_ts - 0.05                       # went downhill a finite amount
        assert np.all(np.diff(energies[1:]) < 1e-6)             # monotonically
        ends.append((at.get_potential_energy(), at.positions.copy()))
    assert abs(ends[0][0] - ends[1][0]) < 5e-3                  # the two tetrahedra are equivalent
    # the two branches fold the rhombus to opposite sides
    fold = [np.cross(p[1] - p[0], p[2] - p[0]) @ (p[3] - p[0]) for _, p in ends]
    assert fold[0] * fold[1] < 0
