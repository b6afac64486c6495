"""Internal-coordinate primitives on the device (csrc/internals.hip) against the CPU restatement
(oracle/sella_oracle/internals.py) and against finite differences — the reference's own test of these
derivatives is FD consistency at rtol = atol = 1e-7 (tests/internal/test_get_internal.py:26-57)."""
import numpy as np
import pytest

from oracle.sella_oracle import internals as orc

KINDS = {'bonds': 2, 'angles': 3, 'dihedrals': 4}


def random_coords(kind, nc, rng, periodic=True):
    na = KINDS[kind]
    pos = rng.normal(size=(nc, na, 3)) * 1.5 + np.arange(na)[None, :, None] * 0.7
    tvec = rng.normal(size=(nc, na - 1, 3)) * (0.4 if periodic else 0.0)
    return pos, tvec


@pytest.mark.parametrize('kind', list(KINDS))
def test_value_grad_hessian_match_oracle(ctx, kind):
    rng = np.random.RandomState(11)
    nc = 40 if ctx.backend == 'emu' else 5000
    pos, tvec = random_coords(kind, nc, rng)
    tan = rng.normal(size=pos.shape)
    q, g, hv, H = ctx.internals_eval(pos, tvec, tan, hessian=True)
    q0, g0, H0 = orc.evaluate_kind(kind, pos, tvec)
    np.testing.assert_allclose(q, orc.value_only(kind, pos, tvec), rtol=0, atol=1e-14)
    np.testing.assert_allclose(q, q0, rtol=0, atol=1e-14)
    scale = max(1.0, np.abs(H0).max())
    np.testing.assert_allclose(g, g0, rtol=0, atol=1e-12 * max(1.0, np.abs(g0).max()))
    np.testing.assert_allclose(H, H0, rtol=0, atol=1e-11 * scale)
    na = KINDS[kind]
    hv0 = np.einsum('ikl,il->ik', H0.reshape(nc, 3 * na, 3 * na), tan.reshape(nc, 3 * na)).reshape(nc, na, 3)
    np.testing.assert_allclose(hv, hv0, rtol=0, atol=1e-11 * scale * 10)
    # the Hessian blocks are symmetric
    Hm = H.reshape(nc, 3 * na, 3 * na)
    np.testing.assert_allclose(Hm, Hm.transpose(0, 2, 1), rtol=0, atol=1e-11 * scale)


@pytest.mark.parametrize('kind', list(KINDS))
def test_finite_difference_consistency(ctx, kind):
    """tests/internal/test_get_internal.py:26-57 re-stated on the device kernel."""
    rng = np.random.RandomState(12)
    nc = 12
    pos, tvec = random_coords(kind, nc, rng)
    na = KINDS[kind]
    q, g, _, H = ctx.internals_eval(pos, tvec, hessian=True)
    h = 1e-5
    gfd = np.zeros_like(g)
    Hfd = np.zeros((nc, 3 * na, 3 * na))
    for e in range(3 * na):
        dp = np.zeros((nc, 3 * na))
        dp[:, e] = h
        dp = dp.reshape(nc, na, 3)
        qp, gp, _, _ = ctx.internals_eval(pos + dp, tvec)
        qm, gm, _, _ = ctx.internals_eval(pos - dp, tvec)
        dq = qp - qm
        if kind == 'dihedrals':
            dq = (dq + np.pi) % (2 * np.pi) - np.pi
        gfd.reshape(nc, 3 * na)[:, e] = dq / (2 * h)
        Hfd[:, :, e] = ((gp - gm) / (2 * h)).reshape(nc, 3 * na)
    np.testing.assert_allclose(g, gfd, rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(H.reshape(nc, 3 * na, 3 * na), Hfd, rtol=1e-6, atol=1e-6)


def test_edge_cases(ctx):
    # no shift vectors given == zero shift vectors
    rng = np.random.RandomState(13)
    pos, _ = random_coords('angles', 5, rng, periodic=False)
    a = ctx.internals_eval(pos, None)
    b = ctx.internals_eval(pos, np.zeros((5, 2, 3)))
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    # right angle, straight dihedral conventions of internal.py:63-80
    ang = np.array([[[1.0, 0, 0], [0, 0, 0], [0, 1.0, 0]]])
    np.testing.assert_allclose(ctx.internals_eval(ang)[0], [np.pi / 2], atol=1e-15)
    trans = np.array([[[1.0, 1, 0], [1.0, 0, 0], [0, 0, 0], [0, -1.0, 0]]])      # trans: |phi| = pi
    cis = np.array([[[1.0, 1, 0], [1.0, 0, 0], [0, 0, 0], [0, 1.0, 0]]])         # cis: phi = 0
    np.testing.assert_allclose(np.abs(ctx.internals_eval(trans)[0]), [np.pi], atol=1e-15)
    np.testing.assert_allclose(ctx.internals_eval(cis)[0], [0.0], atol=1e-15)
    # empty batch
    q, g, _, _ = ctx.internals_eval(np.zeros((0, 2, 3)))
    assert q.shape == (0,) and g.shape == (0, 2, 3)
    with pytest.raises(Exception):
        ctx.internals_eval(np.zeros((1, 5, 3)))
