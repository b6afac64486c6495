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


def _slab(size):
    from sella_amd.atoms import fcc111
    slab = fcc111('Cu', size, vacuum=6.0)
    rng = np.random.RandomState(3)
    slab.positions += 0.05 * rng.normal(size=slab.positions.shape)
    return slab


def test_internal_coordinates_container(ctx):
    """B-matrix, D(v) and ldot of a periodic slab's bonds + angles: finite-difference consistency of
    q(x) -> B and B -> D(v) through the periodic images (config 3 of BASELINE.json in small)."""
    from sella_amd.internal import InternalCoordinates, angles_from_bonds, neighbour_bonds
    slab = _slab((3, 3, 2))
    bonds, bncv = neighbour_bonds(slab, 1.25 * 3.61 / np.sqrt(2))
    assert len(bonds) == 9 * 2 * 3 + 9 * 3                     # 6 in-plane / 2 per atom + 3 interlayer per atom
    angles, ancv = angles_from_bonds(bonds, bncv)
    ic = InternalCoordinates(slab, bonds=bonds, angles=angles, bond_ncvecs=bncv, angle_ncvecs=ancv)
    q0 = ic.calc()
    nb = len(bonds)
    np.testing.assert_allclose(q0[:nb], 3.61 / np.sqrt(2), atol=0.25)          # nearest-neighbour lengths
    assert np.all((q0[nb:] > 0.8) & (q0[nb:] < np.pi + 1e-9))                  # 60 / 90 / 120 / 180 degree families
    B = ic.jacobian()
    x0 = slab.positions.copy()
    rng = np.random.RandomState(8)
    v = rng.normal(size=x0.size)
    h = 1e-6
    slab.positions = x0 + h * v.reshape(-1, 3)
    qp, Bp = ic.calc(), ic.jacobian()
    slab.positions = x0 - h * v.reshape(-1, 3)
    qm, Bm = ic.calc(), ic.jacobian()
    slab.positions = x0
    np.testing.assert_allclose(B @ v, ic.wrap(qp - qm) / (2 * h), rtol=1e-6, atol=1e-7)
    D = ic.hessian_rdot(v)
    np.testing.assert_allclose(D, (Bp - Bm) / (2 * h), rtol=1e-5, atol=1e-6)
    # ldot(w) v == D(v)^T w for any w (linalg.py:601-640)
    w = rng.normal(size=ic.nint)
    np.testing.assert_allclose(ic.hessian().ldot(w) @ v, D.T @ w, rtol=1e-10, atol=1e-10)
    # translation of the whole slab leaves every internal coordinate unchanged: B t = 0
    t = np.tile([0.3, -0.2, 0.5], len(slab))
    np.testing.assert_allclose(B @ t, 0.0, atol=1e-12)


def _g11_blocks(g, i, sizes, order=None):
    """Golden blocks as the product's stacks take them: one group per run of equal-sized coordinates in `order`."""
    order = list(range(len(sizes))) if order is None else list(order)
    jb, hb = [], []
    for k in order:
        idx = g[f'c{i}_idx{k}']
        dofs = (3 * idx[:, None] + np.arange(3)[None, :]).reshape(1, -1)
        m = dofs.shape[1]
        jb.append((dofs, g[f'c{i}_g{k}'].reshape(1, m)))
        hb.append((dofs, g[f'c{i}_h{k}'].reshape(1, m, m)))
    # merge neighbours of equal size into one batched group (the way the classes are fed in production)
    def merge(blocks):
        out = []
        for dofs, val in blocks:
            if out and out[-1][0].shape[1] == dofs.shape[1]:
                out[-1] = (np.vstack((out[-1][0], dofs)), np.concatenate((out[-1][1], val)))
            else:
                out.append((dofs, val))
        return out
    return merge(jb), merge(hb)


def test_golden_sparse_internal(manifest):
    """a17: the scatter / contraction layer of the internal-coordinate Jacobian and Hessians against the reference's
    SparseInternalJacobian / SparseInternalHessian(s) (sella/linalg.py:362-646; fixture g11 generated by
    oracle/make_golden.py from the real classes): asarray, matvec, rmatvec; ldot, rdot, ddot, asarray — fed in the
    reference's mixed order and grouped by size as in production (bonds, angles, dihedrals)."""
    from conftest import load_golden
    from sella_amd.internal import _HessianStack, _JacobianStack
    g = load_golden('g11_sparse_internal')
    for case in manifest['g11_sparse_internal']:
        i, natoms, sizes = case['id'], case['natoms'], case['sizes']
        x, u, y = g[f'c{i}_x'], g[f'c{i}_u'], g[f'c{i}_y']
        for order in (None, np.argsort(sizes, kind='stable')):
            perm = np.arange(len(sizes)) if order is None else np.asarray(order)
            jb, hb = _g11_blocks(g, i, sizes, order)
            J = _JacobianStack(3 * natoms, jb)
            H = _HessianStack(3 * natoms, hb)
            np.testing.assert_allclose(J.asarray(), g[f'c{i}_J'][perm], atol=1e-14)
            np.testing.assert_allclose(J.matvec(x), g[f'c{i}_Jx'][perm], atol=1e-13)
            np.testing.assert_allclose(J.rmatvec(y[perm]), g[f'c{i}_JTy'], atol=1e-13)
            np.testing.assert_allclose(H.asarray(), g[f'c{i}_Hall'][perm], atol=1e-14)
            np.testing.assert_allclose(H.ldot(y[perm]), g[f'c{i}_ldot'], atol=1e-13)
            np.testing.assert_allclose(H.rdot(x), g[f'c{i}_rdot'][perm], atol=1e-13)
            np.testing.assert_allclose(H.ddot(u, x), g[f'c{i}_ddot'][perm], atol=1e-13)
        single = _HessianStack(3 * natoms, _g11_blocks(g, i, sizes[:1])[1])
        np.testing.assert_allclose(single.asarray()[0], g[f'c{i}_H0'], atol=1e-14)
        np.testing.assert_allclose(single.rdot(x)[0], g[f'c{i}_H0x'], atol=1e-13)


def test_container_contractions_agree_with_the_stacks(ctx):
    """The device-evaluated coordinate data through the same pinned scatter layer: `hessian().rdot(v)` (dense
    per-coordinate Hessians) equals `hessian_rdot(v)` (Hessian-vector kernel), `ddot` and `asarray` agree with
    `ldot`, and the CSR B-matrix equals the dense one — incl. a bond between an atom and its own periodic image."""
    from sella_amd.atoms import Atoms
    from sella_amd.internal import InternalCoordinates, angles_from_bonds, neighbour_bonds
    slab = _slab((2, 2, 2))
    bonds, bncv = neighbour_bonds(slab, 1.25 * 3.61 / np.sqrt(2))
    angles, ancv = angles_from_bonds(bonds, bncv)
    ic = InternalCoordinates(slab, bonds=bonds, angles=angles[:40], bond_ncvecs=bncv, angle_ncvecs=ancv[:40])
    rng = np.random.RandomState(4)
    v, u, w = rng.normal(size=ic.ndof), rng.normal(size=ic.ndof), rng.normal(size=ic.nint)
    H = ic.hessian()
    np.testing.assert_allclose(H.rdot(v), ic.hessian_rdot(v), atol=1e-11)
    np.testing.assert_allclose(H.ddot(u, v), ic.hessian_rdot(v) @ u, atol=1e-11)
    np.testing.assert_allclose(np.einsum('i,ijk->jk', w, H.asarray()), H.ldot(w), atol=1e-12)
    np.testing.assert_allclose(ic.jacobian_csr().toarray(), ic.jacobian(), atol=1e-15)
    # one atom in a small periodic cell bonded to its own image: both ends scatter onto the same columns
    at = Atoms(['Cu'], np.array([[0.1, 0.2, 0.3]]), cell=np.diag([2.5, 9.0, 9.0]), pbc=True)
    one = InternalCoordinates(at, bonds=np.array([[0, 0]]), bond_ncvecs=np.array([[[1.0, 0.0, 0.0]]]))
    np.testing.assert_allclose(one.calc(), [2.5], atol=1e-14)
    np.testing.assert_allclose(one.jacobian(), np.zeros((1, 3)), atol=1e-14)          # -e_x + e_x
    np.testing.assert_allclose(one.jacobian_csr().toarray(), np.zeros((1, 3)), atol=1e-14)


def test_uploads_survive_the_pinned_ring_wrapping(ctx):
    """Host-to-device copies go through an 8 MB pinned ring that is rewound behind a synchronisation
    (csrc/context.hip `h2d_async`): far more than 8 MB of distinct payloads, every result still right."""
    rng = np.random.RandomState(5)
    nc = 30000                                              # 30000 bonds x 6 doubles = 1.4 MB per call
    for it in range(14):
        pos = rng.normal(size=(nc, 2, 3))
        q, grad, _, _ = ctx.internals_eval(pos)
        d = pos[:, 1] - pos[:, 0]
        r = np.linalg.norm(d, axis=1)
        np.testing.assert_allclose(q, r, rtol=1e-14)
        np.testing.assert_allclose(grad[:, 1], d / r[:, None], rtol=1e-13, atol=1e-15)
