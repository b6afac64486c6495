"""Device symmetric eigensolver (tridiagonalisation + divide & conquer + compact-WY) against
LAPACK: eigenvalues, residuals and orthogonality, on Hessian-like spectra and on the degenerate
spectra an approximate Hessian really has (scaled identity + low rank)."""
import numpy as np
import pytest

from conftest import load_golden


def check(ctx, A, tol=5e-13):
    n = A.shape[0]
    w, V, Vt = ctx.eigh(ctx.upload(A))
    Vn, Vtn = V.numpy(), Vt.numpy()
    wr = np.linalg.eigvalsh(A)
    scale = max(1.0, np.abs(wr).max())
    assert np.all(np.diff(w) >= 0)
    np.testing.assert_allclose(w, wr, atol=tol * scale * n ** 0.5)
    assert np.abs(A @ Vn - Vn * w).max() <= tol * scale * n
    assert np.abs(Vn.T @ Vn - np.eye(n)).max() <= tol * n
    np.testing.assert_array_equal(Vtn.T, Vn)
    return w


def cases(n, rng):
    Q = np.linalg.qr(rng.normal(size=(n, n)))[0]
    lam = np.exp(rng.uniform(np.log(.05), np.log(50), n))
    lam[0] = -1
    u = rng.normal(size=(n, 3))
    A = rng.normal(size=(n, n))
    yield 'random', A + A.T
    yield 'hessian-like', (Q * lam) @ Q.T
    yield 'scaled identity', 3.0 * np.eye(n)
    yield 'identity + low rank', 2.5 * np.eye(n) + u @ u.T - 0.3 * np.outer(u[:, 0] + 1, u[:, 0] + 1)
    yield 'clusters', (Q * np.repeat(np.arange(1, n // 8 + 2), 8)[:n].astype(float)) @ Q.T
    yield 'diagonal', np.diag(rng.normal(size=n))
    T = np.diag(rng.normal(size=n)) + np.diag(rng.normal(size=n - 1), 1)
    yield 'tridiagonal', T + T.T
    yield 'zero', np.zeros((n, n))


def test_small_sizes(ctx):
    rng = np.random.RandomState(0)
    for leaf in (4, 32):
        ctx.set_option('eigh_leaf', leaf)
        for n in (1, 2, 3, 5, 17, 33, 70):
            A = rng.normal(size=(n, n))
            check(ctx, A + A.T)
    ctx.set_option('eigh_leaf', 16)


def test_panel_widths_and_variants(ctx):
    """Every panel width runs the same algebra (specialised row kernels up to 16 columns, the generic
    loop beyond), and the VALU back-transformation agrees with the matrix-core one."""
    rng = np.random.RandomState(4)
    n = 70
    A = rng.normal(size=(n, n))
    A = A + A.T
    ref = None
    try:
        for nb, wy in ((16, 1), (4, 1), (24, 1), (64, 1), (16, 0)):
            ctx.set_option('eigh_nb', nb)
            ctx.set_option('eigh_wy_mfma', wy)
            w = check(ctx, A)
            if ref is None:
                ref = w
            np.testing.assert_allclose(w, ref, atol=1e-12 * np.abs(ref).max())
        # back-transformation with 32 rows per workgroup (two row tiles), 4 and 8 wavefronts; 16 rows with 8 / 16 wavefronts
        for rows, waves in ((32, 4), (32, 8), (16, 8), (16, 16)):
            ctx.set_option('eigh_wy_rows', rows)
            ctx.set_option('eigh_wy_waves', waves)
            np.testing.assert_allclose(check(ctx, A), ref, atol=1e-12 * np.abs(ref).max())
        # 64 reflectors per compact-WY block (the default from n = 4096 on): full blocks, a ragged last block, one block
        ctx.set_option('eigh_wy_rows', 16)
        ctx.set_option('eigh_wy_waves', 4)
        ctx.set_option('eigh_wy_nb64_min', 1)
        np.testing.assert_allclose(check(ctx, A), ref, atol=1e-12 * np.abs(ref).max())
        for n2 in ((40, 150) if ctx.backend == 'emu' else (40, 130, 150, 515)):
            A2 = rng.normal(size=(n2, n2))
            check(ctx, A2 + A2.T)
    finally:
        ctx.set_option('eigh_nb', 16)
        ctx.set_option('eigh_wy_mfma', 1)
        ctx.set_option('eigh_wy_rows', 16)
        ctx.set_option('eigh_wy_waves', 4)
        ctx.set_option('eigh_wy_nb64_min', 2560)


def test_symmetric_aware_trailing_matvec(ctx):
    """Trailing blocks above `eigh_symv_min` rows read the upper triangle only (tiles of 64 x 256 on an absolute grid,
    partial sums added in a fixed order): same tridiagonal matrix up to rounding, whatever the threshold, for sizes
    that put the diagonal, the trailing origin and the matrix edge at every position inside a tile."""
    rng = np.random.RandomState(11)
    try:
        sizes = ((70, (1,)), (257, (1, 100)), (330, (1, 200)), (515, (1,)), (1100, (1, 600)))
        if ctx.backend == 'emu':
            sizes = ((70, (1, 40)), (67, (1,)), (258, (1,)))                        # the emulator runs fibre by fibre
        for n, thresholds in sizes:
            A = rng.normal(size=(n, n))
            A = A + A.T
            if ctx.backend == 'emu' and n > 100:
                ctx.set_option('eigh_symv_min', 1)
                check(ctx, A)                                             # against LAPACK only
                continue
            ctx.set_option('eigh_symv_min', 0)
            ref = check(ctx, A)
            for thr in thresholds:
                ctx.set_option('eigh_symv_min', thr)
                w = check(ctx, A)
                np.testing.assert_allclose(w, ref, atol=1e-12 * np.abs(ref).max())
                w2 = check(ctx, A)
                np.testing.assert_array_equal(w, w2)                      # fixed summation order: run-to-run identical
        for nb in ((24,) if ctx.backend == 'emu' else (4, 24)):
            ctx.set_option('eigh_nb', nb)
            ctx.set_option('eigh_symv_min', 1)
            ctx.set_option('eigh_symv_tr', 128)                           # 128-row tiles
            n = 100 if ctx.backend == 'emu' else 200
            A = rng.normal(size=(n, n))
            check(ctx, A + A.T)
        if ctx.backend != 'emu':
            A = rng.normal(size=(700, 700))
            check(ctx, A + A.T)
    finally:
        ctx.set_option('eigh_symv_min', 5120)
        ctx.set_option('eigh_symv_tr', 64)
        ctx.set_option('eigh_nb', 16)


def test_one_launch_per_column_chain(ctx):
    """Small trailing blocks run dsytd2's algebra with one launch per column (`trd_upd_kernel`: finish w, rank-2 update of
    the rows the workgroup owns, next row formed on the fly, matvec) — same tridiagonal matrix as the blocked chain up to
    rounding: with and without the LDS tail behind it, entered at the first column or in the middle of the factorisation,
    every rows-per-workgroup variant, one and several chunks per thread, odd and even offsets of the trailing block."""
    rng = np.random.RandomState(21)
    emu = ctx.backend == 'emu'
    try:
        for tail in (0, 128):
            ctx.set_option('eigh_tail_lds', tail)
            for n in ((3, 4, 5, 18, 70) if tail == 0 else ((141,) if emu else (141, 200, 301, 700))):
                A = rng.normal(size=(n, n))
                A = A + A.T
                ctx.set_option('eigh_upd_max', 0)
                ref = check(ctx, A)
                variants = [(4096, 2, 512), (4096, 4, 512), (4096, 8, 512), (40, 2, 512)]
                if n > 140:
                    variants += [(4096, 2, 128), (n // 2, 4, 128)]          # several chunks per thread from n = 257 on
                if emu and n > 100:
                    variants = [variants[0], variants[-1]]
                for upd_max, rows, nt in variants:
                    ctx.set_option('eigh_upd_max', upd_max)
                    ctx.set_option('eigh_upd_rows', rows)
                    ctx.set_option('eigh_upd_nt', nt)
                    w = check(ctx, A)
                    np.testing.assert_allclose(w, ref, atol=1e-12 * max(1.0, np.abs(ref).max()))
                    np.testing.assert_array_equal(w, check(ctx, A))        # fixed summation order: run-to-run identical
        # degenerate inputs through the chain: zero rows (tau = 0 reflectors), a diagonal matrix, identity + low rank
        ctx.set_option('eigh_tail_lds', 0)
        ctx.set_option('eigh_upd_max', 4096)
        ctx.set_option('eigh_upd_rows', 0)
        ctx.set_option('eigh_upd_nt', 512)
        for name, A in cases(40, rng):
            check(ctx, A)
    finally:
        ctx.set_option('eigh_tail_lds', 128)
        ctx.set_option('eigh_upd_max', 1024)
        ctx.set_option('eigh_upd_rows', 0)
        ctx.set_option('eigh_upd_nt', 512)


def test_back_transformation_with_the_strip_in_registers(ctx):
    """`wy_apply_strip_kernel` (the 16-row strip of X kept in the accumulator registers of eight wavefronts for all blocks;
    default from 2048 < n <= 3072, forced here by option value 2) against the streaming kernel and LAPACK."""
    try:
        ctx.set_option('eigh_wy_nb64_min', 1)
        for n in ((64, 192) if ctx.backend == "emu" else (64, 192, 320, 1024, 2112, 3072)):
            rng = np.random.RandomState(n)
            A = rng.normal(size=(n, n))
            A = A + A.T
            out = []
            for strip in (0, 2):
                ctx.set_option('eigh_wy_strip', strip)
                w = check(ctx, A)
                out.append((w, ctx.eigh(ctx.upload(A))[1].numpy()))
            np.testing.assert_array_equal(out[0][0], out[1][0])           # same tridiagonal problem, same eigenvalues
            assert np.abs(out[0][1] - out[1][1]).max() <= 1e-12 * n       # eigenvectors: another summation order only
    finally:
        ctx.set_option('eigh_wy_nb64_min', 2560)
        ctx.set_option('eigh_wy_strip', 1)


def test_spectra(ctx):
    rng = np.random.RandomState(1)
    n = 72 if ctx.backend == 'emu' else 700
    ctx.set_option('eigh_leaf', 8 if ctx.backend == 'emu' else 16)
    for name, A in cases(n, rng):
        check(ctx, A)
    ctx.set_option('eigh_leaf', 16)


def _rank1_check(ctx, D, w, rho, tol=2e-14):
    K = len(D)
    lam, Ut = ctx.rank1_eig(D, w, rho)
    M = np.diag(D) + rho * np.outer(w, w)
    ref = np.linalg.eigvalsh(M)
    scale = max(1.0, np.abs(ref).max())
    np.testing.assert_allclose(lam, ref, atol=tol * scale, rtol=0)
    # strict interlacing D_j < lam_j < D_{j+1}: what the Gu/Eisenstat weights rely on
    assert np.all(lam >= D) and np.all(lam[:-1] <= D[1:])
    assert np.abs(Ut @ Ut.T - np.eye(K)).max() <= 50 * tol
    assert np.abs(Ut @ M @ Ut.T - np.diag(lam)).max() <= 50 * tol * scale


def test_secular_regression_case(ctx):
    """Merge captured from the optimizer leg of bench.py on MI355X (n = 3072, B = lam0*I + low rank):
    a pole pair 1.4e-9 apart with weights of 1e-11, where the earlier Newton-on-tau*g iteration
    kept flipping on the rounding noise of g and ran into its iteration cap."""
    z = load_golden('secular_regression_r01')
    _rank1_check(ctx, z['D'], z['w'], float(z['rho']))


def test_secular_hard_spectra(ctx):
    rng = np.random.RandomState(7)
    ntrial = 12 if ctx.backend == 'emu' else 120
    for trial in range(ntrial):
        K = int(rng.randint(2, 60 if ctx.backend == 'emu' else 300))
        kind = trial % 4
        if kind == 0:
            D = np.sort(rng.normal(size=K))
        elif kind == 1:
            D = np.sort(np.exp(rng.uniform(-20, 3, size=K)))
        elif kind == 2:
            D = np.cumsum(np.exp(rng.uniform(-25, 0, size=K)))
        else:
            D = np.sort(np.round(rng.normal(size=K), 2) + 1e-10 * rng.normal(size=K))
        w = rng.normal(size=K) * np.exp(rng.uniform(-25, 0, size=K))
        w /= np.linalg.norm(w)
        keep = np.abs(w) > 1e-14
        keep[1:] &= np.diff(D) > 0
        D, w = D[keep], w[keep]
        if len(D) < 2 or np.diff(D).min() <= 0:
            continue
        _rank1_check(ctx, D, w, float(np.exp(rng.uniform(-5, 5))))


def test_trailing_update_kernels_agree(ctx):
    """The trailing rank-32 update of the blocked chain with every load issued up front (`rank2k_stream_fixed_kernel`,
    clamped addresses, masked afterwards) against the generic loop kernel: same eigenvalues, for block edges inside a tile,
    odd offsets and the triangle-only variant."""
    rng = np.random.RandomState(33)
    try:
        ctx.set_option('eigh_upd_max', 0)
        ctx.set_option('eigh_tail_lds', 0)
        for n, symv in ((70, 1 << 30), (201, 1 << 30), (150, 64)):
            A = rng.normal(size=(n, n))
            A = A + A.T
            ctx.set_option('eigh_symv_min', symv)
            out = []
            for fixed in (0, 1):
                ctx.set_option('rank2k_fixed', fixed)
                out.append(check(ctx, A))
            np.testing.assert_array_equal(out[0], out[1])        # same k order in every accumulator
    finally:
        ctx.set_option('eigh_upd_max', 1024)
        ctx.set_option('eigh_tail_lds', 128)
        ctx.set_option('eigh_symv_min', 5120)
        ctx.set_option('rank2k_fixed', 1)


def test_trailing_matvec_all_loads_up_front(ctx):
    """`trd_gemv_kernel<NCH>` (every load of the workgroup issued before the first wait, clamped and masked) against its
    loop form: the same sums in the same order, bit for bit — rows shorter than one chunk, ragged last chunks, odd and
    even offsets of the trailing block, panel rows appended."""
    rng = np.random.RandomState(35)
    try:
        ctx.set_option('eigh_upd_max', 0)
        ctx.set_option('eigh_tail_lds', 0)
        for n in (5, 70, 290) if ctx.backend == "emu" else (5, 70, 530, 1100, 2070, 2600):
            A = rng.normal(size=(n, n))
            A = A + A.T
            out = []
            for flat in (0, 1):
                ctx.set_option('eigh_gemv_flat', flat)
                w, V, _ = ctx.eigh(ctx.upload(A))
                out.append((np.array(w), V.numpy()))
            np.testing.assert_array_equal(out[0][0], out[1][0])
            np.testing.assert_array_equal(out[0][1], out[1][1])
    finally:
        ctx.set_option('eigh_upd_max', 1024)
        ctx.set_option('eigh_tail_lds', 128)
        ctx.set_option('eigh_gemv_flat', 1)


def test_divide_and_conquer_one_wait_per_level(ctx):
    """Divide & conquer with the next level's rank-one vectors queued behind the current level (one host wait per level
    instead of two) is a change of schedule only: eigenvalues and eigenvectors bit for bit those of the two-wait loop,
    for trees of depth 0, 1 and more, and for the spectra that deflate heavily."""
    rng = np.random.RandomState(37)
    try:
        for n in (3, 20, 33, 70, 150):
            mats = list(cases(n, rng)) if n == 70 else [('random', None)]
            for name, A in mats:
                if A is None:
                    A = rng.normal(size=(n, n))
                    A = A + A.T
                out = []
                for pipe in (0, 1):
                    ctx.set_option('eigh_dc_pipeline', pipe)
                    w, V, _ = ctx.eigh(ctx.upload(A))
                    out.append((np.array(w), V.numpy()))
                np.testing.assert_array_equal(out[0][0], out[1][0], err_msg=name)
                np.testing.assert_array_equal(out[0][1], out[1][1], err_msg=name)
    finally:
        ctx.set_option('eigh_dc_pipeline', 1)
