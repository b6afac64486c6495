// host_math.h — k x k host-side algebra shared by the Davidson driver and the quasi-Newton
// update (k = number of secant pairs / subspace size, tens at most).  The O(n k) and O(n^2)
// work they parametrise runs on the device; these are the O(k^3) coefficient computations
// that sit between two kernel launches.
#pragma once
#include <math.h>

#include <algorithm>
#include <vector>

#include "bordered.h"
#include "small_linalg.h"

namespace sella {
namespace hostm {

typedef std::vector<double> vec;

// x = pinv(M) b for a symmetric M (lower triangle read), minimum-norm like
// numpy.linalg.lstsq(M, b, rcond=None): singular values below eps*k*max are dropped.
inline void sym_pinv_solve(int m, const double* M, int ldm, const double* b, double* x) {
    if (m == 0) return;
    vec w(m), Z((size_t)m * m), work(m);
    small::sym_eig(m, M, ldm, w.data(), Z.data(), m, work.data());
    double wmax = 0.0;
    for (int i = 0; i < m; ++i) wmax = std::max(wmax, fabs(w[i]));
    const double cut = 2.220446049250313e-16 * m * wmax;
    for (int i = 0; i < m; ++i) x[i] = 0.0;
    for (int j = 0; j < m; ++j) {
        if (fabs(w[j]) <= cut) continue;
        double p = 0.0;
        for (int i = 0; i < m; ++i) p += Z[(size_t)i * m + j] * b[i];
        p /= w[j];
        for (int i = 0; i < m; ++i) x[i] += Z[(size_t)i * m + j] * p;
    }
}

// Coefficients of the secant symmetrisation (sella/hessian_update.py:12-37):
//   Ytilde = Y + Pn X   with Pn = S (symm 0 and 2) or Pn = Y (symm 1), X (k x k).
// Inputs: STS[a][b] = S_a.S_b,  STY[a][b] = S_a.Y_b   (k x k row-major).
// Returns which panel multiplies X: 0 -> S, 1 -> Y, -1 -> identity (X untouched, no correction).
inline int symm_coeffs(int k, const double* STS, const double* STY, int symm, double* X) {
    for (int i = 0; i < k * k; ++i) X[i] = 0.0;
    if (symm < 0 || k == 1) return -1;
    if (symm == 2) {
        // sequential: column i corrected inside span(S[:, :i])          (hessian_update.py:12-24)
        vec L((size_t)k * k), dYTS((size_t)k * k, 0.0), rhs(k), coef(k);
        const bool spd = small::cholesky(k, STS, k, L.data(), k) == 0;
        for (int i = 1; i < k; ++i) {
            // YTS[i, l] = Y_i.S_l = STY[l][i];  YTS[l, i] = Y_l.S_i = STY[i][l]
            for (int l = 0; l < i; ++l)
                rhs[l] = STY[(size_t)l * k + i] - STY[(size_t)i * k + l] - dYTS[(size_t)l * k + i];
            if (spd) {
                for (int l = 0; l < i; ++l) coef[l] = rhs[l];
                small::cholesky_solve_leading(i, L.data(), k, coef.data());
            } else {
                sym_pinv_solve(i, STS, k, rhs.data(), coef.data());
            }
            for (int l = 0; l < i; ++l) X[(size_t)l * k + i] = -coef[l];      // dY_i = -S[:, :i] coef
            for (int c = 0; c < k; ++c) {
                double s = 0.0;
                for (int l = 0; l < i; ++l) s += STS[(size_t)c * k + l] * coef[l];
                dYTS[(size_t)i * k + c] = -s;
            }
        }
        return 0;
    }
    // symm 0 / 1: X = lstsq(G, tril(S^T Y - Y^T S, -1)^T),  G = S^T S (0) or S^T Y (1)
    vec K((size_t)k * k, 0.0);            // K = tril(..., -1)^T  -> strictly upper
    for (int a = 0; a < k; ++a)
        for (int b = 0; b < a; ++b)
            K[(size_t)b * k + a] = STY[(size_t)a * k + b] - STY[(size_t)b * k + a];
    if (symm == 0) {
        vec col(k), sol(k);
        for (int c = 0; c < k; ++c) {
            for (int r = 0; r < k; ++r) col[r] = K[(size_t)r * k + c];
            sym_pinv_solve(k, STS, k, col.data(), sol.data());
            for (int r = 0; r < k; ++r) X[(size_t)r * k + c] = sol[r];
        }
        return 0;
    }
    // symm == 1: general (non-symmetric) k x k system, LU with partial pivoting
    vec G(STY, STY + (size_t)k * k);
    std::vector<int> piv(k);
    for (int i = 0; i < k * k; ++i) X[i] = K[i];
    if (small::lu_factor(k, G.data(), k, piv.data()) == 0) small::lu_solve(k, G.data(), k, piv.data(), X, k, k);
    return 1;
}

// Generalised symmetric-definite eigenproblem A x = lam M x (lower triangles of both are
// read, like scipy.linalg.eigh(A, M)): Cholesky M = L L^T, C = L^-1 A L^-T, eig(C), x = L^-T y.
// W (k x k) gets the eigenvectors as columns (W^T M W = I), lams ascending.  Returns 0 on success.
inline int gen_sym_eig(int k, const double* A, const double* M, double* lams, double* W) {
    vec L((size_t)k * k), C((size_t)k * k), Yv((size_t)k * k), work(k);
    if (small::cholesky(k, M, k, L.data(), k) != 0) return 1;
    // As = symmetric matrix from the lower triangle of A
    vec As((size_t)k * k);
    for (int i = 0; i < k; ++i)
        for (int j = 0; j <= i; ++j) As[(size_t)i * k + j] = As[(size_t)j * k + i] = A[(size_t)i * k + j];
    // T = L^-1 As  (forward substitution on each column)
    vec T((size_t)k * k);
    for (int c = 0; c < k; ++c)
        for (int i = 0; i < k; ++i) {
            double s = As[(size_t)i * k + c];
            for (int l = 0; l < i; ++l) s -= L[(size_t)i * k + l] * T[(size_t)l * k + c];
            T[(size_t)i * k + c] = s / L[(size_t)i * k + i];
        }
    // C = T L^-T  : C^T = L^-1 T^T  -> solve per row of T
    for (int r = 0; r < k; ++r)
        for (int i = 0; i < k; ++i) {
            double s = T[(size_t)r * k + i];
            for (int l = 0; l < i; ++l) s -= L[(size_t)i * k + l] * C[(size_t)r * k + l];
            C[(size_t)r * k + i] = s / L[(size_t)i * k + i];
        }
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < i; ++j) {
            const double v = 0.5 * (C[(size_t)i * k + j] + C[(size_t)j * k + i]);
            C[(size_t)i * k + j] = C[(size_t)j * k + i] = v;
        }
    if (small::sym_eig(k, C.data(), k, lams, Yv.data(), k, work.data()) != 0) return 2;
    // W = L^-T Yv (back substitution per column)
    for (int c = 0; c < k; ++c)
        for (int i = k - 1; i >= 0; --i) {
            double s = Yv[(size_t)i * k + c];
            for (int l = i + 1; l < k; ++l) s -= L[(size_t)l * k + i] * W[(size_t)l * k + c];
            W[(size_t)i * k + c] = s / L[(size_t)i * k + i];
        }
    return 0;
}


// Eigendecomposition of the (m + 1) x (m + 1) arrowhead matrix  [[diag(D), z], [z^T, alpha]]  in O(m^2):
// what the Rayleigh-Ritz step of the Davidson loop becomes once the basis has been rotated into the previous
// Ritz vectors (the old block is diag(theta), the new vector contributes the border).  D need not be sorted.
// lam (m + 1) ascending; W ((m + 1) x (m + 1) row-major) has the eigenvectors as COLUMNS, orthonormal to
// working precision: deflation as in LAPACK's dlaed2 (negligible border entries, coincident poles rotated
// apart), roots of the secular equation by `bordered_root` as (pole, offset) pairs, and the border recomputed
// from the roots (Gu & Eisenstat) so that close roots still give orthogonal vectors.  Returns 0 on success.
inline int arrow_eig(int m, const double* Din, const double* zin, double alpha, double* lam, double* W) {
    const int M = m + 1;
    if (m == 0) { lam[0] = alpha; W[0] = 1.0; return 0; }
    const double EPS = 2.220446049250313e-16;
    std::vector<int> ord(m);
    for (int i = 0; i < m; ++i) ord[i] = i;
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return Din[a] < Din[b]; });
    vec d(m), b(m);
    double scale = fabs(alpha);
    for (int i = 0; i < m; ++i) {
        d[i] = Din[ord[i]];
        b[i] = zin[ord[i]];
        scale = std::max(scale, std::max(fabs(d[i]), fabs(b[i])));
    }
    const double tol = 8.0 * EPS * scale;
    // deflation; rotations recorded as (p, q, c, s): coordinates x_p = c y_p + s y_q, x_q = -s y_p + c y_q
    struct Rot { int p, q; double c, s; };
    std::vector<Rot> rots;
    std::vector<char> active(m, 1);
    for (int i = 0; i < m; ++i)
        if (fabs(b[i]) <= tol) { active[i] = 0; b[i] = 0.0; }
    int prev = -1;
    for (int i = 0; i < m; ++i) {
        if (!active[i]) continue;
        if (prev >= 0 && d[i] - d[prev] <= tol) {
            const double r = hypot(b[prev], b[i]);
            const double c = b[i] / r, sn = b[prev] / r;
            const double dp = c * c * d[prev] + sn * sn * d[i], dq = sn * sn * d[prev] + c * c * d[i];
            d[prev] = dp;
            d[i] = dq;
            b[prev] = 0.0;
            b[i] = r;
            active[prev] = 0;
            rots.push_back({prev, i, c, sn});
        }
        prev = i;
    }
    std::vector<int> act;
    for (int i = 0; i < m; ++i)
        if (active[i]) act.push_back(i);
    const int ma = (int)act.size();
    // eigenvector matrix in the sorted / rotated coordinates: Y (M x M), columns = eigenvectors
    vec Y((size_t)M * M, 0.0), ev(M);
    int col = 0;
    for (int i = 0; i < m; ++i)
        if (!active[i]) {
            ev[col] = d[i];
            Y[(size_t)i * M + col] = 1.0;
            ++col;
        }
    if (ma == 0) {
        ev[col] = alpha;
        Y[(size_t)m * M + col] = 1.0;
        ++col;
    } else {
        vec Dp(ma), bp(ma), tau(ma + 1), bh(ma);
        std::vector<int> org(ma + 1);
        for (int a = 0; a < ma; ++a) { Dp[a] = d[act[a]] - alpha; bp[a] = b[act[a]]; }
        for (int j = 0; j <= ma; ++j) bordered::bordered_root(ma, Dp.data(), bp.data(), j, &org[j], &tau[j], true);
        // mu_j - D_a with full relative accuracy
        auto diff = [&](int j, int a) { return (org[j] >= 0 ? (Dp[org[j]] - Dp[a]) : -Dp[a]) + tau[j]; };
        for (int a = 0; a < ma; ++a) {
            // bhat_a^2 = -(mu_a - D_a)(mu_{a+1} - D_a) prod_{l<a} (mu_l - D_a)/(D_l - D_a) prod_{l>a} (mu_{l+1} - D_a)/(D_l - D_a)
            double p = -diff(a, a) * diff(a + 1, a);
            for (int l = 0; l < a; ++l) p *= diff(l, a) / (Dp[l] - Dp[a]);
            for (int l = a + 1; l < ma; ++l) p *= diff(l + 1, a) / (Dp[l] - Dp[a]);
            const double v = sqrt(fabs(p));
            bh[a] = (bp[a] >= 0.0) ? v : -v;
        }
        for (int j = 0; j <= ma; ++j) {
            double nrm2 = 1.0;
            for (int a = 0; a < ma; ++a) {
                const double x = bh[a] / diff(j, a);
                Y[(size_t)act[a] * M + col] = x;
                nrm2 += x * x;
            }
            const double inv = 1.0 / sqrt(nrm2);
            for (int a = 0; a < ma; ++a) Y[(size_t)act[a] * M + col] *= inv;
            Y[(size_t)m * M + col] = inv;
            ev[col] = alpha + (org[j] >= 0 ? Dp[org[j]] : 0.0) + tau[j];
            ++col;
        }
    }
    // undo the deflating rotations (reverse order) on the coordinate rows
    for (int r = (int)rots.size() - 1; r >= 0; --r) {
        const Rot& g = rots[r];
        double* yp = Y.data() + (size_t)g.p * M;
        double* yq = Y.data() + (size_t)g.q * M;
        for (int cidx = 0; cidx < M; ++cidx) {
            const double a = yp[cidx], bq = yq[cidx];
            yp[cidx] = g.c * a + g.s * bq;
            yq[cidx] = -g.s * a + g.c * bq;
        }
    }
    // ascending order of the eigenvalues, rows back to the caller's order of D
    std::vector<int> eo(M);
    for (int j = 0; j < M; ++j) eo[j] = j;
    std::stable_sort(eo.begin(), eo.end(), [&](int a, int b2) { return ev[a] < ev[b2]; });
    for (int j = 0; j < M; ++j) {
        lam[j] = ev[eo[j]];
        for (int i = 0; i < m; ++i) W[(size_t)ord[i] * M + j] = Y[(size_t)i * M + eo[j]];
        W[(size_t)m * M + j] = Y[(size_t)m * M + eo[j]];
    }
    return 0;
}

// C = A^T B A style helpers on k x k row-major matrices
inline void congruence(int k, const double* W, const double* G, double* out) {
    // out = W^T G W
    vec T((size_t)k * k);
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) {
            double s = 0.0;
            for (int l = 0; l < k; ++l) s += G[(size_t)i * k + l] * W[(size_t)l * k + j];
            T[(size_t)i * k + j] = s;
        }
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) {
            double s = 0.0;
            for (int l = 0; l < k; ++l) s += W[(size_t)l * k + i] * T[(size_t)l * k + j];
            out[(size_t)i * k + j] = s;
        }
}

// Full symmetric eigendecomposition of a small dense matrix with every inner loop running along a contiguous row (the
// k x k Rayleigh-Ritz problem of the block Davidson iteration, k = 48: the column-oriented EISPACK routines of
// small_linalg.h take 135 us there — strided rotations — and sit alone on the iteration's critical path).
//   Householder tridiagonalisation on the FULL symmetric storage (both triangles updated: twice the flops of a
//   one-triangle update, all of them in vector loops), Q accumulated backwards, implicit-shift QL with the rotations
//   applied to the ROWS of Z^T.
// G: k x k, symmetric (the lower triangle is mirrored first), leading dimension ldg.  On return w ascending, row j of Wt
// (leading dimension k) = eigenvector j, largest-magnitude component positive (the convention of small::sym_eig).
// Returns 0, or l + 1 if eigenvalue l did not converge in 60 QL iterations.
inline int sym_eig_rows(int k, const double* G, int ldg, double* w, double* Wt) {
#if defined(__clang__)
#pragma clang fp contract(fast)      // (no trajectory depends on these last bits: block Davidson's parity target is the converged pair)
#endif
    if (k <= 0) return 0;
    if (k == 1) { w[0] = G[0]; Wt[0] = 1.0; return 0; }
    static thread_local vec M, Vs, Qm, p, e, taus;
    M.resize((size_t)k * k); Vs.assign((size_t)k * k, 0.0); Qm.resize((size_t)k * k);
    p.resize(k); e.assign(k, 0.0); taus.assign(k, 0.0);
    for (int i = 0; i < k; ++i)
        for (int j = 0; j <= i; ++j) M[(size_t)i * k + j] = M[(size_t)j * k + i] = G[(size_t)i * ldg + j];
    // ---- tridiagonalisation: column j below the sub-diagonal eliminated by H_j = I - tau v v^T, v = (1, ...) on rows j+1.. ----
    for (int j = 0; j + 2 < k; ++j) {
        const int m = k - j - 1;                              // trailing block rows j+1 .. k-1
        double* x = &M[(size_t)j * k + j + 1];                // row j right of the diagonal = column j below it
        double* v = &Vs[(size_t)j * k + j + 1];
        double sig = 0.0;
        for (int i = 1; i < m; ++i) sig += x[i] * x[i];
        const double alpha = x[0];
        if (sig == 0.0) { e[j] = alpha; taus[j] = 0.0; continue; }
        const double nrm = sqrt(alpha * alpha + sig);
        const double beta = (alpha >= 0.0) ? -nrm : nrm;
        const double tau = (beta - alpha) / beta, sc = 1.0 / (alpha - beta);
        v[0] = 1.0;
        for (int i = 1; i < m; ++i) v[i] = x[i] * sc;
        e[j] = beta;
        taus[j] = tau;
        // p = tau A22 v;  q = p - (tau/2)(p.v) v;  A22 -= v q^T + q v^T
        double pv = 0.0;
        for (int r = 0; r < m; ++r) {
            const double* a = &M[(size_t)(j + 1 + r) * k + j + 1];
            double s = 0.0;
            for (int cidx = 0; cidx < m; ++cidx) s += a[cidx] * v[cidx];
            p[r] = tau * s;
            pv += p[r] * v[r];
        }
        const double hh = 0.5 * tau * pv;
        for (int r = 0; r < m; ++r) p[r] -= hh * v[r];
        for (int r = 0; r < m; ++r) {
            double* a = &M[(size_t)(j + 1 + r) * k + j + 1];
            const double vr = v[r], pr = p[r];
            for (int cidx = 0; cidx < m; ++cidx) a[cidx] -= vr * p[cidx] + pr * v[cidx];
        }
    }
    if (k >= 2) e[k - 2] = M[(size_t)(k - 2) * k + k - 1];
    for (int i = 0; i < k; ++i) w[i] = M[(size_t)i * k + i];
    // ---- Q = H_0 H_1 ... H_{k-3} accumulated backwards (only the block H_j touches is not yet the identity) ----
    std::fill(Qm.begin(), Qm.end(), 0.0);
    for (int i = 0; i < k; ++i) Qm[(size_t)i * k + i] = 1.0;
    for (int j = k - 3; j >= 0; --j) {
        if (taus[j] == 0.0) continue;
        const int m = k - j - 1;
        const double* v = &Vs[(size_t)j * k + j + 1];
        double* y = p.data();                                 // y = v^T B, B = Q[j+1:, j+1:]
        for (int cidx = 0; cidx < m; ++cidx) y[cidx] = 0.0;
        for (int r = 0; r < m; ++r) {
            const double* b = &Qm[(size_t)(j + 1 + r) * k + j + 1];
            const double vr = v[r];
            for (int cidx = 0; cidx < m; ++cidx) y[cidx] += vr * b[cidx];
        }
        for (int r = 0; r < m; ++r) {
            double* b = &Qm[(size_t)(j + 1 + r) * k + j + 1];
            const double f = taus[j] * v[r];
            for (int cidx = 0; cidx < m; ++cidx) b[cidx] -= f * y[cidx];
        }
    }
    for (int i = 0; i < k; ++i)                               // Wt = Q^T
        for (int j = 0; j < k; ++j) Wt[(size_t)i * k + j] = Qm[(size_t)j * k + i];
    // ---- implicit-shift QL (the recurrence of small::tridiag_ql), rotations on rows i, i+1 of Wt ----
    double* d = w;
    e[k - 1] = 0.0;
    for (int l = 0; l < k; ++l) {
        int iter = 0, m;
        do {
            for (m = l; m < k - 1; ++m) {
                const double dd = fabs(d[m]) + fabs(d[m + 1]);
                if (fabs(e[m]) <= 2.220446049250313e-16 * dd) break;
            }
            if (m != l) {
                if (iter++ == 60) return l + 1;
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = sqrt(g * g + 1.0);
                g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? fabs(r) : -fabs(r)));
                double s = 1.0, c = 1.0, pp = 0.0;
                int i;
                for (i = m - 1; i >= l; --i) {
                    double f = s * e[i];
                    const double b = c * e[i];
                    r = sqrt(f * f + g * g);
                    e[i + 1] = r;
                    if (r == 0.0) { d[i + 1] -= pp; e[m] = 0.0; break; }
                    s = f / r;
                    c = g / r;
                    g = d[i + 1] - pp;
                    r = (d[i] - g) * s + 2.0 * c * b;
                    pp = s * r;
                    d[i + 1] = g + pp;
                    g = c * r - b;
                    double* z0 = &Wt[(size_t)i * k];
                    double* z1 = z0 + k;
                    for (int q = 0; q < k; ++q) {
                        const double f2 = z1[q], f1 = z0[q];
                        z1[q] = s * f1 + c * f2;
                        z0[q] = c * f1 - s * f2;
                    }
                }
                if (r == 0.0 && i >= l) continue;
                d[l] -= pp;
                e[l] = g;
                e[m] = 0.0;
            }
        } while (m != l);
    }
    // ---- ascending order, sign convention ----
    static thread_local std::vector<int> ord;
    ord.resize(k);
    for (int i = 0; i < k; ++i) ord[i] = i;
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return d[a] < d[b]; });
    for (int i = 0; i < k; ++i) p[i] = d[ord[i]];
    Qm.assign(Wt, Wt + (size_t)k * k);
    for (int i = 0; i < k; ++i) {
        const double* src = &Qm[(size_t)ord[i] * k];
        double vm = 0.0;
        int im = 0;
        for (int q = 0; q < k; ++q)
            if (fabs(src[q]) > vm) { vm = fabs(src[q]); im = q; }
        const double sg = (src[im] < 0.0) ? -1.0 : 1.0;
        for (int q = 0; q < k; ++q) Wt[(size_t)i * k + q] = sg * src[q];
        w[i] = p[i];
    }
    return 0;
}

}  // namespace hostm
}  // namespace sella
