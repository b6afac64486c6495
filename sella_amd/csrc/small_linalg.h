// small_linalg.h — dense O(k^3) routines for the *small* (k x k, k = subspace size) problems
// of the path.  Plain C++ (no HIP types) so the same code runs inside device kernels
// (leaf eigenproblems of the divide-and-conquer solver) and in the host driver (Rayleigh-Ritz
// matrices of the Davidson loop, secant-symmetrisation coefficients, update coefficients).
//
// Everything is row-major, fp64, caller-provided storage.  Algorithms are the textbook ones
// (Householder tridiagonalisation + implicit-shift QL, Cholesky, LU with partial pivoting),
// written from the mathematical definitions.
#pragma once
#include <math.h>

#ifndef SELLA_HD
#define SELLA_HD __host__ __device__
#endif

namespace sella {
namespace small {

SELLA_HD inline double sign_of(double a, double b) { return b >= 0.0 ? fabs(a) : -fabs(a); }

// ---------------------------------------------------------------------------------------
// Implicit-shift QL on a symmetric tridiagonal matrix (diagonal d[0..n), sub-diagonal
// e[0..n-1) stored in e[0..n-2], e[n-1] is workspace).  If Z != nullptr the rotations are
// accumulated into the columns of Z (n_rows x n, leading dimension ldz), i.e. on exit
// column j of Z_in * (eigenvector matrix of T).  Rows [row0, n_rows) stepping by row_step
// are updated by the caller's thread (lets a wavefront share the scalar recurrence while
// each lane owns a subset of the rows).  Eigenvalues are NOT sorted here.
// Returns 0, or l+1 if eigenvalue l failed to converge in 60 iterations.
// ---------------------------------------------------------------------------------------
SELLA_HD inline int tridiag_ql(int n, double* d, double* e, double* Z, int ldz, int n_rows,
                               int row0 = 0, int row_step = 1, long cs = 1) {
    if (n <= 1) return 0;
    e[n - 1] = 0.0;
    for (int l = 0; l < n; ++l) {
        int iter = 0;
        int m;
        do {
            for (m = l; m < n - 1; ++m) {
                double dd = fabs(d[m]) + fabs(d[m + 1]);
                if (fabs(e[m]) <= 2.220446049250313e-16 * dd) break;
            }
            if (m != l) {
                if (iter++ == 60) return l + 1;
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = hypot(g, 1.0);
                g = d[m] - d[l] + e[l] / (g + sign_of(r, g));
                double s = 1.0, c = 1.0, p = 0.0;
                int i;
                for (i = m - 1; i >= l; --i) {
                    double f = s * e[i];
                    double b = c * e[i];
                    r = hypot(f, g);
                    e[i + 1] = r;
                    if (r == 0.0) {
                        d[i + 1] -= p;
                        e[m] = 0.0;
                        break;
                    }
                    s = f / r;
                    c = g / r;
                    g = d[i + 1] - p;
                    r = (d[i] - g) * s + 2.0 * c * b;
                    p = s * r;
                    d[i + 1] = g + p;
                    g = c * r - b;
                    if (Z) {
                        for (int k = row0; k < n_rows; k += row_step) {
                            double* zr = Z + (long)k * ldz;
                            double f2 = zr[(i + 1) * cs];
                            zr[(i + 1) * cs] = s * zr[i * cs] + c * f2;
                            zr[i * cs] = c * zr[i * cs] - s * f2;
                        }
                    }
                }
                if (r == 0.0 && i >= l) continue;
                d[l] -= p;
                e[l] = g;
                e[m] = 0.0;
            }
        } while (m != l);
    }
    return 0;
}

// ---------------------------------------------------------------------------------------
// Householder reduction of a dense symmetric matrix A (n x n, only the lower triangle is
// read) to tridiagonal form.  On exit Z (n x n, ld = ldz) holds the orthogonal matrix Q with
// Q^T A Q = T, d the diagonal and e[0..n-2] the sub-diagonal of T.  A is not modified.
// ---------------------------------------------------------------------------------------
SELLA_HD inline void householder_tridiag(int n, const double* A, int lda, double* Z, int ldz,
                                         double* d, double* e) {
    // work on Z as a copy of the (symmetrised-from-lower) matrix
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double v = A[(long)i * lda + j];
            Z[(long)i * ldz + j] = v;
            Z[(long)j * ldz + i] = v;
        }
    // EISPACK-style tred2 organisation: eliminate row i using rows 0..i-1
    for (int i = n - 1; i >= 1; --i) {
        int l = i - 1;
        double h = 0.0, scale = 0.0;
        if (l > 0) {
            for (int k = 0; k <= l; ++k) scale += fabs(Z[(long)i * ldz + k]);
            if (scale == 0.0) {
                e[i] = Z[(long)i * ldz + l];
            } else {
                for (int k = 0; k <= l; ++k) {
                    Z[(long)i * ldz + k] /= scale;
                    h += Z[(long)i * ldz + k] * Z[(long)i * ldz + k];
                }
                double f = Z[(long)i * ldz + l];
                double g = (f >= 0.0 ? -sqrt(h) : sqrt(h));
                e[i] = scale * g;
                h -= f * g;
                Z[(long)i * ldz + l] = f - g;
                f = 0.0;
                for (int j = 0; j <= l; ++j) {
                    Z[(long)j * ldz + i] = Z[(long)i * ldz + j] / h;
                    g = 0.0;
                    for (int k = 0; k <= j; ++k) g += Z[(long)j * ldz + k] * Z[(long)i * ldz + k];
                    for (int k = j + 1; k <= l; ++k) g += Z[(long)k * ldz + j] * Z[(long)i * ldz + k];
                    e[j] = g / h;
                    f += e[j] * Z[(long)i * ldz + j];
                }
                double hh = f / (h + h);
                for (int j = 0; j <= l; ++j) {
                    f = Z[(long)i * ldz + j];
                    e[j] = g = e[j] - hh * f;
                    for (int k = 0; k <= j; ++k)
                        Z[(long)j * ldz + k] -= (f * e[k] + g * Z[(long)i * ldz + k]);
                }
            }
        } else {
            e[i] = Z[(long)i * ldz + l];
        }
        d[i] = h;
    }
    d[0] = 0.0;
    e[0] = 0.0;
    for (int i = 0; i < n; ++i) {
        int l = i - 1;
        if (d[i] != 0.0) {
            for (int j = 0; j <= l; ++j) {
                double g = 0.0;
                for (int k = 0; k <= l; ++k) g += Z[(long)i * ldz + k] * Z[(long)k * ldz + j];
                for (int k = 0; k <= l; ++k) Z[(long)k * ldz + j] -= g * Z[(long)k * ldz + i];
            }
        }
        d[i] = Z[(long)i * ldz + i];
        Z[(long)i * ldz + i] = 1.0;
        for (int j = 0; j <= l; ++j) Z[(long)j * ldz + i] = Z[(long)i * ldz + j] = 0.0;
    }
    // shift the sub-diagonal to e[0..n-2]
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
}

// Sort eigenvalues ascending and permute the columns of Z accordingly (selection sort; k small).
SELLA_HD inline void sort_eig(int n, double* w, double* Z, int ldz, int n_rows) {
    for (int i = 0; i < n - 1; ++i) {
        int k = i;
        double p = w[i];
        for (int j = i + 1; j < n; ++j)
            if (w[j] < p) { k = j; p = w[j]; }
        if (k != i) {
            w[k] = w[i];
            w[i] = p;
            if (Z)
                for (int r = 0; r < n_rows; ++r) {
                    double t = Z[(long)r * ldz + i];
                    Z[(long)r * ldz + i] = Z[(long)r * ldz + k];
                    Z[(long)r * ldz + k] = t;
                }
        }
    }
}

// Full symmetric eigendecomposition of a small matrix (lower triangle of A is read).
// w ascending, eigenvectors in the columns of Z; each column's sign is fixed so that its
// largest-magnitude component is positive (LAPACK leaves the sign arbitrary).
// work: n doubles.  Returns 0 on success.
SELLA_HD inline int sym_eig(int n, const double* A, int lda, double* w, double* Z, int ldz,
                            double* work) {
    if (n == 0) return 0;
    if (n == 1) { w[0] = A[0]; Z[0] = 1.0; return 0; }
    householder_tridiag(n, A, lda, Z, ldz, w, work);
    int info = tridiag_ql(n, w, work, Z, ldz, n);
    if (info) return info;
    sort_eig(n, w, Z, ldz, n);
    for (int j = 0; j < n; ++j) {
        int im = 0;
        double vm = 0.0;
        for (int i = 0; i < n; ++i)
            if (fabs(Z[(long)i * ldz + j]) > vm) { vm = fabs(Z[(long)i * ldz + j]); im = i; }
        if (Z[(long)im * ldz + j] < 0.0)
            for (int i = 0; i < n; ++i) Z[(long)i * ldz + j] = -Z[(long)i * ldz + j];
    }
    return 0;
}

// Cholesky factorisation M = L L^T (lower triangle of M read, L written into the lower
// triangle of Lout, strict upper zeroed).  Returns 0, or j+1 if pivot j is not positive.
SELLA_HD inline int cholesky(int n, const double* M, int ldm, double* L, int ldl) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) L[(long)i * ldl + j] = 0.0;
    for (int j = 0; j < n; ++j) {
        double s = M[(long)j * ldm + j];
        for (int k = 0; k < j; ++k) s -= L[(long)j * ldl + k] * L[(long)j * ldl + k];
        if (!(s > 0.0)) return j + 1;
        double ljj = sqrt(s);
        L[(long)j * ldl + j] = ljj;
        for (int i = j + 1; i < n; ++i) {
            double t = M[(long)i * ldm + j];
            for (int k = 0; k < j; ++k) t -= L[(long)i * ldl + k] * L[(long)j * ldl + k];
            L[(long)i * ldl + j] = t / ljj;
        }
    }
    return 0;
}

// Solve (L L^T)[:m,:m] x = b using the leading m x m block of a Cholesky factor (in place).
SELLA_HD inline void cholesky_solve_leading(int m, const double* L, int ldl, double* b) {
    for (int i = 0; i < m; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[(long)i * ldl + k] * b[k];
        b[i] = s / L[(long)i * ldl + i];
    }
    for (int i = m - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < m; ++k) s -= L[(long)k * ldl + i] * b[k];
        b[i] = s / L[(long)i * ldl + i];
    }
}

// LU factorisation with partial pivoting, in place; piv[n].  Returns 0 or j+1 if singular.
SELLA_HD inline int lu_factor(int n, double* A, int lda, int* piv) {
    for (int j = 0; j < n; ++j) {
        int p = j;
        double vmax = fabs(A[(long)j * lda + j]);
        for (int i = j + 1; i < n; ++i)
            if (fabs(A[(long)i * lda + j]) > vmax) { vmax = fabs(A[(long)i * lda + j]); p = i; }
        piv[j] = p;
        if (vmax == 0.0) return j + 1;
        if (p != j)
            for (int c = 0; c < n; ++c) {
                double t = A[(long)j * lda + c];
                A[(long)j * lda + c] = A[(long)p * lda + c];
                A[(long)p * lda + c] = t;
            }
        double inv = 1.0 / A[(long)j * lda + j];
        for (int i = j + 1; i < n; ++i) {
            double f = A[(long)i * lda + j] * inv;
            A[(long)i * lda + j] = f;
            if (f != 0.0)
                for (int c = j + 1; c < n; ++c) A[(long)i * lda + c] -= f * A[(long)j * lda + c];
        }
    }
    return 0;
}

// Solve A X = Bm for nrhs right-hand sides stored as columns of Bm (n x nrhs row-major),
// given lu_factor output.  In place.
SELLA_HD inline void lu_solve(int n, const double* LU, int lda, const int* piv, double* Bm,
                              int ldb, int nrhs) {
    for (int j = 0; j < n; ++j) {
        int p = piv[j];
        if (p != j)
            for (int c = 0; c < nrhs; ++c) {
                double t = Bm[(long)j * ldb + c];
                Bm[(long)j * ldb + c] = Bm[(long)p * ldb + c];
                Bm[(long)p * ldb + c] = t;
            }
    }
    for (int c = 0; c < nrhs; ++c) {
        for (int i = 0; i < n; ++i) {
            double s = Bm[(long)i * ldb + c];
            for (int k = 0; k < i; ++k) s -= LU[(long)i * lda + k] * Bm[(long)k * ldb + c];
            Bm[(long)i * ldb + c] = s;
        }
        for (int i = n - 1; i >= 0; --i) {
            double s = Bm[(long)i * ldb + c];
            for (int k = i + 1; k < n; ++k) s -= LU[(long)i * lda + k] * Bm[(long)k * ldb + c];
            Bm[(long)i * ldb + c] = s / LU[(long)i * lda + i];
        }
    }
}

}  // namespace small
}  // namespace sella
