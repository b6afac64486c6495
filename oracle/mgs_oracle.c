/* ORACLE (test infrastructure, never linked into the product).
 *
 * C restatement of the reference's only native routine on the hot path, the iterated modified
 * Gram-Schmidt `mgs` of sella/utilities/math.pyx:74-140 (a Cython nogil function over BLAS-1
 * ddot/daxpy/dnrm2).  Column-strided access exactly like the reference: X is (n x nx) row-major,
 * column j starts at X + j with stride nx.
 *
 *   returns  m >= 0 : number of columns kept (leading columns of X), the rest zeroed
 *            -1     : shape mismatch (math.pyx:89-90)
 *            -2     : a column did not converge within maxiter sweeps (math.pyx:132-133)
 */
#include <math.h>
#include <stddef.h>

static double col_dot(int n, const double* a, int sa, const double* b, int sb) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += a[(size_t)i * sa] * b[(size_t)i * sb];
    return s;
}

static double col_nrm2(int n, const double* a, int sa) { return sqrt(col_dot(n, a, sa, a, sa)); }

int mgs_oracle(int n, double* X, int nx, double* Y, int ny, int y_rows, double eps1, double eps2,
               int maxiter) {
    if (Y != NULL && y_rows != n) return -1;
    if (Y == NULL) ny = 0;
    int m = 0;
    for (int i = 0; i < nx; ++i) {
        double* xm = X + m;
        if (i != m)
            for (int k = 0; k < n; ++k) xm[(size_t)k * nx] = X[(size_t)k * nx + i];      /* :100-101 */
        double norm = col_nrm2(n, xm, nx);
        for (int k = 0; k < n; ++k) xm[(size_t)k * nx] /= norm;                          /* :102-104 */
        int niter, accepted = 0, dropped = 0;
        for (niter = 0; niter < maxiter; ++niter) {                                      /* :105 */
            double normtot = 1.0;
            for (int pass = 0; pass < 2 && !dropped; ++pass) {
                const double* base = pass ? X : Y;
                const int nb = pass ? m : ny, sb = pass ? nx : ny;
                for (int j = 0; j < nb; ++j) {                                           /* :107-126 */
                    const double* bj = base + j;
                    const double dot = -col_dot(n, bj, sb, xm, nx);
                    for (int k = 0; k < n; ++k) xm[(size_t)k * nx] += dot * bj[(size_t)k * sb];
                    norm = col_nrm2(n, xm, nx);
                    normtot *= norm;
                    if (normtot < eps2) { dropped = 1; break; }
                    for (int k = 0; k < n; ++k) xm[(size_t)k * nx] /= norm;
                }
            }
            if (dropped) break;                                                          /* :116-117,127-128 */
            if (0.0 <= 1.0 - normtot && 1.0 - normtot <= eps1) { accepted = 1; break; }  /* :129-131 */
        }
        if (accepted) ++m;
        else if (!dropped) return -2;                                                    /* :132-133 */
    }
    for (int i = m; i < nx; ++i)
        for (int k = 0; k < n; ++k) X[(size_t)k * nx + i] = 0.0;                         /* :136-138 */
    return m;
}
