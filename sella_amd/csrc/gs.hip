// gs.hip — Gram-Schmidt orthonormalisation on vector-major device panels.
//
// Restates the accept/drop semantics of the reference's native routine `mgs`
// (sella/utilities/math.pyx:74-140) with an MI355X-friendly sweep: instead of k dependent
// (dot, axpy, norm) triples per sweep — k grid-wide reductions — one sweep is two panel
// kernels, c = V t (row-panel matvec, wave64 reductions) and t -= V^T c (coalesced linear
// combination), i.e. iterated classical Gram-Schmidt.  The sweep repeats until the norm stays
// within eps1 of one (math.pyx:129) and the vector is dropped when a sweep shrinks it below
// eps2 (math.pyx:112-117); results agree with the sequential variant to O(eps).
#include "internal.h"

namespace sella {

// One sweep of t against basis rows [0, k): cvec = -(V t); t += V^T cvec; |t|^2 -> dscal[slot];
// t /= |t|.  No host synchronisation.
static int gs_sweep(sella_ctx* c, const double* basis, int ldb, int k, double* t, int n, int slot) {
    double* cvec = c->dscal + DS_CVEC;
    if (k > 0) {
        GemvEpi neg;
        neg.alpha = -1.0;
        SCHK(launch_gemv_rows(c, basis, k, n, ldb, t, ldb, 1, cvec, k, neg));
        SCHK(launch_lincomb(c, n, 1, basis, ldb, k, cvec, 1, nullptr, 0, 0, nullptr, 0, 1.0, t, ldb));
    }
    return launch_normalize(c, t, n, scal_out(c, slot));
}

// Two sweeps of t against the k orthonormal rows of `basis`, normalised; NO host synchronisation: |t|^2 of the input,
// after the first and after the second sweep are left in scalar slots 8, 9, 10 for the caller's next round trip
// (gs_orthonormalise's accept / drop decision is taken from exactly these three numbers).
int gs_project_twice(sella_ctx* c, const double* basis, int ldb, int k, double* t, int n) {
    if (k > DS_STAGE - DS_CVEC) {
        set_error("gram-schmidt: basis of %d vectors exceeds the coefficient buffer", k);
        return SELLA_E_UNSUPPORTED;
    }
    SCHK(launch_normalize(c, t, n, scal_out(c, 8)));
    SCHK(gs_sweep(c, basis, ldb, k, t, n, 9));
    return gs_sweep(c, basis, ldb, k, t, n, 10);
}

// Orthonormalise the n-vector t against the k orthonormal rows of `basis`.
// *kept = 1 if t was accepted (unit norm, orthogonal to the basis), 0 if it was dropped.
// *first_norm (optional) = |t - V V^T t| of the normalised input, the quantity the Davidson
// driver tests against 1e-2 (sella/eigensolvers.py:93).
int gs_orthonormalise(sella_ctx* c, const double* basis, int ldb, int k, double* t, int n,
                      double eps1, double eps2, int maxiter, int* kept, double* first_norm) {
    *kept = 0;
    if (k > DS_STAGE - DS_CVEC) {
        set_error("gram-schmidt: basis of %d vectors exceeds the coefficient buffer", k);
        return SELLA_E_UNSUPPORTED;
    }
    SCHK(launch_normalize(c, t, n, scal_out(c, 8)));
    SCHK(gs_sweep(c, basis, ldb, k, t, n, 9));
    SCHK(gs_sweep(c, basis, ldb, k, t, n, 10));
    SCHK(sync_scalars(c, 8, 3));
    const double n0sq = c->hscal[8];
    double n1 = sqrt(c->hscal[9]), n2 = sqrt(c->hscal[10]);
    if (first_norm) *first_norm = n1;
    if (!(n0sq > 0.0) || n1 != n1) return SELLA_OK;        // zero or NaN input: dropped
    if (n1 < eps2) return SELLA_OK;
    for (int it = 0; it < maxiter; ++it) {
        if (n2 != n2 || n2 < eps2) return SELLA_OK;
        if (fabs(1.0 - n2) <= eps1) { *kept = 1; return SELLA_OK; }
        SCHK(gs_sweep(c, basis, ldb, k, t, n, 10));
        SCHK(sync_scalars(c, 10, 1));
        n2 = sqrt(c->hscal[10]);
    }
    set_error("MGS failed.");
    return SELLA_E_NOCONV;
}

}  // namespace sella

using namespace sella;

// modified_gram_schmidt(X, Y): Y is orthonormalised by itself first (math.pyx:148-151), then
// every column of X against Y and the accepted columns of X.
extern "C" int sella_mgs(sella_ctx* c, const double* X, int n, int nx, const double* Y, int ny,
                         double eps1, double eps2, int maxiter, double* out, int* kept_out) {
    if (!c || !X || !out || !kept_out || n <= 0 || nx < 0 || ny < 0 || (ny > 0 && !Y)) {
        set_error("mgs: invalid arguments");
        return SELLA_E_INVALID;
    }
    *kept_out = 0;
    if (nx == 0) return SELLA_OK;
    const int ld = round_up(n, 8);
    double *P, *src;
    SCHK(scratch_get(c, SCR_V, (size_t)(nx + ny + 1) * ld * sizeof(double), &P));
    SCHK(scratch_get(c, SCR_AV, (size_t)(nx + ny + 1) * ld * sizeof(double), &src));
    if (ny) SCHK(upload_panel(c, Y, n, ny, src, ld));
    SCHK(upload_panel(c, X, n, nx, src + (size_t)ny * ld, ld));
    int k = 0, ykept = 0;
    for (int col = 0; col < ny + nx; ++col) {
        double* slot = P + (size_t)k * ld;
        HIPCHK(s_memcpy(c, slot, src + (size_t)col * ld, (size_t)ld * sizeof(double), hipMemcpyDeviceToDevice));
        int kept = 0;
        SCHK(gs_orthonormalise(c, P, ld, k, slot, n, eps1, eps2, maxiter, &kept, nullptr));
        if (kept) ++k;
        if (col == ny - 1) ykept = k;
    }
    if (ny == 0) ykept = 0;
    const int xkept = k - ykept;
    // out is (n x nx) row-major; kept columns first, the rest zero (math.pyx:136-138)
    for (size_t i = 0; i < (size_t)n * nx; ++i) out[i] = 0.0;
    if (xkept > 0) {
        std::vector<double> tmp((size_t)n * xkept);
        SCHK(download_panel(c, P + (size_t)ykept * ld, ld, n, xkept, tmp.data()));
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < xkept; ++j) out[(size_t)i * nx + j] = tmp[(size_t)i * xkept + j];
    }
    *kept_out = xkept;
    return SELLA_OK;
}
