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

// ---- the whole routine in ONE launch for small problems (round 6) ------------------------------------------------------
// For the searches of an ensemble member (3N = 768, 384 free coordinates, at most a few dozen basis vectors) the routine
// above is 7 launches and a host round trip for the two obligatory sweeps, and 3 launches + a round trip for every further
// one (the reference's acceptance test asks for |1 - |t|| <= 1e-15, which takes a third sweep more often than not): half of
// all launches of a member's Davidson diagonalisation.  One workgroup holds t in LDS and does the sweeps, the norms AND
// the accept / drop decisions of gs_orthonormalise itself: wavefront w takes the dots of the basis rows a = w mod 4 (64
// lanes stride the row, wave reduction), every thread then updates its own elements with all k coefficients (classical
// Gram-Schmidt per sweep, as above).  Same arithmetic per sweep in another summation order; the decisions are the same
// code on the same kind of numbers.  out[0] = |t_in|^2, out[1] = |t|^2 after the first sweep, out[2] = after the last one,
// out[3] = kept (1 / 0), out[4] = 1 if `maxiter` sweeps did not settle the norm ("MGS failed."), out[5] = sweeps made.
constexpr int GSS_MAXN = 2048, GSS_MAXK = 512;
struct GsSmallArgs {
    const double* basis;
    double* t;
    double* out;
    double eps1, eps2;
    int ldb, k, n, maxiter;
};

__device__ __forceinline__ void gs_small_vb(const VB vb, GsSmallArgs a) {
    __shared__ double ts[GSS_MAXN];
    __shared__ double cs[GSS_MAXK];
    __shared__ double red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = a.n, k = a.k;
    (void)vb;
    for (int i = tid; i < n; i += 256) ts[i] = a.t[i];
    __syncthreads();
    int flip = 0;
    // |t|^2 -> return value; t <- t / |t|   (launch_normalize)
    auto normalise = [&]() -> double {
        double sacc = 0.0;
        for (int i = tid; i < n; i += 256) sacc += ts[i] * ts[i];
        sacc = wave_sum64(sacc);
        if (lane == 0) red[flip][wave] = sacc;
        __syncthreads();
        const double tot = (red[flip][0] + red[flip][1]) + (red[flip][2] + red[flip][3]);
        flip ^= 1;                                           // (the next reduction writes the other buffer: no barrier needed here)
        const double f = 1.0 / sqrt(tot);
        for (int i = tid; i < n; i += 256) ts[i] *= f;
        __syncthreads();
        return tot;
    };
    // c = -(V t); t += V^T c   (gs_sweep)
    auto sweep = [&]() {
        if (k <= 0) return;
        for (int r = wave; r < k; r += 4) {
            const double* row = a.basis + (size_t)r * a.ldb;
            double d = 0.0;
            for (int i = lane; i < n; i += 64) d += row[i] * ts[i];
            d = wave_sum64(d);
            if (lane == 0) cs[r] = -d;
        }
        __syncthreads();
        for (int i = tid; i < n; i += 256) {
            double acc = ts[i];
            for (int r = 0; r < k; ++r) acc += cs[r] * a.basis[(size_t)r * a.ldb + i];
            ts[i] = acc;
        }
        __syncthreads();
    };
    const double n0sq = normalise();
    sweep();
    const double n1sq = normalise();
    sweep();
    double n2sq = normalise();
    int sweeps = 2, kept = 0, failed = 0;
    const double n1 = sqrt(n1sq);
    double n2 = sqrt(n2sq);
    if (n0sq > 0.0 && n1 == n1 && !(n1 < a.eps2)) {
        failed = 1;
        for (int it = 0; it < a.maxiter; ++it) {
            if (n2 != n2 || n2 < a.eps2) { failed = 0; break; }
            if (fabs(1.0 - n2) <= a.eps1) { kept = 1; failed = 0; break; }
            sweep();
            n2sq = normalise();
            n2 = sqrt(n2sq);
            ++sweeps;
        }
    }
    for (int i = tid; i < n; i += 256) a.t[i] = ts[i];
    if (tid == 0) {
        a.out[0] = n0sq; a.out[1] = n1sq; a.out[2] = n2sq;
        a.out[3] = (double)kept; a.out[4] = (double)failed; a.out[5] = (double)sweeps;
    }
}
__global__ __launch_bounds__(256) void gs_small_kernel(GsSmallArgs a) { gs_small_vb(vb_hw(), a); }

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
    if (c->opt.gs_small && n <= GSS_MAXN && n <= c->opt.gs_small && k <= GSS_MAXK) {
        GsSmallArgs a;
        a.basis = basis; a.t = t; a.out = scal_out(c, 8);
        a.eps1 = eps1; a.eps2 = eps2; a.ldb = ldb; a.k = k; a.n = n; a.maxiter = maxiter;
        SELLA_LAUNCHB(c, gs_small_kernel, gs_small_vb, 256, dim3(1), dim3(256), 0, a);
        HIPCHK(hipGetLastError());
        SCHK(sync_scalars(c, 8, 6));
        if (first_norm) *first_norm = sqrt(c->hscal[9]);
        if (c->hscal[12] != 0.0) { set_error("MGS failed."); return SELLA_E_NOCONV; }
        *kept = c->hscal[11] != 0.0 ? 1 : 0;
        return SELLA_OK;
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
