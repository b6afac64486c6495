// davidson.hip — Davidson / Rayleigh-Ritz partial diagonalisation, device-resident.
//
// Mirrors rayleigh_ritz + expand of the reference (sella/eigensolvers.py:31-153) with an
// MI355X-first formulation:
//   * the Krylov panels V, AV live on the device, vector-major (k rows x n), and are rotated
//     into the Ritz basis every iteration exactly like eigensolvers.py:62-64;
//   * the k x k Gram matrices V^T V and V^T AV are kept on the host and updated incrementally
//     (one row/column per new vector) instead of being recomputed with O(n k^2) work;
//   * the correction equation (P - theta I)^-1 is applied through the eigendecomposition
//     P = Q diag(d) Q^T that the approximate Hessian already caches on the device:
//     two row-panel matvecs with 2 right-hand sides (16 n^2 bytes) instead of the reference's
//     fresh (n+1)^2 LU per iteration (eigensolvers.py:133-139).  jd0 and jd0_alt are the same
//     algebra (eigensolvers.py:123-132);
//   * orthogonalisation against V is iterated classical Gram-Schmidt (two panel matvecs per
//     sweep) with the reference's accept/drop thresholds (math.pyx:105-133).
// Three host synchronisations per iteration (residual norms, Gram-Schmidt norms, new Gram row).
#include "internal.h"
#include "host_math.h"

#include <chrono>

namespace sella {
namespace {

using hostm::vec;

struct Dav {
    sella_ctx* c = nullptr;
    int n = 0, ld = 0, cap = 0, k = 0;
    double *Vp = nullptr, *AVp = nullptr, *Vq = nullptr, *AVq = nullptr;   // ping-pong panels
    double *Rp = nullptr;        // residual panel (cap rows)
    double *wk = nullptr;        // work vectors: 4 x 8 rows of ld (pinv input / mid / output, t)
    double *dW = nullptr;        // device copy of small coefficient matrices
    int capW = 0;
    vec Gvv, Gva;                // k x k with leading dimension cap
    // operator / preconditioner
    const Mat* A = nullptr;
    sella_matvec_fn matvec = nullptr;
    void* user = nullptr;
    const Mat *Q = nullptr, *Qt = nullptr;
    const double* pevals_dev = nullptr;
    double pscale = 1.0;
    int nmatvec = 0;
    size_t pbytes = 0;           // size of each panel allocation
    vec hv, hav;                 // host staging for the callback operator
};

int dav_alloc(Dav& s, int cap) {
    sella_ctx* c = s.c;
    const size_t pbytes = (size_t)cap * s.ld * sizeof(double);
    // panels are re-allocated on growth; old contents are copied
    double *nV, *nAV, *nVq, *nAVq, *nR;
    // use dedicated allocations (not the scratch pool) so growth can copy old -> new
    if (dev_alloc(c, pbytes, &nV) != SELLA_OK || dev_alloc(c, pbytes, &nAV) != SELLA_OK ||
        dev_alloc(c, pbytes, &nVq) != SELLA_OK || dev_alloc(c, pbytes, &nAVq) != SELLA_OK ||
        dev_alloc(c, pbytes, &nR) != SELLA_OK) {
        set_error("davidson: cannot allocate panels for %d vectors of length %d", cap, s.n);
        return SELLA_E_NOMEM;
    }
    HIPCHK(hipMemsetAsync(nV, 0, pbytes, c->stream));
    HIPCHK(hipMemsetAsync(nAV, 0, pbytes, c->stream));
    HIPCHK(hipMemsetAsync(nVq, 0, pbytes, c->stream));
    HIPCHK(hipMemsetAsync(nAVq, 0, pbytes, c->stream));
    HIPCHK(hipMemsetAsync(nR, 0, pbytes, c->stream));
    if (s.Vp) {
        const size_t old = (size_t)s.k * s.ld * sizeof(double);
        if (old) {
            HIPCHK(hipMemcpyAsync(nV, s.Vp, old, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(hipMemcpyAsync(nAV, s.AVp, old, hipMemcpyDeviceToDevice, c->stream));
        }
        dev_free(c, s.Vp, s.pbytes); dev_free(c, s.AVp, s.pbytes); dev_free(c, s.Vq, s.pbytes);
        dev_free(c, s.AVq, s.pbytes); dev_free(c, s.Rp, s.pbytes);
    }
    s.Vp = nV; s.AVp = nAV; s.Vq = nVq; s.AVq = nAVq; s.Rp = nR;
    s.pbytes = pbytes;
    // Gram matrices: re-layout with the new leading dimension
    vec gvv((size_t)cap * cap, 0.0), gva((size_t)cap * cap, 0.0);
    for (int i = 0; i < s.k; ++i)
        for (int j = 0; j < s.k; ++j) {
            gvv[(size_t)i * cap + j] = s.Gvv[(size_t)i * s.cap + j];
            gva[(size_t)i * cap + j] = s.Gva[(size_t)i * s.cap + j];
        }
    s.Gvv.swap(gvv);
    s.Gva.swap(gva);
    s.cap = cap;
    return SELLA_OK;
}

void dav_free(Dav& s) {
    if (s.c) (void)hipStreamSynchronize(s.c->stream);
    if (s.Vp) {
        dev_free(s.c, s.Vp, s.pbytes); dev_free(s.c, s.AVp, s.pbytes); dev_free(s.c, s.Vq, s.pbytes);
        dev_free(s.c, s.AVq, s.pbytes); dev_free(s.c, s.Rp, s.pbytes);
    }
    s.Vp = nullptr;
}

// Device copy of a small host array through one of three pinned staging slots (a slot is not
// re-used before the next host synchronisation, of which there is at least one per iteration).
int put_small(Dav& s, const double* h, int count, int stage, size_t dev_offset, double** dptr) {
    sella_ctx* c = s.c;
    double* base;
    SCHK(scratch_get(c, SCR_W, (size_t)(4 * (size_t)s.cap * s.cap + 4 * (size_t)s.cap + 256) * sizeof(double), &base));
    double* d = base + dev_offset;
    if (count <= 8192) {
        double* st = c->hscal + DS_STAGE + (size_t)stage * 8192;
        memcpy(st, h, (size_t)count * sizeof(double));
        HIPCHK(hipMemcpyAsync(d, st, (size_t)count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    } else {
        HIPCHK(hipMemcpyAsync(d, h, (size_t)count * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    *dptr = d;
    return SELLA_OK;
}

// y = A x on the device (dense) or through the host callback
int apply_A(Dav& s, const double* x, double* y) {
    sella_ctx* c = s.c;
    s.nmatvec++;
    if (s.A) return launch_gemv_rows(c, s.A->d, s.n, s.n, s.A->ld, x, s.ld, 1, y, s.ld, GemvEpi());
    s.hv.resize(s.n);
    s.hav.resize(s.n);
    HIPCHK(hipMemcpyAsync(s.hv.data(), x, (size_t)s.n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (s.matvec(s.user, s.hv.data(), s.hav.data(), s.n) != 0) {
        set_error("davidson: host matvec callback failed");
        return SELLA_E_CALLBACK;
    }
    HIPCHK(hipMemcpyAsync(y, s.hav.data(), (size_t)s.n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SELLA_OK;
}

// out(m rows) = (P - theta I)^-1 in(m rows), all panels with leading dimension ld
int apply_pinv(Dav& s, double theta, const double* in, int m, double* mid, double* out) {
    sella_ctx* c = s.c;
    if (s.Q == nullptr) {
        for (int h = 0; h < m; ++h)
            SCHK(launch_axpby(c, s.n, 1.0 / (s.pscale - theta), in + (size_t)h * s.ld, 0.0, nullptr,
                              out + (size_t)h * s.ld));
        return SELLA_OK;
    }
    GemvEpi e;
    e.mode = 1;
    e.dvec = s.pevals_dev;
    e.theta = theta;
    SCHK(launch_gemv_rows(c, s.Qt->d, s.n, s.n, s.Qt->ld, in, s.ld, m, mid, s.ld, e));
    return launch_gemv_rows(c, s.Q->d, s.n, s.n, s.Q->ld, mid, s.ld, m, out, s.ld, GemvEpi());
}

// same with the inputs given as separate vectors
int apply_pinv_xp(Dav& s, double theta, const double* const* in, int m, double* mid, double* out) {
    sella_ctx* c = s.c;
    if (s.Q == nullptr) {
        for (int h = 0; h < m; ++h)
            SCHK(launch_axpby(c, s.n, 1.0 / (s.pscale - theta), in[h], 0.0, nullptr, out + (size_t)h * s.ld));
        return SELLA_OK;
    }
    GemvEpi e;
    e.mode = 1;
    e.dvec = s.pevals_dev;
    e.theta = theta;
    SCHK(launch_gemv_rows_xp(c, s.Qt->d, s.n, s.n, s.Qt->ld, in, m, mid, s.ld, e));
    return launch_gemv_rows(c, s.Q->d, s.n, s.n, s.Q->ld, mid, s.ld, m, out, s.ld, GemvEpi());
}

// Orthonormalise t against V[0:k) with the reference's accept / drop rules (gs.hip).
int orthonormalise(Dav& s, double* t, int k, int* kept, double* first_norm) {
    return gs_orthonormalise(s.c, s.Vp, s.ld, k, t, s.n, 1e-15, 1e-6, 100, kept, first_norm);
}

// Append the unit vector in panel slot k (already orthonormalised) and its image A t; update
// the Gram matrices with one synchronisation.
int append_vector(Dav& s) {
    sella_ctx* c = s.c;
    const int k = s.k, cap = s.cap;
    double* t = s.Vp + (size_t)k * s.ld;
    double* At = s.AVp + (size_t)k * s.ld;
    SCHK(apply_A(s, t, At));
    double* ds = scal_out(c, DS_GRAM);
    // rows [0,k]: V_a.t (a = k gives t.t) ; [cap, cap+k]: V_a.At (a = k gives t.At) ; [2cap, 2cap+k): AV_a.t
    const double* xs[2] = {t, At};
    SCHK(launch_gemv_rows_xp(c, s.Vp, k + 1, s.n, s.ld, xs, 2, ds, cap, GemvEpi()));
    if (k > 0) SCHK(launch_gemv_rows_xp(c, s.AVp, k, s.n, s.ld, xs, 1, ds + 2 * (size_t)cap, cap, GemvEpi()));
    SCHK(sync_scalars(c, DS_GRAM, 3 * cap));
    const double* h = c->hscal + DS_GRAM;
    for (int a = 0; a < k; ++a) {
        s.Gvv[(size_t)a * cap + k] = s.Gvv[(size_t)k * cap + a] = h[a];
        s.Gva[(size_t)a * cap + k] = h[cap + a];
        s.Gva[(size_t)k * cap + a] = h[2 * cap + a];
    }
    s.Gvv[(size_t)k * cap + k] = h[k];
    s.Gva[(size_t)k * cap + k] = h[cap + k];
    s.k = k + 1;
    return SELLA_OK;
}

// pack the leading k x k block (ld cap) into a tight k x k array
void pack(const vec& G, int cap, int k, vec& out) {
    out.resize((size_t)k * k);
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) out[(size_t)i * k + j] = G[(size_t)i * cap + j];
}
void unpack(const vec& in, int k, vec& G, int cap) {
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) G[(size_t)i * cap + j] = in[(size_t)i * k + j];
}

}  // namespace
}  // namespace sella

using namespace sella;

extern "C" int sella_davidson(sella_ctx* c, sella_mat hA, sella_matvec_fn matvec, void* user,
                              sella_mat hPvecs, sella_mat hPvecsT, const double* pevals, double pscale,
                              int n, const double* v0, int nv0, double gamma, int method, int maxiter,
                              const double* vref, double vreftol, double* lams_out, double* V_out,
                              double* AV_out, int* k_out, int* nmatvec_out) {
    if (!c || n <= 0 || !v0 || nv0 <= 0 || nv0 > n || !lams_out || !V_out || !AV_out || !k_out) {
        set_error("davidson: invalid arguments");
        return SELLA_E_INVALID;
    }
    if (n > 16000) {
        set_error("davidson: n = %d exceeds the scalar exchange layout (16000)", n);
        return SELLA_E_UNSUPPORTED;
    }
    if (method < SELLA_DAV_LANCZOS || method > SELLA_DAV_MJD0_ALT) {
        set_error("Unknown diagonalization method %d", method);
        return SELLA_E_INVALID;
    }
    if (!(gamma > 0.0)) {
        set_error("davidson: gamma must be > 0 (gamma <= 0 means exact diagonalisation: use sella_eigh)");
        return SELLA_E_INVALID;
    }
    Dav s;
    s.c = c;
    s.n = n;
    s.ld = round_up(n, 8);
    if (hA != SELLA_NO_MAT) {
        s.A = mat_get(c, hA);
        if (!s.A) return SELLA_E_INVALID;
        if (s.A->rows != n || s.A->cols != n) { set_error("davidson: A must be %d x %d", n, n); return SELLA_E_INVALID; }
    } else if (!matvec) {
        set_error("davidson: neither a resident matrix nor a matvec callback was given");
        return SELLA_E_INVALID;
    }
    s.matvec = matvec;
    s.user = user;
    s.pscale = pscale;
    if (hPvecs != SELLA_NO_MAT) {
        s.Q = mat_get(c, hPvecs);
        s.Qt = mat_get(c, hPvecsT);
        if (!s.Q || !s.Qt || !pevals) { set_error("davidson: P needs Pvecs, PvecsT and pevals"); return SELLA_E_INVALID; }
        if (s.Q->rows != n || s.Q->cols != n || s.Qt->rows != n || s.Qt->cols != n) {
            set_error("davidson: eigenvector matrices of P must be %d x %d", n, n);
            return SELLA_E_INVALID;
        }
        double* dev;
        SCHK(scratch_get(c, SCR_C, (size_t)s.ld * sizeof(double), &dev));
        HIPCHK(hipMemcpyAsync(dev, pevals, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        s.pevals_dev = dev;
    }
    if (maxiter <= 0) maxiter = 2 * n + 1;
    const int kstop = (n < maxiter) ? n : maxiter;

    int cap0 = 64;
    while (cap0 < nv0 + 2) cap0 *= 2;
    if (cap0 > n + 1) cap0 = n + 1;
    if (cap0 < nv0 + 1) cap0 = nv0 + 1;
    int st = dav_alloc(s, cap0);
    if (st != SELLA_OK) return st;
    SCHK(scratch_get(c, SCR_T, (size_t)40 * s.ld * sizeof(double), &s.wk));

    auto fail = [&](int code) { dav_free(s); return code; };
#define DCHK(expr) do { int s__ = (expr); if (s__ != SELLA_OK) return fail(s__); } while (0)

    // ---- start block: V = mgs(v0) (eigensolvers.py:44-50), AV = A V ----------------------
    {
        double* tmp;
        DCHK(scratch_get(c, SCR_X, (size_t)nv0 * s.ld * sizeof(double), &tmp));
        DCHK(upload_panel(c, v0, n, nv0, tmp, s.ld));
        for (int j = 0; j < nv0; ++j) {
            double* slot = s.Vp + (size_t)s.k * s.ld;
            if (hipMemcpyAsync(slot, tmp + (size_t)j * s.ld, (size_t)s.ld * sizeof(double), hipMemcpyDeviceToDevice,
                               c->stream) != hipSuccess) return fail(SELLA_E_HIP);
            int kept = 0;
            DCHK(orthonormalise(s, slot, s.k, &kept, nullptr));
            if (kept) DCHK(append_vector(s));
        }
        if (s.k == 0) {
            set_error("davidson: the start block is numerically zero");
            return fail(SELLA_E_INVALID);
        }
    }

    vec lams, W, gvv, gva, X, At, tmpm, coef;
    int seeking = 0;
    unsigned long long lcg = 0x9E3779B97F4A7C15ull;   // deterministic stand-in for np.random.normal (:107)

    double t_host = 0.0;
    const bool dbg_time = getenv("SELLA_DEBUG_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    while (true) {
        const int k = s.k, cap = s.cap;
        double th0 = now();
        // ---- Rayleigh-Ritz (eigensolvers.py:57-64) ---------------------------------------
        pack(s.Gvv, cap, k, gvv);
        pack(s.Gva, cap, k, gva);
        X.assign((size_t)k * k, 0.0);
        hostm::symm_coeffs(k, gvv.data(), gva.data(), 2, X.data());
        // Atilde = V^T (AV + V X) = Gva + Gvv X
        At.assign((size_t)k * k, 0.0);
        for (int a = 0; a < k; ++a)
            for (int b = 0; b < k; ++b) {
                double v = gva[(size_t)a * k + b];
                for (int l = 0; l < b; ++l) v += gvv[(size_t)a * k + l] * X[(size_t)l * k + b];
                At[(size_t)a * k + b] = v;
            }
        lams.assign(k, 0.0);
        W.assign((size_t)k * k, 0.0);
        if (hostm::gen_sym_eig(k, At.data(), gvv.data(), lams.data(), W.data()) != 0) {
            set_error("davidson: Rayleigh-Ritz eigenproblem failed (V^T V not positive definite?)");
            return fail(SELLA_E_NOCONV);
        }
        t_host += now() - th0;
        int nneg = 0;
        for (int i = 0; i < k; ++i) nneg += (lams[i] < 0.0);
        if (nneg < 1) nneg = 1;
        // rotate the panels into the Ritz basis: V <- V W, AV <- AV W
        {
            double* dWp;
            DCHK(put_small(s, W.data(), k * k, 0, 0, &dWp));
            DCHK(launch_lincomb(c, n, k, s.Vp, s.ld, k, dWp, k, nullptr, 0, 0, nullptr, 0, 0.0, s.Vq, s.ld));
            DCHK(launch_lincomb(c, n, k, s.AVp, s.ld, k, dWp, k, nullptr, 0, 0, nullptr, 0, 0.0, s.AVq, s.ld));
            std::swap(s.Vp, s.Vq);
            std::swap(s.AVp, s.AVq);
            th0 = now();
            tmpm.resize((size_t)k * k);
            hostm::congruence(k, W.data(), gvv.data(), tmpm.data());
            unpack(tmpm, k, s.Gvv, cap);
            gvv = tmpm;
            hostm::congruence(k, W.data(), gva.data(), tmpm.data());
            unpack(tmpm, k, s.Gva, cap);
            gva = tmpm;
            t_host += now() - th0;
        }
        if (k >= kstop) break;                                            // :65-66

        // ---- residuals of the leading nneg Ritz pairs (:68-71) ---------------------------
        hostm::symm_coeffs(k, gvv.data(), gva.data(), 2, X.data());
        // R_j = AV_j + sum_{l<j} X[l][j] V_l - lams[j] V_j
        // coefficient block 0 multiplies the V rows, block 1 (identity) selects the AV rows
        coef.assign((size_t)2 * nneg * nneg, 0.0);
        for (int j = 0; j < nneg; ++j) {
            for (int l = 0; l < j; ++l) coef[(size_t)l * nneg + j] = X[(size_t)l * k + j];
            coef[(size_t)j * nneg + j] = -lams[j];
            coef[(size_t)nneg * nneg + (size_t)j * nneg + j] = 1.0;
        }
        {
            double* dC;
            DCHK(put_small(s, coef.data(), 2 * nneg * nneg, 1, (size_t)s.cap * s.cap + 8, &dC));
            DCHK(launch_lincomb(c, n, nneg, s.Vp, s.ld, nneg, dC, nneg, s.AVp, s.ld, nneg, dC + (size_t)nneg * nneg, nneg,
                                0.0, s.Rp, s.ld));
            DCHK(launch_rows_sumsq(c, s.Rp, s.ld, nneg, n, scal_out(c, 0)));
            int nread = nneg;
            if (vref) {
                double* dv = s.wk + 39 * (size_t)s.ld;
                if (hipMemcpyAsync(dv, vref, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess)
                    return fail(SELLA_E_HIP);
                DCHK(launch_gemv_rows(c, s.Vp, 1, n, s.ld, dv, s.ld, 1, scal_out(c, nneg), 1, GemvEpi()));
                nread += 1;
            }
            if (nread > 4000) { set_error("davidson: too many negative Ritz values (%d)", nneg); return fail(SELLA_E_UNSUPPORTED); }
            DCHK(sync_scalars(c, 0, nread));
        }
        if (vref && fabs(c->hscal[nneg]) > vreftol) break;                // :74-77
        seeking = -1;
        for (int i = 0; i < nneg; ++i) {                                  // :80-89
            const double rnorm = sqrt(c->hscal[i]);
            if (k == 1 || rnorm >= gamma * fabs(lams[i])) { seeking = i; break; }
        }
        if (seeking < 0) break;
        const double theta = lams[seeking];
        const double* r = s.Rp + (size_t)seeking * s.ld;
        const double* v = s.Vp + (size_t)seeking * s.ld;

        if (s.k + 1 > s.cap) {
            int ncap = s.cap * 2;
            if (ncap > n + 1) ncap = n + 1;
            DCHK(dav_alloc(s, ncap));
            r = s.Rp + (size_t)seeking * s.ld;    // Rp was re-allocated: recompute the residual rows
            v = s.Vp + (size_t)seeking * s.ld;
            double* dC;
            DCHK(put_small(s, coef.data(), 2 * nneg * nneg, 1, (size_t)s.cap * s.cap + 8, &dC));
            DCHK(launch_lincomb(c, n, nneg, s.Vp, s.ld, nneg, dC, nneg, s.AVp, s.ld, nneg, dC + (size_t)nneg * nneg, nneg,
                                0.0, s.Rp, s.ld));
        }

        // ---- correction vector (expand, :115-153) into panel slot k -----------------------
        double* t = s.Vp + (size_t)s.k * s.ld;
        double* in = s.wk + 8 * (size_t)s.ld;      // up to 8 rows
        double* mid = s.wk + 16 * (size_t)s.ld;
        double* out = s.wk + 24 * (size_t)s.ld;
        auto copy_row = [&](double* dst, const double* src) -> int {
            HIPCHK(hipMemcpyAsync(dst, src, (size_t)s.ld * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            return SELLA_OK;
        };
        if (method == SELLA_DAV_LANCZOS) {
            DCHK(copy_row(t, r));
        } else if (method == SELLA_DAV_GD) {
            const double* rr[1] = {r};
            DCHK(apply_pinv_xp(s, theta, rr, 1, mid, t));
        } else if (method == SELLA_DAV_JD0 || method == SELLA_DAV_JD0_ALT) {
            const double* rv[2] = {r, v};
            DCHK(apply_pinv_xp(s, theta, rv, 2, mid, out));
            // dots[0] = v.x, dots[1] = v.y
            DCHK(launch_gemv_rows(c, v, 1, n, s.ld, out, s.ld, 2, c->dscal + 16, 1, GemvEpi()));
            DCHK(launch_jd_combine(c, out, out + s.ld, c->dscal + 16, t, n));
        } else {
            // mjd0 / mjd0_alt: z = Pinv(V alpha - r), (V^T Pinv V) alpha = V^T Pinv r   (:140-151)
            const int kk = s.k;
            double *pv, *pmid;
            DCHK(scratch_get(c, SCR_V2, (size_t)(kk + 1) * s.ld * sizeof(double), &pv));
            DCHK(scratch_get(c, SCR_AV2, (size_t)(kk + 1) * s.ld * sizeof(double), &pmid));
            double* pin;
            DCHK(scratch_get(c, SCR_R, (size_t)(kk + 1) * s.ld * sizeof(double), &pin));
            HIPCHK(hipMemcpyAsync(pin, s.Vp, (size_t)kk * s.ld * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            DCHK(copy_row(pin + (size_t)kk * s.ld, r));
            DCHK(apply_pinv(s, theta, pin, kk + 1, pmid, pv));
            // G = V^T [Pinv V | Pinv r]  -> (kk+1) columns of kk entries
            double* dg = c->dscal + DS_CVEC;
            if ((kk + 1) * kk > 16000) { set_error("davidson: mjd0 subspace too large"); return fail(SELLA_E_UNSUPPORTED); }
            DCHK(launch_gemv_rows(c, s.Vp, kk, n, s.ld, pv, s.ld, kk + 1, dg, kk, GemvEpi()));
            DCHK(read_scalars(c, DS_CVEC, (kk + 1) * kk));
            // host: G[h*kk + a] = V_a . Pinv(col h)
            vec Gm((size_t)kk * kk), rhs(kk);
            for (int a = 0; a < kk; ++a) {
                for (int b = 0; b < kk; ++b) Gm[(size_t)a * kk + b] = c->hscal[DS_CVEC + (size_t)b * kk + a];
                rhs[a] = c->hscal[DS_CVEC + (size_t)kk * kk + a];
            }
            std::vector<int> piv(kk);
            if (small::lu_factor(kk, Gm.data(), kk, piv.data()) != 0) {
                set_error("davidson: singular projected preconditioner in mjd0");
                return fail(SELLA_E_NOCONV);
            }
            small::lu_solve(kk, Gm.data(), kk, piv.data(), rhs.data(), 1, 1);
            // in = V alpha - r ; t = Pinv(in)
            double* dal;
            DCHK(put_small(s, rhs.data(), kk, 2, 2 * (size_t)s.cap * s.cap + 16, &dal));
            DCHK(launch_axpby(c, n, -1.0, r, 0.0, nullptr, in));
            DCHK(launch_lincomb(c, n, 1, s.Vp, s.ld, kk, dal, 1, nullptr, 0, 0, nullptr, 0, 1.0, in, s.ld));
            DCHK(apply_pinv(s, theta, in, 1, mid, t));
        }

        // ---- normalise, Lanczos safeguard, orthogonalise (:92-109) -------------------------
        int kept = 0;
        double n1 = 0.0;
        DCHK(orthonormalise(s, t, s.k, &kept, &n1));
        if (n1 < 1e-2) {                                                   // :93-95 "Do Lanczos instead"
            DCHK(copy_row(t, r));
            DCHK(orthonormalise(s, t, s.k, &kept, nullptr));
        }
        if (!kept) {                                                       // :100-109
            for (int j = 0; j < nneg && !kept; ++j) {
                DCHK(copy_row(t, s.Rp + (size_t)j * s.ld));
                DCHK(orthonormalise(s, t, s.k, &kept, nullptr));
            }
            if (!kept) {
                vec rnd(n);
                for (int i = 0; i < n; ++i) {   // sum of 12 uniforms - 6: deterministic, ~normal
                    double acc = 0.0;
                    for (int q = 0; q < 12; ++q) {
                        lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
                        acc += (double)(lcg >> 11) * (1.0 / 9007199254740992.0);
                    }
                    rnd[i] = acc - 6.0;
                }
                if (hipMemcpyAsync(t, rnd.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess)
                    return fail(SELLA_E_HIP);
                if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(SELLA_E_HIP);
                DCHK(orthonormalise(s, t, s.k, &kept, nullptr));
                if (!kept) break;
            }
        }
        DCHK(append_vector(s));
    }

    if (dbg_time) fprintf(stderr, "davidson: k=%d total %.3f ms, host k x k algebra %.3f ms\n", s.k, 1e3 * (now() - t_begin), 1e3 * t_host);
    // ---- results: Ritz values, V and AV as (n x k) row-major host arrays ------------------
    const int k = s.k;
    for (int i = 0; i < k; ++i) lams_out[i] = lams[i];
    DCHK(download_panel(c, s.Vp, s.ld, n, k, V_out));
    DCHK(download_panel(c, s.AVp, s.ld, n, k, AV_out));
    *k_out = k;
    if (nmatvec_out) *nmatvec_out = s.nmatvec;
    dav_free(s);
    return SELLA_OK;
#undef DCHK
}
