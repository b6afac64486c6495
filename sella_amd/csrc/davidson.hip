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
    unsigned it = 0;             // iteration counter (position of a cohort member inside the loop, cohort.h)
    double *Vp = nullptr, *AVp = nullptr, *Vq = nullptr, *AVq = nullptr;   // ping-pong panels
    double *Rp = nullptr;        // residual panel (cap rows)
    // carried images of the panels in the eigenbasis of P (fused iteration with an eigenbasis preconditioner):
    // QtV_a = Q^T V_a, QtAV_a = Q^T AV_a, rows [0, k); one allocation, QtAV = QtV + cap * ld
    double *QtV = nullptr, *QtAV = nullptr;
    bool qt_mode = false;
    hipEvent_t ev = nullptr;
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
    int prank = 0;               // explicit eigenpairs of P (== n: full eigendecomposition; < n: the complement of
                                 // their span is the eigenspace of pscale, see apply_pinv)
    int nmatvec = 0;
    size_t pbytes = 0;           // size of each panel allocation
    double* dcoef = nullptr;     // device copy of the residual coefficients of the current iteration
    vec hv, hav;                 // host staging for the callback operator
};

int dav_alloc(Dav& s, int cap) {
    sella_ctx* c = s.c;
    const size_t pbytes = (size_t)cap * s.ld * sizeof(double);
    // panels are re-allocated on growth; old contents are copied
    double *nV, *nAV, *nVq, *nAVq, *nR, *nQt = nullptr;
    // use dedicated allocations (not the scratch pool) so growth can copy old -> new
    if (dev_alloc(c, pbytes, &nV) != SELLA_OK || dev_alloc(c, pbytes, &nAV) != SELLA_OK ||
        dev_alloc(c, pbytes, &nVq) != SELLA_OK || dev_alloc(c, pbytes, &nAVq) != SELLA_OK ||
        dev_alloc(c, pbytes, &nR) != SELLA_OK) {
        set_error("davidson: cannot allocate panels for %d vectors of length %d", cap, s.n);
        return SELLA_E_NOMEM;
    }
    HIPCHK(s_memset0(c, nV, pbytes));
    HIPCHK(s_memset0(c, nAV, pbytes));
    HIPCHK(s_memset0(c, nVq, pbytes));
    HIPCHK(s_memset0(c, nAVq, pbytes));
    HIPCHK(s_memset0(c, nR, pbytes));
    if (s.qt_mode) {
        if (dev_alloc(c, 2 * pbytes, &nQt) != SELLA_OK) { set_error("davidson: cannot allocate the eigenbasis panels"); return SELLA_E_NOMEM; }
        HIPCHK(s_memset0(c, nQt, 2 * pbytes));
    }
    if (s.Vp) {
        const size_t old = (size_t)s.k * s.ld * sizeof(double);
        if (old) {
            HIPCHK(s_memcpy(c, nV, s.Vp, old, hipMemcpyDeviceToDevice));
            HIPCHK(s_memcpy(c, nAV, s.AVp, old, hipMemcpyDeviceToDevice));
            if (s.qt_mode && s.QtV) {
                HIPCHK(s_memcpy(c, nQt, s.QtV, old, hipMemcpyDeviceToDevice));
                HIPCHK(s_memcpy(c, nQt + (size_t)cap * s.ld, s.QtAV, old, hipMemcpyDeviceToDevice));
            }
        }
        if (s.QtV) dev_free(c, s.QtV, 2 * s.pbytes);
        dev_free(c, s.Vp, s.pbytes); dev_free(c, s.AVp, s.pbytes); dev_free(c, s.Vq, s.pbytes);
        dev_free(c, s.AVq, s.pbytes); dev_free(c, s.Rp, s.pbytes);
    }
    s.Vp = nV; s.AVp = nAV; s.Vq = nVq; s.AVq = nAVq; s.Rp = nR;
    s.QtV = nQt;
    s.QtAV = nQt ? nQt + (size_t)cap * s.ld : nullptr;
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
    if (s.c) (void)stream_sync_raw(s.c);
    if (s.Vp) {
        dev_free(s.c, s.Vp, s.pbytes); dev_free(s.c, s.AVp, s.pbytes); dev_free(s.c, s.Vq, s.pbytes);
        dev_free(s.c, s.AVq, s.pbytes); dev_free(s.c, s.Rp, s.pbytes);
        if (s.QtV) dev_free(s.c, s.QtV, 2 * s.pbytes);
    }
    if (s.ev) (void)hipEventDestroy(s.ev);
    s.ev = nullptr;
    s.QtV = s.QtAV = nullptr;
    s.Vp = nullptr;
}

// Device copy of a small host array through one of three pinned staging slots (a slot is not
// re-used before the next host synchronisation, of which there is at least one per iteration).
// pscale - theta with the exact-hit guard of the eigenbasis epilogue (kernels.hip, GemvEpi mode 1)
inline double guarded_shift(double d, double theta) {
    const double den = d - theta;
    return den != 0.0 ? den : 2.220446049250313e-16 * fmax(fabs(theta), 2.2250738585072014e-308);
}

int put_small(Dav& s, const double* h, int count, int stage, size_t dev_offset, double** dptr) {
    sella_ctx* c = s.c;
    double* base;
    SCHK(scratch_get(c, SCR_W, (size_t)(4 * (size_t)s.cap * s.cap + 4 * (size_t)s.cap + 256) * sizeof(double), &base));
    double* d = base + dev_offset;
    if (count <= 8192) {
        double* st = c->hscal + DS_STAGE + (size_t)stage * 8192;
        memcpy(st, h, (size_t)count * sizeof(double));
        SCHK(h2d_pinned(c, d, st, (size_t)count * sizeof(double)));          // (by kernel from 16 KB on)
    } else {
        SCHK(h2d_async(c, d, h, (size_t)count * sizeof(double)));
        SCHK(stream_wait(c));
    }
    *dptr = d;
    return SELLA_OK;
}

// y = A x on the device (dense) or through the host callback
int apply_A(Dav& s, const double* x, double* y) {
    sella_ctx* c = s.c;
    s.nmatvec++;
    if (s.A) return launch_gemv_rows(c, s.A->d, s.n, s.n, s.A->ld, x, s.ld, 1, y, s.ld, GemvEpi());
    cohort_barrier(c, s.it, 2);               // (members of a cohort make the force calls behind their products together)
    s.hv.resize(s.n);
    s.hav.resize(s.n);
    SCHK(d2h_async(c, s.hv.data(), x, (size_t)s.n * sizeof(double)));
    SCHK(stream_wait(c));
    {
        // the callback may re-enter the library on this context (NumericalHessian._matvec, sella/linalg.py:39-95, is
        // allowed to do anything): it runs on a working set of its own (internal.h, sella_ctx::Frame)
        CallbackScope scope(c);
        SCHK(scope.status);
        if (s.matvec(s.user, s.hv.data(), s.hav.data(), s.n) != 0) {
            set_error("davidson: host matvec callback failed");
            return SELLA_E_CALLBACK;
        }
    }
    // (no wait: the payload sits in the pinned ring from here on, the copy is ordered in front of whatever reads y)
    return h2d_async(c, y, s.hav.data(), (size_t)s.n * sizeof(double));
}

// out(m rows) = (P - theta I)^-1 in(m rows), all panels with leading dimension ld
int apply_pinv(Dav& s, double theta, const double* in, int m, double* mid, double* out) {
    sella_ctx* c = s.c;
    if (s.Q == nullptr) {
        for (int h = 0; h < m; ++h)
            SCHK(launch_axpby(c, s.n, 1.0 / guarded_shift(s.pscale, theta), in + (size_t)h * s.ld, 0.0, nullptr,
                              out + (size_t)h * s.ld));
        return SELLA_OK;
    }
    GemvEpi e;
    e.mode = 1;
    e.dvec = s.pevals_dev;
    e.theta = theta;
    if (s.prank < s.n) {
        // P = lam0 (I - W^T W) + W^T diag(mu) W:  (P - theta)^-1 x = x / (lam0 - theta) + W^T [(1/(mu - theta) - 1/(lam0 - theta)) W x]
        const double inv0 = 1.0 / guarded_shift(s.pscale, theta);
        for (int h = 0; h < m; ++h)
            SCHK(launch_axpby(c, s.n, inv0, in + (size_t)h * s.ld, 0.0, nullptr, out + (size_t)h * s.ld));
        if (s.prank == 0) return SELLA_OK;
        e.mode = 5;
        e.beta = inv0;
        SCHK(launch_gemv_rows(c, s.Qt->d, s.prank, s.n, s.Qt->ld, in, s.ld, m, mid, s.ld, e));
        GemvEpi acc;
        acc.mode = 2;
        acc.beta = 1.0;
        return launch_gemv_rows(c, s.Q->d, s.n, s.prank, s.Q->ld, mid, s.ld, m, out, s.ld, acc);
    }
    SCHK(launch_gemv_rows(c, s.Qt->d, s.n, s.n, s.Qt->ld, in, s.ld, m, mid, s.ld, e));
    return launch_gemv_rows(c, s.Q->d, s.n, s.n, s.Q->ld, mid, s.ld, m, out, s.ld, GemvEpi());
}

// same with the inputs given as separate vectors
int apply_pinv_xp(Dav& s, double theta, const double* const* in, int m, double* mid, double* out) {
    sella_ctx* c = s.c;
    if (s.Q == nullptr) {
        for (int h = 0; h < m; ++h)
            SCHK(launch_axpby(c, s.n, 1.0 / guarded_shift(s.pscale, theta), in[h], 0.0, nullptr, out + (size_t)h * s.ld));
        return SELLA_OK;
    }
    GemvEpi e;
    e.mode = 1;
    e.dvec = s.pevals_dev;
    e.theta = theta;
    if (s.prank < s.n) {
        const double inv0 = 1.0 / guarded_shift(s.pscale, theta);
        for (int h = 0; h < m; ++h) SCHK(launch_axpby(c, s.n, inv0, in[h], 0.0, nullptr, out + (size_t)h * s.ld));
        if (s.prank == 0) return SELLA_OK;
        e.mode = 5;
        e.beta = inv0;
        SCHK(launch_gemv_rows_xp(c, s.Qt->d, s.prank, s.n, s.Qt->ld, in, m, mid, s.ld, e));
        GemvEpi acc;
        acc.mode = 2;
        acc.beta = 1.0;
        return launch_gemv_rows(c, s.Q->d, s.n, s.prank, s.Q->ld, mid, s.ld, m, out, s.ld, acc);
    }
    SCHK(launch_gemv_rows_xp(c, s.Qt->d, s.n, s.n, s.Qt->ld, in, m, mid, s.ld, e));
    return launch_gemv_rows(c, s.Q->d, s.n, s.n, s.Q->ld, mid, s.ld, m, out, s.ld, GemvEpi());
}

// mid_h[i] = in_h[i] / (d[i] - theta), h < nh (<= 2): the diagonal of (P - theta)^-1 in P's eigenbasis, with the exact-hit
// guard of the matvec epilogue (kernels.hip, GemvEpi mode 1)
__device__ __forceinline__ void dav_eigscale_vb(const VB vb, int n, int nh, const double* __restrict__ in0,
                                                           const double* __restrict__ in1, const double* __restrict__ d,
                                                           double theta, double* __restrict__ mid, int ld) {
    const int i = vb.x * 256 + threadIdx.x;
    if (i >= n) return;
    double den = d[i] - theta;
    if (den == 0.0) den = 2.220446049250313e-16 * fmax(fabs(theta), 2.2250738585072014e-308);
    mid[i] = in0[i] / den;
    if (nh > 1) mid[(size_t)ld + i] = in1[i] / den;
}
__global__ __launch_bounds__(256) void dav_eigscale_kernel(int n, int nh, const double* __restrict__ in0,
                                                           const double* __restrict__ in1, const double* __restrict__ d,
                                                           double theta, double* __restrict__ mid, int ld) { dav_eigscale_vb(vb_hw(), n, nh, in0, in1, d, theta, mid, ld); }

// rows [row0, row0 + nrows) of the eigenbasis panels from the raw panels: QtV_a = Q^T V_a, QtAV_a = Q^T AV_a
int qt_update(Dav& s, int row0, int nrows) {
    sella_ctx* c = s.c;
    if (nrows == 1 && s.QtAV == s.QtV + (size_t)s.cap * s.ld) {
        // one vector: its image and the image of its product in ONE pass over Q^T (as the fused iteration does)
        const double* xs[2] = {s.Vp + (size_t)row0 * s.ld, s.AVp + (size_t)row0 * s.ld};
        return launch_gemv_rows_xp(c, s.Qt->d, s.n, s.n, s.Qt->ld, xs, 2, s.QtV + (size_t)row0 * s.ld, s.cap * s.ld, GemvEpi());
    }
    for (int r = row0; r < row0 + nrows; r += 8) {
        const int nr = std::min(8, row0 + nrows - r);
        SCHK(launch_gemv_rows(c, s.Qt->d, s.n, s.n, s.Qt->ld, s.Vp + (size_t)r * s.ld, s.ld, nr, s.QtV + (size_t)r * s.ld, s.ld,
                              GemvEpi()));
        SCHK(launch_gemv_rows(c, s.Qt->d, s.n, s.n, s.Qt->ld, s.AVp + (size_t)r * s.ld, s.ld, nr, s.QtAV + (size_t)r * s.ld,
                              s.ld, GemvEpi()));
    }
    return SELLA_OK;
}

// Orthonormalise t against V[0:k) with the reference's accept / drop rules (gs.hip).
int orthonormalise(Dav& s, double* t, int k, int* kept, double* first_norm) {
    return gs_orthonormalise(s.c, s.Vp, s.ld, k, t, s.n, 1e-15, 1e-6, 100, kept, first_norm);
}

// ---- Gram bookkeeping shared by both iteration paths ------------------------------------------------------------
// The device panels stay in the RAW basis (the vectors as they were appended); the host keeps the cumulative
// rotation Wc (k x k, raw -> current Ritz basis, eigensolvers.py:62-64 applied lazily) and the Gram matrices of
// the ROTATED basis, which is what symmetrize_Y2 and the Rayleigh-Ritz step of the reference see.  Dots of a new
// vector against the raw panels are rotated with Wc^T before they enter the Gram matrices.
void rotate_dots(const vec& Wc, int k, const double* raw, double* rot) {
    for (int j = 0; j < k; ++j) {
        double sacc = 0.0;
        for (int a = 0; a < k; ++a) sacc += Wc[(size_t)a * k + j] * raw[a];
        rot[j] = sacc;
    }
}

// raw dots of the new (unit) vector t and its image: vt[a] = V_a.t, vat[a] = V_a.At, avt[a] = AV_a.t (a < k),
// tt = t.t, tat = t.At  ->  new row / column k of the rotated Gram matrices; Wc grows by a unit diagonal entry
void gram_append(Dav& s, vec& Wc, const double* vt, const double* vat, const double* avt, double tt, double tat) {
    const int k = s.k, cap = s.cap;
    vec r1(k), r2(k), r3(k);
    rotate_dots(Wc, k, vt, r1.data());
    rotate_dots(Wc, k, vat, r2.data());
    rotate_dots(Wc, k, avt, r3.data());
    for (int a = 0; a < k; ++a) {
        s.Gvv[(size_t)a * cap + k] = s.Gvv[(size_t)k * cap + a] = r1[a];
        s.Gva[(size_t)a * cap + k] = r2[a];
        s.Gva[(size_t)k * cap + a] = r3[a];
    }
    s.Gvv[(size_t)k * cap + k] = tt;
    s.Gva[(size_t)k * cap + k] = tat;
    vec Wn((size_t)(k + 1) * (k + 1), 0.0);
    for (int a = 0; a < k; ++a)
        for (int b = 0; b < k; ++b) Wn[(size_t)a * (k + 1) + b] = Wc[(size_t)a * k + b];
    Wn[(size_t)k * (k + 1) + k] = 1.0;
    Wc.swap(Wn);
    s.k = k + 1;
}

// Append the unit vector in panel slot k (already orthonormalised) and its image A t; update
// the Gram matrices with one synchronisation.  (Slow path: separate launches.)
int append_vector(Dav& s, vec& Wc) {
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
    gram_append(s, Wc, h, h + cap, h + 2 * cap, h[k], h[cap + k]);
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


// ---- fused kernels of the one-synchronisation iteration -------------------------------------------------------------
// An iteration of the fast path is ONE dependent chain of 9 launches with a single host synchronisation at its end:
//   resid -> Q^T[r v] -> Q[. .] -> V.[x y] -> gs1 -> A.t1 , V.t1 -> gs2 -> final(+dots) -> scalars to the host.
// Everything the host used to decide between launches (convergence, the 1e-2 Lanczos safeguard, the Gram-Schmidt
// accept test) is evaluated from the scalars of that one read-back and VERIFIED after the fact: the chain is launched
// speculatively for the predicted pair, and a verdict that differs from the prediction (converged, another pair to
// pursue, a vector that needs a third sweep or is dropped) discards the speculative vector and repeats the iteration
// on the synchronous path, which is the reference's control flow step by step.  Results are therefore identical to
// the synchronous path's; only the launch/sync structure differs.
constexpr int DF_MAXBLK = 4096;    // partial sums per quantity (stride of the partial buffers): ceil(n / 64), n <= 262144
constexpr int DF_TILE = 512;       // coefficients staged in LDS per trip
constexpr int DF_EL = 64;          // elements per workgroup of the panel-combination kernels: lane = element, the four
                                   // wavefronts split the panel rows (row a belongs to wave a mod 4) and meet in LDS —
                                   // 4x the workgroups and a quarter of the dependent loads per thread of a
                                   // one-thread-per-element layout; these kernels are pure latency (k x n panels, ~1 MB)

__device__ __forceinline__ double df_block_sum(double v, double* red) {
    v = wave_sum64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// sum of up to DF_MAXBLK partials, every thread gets the result
__device__ __forceinline__ double df_sum_partials(const double* __restrict__ p, int nblk, double* red) {
    double v = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) v += p[b];      // (one trip up to n = 16384)
    return df_block_sum(v, red);
}

// combine the per-wave accumulators acc[NA] of element `lane` across the four wavefronts; every thread gets the sums
template <int NA>
__device__ __forceinline__ void df_rowsplit_sum(double (&acc)[NA], double (*xw)[NA][DF_EL]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NA; ++q) xw[wave][q][lane] = acc[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NA; ++q) acc[q] = (xw[0][q][lane] + xw[1][q][lane]) + (xw[2][q][lane] + xw[3][q][lane]);
}

// resid: R_j = sum_a ca_j[a] AV_a + cv_j[a] V_a (j < nneg), v = sum_a cw[a] V_a, from the RAW panels.
// coef = [ca_0 .. ca_{nneg-1} | cv_0 .. cv_{nneg-1} | cw], k doubles each.
template <int NJ>
__device__ __forceinline__ void dav_resid_vb(const VB vb, int n, int k, int nneg, const double* __restrict__ V,
                                                        const double* __restrict__ AV, int ld,
                                                        const double* __restrict__ coef, double* __restrict__ R,
                                                        double* __restrict__ vout, double* __restrict__ part,
                                                        int seek, const double* __restrict__ dscale, double theta,
                                                        double* __restrict__ mid) {
    // seek >= 0 (panels in P's eigenbasis): the rows the correction equation needs are also written divided by
    // (d - theta) — mid row 0 from residual row `seek`, mid row 1 from v — which was a kernel of its own on the chain
    __shared__ double cs[(2 * NJ + 1) * DF_TILE];
    __shared__ double xw[4][NJ + 1][DF_EL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = vb.x * DF_EL + lane;
    const int j0 = vb.y * NJ;
    const bool valid = i < n;
    const int il = valid ? i : n - 1;
    const bool dov = vb.y == 0;
    double acc[NJ + 1];
#pragma unroll
    for (int q = 0; q <= NJ; ++q) acc[q] = 0.0;
    for (int a0 = 0; a0 < k; a0 += DF_TILE) {
        const int jt = (k - a0 < DF_TILE) ? (k - a0) : DF_TILE;
        __syncthreads();
        for (int t = threadIdx.x; t < (2 * NJ + 1) * jt; t += 256) {
            const int q = t / jt, a = t - q * jt;              // q: 0..NJ-1 ca, NJ..2NJ-1 cv, 2NJ cw
            double cval = 0.0;
            if (q < NJ) { if (j0 + q < nneg) cval = coef[(size_t)(j0 + q) * k + a0 + a]; }
            else if (q < 2 * NJ) { if (j0 + q - NJ < nneg) cval = coef[(size_t)(nneg + j0 + q - NJ) * k + a0 + a]; }
            else cval = coef[(size_t)2 * nneg * k + a0 + a];
            cs[q * DF_TILE + a] = cval;
        }
        __syncthreads();
#pragma unroll 4
        for (int a = wave; a < jt; a += 4) {
            const double pv = V[(size_t)(a0 + a) * ld + il], pa = AV[(size_t)(a0 + a) * ld + il];
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) acc[jj] += cs[jj * DF_TILE + a] * pa + cs[(NJ + jj) * DF_TILE + a] * pv;
            acc[NJ] += cs[2 * NJ * DF_TILE + a] * pv;
        }
    }
    df_rowsplit_sum<NJ + 1>(acc, xw);
    if (wave == 0) {
        double den = 1.0;
        if (seek >= 0) {
            den = dscale[il] - theta;                           // exact-hit guard of GemvEpi mode 1 (kernels.hip)
            if (den == 0.0) den = 2.220446049250313e-16 * fmax(fabs(theta), 2.2250738585072014e-308);
        }
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            if (j0 + jj < nneg) {                               // (uniform)
                if (valid) R[(size_t)(j0 + jj) * ld + i] = acc[jj];
                if (valid && j0 + jj == seek) mid[i] = acc[jj] / den;
                const double ss = wave_sum64(valid ? acc[jj] * acc[jj] : 0.0);
                if (lane == 0) part[(size_t)(j0 + jj) * DF_MAXBLK + vb.x] = ss;
            }
        }
        if (dov && valid) {
            vout[i] = acc[NJ];
            if (seek >= 0) mid[(size_t)ld + i] = acc[NJ] / den;
        }
    }
}
template <int NJ>
__global__ __launch_bounds__(256) void dav_resid_kernel(int n, int k, int nneg, const double* __restrict__ V,
                                                        const double* __restrict__ AV, int ld,
                                                        const double* __restrict__ coef, double* __restrict__ R,
                                                        double* __restrict__ vout, double* __restrict__ part,
                                                        int seek, const double* __restrict__ dscale, double theta,
                                                        double* __restrict__ mid) { dav_resid_vb<NJ>(vb_hw(), n, k, nneg, V, AV, ld, coef, R, vout, part, seek, dscale, theta, mid); }

// gs1: the correction vector and its first Gram-Schmidt sweep.
//   mode 0: t = x (lanczos: x = r; gd: x = (P - theta)^-1 r)
//   mode 1: t = y (v.x / v.y) - x with v.x = cw.dx, v.y = cw.dy; |v.y| < 1e-12 -> t = x   (eigensolvers.py:123-139)
//   t1 = t - V^T (V t)  with V t = the same combination of dx = V x, dy = V y.
// Partials: part[blk] = |t|^2, part[DF_MAXBLK + blk] = |t1|^2.  Workgroup 0 also finishes the residual norms.
__device__ __forceinline__ void dav_gs1_vb(const VB vb, int n, int k, const double* __restrict__ V, int ld,
                                                      const double* __restrict__ x, const double* __restrict__ y,
                                                      const double* __restrict__ dx, const double* __restrict__ dy,
                                                      const double* __restrict__ cw, int mode,
                                                      double* __restrict__ t1, double* __restrict__ part,
                                                      const double* __restrict__ rpart, int nneg, int nblk,
                                                      double* __restrict__ scal) {
    __shared__ double cs[DF_TILE];
    __shared__ double xw[4][1][DF_EL];
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = vb.x * DF_EL + lane;
    const bool valid = i < n;
    const int il = valid ? i : n - 1;
    double sfac = 0.0;
    bool alt = true;
    double vy = 0.0;
    if (mode == 1) {
        double px = 0.0, py = 0.0;
        for (int a = threadIdx.x; a < k; a += 256) {
            const double w = cw[a];
            px += w * dx[a];
            py += w * dy[a];
        }
        const double vx = df_block_sum(px, red);
        vy = df_block_sum(py, red);
        alt = fabs(vy) < 1e-12;
        sfac = alt ? 0.0 : vx / vy;
    }
    const double xv = x[il];
    const double tv = alt ? xv : (y[il] * sfac - xv);
    double acc[1] = {0.0};
    for (int a0 = 0; a0 < k; a0 += DF_TILE) {
        const int jt = (k - a0 < DF_TILE) ? (k - a0) : DF_TILE;
        __syncthreads();
        for (int a = threadIdx.x; a < jt; a += 256)
            cs[a] = alt ? dx[a0 + a] : (dy[a0 + a] * sfac - dx[a0 + a]);
        __syncthreads();
#pragma unroll 4
        for (int a = wave; a < jt; a += 4) acc[0] += cs[a] * V[(size_t)(a0 + a) * ld + il];
    }
    df_rowsplit_sum<1>(acc, xw);
    const double t1v = tv - acc[0];
    if (wave == 0) {
        if (valid) t1[i] = t1v;
        const double s0 = wave_sum64(valid ? tv * tv : 0.0);
        const double s1 = wave_sum64(valid ? t1v * t1v : 0.0);
        if (lane == 0) {
            part[vb.x] = s0;
            part[DF_MAXBLK + vb.x] = s1;
        }
    }
    if (vb.x == 0) {
        for (int j = 0; j < nneg; ++j) {
            const double rr = df_sum_partials(rpart + (size_t)j * DF_MAXBLK, nblk, red);
            if (threadIdx.x == 0) scal[8 + j] = rr;
        }
        if (threadIdx.x == 0) {
            scal[4] = sfac;
            scal[5] = vy;
        }
    }
}
__global__ __launch_bounds__(256) void dav_gs1_kernel(int n, int k, const double* __restrict__ V, int ld,
                                                      const double* __restrict__ x, const double* __restrict__ y,
                                                      const double* __restrict__ dx, const double* __restrict__ dy,
                                                      const double* __restrict__ cw, int mode,
                                                      double* __restrict__ t1, double* __restrict__ part,
                                                      const double* __restrict__ rpart, int nneg, int nblk,
                                                      double* __restrict__ scal) { dav_gs1_vb(vb_hw(), n, k, V, ld, x, y, dx, dy, cw, mode, t1, part, rpart, nneg, nblk, scal); }

// gs2: second sweep on t1/|t1| with c2 = V t1, and the image by linearity:
//   t2 = (t1 - V^T c2) / |t1|,   A t2 = (A t1 - AV^T c2) / |t1|;   part2[blk] = |t2|^2
__device__ __forceinline__ void dav_gs2_vb(const VB vb, int n, int k, const double* __restrict__ V,
                                                      const double* __restrict__ AV, int ld,
                                                      const double* __restrict__ t1, const double* __restrict__ At1,
                                                      const double* __restrict__ c2, const double* __restrict__ part1,
                                                      int nblk, double* __restrict__ t2, double* __restrict__ At2,
                                                      double* __restrict__ part2) {
    __shared__ double cs[DF_TILE];
    __shared__ double xw[4][2][DF_EL];
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = vb.x * DF_EL + lane;
    const bool valid = i < n;
    const int il = valid ? i : n - 1;
    const double t1v = t1[il], a1v = At1[il];
    const double n1sq = df_sum_partials(part1 + DF_MAXBLK, nblk, red);
    const double inv = 1.0 / sqrt(n1sq);
    double acc[2] = {0.0, 0.0};
    for (int a0 = 0; a0 < k; a0 += DF_TILE) {
        const int jt = (k - a0 < DF_TILE) ? (k - a0) : DF_TILE;
        __syncthreads();
        for (int a = threadIdx.x; a < jt; a += 256) cs[a] = c2[a0 + a];
        __syncthreads();
#pragma unroll 4
        for (int a = wave; a < jt; a += 4) {
            acc[0] += cs[a] * V[(size_t)(a0 + a) * ld + il];
            acc[1] += cs[a] * AV[(size_t)(a0 + a) * ld + il];
        }
    }
    df_rowsplit_sum<2>(acc, xw);
    const double t2v = (t1v - acc[0]) * inv, a2v = (a1v - acc[1]) * inv;
    if (wave == 0) {
        if (valid) {
            t2[i] = t2v;
            At2[i] = a2v;
        }
        const double s2 = wave_sum64(valid ? t2v * t2v : 0.0);
        if (lane == 0) part2[vb.x] = s2;
    }
}
__global__ __launch_bounds__(256) void dav_gs2_kernel(int n, int k, const double* __restrict__ V,
                                                      const double* __restrict__ AV, int ld,
                                                      const double* __restrict__ t1, const double* __restrict__ At1,
                                                      const double* __restrict__ c2, const double* __restrict__ part1,
                                                      int nblk, double* __restrict__ t2, double* __restrict__ At2,
                                                      double* __restrict__ part2) { dav_gs2_vb(vb_hw(), n, k, V, AV, ld, t1, At1, c2, part1, nblk, t2, At2, part2); }

// final: normalise the new vector and its image into panel slot k and form the raw Gram dots in the same launch.
//   blocks [0, nblke):       V_k = t2 / |t2|, AV_k = At2 / |t2|;  block 0 also stores the three squared norms
//   block nblke + r, r < k:  out[r] = V_r . V_k,  out[cap + r] = V_r . AV_k
//   block nblke + k + r:     out[2 cap + r] = AV_r . V_k
//   block nblke + 2k:        out[cap + k] = V_k . AV_k,  out[k] = V_k . V_k
__device__ __forceinline__ void dav_final_vb(const VB vb, int n, int k, int cap, const double* __restrict__ V,
                                                        const double* __restrict__ AV, int ld,
                                                        const double* __restrict__ t2, const double* __restrict__ At2,
                                                        const double* __restrict__ part1, const double* __restrict__ part2,
                                                        int nblk, int nblke, double* __restrict__ vslot,
                                                        double* __restrict__ avslot, double* __restrict__ out,
                                                        unsigned long long* pword, unsigned long long pseq, unsigned* pcount) {
    __shared__ double red[4];
    // Polled wait (context.hip, poll_wait): the LAST workgroup to finish stores the sequence word into pinned host memory —
    // every workgroup makes its own stores visible system-wide, then counts itself; who counts last resets the counter
    // and publishes.  (No one-thread kernel behind this one: its launch boundary was 3 us of every iteration.)
    auto publish = [&]() {
        if (pword == nullptr) return;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence_system();
            if (atomicAdd(pcount, 1u) == vb.gx - 1) {
                *pcount = 0u;
                __threadfence_system();
                *reinterpret_cast<volatile unsigned long long*>(pword) = pseq;
            }
        }
    };
    const double n2sq = df_sum_partials(part2, nblk, red);
    const double inv2 = 1.0 / sqrt(n2sq);
    const int b = vb.x;
    if (b < nblke) {
        const int i = b * 256 + threadIdx.x;
        if (i < n) {
            vslot[i] = t2[i] * inv2;
            avslot[i] = At2[i] * inv2;
        }
        if (b == 0) {
            const double n0sq = df_sum_partials(part1, nblk, red);
            const double n1sq = df_sum_partials(part1 + DF_MAXBLK, nblk, red);
            if (threadIdx.x == 0) {
                out[3 * (size_t)cap] = n0sq;
                out[3 * (size_t)cap + 1] = n1sq;
                out[3 * (size_t)cap + 2] = n2sq;
            }
        }
        publish();
        return;
    }
    const int r = b - nblke;
    const double* row = (r < k) ? V + (size_t)r * ld : (r < 2 * k) ? AV + (size_t)(r - k) * ld : t2;
    double d1 = 0.0, d2 = 0.0;
    const bool two = r < k;                       // V rows need both right-hand sides
    const double* rhs1 = (r == 2 * k) ? At2 : t2;
    for (int i0 = threadIdx.x; i0 < n; i0 += 1024) {
        double a[4], p[4], q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 256 * u, ic = i < n ? i : n - 1;
            a[u] = row[ic];
            p[u] = rhs1[ic];
            q[u] = two ? At2[ic] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + 256 * u < n) {
                d1 += a[u] * p[u];
                d2 += a[u] * q[u];
            }
    }
    d1 = df_block_sum(d1, red);
    if (two) d2 = df_block_sum(d2, red);
    if (threadIdx.x == 0) {
        if (r < k) {
            out[r] = d1 * inv2;
            out[(size_t)cap + r] = d2 * inv2;
        } else if (r < 2 * k) {
            out[2 * (size_t)cap + r - k] = d1 * inv2;
        } else {
            out[(size_t)cap + k] = d1 * inv2 * inv2;
            out[k] = n2sq * inv2 * inv2;
        }
    }
    publish();
}
__global__ __launch_bounds__(256) void dav_final_kernel(int n, int k, int cap, const double* __restrict__ V,
                                                        const double* __restrict__ AV, int ld,
                                                        const double* __restrict__ t2, const double* __restrict__ At2,
                                                        const double* __restrict__ part1, const double* __restrict__ part2,
                                                        int nblk, int nblke, double* __restrict__ vslot,
                                                        double* __restrict__ avslot, double* __restrict__ out,
                                                        unsigned long long* pword, unsigned long long pseq, unsigned* pcount) { dav_final_vb(vb_hw(), n, k, cap, V, AV, ld, t2, At2, part1, part2, nblk, nblke, vslot, avslot, out, pword, pseq, pcount); }

// Vector-major panels (k rows of stride ld) -> the caller's layout (n x k row-major, vectors as columns), both panels in
// one launch (upper half of grid.y: the second panel), through 32 x 32 LDS tiles: the results then travel in ONE
// contiguous transfer and land in the caller's arrays by plain copies (the host-side transposition of two n x k panels was
// a third of the call's fixed cost).
__device__ __forceinline__ void dav_to_columns_vb(const VB vb, int n, int k, const double* __restrict__ P0,
                                                  const double* __restrict__ P1, int ld, double* __restrict__ out) {
    __shared__ double tile[32][33];
    const unsigned ny = vb.gy / 2, second = vb.y >= ny ? 1u : 0u;
    const double* __restrict__ P = second ? P1 : P0;
    double* __restrict__ o = out + (size_t)second * n * k;
    const int i0 = vb.x * 32, j0 = (vb.y - second * ny) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;              // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int j = j0 + r, i = i0 + tx;
        tile[r][tx] = (j < k && i < n) ? P[(size_t)j * ld + i] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r, j = j0 + tx;
        if (i < n && j < k) o[(size_t)i * k + j] = tile[tx][r];
    }
}
__global__ __launch_bounds__(256) void dav_to_columns_kernel(int n, int k, const double* __restrict__ P0,
                                                             const double* __restrict__ P1, int ld, double* __restrict__ out) { dav_to_columns_vb(vb_hw(), n, k, P0, P1, ld, out); }

// The rotation into the Ritz basis and the change of layout in ONE launch for k <= KM (the usual case: tens of vectors):
//   out[(z n + i) k + j] = sum_a W[a k + j] P_z[a][i],   z = 0: V, 1: AV,
// a thread owns column i of one panel and all k outputs (accumulated over a in ascending order, like launch_lincomb), the
// k x k coefficients come from LDS.  Replaces two lincomb launches and the transposition (three launches, ~15 us).
template <int KM>
__device__ __forceinline__ void dav_rotate_columns_vb(const VB vb, int n, int k, const double* __restrict__ W,
                                                      const double* __restrict__ P0, const double* __restrict__ P1, int ld,
                                                      double* __restrict__ out) {
    __shared__ double ws[KM * KM];
    const double* __restrict__ P = vb.y ? P1 : P0;
    for (int t = threadIdx.x; t < k * k; t += 256) ws[(t / k) * KM + (t % k)] = W[t];
    __syncthreads();
    const int i = vb.x * 256 + threadIdx.x;
    if (i >= n) return;
    double acc[KM];
#pragma unroll
    for (int j = 0; j < KM; ++j) acc[j] = 0.0;
    for (int a = 0; a < k; ++a) {
        const double p = P[(size_t)a * ld + i];
#pragma unroll
        for (int j = 0; j < KM; ++j) acc[j] += (j < k ? ws[a * KM + j] : 0.0) * p;
    }
    double* o = out + ((size_t)vb.y * n + i) * k;
#pragma unroll
    for (int j = 0; j < KM; ++j)
        if (j < k) o[j] = acc[j];
}
template <int KM>
__global__ __launch_bounds__(256) void dav_rotate_columns_kernel(int n, int k, const double* __restrict__ W,
                                                                 const double* __restrict__ P0, const double* __restrict__ P1, int ld,
                                                                 double* __restrict__ out) { dav_rotate_columns_vb<KM>(vb_hw(), n, k, W, P0, P1, ld, out); }

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
    if (n > 64 * DF_MAXBLK) {
        set_error("davidson: n = %d exceeds the partial-sum layout (%d); use sella_davidson_block", n, 64 * DF_MAXBLK);
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
        // Pvecs (n x r) / PvecsT (r x n): r == n is a full eigendecomposition; r < n a structured one whose remaining
        // n - r eigenvalues all equal pscale, with the orthogonal complement of the r vectors as their eigenspace
        const int pr = s.Q->cols;
        if (s.Q->rows != n || pr < 1 || pr > n || s.Qt->rows != pr || s.Qt->cols != n) {
            set_error("davidson: eigenvector matrices of P must be %d x r and r x %d", n, n);
            return SELLA_E_INVALID;
        }
        s.prank = pr;
        double* dev;
        SCHK(scratch_get(c, SCR_C, (size_t)s.ld * sizeof(double), &dev));
        SCHK(h2d_async(c, dev, pevals, (size_t)pr * sizeof(double)));
        s.pevals_dev = dev;
    } else {
        s.prank = n;
    }
    // Fused iteration with an eigenbasis preconditioner: the images Q^T V, Q^T AV of the panels are carried (one
    // 2-right-hand-side pass per NEW vector, queued behind the iteration's last kernel so that it runs while the host
    // does its k x k algebra), and Q^T r, Q^T v of the correction are linear combinations of their rows — the
    // Rayleigh-Ritz-dependent chain then holds two n x n passes (Q, A) instead of three.
    s.qt_mode = s.A != nullptr && s.Q != nullptr && s.prank == n && method >= SELLA_DAV_GD && method <= SELLA_DAV_JD0_ALT &&
                vref == nullptr && !getenv("SELLA_DAV_SYNC") && !getenv("SELLA_DAV_NOQT");
    // Polled waits (dav_poll) read the iteration's scalars where its kernels stored them — in the pinned mirror: the fused
    // iteration then runs with host-visible scalars for the length of this call (restored on every way out).
    struct HostScalarsScope {
        sella_ctx* c; long saved;
        ~HostScalarsScope() { c->opt.host_scalars = saved; }
    } hs_scope{c, c->opt.host_scalars};
    if (s.qt_mode && c->opt.dav_poll) c->opt.host_scalars = 1;
    if (maxiter <= 0) maxiter = 2 * n + 1;
    const int kstop = (n < maxiter) ? n : maxiter;

    const double t_enter = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    int cap0 = 64;
    while (cap0 < nv0 + 2) cap0 *= 2;
    if (cap0 > n + 1) cap0 = n + 1;
    if (cap0 < nv0 + 1) cap0 = nv0 + 1;
    int st = dav_alloc(s, cap0);
    auto fail = [&](int code) { dav_free(s); return code; };
#define DCHK(expr) do { int s__ = (expr); if (s__ != SELLA_OK) return fail(s__); } while (0)
#define DHIP(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); return fail(SELLA_E_HIP); } } while (0)
    if (st != SELLA_OK) return fail(st);
    DCHK(scratch_get(c, SCR_T, (size_t)40 * s.ld * sizeof(double), &s.wk));

    vec Wc;            // cumulative rotation raw -> Ritz basis (k x k)
    // ---- start block: V = mgs(v0) (eigensolvers.py:44-50), AV = A V ----------------------
    {
        double* tmp = nullptr;
        if (nv0 > 1) {
            DCHK(scratch_get(c, SCR_X, (size_t)nv0 * s.ld * sizeof(double), &tmp));
            DCHK(upload_panel(c, v0, n, nv0, tmp, s.ld));
        }
        for (int j = 0; j < nv0; ++j) {
            double* slot = s.Vp + (size_t)s.k * s.ld;
            if (nv0 == 1) DCHK(h2d_async(c, slot, v0, (size_t)n * sizeof(double)));          // (one start vector: straight into its slot)
            else DHIP(s_memcpy(c, slot, tmp + (size_t)j * s.ld, (size_t)s.ld * sizeof(double), hipMemcpyDeviceToDevice));
            int kept = 0;
            if (nv0 == 1 && c->opt.host_scalars && !(c->opt.gs_small && n <= c->opt.gs_small) && !(c->cohort && cohort_in_fiber())) {
                // one start vector, scalars in the pinned mirror: its normalisation and its product A v are queued together
                // and waited for ONCE (the norms of the sweeps are read behind append_vector's wait) — the decision below is
                // gs_orthonormalise's for an empty basis; a vector that fails it leaves through the error exit as before
                DCHK(gs_project_twice(c, s.Vp, s.ld, 0, slot, n));
                DCHK(append_vector(s, Wc));
                const double n0sq = c->hscal[8], n1 = sqrt(c->hscal[9]), n2 = sqrt(c->hscal[10]);
                kept = (n0sq > 0.0) && (n1 == n1) && (n1 >= 1e-6) && (n2 == n2) && (n2 >= 1e-6) && (fabs(1.0 - n2) <= 1e-15);
                if (!kept) {
                    // (rare: a third sweep is needed, or the vector is numerically zero) back to the step-by-step path
                    s.k = 0;
                    s.nmatvec = 0;
                    Wc.clear();
                    DCHK(h2d_async(c, slot, v0, (size_t)n * sizeof(double)));
                    DCHK(orthonormalise(s, slot, s.k, &kept, nullptr));
                    if (kept) DCHK(append_vector(s, Wc));
                }
                continue;
            }
            DCHK(orthonormalise(s, slot, s.k, &kept, nullptr));
            if (kept) DCHK(append_vector(s, Wc));
        }
        if (s.k == 0) {
            set_error("davidson: the start block is numerically zero");
            return fail(SELLA_E_INVALID);
        }
    }

    vec lams, W, gvv, gva, X, At, tmpm, coef, Wn;
    unsigned long long lcg = 0x9E3779B97F4A7C15ull;   // deterministic stand-in for np.random.normal (:107)
    // the fused one-synchronisation iteration needs a resident operator and a correction that is a fixed chain of
    // streams; the bordered multi-vector corrections (mjd0) and the callback operator stay on the synchronous path
    // (vref, "a hack for the optbench.org eigensolver convergence test" eigensolvers.py:73-77, compares the LOWEST Ritz
    // vector whatever pair is pursued: synchronous path only)
    const bool fast_ok = s.A != nullptr && method <= SELLA_DAV_JD0_ALT && vref == nullptr && !getenv("SELLA_DAV_SYNC");
    int last_seek = 0;
    bool wc_identity = false;   // Wc is exactly the identity (set by the device rotation of large subspaces)
    bool old_diag = false;      // rotated Gram matrices are (I, diag(lams)) apart from the newest row / column
    long n_fast = 0, n_slow = 0;

    double t_host = 0.0;
    const bool dbg_time = getenv("SELLA_DEBUG_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    if (s.qt_mode) {
        DCHK(qt_update(s, 0, s.k));
        // (the event of the un-polled wait is created when first needed: creating and destroying one per call was ~10 us)
    }
    while (true) {
        ++s.it;
        cohort_barrier(c, s.it, 0);           // members of a cohort start their iterations together (cohort.h)
        const int k = s.k, cap = s.cap;
        double th0 = now();
        // ---- Rayleigh-Ritz (eigensolvers.py:57-64) ---------------------------------------
        // Once the basis has been rotated into Ritz vectors, V^T V = I and V^T AV = diag(theta) for the old block
        // (to roundoff, for a symmetric operator), and the new vector only adds a border: the (k x k) problem is
        // an ARROWHEAD eigenproblem, solved in O(k^2) (host_math.h arrow_eig) instead of the dense O(k^3) path —
        // symmetrize_Y2 with an orthonormal S just mirrors the lower triangle of S^T Y (hessian_update.py:12-24
        // with S^T S = I: coef = rhs), and scipy's eigh reads that lower triangle (eigensolvers.py:58), so the
        // border is the row AV_a . t.  The dense path remains for an unsymmetric operator (the finite-difference
        // Hessian), for the start block, and whenever the measured asymmetry / non-orthogonality is not roundoff.
        bool arrow = false;
        if (old_diag && k >= 2) {
            const int m = k - 1;
            double dmax = 0.0, asym = 0.0, scale = fabs(s.Gva[(size_t)m * cap + m]), dd = 0.0, db = 0.0, dld = 0.0;
            for (int a = 0; a < m; ++a) {
                const double da = s.Gvv[(size_t)a * cap + m], ba = s.Gva[(size_t)m * cap + a];
                dmax = std::max(dmax, fabs(da));
                asym = std::max(asym, fabs(ba - s.Gva[(size_t)a * cap + m]));
                scale = std::max(scale, std::max(fabs(lams[a]), fabs(ba)));
                dd += da * da;
                db += da * ba;
                dld += lams[a] * da * da;
            }
            const double tt = s.Gvv[(size_t)m * cap + m];
            if (asym <= 4e-12 * scale && dmax <= 1e-10 && fabs(tt - 1.0) <= 1e-10) {
                const double ell = sqrt(tt - dd);
                vec z(m), lprev(lams.begin(), lams.begin() + m), Yv((size_t)k * k);
                for (int a = 0; a < m; ++a)
                    z[a] = (s.Gva[(size_t)m * cap + a] - lprev[a] * s.Gvv[(size_t)a * cap + m]) / ell;
                const double alpha_c = (s.Gva[(size_t)m * cap + m] - 2.0 * db + dld) / (ell * ell);
                lams.assign(k, 0.0);
                if (hostm::arrow_eig(m, lprev.data(), z.data(), alpha_c, lams.data(), Yv.data()) == 0) {
                    W.assign((size_t)k * k, 0.0);
                    for (int j = 0; j < k; ++j) {
                        const double ym = Yv[(size_t)m * k + j] / ell;
                        for (int a = 0; a < m; ++a) W[(size_t)a * k + j] = Yv[(size_t)a * k + j] - s.Gvv[(size_t)a * cap + m] * ym;
                        W[(size_t)m * k + j] = ym;
                    }
                    arrow = true;
                }
            }
        }
        if (!arrow) {
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
        }
        int nneg = 0;
        for (int i = 0; i < k; ++i) nneg += (lams[i] < 0.0);
        if (nneg < 1) nneg = 1;
        // rotate into the Ritz basis — lazily: Wc <- Wc W on the host, Gram matrices by congruence; the device
        // panels stay raw (eigensolvers.py:62-64 rotates V and AV themselves every iteration: O(n k^2) traffic
        // and two launches that nothing downstream needs before the final result)
        {
            if (wc_identity) {
                Wc = W;                                   // panels were just rotated on the device: Wc = I
            } else {
                Wn.assign((size_t)k * k, 0.0);
                for (int a = 0; a < k; ++a) {
                    double* wn = Wn.data() + (size_t)a * k;
                    for (int l = 0; l < k; ++l) {
                        const double w = Wc[(size_t)a * k + l];
                        if (w == 0.0) continue;
                        const double* wl = W.data() + (size_t)l * k;
                        for (int b = 0; b < k; ++b) wn[b] += w * wl[b];
                    }
                }
                Wc.swap(Wn);
            }
            wc_identity = false;
            if (arrow) {
                // exact by construction: the rotated basis diagonalises the (mirrored) projected operator
                for (int a = 0; a < k; ++a)
                    for (int b = 0; b < k; ++b) {
                        s.Gvv[(size_t)a * cap + b] = (a == b) ? 1.0 : 0.0;
                        s.Gva[(size_t)a * cap + b] = (a == b) ? lams[a] : 0.0;
                    }
            } else {
                tmpm.resize((size_t)k * k);
                hostm::congruence(k, W.data(), gvv.data(), tmpm.data());
                gvv = tmpm;
                hostm::congruence(k, W.data(), gva.data(), tmpm.data());
                gva = tmpm;
                // is the rotated pair (I, diag(theta)) to roundoff?  Then snap it (what the arrowhead step assumes)
                double off = 0.0, dev = 0.0, scale = 0.0;
                for (int a = 0; a < k; ++a) {
                    scale = std::max(scale, fabs(lams[a]));
                    for (int b = 0; b < k; ++b) {
                        if (a != b) off = std::max(off, fabs(gva[(size_t)a * k + b]));
                        dev = std::max(dev, fabs(gvv[(size_t)a * k + b] - (a == b ? 1.0 : 0.0)));
                    }
                    off = std::max(off, fabs(gva[(size_t)a * k + a] - lams[a]));
                }
                old_diag = s.A != nullptr && off <= 4e-12 * std::max(scale, 1e-300) && dev <= 1e-12;
                if (old_diag) {
                    for (int a = 0; a < k; ++a)
                        for (int b = 0; b < k; ++b) {
                            gvv[(size_t)a * k + b] = (a == b) ? 1.0 : 0.0;
                            gva[(size_t)a * k + b] = (a == b) ? lams[a] : 0.0;
                        }
                }
                unpack(gvv, k, s.Gvv, cap);
                unpack(gva, k, s.Gva, cap);
            }
        }
        t_host += now() - th0;
        if (k >= kstop) break;                                            // :65-66
        if (k >= 64) s.qt_mode = false;       // (the flush below rotates the raw panels only; old chain from here on)
        if (k >= 64) {
            // large subspaces (converged runs: hundreds of vectors): carrying the cumulative rotation on the host is
            // O(k^3) per iteration; apply it to the device panels instead (two k x k x n products) and restart from
            // the identity — the panels then ARE the Ritz vectors, as in the reference (eigensolvers.py:62-64)
            double* dWp;
            DCHK(put_small(s, Wc.data(), k * k, 0, 0, &dWp));
            DCHK(launch_lincomb(c, n, k, s.Vp, s.ld, k, dWp, k, nullptr, 0, 0, nullptr, 0, 0.0, s.Vq, s.ld));
            DCHK(launch_lincomb(c, n, k, s.AVp, s.ld, k, dWp, k, nullptr, 0, 0, nullptr, 0, 0.0, s.AVq, s.ld));
            std::swap(s.Vp, s.Vq);
            std::swap(s.AVp, s.AVq);
            Wc.assign((size_t)k * k, 0.0);
            for (int a = 0; a < k; ++a) Wc[(size_t)a * k + a] = 1.0;
            wc_identity = true;                           // (gram_append keeps it the identity, one size larger)
        }
        if (nneg > 4000) { set_error("davidson: too many negative Ritz values (%d)", nneg); return fail(SELLA_E_UNSUPPORTED); }

        // ---- residual coefficients in the raw basis (:68-71) -------------------------------
        //   R_j = AV_rot_j + sum_{l<j} X[l][j] V_rot_l - lams_j V_rot_j,   (.)_rot = (.)_raw Wc
        th0 = now();
        const bool have_x = !(arrow || old_diag);               // in the (I, diag) state symmetrize_Y is the identity
        if (have_x) hostm::symm_coeffs(k, gvv.data(), gva.data(), 2, X.data());
        const int seek_pred = (last_seek < nneg) ? last_seek : 0;
        coef.assign((size_t)(2 * nneg + 1) * k, 0.0);
        for (int j = 0; j < nneg; ++j) {
            double* ca = coef.data() + (size_t)j * k;
            double* cv = coef.data() + (size_t)(nneg + j) * k;
            for (int a = 0; a < k; ++a) {
                ca[a] = Wc[(size_t)a * k + j];
                double sacc = -lams[j] * Wc[(size_t)a * k + j];
                if (have_x)
                    for (int l = 0; l < j; ++l) sacc += Wc[(size_t)a * k + l] * X[(size_t)l * k + j];
                cv[a] = sacc;
            }
        }
        auto set_cw = [&](int seek) {
            double* cw = coef.data() + (size_t)2 * nneg * k;
            for (int a = 0; a < k; ++a) cw[a] = Wc[(size_t)a * k + seek];
        };
        int vrow_of = vref ? 0 : seek_pred;              // which Ritz vector the resid kernel leaves in `vrow`
        set_cw(vrow_of);
        t_host += now() - th0;

        if (s.k + 1 > s.cap) {
            int ncap = s.cap * 2;
            if (ncap > n + 1) ncap = n + 1;
            DCHK(dav_alloc(s, ncap));
        }
        const int capn = s.cap;
        double* part;
        DCHK(scratch_get(c, SCR_PART, (size_t)(nneg + 8) * DF_MAXBLK * sizeof(double), &part));
        double* rpart = part + 4 * DF_MAXBLK;            // [0,2) gs1, [2] gs2, [3] vref, [4..) residual rows
        const int nblk = (n + DF_EL - 1) / DF_EL, nblke = (n + 255) / 256;
        double* vrow = s.wk;                             // Ritz vector of the pursued pair
        double* t1 = s.wk + 1 * (size_t)s.ld;
        double* At1 = s.wk + 2 * (size_t)s.ld;
        double* t2 = s.wk + 3 * (size_t)s.ld;
        double* At2 = s.wk + 4 * (size_t)s.ld;
        double* dxd = s.wk + 5 * (size_t)s.ld;           // V x, V y (k entries each)
        double* dyd = s.wk + 6 * (size_t)s.ld;
        double* c2d = s.wk + 7 * (size_t)s.ld;
        double* in = s.wk + 8 * (size_t)s.ld;            // up to 8 rows
        double* mid = s.wk + 16 * (size_t)s.ld;
        double* out = s.wk + 24 * (size_t)s.ld;
        double* dvref = s.wk + 39 * (size_t)s.ld;
        double* dsc = scal_out(c, DS_GRAM);              // scalars of this iteration (device, or pinned host: zero-copy)
        const size_t S0 = 3 * (size_t)capn;
        if (vref) DCHK(h2d_async(c, dvref, vref, (size_t)n * sizeof(double)));

        // coefficients to the device (pinned staging, asynchronous)
        auto launch_resid = [&]() -> int {
            double* dC;
            const int ncoef = (2 * nneg + 1) * k;
            if (c->opt.host_scalars && ncoef <= 8192) {
                // zero-copy: the kernels stage the coefficients into LDS straight from pinned host memory (one
                // coalesced read over PCIe instead of a copy launch on the chain); the slot is rewritten only after
                // this iteration's synchronisation
                dC = c->hscal + DS_STAGE + 8192;
                memcpy(dC, coef.data(), (size_t)ncoef * sizeof(double));
            } else {
                SCHK(put_small(s, coef.data(), ncoef, 1, (size_t)s.cap * s.cap + 8, &dC));
            }
            s.dcoef = dC;
            dim3 grid(nblk, (nneg + 3) / 4);             // nblk workgroups of 64 elements
            if (nneg == 1) {
                grid.y = 1;
                SELLA_LAUNCHB(c, HIP_KERNEL_NAME(dav_resid_kernel<1>), SELLA_BODY(dav_resid_vb<1>), 256, grid, dim3(256), 0, n, k, nneg, s.Vp, s.AVp,
                                   s.ld, dC, s.Rp, vrow, rpart, -1, nullptr, 0.0, nullptr);
            } else {
                SELLA_LAUNCHB(c, HIP_KERNEL_NAME(dav_resid_kernel<4>), SELLA_BODY(dav_resid_vb<4>), 256, grid, dim3(256), 0, n, k, nneg, s.Vp, s.AVp,
                                   s.ld, dC, s.Rp, vrow, rpart, -1, nullptr, 0.0, nullptr);
            }
            HIPCHK(hipGetLastError());
            return SELLA_OK;
        };
        // x (and y) of the correction for the pair `seek` into out rows 0 (and 1); returns the gs1 mode
        auto launch_expand = [&](int seek, double theta, int* mode) -> int {
            const double* r = s.Rp + (size_t)seek * s.ld;
            if (method == SELLA_DAV_LANCZOS) {
                HIPCHK(s_memcpy(c, out, r, (size_t)s.ld * sizeof(double), hipMemcpyDeviceToDevice));
                *mode = 0;
            } else if (method == SELLA_DAV_GD) {
                const double* rr[1] = {r};
                SCHK(apply_pinv_xp(s, theta, rr, 1, mid, out));
                *mode = 0;
            } else {
                const double* rv[2] = {r, vrow};
                SCHK(apply_pinv_xp(s, theta, rv, 2, mid, out));
                *mode = 1;
            }
            return SELLA_OK;
        };

        bool appended = false, stop = false;
        int seeking = -1;
        if (fast_ok) {
            // ---- speculative chain for the predicted pair ---------------------------------------------------
            int mode = 0;
            if (s.qt_mode) {
                // Q^T r_j, Q^T v from the carried eigenbasis panels (same kernel, same coefficients; |Q^T r| = |r| for
                // the convergence test), the diagonal of (P - theta)^-1, then the ONE n x n pass with Q
                double* dC;
                const int ncoef = (2 * nneg + 1) * k;
                if (c->opt.dav_zero_copy && ncoef <= 8192) {
                    // the kernels stage the coefficients into LDS straight from pinned host memory: one coalesced read
                    // over the link instead of a copy launch on the chain; the slot is rewritten only after this
                    // iteration's synchronisation
                    dC = c->hscal + DS_STAGE + 8192;
                    memcpy(dC, coef.data(), (size_t)ncoef * sizeof(double));
                } else {
                    DCHK(put_small(s, coef.data(), ncoef, 1, (size_t)s.cap * s.cap + 8, &dC));
                }
                s.dcoef = dC;
                double* Rq = s.Vq;                           // free between flushes: rows Q^T r_j
                double* vq = s.AVq;                          // row 0: Q^T v
                dim3 grid(nblk, (nneg + 3) / 4);
                const bool fuse = c->opt.dav_fuse_scale;     // the diagonal scaling in the epilogue of the residual kernel
                const int fseek = fuse ? seek_pred : -1;
                if (nneg == 1) {
                    grid.y = 1;
                    SELLA_LAUNCHB(c, HIP_KERNEL_NAME(dav_resid_kernel<1>), SELLA_BODY(dav_resid_vb<1>), 256, grid, dim3(256), 0, n, k, nneg, s.QtV,
                                       s.QtAV, s.ld, dC, Rq, vq, rpart, fseek, s.pevals_dev, lams[seek_pred], mid);
                } else {
                    SELLA_LAUNCHB(c, HIP_KERNEL_NAME(dav_resid_kernel<4>), SELLA_BODY(dav_resid_vb<4>), 256, grid, dim3(256), 0, n, k, nneg, s.QtV,
                                       s.QtAV, s.ld, dC, Rq, vq, rpart, fseek, s.pevals_dev, lams[seek_pred], mid);
                }
                mode = (method == SELLA_DAV_GD) ? 0 : 1;
                if (!fuse)
                    SELLA_LAUNCHB(c, dav_eigscale_kernel, dav_eigscale_vb, 256, dim3(nblke), dim3(256), 0, n, mode == 1 ? 2 : 1,
                                       Rq + (size_t)seek_pred * s.ld, vq, s.pevals_dev, lams[seek_pred], mid, s.ld);
                DHIP(hipGetLastError());
                DCHK(launch_gemv_rows(c, s.Q->d, n, n, s.Q->ld, mid, s.ld, mode == 1 ? 2 : 1, out, s.ld, GemvEpi()));
            } else {
            DCHK(launch_resid());
            DCHK(launch_expand(seek_pred, lams[seek_pred], &mode));
            }
            {
                const double* xs[2] = {out, out + s.ld};           // V.[x y] in one launch: dyd = dxd + ld
                DCHK(launch_gemv_rows_xp(c, s.Vp, k, n, s.ld, xs, mode == 1 ? 2 : 1, dxd, s.ld, GemvEpi()));
            }
            SELLA_LAUNCHB(c, dav_gs1_kernel, dav_gs1_vb, 256, dim3(nblk), dim3(256), 0, n, k, s.Vp, s.ld, out, out + s.ld, dxd,
                               dyd, s.dcoef + (size_t)2 * nneg * k, mode, t1, part, rpart, nneg, nblk, dsc + S0);
            DHIP(hipGetLastError());
            s.nmatvec++;
            DCHK(launch_gemv_rows2(c, s.A->d, n, s.A->ld, s.Vp, k, s.ld, n, t1, At1, c2d));     // [A; V] t1 in one launch
            SELLA_LAUNCHB(c, dav_gs2_kernel, dav_gs2_vb, 256, dim3(nblk), dim3(256), 0, n, k, s.Vp, s.AVp, s.ld, t1, At1, c2d, part,
                               nblk, t2, At2, part + 2 * DF_MAXBLK);
            // (polled wait: the kernel's last workgroup publishes the sequence word itself)
            const bool poll = s.qt_mode && c->opt.dav_poll && c->opt.host_scalars;
            unsigned long long* pword = nullptr;
            unsigned long long pseq = 0;
            unsigned* pcount = nullptr;
            if (poll) DCHK(poll_arm(c, &pword, &pseq, &pcount));
            SELLA_LAUNCHB(c, dav_final_kernel, dav_final_vb, 256, dim3(nblke + 2 * k + 1), dim3(256), 0, n, k, capn, s.Vp, s.AVp, s.ld,
                               t2, At2, part, part + 2 * DF_MAXBLK, nblk, nblke, s.Vp + (size_t)k * s.ld,
                               s.AVp + (size_t)k * s.ld, dsc, pword, pseq, pcount);
            DHIP(hipGetLastError());
            if (s.qt_mode) {
                // read the scalars back (host_scalars: the kernels have written them into pinned host memory themselves
                // and there is nothing to copy), mark that point, queue the eigenbasis images of the new vector behind it
                // (they run while the host decides and does the next Rayleigh-Ritz step), wait for the mark only
                const int cnt = (int)(S0 + 8 + nneg);
                if (!c->opt.host_scalars)
                    DHIP(s_memcpy(c, c->hscal + DS_GRAM, c->dscal + DS_GRAM, (size_t)cnt * sizeof(double), hipMemcpyDeviceToHost, true));
                if (!poll) {
                    if (!s.ev) DHIP(hipEventCreate(&s.ev));
                    DHIP(hipEventRecord(s.ev, c->stream));
                }
                const double* xs[2] = {s.Vp + (size_t)k * s.ld, s.AVp + (size_t)k * s.ld};
                DCHK(launch_gemv_rows_xp(c, s.Qt->d, n, n, s.Qt->ld, xs, 2, s.QtV + (size_t)k * s.ld, capn * s.ld, GemvEpi()));
                if (poll) DCHK(poll_wait(c));
                else DCHK(event_wait(c, s.ev));
            } else {
                DCHK(sync_scalars(c, DS_GRAM, (int)(S0 + 8 + nneg)));
            }
            const double* h = c->hscal + DS_GRAM;
            // ---- verdict: the reference's control flow evaluated after the fact --------------------------------
            for (int i = 0; i < nneg; ++i) {                                   // :80-89
                const double rnorm = sqrt(h[S0 + 8 + i]);
                if (k == 1 || rnorm >= gamma * fabs(lams[i])) { seeking = i; break; }
            }
            if (seeking < 0) { --s.nmatvec; break; }                            // converged: discard the speculative vector
            const double n0sq = h[S0], n1 = sqrt(h[S0 + 1] / h[S0]), n2 = sqrt(h[S0 + 2]);
            const bool gs_ok = (n0sq > 0.0) && (n1 == n1) && (n1 >= 1e-2) && (n2 == n2) && (fabs(1.0 - n2) <= 1e-15);
            if (seeking == seek_pred && gs_ok) {
                gram_append(s, Wc, h, h + capn, h + 2 * capn, h[k], h[capn + k]);
                appended = true;
                ++n_fast;
            } else {
                --s.nmatvec;                                                   // the speculative A t is discarded
            }
        }
        if (!appended) {
            // ---- synchronous path: the reference's steps one by one -------------------------------------------
            ++n_slow;
            if (seeking >= 0 && s.qt_mode) DCHK(launch_resid());      // the fast attempt left Q^T r only: r and v themselves
            if (seeking < 0) {
                DCHK(launch_resid());
                DCHK(launch_rows_sumsq(c, s.Rp, s.ld, nneg, n, scal_out(c, 0)));
                int nread = nneg;
                if (vref) {
                    DCHK(launch_gemv_rows(c, vrow, 1, n, s.ld, dvref, s.ld, 1, scal_out(c, nneg), 1, GemvEpi()));
                    nread += 1;
                }
                DCHK(sync_scalars(c, 0, nread));
                if (vref && fabs(c->hscal[nneg]) > vreftol) break;                // :74-77
                for (int i = 0; i < nneg; ++i) {                                  // :80-89
                    const double rnorm = sqrt(c->hscal[i]);
                    if (k == 1 || rnorm >= gamma * fabs(lams[i])) { seeking = i; break; }
                }
                if (seeking < 0) break;
            }
            const double theta = lams[seeking];
            const double* r = s.Rp + (size_t)seeking * s.ld;
            if (seeking != vrow_of) {
                // Ritz vector of the pair actually pursued
                set_cw(seeking);
                vrow_of = seeking;
                double* dW;
                DCHK(put_small(s, coef.data() + (size_t)2 * nneg * k, k, 2, 2 * (size_t)s.cap * s.cap + 16, &dW));
                DCHK(launch_lincomb(c, n, 1, s.Vp, s.ld, k, dW, 1, nullptr, 0, 0, nullptr, 0, 0.0, vrow, s.ld));
            }
            const double* v = vrow;
            // ---- correction vector (expand, :115-153) into panel slot k -----------------------
            double* t = s.Vp + (size_t)s.k * s.ld;
            auto copy_row = [&](double* dst, const double* src) -> int {
                HIPCHK(s_memcpy(c, dst, src, (size_t)s.ld * sizeof(double), hipMemcpyDeviceToDevice));
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
                // mjd0 / mjd0_alt: z = Pinv(V alpha - r), (V^T Pinv V) alpha = V^T Pinv r   (:140-151); the raw
                // panel spans the same space as the Ritz vectors, and z is invariant under the change of basis
                const int kk = s.k;
                double *pv, *pmid;
                DCHK(scratch_get(c, SCR_V2, (size_t)(kk + 1) * s.ld * sizeof(double), &pv));
                DCHK(scratch_get(c, SCR_AV2, (size_t)(kk + 1) * s.ld * sizeof(double), &pmid));
                double* pin;
                DCHK(scratch_get(c, SCR_R, (size_t)(kk + 1) * s.ld * sizeof(double), &pin));
                DHIP(s_memcpy(c, pin, s.Vp, (size_t)kk * s.ld * sizeof(double), hipMemcpyDeviceToDevice));
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
                    DCHK(h2d_async(c, t, rnd.data(), (size_t)n * sizeof(double)));
                    DCHK(stream_wait(c));
                    DCHK(orthonormalise(s, t, s.k, &kept, nullptr));
                    if (!kept) { stop = true; }
                }
            }
            if (stop) break;
            DCHK(append_vector(s, Wc));
            if (s.qt_mode) DCHK(qt_update(s, s.k - 1, 1));
        }
        last_seek = seeking;
    }
    cohort_barrier(c, 0xffffff, 0);           // (a cohort closes up behind loops of different length)

    const double t_loop_end = now();
    if (dbg_time)
        fprintf(stderr, "davidson: k=%d total %.3f ms, host k x k algebra %.3f ms, fused iterations %ld, synchronous %ld\n", s.k,
                1e3 * (now() - t_begin), 1e3 * t_host, n_fast, n_slow);
    // ---- results: Ritz values, V and AV rotated into the Ritz basis ONCE, as (n x k) row-major host arrays -----
    const int k = s.k;
    for (int i = 0; i < k; ++i) lams_out[i] = lams[i];
    {
        double *dWp, *cols;
        DCHK(put_small(s, Wc.data(), k * k, 0, 0, &dWp));
        // both panels rotated and in the caller's layout on the device, one transfer, plain copies into the caller's arrays
        DCHK(scratch_get(c, SCR_X, 2 * (size_t)n * k * sizeof(double), &cols));
        const dim3 grot((n + 255) / 256, 2);
        if (k <= 16 && c->opt.dav_rotate_fused) {
            SELLA_LAUNCHB(c, HIP_KERNEL_NAME(dav_rotate_columns_kernel<16>), SELLA_BODY(dav_rotate_columns_vb<16>), 256, grot, dim3(256), 0, n, k,
                          (const double*)dWp, (const double*)s.Vp, (const double*)s.AVp, s.ld, cols);
        } else if (k <= 32 && c->opt.dav_rotate_fused) {
            SELLA_LAUNCHB(c, HIP_KERNEL_NAME(dav_rotate_columns_kernel<32>), SELLA_BODY(dav_rotate_columns_vb<32>), 256, grot, dim3(256), 0, n, k,
                          (const double*)dWp, (const double*)s.Vp, (const double*)s.AVp, s.ld, cols);
        } else {
            DCHK(launch_lincomb(c, n, k, s.Vp, s.ld, k, dWp, k, nullptr, 0, 0, nullptr, 0, 0.0, s.Vq, s.ld));
            DCHK(launch_lincomb(c, n, k, s.AVp, s.ld, k, dWp, k, nullptr, 0, 0, nullptr, 0, 0.0, s.AVq, s.ld));
            SELLA_LAUNCHB(c, dav_to_columns_kernel, dav_to_columns_vb, 256, dim3((n + 31) / 32, 2 * ((k + 31) / 32)), dim3(256), 0, n, k,
                          (const double*)s.Vq, (const double*)s.AVq, s.ld, cols);
        }
        DHIP(hipGetLastError());
        if (AV_out == V_out + (size_t)n * k) {
            DCHK(d2h_async(c, V_out, cols, 2 * (size_t)n * k * sizeof(double)));
        } else {
            DCHK(d2h_async(c, V_out, cols, (size_t)n * k * sizeof(double)));
            DCHK(d2h_async(c, AV_out, cols + (size_t)n * k, (size_t)n * k * sizeof(double)));
        }
        DCHK(stream_wait(c));
    }
    *k_out = k;
    if (nmatvec_out) *nmatvec_out = s.nmatvec;
    const double t_res_end = now();
    dav_free(s);
    if (dbg_time)
        fprintf(stderr, "davidson: allocation + start block %.1f us, loop %.1f us, Ritz vectors and images to the host %.1f us, release %.1f us\n",
                1e6 * (t_begin - t_enter), 1e6 * (t_loop_end - t_begin), 1e6 * (t_res_end - t_loop_end), 1e6 * (now() - t_res_end));
    return SELLA_OK;
#undef DCHK
#undef DHIP
}
