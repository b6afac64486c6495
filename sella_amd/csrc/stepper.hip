// stepper.hip — one evaluation of the step families of sella/optimize/stepper.py in the
// eigenbasis of the (projected) approximate Hessian.
//
// The reference evaluates RFO / P-RFO by a fresh dense eigh of the (m+1) x (m+1) scaled
// augmented matrix for every trial alpha (stepper.py:128-131; ~4.8 s per alpha at m = 3072).
// In the eigenbasis H = V diag(lam) V^T that matrix is a BORDERED DIAGONAL
//     [[alpha^2 diag(lam), alpha ghat], [alpha ghat^T, 0]],   ghat = V^T g,
// whose eigenvalues are the roots of  f(mu) = mu + sum_i b_i^2 / (D_i - mu)  and whose
// eigenvectors are (b_i / (mu - D_i), 1) — O(m) host arithmetic per alpha.  The analytic
// d s / d alpha of stepper.py:139-156 (first-order perturbation theory) becomes one
// pseudo-inverse solve with the same bordered-diagonal matrix.  The only O(m^2) work per alpha
// is mapping (s, ds/dalpha) back with one 2-right-hand-side row-panel matvec on the device.
#include "internal.h"
#include "bordered.h"
#include <chrono>
#include <cstdlib>

#include <algorithm>

struct sella_stepper {
    sella_ctx* c = nullptr;
    int kind = 0, m = 0, order = 0, nout = 0;
    sella_mat V = SELLA_NO_MAT;      // m x m eigenvectors (columns)   [not owned]
    sella_mat VU = SELLA_NO_MAT;     // nout x m : U V when a projection U (nout x m) was given [owned]
    sella_mat ownV = SELLA_NO_MAT, ownVt = SELLA_NO_MAT;   // mode matrices built by sella_stepper_create_lr [owned]
    std::vector<double> lam, ghat;
    std::vector<double> d1hat;       // V^T d1 of the IRC quasi-Newton family (kind SELLA_STEP_QN_IRC)
    sella_mat Vt = SELLA_NO_MAT;     // rows = eigenvectors [not owned]; needed for V^T scons in the root finder
    double t_host = 0.0, t_dev = 0.0;     // SELLA_DEBUG_TIMING: seconds in the secular solves / in the device round trip
    long calls = 0, sweeps = 0;
    long rs_calls = 0, rs_rounds = 0, rs_single = 0;   // SELLA_DEBUG_TIMING: root searches, batched round trips, single-alpha round trips
    bool boundary_hint = false;           // sella_opt_step: the previous step ended on the trust boundary
    bool fast_search = false;             // sella_opt_step: interpolating batched search instead of the reference's alpha schedule
    double alpha_hint = 0.0;              // ... where the previous root search of the same saddle search ended (0: unknown)
    double alpha_found = 0.0;             // ... and where this one did
    // Panel form (stepper_on_panel): the modes are rows pidx[i] of a device panel somebody else owns, never gathered into
    // matrices — enough for the search in the orthonormal eigenbasis (trust-region measure), whose only device work is
    // the step itself at the final alpha: s = sum_i shat_i row_i, one launch.
    const double* panel = nullptr;
    int panel_ld = 0;
    std::vector<int> pidx;
    std::vector<double> pscale;           // factor of each row (1 unless the row is stored unnormalised)
};

namespace sella {
namespace {

typedef std::vector<double> vec;

// RFO step in the eigenbasis for a block (lam, ghat) of size mm, eigenpair index `o` of the
// augmented matrix (stepper.py:128-157).  Outputs shat, dshat (mm).  dshat == nullptr: the step only (the bisection
// phase of the trust-radius search never looks at ds/dalpha).  shat == nullptr as well: nothing is written and
// |shat|^2 is returned — for a unit-norm eigenvector (y, 1) / sqrt(1 + |y|^2) the step is alpha y, so its norm is
// alpha |y| and costs one pass behind the root find.
double rfo_block(int mm, const double* lam, const double* ghat, int o, double alpha, double* shat,
                 double* dshat) {
    if (mm == 0) return 0.0;
    static thread_local vec D, b, vh, c1, xp;            // ~50 calls per optimizer step: no allocation per call
    D.resize(mm); b.resize(mm); vh.resize(mm + 1); c1.resize(mm + 1); xp.resize(mm + 1);
    for (int i = 0; i < mm; ++i) { D[i] = alpha * alpha * lam[i]; b[i] = alpha * ghat[i]; }
    int org;
    double tau;
    bordered::bordered_root(mm, D.data(), b.data(), o, &org, &tau, /*wide=*/true);
    const double shift = org >= 0 ? D[org] : 0.0;
    // eigenvector (unnormalised): y_i = b_i / (mu - D_i), eta = 1
    double nrm2 = 1.0;
    bool degenerate = false;
    for (int i = 0; i < mm; ++i) {
        const double den = tau - (D[i] - shift);          // mu - D_i
        if (den == 0.0) { degenerate = true; break; }
        vh[i] = b[i] / den;
        nrm2 += vh[i] * vh[i];
    }
    if (degenerate) {
        // mu coincides with a pole (b_i = 0 there): the eigenvector is e_i, its last component is
        // zero and the reference clamps the denominator at 1e-12 (stepper.py:134-136)
        if (!shat) return (alpha / 1e-12) * (alpha / 1e-12);
        int ip = 0;
        for (int i = 0; i < mm; ++i) if (tau - (D[i] - shift) == 0.0) { ip = i; break; }
        for (int i = 0; i < mm; ++i) shat[i] = 0.0;
        shat[ip] = alpha / 1e-12;
        if (dshat) {
            for (int i = 0; i < mm; ++i) dshat[i] = 0.0;
            dshat[ip] = 1.0 / 1e-12;
        }
        return shat[ip] * shat[ip];
    }
    if (!shat) {
        const double invn = 1.0 / sqrt(nrm2);
        double dn = invn;
        if (fabs(dn) < 1e-12) dn = 1e-12;
        const double f = invn * alpha / dn;              // shat_i = y_i * f
        return f * f * (nrm2 - 1.0);
    }
    const double inv = 1.0 / sqrt(nrm2);
    for (int i = 0; i < mm; ++i) vh[i] *= inv;
    vh[mm] = inv;
    double den = vh[mm];
    if (fabs(den) < 1e-12) den = 1e-12;
    double ss = 0.0;
    for (int i = 0; i < mm; ++i) { shat[i] = vh[i] * alpha / den; ss += shat[i] * shat[i]; }
    if (!dshat) return ss;
    // c = dA/dalpha v : dA = [[2 alpha lam, ghat], [ghat^T, 0]]
    double last = 0.0;
    for (int i = 0; i < mm; ++i) {
        c1[i] = 2.0 * alpha * lam[i] * vh[i] + ghat[i] * vh[mm];
        last += ghat[i] * vh[i];
    }
    c1[mm] = last;
    // x = sum_{j != o} v_j (v_j . c) / (L_j - L_o) = pinv(A - mu) c on the complement of v
    double vc = 0.0;
    for (int i = 0; i <= mm; ++i) vc += vh[i] * c1[i];
    for (int i = 0; i <= mm; ++i) c1[i] -= vc * vh[i];
    // particular solution with eta = 0: (D - mu) y = c1[:mm]
    for (int i = 0; i < mm; ++i) {
        double dd = (D[i] - shift) - tau;
        if (fabs(dd) < 1e-12) dd = (dd >= 0.0) ? 1e-12 : -1e-12;       // stepper.py:143-146 clamp
        xp[i] = c1[i] / dd;
    }
    xp[mm] = 0.0;
    double vx = 0.0;
    for (int i = 0; i <= mm; ++i) vx += vh[i] * xp[i];
    for (int i = 0; i <= mm; ++i) xp[i] -= vx * vh[i];
    for (int i = 0; i < mm; ++i)
        dshat[i] = vh[i] / den + (alpha / den) * xp[i] - (vh[i] * alpha / (den * den)) * xp[mm];
    return ss;
}

}  // namespace
}  // namespace sella

using namespace sella;

// ghat_in != nullptr: V^T g is known already (g is then not read)
static int stepper_make(sella_ctx* c, int kind, sella_mat hV, sella_mat hVt, const double* evals,
                        const double* g, const double* ghat_in, int m, int order, sella_stepper** out) {
    if (!c || !out || !evals || (!g && !ghat_in) || m <= 0 || order < 0 || order > m) {
        set_error("stepper: invalid arguments");
        return SELLA_E_INVALID;
    }
    if (kind < SELLA_STEP_QN || kind > SELLA_STEP_QN_IRC) {
        set_error("Unknown stepper kind %d", kind);
        return SELLA_E_INVALID;
    }
    Mat *V = mat_get(c, hV), *Vt = mat_get(c, hVt);
    if (!V || !Vt) return SELLA_E_INVALID;
    if (V->cols != m || Vt->rows != m || Vt->cols != V->rows) {
        set_error("stepper: eigenvector matrices must be (nout x %d) and (%d x nout)", m, m);
        return SELLA_E_INVALID;
    }
    sella_stepper* st = new sella_stepper();
    st->c = c;
    st->kind = kind;
    st->m = m;
    st->order = order;
    st->nout = V->rows;
    st->V = hV;
    st->Vt = hVt;
    st->lam.assign(evals, evals + m);
    st->ghat.resize(m);
    if (ghat_in) {
        std::copy(ghat_in, ghat_in + m, st->ghat.begin());
        *out = st;
        return SELLA_OK;
    }
    // ghat = V^T g through the row form: rows of Vt are the eigenvectors
    const int nin = Vt->cols;
    const int ldx = round_up(nin, 8), ldy = round_up(m, 8);
    double *dx, *dy;
    int s = scratch_get(c, SCR_STEP0, (size_t)2 * std::max(ldx, ldy) * sizeof(double), &dx);
    if (s == SELLA_OK) s = scratch_get(c, SCR_STEP1, (size_t)2 * std::max(ldx, ldy) * sizeof(double), &dy);
    if (s == SELLA_OK) s = upload_panel(c, g, nin, 1, dx, ldx);
    if (s == SELLA_OK) {
        Vt = mat_get(c, hVt);
        s = launch_gemv_rows(c, Vt->d, m, nin, Vt->ld, dx, ldx, 1, dy, ldy, GemvEpi());
    }
    if (s == SELLA_OK) s = download_panel(c, dy, ldy, m, 1, st->ghat.data());
    if (s != SELLA_OK) { delete st; return s; }
    *out = st;
    return SELLA_OK;
}

// (shat, dshat) of one trial alpha in the eigenbasis — O(m) host arithmetic.  Returns |shat|^2; dshat (and, for the
// RFO families, shat as well) may be null when only that is wanted.
extern "C" int sella_stepper_create(sella_ctx* c, int kind, sella_mat hV, sella_mat hVt, const double* evals,
                                    const double* g, int m, int order, sella_stepper** out) {
    if (!g) { set_error("stepper: invalid arguments"); return SELLA_E_INVALID; }
    return stepper_make(c, kind, hV, hVt, evals, g, nullptr, m, order, out);
}

static double eval_hat(const sella_stepper* st, double alpha, double* shat, double* dshat) {
    const int m = st->m, o = st->order;
    const double* lam = st->lam.data();
    const double* gh = st->ghat.data();
    double ss = 0.0;
    if (st->kind == SELLA_STEP_QN) {                                        // stepper.py:82-96
        for (int i = 0; i < m; ++i) {
            const double sgn = (i < o) ? -1.0 : 1.0;
            const double den = sgn * fabs(lam[i]) + alpha * sgn;
            const double sp = gh[i] / den;
            ss += sp * sp;
            if (shat) shat[i] = -sp;
            if (dshat) dshat[i] = sp / den;
        }
    } else if (st->kind == SELLA_STEP_QN_IRC) {                             // stepper.py:99-111
        const double* dh = st->d1hat.data();
        for (int i = 0; i < m; ++i) {
            const double den = fabs(lam[i]) + alpha;
            const double sp = -(gh[i] + alpha * dh[i]) / den;
            ss += sp * sp;
            if (shat) shat[i] = sp;
            if (dshat) dshat[i] = -(sp + dh[i]) / den;
        }
    } else if (st->kind == SELLA_STEP_RFO) {
        ss = sella::rfo_block(m, lam, gh, o, alpha, shat, dshat);
    } else {                                                                // P-RFO, stepper.py:163-185
        ss = sella::rfo_block(o, lam, gh, o, alpha, shat, dshat);                               // max block: top root
        ss += sella::rfo_block(m - o, lam + o, gh + o, 0, alpha, shat ? shat + o : nullptr,    // min block: lowest root
                               dshat ? dshat + o : nullptr);
    }
    return ss;
}

extern "C" int sella_stepper_get_s(sella_stepper* st, double alpha, double* s_out, double* dsda_out) {
    if (!st || !s_out || !dsda_out) return SELLA_E_INVALID;
    sella_ctx* c = st->c;
    const int m = st->m;
    const auto t0 = std::chrono::steady_clock::now();
    const long sw0 = bordered::g_sweeps;
    std::vector<double> sh(2 * (size_t)m, 0.0);     // [shat | dshat]
    double* shat = sh.data();
    double* dshat = sh.data() + m;
    eval_hat(st, alpha, shat, dshat);
    const auto t1 = std::chrono::steady_clock::now();
    struct Acc {
        sella_stepper* st; std::chrono::steady_clock::time_point a, b; long sw;
        ~Acc() {
            st->t_host += std::chrono::duration<double>(b - a).count();
            st->t_dev += std::chrono::duration<double>(std::chrono::steady_clock::now() - b).count();
            st->calls += 1;
            st->sweeps += sw;
        }
    } acc{st, t0, t1, bordered::g_sweeps - sw0};
    Mat* V = mat_get(c, st->V);
    if (!V) return SELLA_E_INVALID;
    const int nout = st->nout;
    const int ldx = round_up(m, 8), ldy = round_up(nout, 8);
    double *dx, *dy;
    SCHK(scratch_get(c, SCR_STEP0, (size_t)2 * std::max(ldx, ldy) * sizeof(double), &dx));
    SCHK(scratch_get(c, SCR_STEP1, (size_t)2 * std::max(ldx, ldy) * sizeof(double), &dy));
    // The trust-radius search calls this ~50 times per optimizer step, so the round trip is kept short: the two
    // coefficient vectors go down as ONE copy from the pinned exchange buffer (a pageable hipMemcpyAsync is a
    // synchronous staged copy, ~0.1 ms), and the matvec writes its 2 x nout results straight into that pinned,
    // device-visible buffer (each output is written once: posted PCIe writes, no copy-back launches); the
    // stream synchronisation is then the only wait.
    const bool pinned = 2 * ldx <= 16384 && 2 * ldy <= 16384;
    double* hin = c->hscal + DS_STAGE;
    double* hout = c->hscal + DS_STAGE + 16384;
    if (pinned) {
        memcpy(hin, shat, (size_t)m * sizeof(double));
        memcpy(hin + ldx, dshat, (size_t)m * sizeof(double));
        HIPCHK(s_memcpy(c, dx, hin, (size_t)(ldx + m) * sizeof(double), hipMemcpyHostToDevice, true));
        SCHK(launch_gemv_rows(c, V->d, nout, m, V->ld, dx, ldx, 2, hout, ldy, GemvEpi()));
        SCHK(stream_wait(c));
        memcpy(s_out, hout, (size_t)nout * sizeof(double));
        memcpy(dsda_out, hout + ldy, (size_t)nout * sizeof(double));
    } else {
        SCHK(h2d_async(c, dx, shat, (size_t)m * sizeof(double)));
        SCHK(h2d_async(c, dx + ldx, dshat, (size_t)m * sizeof(double)));
        SCHK(launch_gemv_rows(c, V->d, nout, m, V->ld, dx, ldx, 2, dy, ldy, GemvEpi()));
        SCHK(d2h_async(c, s_out, dy, (size_t)nout * sizeof(double)));
        SCHK(d2h_async(c, dsda_out, dy + ldy, (size_t)nout * sizeof(double)));
        SCHK(stream_wait(c));
    }
    return SELLA_OK;
}

extern "C" int sella_stepper_set_d1hat(sella_stepper* st, const double* d1hat, int m) {
    if (!st || !d1hat || m != st->m) {
        set_error("stepper: d1hat must have %d entries", st ? st->m : -1);
        return SELLA_E_INVALID;
    }
    st->d1hat.assign(d1hat, d1hat + m);
    return SELLA_OK;
}

namespace sella {
namespace {

// ---------------------------------------------------------------------------------------------------------------------
// Batched trial alphas.  Once the reference's schedule has degenerated into pure bisection (restricted_step.py:100-110:
// niter > 4 and a family that is not newton_safe) the next L levels of trial alphas form a binary tree of 2^L - 1
// midpoints that is known before any of them is evaluated.  The whole tree is evaluated in ONE round trip:
//   rs_batch_kernel    one workgroup per candidate: shat(alpha) in the eigenbasis (the bordered-diagonal root find with
//                      its O(m) sums as block reductions, bordered.h) -> a 16-row panel
//   panel16            V . panel on the matrix cores: the eigenvector matrix is streamed once for all candidates
//   rs_measure_kernel  one workgroup per candidate: the constraint value
// and the host then walks the tree with the reference's control flow.  The alphas visited are exactly the reference's.
// ---------------------------------------------------------------------------------------------------------------------
struct RsBatchArgs {
    int kind, m, order, ncand, ldx;
    double alpha[16];
    const double* lam; const double* ghat; const double* d1hat;
    double* X;                                  // 16 x ldx panel, rows >= ncand and columns >= m stay zero
};

// block-wide sums of 4 values (256 threads); every thread gets the result
__device__ __forceinline__ void block_sum4_256(double v[4], double (*red)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = wave_sum64(v[q]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int q = 0; q < 4; ++q) red[threadIdx.x >> 6][q] = v[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = red[0][q] + red[1][q] + red[2][q] + red[3][q];
}

// shat of one RFO block (rfo_block above, without ds/dalpha).  WAVE = false: by the whole workgroup, the O(m) sums as block
// reductions (two barriers each).  WAVE = true: by ONE wavefront (the caller lets only wavefront 0 in), the sums as
// wavefront reductions — a root costs 6-10 evaluations of the secular function, and for the few dozen to few hundred
// modes of a structured eigendecomposition the barriers were the whole cost (14 us per candidate and block).
template <bool WAVE>
__device__ void rfo_block_dev(int mm, const double* __restrict__ lam, const double* __restrict__ gh, int o, double alpha,
                              double* __restrict__ shat, double (*red)[4]) {
    if (mm == 0) return;
    const int tid = WAVE ? (threadIdx.x & 63) : threadIdx.x;
    constexpr int NT = WAVE ? 64 : 256;
    auto sum4 = [&](double v[4]) {
        if (WAVE) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = wave_sum64(v[q]);
        } else {
            block_sum4_256(v, red);
        }
    };
    const double a2 = alpha * alpha;
    auto Dat = [&](int i) { return a2 * lam[i]; };
    auto bat = [&](int i) { return alpha * gh[i]; };
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = tid; i < mm; i += NT) { const double b = bat(i); acc[0] += b * b; }
    sum4(acc);
    const double bb = acc[0];
    auto eval = [&](double shift, double t) {
        double v[4] = {0.0, 0.0, 0.0, 0.0};                  // s, sum |q|, dl, dr
        for (int i = tid; i < mm; i += NT) {
            const double b = bat(i);
            const double r = 1.0 / ((Dat(i) - shift) - t);
            const double q = b * b * r;
            v[0] += q;
            v[1] += fabs(q);
            if (i < o) v[2] += q * r; else v[3] += q * r;
        }
        sum4(v);
        bordered::Ev e;
        e.f = (shift + t) + v[0];
        e.noise = fabs(shift + t) + v[1];
        e.dl = v[2];
        e.dr = v[3];
        return e;
    };
    int org;
    double tau;
    bordered::bordered_root_core(mm, Dat, bat, bb, o, eval, &org, &tau);
    const double shift = org >= 0 ? Dat(org) : 0.0;
    // eigenvector (unnormalised): y_i = b_i / (mu - D_i), eta = 1
    double w[4] = {0.0, 0.0, 0.0, 0.0};                      // sum y^2, degenerate flag
    for (int i = tid; i < mm; i += NT) {
        const double den = tau - (Dat(i) - shift);
        if (den == 0.0) { w[1] += 1.0; }
        else { const double y = bat(i) / den; w[0] += y * y; }
    }
    sum4(w);
    if (w[1] > 0.0) {
        // mu coincides with a pole: the eigenvector is e_i of the FIRST such pole, last component zero, denominator
        // clamped at 1e-12 (stepper.py:134-136)
        int mine = 0x7fffffff;
        for (int i = tid; i < mm; i += NT)
            if (tau - (Dat(i) - shift) == 0.0 && i < mine) mine = i;
        int first;
        if (WAVE) {
            for (int off = 32; off > 0; off >>= 1) { const int other = __shfl_xor(mine, off); mine = other < mine ? other : mine; }
            first = mine;
        } else {
            __shared__ int sfirst;
            if (tid == 0) sfirst = 0x7fffffff;
            __syncthreads();
            if (mine != 0x7fffffff) atomicMin(&sfirst, mine);
            __syncthreads();
            first = sfirst;
        }
        for (int i = tid; i < mm; i += NT) shat[i] = (i == first) ? alpha / 1e-12 : 0.0;
        if (!WAVE) __syncthreads();
        return;
    }
    const double inv = 1.0 / sqrt(1.0 + w[0]);
    double den = inv;
    if (fabs(den) < 1e-12) den = 1e-12;
    for (int i = tid; i < mm; i += NT) {
        const double y = bat(i) / (tau - (Dat(i) - shift));
        shat[i] = (y * inv) * alpha / den;
    }
}

constexpr int RS_LDS_MAX = 2048;

__device__ __forceinline__ void rs_batch_vb(const VB vb, RsBatchArgs a) {
    __shared__ double red[4][4];
    __shared__ double sLam[RS_LDS_MAX], sG[RS_LDS_MAX];
    const int cand = vb.x, tid = threadIdx.x;
    if (cand >= a.ncand) return;
    const double alpha = a.alpha[cand];
    double* shat = a.X + (size_t)cand * a.ldx;
    const int m = a.m, o = a.order;
    // every evaluation of the secular function walks the poles and weights again (6-10 times per root): from LDS, not
    // from global memory — a round trip per evaluation was most of this kernel's 13 us
    if (m <= RS_LDS_MAX && (a.kind == SELLA_STEP_RFO || a.kind == SELLA_STEP_PRFO)) {
        for (int i = tid; i < m; i += 256) { sLam[i] = a.lam[i]; sG[i] = a.ghat[i]; }
        __syncthreads();
        a.lam = sLam;
        a.ghat = sG;
    }
    if (a.kind == SELLA_STEP_QN) {
        for (int i = tid; i < m; i += 256) {
            const double sgn = (i < o) ? -1.0 : 1.0;
            shat[i] = -(a.ghat[i] / (sgn * fabs(a.lam[i]) + alpha * sgn));
        }
    } else if (a.kind == SELLA_STEP_QN_IRC) {
        for (int i = tid; i < m; i += 256) shat[i] = -(a.ghat[i] + alpha * a.d1hat[i]) / (fabs(a.lam[i]) + alpha);
    } else if (m <= 1024) {
        // few modes (the structured eigendecompositions: r + 1 of them): one wavefront, no barriers
        if (tid < 64) {
            if (a.kind == SELLA_STEP_RFO) {
                rfo_block_dev<true>(m, a.lam, a.ghat, o, alpha, shat, red);
            } else {
                rfo_block_dev<true>(o, a.lam, a.ghat, o, alpha, shat, red);
                rfo_block_dev<true>(m - o, a.lam + o, a.ghat + o, 0, alpha, shat + o, red);
            }
        }
    } else if (a.kind == SELLA_STEP_RFO) {
        rfo_block_dev<false>(m, a.lam, a.ghat, o, alpha, shat, red);
    } else {
        rfo_block_dev<false>(o, a.lam, a.ghat, o, alpha, shat, red);
        rfo_block_dev<false>(m - o, a.lam + o, a.ghat + o, 0, alpha, shat + o, red);
    }
}
__global__ __launch_bounds__(256) void rs_batch_kernel(RsBatchArgs a) { rs_batch_vb(vb_hw(), a); }

// value of the constraint for candidate blockIdx.x: y = V shat (row blockIdx.x of Y), optionally scattered through the
// inverse selection map (inv[i] = position of full coordinate i in the family's space, -1: not a free coordinate)
__device__ __forceinline__ void rs_measure_vb(const VB vb, int cons, int nout, const double* __restrict__ Y, int ldy,
                                                         const double* __restrict__ scons, const double* __restrict__ w,
                                                         const double* __restrict__ d1, const int* __restrict__ inv,
                                                         double* out) {
    __shared__ double red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* y = Y + (size_t)vb.x * ldy;
    auto stot = [&](int i) {
        double s;
        if (inv) { const int p = inv[i]; s = (p >= 0) ? y[p] : 0.0; }
        else s = y[i];
        return s + (scons ? scons[i] : 0.0);
    };
    double acc = 0.0;
    if (cons == 0 || cons == 3) {
        for (int i = tid; i < nout; i += 256) {
            double x = stot(i);
            if (cons == 3) x = (x + d1[i]) * w[i];
            acc += x * x;
        }
        acc = wave_sum64(acc);
    } else {
        if (cons == 1) {
            for (int at = tid; at < nout / 3; at += 256) {
                double n2 = 0.0;
#pragma unroll
                for (int q = 0; q < 3; ++q) { const double st = stot(3 * at + q); n2 += st * st; }
                acc = fmax(acc, sqrt(n2));
            }
        } else {
            for (int i = tid; i < nout; i += 256) acc = fmax(acc, fabs(stot(i) * w[i]));
        }
        for (int off = 32; off > 0; off >>= 1) acc = fmax(acc, __shfl_xor(acc, off));
    }
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid == 0) {
        if (cons == 0 || cons == 3) out[vb.x] = sqrt(red[0] + red[1] + red[2] + red[3]);
        else out[vb.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    }
}
__global__ __launch_bounds__(256) void rs_measure_kernel(int cons, int nout, const double* __restrict__ Y, int ldy,
                                                         const double* __restrict__ scons, const double* __restrict__ w,
                                                         const double* __restrict__ d1, const int* __restrict__ inv,
                                                         double* out) { rs_measure_vb(vb_hw(), cons, nout, Y, ldy, scons, w, d1, inv, out); }

// Constraint measure of the total step and its derivative along the family, in ONE workgroup:
//   stot = s + scons (written back to `stot`),  out[0] = val, out[1] = dval.
//   cons 0: |stot|, dsda.stot / |stot|;   cons 3: the same for (stot + d1) * w and dsda * w;
//   cons 1: atom with the largest |stot_a|: (|stot_a|, dsda_a.stot_a / |stot_a|);
//   cons 2: component with the largest |stot_i w_i|: (that value, sign(stot_i) dsda_i w_i).
// Ties resolve to the lowest index (np.argmax).  val is clamped at 1e-12 in the denominators like the reference.
__device__ __forceinline__ void rs_cons_vb(const VB vb, int cons, int nout, const double* s_in, const double* dsda_in,
                                                       const double* __restrict__ scons, const double* __restrict__ w,
                                                       const double* __restrict__ d1, double* stot, double* out,
                                                       const int* __restrict__ sel, int m, double* sfull, double* dfull) {
    __shared__ double r1[16], r2[16];
    __shared__ int ri[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* s = s_in;
    const double* dsda = dsda_in;
    if (sel != nullptr) {
        // the family works in the subspace of the free coordinates (basis = columns of the identity): scatter
        // (s, ds/dalpha) into the full space first, zeros elsewhere
        for (int i = tid; i < nout; i += 1024) { sfull[i] = 0.0; dfull[i] = 0.0; }
        __syncthreads();
        for (int i = tid; i < m; i += 1024) { sfull[sel[i]] = s_in[i]; dfull[sel[i]] = dsda_in[i]; }
        __syncthreads();
        s = sfull;
        dsda = dfull;
    }
    if (cons == 0 || cons == 3) {
        double a = 0.0, b = 0.0;
        for (int i = tid; i < nout; i += 1024) {
            const double st = s[i] + (scons ? scons[i] : 0.0);
            stot[i] = st;
            double x = st, d = dsda[i];
            if (cons == 3) { x = (st + d1[i]) * w[i]; d *= w[i]; }
            a += x * x;
            b += d * x;
        }
        a = wave_sum64(a);
        b = wave_sum64(b);
        if (lane == 0) { r1[wave] = a; r2[wave] = b; }
        __syncthreads();
        if (tid == 0) {
            double A = 0.0, B = 0.0;
            for (int q = 0; q < 16; ++q) { A += r1[q]; B += r2[q]; }
            const double val = sqrt(A);
            out[0] = val;
            out[1] = B / (val > 1e-12 ? val : 1e-12);
        }
        return;
    }
    // argmax kinds: every thread scans its groups, keeps (value, index, derivative numerator)
    double best = -1.0, bd = 0.0;
    int bi = 0x7fffffff;
    if (cons == 1) {
        const int natoms = nout / 3;
        for (int a = tid; a < natoms; a += 1024) {
            double n2 = 0.0, dd = 0.0;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int i = 3 * a + q;
                const double st = s[i] + (scons ? scons[i] : 0.0);
                stot[i] = st;
                n2 += st * st;
                dd += dsda[i] * st;
            }
            const double v = sqrt(n2);
            if (v > best) { best = v; bi = a; bd = dd; }
        }
    } else {
        for (int i = tid; i < nout; i += 1024) {
            const double st = s[i] + (scons ? scons[i] : 0.0);
            stot[i] = st;
            const double v = fabs(st * w[i]);
            if (v > best) { best = v; bi = i; bd = (st >= 0.0 ? 1.0 : -1.0) * dsda[i] * w[i]; }
        }
    }
    // wave reduction by shuffles (64 lanes), then across the 16 waves; lower index wins ties
    for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(best, off);
        const double od = __shfl_xor(bd, off);
        const int oi = __shfl_xor(bi, off);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bd = od; bi = oi; }
    }
    if (lane == 0) { r1[wave] = best; r2[wave] = bd; ri[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int q = 1; q < 16; ++q)
            if (r1[q] > best || (r1[q] == best && ri[q] < bi)) { best = r1[q]; bd = r2[q]; bi = ri[q]; }
        out[0] = best;
        out[1] = (cons == 1) ? bd / (best > 1e-12 ? best : 1e-12) : bd;
    }
}
__global__ __launch_bounds__(1024) void rs_cons_kernel(int cons, int nout, const double* s_in, const double* dsda_in,
                                                       const double* __restrict__ scons, const double* __restrict__ w,
                                                       const double* __restrict__ d1, double* stot, double* out,
                                                       const int* __restrict__ sel, int m, double* sfull, double* dfull) { rs_cons_vb(vb_hw(), cons, nout, s_in, dsda_in, scons, w, d1, stot, out, sel, m, sfull, dfull); }

}  // namespace
}  // namespace sella

// Y[q * ldy + c] = sum_i X[q * ldx + i] * scale[i] * panel[idx[i] * ld + c],  q < nq (<= 16): trial steps of a family in panel
// form — its modes are rows of a panel somebody else owns, read where they are.  `aux` = m doubles (factor of each row),
// then m ints (row of each mode), uploaded once per family.  Thread = coordinate; the coefficients go through LDS in
// tiles of 128 modes (read back as broadcasts).
__device__ __forceinline__ void rs_panel_apply_vb(const VB vb, const double* __restrict__ panel, int ld, int m, int n,
                                                             const double* __restrict__ aux, const double* __restrict__ X,
                                                             int ldx, int nq, double* __restrict__ Y, int ldy) {
    constexpr int TI = 128;
    __shared__ double cs[TI][16];
    __shared__ int si[TI];
    const int* idx = reinterpret_cast<const int*>(aux + m);
    const int c = vb.x * 256 + threadIdx.x;
    const int cl = c < n ? c : n - 1;
    double acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0;
    for (int i0 = 0; i0 < m; i0 += TI) {
        const int nb = m - i0 < TI ? m - i0 : TI;
        __syncthreads();
        for (int e = threadIdx.x; e < nb * 16; e += 256) {
            const int i = e >> 4, q = e & 15;
            cs[i][q] = q < nq ? X[(size_t)q * ldx + i0 + i] * aux[i0 + i] : 0.0;
        }
        for (int i = threadIdx.x; i < nb; i += 256) si[i] = idx[i0 + i];
        __syncthreads();
        if (nq <= 2) {
            for (int i = 0; i < nb; ++i) {
                const double v = panel[(size_t)si[i] * ld + cl];
                acc[0] += cs[i][0] * v;
                acc[1] += cs[i][1] * v;
            }
        } else {
            for (int i = 0; i < nb; ++i) {
                const double v = panel[(size_t)si[i] * ld + cl];
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] += cs[i][q] * v;
            }
        }
    }
    if (c < n)
        for (int q = 0; q < nq; ++q) Y[(size_t)q * ldy + c] = acc[q];
}
__global__ __launch_bounds__(256) void rs_panel_apply_kernel(const double* __restrict__ panel, int ld, int m, int n,
                                                             const double* __restrict__ aux, const double* __restrict__ X,
                                                             int ldx, int nq, double* __restrict__ Y, int ldy) { rs_panel_apply_vb(vb_hw(), panel, ld, m, n, aux, X, ldx, nq, Y, ldy); }

extern "C" int sella_restricted_step(sella_stepper* st, int cons, double delta, const double* scons, const double* w,
                                     const double* d1, double alpha0, double alphamin, double alphamax, double slope,
                                     int newton_safe, int orthonormal, double tol, int maxiter, const int* sel,
                                     int nfull, double* s_out, double* val_out, double* alphas, int* nalpha) {
    if (!st || !s_out || !val_out || cons < 0 || cons > 3 || maxiter < 0) {
        set_error("restricted_step: invalid arguments");
        return SELLA_E_INVALID;
    }
    if ((cons >= 2 && !w) || (cons == 3 && !d1)) {
        set_error("restricted_step: constraint kind %d needs weights%s", cons, cons == 3 ? " and d1" : "");
        return SELLA_E_INVALID;
    }
    if (st->kind == SELLA_STEP_QN_IRC && (int)st->d1hat.size() != st->m) {
        set_error("restricted_step: the IRC family needs sella_stepper_set_d1hat first");
        return SELLA_E_INVALID;
    }
    sella_ctx* c = st->c;
    Mat* V = st->panel ? nullptr : mat_get(c, st->V);
    if (!V && !st->panel) return SELLA_E_INVALID;
    if (st->panel && !(orthonormal && cons <= 1 && !scons)) {
        set_error("restricted_step: a family in panel form serves the trust-region and per-atom measures without a constraint correction only");
        return SELLA_E_INVALID;
    }
    // panel form: factors and row indices of the modes on the device (once per family)
    double* daux = nullptr;
    const int ldxp = round_up(st->m, 8);
    auto panel_apply = [&](const double* dX, int ldX, int nq, double* dYo, int ldYo) -> int {
        if (!daux) {
            const int mm = st->m;
            std::vector<double> pack((size_t)mm + (size_t)(mm + 1) / 2 + 1, 0.0);
            for (int i = 0; i < mm; ++i) pack[i] = st->pscale[i];
            memcpy(pack.data() + mm, st->pidx.data(), (size_t)mm * sizeof(int));
            SCHK(scratch_get(c, SCR_PSMALL, pack.size() * sizeof(double), &daux));
            SCHK(h2d_async(c, daux, pack.data(), pack.size() * sizeof(double)));
        }
        SELLA_LAUNCHB(c, rs_panel_apply_kernel, rs_panel_apply_vb, 256, dim3((st->nout + 255) / 256), dim3(256), 0, st->panel, st->panel_ld,
                           st->m, st->nout, daux, dX, ldX, nq, dYo, ldYo);
        HIPCHK(hipGetLastError());
        return SELLA_OK;
    };
    (void)ldxp;
    const int m = st->m;
    const int nfam = st->nout;                         // rows of the family's eigenvector matrix (m modes of length nfam;
                                                       // m < nfam for a structured eigendecomposition, sella_stepper_create_lr)
    if (sel && nfull < nfam) { set_error("restricted_step: a selection needs nfull >= the family's dimension"); return SELLA_E_INVALID; }
    const int nout = sel ? nfull : nfam;               // dimension of the step handed back
    if (cons == 1 && nout % 3 != 0) { set_error("restricted_step: per-atom measure needs 3 N components"); return SELLA_E_INVALID; }
    const int ldx = round_up(m, 8), ldy = round_up(std::max(nout, nfam), 8);
    double *dx, *dy, *dv;
    SCHK(scratch_get(c, SCR_STEP0, (size_t)2 * std::max(ldx, ldy) * sizeof(double), &dx));
    SCHK(scratch_get(c, SCR_STEP1, (size_t)2 * std::max(ldx, ldy) * sizeof(double), &dy));
    SCHK(scratch_get(c, SCR_QR0, (size_t)7 * ldy * sizeof(double), &dv));     // scons | w | d1 | stot | sfull | dfull | sel
    double* dscons = scons ? dv : nullptr;
    double* dw = w ? dv + ldy : nullptr;
    double* dd1 = d1 ? dv + 2 * (size_t)ldy : nullptr;
    double* dstot = dv + 3 * (size_t)ldy;
    double* dsfull = dv + 4 * (size_t)ldy;
    double* ddfull = dv + 5 * (size_t)ldy;
    int* dsel = sel ? reinterpret_cast<int*>(dv + 6 * (size_t)ldy) : nullptr;
    if (sel) SCHK(h2d_async(c, dsel, sel, (size_t)nfam * sizeof(int)));
    const bool eig_only = orthonormal && cons == 0;
    std::vector<double> chat;                 // V^T scons (eigenbasis measure)
    double scons2 = 0.0;
    if (scons) {
        for (int i = 0; i < nout; ++i) scons2 += scons[i] * scons[i];
        SCHK(h2d_async(c, dscons, scons, (size_t)nout * sizeof(double)));
    }
    if (w) SCHK(h2d_async(c, dw, w, (size_t)nout * sizeof(double)));
    if (d1) SCHK(h2d_async(c, dd1, d1, (size_t)nout * sizeof(double)));
    if (eig_only && scons && scons2 > 0.0) {
        Mat* Vt = mat_get(c, st->Vt);
        if (!Vt) return SELLA_E_INVALID;
        chat.resize(m);
        const double* dsrc = dscons;
        if (sel) {                                      // the family sees the free components of the correction only
            std::vector<double> sc(nfam);
            for (int i = 0; i < nfam; ++i) sc[i] = scons[sel[i]];
            SCHK(upload_panel(c, sc.data(), nfam, 1, dsfull, ldy));
            dsrc = dsfull;
        }
        SCHK(launch_gemv_rows(c, Vt->d, m, nfam, Vt->ld, dsrc, ldy, 1, dx, ldx, GemvEpi()));
        SCHK(download_panel(c, dx, ldx, m, 1, chat.data()));
    }
    const bool pinned = 2 * ldx <= 16384;
    double* hin = c->hscal + DS_STAGE;
    double* hres = c->hscal + DS_MISC + 64;            // val, dval land here (written by the kernel, pinned host memory)
    std::vector<double> sh(2 * (size_t)m, 0.0);
    double* shat = sh.data();
    double* dshat = sh.data() + m;
    int ntrial = 0;
    bool hat_stale = false;                   // shat on the host does not belong to the last trial alpha
    // one trial alpha -> (val, dval); leaves [shat | dshat] on the host and (unless eig_only) stot on the device
    // value_only: the caller is in the pure-bisection phase and ignores dval (restricted_step.py:100-110)
    auto evaluate = [&](double alpha, double* val, double* dval, bool value_only = false) -> int {
        if (alphas && ntrial <= maxiter) alphas[ntrial] = alpha;
        ++ntrial;
        const auto th0 = std::chrono::steady_clock::now();
        struct HostTime {                      // SELLA_DEBUG_TIMING: host arithmetic up to `mark`, the rest is the device round trip
            sella_stepper* st; std::chrono::steady_clock::time_point a, mark;
            ~HostTime() {
                st->t_host += std::chrono::duration<double>(mark - a).count();
                st->t_dev += std::chrono::duration<double>(std::chrono::steady_clock::now() - mark).count();
            }
        } ht{st, th0, th0};
        const long sw0 = bordered::g_sweeps;
        if (eig_only && value_only) {
            // |s + scons|^2 = |shat|^2 + 2 shat.chat + |scons|^2 in the orthonormal eigenbasis; without a constraint
            // correction the step vector itself is not needed
            double ss;
            if (chat.empty()) {
                ss = scons2 + eval_hat(st, alpha, nullptr, nullptr);
            } else {
                ss = scons2 + eval_hat(st, alpha, shat, nullptr);
                for (int i = 0; i < m; ++i) ss += 2.0 * shat[i] * chat[i];
            }
            *val = sqrt(ss > 0.0 ? ss : 0.0);
            *dval = 0.0;
            hat_stale = true;
            ht.mark = std::chrono::steady_clock::now();
            st->sweeps += bordered::g_sweeps - sw0;
            return SELLA_OK;
        }
        eval_hat(st, alpha, shat, dshat);
        hat_stale = false;
        ht.mark = std::chrono::steady_clock::now();
        st->sweeps += bordered::g_sweeps - sw0;
        if (eig_only) {
            double ss = scons2, sd = 0.0;
            for (int i = 0; i < m; ++i) {
                const double ci = chat.empty() ? 0.0 : chat[i];
                ss += shat[i] * (shat[i] + 2.0 * ci);
                sd += dshat[i] * (shat[i] + ci);
            }
            *val = sqrt(ss > 0.0 ? ss : 0.0);
            *dval = sd / (*val > 1e-12 ? *val : 1e-12);
            return SELLA_OK;
        }
        if (pinned) {
            memcpy(hin, shat, (size_t)m * sizeof(double));
            memcpy(hin + ldx, dshat, (size_t)m * sizeof(double));
            HIPCHK(s_memcpy(c, dx, hin, (size_t)(ldx + m) * sizeof(double), hipMemcpyHostToDevice, true));
        } else {
            SCHK(h2d_async(c, dx, shat, (size_t)m * sizeof(double)));
            SCHK(h2d_async(c, dx + ldx, dshat, (size_t)m * sizeof(double)));
        }
        if (st->panel) SCHK(panel_apply(dx, ldx, 2, dy, ldy));
        else SCHK(launch_gemv_rows(c, V->d, nfam, m, V->ld, dx, ldx, 2, dy, ldy, GemvEpi()));
        SELLA_LAUNCHB(c, rs_cons_kernel, rs_cons_vb, 1024, dim3(1), dim3(1024), 0, cons, nout, dy, dy + ldy, dscons, dw, dd1, dstot,
                           hres, dsel, nfam, dsfull, ddfull);
        HIPCHK(hipGetLastError());
        SCHK(stream_wait(c));
        ++st->rs_single;
        *val = hres[0];
        *dval = hres[1];
        return SELLA_OK;
    };
    // ---- a tree of bisection midpoints in one round trip (kernels above) ------------------------------------------
    // Candidates in heap order: node 1 = mid(lower, upper), node 2q = midpoint of the lower half of node q's bracket,
    // node 2q + 1 of its upper half; computed with the reference's own expression 0.5 * (lower + upper).
    constexpr int BATCH_LEVELS = 4, BATCH_NODES = (1 << BATCH_LEVELS) - 1;
    const bool can_batch = !eig_only && !newton_safe && c->opt.rs_batch && (st->panel || ((V->ld % 4 == 0) && m <= V->ld));
    double* dbatch = nullptr;                  // lam | ghat | d1hat | X (16 x ldb) | inv | Y (16 x ldy)
    double* batch_Y = nullptr;                 // Y of the last batch
    int ldb = 0;
    bool batch_ready = false;
    double cand[BATCH_NODES + 1], cval[BATCH_NODES + 1];
    int nbatch = 0;
    // device layout: lam | ghat | d1hat | X (16 x ldb, zero) | inv (nout ints, padded to ldy / 2 + 8 doubles) | Y (16 x ldy)
    const size_t inv_doubles = (size_t)ldy / 2 + 8;
    auto batch_setup = [&]() -> int {
        ldb = st->panel ? round_up(m, 8) : V->ld;
        const size_t nup = (size_t)3 * ldx + (size_t)16 * ldb + inv_doubles;      // what comes from the host, in front of Y
        const size_t need = nup + (size_t)16 * ldy + 64;
        SCHK(scratch_get(c, SCR_STEP2, need * sizeof(double), &dbatch));
        // ONE transfer for the spectrum, the zero-filled candidate matrix and the index map of the view (round 6: they
        // were an upload, a fill and an upload — three small operations and their gaps in front of the first round)
        void* slot = nullptr;
        if (h2d_begin(c, nup * sizeof(double), &slot) == SELLA_OK) {           // (zeroed by h2d_begin)
            double* pk = static_cast<double*>(slot);
            std::copy(st->lam.begin(), st->lam.end(), pk);
            std::copy(st->ghat.begin(), st->ghat.end(), pk + ldx);
            if ((int)st->d1hat.size() == m) std::copy(st->d1hat.begin(), st->d1hat.end(), pk + 2 * (size_t)ldx);
            if (sel) {
                int* inv = reinterpret_cast<int*>(pk + 3 * (size_t)ldx + (size_t)16 * ldb);
                for (int i = 0; i < nout; ++i) inv[i] = -1;
                for (int i = 0; i < nfam; ++i) inv[sel[i]] = i;
            }
            SCHK(h2d_end(c, dbatch, slot, nup * sizeof(double)));
        } else {                                                                // (beyond half the ring: the separate operations)
            std::vector<double> pack((size_t)3 * ldx, 0.0);
            std::copy(st->lam.begin(), st->lam.end(), pack.begin());
            std::copy(st->ghat.begin(), st->ghat.end(), pack.begin() + ldx);
            if ((int)st->d1hat.size() == m) std::copy(st->d1hat.begin(), st->d1hat.end(), pack.begin() + 2 * (size_t)ldx);
            SCHK(h2d_async(c, dbatch, pack.data(), pack.size() * sizeof(double)));
            HIPCHK(s_memset0(c, dbatch + 3 * (size_t)ldx, (size_t)16 * ldb * sizeof(double)));
            if (sel) {
                std::vector<int> inv(nout, -1);
                for (int i = 0; i < nfam; ++i) inv[sel[i]] = i;
                SCHK(h2d_async(c, dbatch + 3 * (size_t)ldx + (size_t)16 * ldb, inv.data(), (size_t)nout * sizeof(int)));
            }
        }
        batch_ready = true;
        return SELLA_OK;
    };
    // given: cand[1 .. BATCH_NODES] are set by the caller; otherwise the tree of midpoints of (lower, upper)
    auto batch_evaluate = [&](double lower, double upper, bool given = false) -> int {
        cohort_barrier(c, 1 + nbatch, 0);                     // members of a cohort take the rounds of their searches together
        if (!batch_ready) SCHK(batch_setup());
        double lo[BATCH_NODES + 1], hi[BATCH_NODES + 1];
        lo[1] = lower; hi[1] = upper;
        for (int q = 1; q <= BATCH_NODES && !given; ++q) {
            cand[q] = 0.5 * (lo[q] + hi[q]);
            if (2 * q + 1 <= BATCH_NODES) {
                lo[2 * q] = lo[q]; hi[2 * q] = cand[q];
                lo[2 * q + 1] = cand[q]; hi[2 * q + 1] = hi[q];
            }
        }
        RsBatchArgs ba;
        ba.kind = st->kind; ba.m = m; ba.order = st->order; ba.ncand = BATCH_NODES; ba.ldx = ldb;
        for (int q = 0; q < BATCH_NODES; ++q) ba.alpha[q] = cand[q + 1];
        ba.alpha[15] = 0.0;
        ba.lam = dbatch; ba.ghat = dbatch + ldx; ba.d1hat = dbatch + 2 * (size_t)ldx;
        ba.X = dbatch + 3 * (size_t)ldx;
        double* dY = ba.X + (size_t)16 * ldb + inv_doubles;
        batch_Y = dY;
        const int* dinv = sel ? reinterpret_cast<const int*>(ba.X + (size_t)16 * ldb) : nullptr;
        SELLA_LAUNCHB(c, rs_batch_kernel, rs_batch_vb, 256, dim3(BATCH_NODES), dim3(256), 0, ba);
        HIPCHK(hipGetLastError());
        if (st->panel) SCHK(panel_apply(ba.X, ldb, BATCH_NODES, dY, ldy));
        else SCHK(launch_panel16(c, V->d, nfam, m, V->ld, ba.X, BATCH_NODES, dY, ldy));
        SELLA_LAUNCHB(c, rs_measure_kernel, rs_measure_vb, 256, dim3(BATCH_NODES), dim3(256), 0, cons, nout, dY, ldy, dscons, dw, dd1,
                           dinv, hres + 2);
        HIPCHK(hipGetLastError());
        // (the values land in pinned host memory, written by the kernel itself: the wait polls a sequence word stored
        //  behind it instead of sleeping in the runtime — nothing else is pending on the stream in a round)
        if (c->opt.rs_poll && c->d2h_pending.empty()) { SCHK(poll_mark(c)); SCHK(poll_wait(c)); }
        else SCHK(stream_wait(c));
        for (int q = 1; q <= BATCH_NODES; ++q) cval[q] = hres[1 + q];
        ++nbatch;
        return SELLA_OK;
    };
    // ---- restricted_step.py:78-120 --------------------------------------------------------------------------------
    double alpha = alpha0, val = 0.0, dval = 0.0;
    bool inside = false;
    bool stale = false;                        // the device no longer holds stot of the final alpha (batched evaluations)
    bool fast_done = false;
    int batch_row = -1;                        // >= 0: the step of the final alpha is row `batch_row` of the LAST batch's output
    const bool fast_ok = st->fast_search && can_batch && alphamin == 0.0 && std::isfinite(alphamax) && slope > 0.0;
    // The caller expects the step on the trust boundary (the previous one was: sella_opt_step's hint) — then the start value
    // alpha0 rides in the first batch instead of a round trip of its own, and the final step is read from the batch that
    // produced it whenever it is the last one.  Same trial values, same root; two round trips fewer on a boundary step.
    const bool merged = fast_ok && st->boundary_hint && !scons && alpha0 > 0.0 && c->opt.rs_batch_result;
    if (!merged) {
        SCHK(evaluate(alpha, &val, &dval));
        inside = val < delta;
    }
    if (merged || (!inside && fast_ok && fabs(val - delta) > tol)) {
        // ---- the one-call optimizer step: same root, fewer round trips ------------------------------------------------
        // The measure grows with alpha and is smooth between the points where the largest component changes hands, so
        // the 15 candidates of a round trip go where the secant through the bracket ends puts the root, at offsets of
        // 1e-1 ... 1e-7 of the bracket on both sides: the bracket shrinks by the accuracy of the secant (at least
        // tenfold) per round instead of sixteenfold, and a handful of rounds replace the dozen of the bisection tree.
        // The root is the reference's — the alpha where the measure crosses the radius, to the last bits the measure
        // itself resolves — reached through other trial points.
        double lower = 0.0, upper = alpha, flo = -delta, fhi = merged ? 0.0 : val - delta;
        bool first = true, ok = false, have_top = !merged;
        int round_no = 0, src_lo_round = -1, src_lo_row = -1, src_hi_round = -1, src_hi_row = -1, hit_row = -1;
        // a third evaluated point next to the bracket (the nearest one outside it, from the last round): with it the next
        // trial value comes from inverse QUADRATIC interpolation instead of the secant — the measure is smooth between the
        // points where its largest component changes hands, and the bracket then shrinks by the cube of its width per round
        bool have3 = false;
        double x3 = 0.0, f3 = 0.0;
        for (int round = 0; round < 48; ++round) {
            const double width = upper - lower;
            if (have_top && (!(width > 0.0) || nextafter(nextafter(lower, upper), upper) >= upper)) { ok = true; break; }
            int nc = 0;
            double pts[BATCH_NODES];
            const int nlog = have_top ? BATCH_NODES : BATCH_NODES - 1;
            const double hint = st->alpha_hint;
            if (first && c->opt.rs_hint && hint > lower && hint < upper && lower == 0.0) {
                // The previous root search of this saddle search ended at `hint`: consecutive steps of a search sit on the same trust
                // boundary with slowly changing curvature, so the first round spends half of its trial values around that
                // value (offsets of 10^-1 ... 10^-4 to both sides, clipped into the bracket) and the other half on the decades
                // below the start value as before.  Other trial points, the same root (restricted_step.py:78-120); a hint that
                // is off costs nothing but the round it would have taken anyway.
                pts[nc++] = hint;
                for (int k = 1; k <= 4 && nc + 2 <= nlog; ++k) {
                    const double off = hint * pow(10.0, -(double)k);
                    pts[nc++] = hint - off;
                    pts[nc++] = hint + off;
                }
                for (int k = 1; nc < nlog; ++k) pts[nc++] = lower + width * pow(10.0, -1.0 * k);
            } else if (first) {
                for (int k = 1; k <= nlog; ++k) pts[nc++] = lower + width * pow(10.0, -0.5 * k);
            } else {
                double rs = lower - flo * width / (fhi - flo);
                if (c->opt.rs_hint && have3 && f3 != flo && f3 != fhi && flo != fhi) {
                    // x as a quadratic in f through (flo, lower), (fhi, upper), (f3, x3), evaluated at f = 0
                    const double q = lower * (fhi * f3) / ((flo - fhi) * (flo - f3)) + upper * (flo * f3) / ((fhi - flo) * (fhi - f3)) +
                                     x3 * (flo * fhi) / ((f3 - flo) * (f3 - fhi));
                    if (q > lower && q < upper) rs = q;
                }
                if (!(rs > lower && rs < upper)) rs = 0.5 * (lower + upper);
                pts[nc++] = rs;
                // offsets to both sides of the estimate: decades 10^-1 ... 10^-7 of the bracket, or — with the better
                // estimate — every other decade down to 10^-13 (an estimate that is good to 10^-9 of the bracket is
                // then bracketed that tightly instead of at 10^-7)
                const double step10 = (c->opt.rs_hint && have3) ? 2.0 : 1.0;
                for (int k = 1; k <= 7; ++k) {
                    const double off = width * pow(10.0, -step10 * k + (step10 - 1.0));
                    pts[nc++] = rs - off;
                    pts[nc++] = rs + off;
                }
            }
            std::sort(pts, pts + nc);
            // keep the candidates strictly inside the bracket and distinct; fill up with an even subdivision
            int kept = 0;
            for (int k = 0; k < nc; ++k)
                if (pts[k] > lower && pts[k] < upper && (kept == 0 || pts[k] > pts[kept - 1])) pts[kept++] = pts[k];
            for (int k = 1; kept < nlog; ++k) {
                const double p = lower + width * k / (BATCH_NODES + 1.0);
                if (k > 4 * BATCH_NODES) break;
                bool dup = !(p > lower && p < upper);
                for (int q = 0; q < kept && !dup; ++q) dup = pts[q] == p;
                if (!dup) pts[kept++] = p;
            }
            std::sort(pts, pts + kept);
            if (!have_top) pts[kept++] = upper;                    // the start value itself: the largest candidate
            for (int q = 0; q < BATCH_NODES; ++q) cand[q + 1] = q < kept ? pts[q] : pts[kept - 1];
            SCHK(batch_evaluate(lower, upper, true));
            ++round_no;
            stale = true;
            ntrial += kept;
            first = false;
            int nscan = kept;
            if (!have_top) {
                // what the evaluation at alpha0 would have said (restricted_step.py:78-84)
                have_top = true;
                nscan = kept - 1;
                val = cval[kept];
                inside = val < delta;
                if (inside || fabs(val - delta) <= tol) { hit_row = kept - 1; ok = true; fast_done = true; break; }
                fhi = val - delta;
                src_hi_round = round_no; src_hi_row = kept - 1;
            }
            bool hit = false;
            const double lower_in = lower, upper_in = upper, flo_in = flo, fhi_in = fhi;
            int q_lo = -1, q_hi = -1;                              // rows of this round that became the bracket ends
            for (int q = 0; q < nscan; ++q) {
                const double e = cval[q + 1] - delta;
                if (fabs(e) <= tol) { alpha = pts[q]; val = cval[q + 1]; hit = true; hit_row = q; break; }
                if (e > 0.0) { if (pts[q] < upper) { upper = pts[q]; fhi = e; src_hi_round = round_no; src_hi_row = q; q_hi = q; } }
                else if (pts[q] > lower) { lower = pts[q]; flo = e; src_lo_round = round_no; src_lo_row = q; q_lo = q; }
            }
            if (hit) { ok = true; fast_done = true; break; }
            {
                // the nearest evaluated neighbours outside the new bracket: the row below the new lower end (or the old lower
                // end), the row above the new upper end (or the old upper end); the one with the smaller residual is kept
                double xa = 0.0, fa = 0.0, xb = 0.0, fb = 0.0;
                bool ha = false, hb = false;
                if (q_lo > 0) { xa = pts[q_lo - 1]; fa = cval[q_lo] - delta; ha = true; }
                else if (q_lo == 0 && lower_in > 0.0) { xa = lower_in; fa = flo_in; ha = true; }
                if (q_hi >= 0 && q_hi + 1 < nscan) { xb = pts[q_hi + 1]; fb = cval[q_hi + 2] - delta; hb = true; }
                else if (q_hi >= 0) { xb = upper_in; fb = fhi_in; hb = fhi_in != 0.0; }
                have3 = ha || hb;
                if (ha && (!hb || fabs(fa) <= fabs(fb))) { x3 = xa; f3 = fa; }
                else if (hb) { x3 = xb; f3 = fb; }
            }
        }
        if (!ok) { set_error("Restricted step failed to converge!"); return SELLA_E_NOCONV; }
        if (!fast_done) {
            // bracket collapsed: the end nearer to the radius
            if (lower > 0.0 && fabs(flo) < fabs(fhi)) { alpha = lower; val = flo + delta; if (src_lo_round == round_no) batch_row = src_lo_row; }
            else { alpha = upper; val = fhi + delta; if (src_hi_round == round_no) batch_row = src_hi_row; }
            fast_done = true;
        } else {
            batch_row = hit_row;                                   // found in the batch just evaluated
        }
        st->alpha_found = alpha;                                   // where this search ended: the next one starts looking there
        if (!merged) batch_row = -1;                               // (the step from the batch output: with the merged flow only)
        if (batch_row < 0) {
            const int keep = ntrial;
            const double vkeep = val;
            SCHK(evaluate(alpha, &val, &dval));                    // s (and stot on the device) of the final alpha
            ntrial = keep;
            val = vkeep;
            stale = false;
        }
    }
    if (!inside && !fast_done) {
        double err = val - delta, lower = alphamin, upper = alphamax;
        bool converged = false;
        int node = 0;                          // position in the current candidate tree (0: none)
        for (int niter = 0; niter < maxiter; ++niter) {
            if (fabs(err) <= tol) { converged = true; break; }
            if (nextafter(lower, upper) >= upper) { converged = true; break; }
            if (err * slope > 0.0) upper = alpha; else lower = alpha;
            if (can_batch && niter > 4 && std::isfinite(lower) && std::isfinite(upper)) {
                // pure bisection from here on: take this trial alpha from the tree, refilling it when exhausted
                if (node == 0 || node > BATCH_NODES) {
                    SCHK(batch_evaluate(lower, upper));
                    node = 1;
                }
                alpha = cand[node];                                  // == 0.5 * (lower + upper)
                val = cval[node];
                err = val - delta;
                if (alphas && ntrial <= maxiter) alphas[ntrial] = alpha;
                ++ntrial;
                stale = true;
                // the next bracket update decides which child follows: upper = alpha -> lower half
                node = 2 * node + ((err * slope > 0.0) ? 0 : 1);
                continue;
            }
            const double newton = alpha - err / dval;
            const bool bisect = (newton != newton) || newton <= lower || newton >= upper || (niter > 4 && !newton_safe);
            if (bisect) {
                const double mid = 0.5 * (lower + upper);
                if (std::isinf(mid)) alpha = alpha + std::max(1.0, 0.5 * alpha) * (mid > 0.0 ? 1.0 : -1.0);
                else alpha = mid;
            } else {
                alpha = newton;
            }
            SCHK(evaluate(alpha, &val, &dval, niter >= 4 && !newton_safe));
            err = val - delta;
        }
        if (!converged) {
            set_error("Restricted step failed to converge!");
            return SELLA_E_NOCONV;
        }
        if (stale || hat_stale) {
            // s (and stot on the device) of the final alpha; not a trial of the schedule
            const int keep = ntrial;
            const double vkeep = val;
            SCHK(evaluate(alpha, &val, &dval));
            ntrial = keep;
            val = vkeep;
        }
    }
    // ---- the step at the final alpha ------------------------------------------------------------------------------
    cohort_barrier(c, 0xffffff, 0);                           // (a cohort closes up behind searches of different length)
    if (eig_only && st->panel) {
        // coefficients up, the step in one launch over the panel's own rows
        if (pinned) {
            memcpy(hin, shat, (size_t)m * sizeof(double));
            HIPCHK(s_memcpy(c, dx, hin, (size_t)m * sizeof(double), hipMemcpyHostToDevice, true));
        } else {
            SCHK(h2d_async(c, dx, shat, (size_t)m * sizeof(double)));
        }
        SCHK(panel_apply(dx, ldx, 1, dy, ldy));
    } else if (eig_only) {
        if (pinned) {
            memcpy(hin, shat, (size_t)m * sizeof(double));
            HIPCHK(s_memcpy(c, dx, hin, (size_t)m * sizeof(double), hipMemcpyHostToDevice, true));
        } else {
            SCHK(h2d_async(c, dx, shat, (size_t)m * sizeof(double)));
        }
        SCHK(launch_gemv_rows(c, V->d, nfam, m, V->ld, dx, ldx, 1, dy, ldy, GemvEpi()));
    }
    if (eig_only) {
        if (sel) {
            std::vector<double> sp(nfam);
            SCHK(d2h_async(c, sp.data(), dy, (size_t)nfam * sizeof(double)));
            SCHK(stream_wait(c));
            for (int i = 0; i < nout; ++i) s_out[i] = 0.0;
            for (int i = 0; i < nfam; ++i) s_out[sel[i]] = sp[i];
        } else {
            SCHK(d2h_async(c, s_out, dy, (size_t)nout * sizeof(double)));
            SCHK(stream_wait(c));
        }
        if (scons)
            for (int i = 0; i < nout; ++i) s_out[i] += scons[i];
    } else if (batch_row >= 0) {
        // the step of the final alpha as the last batch computed it (family space; pinned coordinates stay zero)
        std::vector<double> sp(nfam);
        SCHK(d2h_async(c, sp.data(), batch_Y + (size_t)batch_row * ldy, (size_t)nfam * sizeof(double)));
        SCHK(stream_wait(c));
        if (sel) {
            for (int i = 0; i < nout; ++i) s_out[i] = 0.0;
            for (int i = 0; i < nfam; ++i) s_out[sel[i]] = sp[i];
        } else {
            for (int i = 0; i < nout; ++i) s_out[i] = sp[i];
        }
    } else {
        SCHK(d2h_async(c, s_out, dstot, (size_t)nout * sizeof(double)));
        SCHK(stream_wait(c));
    }
    *val_out = inside ? val : delta;
    if (nalpha) *nalpha = ntrial;
    st->calls += ntrial;
    ++st->rs_calls;
    st->rs_rounds += nbatch;
    return SELLA_OK;
}

extern "C" int sella_stepper_destroy(sella_stepper* st) {
    if (!st) return SELLA_OK;
    if (st->calls && getenv("SELLA_DEBUG_TIMING"))
        fprintf(stderr, "stepper m=%d nout=%d: %ld get_s calls, %.1f us host solve (%.1f sweeps) + %.1f us device round trip per call; %ld root searches, %ld batched and %ld single-alpha round trips\n",
                st->m, st->nout, st->calls, 1e6 * st->t_host / st->calls, (double)st->sweeps / st->calls,
                1e6 * st->t_dev / st->calls, st->rs_calls, st->rs_rounds, st->rs_single);
    if (st->ownV != SELLA_NO_MAT) sella_mat_free(st->c, st->ownV);
    if (st->ownVt != SELLA_NO_MAT) sella_mat_free(st->c, st->ownVt);
    delete st;
    return SELLA_OK;
}

// Step family on a STRUCTURED eigendecomposition (r explicit eigenpairs mu / rows of Wt, eigenvalue lam0 on the
// complement of their span; eigh.hip lr_lowrank_update).  Every family of stepper.py gives a mode whose gradient
// component vanishes a zero step, so the step lives in span{W, g_perp}, g_perp = g - W^T W g: the n - r eigenvectors of
// the cluster are REPLACED by one, g_perp / |g_perp| (weight |g_perp|), plus min(order, n - r - 1) weightless copies
// so that "the `order` lowest modes" (P-RFO's partition, RFO's root index) still counts the cluster's multiplicity.
// The compressed problem — r + 1 + copies modes of length n — is handed to the dense machinery unchanged
// (sella_stepper_create on rectangular mode matrices), so sella_stepper_get_s / sella_restricted_step work as is.
extern "C" int sella_stepper_create_lr(sella_ctx* c, int kind, sella_mat hWt, int r, const double* mu, double lam0,
                                       const double* g, int n, int order, sella_stepper** out) {
    if (!c || !out || !g || n <= 0 || r < 0 || r > n || order < 0 || order > n || (r > 0 && !mu)) {
        set_error("stepper (structured): invalid arguments");
        return SELLA_E_INVALID;
    }
    Mat* Wm = mat_get(c, hWt);
    if (r > 0 && (!Wm || Wm->cols != n || Wm->rows < r)) { set_error("stepper (structured): bad eigenvector handle"); return SELLA_E_INVALID; }
    const int ld = round_up(n, 8);
    const int ncl = n - r;                                   // multiplicity of lam0
    const int ncopy = ncl > 0 ? std::min(order, ncl - 1) : 0;
    const int m = r + (ncl > 0 ? 1 : 0) + ncopy;
    // source panel: [W rows | g_perp normalised | zero row]
    double* src;
    SCHK(scratch_get(c, SCR_STEP2, (size_t)(r + 2) * ld * sizeof(double), &src));
    if (r > 0) {
        Wm = mat_get(c, hWt);
        SCHK(launch_axpby2d(c, r, n, 1.0, Wm->d, Wm->ld, 0.0, nullptr, 0, src, ld));
    }
    double* gp = src + (size_t)r * ld;
    HIPCHK(s_memset0(c, gp, (size_t)2 * ld * sizeof(double)));
    // ONE round trip: a = W g (weights of the explicit modes), then the normalised component of g outside span(W) by two
    // Gram-Schmidt sweeps whose norms come back with a — |g_perp| = |g| n1 n2 is the weight of the cluster's mode
    bool have_perp = false;
    std::vector<double> aw(r + 1, 0.0);
    double gperp = 0.0;
    {
        SCHK(h2d_async(c, gp, g, (size_t)n * sizeof(double)));
        double* da = c->dscal + DS_CVEC + 8192;                       // r <= 8192 explicit pairs
        if (r > 8192) { set_error("stepper (structured): too many explicit eigenpairs"); return SELLA_E_UNSUPPORTED; }
        if (r > 0) {
            SCHK(launch_gemv_rows(c, src, r, n, ld, gp, ld, 1, da, std::max(r, 1), GemvEpi()));
            HIPCHK(s_memcpy(c, c->hscal + DS_CVEC + 8192, da, (size_t)r * sizeof(double), hipMemcpyDeviceToHost, true));
        }
        if (ncl > 0) SCHK(gs_project_twice(c, src, ld, r, gp, n));
        SCHK(sync_scalars(c, 8, 3));
        for (int i = 0; i < r; ++i) aw[i] = c->hscal[DS_CVEC + 8192 + i];
        if (ncl > 0) {
            const double n0sq = c->hscal[8], n1 = sqrt(c->hscal[9]), n2 = sqrt(c->hscal[10]);
            have_perp = (n0sq > 0.0) && (n1 == n1) && !(n1 < 1e-13) && (n2 == n2) && !(n2 < 1e-13);
            if (have_perp && fabs(1.0 - n2) > 1e-15) {
                int kept = 0;                                           // rare: more sweeps, synchronously
                SCHK(gs_orthonormalise(c, src, ld, r, gp, n, 1e-15, 1e-13, 100, &kept, nullptr));
                have_perp = kept != 0;
                if (have_perp) {
                    // weight of the finished row: gp . g
                    double* gd;
                    SCHK(scratch_get(c, SCR_STEP1, (size_t)2 * ld * sizeof(double), &gd));
                    SCHK(h2d_async(c, gd, g, (size_t)n * sizeof(double)));
                    SCHK(launch_gemv_rows(c, gp, 1, n, ld, gd, ld, 1, c->dscal + DS_CVEC, 1, GemvEpi()));
                    SCHK(read_scalars(c, DS_CVEC, 1));
                    gperp = c->hscal[DS_CVEC];
                }
            } else if (have_perp) {
                gperp = sqrt(n0sq) * n1 * n2;
            }
            if (!have_perp) HIPCHK(s_memset0(c, gp, (size_t)ld * sizeof(double)));
        }
    }
    // modes in ascending order of eigenvalue; within the cluster the weighted mode first
    std::vector<double> ev(m);
    std::vector<int> idx(m);
    {
        int i = 0, p = 0;
        while (i < r && mu[i] < lam0) { ev[p] = mu[i]; idx[p++] = i++; }
        if (ncl > 0) {
            ev[p] = lam0; idx[p++] = r;
            for (int q = 0; q < ncopy; ++q) { ev[p] = lam0; idx[p++] = r + 1; }
        }
        while (i < r) { ev[p] = mu[i]; idx[p++] = i++; }
    }
    std::vector<double> gh(m);
    for (int p = 0; p < m; ++p) gh[p] = idx[p] < r ? aw[idx[p]] : (idx[p] == r ? gperp : 0.0);
    return stepper_from_panel(c, kind, src, ld, idx.data(), m, n, ev.data(), gh.data(), order, out);
}

// Step family in panel form: modes = rows idx[i] of `src` (not copied: the panel must outlive the family), eigenvalues
// ev, gradient components gh.  For sella_restricted_step with the trust-region measure only (see sella_stepper).
int sella::stepper_on_panel(sella_ctx* c, int kind, const double* src, int ld, const int* idx, int m, int n, const double* ev,
                            const double* gh, int order, sella_stepper** out) {
    if (!c || !out || !src || !idx || !ev || !gh || m <= 0 || order < 0 || order > m || kind < SELLA_STEP_QN ||
        kind > SELLA_STEP_PRFO) {
        set_error("stepper: invalid arguments");
        return SELLA_E_INVALID;
    }
    sella_stepper* st = new sella_stepper();
    st->c = c; st->kind = kind; st->m = m; st->order = order; st->nout = n;
    st->lam.assign(ev, ev + m);
    st->ghat.assign(gh, gh + m);
    st->panel = src; st->panel_ld = ld;
    st->pidx.assign(idx, idx + m);
    st->pscale.assign(m, 1.0);
    *out = st;
    return SELLA_OK;
}

void sella::stepper_panel_scale(sella_stepper* st, int mode, double factor) {
    if (st && mode >= 0 && mode < (int)st->pscale.size()) st->pscale[mode] = factor;
}

void sella::stepper_set_alpha_hint(sella_stepper* st, double hint) { if (st) st->alpha_hint = hint; }
double sella::stepper_alpha_found(const sella_stepper* st) { return st ? st->alpha_found : 0.0; }

void sella::stepper_set_fast_search(sella_stepper* st, bool on, bool boundary_hint) {
    if (st) { st->fast_search = on; st->boundary_hint = boundary_hint; }
}

// Step family whose m modes are rows idx[0..m) of a device panel (ascending eigenvalues ev, gradient components gh):
// the rows are gathered into mode matrices the stepper owns.
int sella::stepper_from_panel(sella_ctx* c, int kind, const double* src, int ld, const int* idx, int m, int n,
                              const double* ev, const double* gh, int order, sella_stepper** out) {
    sella_mat hVt = SELLA_NO_MAT, hV = SELLA_NO_MAT;
    SCHK(mat_new(c, m, n, &hVt));
    int st = mat_new(c, n, m, &hV);
    if (st != SELLA_OK) { sella_mat_free(c, hVt); return st; }
    auto bail = [&](int code) { sella_mat_free(c, hVt); sella_mat_free(c, hV); return code; };
    int* didx;
    {
        double* ib;
        st = scratch_get(c, SCR_STEP0, (size_t)(2 * std::max(ld, round_up(m, 8)) + m / 2 + 8) * sizeof(double), &ib);
        if (st != SELLA_OK) return bail(st);
        didx = reinterpret_cast<int*>(ib + 2 * (size_t)std::max(ld, round_up(m, 8)));
    }
    st = h2d_async(c, didx, idx, (size_t)m * sizeof(int));
    if (st != SELLA_OK) return bail(st);
    Mat* Vt = mat_get(c, hVt);
    st = launch_gather_rows(c, src, ld, didx, m, n, Vt->d, Vt->ld);
    if (st != SELLA_OK) return bail(st);
    Mat* V = mat_get(c, hV);
    Vt = mat_get(c, hVt);
    st = launch_transpose(c, Vt->d, m, n, Vt->ld, V->d, V->ld);
    if (st != SELLA_OK) return bail(st);
    st = stepper_make(c, kind, hV, hVt, ev, nullptr, gh, m, order, out);
    if (st != SELLA_OK) return bail(st);
    (*out)->ownV = hV;
    (*out)->ownVt = hVt;
    return SELLA_OK;
}
