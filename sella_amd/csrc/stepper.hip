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
#include <chrono>
#include <cstdlib>

#include <algorithm>

struct sella_stepper {
    sella_ctx* c = nullptr;
    int kind = 0, m = 0, order = 0, nout = 0;
    sella_mat V = SELLA_NO_MAT;      // m x m eigenvectors (columns)   [not owned]
    sella_mat VU = SELLA_NO_MAT;     // nout x m : U V when a projection U (nout x m) was given [owned]
    std::vector<double> lam, ghat;
    double t_host = 0.0, t_dev = 0.0;     // SELLA_DEBUG_TIMING: seconds in the secular solves / in the device round trip
    long calls = 0, sweeps = 0;
};

namespace sella {
namespace {

typedef std::vector<double> vec;

// Root number j (ascending, 0..mm) of f(mu) = mu + sum b_i^2 / (D_i - mu), D ascending.
// Returned as (origin, tau): mu = D_origin + tau with the origin the closer pole
// (origin = -1: mu = tau, used for the two exterior roots far from every pole).
long g_sweeps = 0;      // SELLA_DEBUG_TIMING statistics only

void bordered_root(int mm, const double* D, const double* b, int j, int* origin, double* tau) {
    double bb = 0.0;
    for (int i = 0; i < mm; ++i) bb += b[i] * b[i];
    // value, noise scale and the derivative split at pole index j (left part: poles i < j, right part: i >= j)
    struct Ev { double f, noise, dl, dr; };
    const int ext = (j == 0) ? 0 : (j == mm ? mm - 1 : -1);      // nearest pole of an exterior root
    auto eval = [&](double shift, double t) {
        Ev e;
        double s = 0.0, sa = 0.0, dl = 0.0, dr = 0.0;
        for (int i = 0; i < j; ++i) {                 // poles left of the root (two plain loops: both vectorise)
            const double r = 1.0 / ((D[i] - shift) - t);
            const double q = b[i] * b[i] * r;
            s += q;
            sa += fabs(q);
            dl += q * r;
        }
        for (int i = j; i < mm; ++i) {                // poles right of the root
            const double r = 1.0 / ((D[i] - shift) - t);
            const double q = b[i] * b[i] * r;
            s += q;
            sa += fabs(q);
            dr += q * r;
        }
        e.f = (shift + t) + s;
        e.noise = fabs(shift + t) + sa;
        e.dl = dl;
        e.dr = dr;
        return e;
    };
    double shift, lo, hi, t;
    int org;
    if (mm == 0) { *origin = -1; *tau = 0.0; return; }
    if (j == 0 || j == mm) {
        // Exterior root.  Brackets from two one-pole problems mu + c / (d - mu) = 0 with d the nearest pole D_e:
        // all weight on that pole (c = |b|^2) overshoots the root, only that pole's own weight (c = b_e^2) falls
        // short of it — every term of the sum has the same sign on this side of the spectrum.
        const int e = (j == 0) ? 0 : mm - 1;
        const double sg = (j == 0) ? -1.0 : 1.0;
        org = e;
        shift = D[e];
        const double far = 0.5 * (-shift + sg * sqrt(shift * shift + 4.0 * bb));          // t of the overshooting model
        const double near = 0.5 * (-shift + sg * sqrt(shift * shift + 4.0 * b[e] * b[e]));
        if (j == 0) { lo = far; hi = std::min(near, 0.0); }
        else { lo = std::max(near, 0.0); hi = far; }
        if (!(hi > lo)) { *origin = org; *tau = 0.5 * (lo + hi); return; }                 // b = 0: mu = min/max(D_e, 0)
        t = far;
        if (t == 0.0) t = 0.5 * (lo + hi);
    } else {
        const double delta = D[j] - D[j - 1];
        if (delta <= 0.0) { *origin = j; *tau = 0.0; return; }      // coincident poles: mu = D_j
        const double fm = eval(D[j - 1], 0.5 * delta).f;
        if (fm >= 0.0) { org = j - 1; shift = D[j - 1]; lo = 0.0; hi = 0.5 * delta; }
        else { org = j; shift = D[j]; lo = -0.5 * delta; hi = 0.0; }
        t = 0.5 * (lo + hi);
    }
    // f is increasing between poles: f(lo) <= 0 <= f(hi) (pole ends are never evaluated).  Interior roots: as in
    // the eigensolver's secular equation (secular.h), the two poles next to the root are kept exact and the rest of
    // the sum is frozen at value and slope ("middle way" rational model).  Exterior roots: the nearest pole is
    // kept exact and the rest of the sum is replaced by the one-pole function that matches its value and slope (all terms have
    // the same sign there) next to the exact nearest pole — with the nearest pole alone these roots took 30-60
    // sweeps at the sizes of a slab search, now 3-6.  Bracket + bisection as the safeguard, stop at |f| below its rounding noise.
    const double EPS = 2.220446049250313e-16;
    for (int it = 0; it < 200; ++it) {
        const Ev e = eval(shift, t);
        ++g_sweeps;
        const double fv = e.f;
        if (!(fabs(fv) > 8.0 * EPS * e.noise)) break;
        if (fv < 0.0) lo = t; else hi = t;
        const double df = 1.0 + e.dl + e.dr;
        double eta;
        if (j > 0 && j < mm) {
            const double D1 = (D[j - 1] - shift) - t, D2 = (D[j] - shift) - t;     // < 0 < 
            const double c_ = fv - D1 * e.dl - D2 * e.dr;
            const double a_ = (D1 + D2) * fv - D1 * D2 * (e.dl + e.dr);
            const double b_ = D1 * D2 * fv;
            if (c_ == 0.0) eta = (a_ != 0.0) ? b_ / a_ : -fv / df;
            else {
                const double disc = sqrt(fabs(a_ * a_ - 4.0 * b_ * c_));
                eta = (a_ <= 0.0) ? (a_ - disc) / (2.0 * c_) : 2.0 * b_ / (a_ + disc);
            }
        } else {
            // model: (mu + eta) + a1 / (p1 - eta) + a2 / (p2 - eta) = 0 with the nearest pole exact
            // (a1 = b_e^2, p1 = D_e - mu) and the REST of the sum R replaced by the one-pole function that
            // matches R and R' at the current point (p2 = R / R', a2 = R p2).  Solved for eta by a safeguarded
            // scalar Newton iteration inside the bracket — O(1) work per sweep.
            const double mu = shift + t;
            const double p1 = (D[ext] - shift) - t, a1 = b[ext] * b[ext];
            const double q1 = a1 / p1;
            const double R = (fv - mu) - q1, Rp = (e.dl + e.dr) - q1 / p1;
            double a2 = 0.0, p2 = 1.0;
            if (Rp > 0.0 && R != 0.0) { p2 = R / Rp; a2 = R * p2; }
            double elo = lo - t, ehi = hi - t, x = 0.0, Fx = fv;
            eta = -fv / df;
            for (int in = 0; in < 40; ++in) {
                if (Fx < 0.0) elo = x; else ehi = x;
                const double r1 = 1.0 / (p1 - x), r2 = 1.0 / (p2 - x);
                const double dF = 1.0 + a1 * r1 * r1 + a2 * r2 * r2;
                double xn = x - Fx / dF;
                if (!(xn > elo && xn < ehi)) xn = 0.5 * (elo + ehi);
                if (xn == x) break;
                x = xn;
                Fx = (mu + x) + a1 / (p1 - x) + a2 / (p2 - x);
                if (fabs(Fx) <= 4.0 * EPS * (fabs(mu + x) + fabs(a1 / (p1 - x)) + fabs(a2 / (p2 - x)))) break;
            }
            if (x != 0.0) eta = x;
        }
        if (!(fv * eta < 0.0)) eta = -fv / df;
        double tn = t + eta;
        if (!(tn > lo && tn < hi)) tn = 0.5 * (lo + hi);
        if (tn == lo || tn == hi || tn == t) { t = tn; break; }
        t = tn;
        if (hi - lo <= EPS * std::max(fabs(lo), fabs(hi))) break;
    }
    *origin = org;
    *tau = t;
}

// RFO step in the eigenbasis for a block (lam, ghat) of size mm, eigenpair index `o` of the
// augmented matrix (stepper.py:128-157).  Outputs shat, dshat (mm).
void rfo_block(int mm, const double* lam, const double* ghat, int o, double alpha, double* shat,
               double* dshat) {
    if (mm == 0) return;
    vec D(mm), b(mm), vh(mm + 1), c1(mm + 1), xp(mm + 1);
    for (int i = 0; i < mm; ++i) { D[i] = alpha * alpha * lam[i]; b[i] = alpha * ghat[i]; }
    int org;
    double tau;
    bordered_root(mm, D.data(), b.data(), o, &org, &tau);
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
        int ip = 0;
        for (int i = 0; i < mm; ++i) if (tau - (D[i] - shift) == 0.0) { ip = i; break; }
        for (int i = 0; i < mm; ++i) { shat[i] = 0.0; dshat[i] = 0.0; }
        shat[ip] = alpha / 1e-12;
        dshat[ip] = 1.0 / 1e-12;
        return;
    }
    const double inv = 1.0 / sqrt(nrm2);
    for (int i = 0; i < mm; ++i) vh[i] *= inv;
    vh[mm] = inv;
    double den = vh[mm];
    if (fabs(den) < 1e-12) den = 1e-12;
    for (int i = 0; i < mm; ++i) shat[i] = vh[i] * alpha / den;
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
}

}  // namespace
}  // namespace sella

using namespace sella;

extern "C" int sella_stepper_create(sella_ctx* c, int kind, sella_mat hV, sella_mat hVt, const double* evals,
                                    const double* g, int m, int order, sella_stepper** out) {
    if (!c || !out || !evals || !g || m <= 0 || order < 0 || order > m) {
        set_error("stepper: invalid arguments");
        return SELLA_E_INVALID;
    }
    if (kind < SELLA_STEP_QN || kind > SELLA_STEP_PRFO) {
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
    st->lam.assign(evals, evals + m);
    st->ghat.resize(m);
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

extern "C" int sella_stepper_get_s(sella_stepper* st, double alpha, double* s_out, double* dsda_out) {
    if (!st || !s_out || !dsda_out) return SELLA_E_INVALID;
    sella_ctx* c = st->c;
    const int m = st->m, o = st->order;
    const auto t0 = std::chrono::steady_clock::now();
    const long sw0 = g_sweeps;
    std::vector<double> sh(2 * (size_t)m, 0.0);     // [shat | dshat]
    double* shat = sh.data();
    double* dshat = sh.data() + m;
    const double* lam = st->lam.data();
    const double* gh = st->ghat.data();
    if (st->kind == SELLA_STEP_QN) {                                        // stepper.py:82-96
        for (int i = 0; i < m; ++i) {
            const double sgn = (i < o) ? -1.0 : 1.0;
            const double den = sgn * fabs(lam[i]) + alpha * sgn;
            const double sp = gh[i] / den;
            shat[i] = -sp;
            dshat[i] = sp / den;
        }
    } else if (st->kind == SELLA_STEP_RFO) {
        rfo_block(m, lam, gh, o, alpha, shat, dshat);
    } else {                                                                // P-RFO, stepper.py:163-185
        rfo_block(o, lam, gh, o, alpha, shat, dshat);                       // max block: top root
        rfo_block(m - o, lam + o, gh + o, 0, alpha, shat + o, dshat + o);   // min block: lowest root
    }
    const auto t1 = std::chrono::steady_clock::now();
    struct Acc {
        sella_stepper* st; std::chrono::steady_clock::time_point a, b; long sw;
        ~Acc() {
            st->t_host += std::chrono::duration<double>(b - a).count();
            st->t_dev += std::chrono::duration<double>(std::chrono::steady_clock::now() - b).count();
            st->calls += 1;
            st->sweeps += sw;
        }
    } acc{st, t0, t1, g_sweeps - sw0};
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
        HIPCHK(hipMemcpyAsync(dx, hin, (size_t)(ldx + m) * sizeof(double), hipMemcpyHostToDevice, c->stream));
        SCHK(launch_gemv_rows(c, V->d, nout, m, V->ld, dx, ldx, 2, hout, ldy, GemvEpi()));
        HIPCHK(hipStreamSynchronize(c->stream));
        memcpy(s_out, hout, (size_t)nout * sizeof(double));
        memcpy(dsda_out, hout + ldy, (size_t)nout * sizeof(double));
    } else {
        HIPCHK(hipMemcpyAsync(dx, shat, (size_t)m * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(dx + ldx, dshat, (size_t)m * sizeof(double), hipMemcpyHostToDevice, c->stream));
        SCHK(launch_gemv_rows(c, V->d, nout, m, V->ld, dx, ldx, 2, dy, ldy, GemvEpi()));
        HIPCHK(hipMemcpyAsync(s_out, dy, (size_t)nout * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(dsda_out, dy + ldy, (size_t)nout * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return SELLA_OK;
}

extern "C" int sella_stepper_destroy(sella_stepper* st) {
    if (!st) return SELLA_OK;
    if (st->calls && getenv("SELLA_DEBUG_TIMING"))
        fprintf(stderr, "stepper m=%d nout=%d: %ld get_s calls, %.1f us host solve (%.1f sweeps) + %.1f us device round trip per call\n",
                st->m, st->nout, st->calls, 1e6 * st->t_host / st->calls, (double)st->sweeps / st->calls,
                1e6 * st->t_dev / st->calls);
    delete st;
    return SELLA_OK;
}
