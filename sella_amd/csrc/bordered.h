// bordered.h — roots of the secular equation of a bordered diagonal (arrowhead) matrix
//     [[diag(D), b], [b^T, 0]]  :  f(mu) = mu + sum_i b_i^2 / (D_i - mu) = 0.
// Shared by the RFO / P-RFO step solve (stepper.hip: on the host per trial alpha, on the device for batches of
// trial alphas) and the O(k^2) Rayleigh-Ritz update of the Davidson loop (host_math.h `arrow_eig`).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include <algorithm>

namespace sella {
namespace bordered {

inline thread_local long g_sweeps = 0;      // SELLA_DEBUG_TIMING statistics (one context per host thread)

// Root number j (ascending, 0..mm) of f(mu) = mu + sum b_i^2 / (D_i - mu), D ascending.
// Returned as (origin, tau): mu = D_origin + tau with the origin the closer pole
// (origin = -1: mu = tau, used for the two exterior roots far from every pole).
//
// The O(1) logic is shared by the host (plain loops over arrays) and the device (one workgroup per problem, the
// sums as block reductions — stepper.hip `rs_batch_kernel`): `Dat(i)`, `bat(i)` return single entries, `bb` is
// sum b_i^2, and `eval(shift, t)` returns the value, its noise scale and the derivative split at pole index j
// (dl: poles i < j, dr: poles i >= j) of f at mu = shift + t.
struct Ev { double f, noise, dl, dr; };

template <class DAt, class BAt, class Eval>
__host__ __device__ inline void bordered_root_core(int mm, DAt Dat, BAt bat, double bb, int j, Eval eval, int* origin,
                                                   double* tau) {
    const int ext = (j == 0) ? 0 : (j == mm ? mm - 1 : -1);      // nearest pole of an exterior root
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
        shift = Dat(e);
        // t = (-shift + sg sqrt(shift^2 + 4 c)) / 2, written without cancellation when -shift and sg have opposite
        // signs (a border that is tiny next to |shift| — every nearly converged Ritz pair — would otherwise give 0)
        auto qroot = [&](double cw) {
            const double R = sqrt(shift * shift + 4.0 * cw);
            return (-shift * sg >= 0.0) ? 0.5 * (-shift + sg * R) : 2.0 * cw / (shift + sg * R);
        };
        const double far = qroot(bb);                         // t of the overshooting model
        const double near = qroot(bat(e) * bat(e));
        if (j == 0) { lo = far; hi = fmin(near, 0.0); }
        else { lo = fmax(near, 0.0); hi = far; }
        if (!(hi > lo)) { *origin = org; *tau = 0.5 * (lo + hi); return; }                 // b = 0: mu = min/max(D_e, 0)
        t = far;
        if (t == 0.0) t = 0.5 * (lo + hi);
    } else {
        const double delta = Dat(j) - Dat(j - 1);
        if (delta <= 0.0) { *origin = j; *tau = 0.0; return; }      // coincident poles: mu = D_j
        const double fm = eval(Dat(j - 1), 0.5 * delta).f;
        if (fm >= 0.0) { org = j - 1; shift = Dat(j - 1); lo = 0.0; hi = 0.5 * delta; }
        else { org = j; shift = Dat(j); lo = -0.5 * delta; hi = 0.0; }
        t = 0.5 * (lo + hi);
    }
    // f is increasing between poles: f(lo) <= 0 <= f(hi) (pole ends are never evaluated).  Interior roots: as in
    // the eigensolver's secular equation (secular.h), the two poles next to the root are kept exact and the rest of
    // the sum is frozen at value and slope ("middle way" rational model).  Exterior roots: the nearest pole is
    // kept exact and the rest of the sum is replaced by the one-pole function that matches its value and slope (all terms have
    // the same sign there) next to the exact nearest pole — with the nearest pole alone these roots took 30-60
    // sweeps at the sizes of a slab search, now 3-6.  Bracket + bisection as the safeguard, stop at |f| below its rounding noise.
    const double EPS = 2.220446049250313e-16;
    // (the root of the rational MODEL is only the next trial point: it is iterated to a relative step of 1e-6, not to the
    //  last bit — round 6: the scalar model solves were 40 % of the Davidson loop's k x k Rayleigh-Ritz time and the O(1)
    //  part of every trial alpha of the step families; the number of sweeps over the sum is unchanged)
    const double MODEL_TOL = 1e-6;
    for (int it = 0; it < 200; ++it) {
        const Ev e = eval(shift, t);
        const double fv = e.f;
        if (!(fabs(fv) > 8.0 * EPS * e.noise)) break;
        if (fv < 0.0) lo = t; else hi = t;
        const double df = 1.0 + e.dl + e.dr;
        double eta;
        if (j > 0 && j < mm) {
            // Interior root: both neighbouring poles exact, the rest of each side's sum frozen at value and slope
            // ("middle way"), AND the linear term mu kept exactly — without it the quadratic model overshoots on
            // alternate sides whenever mu' = 1 matters next to the pole slopes, and the bracket shrinks by a few
            // per cent per sweep.  Model:  F(x) = (mu + x) + c0 + a1 / (p1 - x) + a2 / (p2 - x),  p1 < 0 < p2,
            // increasing between the poles; its root by safeguarded scalar Newton (O(1) per sweep).
            const double mu = shift + t;
            const double p1 = (Dat(j - 1) - shift) - t, p2 = (Dat(j) - shift) - t;
            const double a1 = e.dl * p1 * p1, a2 = e.dr * p2 * p2;
            const double c0 = (fv - mu) - e.dl * p1 - e.dr * p2;
            double elo = fmax(lo - t, p1), ehi = fmin(hi - t, p2), x = 0.0, Fx = fv;
            eta = -fv / df;
            for (int in = 0; in < 60; ++in) {
                if (Fx < 0.0) elo = x; else ehi = x;
                const double r1 = 1.0 / (p1 - x), r2 = 1.0 / (p2 - x);
                const double dF = 1.0 + a1 * r1 * r1 + a2 * r2 * r2;
                double xn = x - Fx / dF;
                if (!(xn > elo && xn < ehi)) xn = 0.5 * (elo + ehi);
                if (xn == x) break;
                const double moved = fabs(xn - x);
                x = xn;
                if (moved <= MODEL_TOL * fabs(x)) break;
                const double q1 = a1 / (p1 - x), q2 = a2 / (p2 - x);
                Fx = ((mu + x) + c0) + q1 + q2;
                if (fabs(Fx) <= 4.0 * EPS * (fabs(mu + x) + fabs(c0) + fabs(q1) + fabs(q2))) break;
            }
            if (x != 0.0) eta = x;
        } else {
            // model: (mu + eta) + a1 / (p1 - eta) + a2 / (p2 - eta) = 0 with the nearest pole exact
            // (a1 = b_e^2, p1 = D_e - mu) and the REST of the sum R replaced by the one-pole function that
            // matches R and R' at the current point (p2 = R / R', a2 = R p2).  Solved for eta by a safeguarded
            // scalar Newton iteration inside the bracket — O(1) work per sweep.
            const double mu = shift + t;
            const double p1 = (Dat(ext) - shift) - t, a1 = bat(ext) * bat(ext);
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
                const double moved = fabs(xn - x);
                x = xn;
                if (moved <= MODEL_TOL * fabs(x)) break;
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
        if (hi - lo <= EPS * fmax(fabs(lo), fabs(hi))) break;
    }
    *origin = org;
    *tau = t;
}

// host form: arrays D (ascending), b
// wide: sum four terms at a time (the step families' O(m) problems and, since round 6, the Davidson loop's k x k
// Rayleigh-Ritz: once the model solves above stopped dominating, the divisions of the sums are what is left).
inline void bordered_root(int mm, const double* D, const double* b, int j, int* origin, double* tau, bool wide = false) {
    double bb = 0.0;
    for (int i = 0; i < mm; ++i) bb += b[i] * b[i];
    // The sums are the whole cost (one division per term, 3-6 sweeps per root, one root per Ritz value and iteration in
    // the Davidson loop): four terms at a time through the compiler's vector types — vdivpd on the AVX2 hosts the library
    // is built for, two SSE2 halves elsewhere — with scalar loops for the ragged ends.
    typedef double v4d __attribute__((vector_size(32)));
    typedef long long v4i __attribute__((vector_size(32)));
    auto range = [&](int lo, int hi, double shift, double t, double* s, double* sa, double* dd) {
        v4d s4 = {0.0, 0.0, 0.0, 0.0}, a4 = s4, d4 = s4;
        const v4i mask = {0x7fffffffffffffffLL, 0x7fffffffffffffffLL, 0x7fffffffffffffffLL, 0x7fffffffffffffffLL};
        int i = lo;
        for (; wide && i + 4 <= hi; i += 4) {
            v4d Dv, bv;
            memcpy(&Dv, D + i, sizeof(v4d));
            memcpy(&bv, b + i, sizeof(v4d));
            const v4d r = 1.0 / ((Dv - shift) - t);
            const v4d q = bv * bv * r;
            v4i qi;
            memcpy(&qi, &q, sizeof(v4d));
            qi &= mask;
            v4d aq;
            memcpy(&aq, &qi, sizeof(v4d));
            s4 += q;
            a4 += aq;
            d4 += q * r;
        }
        // the scalar path continues the running sums term by term (the order of the plain loops this replaces)
        double ss = *s, sas = *sa, ds = *dd;
        if (i > lo) {
            ss += (s4[0] + s4[1]) + (s4[2] + s4[3]);
            sas += (a4[0] + a4[1]) + (a4[2] + a4[3]);
            ds += (d4[0] + d4[1]) + (d4[2] + d4[3]);
        }
        for (; i < hi; ++i) {
            const double r = 1.0 / ((D[i] - shift) - t);
            const double q = b[i] * b[i] * r;
            ss += q;
            sas += fabs(q);
            ds += q * r;
        }
        *s = ss;
        *sa = sas;
        *dd = ds;
    };
    auto eval = [&](double shift, double t) {
        Ev e;
        double s = 0.0, sa = 0.0, dl = 0.0, dr = 0.0;
        range(0, j, shift, t, &s, &sa, &dl);          // poles left of the root
        range(j, mm, shift, t, &s, &sa, &dr);         // poles right of the root
        e.f = (shift + t) + s;
        e.noise = fabs(shift + t) + sa;
        e.dl = dl;
        e.dr = dr;
        ++g_sweeps;
        return e;
    };
    bordered_root_core(mm, [&](int i) { return D[i]; }, [&](int i) { return b[i]; }, bb, j, eval, origin, tau);
}

}  // namespace bordered
}  // namespace sella
