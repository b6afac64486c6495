// secular.h — roots of the secular equation of a rank-one modified diagonal matrix,
//     f(x) = 1 + rho * sum_i w_i^2 / (D_i - x) = 0,   D ascending and distinct, rho > 0, w_i != 0,
// used by the merge step of the divide-and-conquer symmetric eigensolver (eigh.hip).
// Host/device code, fp64.
//
// Root j lies in (D_j, D_{j+1}) (the last one in (D_{K-1}, D_{K-1} + rho*|w|^2]).  Each root
// is returned as (origin, tau) with lambda_j = D_origin + tau where D_origin is the CLOSER
// pole, so that the differences D_i - lambda_j = (D_i - D_origin) - tau keep full relative
// accuracy — the property the Gu/Eisenstat eigenvector formula needs.
//
// The O(K) sums are written for a cooperating group: the caller owns terms i0, i0+istep, ... and
// supplies `sum` / `prod` functors that combine the partial results across the group (a wavefront
// on the device, the identity on the host).  All members of the group follow identical control flow.
#pragma once
#include <math.h>

#ifndef SELLA_HD
#define SELLA_HD __host__ __device__
#endif

namespace sella {
namespace secular {

struct Alone {
    SELLA_HD double operator()(double v) const { return v; }
};

// Pieces of the shifted function g(tau) = 1 + rho * sum_i w_i^2 / ((D_i - D_o) - tau) at tau:
// the sum split at pole index j (psi: i <= j, phi: i > j) with its derivative parts, and the sum of
// the absolute values of the terms (the evaluation-noise scale of g).
struct Eval {
    double g, dpsi, dphi, noise;
};

template <class Sum>
SELLA_HD inline Eval eval(int K, const double* D, const double* w, double rho, int o, int j, double tau,
                          int i0, int istep, Sum sum) {
    const double Do = D[o];
    double s = 0.0, sa = 0.0, d1 = 0.0, d2 = 0.0;
    for (int i = i0; i < K; i += istep) {
        const double r = 1.0 / ((D[i] - Do) - tau);
        const double t = w[i] * w[i] * r;
        s += t;
        sa += fabs(t);
        if (i <= j) d1 += t * r; else d2 += t * r;
    }
    Eval e;
    e.g = 1.0 + rho * sum(s);
    e.noise = 1.0 + rho * sum(sa);
    e.dpsi = rho * sum(d1);
    e.dphi = rho * sum(d2);
    return e;
}

// Solve for root j.  Returns the number of iterations used (negative if the cap was hit).
//
// Iteration: the two poles that bracket the root are kept exact and the remaining sums are replaced
// by constants fitted to value and slope ("middle way" rational interpolation, R.-C. Li 1994, the
// scheme of LAPACK's dlaed4); the update is the root of the resulting quadratic, safeguarded by the
// sign bracket [lo, hi] with bisection as the fallback.  Convergence is declared when |g| drops
// below its own rounding noise, 8 eps (1 + rho sum |terms|): beyond that point the sign of g is
// random and neither the bracket nor the iteration can make progress.
template <class Sum>
SELLA_HD inline int solve_root(int K, const double* D, const double* w, double rho, int j,
                               int* origin, double* tau_out, int i0, int istep, Sum sum) {
    const double EPS = 2.220446049250313e-16;
    const bool last = (j == K - 1);
    double lo, hi;
    int o;
    if (!last) {
        const double delta = D[j + 1] - D[j];
        const Eval m = eval(K, D, w, rho, j, j, 0.5 * delta, i0, istep, sum);
        if (m.g >= 0.0) { o = j; lo = 0.0; hi = 0.5 * delta; }
        else { o = j + 1; lo = -0.5 * delta; hi = 0.0; }
    } else {
        double ww = 0.0;
        for (int i = i0; i < K; i += istep) ww += w[i] * w[i];
        ww = sum(ww);
        o = K - 1;
        lo = 0.0;
        hi = rho * ww;
        hi += 4.0 * EPS * fabs(hi) + 1e-300;   // never round below the root
    }
    // poles either side of the root, relative to the origin
    const double p1 = D[j] - D[o];
    const double p2 = last ? 0.0 : D[j + 1] - D[o];
    // invariant: g(lo) < 0 <= g(hi)  (one end may be the pole itself, never evaluated)
    double tau = 0.5 * (lo + hi);
    {   // first guess that respects the nearby pole: tau ~ rho w_o^2 / (1 + rho * S_rest(0))
        double s = 0.0;
        const double Do = D[o];
        for (int i = i0; i < K; i += istep)
            if (i != o) s += w[i] * w[i] / (D[i] - Do);
        s = sum(s);
        const double den = 1.0 + rho * s;
        if (den != 0.0) {
            const double t0 = rho * w[o] * w[o] / den;
            if (t0 > lo && t0 < hi) tau = t0;
        }
    }
    int it;
    for (it = 0; it < 100; ++it) {
        const Eval e = eval(K, D, w, rho, o, j, tau, i0, istep, sum);
        const double g = e.g;
        if (!(fabs(g) > 8.0 * EPS * e.noise)) break;          // also leaves on NaN
        if (g < 0.0) lo = tau; else hi = tau;
        const double dg = e.dpsi + e.dphi;
        const double D1 = p1 - tau;                             // < 0
        double eta;
        if (!last) {
            const double D2 = p2 - tau;                         // > 0
            const double c = g - D1 * e.dpsi - D2 * e.dphi;
            const double a = (D1 + D2) * g - D1 * D2 * dg;
            const double b = D1 * D2 * g;
            if (c == 0.0) {
                eta = (a != 0.0) ? b / a : -g / dg;
            } else {
                const double disc = sqrt(fabs(a * a - 4.0 * b * c));
                eta = (a <= 0.0) ? (a - disc) / (2.0 * c) : 2.0 * b / (a + disc);
            }
        } else {
            // single pole on the left: g ~ c + a/(p1 - x), a = dpsi D1^2, c = g - D1 dpsi
            const double c = g - D1 * e.dpsi;
            const double a = e.dpsi * D1 * D1;
            eta = (c > 0.0) ? (D1 + a / c) : -g / dg;
        }
        if (!(g * eta < 0.0)) eta = -g / dg;                    // wrong direction: Newton
        double tn = tau + eta;
        if (!(tn > lo && tn < hi)) tn = 0.5 * (lo + hi);
        if (tn == lo || tn == hi || tn == tau) { tau = tn; break; }
        tau = tn;
        if (hi - lo <= EPS * fmax(fabs(lo), fabs(hi))) break;
    }
    *origin = o;
    *tau_out = tau;
    return it < 100 ? it : -it;
}

// zhat_i^2 = (lam_i - D_i) * prod_{j != i} (lam_j - D_i) / (D_j - D_i)       (Gu & Eisenstat)
// with lam_j - D_i = (D_{org_j} - D_i) + tau_j.  Returns zhat_i with the sign of w_i.
template <class Prod>
SELLA_HD inline double zhat(int K, const double* D, const double* w, const double* tau,
                            const int* org, int i, int j0, int jstep, Prod prod) {
    const double Di = D[i];
    double p = 1.0;
    for (int j = j0; j < K; j += jstep) {
        const double num = (D[org[j]] - Di) + tau[j];
        p *= (j == i) ? num : num / (D[j] - Di);
    }
    p = prod(p);
    const double z = sqrt(fabs(p));
    return w[i] >= 0.0 ? z : -z;
}

}  // namespace secular
}  // namespace sella
