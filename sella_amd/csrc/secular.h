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

// g(tau) and g'(tau) for the shifted function, origin o
template <class Sum>
SELLA_HD inline void eval(int K, const double* D, const double* w, double rho, int o, double tau,
                          double* g, double* dg, int i0, int istep, Sum sum) {
    const double Do = D[o];
    double s = 0.0, ds = 0.0;
    for (int i = i0; i < K; i += istep) {
        const double r = 1.0 / ((D[i] - Do) - tau);
        const double t = w[i] * w[i] * r;
        s += t;
        ds += t * r;
    }
    s = sum(s);
    ds = sum(ds);
    *g = 1.0 + rho * s;
    *dg = rho * ds;
}

// Solve for root j.  Returns the number of iterations used (negative if the cap was hit).
template <class Sum>
SELLA_HD inline int solve_root(int K, const double* D, const double* w, double rho, int j,
                               int* origin, double* tau_out, int i0, int istep, Sum sum) {
    double lo, hi;
    int o;
    if (j < K - 1) {
        const double delta = D[j + 1] - D[j];
        double gm, dgm;
        eval(K, D, w, rho, j, 0.5 * delta, &gm, &dgm, i0, istep, sum);
        if (gm >= 0.0) { o = j; lo = 0.0; hi = 0.5 * delta; }
        else { o = j + 1; lo = -0.5 * delta; hi = 0.0; }
    } else {
        double ww = 0.0;
        for (int i = i0; i < K; i += istep) ww += w[i] * w[i];
        ww = sum(ww);
        o = K - 1;
        lo = 0.0;
        hi = rho * ww;
        hi += 4.0 * 2.220446049250313e-16 * fabs(hi) + 1e-300;   // never round below the root
    }
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
    for (it = 0; it < 200; ++it) {
        double g, dg;
        eval(K, D, w, rho, o, tau, &g, &dg, i0, istep, sum);
        if (g == 0.0) break;
        if (g < 0.0) lo = tau; else hi = tau;
        // Newton on h(tau) = tau * g(tau): smooth across the origin pole
        const double h = tau * g;
        const double dh = g + tau * dg;
        double tn = (dh != 0.0) ? tau - h / dh : 0.5 * (lo + hi);
        if (!(tn > lo && tn < hi)) tn = 0.5 * (lo + hi);
        if (tn == lo || tn == hi || tn == tau) { tau = tn; break; }
        const double step = fabs(tn - tau);
        tau = tn;
        if (step <= 2.220446049250313e-16 * fabs(tau)) break;
        if (hi - lo <= 2.220446049250313e-16 * fmax(fabs(lo), fabs(hi))) break;
    }
    *origin = o;
    *tau_out = tau;
    return it < 200 ? it : -it;
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
