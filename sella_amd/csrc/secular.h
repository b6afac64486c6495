// secular.h — roots of the secular equation of a rank-one modified diagonal matrix,
//     f(x) = 1 + rho * sum_i w_i^2 / (D_i - x) = 0,   D ascending and distinct, rho > 0, w_i != 0,
// used by the merge step of the divide-and-conquer symmetric eigensolver (eigh.hip) and by
// the RFO step solve (bordered diagonal matrices).  Host/device code, fp64.
//
// Root j lies in (D_j, D_{j+1}) (the last one in (D_{K-1}, D_{K-1} + rho*|w|^2]).  Each root
// is returned as (origin, tau) with lambda_j = D_origin + tau where D_origin is the CLOSER
// pole, so that the differences D_i - lambda_j = (D_i - D_origin) - tau keep full relative
// accuracy — the property the Gu/Eisenstat eigenvector formula needs.
#pragma once
#include <math.h>

#ifndef SELLA_HD
#define SELLA_HD __host__ __device__
#endif

namespace sella {
namespace secular {

// g(tau) and g'(tau) for the shifted function, origin o
SELLA_HD inline void eval(int K, const double* D, const double* w, double rho, int o, double tau,
                          double* g, double* dg) {
    const double Do = D[o];
    double s = 0.0, ds = 0.0;
    for (int i = 0; i < K; ++i) {
        const double r = 1.0 / ((D[i] - Do) - tau);
        const double t = w[i] * w[i] * r;
        s += t;
        ds += t * r;
    }
    *g = 1.0 + rho * s;
    *dg = rho * ds;
}

// Solve for root j.  Returns the number of iterations used (negative if the cap was hit).
SELLA_HD inline int solve_root(int K, const double* D, const double* w, double rho, int j,
                               int* origin, double* tau_out) {
    double lo, hi;
    int o;
    if (j < K - 1) {
        const double delta = D[j + 1] - D[j];
        double gm, dgm;
        eval(K, D, w, rho, j, 0.5 * delta, &gm, &dgm);
        if (gm >= 0.0) { o = j; lo = 0.0; hi = 0.5 * delta; }
        else { o = j + 1; lo = -0.5 * delta; hi = 0.0; }
    } else {
        double ww = 0.0;
        for (int i = 0; i < K; ++i) ww += w[i] * w[i];
        o = K - 1;
        lo = 0.0;
        hi = rho * ww;
        // guard against hi rounding below the root
        hi += 4.0 * 2.220446049250313e-16 * fabs(hi) + 1e-300;
    }
    // invariant: g(lo) < 0 <= g(hi)  (one end may be the pole itself, never evaluated)
    double tau = 0.5 * (lo + hi);
    // a first guess that respects the nearby pole: tau ~ rho w_o^2 / (1 + rho * S_rest(0))
    {
        double s = 0.0;
        const double Do = D[o];
        for (int i = 0; i < K; ++i)
            if (i != o) s += w[i] * w[i] / (D[i] - Do);
        const double den = 1.0 + rho * s;
        if (den != 0.0) {
            const double t0 = rho * w[o] * w[o] / den;
            if (t0 > lo && t0 < hi) tau = t0;
        }
    }
    int it;
    for (it = 0; it < 200; ++it) {
        double g, dg;
        eval(K, D, w, rho, o, tau, &g, &dg);
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
        if (step <= 2.220446049250313e-16 * fabs(tau)) {
            // converged to an ulp: polish the bracket side once more and stop
            break;
        }
        if (hi - lo <= 2.220446049250313e-16 * fmax(fabs(lo), fabs(hi))) break;
    }
    *origin = o;
    *tau_out = tau;
    return it < 200 ? it : -it;
}

// zhat_i^2 = (lam_i - D_i) * prod_{j != i} (lam_j - D_i) / (D_j - D_i)       (Gu & Eisenstat)
// with lam_j - D_i = (D_{org_j} - D_i) + tau_j.  Returns zhat_i with the sign of w_i.
SELLA_HD inline double zhat(int K, const double* D, const double* w, const double* tau,
                            const int* org, int i) {
    const double Di = D[i];
    double p = (D[org[i]] - Di) + tau[i];
    for (int j = 0; j < K; ++j) {
        if (j == i) continue;
        p *= ((D[org[j]] - Di) + tau[j]) / (D[j] - Di);
    }
    const double z = sqrt(fabs(p));
    return w[i] >= 0.0 ? z : -z;
}

}  // namespace secular
}  // namespace sella
