// The far side of the calculator boundary, for calculators that live in this library (sella/peswrapper.py:413-418 calls
// `atoms.get_potential_energy()` / `get_forces()`; any ASE calculator stays a host-language callback), and the
// finite-difference Hessian operator of sella/linalg.py:14-101 on top of one — so that an iterative diagonalisation
// (peswrapper.py:508-556) through such a calculator is ONE library call: sella_davidson with sella_fd_matvec as its
// operator, no host-language frame between the force calls.
//
//   sella_calc_model_*   f(x) = 1/2 x^T A x + c/3 sum_j (u_j . x)^3, A resident (the model PES of SURVEY.md section 8d)
//   sella_calc_emt_*     effective-medium theory (emt.hip, sella_emt_eval)
//   sella_fd_*           H v ~ (g(x0 + eta v^) - g0) / eta (or the central form), seen through a selection of free
//                        coordinates; every product is remembered as a secant pair for the Hessian update afterwards
#include "internal.h"

#include <cmath>
#include <vector>

using namespace sella;

struct sella_calc {
    sella_ctx* c = nullptr;
    int kind = 0;                     // 0 model, 1 EMT
    int n = 0;
    long ncalls = 0;
    // model
    sella_mat A = SELLA_NO_MAT;
    std::vector<double> U;
    int nu = 0;
    double cc = 0.0;
    // EMT
    int natoms = 0, nshift = 0;
    std::vector<double> par, shifts;
    double rc = 0, acut = 0, cutoff = 0, beta = 0;
    std::vector<double> work;
    double* dconst = nullptr;         // EMT: parameter table + shift vectors, resident; model: the rows u_j (nu x ld)
    size_t dconst_bytes = 0;
};

namespace {
// g_i = (A x)_i + sum_j c p_j^2 u_j[i],  p = U x: the gradient of the cubic terms, the rows u_j taken in order
__device__ __forceinline__ void model_grad_vb(const VB vb, int n, int nu, int ld, double cc, const double* __restrict__ Ax,
                                                         const double* __restrict__ p, const double* __restrict__ U,
                                                         double* __restrict__ g) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const int i = vb.x * 256 + threadIdx.x;
    if (i >= n) return;
    double v = Ax[i];
    for (int j = 0; j < nu; ++j) {
        const double w = cc * p[j] * p[j];
        v += w * U[(size_t)j * ld + i];
    }
    g[i] = v;
}
__global__ __launch_bounds__(256) void model_grad_kernel(int n, int nu, int ld, double cc, const double* __restrict__ Ax,
                                                         const double* __restrict__ p, const double* __restrict__ U,
                                                         double* __restrict__ g) { model_grad_vb(vb_hw(), n, nu, ld, cc, Ax, p, U, g); }
}  // namespace

extern "C" int sella_calc_model_create(sella_ctx* c, sella_mat A, const double* U, int nu, int n, double cc, sella_calc** out) {
    Mat* a = mat_get(c, A);
    if (!c || !out || !a || a->rows != n || a->cols != n || nu < 0 || (nu > 0 && !U)) {
        set_error("calc_model: invalid arguments");
        return SELLA_E_INVALID;
    }
    sella_calc* k = new sella_calc();
    k->c = c; k->kind = 0; k->n = n; k->A = A; k->nu = nu; k->cc = cc;
    k->U.assign(U, U + (size_t)nu * n);
    k->work.resize((size_t)n);
    if (nu > 0) {
        const int ld = round_up(n, 8);
        k->dconst_bytes = ((size_t)nu + 2) * ld * sizeof(double);
        int st = dev_alloc(c, k->dconst_bytes, &k->dconst);
        if (st == SELLA_OK) st = s_memset0(c, k->dconst, k->dconst_bytes) == hipSuccess ? SELLA_OK : SELLA_E_HIP;
        for (int j = 0; j < nu && st == SELLA_OK; ++j)
            st = h2d_async(c, k->dconst + (size_t)j * ld, U + (size_t)j * n, (size_t)n * sizeof(double));
        if (st == SELLA_OK) st = stream_wait(c);
        if (st != SELLA_OK) { delete k; return st; }
    }
    *out = k;
    return SELLA_OK;
}

int sella::calc_queue(sella_calc* k, const double* x, double** g_dev, double** aux_dev, int* naux) {
    sella_ctx* c = k->c;
    ++k->ncalls;
    if (k->kind == 1) {
        double *dea, *dgr;
        SCHK(emt_queue(c, k->natoms, x, k->par.data(), k->nshift, k->shifts.data(), k->dconst, k->rc, k->acut, k->cutoff, k->beta,
                       &dea, &dgr));
        *g_dev = dgr;
        *aux_dev = dea;
        *naux = k->natoms;
        return SELLA_OK;
    }
    const int n = k->n, ld = round_up(n, 8), nu = k->nu;
    Mat* A = mat_get(c, k->A);
    if (!A) return SELLA_E_INVALID;
    double* buf;                                       // x | A x | p (8-padded) | g
    const int ldp = round_up(nu > 0 ? nu : 1, 8);
    SCHK(scratch_get(c, SCR_MISC0, ((size_t)3 * ld + ldp) * sizeof(double), &buf));
    double *dx = buf, *dAx = buf + ld, *dp = buf + 2 * (size_t)ld, *dg = dp + ldp;
    SCHK(h2d_async(c, dx, x, (size_t)n * sizeof(double)));
    SCHK(launch_gemv_rows(c, A->d, n, n, A->ld, dx, ld, 1, dAx, ld, GemvEpi()));
    if (nu > 0) SCHK(launch_gemv_rows(c, k->dconst, nu, n, ld, dx, ld, 1, dp, ldp, GemvEpi()));
    SELLA_LAUNCHB(c, model_grad_kernel, model_grad_vb, 256, dim3((n + 255) / 256), dim3(256), 0, n, nu, ld, k->cc, dAx, dp, k->dconst, dg);
    HIPCHK(hipGetLastError());
    *g_dev = dg;
    *aux_dev = dAx;                                    // A x (ld entries) then p: one read-back
    *naux = ld + ldp;
    return SELLA_OK;
}

double sella::calc_finish(sella_calc* k, const double* x, const double* aux) {
    if (k->kind == 1) {
        double e = 0.0;
        for (int i = 0; i < k->natoms; ++i) e += aux[i];
        return e;
    }
    const int n = k->n, ld = round_up(n, 8);
    double e = 0.0;
    for (int i = 0; i < n; ++i) e += x[i] * aux[i];
    e *= 0.5;
    double cub = 0.0;
    for (int j = 0; j < k->nu; ++j) { const double p = aux[ld + j]; cub += p * p * p; }
    return e + k->cc / 3.0 * cub;
}

extern "C" int sella_calc_emt_create(sella_ctx* c, int natoms, const double* par, int nshift, const double* shifts, double rc,
                                     double acut, double cutoff, double beta, sella_calc** out) {
    if (!c || !out || natoms <= 0 || !par || nshift <= 0 || !shifts) {
        set_error("calc_emt: invalid arguments");
        return SELLA_E_INVALID;
    }
    sella_calc* k = new sella_calc();
    k->c = c; k->kind = 1; k->n = 3 * natoms; k->natoms = natoms; k->nshift = nshift;
    k->par.assign(par, par + (size_t)9 * natoms);
    k->shifts.assign(shifts, shifts + (size_t)3 * nshift);
    k->rc = rc; k->acut = acut; k->cutoff = cutoff; k->beta = beta;
    k->dconst_bytes = ((size_t)9 * natoms + (size_t)3 * nshift) * sizeof(double);
    if (dev_alloc(c, k->dconst_bytes, &k->dconst) == SELLA_OK) {
        int st = h2d_async(c, k->dconst, par, (size_t)9 * natoms * sizeof(double));
        if (st == SELLA_OK) st = h2d_async(c, k->dconst + (size_t)9 * natoms, shifts, (size_t)3 * nshift * sizeof(double));
        if (st == SELLA_OK) st = stream_wait(c);
        if (st != SELLA_OK) { dev_free(c, k->dconst, k->dconst_bytes); k->dconst = nullptr; }
    } else {
        k->dconst = nullptr;
    }
    *out = k;
    return SELLA_OK;
}

// energy and gradient dE/dx at x (n entries)
extern "C" int sella_calc_eval(sella_calc* k, const double* x, double* f, double* g) {
    if (!k || !x || !f || !g) return SELLA_E_INVALID;
    double *dg, *daux;
    int naux = 0;
    SCHK(calc_queue(k, x, &dg, &daux, &naux));
    std::vector<double>& aux = k->work;
    aux.resize((size_t)naux);
    SCHK(d2h_async(k->c, aux.data(), daux, (size_t)naux * sizeof(double)));
    SCHK(d2h_async(k->c, g, dg, (size_t)k->n * sizeof(double)));
    SCHK(stream_wait(k->c));
    *f = calc_finish(k, x, aux.data());
    return SELLA_OK;
}

extern "C" long sella_calc_ncalls(sella_calc* k) { return k ? k->ncalls : 0; }
extern "C" int sella_calc_dim(sella_calc* k) { return k ? k->n : 0; }
extern "C" int sella_calc_destroy(sella_calc* k) {
    if (k && k->dconst) dev_free(k->c, k->dconst, k->dconst_bytes);
    delete k;
    return SELLA_OK;
}

// ---- finite-difference Hessian operator (sella/linalg.py:14-101) ---------------------------------------------------------
struct sella_fd {
    sella_calc* calc = nullptr;
    int n = 0, m = 0;                 // full dimension, dimension the eigensolver sees
    double eta = 0.0;
    int threepoint = 0;
    std::vector<double> x0, g0, vfull, xd, ahead, behind;
    std::vector<int> idx;             // free coordinates (empty: all)
    std::vector<double> Vs, AVs;      // recorded pairs, k x n each (pair-major)
    int npairs = 0;
    long calls = 0;
};

extern "C" int sella_fd_create(sella_calc* calc, int n, const double* x0, const double* g0, double eta, int threepoint,
                               const int* idx, int m, sella_fd** out) {
    if (!calc || !out || !x0 || !g0 || n <= 0 || calc->n != n || !(eta > 0.0) || (idx && (m <= 0 || m > n))) {
        set_error("fd operator: invalid arguments");
        return SELLA_E_INVALID;
    }
    sella_fd* o = new sella_fd();
    o->calc = calc; o->n = n; o->eta = eta; o->threepoint = threepoint;
    o->x0.assign(x0, x0 + n);
    o->g0.assign(g0, g0 + n);
    if (idx) o->idx.assign(idx, idx + m);
    o->m = idx ? m : n;
    o->vfull.resize(n); o->xd.resize(n); o->ahead.resize(n); o->behind.resize(n);
    *out = o;
    return SELLA_OK;
}

// sella_matvec_fn: Av = U^T H U v through finite differences of the calculator's gradient
extern "C" int sella_fd_matvec(void* user, const double* v, double* Av, int m) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    sella_fd* o = static_cast<sella_fd*>(user);
    if (!o || !v || !Av || m != o->m) return SELLA_E_INVALID;
    const int n = o->n;
    ++o->calls;
    double* vf = o->vfull.data();
    if (o->idx.empty()) {
        for (int i = 0; i < n; ++i) vf[i] = v[i];
    } else {
        for (int i = 0; i < n; ++i) vf[i] = 0.0;
        for (int q = 0; q < m; ++q) vf[o->idx[q]] = v[q];
    }
    double len2 = 0.0;
    for (int i = 0; i < n; ++i) len2 += vf[i] * vf[i];
    const double length = std::sqrt(len2);
    if (length < 1e-12) {
        for (int q = 0; q < m; ++q) Av[q] = 0.0;
        return SELLA_OK;
    }
    // which of +-v is displaced along (linalg.py:45-73): downhill if v has a gradient component, else towards the origin,
    // else so that the first significant component is positive
    double sign = 0.0;
    const double* refs[2] = {o->g0.data(), o->x0.data()};
    for (int t = 0; t < 2 && sign == 0.0; ++t) {
        double proj = 0.0;
        for (int i = 0; i < n; ++i) proj += vf[i] * refs[t][i];
        if (std::fabs(proj) > 1e-4) sign = proj > 0.0 ? -1.0 : 1.0;
    }
    if (sign == 0.0) {
        sign = 1.0;
        for (int i = 0; i < n; ++i)
            if (std::fabs(vf[i]) > 1e-4) { sign = vf[i] < 0.0 ? -1.0 : 1.0; break; }
    }
    const double scale = sign * length;
    double f;
    for (int i = 0; i < n; ++i) o->xd[i] = o->x0[i] + (o->eta * vf[i]) / scale;
    SCHK(sella_calc_eval(o->calc, o->xd.data(), &f, o->ahead.data()));
    o->Vs.insert(o->Vs.end(), vf, vf + n);
    const size_t at = o->AVs.size();
    o->AVs.resize(at + n);
    double* out = o->AVs.data() + at;
    if (o->threepoint) {
        for (int i = 0; i < n; ++i) o->xd[i] = o->x0[i] - (o->eta * vf[i]) / scale;
        SCHK(sella_calc_eval(o->calc, o->xd.data(), &f, o->behind.data()));
        for (int i = 0; i < n; ++i) out[i] = (scale * (o->ahead[i] - o->behind[i])) / (2 * o->eta);
    } else {
        for (int i = 0; i < n; ++i) out[i] = (scale * (o->ahead[i] - o->g0[i])) / o->eta;
    }
    ++o->npairs;
    if (o->idx.empty()) for (int i = 0; i < n; ++i) Av[i] = out[i];
    else for (int q = 0; q < m; ++q) Av[q] = out[o->idx[q]];
    return SELLA_OK;
}

extern "C" int sella_fd_npairs(sella_fd* o) { return o ? o->npairs : 0; }
extern "C" long sella_fd_calls(sella_fd* o) { return o ? o->calls : 0; }

// recorded pairs as (n x k) row-major matrices (columns = products in call order)
extern "C" int sella_fd_pairs(sella_fd* o, double* Vs, double* AVs) {
    if (!o || !Vs || !AVs) return SELLA_E_INVALID;
    const int n = o->n, k = o->npairs;
    for (int p = 0; p < k; ++p)
        for (int i = 0; i < n; ++i) {
            Vs[(size_t)i * k + p] = o->Vs[(size_t)p * n + i];
            AVs[(size_t)i * k + p] = o->AVs[(size_t)p * n + i];
        }
    return SELLA_OK;
}

extern "C" int sella_fd_destroy(sella_fd* o) {
    delete o;
    return SELLA_OK;
}
