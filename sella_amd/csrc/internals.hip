// internals.hip — batched internal-coordinate primitives: value, gradient, Hessian-vector product and
// Hessian blocks of bonds, angles and dihedrals.
//
// Replaces the JAX functions of sella/internal.py:58-135: `_bond_value` / `_angle_value` /
// `_dihedral_value` (:58-80), their `grad` (:85-87), `jacfwd(grad)` (:95-97) and `jvp(grad)` (:106-135),
// all vmapped over the coordinates of one kind.  The value functions below are written once, as
// templates over the scalar type, in exactly the reference's form (same vectors, same clip, same
// arctan2 arguments), and are differentiated by forward-mode hyper-dual arithmetic
//     x + a e1 + b e2 + c e1e2,   e1^2 = e2^2 = 0,
// seeded with e1 = unit vector k and e2 = the tangent (HVP) or unit vector l (Hessian): the e1 part of
// the result is dq/dx_k, the e1e2 part is (H t)_k or H_kl.  One thread evaluates one (coordinate, k[, l])
// pair — 4 doubles per intermediate instead of a 12-wide gradient per intermediate, everything stays in
// registers, and the outputs are written coalesced in the (nc, natoms, 3) layout of the reference.
#include "internal.h"

namespace sella {
namespace {

struct HDual {
    double v, a, b, c;           // value, d/de1, d/de2, d2/de1de2
};

__device__ __forceinline__ HDual hd(double v) { return HDual{v, 0.0, 0.0, 0.0}; }
__device__ __forceinline__ HDual operator+(HDual x, HDual y) { return HDual{x.v + y.v, x.a + y.a, x.b + y.b, x.c + y.c}; }
__device__ __forceinline__ HDual operator-(HDual x, HDual y) { return HDual{x.v - y.v, x.a - y.a, x.b - y.b, x.c - y.c}; }
__device__ __forceinline__ HDual operator-(HDual x) { return HDual{-x.v, -x.a, -x.b, -x.c}; }
__device__ __forceinline__ HDual operator+(HDual x, double y) { return HDual{x.v + y, x.a, x.b, x.c}; }
__device__ __forceinline__ HDual operator*(HDual x, HDual y) {
    return HDual{x.v * y.v, x.a * y.v + x.v * y.a, x.b * y.v + x.v * y.b,
                 x.c * y.v + x.a * y.b + x.b * y.a + x.v * y.c};
}
// phi(x) with phi' = d1, phi'' = d2 at x.v
__device__ __forceinline__ HDual chain(HDual x, double f, double d1, double d2) {
    return HDual{f, d1 * x.a, d1 * x.b, d2 * x.a * x.b + d1 * x.c};
}
__device__ __forceinline__ HDual recip(HDual x) {
    const double r = 1.0 / x.v;
    return chain(x, r, -r * r, 2.0 * r * r * r);
}
__device__ __forceinline__ HDual operator/(HDual x, HDual y) { return x * recip(y); }
__device__ __forceinline__ HDual sqrt(HDual x) {
    const double s = ::sqrt(x.v);
    return chain(x, s, 0.5 / s, -0.25 / (s * x.v));
}
__device__ __forceinline__ HDual clip1(HDual x) {                  // jnp.clip(x, -1, 1): flat outside
    if (x.v > 1.0) return hd(1.0);
    if (x.v < -1.0) return hd(-1.0);
    return x;
}
__device__ __forceinline__ HDual acos(HDual x) {
    const double om = 1.0 - x.v * x.v;
    const double d1 = -1.0 / ::sqrt(om);
    return chain(x, ::acos(x.v), d1, x.v * d1 / om);
}
__device__ __forceinline__ HDual atan2(HDual y, HDual x) {
    // d atan2 = (x dy - y dx) / (x^2 + y^2): built from the hyper-dual quotient so that second
    // derivatives follow from the same arithmetic
    const HDual r2 = x * x + y * y;
    const HDual u = x / r2, w = -(y / r2);                         // d/dy and d/dx of atan2
    HDual out;
    out.v = ::atan2(y.v, x.v);
    out.a = u.v * y.a + w.v * x.a;
    out.b = u.v * y.b + w.v * x.b;
    out.c = u.v * y.c + w.v * x.c + u.b * y.a + w.b * x.a;
    return out;
}

template <class T>
struct V3 {
    T x, y, z;
};
template <class T>
__device__ __forceinline__ V3<T> sub(V3<T> p, V3<T> q, const double* t) {   // p - q + t
    return V3<T>{p.x - q.x + t[0], p.y - q.y + t[1], p.z - q.z + t[2]};
}
template <class T>
__device__ __forceinline__ T dot(V3<T> p, V3<T> q) { return p.x * q.x + p.y * q.y + p.z * q.z; }
template <class T>
__device__ __forceinline__ V3<T> cross(V3<T> p, V3<T> q) {
    return V3<T>{p.y * q.z - p.z * q.y, p.z * q.x - p.x * q.z, p.x * q.y - p.y * q.x};
}
template <class T>
__device__ __forceinline__ V3<T> neg(V3<T> p) { return V3<T>{-p.x, -p.y, -p.z}; }
template <class T>
__device__ __forceinline__ T norm(V3<T> p) { return sqrt(dot(p, p)); }

// internal.py:58-60
template <class T>
__device__ __forceinline__ T bond_value(const V3<T>* p, const double* t) { return norm(sub(p[1], p[0], t)); }
// internal.py:63-70
template <class T>
__device__ __forceinline__ T angle_value(const V3<T>* p, const double* t) {
    const V3<T> dx1 = neg(sub(p[1], p[0], t));
    const V3<T> dx2 = sub(p[2], p[1], t + 3);
    return acos(clip1(dot(dx1, dx2) / (norm(dx1) * norm(dx2))));
}
// internal.py:73-80
template <class T>
__device__ __forceinline__ T dihedral_value(const V3<T>* p, const double* t) {
    const V3<T> dx1 = sub(p[1], p[0], t);
    const V3<T> dx2 = sub(p[2], p[1], t + 3);
    const V3<T> dx3 = sub(p[3], p[2], t + 6);
    const V3<T> c12 = cross(dx1, dx2), c23 = cross(dx2, dx3);
    const T numer = dot(dx2, cross(c12, c23));
    const T denom = norm(dx2) * dot(c12, c23);
    return atan2(numer, denom);
}

template <int NA>
__device__ __forceinline__ HDual eval_hd(const V3<HDual>* p, const double* t) {
    if constexpr (NA == 2) return bond_value(p, t);
    else if constexpr (NA == 3) return angle_value(p, t);
    else return dihedral_value(p, t);
}

// mode 0: thread (i, k): q[i] (k == 0), grad[i][k], hvp[i][k] = sum_l H_kl tangent_l (if tangent)
// mode 1: thread (i, k, l): hess[i][k][l]
template <int NA>
__global__ __launch_bounds__(256) void internals_kernel(int nc, int mode, const double* __restrict__ pos,
                                                        const double* __restrict__ tvec,
                                                        const double* __restrict__ tangent,
                                                        double* __restrict__ q, double* __restrict__ grad,
                                                        double* __restrict__ hvp, double* __restrict__ hess) {
    constexpr int NV = 3 * NA;
    const long per = (mode == 0) ? NV : NV * NV;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)nc * per) return;
    const int i = (int)(gid / per);
    const int rem = (int)(gid % per);
    const int k = (mode == 0) ? rem : rem / NV;
    const int l = (mode == 0) ? -1 : rem % NV;
    const double* pi = pos + (size_t)i * NV;
    double tv[9];
#pragma unroll
    for (int e = 0; e < 3 * (NA - 1); ++e) tv[e] = tvec ? tvec[(size_t)i * 3 * (NA - 1) + e] : 0.0;
    V3<HDual> p[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        HDual c[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int e = 3 * a + d;
            c[d].v = pi[e];
            c[d].a = (e == k) ? 1.0 : 0.0;
            c[d].b = (mode == 0) ? (tangent ? tangent[(size_t)i * NV + e] : 0.0) : ((e == l) ? 1.0 : 0.0);
            c[d].c = 0.0;
        }
        p[a] = V3<HDual>{c[0], c[1], c[2]};
    }
    const HDual out = eval_hd<NA>(p, tv);
    if (mode == 0) {
        if (k == 0) q[i] = out.v;
        grad[(size_t)i * NV + k] = out.a;
        if (tangent) hvp[(size_t)i * NV + k] = out.c;
    } else {
        hess[((size_t)i * NV + k) * NV + l] = out.c;
    }
}

template <int NA>
int run_kind(sella_ctx* c, int nc, const double* dpos, const double* dtv, const double* dtan, double* dq,
             double* dgrad, double* dhvp, double* dhess) {
    constexpr int NV = 3 * NA;
    const long n0 = (long)nc * NV;
    // algorithmic bytes: positions + shift vectors (+ tangent) in, value + gradient (+ H t) out
    const double bytes = 8.0 * nc * (NV + 3.0 * (NA - 1) + (dtan ? NV : 0) + 1.0 + NV + (dtan ? NV : 0));
    prof_begin(c, PROF_OTHER, bytes, 0.0);
    SELLA_LAUNCH(c, HIP_KERNEL_NAME(internals_kernel<NA>), dim3((unsigned)((n0 + 255) / 256)), dim3(256), 0,
                 nc, 0, dpos, dtv, dtan, dq, dgrad, dhvp, dhess);
    prof_end(c);
    if (dhess) {
        const long n1 = (long)nc * NV * NV;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(internals_kernel<NA>), dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0,
                           c->stream, nc, 1, dpos, dtv, dtan, dq, dgrad, dhvp, dhess);
    }
    HIPCHK(hipGetLastError());
    return SELLA_OK;
}

}  // namespace
}  // namespace sella

using namespace sella;

extern "C" int sella_internals_eval(sella_ctx* c, int natoms, int nc, const double* pos, const double* tvec,
                                    const double* tangent, double* q, double* grad, double* hvp, double* hess) {
    if (!c || natoms < 2 || natoms > 4 || nc < 0 || !pos || !q || !grad || (tangent && !hvp)) {
        set_error("internals_eval: invalid arguments (natoms must be 2, 3 or 4)");
        return SELLA_E_INVALID;
    }
    if (nc == 0) return SELLA_OK;
    const size_t nv = 3 * (size_t)natoms, ntv = 3 * (size_t)(natoms - 1);
    // one scratch block: pos | tvec | tangent | q | grad | hvp | hess
    const size_t words = (size_t)nc * (nv + ntv + nv + 1 + nv + nv + (hess ? nv * nv : 0)) + 64;
    double* buf;
    SCHK(scratch_get(c, SCR_MISC0, words * sizeof(double), &buf));
    double* dpos = buf;
    double* dtv = dpos + (size_t)nc * nv;
    double* dtan = dtv + (size_t)nc * ntv;
    double* dq = dtan + (size_t)nc * nv;
    double* dgrad = dq + nc;
    double* dhvp = dgrad + (size_t)nc * nv;
    double* dhess = dhvp + (size_t)nc * nv;
    SCHK(h2d_async(c, dpos, pos, (size_t)nc * nv * sizeof(double)));
    if (tvec) SCHK(h2d_async(c, dtv, tvec, (size_t)nc * ntv * sizeof(double)));
    if (tangent) SCHK(h2d_async(c, dtan, tangent, (size_t)nc * nv * sizeof(double)));
    const double* atv = tvec ? dtv : nullptr;
    const double* atan_ = tangent ? dtan : nullptr;
    int st;
    if (natoms == 2) st = run_kind<2>(c, nc, dpos, atv, atan_, dq, dgrad, dhvp, hess ? dhess : nullptr);
    else if (natoms == 3) st = run_kind<3>(c, nc, dpos, atv, atan_, dq, dgrad, dhvp, hess ? dhess : nullptr);
    else st = run_kind<4>(c, nc, dpos, atv, atan_, dq, dgrad, dhvp, hess ? dhess : nullptr);
    SCHK(st);
    SCHK(d2h_async(c, q, dq, (size_t)nc * sizeof(double)));
    SCHK(d2h_async(c, grad, dgrad, (size_t)nc * nv * sizeof(double)));
    if (tangent) SCHK(d2h_async(c, hvp, dhvp, (size_t)nc * nv * sizeof(double)));
    if (hess) SCHK(d2h_async(c, hess, dhess, (size_t)nc * nv * nv * sizeof(double)));
    SCHK(stream_wait(c));
    return SELLA_OK;
}
