// Device-resident rank-two update of a STRUCTURED eigendecomposition  B = lam0 (I - W^T W) + W^T diag(mu) W
// (r explicit pairs, see eigh.hip lr_lowrank_update for the host-planned form) — the per-step quasi-Newton update of
// sella/hessian_update.py:114-126 (TS-BFGS, one secant pair) and sella/linalg.py:274-304 with NO host decision inside:
// the two new directions, the TS-BFGS vectors in coordinates, their rank-one terms, deflation, secular roots,
// Gu/Eisenstat vectors and the ordering of the new spectrum are all decided on the device; the host queues a fixed
// sequence of launches and reads the result back once.
//
// The algebra.  With c_s = W s, c_y = W y and the parts of s, y outside span(W) orthonormalised into two new rows
// e1, e2, every vector of the update lives in span(E), E = [W; e1; e2] (r + 2 rows), and B acts on span(E) as
// D = diag(mu, lam0, lam0) in these coordinates:  s~ = (c_s, |s_perp|, 0), y~ = (c_y, y.e1, y.e2),
//     B s -> D s~,  |B| s -> |D| s~,  j~ = y~ - D s~,  m1 = s.y, m2 = s~.|D| s~,  u~ = (m1 y~ + m2 |D| s~) / (m1^2 + m2^2),
//     z~ = j~ - (j.s / 2) u~,     B+ = B + u z^T + z u^T      (hessian_update.py:120-126),
// so the update is the (r + 2)-dimensional problem  T = D + u~ z~^T + z~ u~^T  = D + sigma_1 p_1 p_1^T + sigma_2 p_2 p_2^T
// (closed form in the plane of u~, z~), solved as two rank-one modifications IN COORDINATES — the eigenvector panel is
// touched once, by the final product  W+ = Q^T E  on the matrix cores.  No n x n object appears: the dense mirror of B
// is rebuilt from (W, mu, lam0) when somebody asks for it (sella_lr_materialize).
//
// Kernels: lr_pre_kernel (one workgroup: coordinates, TS-BFGS scalars, the two rank-one terms), lr_plan_kernel (one
// workgroup: z = Q^T p, ordering, LAPACK dlaed2's deflation rules, Givens rotations on the columns of Q),
// lr_secular_kernel / lr_zhat_kernel (one wavefront per root, secular.h), lr_apply_kernel (one workgroup per new
// eigenpair: its vector, its place in the ascending order, its column of the new Q).
#include "internal.h"
#include <chrono>
#include "secular.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <vector>

namespace sella {
namespace {

constexpr int LR_DEV_MAX = 512;          // rows (r + 2) up to which the coordinate kernels are used
constexpr int LR_SMALL = 128;            // ... and up to which they take their merged forms (fused chain)

__device__ __forceinline__ double blk_sum(double v, double* red) {
    v = wave_sum64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
// two sums with one pair of barriers (each added up in blk_sum's order); red: 8 doubles
__device__ __forceinline__ void blk_sum2(double v1, double v2, double* red, double* o1, double* o2) {
    v1 = wave_sum64(v1);
    v2 = wave_sum64(v2);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = v1; red[4 + (threadIdx.x >> 6)] = v2; }
    __syncthreads();
    *o1 = (red[0] + red[1]) + (red[2] + red[3]);
    *o2 = (red[4] + red[5]) + (red[6] + red[7]);
}
__device__ __forceinline__ double blk_max(double v, double* red) {
    for (int m = 32; m > 0; m >>= 1) v = fmax(v, __shfl_xor(v, m, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// scalar slots of the workspace (doubles)
enum { SC_M1 = 0, SC_M2, SC_JS, SC_SBS, SC_SIG1, SC_SIG2, SC_KEEP1, SC_KEEP2, SC_FAIL, SC_GPERP2 = 16, SC_N = 32 };
// gram slots: host / device Gram of the input rows and of the residual rows
enum { G_SS = 0, G_SY, G_YY, G_A11 = 8, G_A12, G_A22 = 11, G_R0G = 12, G_R1G = 13 };

struct PreArgs {
    int r, nr, ldr, mode;                 // mode 0: rows are (s, y) -> TS-BFGS; 1: rows are (u, z) themselves
    const double *C, *C2, *G, *mu;        // C[h * ldr + i], C2[h * ldr + i]
    const double* C3;                     // negated coefficients of the clean-up sweep of the second new row (r + 1 entries)
    double* row2;                         // that row (n entries): measured, normalised (or zeroed) here
    int n;
    double lam0;
    double *sc, *ec, *UZ, *P, *D;
    // fused chain: |row2|^2 arrives as per-workgroup partials of the clean-up launch (null: measured here), and the
    // components of a third row g along the rows of E are assembled from dots the earlier launches left (eg, nr entries):
    // CG = the negated W g (row 2 of C), R0.g / R1.g in the Gram block
    const double* NP = nullptr;
    int parts = 0;
    const double* CG = nullptr;
    double* eg = nullptr;
};

// Fast form for nr <= 256 (one coordinate per thread): every global operand — the coefficient rows, mu, the Gram scalars,
// the partial sums, the second new row itself — is loaded ONCE, up front, all loads in flight together; s~, y~, u~, z~ stay
// in registers between the three block-wide reductions.  The general form re-read them from global memory behind every
// barrier: nine dependent load phases of 1.5 - 2 us in a kernel of 14.  shP / shD (LDS, may be null): p_1 and D for the
// plan that follows in the same launch.  Same arithmetic, same summation orders as the general form.
constexpr int LR_ROW2_REGS = 16;           // the second row rides in registers up to 256 x 16 = 4096 entries
__device__ __forceinline__ void lr_pre_fast(const PreArgs& a, double* shP, double* shD) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    __shared__ double red[8];
    const int tid = threadIdx.x, r = a.r, nr = a.nr, ldr = a.ldr;
    // ---- every load of the kernel ----
    double c0i = 0.0, c20i = 0.0, c1i = 0.0, c21i = 0.0, c3i = 0.0, mui = 0.0, cgi = 0.0;
    if (tid < r) {
        c0i = a.C[tid]; c20i = a.C2[tid]; c1i = a.C[ldr + tid]; c21i = a.C2[ldr + tid]; c3i = a.C3[tid]; mui = a.mu[tid];
        if (a.eg) cgi = a.CG[tid];
    }
    const double c3r = a.C3[r];
    const double ss = a.G[G_SS], sy = a.G[G_SY], yy = a.G[G_YY];
    const double a11 = a.G[G_A11], a12 = a.G[G_A12];
    const double r0g = a.eg ? a.G[G_R0G] : 0.0;
    double rv[LR_ROW2_REGS];
#pragma unroll
    for (int q = 0; q < LR_ROW2_REGS; ++q) {
        const int i = tid + 256 * q;
        rv[q] = (i < a.n) ? a.row2[i] : 0.0;
    }
    double rho2sq = 0.0, row2g = 0.0;
    if (a.NP) {
        for (int p = 0; p < a.parts; ++p) { rho2sq += a.NP[2 * p]; row2g += a.NP[2 * p + 1]; }
    } else {
        double pr = 0.0;
#pragma unroll
        for (int q = 0; q < LR_ROW2_REGS; ++q) pr += rv[q] * rv[q];
        rho2sq = blk_sum(pr, red);
    }
    const bool keep1 = ss > 0.0 && a11 > 1e-26 * ss;
    const double r11 = keep1 ? sqrt(a11) : 0.0;
    const bool keep2 = yy > 0.0 && rho2sq > 1e-26 * yy;
    const double r22 = keep2 ? sqrt(rho2sq) : 0.0;
    const double inv22 = keep2 ? 1.0 / r22 : 0.0;
#pragma unroll
    for (int q = 0; q < LR_ROW2_REGS; ++q) {
        const int i = tid + 256 * q;
        if (i < a.n) a.row2[i] = rv[q] * inv22;
    }
    const double y1 = keep1 ? a12 / r11 - c3r : 0.0;
    if (tid == 0) {
        a.sc[SC_KEEP1] = keep1 ? 1.0 : 0.0;
        a.sc[SC_KEEP2] = keep2 ? 1.0 : 0.0;
        a.sc[SC_FAIL] = 0.0;
        a.sc[SC_GPERP2] = 0.0;
    }
    if (a.eg) {
        if (tid < r) a.eg[tid] = -cgi;
        if (tid == 0) {
            a.eg[r] = keep1 ? r0g / r11 : 0.0;
            a.eg[r + 1] = keep2 ? row2g * inv22 : 0.0;
        }
    }
    const bool own = tid < nr;
    const double di = tid < r ? mui : a.lam0;
    const double si = tid < r ? -(c0i + c20i) : (tid == r ? r11 : 0.0);
    const double yi = tid < r ? -((c1i + c21i) + c3i) : (tid == r ? y1 : r22);
    double c0 = 0.0, c1 = 0.0, c2 = 0.0;
    if (a.mode == 0) {
        double m2, sBs;
        blk_sum2(own ? fabs(di) * si * si : 0.0, own ? di * si * si : 0.0, red, &m2, &sBs);
        const double m1 = sy, js = sy - sBs;
        const double g = m1 * m1 + m2 * m2;
        const double gp = (g > 0.0) ? 1.0 / g : 0.0;
        c0 = gp * m1;
        c1 = gp * m2;
        c2 = -0.5 * js;
        if (tid == 0) { a.sc[SC_M1] = m1; a.sc[SC_M2] = m2; a.sc[SC_JS] = js; a.sc[SC_SBS] = sBs; }
    }
    double ui = 0.0, zi = 0.0;
    if (own) {
        if (a.mode == 0) {
            ui = c0 * yi + c1 * fabs(di) * si;
            zi = (yi - di * si) + c2 * ui;
        } else {
            ui = si;
            zi = yi;
        }
        a.UZ[2 * tid] = ui;
        a.UZ[2 * tid + 1] = zi;
        a.D[tid] = di;
        if (shD) shD[tid] = di;
    }
    double uu, uz;
    blk_sum2(ui * ui, ui * zi, red, &uu, &uz);
    const double proj = uu > 0.0 ? uz / uu : 0.0;
    const double zp = zi - proj * ui;
    const double n2 = blk_sum(own ? zp * zp : 0.0, red);
    const double un = sqrt(uu), zn = sqrt(n2), s = un * zn, b = uz;
    double sig[2] = {0.0, 0.0}, w1[2] = {0.0, 0.0}, w2[2] = {0.0, 0.0};        // p_t = w1[t] f1 + w2[t] f2
    if (uu > 0.0) {
        if (s > 0.0) {
            const double root = sqrt(b * b + s * s);
            if (b >= 0.0) { sig[0] = b + root; sig[1] = -(s * s) / sig[0]; }
            else { sig[1] = b - root; sig[0] = -(s * s) / sig[1]; }
            for (int t = 0; t < 2; ++t) {
                const double nrm = sqrt(sig[t] * sig[t] + s * s);
                w1[t] = sig[t] / nrm;
                w2[t] = s / nrm;
            }
        } else {
            sig[0] = 2.0 * b;
            w1[0] = 1.0;
        }
    }
    if (own) {
        const double f1 = uu > 0.0 ? ui / un : 0.0;
        const double f2 = zn > 0.0 ? zp / zn : 0.0;
        const double p1 = w1[0] * f1 + w2[0] * f2;
        a.P[tid] = p1;
        a.P[ldr + tid] = w1[1] * f1 + w2[1] * f2;
        if (shP) shP[tid] = p1;
    }
    if (tid == 0) { a.sc[SC_SIG1] = sig[0]; a.sc[SC_SIG2] = sig[1]; }
}

__device__ __forceinline__ void lr_pre_body(const PreArgs& a, double* shP = nullptr, double* shD = nullptr) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    if (a.nr <= 256 && a.n <= 256 * LR_ROW2_REGS) { lr_pre_fast(a, shP, shD); return; }
    __shared__ double red[4];
    const int tid = threadIdx.x, r = a.r, nr = a.nr;
    const double ss = a.G[G_SS], sy = a.G[G_SY], yy = a.G[G_YY];
    const double a11 = a.G[G_A11], a12 = a.G[G_A12];
    // The two new rows.  e1 = s_perp / |s_perp| was written by lr_e1_kernel; the second arrives as
    // y_perp - (a12 / a11) s_perp after one more sweep against [W; e1]: the two residuals are often parallel (always in
    // a view, where both update vectors leave span(W) along one direction), what is left is then a difference of nearly
    // equal vectors, and only the explicit sweep makes it orthogonal to the rest at roundoff.  Its norm is measured
    // here, on the vector itself.  A part below 1e-13 of the original vector is dropped (the thresholds of
    // math.pyx:112-117 as gs.hip uses them): the row is zeroed and takes no part.
    const bool keep1 = ss > 0.0 && a11 > 1e-26 * ss;
    const double r11 = keep1 ? sqrt(a11) : 0.0;
    double rho2sq;
    double row2g = 0.0;
    if (a.NP) {
        rho2sq = 0.0;
        for (int p = 0; p < a.parts; ++p) { rho2sq += a.NP[2 * p]; row2g += a.NP[2 * p + 1]; }
    } else {
        double pr = 0.0;
        for (int i = tid; i < a.n; i += 256) pr += a.row2[i] * a.row2[i];
        rho2sq = blk_sum(pr, red);
    }
    const bool keep2 = yy > 0.0 && rho2sq > 1e-26 * yy;
    const double r22 = keep2 ? sqrt(rho2sq) : 0.0;
    const double inv22 = keep2 ? 1.0 / r22 : 0.0;
    for (int i = tid; i < a.n; i += 256) a.row2[i] *= inv22;
    const double y1 = keep1 ? a12 / r11 - a.C3[r] : 0.0;             // (C3 negated, like C and C2)
    if (tid == 0) {
        a.sc[SC_KEEP1] = keep1 ? 1.0 : 0.0;
        a.sc[SC_KEEP2] = keep2 ? 1.0 : 0.0;
        a.sc[SC_FAIL] = 0.0;
        a.sc[SC_GPERP2] = 0.0;
    }
    if (a.eg) {
        // E g: along the old rows measured by the first launch; along e1 = R0 / r11 from the dot R0.g of the second sweep;
        // along e2 as measured on the cleaned row itself by the clean-up launch
        const double r0g = a.G[G_R0G];
        for (int i = tid; i < r; i += 256) a.eg[i] = -a.CG[i];
        if (tid == 0) {
            a.eg[r] = keep1 ? r0g / r11 : 0.0;
            a.eg[r + 1] = keep2 ? row2g * inv22 : 0.0;
        }
    }
    auto Dof = [&](int i) { return i < r ? a.mu[i] : a.lam0; };
    auto s_of = [&](int i) { return i < r ? -(a.C[i] + a.C2[i]) : (i == r ? r11 : 0.0); };      // (C, C2: negated W x)
    auto y_of = [&](int i) { return i < r ? -((a.C[a.ldr + i] + a.C2[a.ldr + i]) + a.C3[i]) : (i == r ? y1 : r22); };
    double c0 = 0.0, c1 = 0.0, c2 = 0.0;
    if (a.mode == 0) {
        double p2 = 0.0, p3 = 0.0;
        for (int i = tid; i < nr; i += 256) {
            const double si = s_of(i), di = Dof(i);
            p2 += fabs(di) * si * si;
            p3 += di * si * si;
        }
        const double m2 = blk_sum(p2, red), sBs = blk_sum(p3, red);
        const double m1 = sy, js = sy - sBs;
        const double g = m1 * m1 + m2 * m2;
        const double gp = (g > 0.0) ? 1.0 / g : 0.0;
        c0 = gp * m1;
        c1 = gp * m2;
        c2 = -0.5 * js;
        if (tid == 0) { a.sc[SC_M1] = m1; a.sc[SC_M2] = m2; a.sc[SC_JS] = js; a.sc[SC_SBS] = sBs; }
    }
    // u~, z~ (kept in UZ: the coefficient matrix of u, z over the rows of E, nr x 2)
    double pa = 0.0, pb = 0.0;
    for (int i = tid; i < nr; i += 256) {
        const double si = s_of(i), yi = y_of(i);
        double ui, zi;
        if (a.mode == 0) {
            const double di = Dof(i);
            ui = c0 * yi + c1 * fabs(di) * si;
            zi = (yi - di * si) + c2 * ui;
        } else {
            ui = si;
            zi = yi;
        }
        a.UZ[2 * i] = ui;
        a.UZ[2 * i + 1] = zi;
        a.D[i] = Dof(i);
        pa += ui * ui;
        pb += ui * zi;
    }
    const double uu = blk_sum(pa, red), uz = blk_sum(pb, red);
    // u z^T + z u^T in the plane (f1, f2), f1 = u / |u|, f2 = z_perp / |z_perp|:  [[2b, s], [s, 0]], b = u.z, s = |u| |z_perp|
    double pn = 0.0;
    const double proj = uu > 0.0 ? uz / uu : 0.0;
    for (int i = tid; i < nr; i += 256) {
        const double zp = a.UZ[2 * i + 1] - proj * a.UZ[2 * i];
        pn += zp * zp;
    }
    const double n2 = blk_sum(pn, red);
    const double un = sqrt(uu), zn = sqrt(n2), s = un * zn, b = uz;
    double sig[2] = {0.0, 0.0}, w1[2] = {0.0, 0.0}, w2[2] = {0.0, 0.0};        // p_t = w1[t] f1 + w2[t] f2
    if (uu > 0.0) {
        if (s > 0.0) {
            const double root = sqrt(b * b + s * s);           // (the smaller root from the product, -s^2: no cancellation)
            if (b >= 0.0) { sig[0] = b + root; sig[1] = -(s * s) / sig[0]; }
            else { sig[1] = b - root; sig[0] = -(s * s) / sig[1]; }
            for (int t = 0; t < 2; ++t) {
                const double nrm = sqrt(sig[t] * sig[t] + s * s);
                w1[t] = sig[t] / nrm;
                w2[t] = s / nrm;
            }
        } else {
            sig[0] = 2.0 * b;
            w1[0] = 1.0;
        }
    }
    for (int i = tid; i < nr; i += 256) {
        const double f1 = uu > 0.0 ? a.UZ[2 * i] / un : 0.0;
        const double f2 = zn > 0.0 ? (a.UZ[2 * i + 1] - proj * a.UZ[2 * i]) / zn : 0.0;
        a.P[i] = w1[0] * f1 + w2[0] * f2;
        a.P[a.ldr + i] = w1[1] * f1 + w2[1] * f2;
    }
    if (tid == 0) { a.sc[SC_SIG1] = sig[0]; a.sc[SC_SIG2] = sig[1]; }
}

__device__ __forceinline__ void lr_pre_vb(const VB vb, PreArgs a) { lr_pre_body(a); }
__global__ __launch_bounds__(256) void lr_pre_kernel(PreArgs a) { lr_pre_vb(vb_hw(), a); }

struct PlanArgs {
    int fill = 0;                         // with `first`: Q is NOT initialised — the kernel writes the identity itself if (and
                                          // only if) it has rotations to apply to it; lr_apply_kernel reads cnt[1] to know
    int nr, ldr, ldq, first;              // first: Q is the identity (z = p)
    const double* p;
    const double* sigma;                  // device scalar
    const double* Dcur;
    double* Q;                            // nr x nr, columns = current eigenvectors in coordinates (rotated in place)
    double *z, *Dp, *zz, *Dd, *wd, *cs, *pl;
    int *perm, *nd, *df, *i1, *i2, *cnt;
};

// One workgroup: weights of the term in the current eigenbasis, ascending order of the (signed) poles, deflation by
// LAPACK dlaed2's rules (a negligible weight; of two nearly equal poles one rotated out), the rotations applied to the
// columns of Q, the compact secular problem.  A negative sigma is solved as -(-D + |sigma| z z^T).  The sequential part
// (one thread walks the poles in order) works on LDS copies: a chain of dependent global loads costs ~0.5 us a link.
__device__ __forceinline__ void lr_plan_body(const PlanArgs& a) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    __shared__ double red[4];
    __shared__ double sD[LR_DEV_MAX], sZ[LR_DEV_MAX], sP[LR_DEV_MAX], sCS[2 * LR_DEV_MAX], sZs[LR_DEV_MAX];
    __shared__ int sPerm[LR_DEV_MAX], sNd[LR_DEV_MAX], sDf[LR_DEV_MAX], sI1[LR_DEV_MAX], sI2[LR_DEV_MAX];
    __shared__ int sK, sRot, sNdf;
    __shared__ double sZq[4][64];
    const int tid = threadIdx.x, nr = a.nr;
    const double sigma = a.sigma[0];
    for (int i = tid; i < nr; i += 256) { sP[i] = a.p[i]; sD[i] = a.Dcur[i]; }     // both vectors in flight together
    __syncthreads();
    double pz = 0.0;
    if (a.first) {
        for (int i = tid; i < nr; i += 256) {
            const double zi = sP[i];
            sZ[i] = zi;
            pz += zi * zi;
        }
    } else if (nr <= 128) {
        // z = Q^T p for up to 128 rows: lane -> columns lane and 64 + lane, wavefront w -> the w-th quarter of the summation
        // index, EVERY load of the product in flight at once (two passes of two batches of sixteen were four dependent load
        // phases of a 19 us kernel); the quarters meet in LDS once, summed in the order of the general form below
        constexpr int QB = 32;
        __shared__ double sZq2[2][4][64];
        const int lane = tid & 63, wave = tid >> 6;
        const int chunk = (nr + 3) / 4, kb = wave * chunk, ke = (kb + chunk < nr) ? kb + chunk : nr;
        const int il0 = lane < nr ? lane : nr - 1, il1 = 64 + lane < nr ? 64 + lane : nr - 1;
        const bool two = nr > 64;
        double q0[QB], q1[QB];
#pragma unroll
        for (int u = 0; u < QB; ++u) {
            const int k = kb + u, kc = k < nr ? k : nr - 1;
            q0[u] = a.Q[(size_t)kc * a.ldq + il0];
            q1[u] = two ? a.Q[(size_t)kc * a.ldq + il1] : 0.0;
        }
        double za[4] = {0.0, 0.0, 0.0, 0.0}, zb[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < QB; ++u) {
            const int k = kb + u;
            const double pk = k < ke ? sP[k] : 0.0;
            za[u & 3] += k < ke ? q0[u] * pk : 0.0;
            zb[u & 3] += k < ke ? q1[u] * pk : 0.0;
        }
        sZq2[0][wave][lane] = (za[0] + za[1]) + (za[2] + za[3]);
        sZq2[1][wave][lane] = (zb[0] + zb[1]) + (zb[2] + zb[3]);
        __syncthreads();
        if (wave == 0) {
            for (int ps = 0; ps < (two ? 2 : 1); ++ps) {
                const int i = 64 * ps + lane;
                if (i < nr) {
                    const double zi = (sZq2[ps][0][lane] + sZq2[ps][1][lane]) + (sZq2[ps][2][lane] + sZq2[ps][3][lane]);
                    sZ[i] = zi;
                    pz += zi * zi;
                }
            }
        }
        __syncthreads();
    } else {
        // z = Q^T p: lane -> column i (64 at a time), wavefront w -> the w-th quarter of the summation index, sixteen
        // loads in flight at a time (a plain loop waits for every load in turn); the quarters meet in LDS in a fixed order
        const int lane = tid & 63, wave = tid >> 6;
        const int chunk = (nr + 3) / 4, kb = wave * chunk, ke = (kb + chunk < nr) ? kb + chunk : nr;
        for (int i0 = 0; i0 < nr; i0 += 64) {
            const int i = i0 + lane, il = i < nr ? i : nr - 1;
            double z0 = 0.0, z1 = 0.0, z2 = 0.0, z3 = 0.0;
            for (int k0 = kb; k0 < ke; k0 += 16) {
                double qv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int k = k0 + u;
                    qv[u] = a.Q[(size_t)(k < ke ? k : ke - 1) * a.ldq + il];
                }
#pragma unroll
                for (int u = 0; u < 16; u += 4) {
                    auto term = [&](int v) -> double { const int k = k0 + v; return k < ke ? qv[v] * sP[k] : 0.0; };
                    z0 += term(u);
                    z1 += term(u + 1);
                    z2 += term(u + 2);
                    z3 += term(u + 3);
                }
            }
            sZq[wave][lane] = (z0 + z1) + (z2 + z3);
            __syncthreads();
            if (wave == 0 && i < nr) {
                const double zi = (sZq[0][lane] + sZq[1][lane]) + (sZq[2][lane] + sZq[3][lane]);
                sZ[i] = zi;
                pz += zi * zi;
            }
            __syncthreads();
        }
    }
    const double zn2 = blk_sum(pz, red);
    const double sgn = sigma < 0.0 ? -1.0 : 1.0;
    const bool idle = !(sigma != 0.0) || !(zn2 > 0.0);
    const double zn = idle ? 1.0 : sqrt(zn2);
    double dmax = 0.0, zmax = 0.0;
    for (int i = tid; i < nr; i += 256) {
        const double d = sgn * sD[i], w = sZ[i] / zn;
        sD[i] = d;
        sZ[i] = w;
        dmax = fmax(dmax, fabs(d));
        zmax = fmax(zmax, fabs(w));
    }
    dmax = blk_max(dmax, red);
    zmax = blk_max(zmax, red);
    __syncthreads();
    // stable ascending order of the poles: perm[rank] = index
    for (int i = tid; i < nr; i += 256) {
        const double di = sD[i];
        int rank = 0;
        for (int j = 0; j < nr; ++j) {
            const double dj = sD[j];
            rank += (dj < di || (dj == di && j < i)) ? 1 : 0;
        }
        sPerm[rank] = i;
    }
    __syncthreads();
    const double rho = fabs(sigma) * zn2;
    // the walk below reads poles and weights IN ORDER from sorted copies (sP is free by now: z is formed)
    double* sDs = sP;                                        // p is consumed (z is formed)
    for (int jj = tid; jj < nr; jj += 256) { sDs[jj] = sD[sPerm[jj]]; sZs[jj] = sZ[sPerm[jj]]; }
    __syncthreads();
    // The usual case needs no walk at all: no negligible weight and no pair of neighbouring poles close enough to rotate
    // means every pole stays, in order.  Each thread tests its own position against its left neighbour (with nothing
    // deflated in between the neighbour IS the pending pole of the walk); one block-wide maximum decides.
    bool plain = false;
    {
        const double tol0 = 8.0 * 2.220446049250313e-16 * fmax(dmax, zmax);
        double flag = (idle || rho * zmax <= tol0) ? 1.0 : 0.0;
        for (int jj = tid; jj < nr; jj += 256) {
            const double zj = sZs[jj];
            if (rho * fabs(zj) <= tol0) flag = 1.0;
            if (jj > 0) {
                const double zq = sZs[jj - 1], t = sDs[jj] - sDs[jj - 1];
                if (fabs(t * zj * zq) <= tol0 * (zj * zj + zq * zq)) flag = 1.0;
            }
        }
        plain = !(blk_max(flag, red) > 0.0);
        __syncthreads();
    }
    if (plain) {
        for (int jj = tid; jj < nr; jj += 256) sNd[jj] = sPerm[jj];
        if (tid == 0) {
            sK = nr; sRot = 0; sNdf = 0;
            a.cnt[0] = nr;
            a.cnt[1] = 0;
            a.pl[0] = rho;
            a.pl[1] = sgn;
        }
    } else if (tid == 0) {
        const double eps = 2.220446049250313e-16;
        const double tol = 8.0 * eps * fmax(dmax, zmax);
        int K = 0, nd = 0, nrot = 0;
        if (idle || rho * zmax <= tol) {
            for (int jj = 0; jj < nr; ++jj) sDf[nd++] = sPerm[jj];
        } else {
            // One thread, the poles in ascending order, dlaed2's rules.  The pending pole (position pjj) lives in
            // registers; the closeness test |t c s| <= tol with c = cc / tau, s = -ss / tau is taken in the form
            // |t cc ss| <= tol (cc^2 + ss^2), so the common case (distinct poles) costs no square root and no division —
            // with hypot and two divisions per pole this walk was 25 us at 60 poles, most of a quasi-Newton update.
            int pjj = -1;
            double dp = 0.0, zp = 0.0;
            for (int jj = 0; jj < nr; ++jj) {
                const double zj = sZs[jj], dj = sDs[jj];
                if (rho * fabs(zj) <= tol) { sDf[nd++] = sPerm[jj]; continue; }
                if (pjj < 0) { pjj = jj; dp = dj; zp = zj; continue; }
                const double t = dj - dp;
                if (fabs(t * zj * zp) <= tol * (zj * zj + zp * zp)) {
                    const double tau = hypot(zj, zp);
                    const double cc = zj / tau, sn = -zp / tau;
                    const int pj = sPerm[pjj], nj = sPerm[jj];
                    sI1[nrot] = pj;
                    sI2[nrot] = nj;
                    sCS[2 * nrot] = cc;
                    sCS[2 * nrot + 1] = sn;
                    ++nrot;
                    sD[pj] = dp * cc * cc + dj * sn * sn;
                    sZ[pj] = 0.0;
                    sDf[nd++] = pj;
                    dp = dp * sn * sn + dj * cc * cc;
                    zp = tau;
                    pjj = jj;
                } else {
                    const int pj = sPerm[pjj];
                    sD[pj] = dp;
                    sZ[pj] = zp;
                    sNd[K++] = pj;
                    pjj = jj; dp = dj; zp = zj;
                }
            }
            if (pjj >= 0) {
                const int pj = sPerm[pjj];
                sD[pj] = dp;
                sZ[pj] = zp;
                sNd[K++] = pj;
            }
        }
        sK = K;
        sRot = nrot;
        sNdf = nd;
        a.cnt[0] = K;
        a.cnt[1] = nrot;
        a.pl[0] = rho;
        a.pl[1] = sgn;
    }
    __syncthreads();
    const int K = sK, nrot = sRot, ndf = sNdf;
    if (a.first && a.fill && nrot > 0) {
        for (int e = tid; e < nr * a.ldq; e += 256) a.Q[e] = ((e / a.ldq) == (e % a.ldq)) ? 1.0 : 0.0;
        __syncthreads();
    }
    // rotations on column pairs of Q (x' = c x + s y, y' = c y - s x: the row rotations of eigh.hip, transposed)
    for (int q = 0; q < nrot; ++q) {
        const int c1 = sI1[q], c2 = sI2[q];
        const double cc = sCS[2 * q], s = sCS[2 * q + 1];
        for (int k = tid; k < nr; k += 256) {
            double* row = a.Q + (size_t)k * a.ldq;
            const double x = row[c1], y = row[c2];
            row[c1] = cc * x + s * y;
            row[c2] = cc * y - s * x;
        }
        __syncthreads();
    }
    for (int i = tid; i < nr; i += 256) a.Dp[i] = sD[i];
    for (int p = tid; p < K; p += 256) {
        a.nd[p] = sNd[p];
        a.Dd[p] = sD[sNd[p]];
        a.wd[p] = sZ[sNd[p]];
    }
    for (int p = tid; p < ndf; p += 256) a.df[p] = sDf[p];
}

__device__ __forceinline__ void lr_plan_vb(const VB vb, PlanArgs a) { lr_plan_body(a); }
__global__ __launch_bounds__(256) void lr_plan_kernel(PlanArgs a) { lr_plan_vb(vb_hw(), a); }

// the first term's plan needs nothing but what lr_pre_kernel leaves: one workgroup does both (what the first writes to
// global memory is visible to the whole workgroup behind the barrier)
__device__ __forceinline__ void lr_pre_plan_vb(const VB vb, PreArgs pa, PlanArgs pl) {
    // (fast form: the plan reads p_1 and D where the first half left them in LDS instead of through L2)
    __shared__ double shP[256], shD[256];
    const bool fast = pa.nr <= 256 && pa.n <= 256 * LR_ROW2_REGS;
    lr_pre_body(pa, fast ? shP : nullptr, fast ? shD : nullptr);
    __syncthreads();
    if (fast) { pl.p = shP; pl.Dcur = shD; }
    lr_plan_body(pl);
}
__global__ __launch_bounds__(256) void lr_pre_plan_kernel(PreArgs pa, PlanArgs pl) { lr_pre_plan_vb(vb_hw(), pa, pl); }

struct WaveSum2 {
    __device__ double operator()(double v) const { return wave_sum64(v); }
};
struct WaveProd2 {
    __device__ double operator()(double v) const {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) v *= __shfl_xor(v, m, 64);
        return v;
    }
};

// one wavefront per root; K and rho come from the plan kernel
__device__ __forceinline__ void lr_secular_vb(const VB vb, const int* __restrict__ cnt, const double* __restrict__ pl,
                                                         const double* __restrict__ D, const double* __restrict__ w,
                                                         double* __restrict__ tau, int* __restrict__ org,
                                                         double* __restrict__ lam, double* __restrict__ fail) {
    const int K = cnt[0];
    const int j = vb.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (j >= K) return;
    int o;
    double t;
    const int it = secular::solve_root(K, D, w, pl[0], j, &o, &t, lane, 64, WaveSum2());
    if (lane == 0) {
        tau[j] = t;
        org[j] = o;
        lam[j] = D[o] + t;
        if (it < 0) fail[0] = j + 1.0;
    }
}
__global__ __launch_bounds__(256) void lr_secular_kernel(const int* __restrict__ cnt, const double* __restrict__ pl,
                                                         const double* __restrict__ D, const double* __restrict__ w,
                                                         double* __restrict__ tau, int* __restrict__ org,
                                                         double* __restrict__ lam, double* __restrict__ fail) { lr_secular_vb(vb_hw(), cnt, pl, D, w, tau, org, lam, fail); }

__device__ __forceinline__ void lr_zhat_vb(const VB vb, const int* __restrict__ cnt, const double* __restrict__ D,
                                                      const double* __restrict__ w, const double* __restrict__ tau,
                                                      const int* __restrict__ org, double* __restrict__ zh) {
    const int K = cnt[0];
    const int i = vb.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= K) return;
    const double z = secular::zhat(K, D, w, tau, org, i, lane, 64, WaveProd2());
    if (lane == 0) zh[i] = z;
}
__global__ __launch_bounds__(256) void lr_zhat_kernel(const int* __restrict__ cnt, const double* __restrict__ D,
                                                      const double* __restrict__ w, const double* __restrict__ tau,
                                                      const int* __restrict__ org, double* __restrict__ zh) { lr_zhat_vb(vb_hw(), cnt, D, w, tau, org, zh); }

struct ApplyArgs {
    int nr, ldq;
    const int *cnt, *nd, *df, *org;
    const double *pl, *Dd, *Dp, *zh, *tau, *lam;
    const double* Qin;
    double *Qout, *Dnext;
    int first = 0;                              // Qin is the identity unless the plan had to rotate it (cnt[1] > 0): then never read
    const double* wd = nullptr;                 // merged form: the Gu / Eisenstat weights are computed here, by every
                                                // workgroup for itself (K <= LR_SMALL), instead of by a launch of their own
};

// Workgroup j: new eigenpair j of the term — an updated one (j < K: Gu/Eisenstat vector over the non-deflated columns)
// or a deflated one (copied) — goes to its place in the ascending order of the new spectrum.
__device__ __forceinline__ void lr_apply_vb(const VB vb, ApplyArgs a) {
    __shared__ double red[4];
    __shared__ double u[LR_DEV_MAX];
    __shared__ double zhs[LR_SMALL + 8];
    const int tid = threadIdx.x, nr = a.nr, j = vb.x;
    const int K = a.cnt[0];
    const bool ident = a.first && a.cnt[1] == 0;
    if (a.wd && j < K) {
        const int lane = tid & 63, wave = tid >> 6;
        for (int i = wave; i < K; i += 4) {
            const double z = secular::zhat(K, a.Dd, a.wd, a.tau, a.org, i, lane, 64, WaveProd2());
            if (lane == 0) zhs[i] = z;
        }
        __syncthreads();
    }
    const double* zh = a.wd ? zhs : a.zh;
    const double sgn = a.pl[1];
    auto value = [&](int t) { return sgn * (t < K ? a.lam[t] : a.Dp[a.df[t - K]]); };
    const double vj = value(j);
    double cntl = 0.0;
    for (int t = tid; t < nr; t += 256) {
        const double vt = value(t);
        cntl += (vt < vj || (vt == vj && t < j)) ? 1.0 : 0.0;
    }
    const int pos = (int)(blk_sum(cntl, red) + 0.5);
    if (tid == 0) a.Dnext[pos] = vj;
    if (j >= K) {
        const int src = a.df[j - K];
        if (ident) for (int k = tid; k < nr; k += 256) a.Qout[(size_t)k * a.ldq + pos] = (k == src) ? 1.0 : 0.0;
        else for (int k = tid; k < nr; k += 256) a.Qout[(size_t)k * a.ldq + pos] = a.Qin[(size_t)k * a.ldq + src];
        return;
    }
    const double Do = a.Dd[a.org[j]], tj = a.tau[j];
    double ssq = 0.0;
    for (int i = tid; i < K; i += 256) {
        const double ui = zh[i] / ((a.Dd[i] - Do) - tj);
        u[i] = ui;
        ssq += ui * ui;
    }
    const double inv = 1.0 / sqrt(blk_sum(ssq, red));
    if (ident) {
        // columns of the identity: row nd[i] of the new vector is u[i], every other row zero
        for (int k = tid; k < nr; k += 256) a.Qout[(size_t)k * a.ldq + pos] = 0.0;
        __syncthreads();
        for (int i = tid; i < K; i += 256) a.Qout[(size_t)a.nd[i] * a.ldq + pos] = u[i] * inv;
        return;
    }
    for (int k = tid; k < nr; k += 256) {
        const double* row = a.Qin + (size_t)k * a.ldq;
        double acc = 0.0;
        for (int i = 0; i < K; ++i) acc += row[a.nd[i]] * u[i];
        a.Qout[(size_t)k * a.ldq + pos] = acc * inv;
    }
}
__global__ __launch_bounds__(256) void lr_apply_kernel(ApplyArgs a) { lr_apply_vb(vb_hw(), a); }

// e1 = R0 / |R0| -> out row 0;  R1 - (R0.R1 / R0.R0) R0 -> out row 1 (unnormalised; cleaned and measured afterwards).
// Every workgroup derives the two scalars from the Gram matrix of the residual rows itself.
__device__ __forceinline__ void lr_e1_vb(const VB vb, const double* __restrict__ R, int ldr_, int n,
                                                    const double* __restrict__ G, double* __restrict__ out, int ldo) {
    const int i = vb.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double ss = G[G_SS], a11 = G[G_A11], a12 = G[G_A12];
    const bool keep1 = ss > 0.0 && a11 > 1e-26 * ss;
    const double inv = keep1 ? 1.0 / sqrt(a11) : 0.0, f = keep1 ? a12 / a11 : 0.0;
    const double r0 = R[i], r1 = R[ldr_ + i];
    out[i] = r0 * inv;
    out[ldo + i] = r1 - f * r0;
}
__global__ __launch_bounds__(256) void lr_e1_kernel(const double* __restrict__ R, int ldr_, int n,
                                                    const double* __restrict__ G, double* __restrict__ out, int ldo) { lr_e1_vb(vb_hw(), R, ldr_, n, G, out, ldo); }

// R_h -= sum_j C[h * ldc + j] W_j for both residual rows in ONE pass over W (C holds the negated coefficients: added)
__device__ __forceinline__ void lr_proj2_vb(const VB vb, const double* __restrict__ W, int ldw, int r, int n,
                                                       const double* __restrict__ C, int ldc, double* __restrict__ R,
                                                       int ldr_) {
    __shared__ double c0[LR_DEV_MAX], c1[LR_DEV_MAX];
    for (int j = threadIdx.x; j < r; j += 256) { c0[j] = C[j]; c1[j] = C[ldc + j]; }
    __syncthreads();
    const int i = vb.x * 256 + threadIdx.x;
    if (i >= n) return;
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
    int j = 0;
    for (; j + 1 < r; j += 2) {
        const double w0 = W[(size_t)j * ldw + i], w1 = W[(size_t)(j + 1) * ldw + i];
        a0 += c0[j] * w0;
        b0 += c0[j + 1] * w1;
        a1 += c1[j] * w0;
        b1 += c1[j + 1] * w1;
    }
    if (j < r) {
        const double w0 = W[(size_t)j * ldw + i];
        a0 += c0[j] * w0;
        a1 += c1[j] * w0;
    }
    R[i] += a0 + b0;
    R[ldr_ + i] += a1 + b1;
}
__global__ __launch_bounds__(256) void lr_proj2_kernel(const double* __restrict__ W, int ldw, int r, int n,
                                                       const double* __restrict__ C, int ldc, double* __restrict__ R,
                                                       int ldr_) { lr_proj2_vb(vb_hw(), W, ldw, r, n, C, ldc, R, ldr_); }


// ---- the fused chain (option lr_chain, default): the O(n r) passes of a job as five launches ----------------------------
// Every pass works on chunks of 64 coordinates (lane = coordinate; the four wavefronts of a workgroup split the rows of W
// four ways and meet in LDS): it applies the coefficients of the previous pass (summed from that pass's per-workgroup
// partials, in a fixed order, by every workgroup itself) and leaves the partial dots the next pass needs.  So a
// Gram-Schmidt sweep is ONE launch (projection + the dots of the next sweep) instead of two, and no launch exists only to
// reduce something.  n / 64 workgroups: up to 64 x 64 = 4096 coordinates (larger jobs take the chain of round 3).
constexpr int LR_CHUNK = 64;
constexpr int LR_MAXPARTS = 64;           // workgroups of a pass
// slots of a pass's scalar partials (GP[wg * 8 + k])
enum { GP_A11 = 0, GP_A12, GP_A22, GP_R0G, GP_R1G };

__device__ __forceinline__ double wave4_sum(double v, double (*xs)[LR_CHUNK]) {      // sum over the 4 wavefronts, per lane
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    xs[wave][lane] = v;
    __syncthreads();
    return (xs[0][lane] + xs[1][lane]) + (xs[2][lane] + xs[3][lane]);
}
// sum over all 256 threads of a per-lane value that is ALREADY the same in the four wavefronts' lanes (one wave sums)
__device__ __forceinline__ double lanes_sum(double v) { return wave_sum64(v); }

// rows j = wave, wave + 4, ... < nrows of W against this lane's coordinate: lincomb with two coefficient vectors
__device__ __forceinline__ void rows_lincomb(const double* __restrict__ W, int ldw, int nrows, int cl, const double* c0,
                                             const double* c1, double* p0, double* p1) {
    const int wave = threadIdx.x >> 6;
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
    int j = wave;
    for (; j + 4 < nrows; j += 8) {
        const double w0 = W[(size_t)j * ldw + cl], w1 = W[(size_t)(j + 4) * ldw + cl];
        a0 += c0[j] * w0;
        b0 += c0[j + 4] * w1;
        if (c1) { a1 += c1[j] * w0; b1 += c1[j + 4] * w1; }
    }
    if (j < nrows) {
        const double w0 = W[(size_t)j * ldw + cl];
        a0 += c0[j] * w0;
        if (c1) a1 += c1[j] * w0;
    }
    *p0 = a0 + b0;
    if (p1) *p1 = a1 + b1;
}
// out_h[j] = -sum_lanes W_j[c] x_h   for the rows of this wavefront (x1 unused when out1 is null)
__device__ __forceinline__ void rows_dots(const double* __restrict__ W, int ldw, int nrows, int cl, bool valid, double x0,
                                          double x1, double* out0, double* out1) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = wave; j < nrows; j += 8) {
        const bool two = j + 4 < nrows;
        const double wa = valid ? W[(size_t)j * ldw + cl] : 0.0;
        const double wb = (valid && two) ? W[(size_t)(j + 4) * ldw + cl] : 0.0;
        const double a0 = wave_sum64(wa * x0), b0 = wave_sum64(wb * x0);
        double a1 = 0.0, b1 = 0.0;
        if (out1) { a1 = wave_sum64(wa * x1); b1 = wave_sum64(wb * x1); }
        if (lane == 0) {
            out0[j] = -a0;
            if (out1) out1[j] = -a1;
            if (two) { out0[j + 4] = -b0; if (out1) out1[j + 4] = -b1; }
        }
    }
}

struct SweepArgs {
    const double* W; int ldw, r, n;
    const double* Cin; int parts, ldc;          // parts == 0: final coefficients Cin[h * ldc + j]; else Cin[(p * 2 + h) * ldc + j]
    double* R; int ldr_;
    double* Cout;                               // partial dots of the NEW rows: Cout[(wg * 2 + h) * ldc + j] (null: not wanted)
    double* GP;                                 // scalar partials of the new rows (null: not wanted)
    const double* g;                            // with GP: R_h . g as well (null: none)
    double* Csum;                               // the coefficients applied here, summed (workgroup 0 writes them; null: not wanted)
};

__device__ __forceinline__ void lr_sweep_vb(const VB vb, SweepArgs a) {
    __shared__ double c0[LR_DEV_MAX], c1[LR_DEV_MAX], xs[4][LR_CHUNK];
    const int tid = threadIdx.x, r = a.r, lane = tid & 63;
    for (int j = tid; j < r; j += 256) {
        if (a.parts == 0) {
            c0[j] = a.Cin[j];
            c1[j] = a.Cin[a.ldc + j];
        } else {
            double s0 = 0.0, s1 = 0.0, t0 = 0.0, t1 = 0.0;
            int p = 0;
            for (; p + 1 < a.parts; p += 2) {
                s0 += a.Cin[(size_t)(2 * p) * a.ldc + j];
                s1 += a.Cin[(size_t)(2 * p + 1) * a.ldc + j];
                t0 += a.Cin[(size_t)(2 * p + 2) * a.ldc + j];
                t1 += a.Cin[(size_t)(2 * p + 3) * a.ldc + j];
            }
            if (p < a.parts) { s0 += a.Cin[(size_t)(2 * p) * a.ldc + j]; s1 += a.Cin[(size_t)(2 * p + 1) * a.ldc + j]; }
            c0[j] = s0 + t0;
            c1[j] = s1 + t1;
        }
        if (a.Csum && vb.x == 0) { a.Csum[j] = c0[j]; a.Csum[a.ldc + j] = c1[j]; }
    }
    const int c = vb.x * LR_CHUNK + lane;
    const bool valid = c < a.n;
    const int cl = valid ? c : a.n - 1;
    const double x0 = valid ? a.R[c] : 0.0, x1 = valid ? a.R[a.ldr_ + c] : 0.0;      // (in flight across the barrier)
    __syncthreads();
    double p0, p1;
    rows_lincomb(a.W, a.ldw, r, cl, c0, c1, &p0, &p1);
    const double sum0 = wave4_sum(p0, xs), sum1 = wave4_sum(p1, xs);
    const double r0 = valid ? x0 + sum0 : 0.0;
    const double r1 = valid ? x1 + sum1 : 0.0;
    if (valid && tid < 64) { a.R[c] = r0; a.R[a.ldr_ + c] = r1; }
    if (a.Cout)
        rows_dots(a.W, a.ldw, r, cl, valid, r0, r1, a.Cout + (size_t)(2 * vb.x) * a.ldc,
                  a.Cout + (size_t)(2 * vb.x + 1) * a.ldc);
    if (a.GP && tid < 64) {
        const double gv = (a.g && valid) ? a.g[c] : 0.0;
        const double s11 = lanes_sum(r0 * r0), s12 = lanes_sum(r0 * r1), s22 = lanes_sum(r1 * r1);
        const double s0g = lanes_sum(r0 * gv), s1g = lanes_sum(r1 * gv);
        if (tid == 0) {
            double* o = a.GP + 8 * vb.x;
            o[GP_A11] = s11; o[GP_A12] = s12; o[GP_A22] = s22; o[GP_R0G] = s0g; o[GP_R1G] = s1g;
        }
    }
}
__global__ __launch_bounds__(256) void lr_sweep_kernel(SweepArgs a) { lr_sweep_vb(vb_hw(), a); }

struct ERowsArgs {
    const double* W; int ldw, r, n;
    const double* R; int ldr_;
    const double* GP; int parts;                // scalar partials of the residual rows (from the second sweep)
    double* G;                                  // Gram block: entries 0..2 given; G_A11.., G_R0G, G_R1G written by workgroup 0
    double* Erow;                               // W + r * ldw: e1, then the second new row (unnormalised, before its clean-up)
    double* C3part; int ldc;                    // partial clean-up dots of the second row against [W; e1]: r + 1 entries each
    const double* SY = nullptr; int syparts = 0;    // Gram of the input rows still in partials (3 per part: x0.x0, x0.x1, x1.x1):
                                                    // summed here, written to G[0..2] by workgroup 0 (pipelined force call, view jobs)
};

// e1 = R0 / |R0| and the second row R1 - (a12 / a11) R0 (what lr_e1_kernel writes), plus the partial dots of that
// second row against [W; e1] — the clean-up sweep of lr_pre_kernel's comment, whose coefficients the next launch sums
__device__ __forceinline__ void lr_erows_vb(const VB vb, ERowsArgs a) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int c = vb.x * LR_CHUNK + lane;
    const bool valid = c < a.n;
    const int cl = valid ? c : a.n - 1;
    const double r0 = valid ? a.R[c] : 0.0, r1 = valid ? a.R[a.ldr_ + c] : 0.0;
    // the scalar partials of the second sweep: lane p of every wavefront fetches part p (parts <= 64), one wavefront-wide
    // sum each — a fixed tree, the same in every workgroup
    __shared__ double gs[8];
    {
        const double* o = a.GP + 8 * (lane < a.parts ? lane : 0);
        const bool on = lane < a.parts;
        const double t11 = wave_sum64(on ? o[GP_A11] : 0.0), t12 = wave_sum64(on ? o[GP_A12] : 0.0);
        const double t22 = wave_sum64(on ? o[GP_A22] : 0.0), t0g = wave_sum64(on ? o[GP_R0G] : 0.0);
        const double t1g = wave_sum64(on ? o[GP_R1G] : 0.0);
        if (tid == 0) { gs[0] = t11; gs[1] = t12; gs[2] = t22; gs[3] = t0g; gs[4] = t1g; }
    }
    __syncthreads();
    const double a11 = gs[0], a12 = gs[1], a22 = gs[2], r0g = gs[3], r1g = gs[4];
    double ss = 0.0;
    if (a.SY) for (int p = 0; p < a.syparts; ++p) ss += a.SY[3 * p];
    else ss = a.G[G_SS];
    const bool keep1 = ss > 0.0 && a11 > 1e-26 * ss;
    const double inv = keep1 ? 1.0 / sqrt(a11) : 0.0, f = keep1 ? a12 / a11 : 0.0;
    const double e1 = r0 * inv, row2 = r1 - f * r0;
    if (valid && tid < 64) { a.Erow[c] = e1; a.Erow[a.ldw + c] = row2; }
    double* out = a.C3part + (size_t)vb.x * a.ldc;
    rows_dots(a.W, a.ldw, a.r, cl, valid, row2, 0.0, out, nullptr);
    if (tid < 64) {
        const double de = lanes_sum(e1 * row2);      // against e1 itself: its chunk is in registers, the row is being written here
        if (tid == 0) out[a.r] = -de;
    }
    if (vb.x == 0 && tid == 0) {
        a.G[G_A11] = a11; a.G[G_A12] = a12; a.G[G_A12 + 1] = a12; a.G[G_A22] = a22;
        a.G[G_R0G] = r0g; a.G[G_R1G] = r1g;
        if (a.SY) {
            double sy = 0.0, yy = 0.0;
            for (int p = 0; p < a.syparts; ++p) { sy += a.SY[3 * p + 1]; yy += a.SY[3 * p + 2]; }
            a.G[G_SS] = ss;
            a.G[G_SY] = sy;
            a.G[G_YY] = yy;
        }
    }
}
__global__ __launch_bounds__(256) void lr_erows_kernel(ERowsArgs a) { lr_erows_vb(vb_hw(), a); }

struct CleanArgs {
    const double* W; int ldw, r, n;             // rows 0..r (row r = e1)
    const double* C3part; int parts, ldc;
    double* row2;                               // W + (r + 1) * ldw
    double* C3;                                 // summed coefficients (r + 1), written by workgroup 0
    double* NP;                                 // partial |row2|^2 after the clean-up: NP[2 * wg]; NP[2 * wg + 1]: row2 . g
    const double* g;                            // (null: none) — measured on the vector itself: R1.g - f R0.g cancels as badly as the norm
};

__device__ __forceinline__ void lr_clean_vb(const VB vb, CleanArgs a) {
    __shared__ double c3[LR_DEV_MAX + 8], xs[4][LR_CHUNK];
    const int tid = threadIdx.x, nr1 = a.r + 1, lane = tid & 63;
    for (int j = tid; j < nr1; j += 256) {
        double s = 0.0, t = 0.0;
        int p = 0;
        for (; p + 1 < a.parts; p += 2) { s += a.C3part[(size_t)p * a.ldc + j]; t += a.C3part[(size_t)(p + 1) * a.ldc + j]; }
        if (p < a.parts) s += a.C3part[(size_t)p * a.ldc + j];
        c3[j] = s + t;
        if (vb.x == 0) a.C3[j] = s + t;
    }
    const int c = vb.x * LR_CHUNK + lane;
    const bool valid = c < a.n;
    const int cl = valid ? c : a.n - 1;
    const double x = valid ? a.row2[c] : 0.0;
    __syncthreads();
    double p0;
    rows_lincomb(a.W, a.ldw, nr1, cl, c3, nullptr, &p0, nullptr);
    const double sum = wave4_sum(p0, xs);
    const double v = valid ? x + sum : 0.0;
    if (tid < 64) {
        if (valid) a.row2[c] = v;
        const double gv = (a.g && valid) ? a.g[c] : 0.0;
        const double s2 = lanes_sum(v * v), sg = lanes_sum(v * gv);
        if (tid == 0) { a.NP[2 * vb.x] = s2; a.NP[2 * vb.x + 1] = sg; }
    }
}
__global__ __launch_bounds__(256) void lr_clean_kernel(CleanArgs a) { lr_clean_vb(vb_hw(), a); }

struct GperpArgs {
    const double* Wnew; int ldw, nr, n;         // the nr new eigenvector rows; rows nr (g_perp) and nr + 1 (zero) are written here
    const double* Q; int ldq;                   // final coordinates (columns = new eigenvectors)
    const double* eg;                           // E g in coordinates (nr entries, lr_pre_kernel)
    const double* g;
    double* ghat;                               // -(Wnew g) = -(Q^T E g), written by workgroup 0
    double* NPg;                                // partial |g_perp|^2
};

// components of the gradient along the new eigenvectors WITHOUT a pass over them (they are Q^T applied to the components
// along the old rows, which the first launch of the job measured), then the part of g outside their span and its norm
__device__ __forceinline__ void lr_gperp_vb(const VB vb, GperpArgs a) {
    __shared__ double se[LR_SMALL + 8], gh[LR_SMALL + 8], xs[4][LR_CHUNK];
    const int tid = threadIdx.x, nr = a.nr, lane = tid & 63;
    const int c = vb.x * LR_CHUNK + lane;
    const bool valid = c < a.n;
    const int cl = valid ? c : a.n - 1;
    const double gv = valid ? a.g[c] : 0.0;
    for (int k = tid; k < nr; k += 256) se[k] = a.eg[k];
    __syncthreads();
    for (int i = tid; i < nr; i += 256) {
        double a0 = 0.0, a1 = 0.0;
        int k = 0;
        for (; k + 1 < nr; k += 2) {
            a0 += a.Q[(size_t)k * a.ldq + i] * se[k];
            a1 += a.Q[(size_t)(k + 1) * a.ldq + i] * se[k + 1];
        }
        if (k < nr) a0 += a.Q[(size_t)k * a.ldq + i] * se[k];
        const double v = -(a0 + a1);
        gh[i] = v;
        if (vb.x == 0) a.ghat[i] = v;
    }
    __syncthreads();
    double p0;
    rows_lincomb(a.Wnew, a.ldw, nr, cl, gh, nullptr, &p0, nullptr);
    const double sum = wave4_sum(p0, xs);
    const double v = valid ? gv + sum : 0.0;
    // (rows nr and nr + 1 of the panel: g_perp, and the zero row the weightless copies of the cluster point at; the
    // padding columns of both are zeroed too — the launch covers the whole leading dimension)
    if (tid < 64) {
        if (c < a.ldw) {
            ((double*)a.Wnew)[(size_t)nr * a.ldw + c] = v;
            ((double*)a.Wnew)[(size_t)(nr + 1) * a.ldw + c] = 0.0;
        }
        const double s2 = lanes_sum(v * v);
        if (tid == 0) a.NPg[vb.x] = s2;
    }
}
__global__ __launch_bounds__(256) void lr_gperp_kernel(GperpArgs a) { lr_gperp_vb(vb_hw(), a); }

// The secant pair formed on the device (pipelined force call): X1 holds g_old on entry; y = g - g_old -> X1 and X4, g -> X2;
// per-workgroup partials of s.s, s.y and y.y -> SY[3 * wg], SY[3 * wg + 1], SY[3 * wg + 2] (lr_erows_kernel sums them into the
// Gram block)
__device__ __forceinline__ void lr_secant_vb(const VB vb, double* __restrict__ X, int ld, int n, const double* __restrict__ g,
                                                        double* __restrict__ SY) {
    __shared__ double red[4];
    const int i = vb.x * 256 + threadIdx.x;
    const bool valid = i < n;
    const double gi = valid ? g[i] : 0.0;
    const double y = valid ? gi - X[ld + i] : 0.0;
    const double s = valid ? X[i] : 0.0;
    if (valid) { X[ld + i] = y; X[2 * (size_t)ld + i] = gi; X[4 * (size_t)ld + i] = y; }
    const double ss = blk_sum(s * s, red), sy = blk_sum(s * y, red), yy = blk_sum(y * y, red);
    if (threadIdx.x == 0) { SY[3 * vb.x] = ss; SY[3 * vb.x + 1] = sy; SY[3 * vb.x + 2] = yy; }
}
__global__ __launch_bounds__(256) void lr_secant_kernel(double* __restrict__ X, int ld, int n, const double* __restrict__ g,
                                                        double* __restrict__ SY) { lr_secant_vb(vb_hw(), X, ld, n, g, SY); }

// The update vectors of a view job formed where they are needed (one launch instead of a lincomb, two gathers, a zero
// fill, a copy and two Gram launches): u = sum_i UZ[2i] E_i, z = sum_i UZ[2i+1] E_i restricted to the view's coordinates
// idx, written as rows 0, 1 (inputs) and 3, 4 (residual rows) of the packed block Xs; row 2 = g[idx] when the gradient
// is on the device (else the caller uploads it); padding columns zero; partials of u.u, u.z, z.z -> SY[3 wg ..].
struct ViewRowsArgs {
    const double* E; int lde, nr;
    const double* UZ;
    const int* idx; int m, lds;
    const double* gsrc;
    double* Xs;
    double* SY;
};

__device__ __forceinline__ void lr_view_rows_vb(const VB vb, ViewRowsArgs a) {
    __shared__ double cu[LR_DEV_MAX + 8], cz[LR_DEV_MAX + 8], xs[4][LR_CHUNK];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < a.nr; i += 256) { cu[i] = a.UZ[2 * i]; cz[i] = a.UZ[2 * i + 1]; }
    const int c = vb.x * LR_CHUNK + lane;
    const bool valid = c < a.m;
    const int col = a.idx[valid ? c : a.m - 1];
    __syncthreads();
    double pu, pz;
    rows_lincomb(a.E, a.lde, a.nr, col, cu, cz, &pu, &pz);
    const double su = wave4_sum(pu, xs), sz = wave4_sum(pz, xs);
    const double u = valid ? su : 0.0, z = valid ? sz : 0.0;
    if (tid < 64) {
        if (c < a.lds) {
            a.Xs[c] = u; a.Xs[(size_t)a.lds + c] = z;
            a.Xs[3 * (size_t)a.lds + c] = u; a.Xs[4 * (size_t)a.lds + c] = z;
            if (a.gsrc) a.Xs[2 * (size_t)a.lds + c] = valid ? a.gsrc[col] : 0.0;
        }
        const double uu = lanes_sum(u * u), uz = lanes_sum(u * z), zz = lanes_sum(z * z);
        if (tid == 0) { a.SY[3 * vb.x] = uu; a.SY[3 * vb.x + 1] = uz; a.SY[3 * vb.x + 2] = zz; }
    }
}
__global__ __launch_bounds__(256) void lr_view_rows_kernel(ViewRowsArgs a) { lr_view_rows_vb(vb_hw(), a); }

__device__ __forceinline__ void lr_identity_vb(const VB vb, double* __restrict__ Q, int nr, int ldq) {
    const int i = vb.x * 256 + threadIdx.x;
    if (i < nr * ldq) Q[i] = ((i / ldq) == (i % ldq)) ? 1.0 : 0.0;
}
__global__ __launch_bounds__(256) void lr_identity_kernel(double* __restrict__ Q, int nr, int ldq) { lr_identity_vb(vb_hw(), Q, nr, ldq); }

// rows i < r of out scaled copies of W: out_i = (mu_i - lam0) W_i   (dense mirror: B = lam0 I + W^T out)
__device__ __forceinline__ void lr_scale_rows_vb(const VB vb, const double* __restrict__ W, int ldw, int r, int n,
                                                            const double* __restrict__ mu, double lam0,
                                                            double* __restrict__ out, int ldo) {
    const int i = vb.x * 256 + threadIdx.x, row = vb.y;
    if (i < n && row < r) out[(size_t)row * ldo + i] = (mu[row] - lam0) * W[(size_t)row * ldw + i];
}
__global__ __launch_bounds__(256) void lr_scale_rows_kernel(const double* __restrict__ W, int ldw, int r, int n,
                                                            const double* __restrict__ mu, double lam0,
                                                            double* __restrict__ out, int ldo) { lr_scale_rows_vb(vb_hw(), W, ldw, r, n, mu, lam0, out, ldo); }

}  // namespace

// Workspace of one job, carved out of one scratch slot.
struct LrWork {
    int nr, ldr, ldq;
    double *C, *C2, *C3, *G, *sc, *ec, *UZ, *P, *D0, *D1, *z, *Dp, *zz, *Dd, *wd, *tau, *zh, *lam, *cs, *pl, *mu, *ghat, *Qa, *Qb;
    double *npg, *eg, *np, *gp, *cpart, *c3part;       // fused chain: partials of the passes (npg directly behind ghat: read back with it)
    int *perm, *nd, *df, *i1, *i2, *org, *cnt;
};

static int lr_work(sella_ctx* c, int slot, int nr, LrWork& w) {
    w.nr = nr;
    w.ldr = round_up(nr + 2, 8);
    w.ldq = w.ldr;
    const size_t ldr = w.ldr;
    const size_t nchain = LR_MAXPARTS + ldr + 2 * LR_MAXPARTS + 8 * LR_MAXPARTS + 2 * LR_MAXPARTS * ldr + LR_MAXPARTS * ldr;
    const size_t ndbl = 3 * ldr + 3 * ldr + 16 + SC_N + 8 + 2 * ldr + 2 * ldr + 14 * ldr + 8 + 2 * ldr * ldr + nchain;
    const size_t nint = 7 * ldr + 8;
    double* base;
    SCHK(scratch_get(c, slot, (ndbl + nint / 2 + 8) * sizeof(double), &base));
    double* p = base;
    auto take = [&](size_t k) { double* q = p; p += k; return q; };
    w.C = take(3 * ldr); w.C2 = take(2 * ldr); w.C3 = take(ldr); w.G = take(16); w.ec = take(8);
    w.sc = take(SC_N); w.D0 = take(ldr); w.ghat = take(ldr);           // read back as ONE block: sc | D0 | ghat | npg
    w.npg = take(LR_MAXPARTS);
    w.UZ = take(2 * ldr); w.P = take(2 * ldr);
    w.D1 = take(ldr); w.z = take(ldr); w.Dp = take(ldr); w.zz = take(ldr); w.Dd = take(ldr);
    w.wd = take(ldr); w.tau = take(ldr); w.zh = take(ldr); w.lam = take(ldr); w.cs = take(2 * ldr); w.mu = take(ldr);
    w.pl = take(8);
    w.Qa = take(ldr * ldr); w.Qb = take(ldr * ldr);
    w.eg = take(ldr); w.np = take(2 * LR_MAXPARTS); w.gp = take(8 * LR_MAXPARTS);
    w.cpart = take(2 * LR_MAXPARTS * ldr); w.c3part = take(LR_MAXPARTS * ldr);
    int* ip = reinterpret_cast<int*>(p);
    w.perm = ip; w.nd = ip + ldr; w.df = ip + 2 * ldr; w.i1 = ip + 3 * ldr; w.i2 = ip + 4 * ldr; w.org = ip + 5 * ldr;
    w.cnt = ip + 6 * ldr;
    return SELLA_OK;
}

// One job: the structured decomposition (Wt rows 0..r-1, mu, lam0) receives the update defined by the two rows of Xd
// (mode 0: s and y, TS-BFGS; mode 1: u and z themselves).  Xd row 2 (optional, want_modes): the gradient whose
// components along the NEW eigenvectors (and whose part outside their span) the next step family needs.
// Everything is queued on the stream; results are valid after stream_wait:
//   Wnew: panel of nr + 2 rows: the r + 2 new eigenvector rows (ascending eigenvalues), then the unnormalised g_perp row, then
//   a zero row;  hD (nr) new eigenvalues, hghat (nr) = -(Wnew g), hsc (SC_N) scalars, hcnt (4 ints).
struct LrJob {
    Mat* Wt;
    int r, n, mode;
    const double* mu;
    double lam0;
    const double* Xd;          // device rows: (s | u), (y | z), g
    int ldx;
    const double* gram;        // host: ss, sy, yy of the two rows (mode 0) or nullptr (computed on the device)
    bool want_modes;
    int slot_ws, slot_panel;
    // optional: what the caller has put on the device already in ONE transfer — the residual rows as copies of the
    // inputs (2 rows, stride ldx), the eigenvalues and a 16-double Gram block (entries 0..2 set) — instead of four copies here
    double* Rpre = nullptr;
    double* mu_dev = nullptr;
    double* G_dev = nullptr;
    hipEvent_t fork_ev = nullptr;      // recorded behind the launch after which a dependent job on another stream may start
    const double* SY = nullptr;        // pipelined force call: partials of s.y, y.y still to be summed into the Gram block
    int syparts = 0;
    // results
    LrWork w;
    double* Wnew;
    int ldw;
    std::vector<double> hout;          // sc | D | ghat (| partials of |g_perp|^2) as read back in one transfer
    const double *hD, *hghat, *hsc;
    int gparts = 0;                    // > 0: |g_perp|^2 = sum of that many partials behind ghat (fused chain)
    double gperp2() const {
        if (gparts == 0) return hsc[SC_GPERP2];
        double s = 0.0;
        for (int p = 0; p < gparts; ++p) s += hghat[w.ldr + p];
        return s;
    }
};

static int lr_job_queue(sella_ctx* c, LrJob& j) {
    const int r = j.r, n = j.n, nr = r + 2;
    Mat* Wm = j.Wt;
    const int ld = Wm->ld;
    if (Wm->rows < nr + 2) { set_error("structured update: eigenvector panel too small"); return SELLA_E_INVALID; }
    SCHK(lr_work(c, j.slot_ws, nr, j.w));
    LrWork& w = j.w;
    double* panel;
    SCHK(scratch_get(c, j.slot_panel, ((size_t)(nr + 2) + 2) * ld * sizeof(double), &panel));
    double* R = panel;                               // residual rows (2)
    j.Wnew = panel + 2 * (size_t)ld;
    j.ldw = ld;
    double* W = Wm->d;
    const bool packed = j.Rpre && j.mu_dev && j.G_dev && j.ldx == ld;
    if (packed) {
        R = j.Rpre;
        w.mu = j.mu_dev;
        w.G = j.G_dev;
    }
    // small uploads: mu and the Gram of the input rows
    if (packed) {
    } else if (r > 0) SCHK(h2d_async(c, w.mu, j.mu, (size_t)r * sizeof(double)));
    if (packed) {
    } else if (j.gram) {
        SCHK(h2d_async(c, w.G, j.gram, 3 * sizeof(double)));
    } else {
        // G[h * 2 + i] = X_i . X_h -> ss = G[0], sy = G[1] ... laid out to match (ss, sy, yy) needs a shuffle: use 3 dots
        SCHK(launch_gemv_rows(c, j.Xd, 1, n, j.ldx, j.Xd, j.ldx, 2, w.G, 1, GemvEpi()));                  // ss, sy
        SCHK(launch_gemv_rows(c, j.Xd + j.ldx, 1, n, j.ldx, j.Xd + j.ldx, j.ldx, 1, w.G + 2, 1, GemvEpi()));   // yy
    }
    // residual rows start as copies of the inputs
    if (!packed) SCHK(launch_axpby2d(c, 2, n, 1.0, j.Xd, j.ldx, 0.0, nullptr, 0, R, ld));
    const int parts = (n + LR_CHUNK - 1) / LR_CHUNK;
    const bool chain = c->opt.lr_chain && parts <= LR_MAXPARTS;
    const bool small = chain && nr <= LR_SMALL;            // merged coordinate kernels
    double* Erow = W + (size_t)r * ld;
    if (chain) {
        // ---- fused chain: dots, two sweeps, the new rows, their clean-up (five launches; see the kernels) ----------
        const double* g = j.want_modes ? j.Xd + 2 * (size_t)j.ldx : nullptr;
        if (r > 0) {
            GemvEpi neg;
            neg.alpha = -1.0;
            const double* xs[3] = {R, R + ld, g};
            SCHK(launch_gemv_rows_xp(c, W, r, n, ld, xs, g ? 3 : 2, w.C, w.ldr, neg));     // C (and the negated W g)
        }
        SweepArgs sw;
        sw.W = W; sw.ldw = ld; sw.r = r; sw.n = n; sw.R = R; sw.ldr_ = ld; sw.ldc = w.ldr;
        if (r > 0) {
            sw.Cin = w.C; sw.parts = 0; sw.Cout = w.cpart; sw.GP = nullptr; sw.g = nullptr; sw.Csum = nullptr;
            SELLA_LAUNCHB(c, lr_sweep_kernel, lr_sweep_vb, 256, dim3(parts), dim3(256), 0, sw);
        }
        sw.Cin = w.cpart; sw.parts = r > 0 ? parts : 0; sw.Cout = nullptr; sw.GP = w.gp; sw.g = g; sw.Csum = r > 0 ? w.C2 : nullptr;
        if (r == 0) sw.Cin = w.C;                          // (never read: no rows)
        SELLA_LAUNCHB(c, lr_sweep_kernel, lr_sweep_vb, 256, dim3(parts), dim3(256), 0, sw);
        ERowsArgs er;
        er.W = W; er.ldw = ld; er.r = r; er.n = n; er.R = R; er.ldr_ = ld; er.GP = w.gp; er.parts = parts; er.G = w.G;
        er.Erow = Erow; er.C3part = w.c3part; er.ldc = w.ldr; er.SY = j.SY; er.syparts = j.syparts;
        SELLA_LAUNCHB(c, lr_erows_kernel, lr_erows_vb, 256, dim3(parts), dim3(256), 0, er);
        CleanArgs cl;
        cl.W = W; cl.ldw = ld; cl.r = r; cl.n = n; cl.C3part = w.c3part; cl.parts = parts; cl.ldc = w.ldr;
        cl.row2 = Erow + ld; cl.C3 = w.C3; cl.NP = w.np; cl.g = g;
        SELLA_LAUNCHB(c, lr_clean_kernel, lr_clean_vb, 256, dim3(parts), dim3(256), 0, cl);
        HIPCHK(hipGetLastError());
    } else {
    if (r > 0) {
        // two classical Gram-Schmidt sweeps of both rows against W
        GemvEpi neg;                                  // C, C2 hold the NEGATED coefficients: lincomb adds them
        neg.alpha = -1.0;
        SCHK(launch_gemv_rows(c, W, r, n, ld, R, ld, 2, w.C, w.ldr, neg));
        SELLA_LAUNCHB(c, lr_proj2_kernel, lr_proj2_vb, 256, dim3((n + 255) / 256), dim3(256), 0, W, ld, r, n, w.C, w.ldr, R, ld);
        SCHK(launch_gemv_rows(c, W, r, n, ld, R, ld, 2, w.C2, w.ldr, neg));
        SELLA_LAUNCHB(c, lr_proj2_kernel, lr_proj2_vb, 256, dim3((n + 255) / 256), dim3(256), 0, W, ld, r, n, w.C2, w.ldr, R, ld);
        HIPCHK(hipGetLastError());
    }
    // Gram of the residual rows: G[8 + h * 2 + i] = R_i . R_h -> a11 = G[8], a12 = G[9] (= G[10]), a22 = G[11]
    SCHK(launch_gemv_rows(c, R, 2, n, ld, R, ld, 2, w.G + G_A11, 2, GemvEpi()));
    // the two new rows of E, in place behind W: e1, and the second one after a clean-up sweep against [W; e1]
    SELLA_LAUNCHB(c, lr_e1_kernel, lr_e1_vb, 256, dim3((n + 255) / 256), dim3(256), 0, R, ld, n, w.G, Erow, ld);
    HIPCHK(hipGetLastError());
    {
        GemvEpi neg;
        neg.alpha = -1.0;
        SCHK(launch_gemv_rows(c, W, r + 1, n, ld, Erow + ld, ld, 1, w.C3, w.ldr, neg));
        SCHK(launch_lincomb(c, n, 1, W, ld, r + 1, w.C3, 1, nullptr, 0, 0, nullptr, 0, 1.0, Erow + ld, ld));
    }
    }
    PreArgs pa;
    pa.r = r; pa.nr = nr; pa.ldr = w.ldr; pa.mode = j.mode;
    pa.C = w.C; pa.C2 = w.C2; pa.C3 = w.C3; pa.G = w.G; pa.mu = w.mu; pa.lam0 = j.lam0;
    pa.row2 = Erow + ld; pa.n = n;
    pa.sc = w.sc; pa.ec = w.ec; pa.UZ = w.UZ; pa.P = w.P; pa.D = w.D0;
    if (chain) {
        pa.NP = w.np; pa.parts = parts;
        if (j.want_modes && small) { pa.CG = w.C + 2 * (size_t)w.ldr; pa.eg = w.eg; }
    }
    // two rank-one terms in coordinates
    double *Qin = w.Qa, *Qout = w.Qb, *Din = w.D0, *Dout = w.D1;
    if (!chain) {
        SELLA_LAUNCHB(c, lr_pre_kernel, lr_pre_vb, 256, dim3(1), dim3(256), 0, pa);
        SELLA_LAUNCHB(c, lr_identity_kernel, lr_identity_vb, 256, dim3((nr * w.ldq + 255) / 256), dim3(256), 0, w.Qa, nr, w.ldq);
        HIPCHK(hipGetLastError());
    }
    for (int t = 0; t < 2; ++t) {
        PlanArgs pl;
        pl.nr = nr; pl.ldr = w.ldr; pl.ldq = w.ldq; pl.first = (t == 0); pl.fill = chain ? 1 : 0;
        pl.p = w.P + (size_t)t * w.ldr; pl.sigma = w.sc + SC_SIG1 + t; pl.Dcur = Din; pl.Q = Qin;
        pl.z = w.z; pl.Dp = w.Dp; pl.zz = w.zz; pl.Dd = w.Dd; pl.wd = w.wd; pl.cs = w.cs; pl.pl = w.pl;
        pl.perm = w.perm; pl.nd = w.nd; pl.df = w.df; pl.i1 = w.i1; pl.i2 = w.i2; pl.cnt = w.cnt;
        if (chain && t == 0) {
            SELLA_LAUNCHB(c, lr_pre_plan_kernel, lr_pre_plan_vb, 256, dim3(1), dim3(256), 0, pa, pl);
            // (from here on the job only works in coordinates and on its own panel: the two new rows of E and the update
            // vectors' coordinates are final)
            if (j.fork_ev) HIPCHK(hipEventRecord(j.fork_ev, c->stream));
        } else SELLA_LAUNCHB(c, lr_plan_kernel, lr_plan_vb, 256, dim3(1), dim3(256), 0, pl);
        SELLA_LAUNCHB(c, lr_secular_kernel, lr_secular_vb, 256, dim3((nr + 3) / 4), dim3(256), 0, w.cnt, w.pl, w.Dd, w.wd, w.tau,
                           w.org, w.lam, w.sc + SC_FAIL);
        // (the Gu / Eisenstat weights computed by every apply workgroup for itself were measured: the apply kernel grows by
        // more than the launch saves — 11.7 / 14.8 us against 6.5 + 4.5 — so they keep their launch; ApplyArgs::wd stays as
        // the switch)
        SELLA_LAUNCHB(c, lr_zhat_kernel, lr_zhat_vb, 256, dim3((nr + 3) / 4), dim3(256), 0, w.cnt, w.Dd, w.wd, w.tau, w.org,
                               w.zh);
        ApplyArgs ap;
        ap.nr = nr; ap.ldq = w.ldq; ap.cnt = w.cnt; ap.nd = w.nd; ap.df = w.df; ap.org = w.org; ap.pl = w.pl;
        ap.Dd = w.Dd; ap.Dp = w.Dp; ap.zh = w.zh; ap.tau = w.tau; ap.lam = w.lam; ap.Qin = Qin; ap.Qout = Qout; ap.Dnext = Dout;
        ap.first = (chain && t == 0) ? 1 : 0;
        ap.wd = nullptr;
        SELLA_LAUNCHB(c, lr_apply_kernel, lr_apply_vb, 256, dim3(nr), dim3(256), 0, ap);
        HIPCHK(hipGetLastError());
        std::swap(Qin, Qout);
        std::swap(Din, Dout);
    }
    // (after two terms: Qin == Qa, Din == D0 again)
    // W+ = Q^T E on the matrix cores
    SCHK(launch_gemm(c, 1, 0, nr, n, nr, 1.0, Qin, w.ldq, W, ld, 0.0, j.Wnew, ld));
    j.hout.assign((size_t)SC_N + 2 * w.ldr + LR_MAXPARTS, 0.0);
    j.gparts = 0;
    if (j.want_modes && small) {
        // components along the new eigenvectors in coordinates, g_perp and its norm: one launch (it also writes the zero row)
        GperpArgs ga;
        ga.Wnew = j.Wnew; ga.ldw = ld; ga.nr = nr; ga.n = n; ga.Q = Qin; ga.ldq = w.ldq; ga.eg = w.eg;
        ga.g = j.Xd + 2 * (size_t)j.ldx; ga.ghat = w.ghat; ga.NPg = w.npg;
        j.gparts = (ld + LR_CHUNK - 1) / LR_CHUNK;
        if (j.gparts > LR_MAXPARTS) { set_error("structured update: too many chunks"); return SELLA_E_INVALID; }
        SELLA_LAUNCHB(c, lr_gperp_kernel, lr_gperp_vb, 256, dim3(j.gparts), dim3(256), 0, ga);
        HIPCHK(hipGetLastError());
    } else {
        HIPCHK(s_memset0(c, j.Wnew + (size_t)nr * ld, (size_t)2 * ld * sizeof(double)));
    if (j.want_modes) {
        const double* g = j.Xd + 2 * (size_t)j.ldx;
        GemvEpi e;
        e.alpha = -1.0;                               // -(W+ g): the coefficients lincomb adds to g
        SCHK(launch_gemv_rows(c, j.Wnew, nr, n, ld, g, j.ldx, 1, w.ghat, w.ldr, e));
        double* gp = j.Wnew + (size_t)nr * ld;
        SCHK(launch_axpby2d(c, 1, n, 1.0, g, j.ldx, 0.0, nullptr, 0, gp, ld));
        SCHK(launch_lincomb(c, n, 1, j.Wnew, ld, nr, w.ghat, 1, nullptr, 0, 0, nullptr, 0, 1.0, gp, ld));
        SCHK(launch_rows_sumsq(c, gp, ld, 1, n, w.sc + SC_GPERP2));
    }
    }
    if (Din != w.D0) { set_error("structured update: eigenvalue buffers out of step"); return SELLA_E_INVALID; }
    SCHK(d2h_async(c, j.hout.data(), w.sc, j.hout.size() * sizeof(double)));
    j.hsc = j.hout.data();
    j.hD = j.hout.data() + SC_N;
    j.hghat = j.hD + w.ldr;
    return SELLA_OK;
}

__device__ __forceinline__ void lr_gather_cols_vb(const VB vb, const double* __restrict__ P, int ldp, int rows,
                                                             const int* __restrict__ idx, int m,
                                                             double* __restrict__ out, int ldo) {
    const int i = vb.x * 256 + threadIdx.x;
    const int r = vb.y;
    if (i < m && r < rows) out[(size_t)r * ldo + i] = P[(size_t)r * ldp + idx[i]];
}
__global__ __launch_bounds__(256) void lr_gather_cols_kernel(const double* __restrict__ P, int ldp, int rows,
                                                             const int* __restrict__ idx, int m,
                                                             double* __restrict__ out, int ldo) { lr_gather_cols_vb(vb_hw(), P, ldp, rows, idx, m, out, ldo); }

// After the wait: the new explicit pairs of one job (rows whose eigenvalue is exactly lam0 belong to the cluster again:
// dropped, zero rows of a direction that was already in span(W) included) written back into the decomposition.
static int lr_job_commit(sella_ctx* c, LrJob& j, int* r_io, double* mu, std::vector<int>& kept) {
    const int nr = j.r + 2, n = j.n;
    kept.clear();
    for (int i = 0; i < nr; ++i)
        if (j.hD[i] != j.lam0) kept.push_back(i);
    const int rn = (int)kept.size();
    Mat* Wm = j.Wt;
    if (rn == nr) {
        SCHK(launch_axpby2d(c, nr, n, 1.0, j.Wnew, j.ldw, 0.0, nullptr, 0, Wm->d, Wm->ld));
    } else if (rn > 0) {
        int* didx = j.w.perm;                                          // (free again after the wait)
        SCHK(h2d_async(c, didx, kept.data(), (size_t)rn * sizeof(int)));
        SCHK(launch_gather_rows(c, j.Wnew, j.ldw, didx, rn, n, Wm->d, Wm->ld));
    }
    for (int i = 0; i < rn; ++i) mu[i] = j.hD[kept[i]];
    *r_io = rn;
    return SELLA_OK;
}

// The fast form of sella_opt_step (optstep.hip): TS-BFGS, one secant pair, structured decompositions small enough for
// the coordinate kernels.  *handled = false: nothing was changed and the caller takes the general route.
int lr_fused_step(sella_ctx* c, sella_opt_step_t* a, bool* handled, CalcPipe* pipe) {
    *handled = false;
    static const bool step_timing = getenv("SELLA_STEP_TIMING") != nullptr;   // host side of a step, phase by phase (stderr)
    auto stamp = [] { return std::chrono::steady_clock::now(); };
    const auto ts0 = stamp();
    const int n = a->n;
    const bool view = a->idx != nullptr && a->m > 0;
    if (!(a->flags & SELLA_OPT_LEARN) || a->update_method != SELLA_UPD_TS_BFGS || !c->opt.lr_dev) return SELLA_OK;
    if (*a->r + 2 > LR_DEV_MAX || (view && (!a->r_sub || *a->r_sub + 2 > LR_DEV_MAX))) return SELLA_OK;
    Mat* Wm = mat_get(c, a->Wt);
    Mat* Ws = view ? mat_get(c, a->Wt_sub) : nullptr;
    if (!Wm || Wm->cols != n || Wm->rows < *a->r + 4 || (view && (!Ws || Ws->cols != a->m || Ws->rows < *a->r_sub + 4)))
        return SELLA_OK;
    const bool propose = (a->flags & SELLA_OPT_PROPOSE) != 0;
    const bool on_boundary = a->smag == a->delta;          // the step just taken ended on the trust boundary (val_out = delta)
    const int ld = round_up(n, 8);
    // pipelined force call: only where the secant pair can be formed on the device (fused chain, packed staging)
    const bool piped = pipe && pipe->calc && c->opt.lr_chain && c->opt.lr_pipe && Wm->ld == ld &&
                       (n + 63) / 64 <= LR_MAXPARTS && (n + 255) / 256 <= LR_MAXPARTS;
    // host side of the secant pair
    std::vector<double>& y = c->hbuf_a;
    y.resize((size_t)n);
    double gram[3] = {0.0, 0.0, 0.0}, gd = 0.0, gg = 0.0;
    for (int i = 0; i < n; ++i) {
        gram[0] += a->dx[i] * a->dx[i];
        gd += a->g_old[i] * a->dx[i];
    }
    if (!piped) {
        if (pipe && pipe->calc) return SELLA_OK;                    // (the caller makes the force call and comes back)
        for (int i = 0; i < n; ++i) {
            y[i] = a->g_new[i] - a->g_old[i];
            gram[1] += a->dx[i] * y[i];
            gram[2] += y[i] * y[i];
            gg += a->g_new[i] * a->g_new[i];
        }
    }
    if (!(std::sqrt(gram[0]) >= 1e-8)) return SELLA_OK;             // B is left alone: the general route knows how
    double *gdev = nullptr, *auxdev = nullptr;
    int naux = 0;
    std::vector<double> auxh;
    if (piped) {
        // the force call first, nothing waited for: its kernels run while the host stages and queues the update
        SCHK(calc_queue(pipe->calc, pipe->x, &gdev, &auxdev, &naux));
        auxh.resize((size_t)naux);
    }
    // ONE transfer for everything the full-space job reads from the host: the rows s, y, g; s, y again as the residual
    // rows the two sweeps work on; the eigenvalues; the Gram entries of (s, y)
    const int ldmu = round_up(*a->r + 2, 8);
    const size_t nstage = (size_t)5 * ld + ldmu + 16;
    double* X;
    SCHK(scratch_get(c, SCR_UPD0, nstage * sizeof(double), &X));
    {
        // composed directly in the pinned ring (no intermediate buffer)
        void* slot = nullptr;
        std::vector<double>& hsv = c->hbuf_b;
        double* hs;
        const bool ring = h2d_begin(c, nstage * sizeof(double), &slot) == SELLA_OK;
        if (ring) hs = static_cast<double*>(slot);
        else { hsv.assign(nstage, 0.0); hs = hsv.data(); }
        std::copy(a->dx, a->dx + n, hs);
        if (piped) {
            std::copy(a->g_old, a->g_old + n, hs + ld);               // becomes y on the device (lr_secant_kernel)
        } else {
            std::copy(y.begin(), y.end(), hs + ld);
            std::copy(a->g_new, a->g_new + n, hs + 2 * (size_t)ld);
            std::copy(y.begin(), y.end(), hs + 4 * (size_t)ld);
        }
        std::copy(a->dx, a->dx + n, hs + 3 * (size_t)ld);
        std::copy(a->mu, a->mu + *a->r, hs + 5 * (size_t)ld);
        std::copy(gram, gram + 3, hs + 5 * (size_t)ld + ldmu);
        if (ring) SCHK(h2d_end(c, X, slot, nstage * sizeof(double)));
        else SCHK(h2d_async(c, X, hs, nstage * sizeof(double)));
    }
    LrJob F;
    F.Wt = Wm; F.r = *a->r; F.n = n; F.mode = 0; F.mu = a->mu; F.lam0 = a->lam0; F.Xd = X; F.ldx = ld; F.gram = gram;
    if (Wm->ld == ld) { F.Rpre = X + 3 * (size_t)ld; F.mu_dev = X + 5 * (size_t)ld; F.G_dev = F.mu_dev + ldmu; }
    double* SYp = nullptr;
    if (piped) {
        const int syparts = (n + 255) / 256;
        SCHK(scratch_get(c, SCR_UPD3, (size_t)6 * LR_MAXPARTS * sizeof(double), &SYp));
        SELLA_LAUNCHB(c, lr_secant_kernel, lr_secant_vb, 256, dim3(syparts), dim3(256), 0, X, ld, n, gdev, SYp);
        HIPCHK(hipGetLastError());
        F.SY = SYp;
        F.syparts = syparts;
    }
    F.want_modes = propose && !view; F.slot_ws = SCR_EIG0; F.slot_panel = SCR_EIG1;
    // the view job depends on the full-space job up to its lr_pre_plan_kernel only: from there the two run side by side
    bool overlap = false;
    if (view && c->opt.lr_overlap && c->opt.lr_chain && (n + LR_CHUNK - 1) / LR_CHUNK <= LR_MAXPARTS) {
        if (!c->stream2) {
            if (hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
                set_error("structured update: second stream could not be created");
                return SELLA_E_HIP;
            }
        }
        overlap = true;
        F.fork_ev = c->ev_fork;
    }
    SCHK(lr_job_queue(c, F));
    LrJob S;
    std::vector<double> gsub;
    struct StreamSwap {                       // every launch helper queues on c->stream: the view job is queued with the second
        sella_ctx* c; hipStream_t saved; bool on;                      // stream installed there (restored on every path)
        ~StreamSwap() { if (on) c->stream = saved; }
    } swap{c, c->stream, false};
    if (view) {
        if (overlap) {
            HIPCHK(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
            c->stream = c->stream2;
            swap.on = true;
        }
        const int m = a->m, lds = round_up(m, 8), nr = F.r + 2;
        double *UZp, *Xs;
        const int ldmus = round_up(*a->r_sub + 2, 8);
        const int vparts = (lds + LR_CHUNK - 1) / LR_CHUNK;
        const bool vfused = c->opt.lr_chain && Ws->ld == lds && vparts <= LR_MAXPARTS && nr <= LR_DEV_MAX;
        SCHK(scratch_get(c, SCR_UPD1, (size_t)2 * ld * sizeof(double), &UZp));
        SCHK(scratch_get(c, SCR_UPD2, ((size_t)5 * lds + ldmus + 16 + (size_t)m / 2 + 8) * sizeof(double), &Xs));
        int* didx = reinterpret_cast<int*>(Xs + 5 * (size_t)lds + ldmus + 16);
        if (!vfused) SCHK(h2d_async(c, didx, a->idx, (size_t)m * sizeof(int)));          // (fused: rides with the eigenvalues below)
        gsub.resize((size_t)m);
        S.Wt = Ws; S.r = *a->r_sub; S.n = m; S.mode = 1; S.mu = a->mu_sub; S.lam0 = a->lam0; S.Xd = Xs; S.ldx = lds;
        S.gram = nullptr; S.want_modes = propose; S.slot_ws = SCR_EIG2; S.slot_panel = SCR_EIG3;
        if (vfused) {
            // u, z on the view's coordinates, their copies as residual rows, their Gram partials, the gradient: one launch
            // eigenvalues of the view + the index map behind them: one transfer (they are neighbours on the device)
            std::vector<double> small((size_t)ldmus + 16 + (size_t)(m + 1) / 2, 0.0);
            std::copy(a->mu_sub, a->mu_sub + *a->r_sub, small.begin());
            memcpy(small.data() + ldmus + 16, a->idx, (size_t)m * sizeof(int));
            SCHK(h2d_async(c, Xs + 5 * (size_t)lds, small.data(), ((size_t)ldmus + 16) * sizeof(double) + (size_t)m * sizeof(int)));
            ViewRowsArgs va;
            va.E = Wm->d; va.lde = Wm->ld; va.nr = nr; va.UZ = F.w.UZ; va.idx = didx; va.m = m; va.lds = lds;
            va.gsrc = piped ? X + 2 * (size_t)ld : nullptr; va.Xs = Xs;
            double* SYv = SYp ? SYp + 3 * LR_MAXPARTS : nullptr;
            if (!SYv) { SCHK(scratch_get(c, SCR_UPD3, (size_t)6 * LR_MAXPARTS * sizeof(double), &SYv)); SYv += 3 * LR_MAXPARTS; }
            va.SY = SYv;
            SELLA_LAUNCHB(c, lr_view_rows_kernel, lr_view_rows_vb, 256, dim3(vparts), dim3(256), 0, va);
            HIPCHK(hipGetLastError());
            if (!piped) {
                for (int q = 0; q < m; ++q) gsub[q] = a->g_new[a->idx[q]];
                // whole row, padding zeroed: the scratch may hold another search's layout beyond column m
                double* slot = nullptr;
                SCHK(h2d_begin(c, (size_t)lds * sizeof(double), reinterpret_cast<void**>(&slot)));
                memcpy(slot, gsub.data(), (size_t)m * sizeof(double));
                SCHK(h2d_end(c, Xs + 2 * (size_t)lds, slot, (size_t)lds * sizeof(double)));
            }
            S.Rpre = Xs + 3 * (size_t)lds; S.mu_dev = Xs + 5 * (size_t)lds; S.G_dev = S.mu_dev + ldmus;
            S.SY = SYv; S.syparts = vparts;
        } else {
        // u, z as vectors (rows of E weighted by their coordinates), restricted to the view's coordinates
        SCHK(launch_lincomb(c, n, 2, Wm->d, Wm->ld, nr, F.w.UZ, 2, nullptr, 0, 0, nullptr, 0, 0.0, UZp, ld));
        HIPCHK(s_memset0(c, Xs, (size_t)3 * lds * sizeof(double)));
        SELLA_LAUNCHB(c, lr_gather_cols_kernel, lr_gather_cols_vb, 256, dim3((m + 255) / 256, 2), dim3(256), 0, UZp, ld, 2, didx, m, Xs, lds);
        HIPCHK(hipGetLastError());
        if (piped) {                                                   // the gradient is still on its way: gathered on the device
            SELLA_LAUNCHB(c, lr_gather_cols_kernel, lr_gather_cols_vb, 256, dim3((m + 255) / 256, 1), dim3(256), 0, X + 2 * (size_t)ld, ld, 1,
                               didx, m, Xs + 2 * (size_t)lds, lds);
            HIPCHK(hipGetLastError());
        } else {
            for (int q = 0; q < m; ++q) gsub[q] = a->g_new[a->idx[q]];
            SCHK(h2d_async(c, Xs + 2 * (size_t)lds, gsub.data(), (size_t)m * sizeof(double)));
        }
        }
        SCHK(lr_job_queue(c, S));
        if (overlap) {
            HIPCHK(hipEventRecord(c->ev_join, c->stream));          // (c->stream is the second stream here)
            c->stream = swap.saved;
            swap.on = false;
            HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0));   // every later wait on the main stream covers the view job
        }
    }
    if (piped) {
        // gradient and energy terms come back with the results of the update, delivered by the one wait
        SCHK(d2h_async(c, pipe->g_out, X + 2 * (size_t)ld, (size_t)n * sizeof(double)));
        SCHK(d2h_async(c, auxh.data(), auxdev, (size_t)naux * sizeof(double)));
    }
    const auto ts1 = stamp();
    SCHK(stream_wait(c));
    const auto ts2 = stamp();
    if (piped) {
        // the force call is complete whatever becomes of the update
        pipe->f = calc_finish(pipe->calc, pipe->x, auxh.data());
        pipe->done = true;
        a->f_new = pipe->f;
        for (int i = 0; i < n; ++i) gg += pipe->g_out[i] * pipe->g_out[i];
        if (view) for (int q = 0; q < a->m; ++q) gsub[q] = pipe->g_out[a->idx[q]];
    }
    auto sound = [](const LrJob& j) {
        if (j.hsc[SC_FAIL] != 0.0) return false;
        for (int i = 0; i < j.r + 2; ++i) if (!(j.hD[i] == j.hD[i])) return false;
        return true;
    };
    if (!sound(F) || (view && !sound(S))) {                            // nothing committed: the general route takes the step
        if (getenv("SELLA_DEBUG_TIMING"))
            fprintf(stderr, "opt_step: coordinate update set aside (secular solver flag: full %g, view %g)\n", F.hsc[SC_FAIL],
                    view ? S.hsc[SC_FAIL] : 0.0);
        return SELLA_OK;
    }
    if (getenv("SELLA_DEBUG_LR")) {
        auto dump = [](const char* tag, const LrJob& j) {
            const int nr = j.r + 2;
            double sd = 0.0, sg = 0.0;
            for (int i = 0; i < nr; ++i) { sd += j.hD[i] * (i + 1); sg += j.hghat[i] * j.hghat[i]; }
            fprintf(stderr, "LR %s r=%d sumD=%.15e |ghat|^2=%.15e gp2=%.15e m1=%.15e m2=%.15e js=%.15e sBs=%.15e sig=%.15e %.15e keep=%g %g\n", tag, j.r, sd,
                    sg, j.gperp2(), j.hsc[SC_M1], j.hsc[SC_M2], j.hsc[SC_JS], j.hsc[SC_SBS], j.hsc[SC_SIG1], j.hsc[SC_SIG2],
                    j.hsc[SC_KEEP1], j.hsc[SC_KEEP2]);
        };
        dump("F", F);
        if (view) dump("S", S);
    }
    *handled = true;
    std::vector<int> keptF, keptS;
    SCHK(lr_job_commit(c, F, a->r, a->mu, keptF));
    if (view) SCHK(lr_job_commit(c, S, a->r_sub, a->mu_sub, keptS));
    a->updated = 1;
    a->nrank1 = 2;
    a->nrank1_sub = view ? 2 : 0;
    a->B_stale = 1;
    if (view) a->Bsub_stale = 1;
    // model prediction with s.B s from the coordinates (peswrapper.py:446-449), ratio, radius (optimize.py:413-434)
    const double predicted = gd + 0.5 * F.hsc[SC_SBS];
    a->df_pred = predicted;
    a->ratio_valid = 0;
    if (std::fabs(predicted) >= 1e-14) {
        a->ratio = (a->f_new - a->f_old) / predicted;
        a->ratio_valid = 1;
    }
    if (!a->ratio_valid) {
        a->rho = 1.0;
    } else {
        const double rho = a->ratio;
        if (!(1.0 / a->rho_dec <= rho && rho <= a->rho_dec)) a->delta = std::fmax(a->smag * a->sigma_dec, a->delta_min);
        else if (1.0 / a->rho_inc < rho && rho < a->rho_inc) a->delta = std::fmax(a->sigma_inc * a->smag, a->delta);
        a->rho = rho;
    }
    const auto ts3 = stamp();
    if (!propose) return SELLA_OK;
    // step family on the new modes: explicit pairs + the part of g outside their span (+ weightless copies), as
    // sella_stepper_create_lr builds it, from what came back with the update
    LrJob& J = view ? S : F;
    const std::vector<int>& kept = view ? keptS : keptF;
    const int nd = J.n, nrj = J.r + 2, rn = (int)kept.size();
    double g2 = gg;
    if (view) { g2 = 0.0; for (double v : gsub) g2 += v * v; }
    const double gp2 = J.gperp2();
    const int ncl = nd - rn;
    const bool have_perp = ncl > 0 && gp2 > 0.0 && gp2 > 1e-26 * g2;
    double* gprow = J.Wnew + (size_t)nrj * J.ldw;
    const bool on_panel = a->cons <= 1 && c->opt.lr_chain;       // (see below: the family then reads the panel's rows in place)
    if (have_perp && on_panel && J.gparts > 0) { /* the row stays unnormalised: its factor goes into the step's coefficient */ }
    else if (have_perp && J.gparts > 0) SCHK(launch_axpby(c, nd, 1.0 / std::sqrt(gp2), gprow, 0.0, nullptr, gprow));
    else if (have_perp) SCHK(launch_scale_by(c, gprow, nd, J.w.sc + SC_GPERP2, 0));
    else if (!(on_panel && J.gparts > 0)) HIPCHK(s_memset0(c, gprow, (size_t)J.ldw * sizeof(double)));
    const int ncopy = ncl > 0 ? std::min(a->order, ncl - 1) : 0;
    const int mm = rn + (ncl > 0 ? 1 : 0) + ncopy;
    std::vector<double> ev(mm), gh(mm);
    std::vector<int> idx(mm);
    {
        int i = 0, p = 0;
        auto mu_of = [&](int q) { return J.hD[kept[q]]; };
        auto put = [&](int q) { ev[p] = mu_of(q); idx[p] = kept[q]; gh[p] = -J.hghat[kept[q]]; ++p; };
        while (i < rn && mu_of(i) < J.lam0) put(i++);
        if (ncl > 0) {
            ev[p] = J.lam0; idx[p] = nrj; gh[p] = have_perp ? std::sqrt(gp2) : 0.0; ++p;
            for (int q = 0; q < ncopy; ++q) { ev[p] = J.lam0; idx[p] = nrj + 1; gh[p] = 0.0; ++p; }
        }
        while (i < rn) put(i++);
    }
    sella_stepper* st = nullptr;
    // the family reads its modes where they are — rows of the panel the update just wrote (trial steps of the per-atom
    // measure, the final step of the trust-region one): no gather, no transpose, no zero-filled mode matrices
    if (on_panel) {
        SCHK(stepper_on_panel(c, a->stepper_kind, J.Wnew, J.ldw, idx.data(), mm, nd, ev.data(), gh.data(), a->order, &st));
        if (have_perp && J.gparts > 0)
            for (int q = 0; q < mm; ++q)
                if (idx[q] == nrj) stepper_panel_scale(st, q, 1.0 / std::sqrt(gp2));
    } else
        SCHK(stepper_from_panel(c, a->stepper_kind, J.Wnew, J.ldw, idx.data(), mm, nd, ev.data(), gh.data(), a->order, &st));
    stepper_set_fast_search(st, c->opt.rs_fast != 0, on_boundary);
    stepper_set_alpha_hint(st, c->opt.rs_hint ? a->alpha_hint : 0.0);
    const bool qn = a->stepper_kind == SELLA_STEP_QN;
    const double alpha0 = qn ? 0.0 : 1.0, alphamax = qn ? std::numeric_limits<double>::infinity() : 1.0;
    const auto ts4 = stamp();
    const int rc = sella_restricted_step(st, a->cons, a->delta, nullptr, nullptr, nullptr, alpha0, 0.0, alphamax,
                                         qn ? -1.0 : 1.0, qn ? 1 : 0, 1, a->tol, a->maxiter, view ? a->idx : nullptr,
                                         view ? n : 0, a->s_out, &a->smag_out, nullptr, &a->nalpha);
    if (rc == SELLA_OK && stepper_alpha_found(st) > 0.0) a->alpha_hint = stepper_alpha_found(st);
    sella_stepper_destroy(st);
    if (step_timing) {
        const auto ts5 = stamp();
        auto us = [](auto x, auto y) { return std::chrono::duration<double, std::micro>(y - x).count(); };
        fprintf(stderr, "step host side: stage + queue %.1f us, wait %.1f us, commit + ratio %.1f us, family %.1f us, restricted step %.1f us\n",
                us(ts0, ts1), us(ts1, ts2), us(ts2, ts3), us(ts3, ts4), us(ts4, ts5));
    }
    return rc;
}

}  // namespace sella

using namespace sella;

// Dense mirror of a structured decomposition: B = lam0 I + W^T diag(mu - lam0) W.
extern "C" int sella_lr_materialize(sella_ctx* c, sella_mat hB, sella_mat hWt, int r, const double* mu, double lam0) {
    Mat *B = mat_get(c, hB), *Wm = mat_get(c, hWt);
    if (!B || B->rows != B->cols || r < 0 || (r > 0 && (!Wm || !mu || Wm->cols != B->rows || Wm->rows < r))) {
        set_error("lr_materialize: invalid arguments");
        return SELLA_E_INVALID;
    }
    const int n = B->rows;
    HIPCHK(s_memset0(c, B->d, (size_t)n * B->ld * sizeof(double)));
    if (r > 0) {
        const int ld = Wm->ld;
        double *scaled, *dmu;
        SCHK(scratch_get(c, SCR_MISC0, (size_t)r * ld * sizeof(double), &scaled));
        SCHK(scratch_get(c, SCR_MISC1, (size_t)round_up(r, 8) * sizeof(double), &dmu));
        SCHK(h2d_async(c, dmu, mu, (size_t)r * sizeof(double)));
        B = mat_get(c, hB);
        Wm = mat_get(c, hWt);
        SELLA_LAUNCHB(c, lr_scale_rows_kernel, lr_scale_rows_vb, 256, dim3((n + 255) / 256, r), dim3(256), 0, Wm->d, ld, r, n, dmu, lam0,
                           scaled, ld);
        HIPCHK(hipGetLastError());
        SCHK(launch_gemm(c, 1, 0, n, n, r, 1.0, Wm->d, ld, scaled, ld, 0.0, B->d, B->ld));
        SCHK(launch_symmetrize(c, B->d, n, B->ld));          // (the two triangles are summed in different orders)
    }
    return sella_mat_add_diag(c, hB, lam0);
}
