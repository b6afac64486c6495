// kernels.hip — streaming kernels of libsella_hip for gfx950 (CDNA4, wave64).
//
// Everything on the Sella hot path is fp64 and, at 3N ~ 3072, HBM/L2-bandwidth bound
// (SURVEY.md §8d): the work-horse is a row-panel matvec that streams a row-major matrix
// once with 16-byte coalesced loads, keeps the right-hand sides in LDS and reduces each row
// inside one 64-lane wavefront.  Panels of Krylov vectors are stored vector-major
// (k rows x n), so the same kernel computes V^T t, and all O(n k) updates are coalesced.
#include "internal.h"

namespace sella {

// ------------------------------------------------------------------------------------
// wave64 butterfly sum
// ------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) { return wave_sum64(v); }

// ------------------------------------------------------------------------------------
// K1: row-panel matvec.  A 256-thread workgroup owns RB consecutive rows; its four wavefronts
// stride each row together in 16-byte pieces (4 KiB per workgroup-instruction, fully coalesced,
// two such loads in flight per lane per row), the right-hand sides are read straight from
// global memory (24 KiB per vector at n = 3072: L1/L2 resident), each wavefront reduces with
// the wave64 butterfly and the four partials meet in LDS.  Measured on MI355X at n = 3072
// (tools/lab/gemv_lab.hip): 6.07 TB/s for 1 RHS, 5.7 TB/s for 2 — against 4.9 / 4.0 TB/s for the
// wavefront-per-row + LDS-staged-x variant this replaced.  Algorithmic traffic: 8*rows*cols bytes.
// ------------------------------------------------------------------------------------
template <int NRHS, int RB>
__device__ __forceinline__ void gemv_rows_vb(const VB vb, const double* __restrict__ A, int rows,
                                                        int cols, int lda,
                                                        GemvX xp,
                                                        double* __restrict__ Y, int ldy,
                                                        GemvEpi epi) {
    __shared__ double red[4][RB][NRHS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = vb.x * RB;
    const double2* arow[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        int rr = row0 + r;
        if (rr > rows - 1) rr = rows - 1;
        arow[r] = reinterpret_cast<const double2*>(A + (size_t)rr * lda);
    }
    double acc[RB][NRHS];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int h = 0; h < NRHS; ++h) acc[r][h] = 0.0;
    const int n2 = (cols + 1) >> 1;
    const bool odd = cols & 1;
    for (int j0 = threadIdx.x; j0 < n2; j0 += 512) {
        const int j1 = j0 + 256;
        const bool has1 = j1 < n2;
        double2 a0[RB], a1[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            a0[r] = arow[r][j0];
            a1[r] = has1 ? arow[r][j1] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int h = 0; h < NRHS; ++h) {
            const double* xh = xp.p[h];
            double2 x0, x1;
            x0.x = xh[2 * j0];
            x0.y = (odd && j0 == n2 - 1) ? 0.0 : xh[2 * j0 + 1];
            if (has1) {
                x1.x = xh[2 * j1];
                x1.y = (odd && j1 == n2 - 1) ? 0.0 : xh[2 * j1 + 1];
            } else {
                x1 = make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int r = 0; r < RB; ++r)
                acc[r][h] += a0[r].x * x0.x + a0[r].y * x0.y + a1[r].x * x1.x + a1[r].y * x1.y;
        }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int h = 0; h < NRHS; ++h) {
            const double v = wave_sum(acc[r][h]);
            if (lane == 0) red[wave][r][h] = v;
        }
    __syncthreads();
    if (threadIdx.x < RB * NRHS) {
        const int r = threadIdx.x / NRHS, h = threadIdx.x % NRHS;
        const int rr = row0 + r;
        if (rr < rows) {
            double v = epi.alpha * (red[0][r][h] + red[1][r][h] + red[2][r][h] + red[3][r][h]);
            if (epi.mode == 1) {
                // (P - theta I)^-1 in P's eigenbasis; a shift that hits an eigenvalue exactly (the reference's LU solve
                // raises LinAlgError there) gets the denominator a nearly singular solve would see instead of 1 / 0
                double den = epi.dvec[rr] - epi.theta;
                if (den == 0.0) den = 2.220446049250313e-16 * fmax(fabs(epi.theta), 2.2250738585072014e-308);
                v = v / den;
            }
            else if (epi.mode == 2) v += epi.beta * Y[(size_t)h * ldy + rr];
            else if (epi.mode == 3) v *= fabs(epi.dvec[rr]);
            else if (epi.mode == 4) v *= epi.dvec[rr];
            else if (epi.mode == 5) {
                // structured (P - theta I)^-1: explicit eigenpair i contributes 1 / (d_i - theta) - 1 / (lam0 - theta) on
                // top of the scaled identity of the complement; beta carries 1 / (lam0 - theta)
                double den = epi.dvec[rr] - epi.theta;
                if (den == 0.0) den = 2.220446049250313e-16 * fmax(fabs(epi.theta), 2.2250738585072014e-308);
                v *= 1.0 / den - epi.beta;
            }
            Y[(size_t)h * ldy + rr] = v;
        }
    }
}
template <int NRHS, int RB>
__global__ __launch_bounds__(256) void gemv_rows_kernel(const double* __restrict__ A, int rows,
                                                        int cols, int lda,
                                                        GemvX xp,
                                                        double* __restrict__ Y, int ldy,
                                                        GemvEpi epi) { gemv_rows_vb<NRHS, RB>(vb_hw(), A, rows, cols, lda, xp, Y, ldy, epi); }

template <int NRHS>
static int gemv_rows_dispatch_rw(sella_ctx* c, int rb, const double* A, int rows, int cols, int lda,
                                 const GemvX& xp, double* Y, int ldy, const GemvEpi& epi) {
    if (rb == 4) {
        SELLA_LAUNCHB_PROF(c, HIP_KERNEL_NAME(gemv_rows_kernel<NRHS, 4>), SELLA_BODY(gemv_rows_vb<NRHS, 4>), 256, dim3((rows + 3) / 4), dim3(256), 0,
                     A, rows, cols, lda, xp, Y, ldy, epi);
    } else if (rb == 2) {
        SELLA_LAUNCHB_PROF(c, HIP_KERNEL_NAME(gemv_rows_kernel<NRHS, 2>), SELLA_BODY(gemv_rows_vb<NRHS, 2>), 256, dim3((rows + 1) / 2), dim3(256), 0,
                     A, rows, cols, lda, xp, Y, ldy, epi);
    } else {
        SELLA_LAUNCHB_PROF(c, HIP_KERNEL_NAME(gemv_rows_kernel<NRHS, 1>), SELLA_BODY(gemv_rows_vb<NRHS, 1>), 256, dim3(rows), dim3(256), 0,
                     A, rows, cols, lda, xp, Y, ldy, epi);
    }
    HIPCHK(hipGetLastError());
    return SELLA_OK;
}

int launch_gemv_rows_xp(sella_ctx* c, const double* A, int rows, int cols, int lda,
                        const double* const* xs, int nrhs, double* Y, int ldy, const GemvEpi& epi) {
    if (rows <= 0 || cols <= 0 || nrhs <= 0) return SELLA_OK;
    if ((lda & 1) || (((uintptr_t)A) & 15)) {
        set_error("gemv_rows: matrix must be 16-byte aligned with even leading dimension");
        return SELLA_E_INVALID;
    }
    int rw = (int)c->opt.gemv_rw;
    // 0 = by size: beyond ~4000 rows fewer, fatter workgroups stream faster (n = 6144, 2 right-hand sides: 4 rows per
    // workgroup 47 us = 6.4 TB/s against 55 us with 2; equal at n = 3072 — tools/lab/gemv_rw.py)
    if (rw == 0) rw = rows >= 4096 ? 4 : 2;
    // few rows (panel dots, small matrices): one row per workgroup keeps more of the chip busy
    if (rows < 1024) rw = 1;
    // 8 right-hand sides x 4 rows would need 64 accumulators per lane
    if (nrhs > 4 && rw == 4) rw = 2;
    const int kind = (rows >= 1024 && cols >= 1024) ? PROF_GEMV : PROF_GEMV_SMALL;
    for (int h0 = 0; h0 < nrhs; h0 += 8) {
        const int nh = (nrhs - h0 < 8) ? (nrhs - h0) : 8;
        GemvX xp;
        for (int h = 0; h < 8; ++h) xp.p[h] = xs[h0 + (h < nh ? h : 0)];
        double* Yh = Y + (size_t)h0 * ldy;
        prof_begin(c, kind, 8.0 * rows * (double)cols, 2.0 * rows * (double)cols * nh);
        int st;
        switch (nh) {
            case 1: st = gemv_rows_dispatch_rw<1>(c, rw, A, rows, cols, lda, xp, Yh, ldy, epi); break;
            case 2: st = gemv_rows_dispatch_rw<2>(c, rw, A, rows, cols, lda, xp, Yh, ldy, epi); break;
            case 3: st = gemv_rows_dispatch_rw<3>(c, rw, A, rows, cols, lda, xp, Yh, ldy, epi); break;
            case 4: st = gemv_rows_dispatch_rw<4>(c, rw, A, rows, cols, lda, xp, Yh, ldy, epi); break;
            case 5: st = gemv_rows_dispatch_rw<5>(c, rw, A, rows, cols, lda, xp, Yh, ldy, epi); break;
            case 6: st = gemv_rows_dispatch_rw<6>(c, rw, A, rows, cols, lda, xp, Yh, ldy, epi); break;
            case 7: st = gemv_rows_dispatch_rw<7>(c, rw, A, rows, cols, lda, xp, Yh, ldy, epi); break;
            default: st = gemv_rows_dispatch_rw<8>(c, rw, A, rows, cols, lda, xp, Yh, ldy, epi); break;
        }
        prof_end(c);
        SCHK(st);
    }
    return SELLA_OK;
}

int launch_gemv_rows(sella_ctx* c, const double* A, int rows, int cols, int lda, const double* X,
                     int ldx, int nrhs, double* Y, int ldy, const GemvEpi& epi) {
    if (nrhs <= 0) return SELLA_OK;
    std::vector<const double*> xs(nrhs);
    for (int h = 0; h < nrhs; ++h) xs[h] = X + (size_t)h * ldx;
    return launch_gemv_rows_xp(c, A, rows, cols, lda, xs.data(), nrhs, Y, ldy, epi);
}

// Two row sources, one right-hand side: rows [0, rows) come from A, rows [rows, rows + rows2) from A2 (same column
// count); same streaming structure as gemv_rows_kernel<1, 2>.
__device__ __forceinline__ void gemv_rows2_vb(const VB vb, const double* __restrict__ A, int rows, int lda,
                                                         const double* __restrict__ A2, int rows2, int lda2, int cols,
                                                         const double* __restrict__ x, double* __restrict__ y,
                                                         double* __restrict__ y2) {
    constexpr int RB = 2;
    __shared__ double red[4][RB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = vb.x * RB, rtot = rows + rows2;
    const double2* arow[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        int rr = row0 + r;
        if (rr > rtot - 1) rr = rtot - 1;
        arow[r] = reinterpret_cast<const double2*>(rr < rows ? A + (size_t)rr * lda : A2 + (size_t)(rr - rows) * lda2);
    }
    double acc[RB] = {0.0, 0.0};
    const int n2 = (cols + 1) >> 1;
    const bool odd = cols & 1;
    for (int j0 = threadIdx.x; j0 < n2; j0 += 512) {
        const int j1 = j0 + 256;
        const bool has1 = j1 < n2;
        double2 a0[RB], a1[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            a0[r] = arow[r][j0];
            a1[r] = has1 ? arow[r][j1] : make_double2(0.0, 0.0);
        }
        double2 x0, x1;
        x0.x = x[2 * j0];
        x0.y = (odd && j0 == n2 - 1) ? 0.0 : x[2 * j0 + 1];
        if (has1) {
            x1.x = x[2 * j1];
            x1.y = (odd && j1 == n2 - 1) ? 0.0 : x[2 * j1 + 1];
        } else {
            x1 = make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] += a0[r].x * x0.x + a0[r].y * x0.y + a1[r].x * x1.x + a1[r].y * x1.y;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const double v = wave_sum(acc[r]);
        if (lane == 0) red[wave][r] = v;
    }
    __syncthreads();
    if (threadIdx.x < RB) {
        const int rr = row0 + threadIdx.x;
        if (rr < rtot) {
            const double v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
            if (rr < rows) y[rr] = v; else y2[rr - rows] = v;
        }
    }
}
__global__ __launch_bounds__(256) void gemv_rows2_kernel(const double* __restrict__ A, int rows, int lda,
                                                         const double* __restrict__ A2, int rows2, int lda2, int cols,
                                                         const double* __restrict__ x, double* __restrict__ y,
                                                         double* __restrict__ y2) { gemv_rows2_vb(vb_hw(), A, rows, lda, A2, rows2, lda2, cols, x, y, y2); }

int launch_gemv_rows2(sella_ctx* c, const double* A, int rows, int lda, const double* A2, int rows2, int lda2, int cols,
                      const double* x, double* y, double* y2) {
    if (rows + rows2 <= 0 || cols <= 0) return SELLA_OK;
    if ((lda & 1) || (lda2 & 1) || (((uintptr_t)A) & 15) || (((uintptr_t)A2) & 15)) {
        set_error("gemv_rows2: matrices must be 16-byte aligned with even leading dimensions");
        return SELLA_E_INVALID;
    }
    prof_begin(c, PROF_GEMV, 8.0 * (rows + rows2) * (double)cols, 2.0 * (rows + rows2) * (double)cols);
    SELLA_LAUNCHB_PROF(c, gemv_rows2_kernel, gemv_rows2_vb, 256, dim3((rows + rows2 + 1) / 2), dim3(256), 0, A, rows, lda, A2, rows2, lda2, cols, x, y, y2);
    prof_end(c);
    HIPCHK(hipGetLastError());
    return SELLA_OK;
}

// |x|^2 -> out[0], then x /= |x|  in ONE single-workgroup launch (n up to a few 10^4)
__device__ __forceinline__ void normalize_vb(const VB vb, double* __restrict__ x, int n, double* __restrict__ out) {
    __shared__ double red[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) s += x[i] * x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += red[w];
    const double f = 1.0 / sqrt(tot);
    for (int i = threadIdx.x; i < n; i += 1024) x[i] *= f;
    if (threadIdx.x == 0) out[0] = tot;
}
__global__ __launch_bounds__(1024) void normalize_kernel(double* __restrict__ x, int n, double* __restrict__ out) { normalize_vb(vb_hw(), x, n, out); }

int launch_normalize(sella_ctx* c, double* x, int n, double* out) {
    SELLA_LAUNCHB(c, normalize_kernel, normalize_vb, 1024, dim3(1), dim3(1024), 0, x, n, out);
    HIPCHK(hipGetLastError());
    return SELLA_OK;
}

// ------------------------------------------------------------------------------------
// K1t: transposed product Y = A^T X without a transposed copy.  Each thread owns two
// adjacent columns (16-byte coalesced loads along the row), row blocks are split over
// blockIdx.y into a partial buffer that a second kernel sums in a fixed order
// (deterministic; no atomics).
// ------------------------------------------------------------------------------------
constexpr int GEMVT_ROWS = 32;

template <int NRHS>
__device__ __forceinline__ void gemv_cols_partial_vb(const VB vb, const double* __restrict__ A,
                                                                int rows, int cols, int lda,
                                                                const double* __restrict__ X,
                                                                int ldx, double* __restrict__ part,
                                                                int ldpart) {
    __shared__ double xs[NRHS * GEMVT_ROWS];
    const int r0 = vb.y * GEMVT_ROWS;
    const int nr = (rows - r0 < GEMVT_ROWS) ? (rows - r0) : GEMVT_ROWS;
    for (int t = threadIdx.x; t < NRHS * GEMVT_ROWS; t += 256) {
        const int h = t / GEMVT_ROWS, r = t % GEMVT_ROWS;
        xs[t] = (r < nr) ? X[(size_t)h * ldx + r0 + r] : 0.0;
    }
    __syncthreads();
    const int cp = vb.x * 256 + threadIdx.x;
    const int cols2 = (cols + 1) & ~1;
    if (2 * cp >= cols2) return;
    double2 acc[NRHS];
#pragma unroll
    for (int h = 0; h < NRHS; ++h) acc[h] = make_double2(0.0, 0.0);
    const double* ap = A + (size_t)r0 * lda + 2 * cp;
#pragma unroll 4
    for (int r = 0; r < nr; ++r) {
        const double2 a = *reinterpret_cast<const double2*>(ap + (size_t)r * lda);
#pragma unroll
        for (int h = 0; h < NRHS; ++h) {
            const double x = xs[h * GEMVT_ROWS + r];
            acc[h].x += a.x * x;
            acc[h].y += a.y * x;
        }
    }
#pragma unroll
    for (int h = 0; h < NRHS; ++h)
        *reinterpret_cast<double2*>(part + ((size_t)vb.y * NRHS + h) * ldpart + 2 * cp) = acc[h];
}
template <int NRHS>
__global__ __launch_bounds__(256) void gemv_cols_partial_kernel(const double* __restrict__ A,
                                                                int rows, int cols, int lda,
                                                                const double* __restrict__ X,
                                                                int ldx, double* __restrict__ part,
                                                                int ldpart) { gemv_cols_partial_vb<NRHS>(vb_hw(), A, rows, cols, lda, X, ldx, part, ldpart); }

__device__ __forceinline__ void gemv_cols_reduce_vb(const VB vb, const double* __restrict__ part,
                                                               int ldpart, int nsplit, int nrhs,
                                                               int cols, double* __restrict__ Y,
                                                               int ldy) {
    const int j = vb.x * 256 + threadIdx.x;
    const int h = vb.y;
    if (j >= cols) return;
    double s = 0.0;
    for (int sp = 0; sp < nsplit; ++sp) s += part[((size_t)sp * nrhs + h) * ldpart + j];
    Y[(size_t)h * ldy + j] = s;
}
__global__ __launch_bounds__(256) void gemv_cols_reduce_kernel(const double* __restrict__ part,
                                                               int ldpart, int nsplit, int nrhs,
                                                               int cols, double* __restrict__ Y,
                                                               int ldy) { gemv_cols_reduce_vb(vb_hw(), part, ldpart, nsplit, nrhs, cols, Y, ldy); }

template <int NRHS>
static int gemv_cols_run(sella_ctx* c, const double* A, int rows, int cols, int lda, const double* X,
                         int ldx, double* Y, int ldy) {
    const int cols2 = (cols + 1) & ~1;
    const int ldpart = round_up(cols2, 8);
    const int nsplit = (rows + GEMVT_ROWS - 1) / GEMVT_ROWS;
    double* part;
    SCHK(scratch_get(c, SCR_PART, (size_t)nsplit * NRHS * ldpart * sizeof(double), &part));
    dim3 grid((cols2 / 2 + 255) / 256, nsplit);
    prof_begin(c, PROF_GEMV, 8.0 * rows * (double)cols, 2.0 * rows * (double)cols * NRHS);
    SELLA_LAUNCHB_PROF(c, HIP_KERNEL_NAME(gemv_cols_partial_kernel<NRHS>), SELLA_BODY(gemv_cols_partial_vb<NRHS>), 256, grid, dim3(256), 0,
                 A, rows, cols, lda, X, ldx, part, ldpart);
    prof_end(c);
    HIPCHK(hipGetLastError());
    SELLA_LAUNCHB(c, gemv_cols_reduce_kernel, gemv_cols_reduce_vb, 256, dim3((cols + 255) / 256, NRHS), dim3(256), 0,
                       part, ldpart, nsplit, NRHS, cols, Y, ldy);
    HIPCHK(hipGetLastError());
    return SELLA_OK;
}

int launch_gemv_cols(sella_ctx* c, const double* A, int rows, int cols, int lda, const double* X,
                     int ldx, int nrhs, double* Y, int ldy) {
    if (rows <= 0 || cols <= 0 || nrhs <= 0) return SELLA_OK;
    if ((lda & 1) || (((uintptr_t)A) & 15)) {
        set_error("gemv_cols: matrix must be 16-byte aligned with even leading dimension");
        return SELLA_E_INVALID;
    }
    for (int h0 = 0; h0 < nrhs; h0 += 4) {
        const int nh = (nrhs - h0 < 4) ? (nrhs - h0) : 4;
        const double* Xh = X + (size_t)h0 * ldx;
        double* Yh = Y + (size_t)h0 * ldy;
        switch (nh) {
            case 1: SCHK(gemv_cols_run<1>(c, A, rows, cols, lda, Xh, ldx, Yh, ldy)); break;
            case 2: SCHK(gemv_cols_run<2>(c, A, rows, cols, lda, Xh, ldx, Yh, ldy)); break;
            case 3: SCHK(gemv_cols_run<3>(c, A, rows, cols, lda, Xh, ldx, Yh, ldy)); break;
            default: SCHK(gemv_cols_run<4>(c, A, rows, cols, lda, Xh, ldx, Yh, ldy)); break;
        }
    }
    return SELLA_OK;
}

// ------------------------------------------------------------------------------------
// K2: linear combinations of panel rows (basis rotation V <- V W, residuals, Gram-Schmidt
// updates).  Thread i owns element i of every vector: all loads/stores are coalesced.
// ------------------------------------------------------------------------------------
constexpr int LC_NT = 8;     // outputs per thread
constexpr int LC_JT = 128;   // panel rows per LDS tile of coefficients

__device__ __forceinline__ void lincomb_vb(const VB vb, int n, int nout, const double* __restrict__ P1,
                                                      int ldp1, int k1,
                                                      const double* __restrict__ W1, int ldw1,
                                                      const double* __restrict__ P2, int ldp2,
                                                      int k2, const double* __restrict__ W2,
                                                      int ldw2, double beta,
                                                      double* __restrict__ out, int ldo) {
    __shared__ double ws[LC_JT * LC_NT];
    const int i = vb.x * 256 + threadIdx.x;
    const int c0 = vb.y * LC_NT;
    double acc[LC_NT];
#pragma unroll
    for (int c = 0; c < LC_NT; ++c) acc[c] = 0.0;
    for (int pass = 0; pass < 2; ++pass) {
        const double* P = pass ? P2 : P1;
        const double* W = pass ? W2 : W1;
        const int ldp = pass ? ldp2 : ldp1;
        const int ldw = pass ? ldw2 : ldw1;
        const int k = pass ? k2 : k1;
        if (P == nullptr || k <= 0) continue;
        for (int j0 = 0; j0 < k; j0 += LC_JT) {
            const int jt = (k - j0 < LC_JT) ? (k - j0) : LC_JT;
            __syncthreads();
            for (int t = threadIdx.x; t < jt * LC_NT; t += 256) {
                const int j = t / LC_NT, cc = t % LC_NT;
                ws[t] = (c0 + cc < nout) ? W[(size_t)(j0 + j) * ldw + c0 + cc] : 0.0;
            }
            __syncthreads();
            if (i < n) {
                // sixteen panel rows in flight at a time (a loop unrolled by 4 is k / 4 memory round trips: 16 us for
                // k = 60 rows of 3072 — these passes sit between the kernels of an optimizer step)
                for (int jb = 0; jb < jt; jb += 16) {
                    double pv[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int j = jb + u;
                        pv[u] = P[(size_t)(j0 + (j < jt ? j : jt - 1)) * ldp + i];
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int j = jb + u;
                        if (j < jt) {
#pragma unroll
                            for (int c = 0; c < LC_NT; ++c) acc[c] += ws[j * LC_NT + c] * pv[u];
                        }
                    }
                }
            }
        }
    }
    if (i < n) {
#pragma unroll
        for (int c = 0; c < LC_NT; ++c) {
            if (c0 + c < nout) {
                double* o = out + (size_t)(c0 + c) * ldo + i;
                *o = (beta == 0.0) ? acc[c] : (beta * (*o) + acc[c]);
            }
        }
    }
}
__global__ __launch_bounds__(256) void lincomb_kernel(int n, int nout, const double* __restrict__ P1,
                                                      int ldp1, int k1,
                                                      const double* __restrict__ W1, int ldw1,
                                                      const double* __restrict__ P2, int ldp2,
                                                      int k2, const double* __restrict__ W2,
                                                      int ldw2, double beta,
                                                      double* __restrict__ out, int ldo) { lincomb_vb(vb_hw(), n, nout, P1, ldp1, k1, W1, ldw1, P2, ldp2, k2, W2, ldw2, beta, out, ldo); }

int launch_lincomb(sella_ctx* c, int n, int nout, const double* P1, int ldp1, int k1,
                   const double* W1, int ldw1, const double* P2, int ldp2, int k2,
                   const double* W2, int ldw2, double beta, double* out, int ldo) {
    if (n <= 0 || nout <= 0) return SELLA_OK;
    dim3 grid((n + 255) / 256, (nout + LC_NT - 1) / LC_NT);
    SELLA_LAUNCHB(c, lincomb_kernel, lincomb_vb, 256, grid, dim3(256), 0, n, nout, P1, ldp1, k1, W1, ldw1,
                       P2, ldp2, k2, W2, ldw2, beta, out, ldo);
    HIPCHK(hipGetLastError());
    return SELLA_OK;
}

// ------------------------------------------------------------------------------------
// small vector kernels
// ------------------------------------------------------------------------------------
// one workgroup per row: 256 lanes take 16-byte pieces, four loads in flight each
__device__ __forceinline__ void rows_sumsq_vb(const VB vb, const double* __restrict__ P, int ldp,
                                                         int nrows, int n, double* __restrict__ out) {
    __shared__ double red[4];
    const double* p = P + (size_t)vb.x * ldp;
    double s = 0.0;
    // usually ONE row (the residual of the lowest Ritz pair): a single workgroup walks it, so the loads are
    // issued four at a time instead of one per trip (12 dependent trips at n = 3072 cost 12 us)
    for (int i0 = threadIdx.x; i0 < n; i0 += 1024) {
        double v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + 256 * k;
            v[k] = p[i < n ? i : n - 1];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) s += (i0 + 256 * k < n) ? v[k] * v[k] : 0.0;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[vb.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void rows_sumsq_kernel(const double* __restrict__ P, int ldp,
                                                         int nrows, int n, double* __restrict__ out) { rows_sumsq_vb(vb_hw(), P, ldp, nrows, n, out); }

int launch_rows_sumsq(sella_ctx* c, const double* P, int ldp, int nrows, int n, double* out) {
    if (nrows <= 0) return SELLA_OK;
    SELLA_LAUNCHB(c, rows_sumsq_kernel, rows_sumsq_vb, 256, dim3(nrows), dim3(256), 0, P, ldp, nrows, n, out);
    HIPCHK(hipGetLastError());
    return SELLA_OK;
}

__device__ __forceinline__ void scale_by_vb(const VB vb, double* __restrict__ x, int n,
                                                       const double* __restrict__ scal, int mode) {
    const int i = vb.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double s = scal[0];
    double f;
    if (mode == 0) f = 1.0 / sqrt(s);
    else if (mode == 1) f = 1.0 / s;
    else f = s;
    x[i] *= f;
}
__global__ __launch_bounds__(256) void scale_by_kernel(double* __restrict__ x, int n,
                                                       const double* __restrict__ scal, int mode) { scale_by_vb(vb_hw(), x, n, scal, mode); }

int launch_scale_by(sella_ctx* c, double* x, int n, const double* scal, int mode) {
    SELLA_LAUNCHB(c, scale_by_kernel, scale_by_vb, 256, dim3((n + 255) / 256), dim3(256), 0, x, n, scal, mode);
    HIPCHK(hipGetLastError());
    return SELLA_OK;
}

__device__ __forceinline__ void jd_combine_vb(const VB vb, const double* __restrict__ x,
                                                         const double* __restrict__ y,
                                                         const double* __restrict__ dots,
                                                         double* __restrict__ t, int n) {
    const int i = vb.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double num = dots[0], den = dots[1];
    // eigensolvers.py:127-132: t = y*(v.x / v.y) - x, or x itself when v.y ~ 0
    t[i] = (fabs(den) < 1e-12) ? x[i] : (y[i] * (num / den) - x[i]);
}
__global__ __launch_bounds__(256) void jd_combine_kernel(const double* __restrict__ x,
                                                         const double* __restrict__ y,
                                                         const double* __restrict__ dots,
                                                         double* __restrict__ t, int n) { jd_combine_vb(vb_hw(), x, y, dots, t, n); }

int launch_jd_combine(sella_ctx* c, const double* x, const double* y, const double* dots, double* t,
                      int n) {
    SELLA_LAUNCHB(c, jd_combine_kernel, jd_combine_vb, 256, dim3((n + 255) / 256), dim3(256), 0, x, y, dots, t, n);
    HIPCHK(hipGetLastError());
    return SELLA_OK;
}

// (no __restrict__: callers may update in place)
__device__ __forceinline__ void axpby_vb(const VB vb, int n, double a, const double* x, double b,
                                                    const double* y, double* z) {
    const int i = vb.x * 256 + threadIdx.x;
    if (i >= n) return;
    double v = a * x[i];
    if (y != nullptr) v += b * y[i];
    z[i] = v;
}
__global__ __launch_bounds__(256) void axpby_kernel(int n, double a, const double* x, double b,
                                                    const double* y, double* z) { axpby_vb(vb_hw(), n, a, x, b, y, z); }

int launch_axpby(sella_ctx* c, int n, double a, const double* x, double b, const double* y, double* z) {
    if (n <= 0) return SELLA_OK;
    SELLA_LAUNCHB(c, axpby_kernel, axpby_vb, 256, dim3((n + 255) / 256), dim3(256), 0, n, a, x, b, y, z);
    HIPCHK(hipGetLastError());
    return SELLA_OK;
}

__device__ __forceinline__ void axpby2d_vb(const VB vb, int rows, int cols, double a,
                                                      const double* __restrict__ A, int lda, double b,
                                                      const double* __restrict__ B, int ldb,
                                                      double* __restrict__ C, int ldc) {
    const int j = vb.x * 256 + threadIdx.x;
    const int i = vb.y;
    if (j >= cols || i >= rows) return;
    double v = a * A[(size_t)i * lda + j];
    if (B != nullptr) v += b * B[(size_t)i * ldb + j];
    C[(size_t)i * ldc + j] = v;
}
__global__ __launch_bounds__(256) void axpby2d_kernel(int rows, int cols, double a,
                                                      const double* __restrict__ A, int lda, double b,
                                                      const double* __restrict__ B, int ldb,
                                                      double* __restrict__ C, int ldc) { axpby2d_vb(vb_hw(), rows, cols, a, A, lda, b, B, ldb, C, ldc); }

int launch_axpby2d(sella_ctx* c, int rows, int cols, double a, const double* A, int lda, double b,
                   const double* B, int ldb, double* C, int ldc) {
    if (rows <= 0 || cols <= 0) return SELLA_OK;
    SELLA_LAUNCHB(c, axpby2d_kernel, axpby2d_vb, 256, dim3((cols + 255) / 256, rows), dim3(256), 0, rows,
                       cols, a, A, lda, b, B, ldb, C, ldc);
    HIPCHK(hipGetLastError());
    return SELLA_OK;
}

// 32x32 tiles through LDS (+1 padding), coalesced on both sides.
__device__ __forceinline__ void transpose_vb(const VB vb, const double* __restrict__ A, int rows,
                                                        int cols, int lda, double* __restrict__ At,
                                                        int ldat) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int r0 = vb.y * 32, c0 = vb.x * 32;
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, cc = c0 + tx;
        tile[k][tx] = (r < rows && cc < cols) ? A[(size_t)r * lda + cc] : 0.0;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int cc = c0 + k, r = r0 + tx;   // At[cc][r]
        if (cc < cols && r < rows) At[(size_t)cc * ldat + r] = tile[tx][k];
    }
}
__global__ __launch_bounds__(256) void transpose_kernel(const double* __restrict__ A, int rows,
                                                        int cols, int lda, double* __restrict__ At,
                                                        int ldat) { transpose_vb(vb_hw(), A, rows, cols, lda, At, ldat); }

int launch_transpose(sella_ctx* c, const double* A, int rows, int cols, int lda, double* At, int ldat) {
    if (rows <= 0 || cols <= 0) return SELLA_OK;
    SELLA_LAUNCHB(c, transpose_kernel, transpose_vb, 256, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, A, rows, cols, lda, At, ldat);
    HIPCHK(hipGetLastError());
    return SELLA_OK;
}

// In-place symmetrisation B <- (B + B^T)/2: block (bi, bj) with bi <= bj handles both tiles.
__device__ __forceinline__ void symmetrize_vb(const VB vb, double* __restrict__ B, int n, int ld) {
    if (vb.x < vb.y) return;   // uniform per block
    __shared__ double t1[32][33];
    __shared__ double t2[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = vb.y * 32, c0 = vb.x * 32;     // tile (r0, c0), mirror (c0, r0)
    for (int k = ty; k < 32; k += 8) {
        int r = r0 + k, cc = c0 + tx;
        t1[k][tx] = (r < n && cc < n) ? B[(size_t)r * ld + cc] : 0.0;
        r = c0 + k; cc = r0 + tx;
        t2[k][tx] = (r < n && cc < n) ? B[(size_t)r * ld + cc] : 0.0;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        int r = r0 + k, cc = c0 + tx;
        if (r < n && cc < n) B[(size_t)r * ld + cc] = 0.5 * (t1[k][tx] + t2[tx][k]);
        r = c0 + k; cc = r0 + tx;
        if (r < n && cc < n) B[(size_t)r * ld + cc] = 0.5 * (t2[k][tx] + t1[tx][k]);
    }
}
__global__ __launch_bounds__(256) void symmetrize_kernel(double* __restrict__ B, int n, int ld) { symmetrize_vb(vb_hw(), B, n, ld); }

int launch_symmetrize(sella_ctx* c, double* B, int n, int ld) {
    if (n <= 0) return SELLA_OK;
    const int nb = (n + 31) / 32;
    SELLA_LAUNCHB(c, symmetrize_kernel, symmetrize_vb, 256, dim3(nb, nb), dim3(256), 0, B, n, ld);
    HIPCHK(hipGetLastError());
    return SELLA_OK;
}

__device__ __forceinline__ void gather_rows_vb(const VB vb, const double* __restrict__ in, int ldi,
                                                          const int* __restrict__ idx, int nrows,
                                                          int ncols, double* __restrict__ out,
                                                          int ldo) {
    const int j = vb.x * 256 + threadIdx.x;
    const int r = vb.y;
    if (j >= ncols || r >= nrows) return;
    out[(size_t)r * ldo + j] = in[(size_t)idx[r] * ldi + j];
}
__global__ __launch_bounds__(256) void gather_rows_kernel(const double* __restrict__ in, int ldi,
                                                          const int* __restrict__ idx, int nrows,
                                                          int ncols, double* __restrict__ out,
                                                          int ldo) { gather_rows_vb(vb_hw(), in, ldi, idx, nrows, ncols, out, ldo); }

int launch_gather_rows(sella_ctx* c, const double* in, int ldi, const int* idx, int nrows, int ncols,
                       double* out, int ldo) {
    if (nrows <= 0 || ncols <= 0) return SELLA_OK;
    SELLA_LAUNCHB(c, gather_rows_kernel, gather_rows_vb, 256, dim3((ncols + 255) / 256, nrows), dim3(256), 0, in,
                       ldi, idx, nrows, ncols, out, ldo);
    HIPCHK(hipGetLastError());
    return SELLA_OK;
}

// ------------------------------------------------------------------------------------
// GEMM, fp64.  C (M x N) = alpha * op(A) op(B) + beta * C, everything row-major.
// 64x64 output tile per 256-thread workgroup, K stepped by 16 through LDS (k-major tiles,
// row stride 80 doubles so the two 16-lane halves of a 32-lane LDS group hit disjoint banks).
//   MFMA variant : each wavefront owns a 32x32 quadrant = 2x2 v_mfma_f64_16x16x4_f64 tiles
//                  (A frag: lane l holds A[l&15][l>>4]; B frag: B[l>>4][l&15];
//                   C/D: col = l&15, row = (l>>4) + 4*reg).
//   VALU variant : each thread owns a 4x4 register tile.
// Edge tiles are zero-filled on load and masked on store, so any M, N, K, ld are accepted.
// ------------------------------------------------------------------------------------
constexpr int GM_BM = 64, GM_BN = 64, GM_BK = 16, GM_LD = 80;

typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void gemm_load_tiles(const double* __restrict__ A, int lda, int transA,
                                                const double* __restrict__ B, int ldb, int transB,
                                                int M, int N, int K, int m0, int n0, int k0,
                                                double (*As)[GM_LD], double (*Bs)[GM_LD]) {
    // As[k][m] = op(A)[m0+m][k0+k];  Bs[k][n] = op(B)[k0+k][n0+n];  1024 elements each
    for (int t = threadIdx.x; t < GM_BM * GM_BK; t += 256) {
        int m, k;
        if (transA) { m = t & 63; k = t >> 6; }     // contiguous along m in memory
        else { k = t & 15; m = t >> 4; }            // contiguous along k in memory
        const int gm = m0 + m, gk = k0 + k;
        double v = 0.0;
        if (gm < M && gk < K) v = transA ? A[(size_t)gk * lda + gm] : A[(size_t)gm * lda + gk];
        As[k][m] = v;
    }
    for (int t = threadIdx.x; t < GM_BN * GM_BK; t += 256) {
        int n, k;
        if (transB) { k = t & 15; n = t >> 4; }     // B is N x K: contiguous along k
        else { n = t & 63; k = t >> 6; }            // B is K x N: contiguous along n
        const int gn = n0 + n, gk = k0 + k;
        double v = 0.0;
        if (gn < N && gk < K) v = transB ? B[(size_t)gn * ldb + gk] : B[(size_t)gk * ldb + gn];
        Bs[k][n] = v;
    }
}

__device__ __forceinline__ void gemm_mfma_tile(int transA, int transB, int M, int N, int K, double alpha,
                                               const double* __restrict__ A, int lda,
                                               const double* __restrict__ B, int ldb, double beta,
                                               double* __restrict__ C, int ldc, int m0, int n0,
                                               double (*As)[GM_LD], double (*Bs)[GM_LD]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int li = lane & 15, lk = lane >> 4;
    f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (f64x4){0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < K; k0 += GM_BK) {
        __syncthreads();
        gemm_load_tiles(A, lda, transA, B, ldb, transB, M, N, K, m0, n0, k0, As, Bs);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GM_BK; kk += 4) {
            double af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = As[kk + lk][wm + a * 16 + li];
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b] = Bs[kk + lk][wn + b * 16 + li];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = m0 + wm + a * 16 + lk + 4 * r;
                const int gn = n0 + wn + b * 16 + li;
                if (gm < M && gn < N) {
                    double* cp = C + (size_t)gm * ldc + gn;
                    const double v = alpha * acc[a][b][r];
                    *cp = (beta == 0.0) ? v : (v + beta * (*cp));
                }
            }
}

__device__ __forceinline__ void gemm_mfma_vb(const VB vb, int transA, int transB, int M, int N, int K,
                                                        double alpha, const double* __restrict__ A,
                                                        int lda, const double* __restrict__ B,
                                                        int ldb, double beta, double* __restrict__ C,
                                                        int ldc) {
    __shared__ double As[GM_BK][GM_LD];
    __shared__ double Bs[GM_BK][GM_LD];
    gemm_mfma_tile(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, vb.y * GM_BM,
                   vb.x * GM_BN, As, Bs);
}
__global__ __launch_bounds__(256) void gemm_mfma_kernel(int transA, int transB, int M, int N, int K,
                                                        double alpha, const double* __restrict__ A,
                                                        int lda, const double* __restrict__ B,
                                                        int ldb, double beta, double* __restrict__ C,
                                                        int ldc) { gemm_mfma_vb(vb_hw(), transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc); }

// one launch for all merges of a divide-and-conquer level (blockIdx.z = merge)
__global__ __launch_bounds__(256) void gemm_merge_batched_kernel(const int* __restrict__ desc,
                                                                 const double* __restrict__ A,
                                                                 const double* __restrict__ B,
                                                                 double* __restrict__ C, int ld) {
    __shared__ double As[GM_BK][GM_LD];
    __shared__ double Bs[GM_BK][GM_LD];
    const int lo = desc[4 * blockIdx.z], N = desc[4 * blockIdx.z + 1], K = desc[4 * blockIdx.z + 2];
    const int m0 = blockIdx.y * GM_BM, n0 = blockIdx.x * GM_BN;
    if (m0 >= K || n0 >= N) return;                          // whole workgroup leaves together
    const size_t off = (size_t)lo * ld + lo;
    gemm_mfma_tile(0, 0, K, N, K, 1.0, A + off, ld, B + off, ld, 0.0, C + off, ld, m0, n0, As, Bs);
}

// ---- 128 x 128 x 16 tiles, double-buffered -----------------------------------------------------------
// For the large NN / TN products (divide-and-conquer merges, eigen-update merges, U^T H U): a workgroup
// owns a 128 x 128 tile of C, each wavefront a 64 x 64 quadrant = 4 x 4 MFMA tiles (64 MFMAs between two
// barriers instead of 16), and the global loads of k-step t+1 are issued into registers before the MFMAs
// of step t and stored to the other LDS buffer afterwards, so one workgroup hides its own load latency.
constexpr int G2_BM = 128, G2_BN = 128, G2_BK = 16, G2_LD = 129;

__device__ __forceinline__ void gemm128_tile(int transA, int M, int N, int K, double alpha,
                                             const double* __restrict__ A, int lda,
                                             const double* __restrict__ B, int ldb, double beta,
                                             double* __restrict__ C, int ldc, int m0, int n0,
                                             double (*As)[G2_BK][G2_LD], double (*Bs)[G2_BK][G2_LD]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int li = lane & 15, lk = lane >> 4;
    f64x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f64x4){0.0, 0.0, 0.0, 0.0};
    double ra[8], rb[8];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int e = tid + 256 * p;
            int m, k;
            if (transA) { m = e & 127; k = e >> 7; }          // A is K x M: contiguous along m
            else { k = e & 15; m = e >> 4; }                  // A is M x K: contiguous along k
            const int gm = m0 + m, gk = k0 + k;
            ra[p] = (gm < M && gk < K) ? (transA ? A[(size_t)gk * lda + gm] : A[(size_t)gm * lda + gk]) : 0.0;
            const int nn = e & 127, kb = e >> 7;              // B is K x N: contiguous along n
            const int gn = n0 + nn, gkb = k0 + kb;
            rb[p] = (gn < N && gkb < K) ? B[(size_t)gkb * ldb + gn] : 0.0;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int e = tid + 256 * p;
            if (transA) As[buf][e >> 7][e & 127] = ra[p];
            else As[buf][e & 15][e >> 4] = ra[p];
            Bs[buf][e >> 7][e & 127] = rb[p];
        }
    };
    fetch(0);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += G2_BK) {
        const bool more = k0 + G2_BK < K;
        if (more) fetch(k0 + G2_BK);
#pragma unroll
        for (int kk = 0; kk < G2_BK; kk += 4) {
            double af[4], bf[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) af[a] = As[buf][kk + lk][wm + a * 16 + li];
#pragma unroll
            for (int b = 0; b < 4; ++b) bf[b] = Bs[buf][kk + lk][wn + b * 16 + li];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = m0 + wm + a * 16 + lk + 4 * r;
                const int gn = n0 + wn + b * 16 + li;
                if (gm < M && gn < N) {
                    double* cp = C + (size_t)gm * ldc + gn;
                    const double v = alpha * acc[a][b][r];
                    *cp = (beta == 0.0) ? v : (v + beta * (*cp));
                }
            }
}

__global__ __launch_bounds__(256) void gemm128_kernel(int transA, int M, int N, int K, double alpha,
                                                      const double* __restrict__ A, int lda,
                                                      const double* __restrict__ B, int ldb, double beta,
                                                      double* __restrict__ C, int ldc) {
    __shared__ double As[2][G2_BK][G2_LD];
    __shared__ double Bs[2][G2_BK][G2_LD];
    gemm128_tile(transA, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, blockIdx.y * G2_BM, blockIdx.x * G2_BN, As, Bs);
}

__global__ __launch_bounds__(256) void gemm128_merge_batched_kernel(const int* __restrict__ desc,
                                                                    const double* __restrict__ A,
                                                                    const double* __restrict__ B,
                                                                    double* __restrict__ C, int ld) {
    __shared__ double As[2][G2_BK][G2_LD];
    __shared__ double Bs[2][G2_BK][G2_LD];
    const int lo = desc[4 * blockIdx.z], N = desc[4 * blockIdx.z + 1], K = desc[4 * blockIdx.z + 2];
    const int m0 = blockIdx.y * G2_BM, n0 = blockIdx.x * G2_BN;
    if (m0 >= K || n0 >= N) return;
    const size_t off = (size_t)lo * ld + lo;
    gemm128_tile(0, K, N, K, 1.0, A + off, ld, B + off, ld, 0.0, C + off, ld, m0, n0, As, Bs);
}

__global__ __launch_bounds__(256) void gemm_valu_kernel(int transA, int transB, int M, int N, int K,
                                                        double alpha, const double* __restrict__ A,
                                                        int lda, const double* __restrict__ B,
                                                        int ldb, double beta, double* __restrict__ C,
                                                        int ldc) {
    __shared__ double As[GM_BK][GM_LD];
    __shared__ double Bs[GM_BK][GM_LD];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;     // 16 x 16 threads, 4x4 each
    const int m0 = blockIdx.y * GM_BM, n0 = blockIdx.x * GM_BN;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
    for (int k0 = 0; k0 < K; k0 += GM_BK) {
        __syncthreads();
        gemm_load_tiles(A, lda, transA, B, ldb, transB, M, N, K, m0, n0, k0, As, Bs);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GM_BK; ++kk) {
            double af[4], bf[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) af[a] = As[kk][ty * 4 + a];
#pragma unroll
            for (int b = 0; b < 4; ++b) bf[b] = Bs[kk][tx * 4 + b];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] += af[a] * bf[b];
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int gm = m0 + ty * 4 + a, gn = n0 + tx * 4 + b;
            if (gm < M && gn < N) {
                double* cp = C + (size_t)gm * ldc + gn;
                const double v = alpha * acc[a][b];
                *cp = (beta == 0.0) ? v : (v + beta * (*cp));
            }
        }
}

int launch_gemm(sella_ctx* c, int transA, int transB, int M, int N, int K, double alpha,
                const double* A, int lda, const double* B, int ldb, double beta, double* C, int ldc) {
    if (M <= 0 || N <= 0) return SELLA_OK;
    dim3 grid((N + GM_BN - 1) / GM_BN, (M + GM_BM - 1) / GM_BM);
    prof_begin(c, PROF_GEMM, 8.0 * ((double)M * K + (double)K * N + 2.0 * M * N), 2.0 * M * (double)N * K);
    if (c->opt.gemm_mfma && !transB && M >= 192 && N >= 192 && c->opt.gemm_tile128) {
        dim3 g2((N + G2_BN - 1) / G2_BN, (M + G2_BM - 1) / G2_BM);
        SELLA_LAUNCH(c, gemm128_kernel, g2, dim3(256), 0, transA, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
    } else if (c->opt.gemm_mfma)
        SELLA_LAUNCHB_PROF(c, gemm_mfma_kernel, gemm_mfma_vb, 256, grid, dim3(256), 0, transA, transB, M, N, K, alpha,
                     A, lda, B, ldb, beta, C, ldc);
    else
        SELLA_LAUNCH(c, gemm_valu_kernel, grid, dim3(256), 0, transA, transB, M, N, K, alpha,
                     A, lda, B, ldb, beta, C, ldc);
    prof_end(c);
    HIPCHK(hipGetLastError());
    return SELLA_OK;
}

int launch_gemm_merge_batched(sella_ctx* c, int nbatch, const int* desc, int maxN, int maxK, const double* A,
                              const double* B, double* C, int ld) {
    if (nbatch <= 0 || maxN <= 0 || maxK <= 0) return SELLA_OK;
    if (maxK >= 192 && c->opt.gemm_tile128) {
        dim3 g2((maxN + G2_BN - 1) / G2_BN, (maxK + G2_BM - 1) / G2_BM, nbatch);
        hipLaunchKernelGGL(gemm128_merge_batched_kernel, g2, dim3(256), 0, c->stream, desc, A, B, C, ld);
    } else {
        dim3 grid((maxN + GM_BN - 1) / GM_BN, (maxK + GM_BM - 1) / GM_BM, nbatch);
        hipLaunchKernelGGL(gemm_merge_batched_kernel, grid, dim3(256), 0, c->stream, desc, A, B, C, ld);
    }
    HIPCHK(hipGetLastError());
    return SELLA_OK;
}

// ------------------------------------------------------------------------------------
// Panel product on the matrix cores: Y[h][r] = sum_c A[r][c] X[h][c] for a 16-vector panel X
// (vector-major, rows beyond the live ones zero).  HBM bound like the row-panel matvec, but the matrix
// is streamed ONCE for all 16 right-hand sides (block H.V, BASELINE configs[4]).  A workgroup owns 32
// rows; its four wavefronts split the columns by 64-column chunk and meet in LDS.  Operands come
// straight from global memory as 32-byte vectors: lane (i, g) takes columns 4g..4g+3 of a 16-column
// group for four successive v_mfma_f64_16x16x4_f64 (the summation index may be permuted freely).
// ------------------------------------------------------------------------------------
// 16 RT rows per workgroup (RT = 1 .. 4, chosen by the launcher so that there are about 256 workgroups)
template <int RT>
__device__ __forceinline__ void panel16_mfma_vb(const VB vb, const double* __restrict__ A, int rows, int ld,
                                                           const double* __restrict__ Xp, int nrhs,
                                                           double* __restrict__ Y, int ldy) {
    constexpr int WR = 16 * RT;
    __shared__ double part[4][WR][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int r0 = vb.x * WR;
    const double* arow[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int r = r0 + 16 * t + li;
        arow[t] = A + (size_t)(r < rows ? r : rows - 1) * ld;
    }
    const double* xr = Xp + (size_t)li * ld;
    f64x4 acc[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
    for (int cc = 64 * wave; cc < ld; cc += 256) {
        double4 va[RT][4], vx[4];
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) {
            const int col = cc + 16 * sg + 4 * lg;
            const bool ok = col < ld;
            const int colc = ok ? col : 0;
#pragma unroll
            for (int t = 0; t < RT; ++t) va[t][sg] = *reinterpret_cast<const double4*>(arow[t] + colc);
            vx[sg] = *reinterpret_cast<const double4*>(xr + colc);
            if (!ok) vx[sg] = make_double4(0.0, 0.0, 0.0, 0.0);
        }
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) {
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[t][sg].x, vx[sg].x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[t][sg].y, vx[sg].y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[t][sg].z, vx[sg].z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[t][sg].w, vx[sg].w, acc[t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][16 * t + lg + 4 * r][li] = acc[t][r];
    __syncthreads();
    for (int e = tid; e < WR * 16; e += 256) {
        const int h = e / WR, row = e % WR;               // consecutive threads -> consecutive rows of Y[h]
        if (h < nrhs && r0 + row < rows)
            Y[(size_t)h * ldy + r0 + row] = part[0][row][h] + part[1][row][h] + part[2][row][h] + part[3][row][h];
    }
}
template <int RT>
__global__ __launch_bounds__(256) void panel16_mfma_kernel(const double* __restrict__ A, int rows, int ld,
                                                           const double* __restrict__ Xp, int nrhs,
                                                           double* __restrict__ Y, int ldy) { panel16_mfma_vb<RT>(vb_hw(), A, rows, ld, Xp, nrhs, Y, ldy); }

// ---- short-and-wide panel products -----------------------------------------------------------------------------
// Gram blocks and projections of the block methods: rows <= 64 (the basis), nrhs <= 16 (the block), cols = n long.
// The row-parallel kernels above give such a product 1 - 4 workgroups, each streaming whole rows: 82 us for a 48 x 12288
// panel (57 GB/s) — five of them per block-Davidson iteration were its largest serial part.  Here the LONG index is
// split: workgroup w takes columns [w cpw, (w + 1) cpw) in tiles of 64 staged through LDS, thread (r, hg) accumulates the
// four outputs (r, 4 hg .. 4 hg + 3); the per-workgroup partial results are summed by a second, single-workgroup
// kernel in workgroup order (deterministic, no atomics).
constexpr int PS_ROWS = 64, PS_TILE = 64;

__global__ __launch_bounds__(256) void panel_small_partial_kernel(const double* __restrict__ A, int rows, int cols, int lda,
                                                                  const double* __restrict__ X, int ldx, int nrhs, int cpw,
                                                                  double* __restrict__ part) {
    __shared__ double As[PS_ROWS][PS_TILE + 1];
    __shared__ double Xs[16][PS_TILE];
    const int tid = threadIdx.x, r = tid & 63, hg = tid >> 6;
    const int c0 = blockIdx.x * cpw, c1 = (c0 + cpw < cols) ? c0 + cpw : cols;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int t0 = c0; t0 < c1; t0 += PS_TILE) {
        __syncthreads();
        // stage: 64 lanes along the columns (coalesced), 4 rows per pass
        for (int rr = tid >> 6; rr < PS_ROWS; rr += 4) {
            const int j = t0 + (tid & 63);
            As[rr][tid & 63] = (rr < rows && j < c1) ? A[(size_t)rr * lda + j] : 0.0;
        }
        for (int hh = tid >> 6; hh < 16; hh += 4) {
            const int j = t0 + (tid & 63);
            Xs[hh][tid & 63] = (hh < nrhs && j < c1) ? X[(size_t)hh * ldx + j] : 0.0;
        }
        __syncthreads();
#pragma unroll 8
        for (int j = 0; j < PS_TILE; ++j) {
            const double a = As[r][j];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += a * Xs[4 * hg + q][j];
        }
    }
    double* out = part + (size_t)blockIdx.x * 16 * PS_ROWS;
#pragma unroll
    for (int q = 0; q < 4; ++q) out[(4 * hg + q) * PS_ROWS + r] = acc[q];
}

// one workgroup per right-hand side h: thread (r, g) sums every fourth partial of output (h, r) with four independent
// accumulators (the loads of a sequential sum were this kernel's whole time: 174 us for 192 partials), the four groups
// are combined through LDS in a fixed order
// (pword != null: the LAST workgroup to finish publishes the sequence word of a polled wait — context.hip poll_arm /
// poll_wait — so that a result the host consumes needs no mark kernel behind it)
__global__ __launch_bounds__(256) void panel_small_reduce_kernel(const double* __restrict__ part, int nsplit, int rows,
                                                                 int nrhs, double* __restrict__ Y, int ldy,
                                                                 unsigned long long* pword, unsigned long long pseq, unsigned* pcount) {
    __shared__ double red[4][PS_ROWS];
    const int tid = threadIdx.x, r = tid & 63, g = tid >> 6, h = blockIdx.x;
    const double* p = part + (size_t)h * PS_ROWS + r;
    const size_t stride = (size_t)16 * PS_ROWS;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int w = g;
    for (; w + 12 < nsplit; w += 16) {
        s0 += p[(size_t)w * stride];
        s1 += p[(size_t)(w + 4) * stride];
        s2 += p[(size_t)(w + 8) * stride];
        s3 += p[(size_t)(w + 12) * stride];
    }
    for (; w < nsplit; w += 4) s0 += p[(size_t)w * stride];
    red[g][r] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && r < rows) Y[(size_t)h * ldy + r] = (red[0][r] + red[1][r]) + (red[2][r] + red[3][r]);
    if (pword != nullptr) {
        __syncthreads();
        if (tid == 0) {
            __threadfence_system();
            if (atomicAdd(pcount, 1u) == gridDim.x - 1) {
                *pcount = 0u;
                __threadfence_system();
                *reinterpret_cast<volatile unsigned long long*>(pword) = pseq;
            }
        }
    }
}

static int launch_panel_small(sella_ctx* c, const double* A, int rows, int cols, int lda, const double* X, int ldx, int nrhs,
                              double* Y, int ldy, bool publish = false) {
    int nsplit = (cols + PS_TILE - 1) / PS_TILE;
    if (nsplit > 256) nsplit = 256;
    int cpw = (cols + nsplit - 1) / nsplit;
    cpw = (cpw + PS_TILE - 1) / PS_TILE * PS_TILE;
    nsplit = (cols + cpw - 1) / cpw;
    double* part;
    SCHK(scratch_get(c, SCR_PSMALL, (size_t)256 * 16 * PS_ROWS * sizeof(double), &part));
    hipLaunchKernelGGL(panel_small_partial_kernel, dim3(nsplit), dim3(256), 0, c->stream, A, rows, cols, lda, X, ldx, nrhs, cpw, part);
    unsigned long long* pword = nullptr;
    unsigned long long pseq = 0;
    unsigned* pcount = nullptr;
    if (publish) SCHK(poll_arm(c, &pword, &pseq, &pcount));
    hipLaunchKernelGGL(panel_small_reduce_kernel, dim3(nrhs), dim3(256), 0, c->stream, part, nsplit, rows, nrhs, Y, ldy, pword, pseq, pcount);
    HIPCHK(hipGetLastError());
    return SELLA_OK;
}

// launch_panel16 for a result the host waits for by polling (Y in pinned host memory): the product's last kernel
// publishes the sequence word where it can, a mark kernel follows otherwise; poll_wait(c) afterwards.
int launch_panel16_marked(sella_ctx* c, const double* A, int rows, int cols, int lda, const double* Xp, int nrhs, double* Y, int ldy) {
    if (rows <= 0 || nrhs <= 0) return poll_mark(c);
    if (rows <= PS_ROWS && nrhs <= 16 && c->opt.panel_small > 0 && cols >= c->opt.panel_small && !(c->cohort && cohort_in_fiber()))
        return launch_panel_small(c, A, rows, cols, lda, Xp, lda, nrhs, Y, ldy, true);
    SCHK(launch_panel16(c, A, rows, cols, lda, Xp, nrhs, Y, ldy));
    return poll_mark(c);
}

// Y (nrhs rows, vector-major) = A X^T for a zero-padded 16-row panel Xp with the matrix's leading dimension
int launch_panel16(sella_ctx* c, const double* A, int rows, int cols, int lda, const double* Xp, int nrhs,
                   double* Y, int ldy) {
    if (rows <= 0 || nrhs <= 0) return SELLA_OK;
    if (rows <= PS_ROWS && nrhs <= 16 && c->opt.panel_small > 0 && cols >= c->opt.panel_small)
        return launch_panel_small(c, A, rows, cols, lda, Xp, lda, nrhs, Y, ldy);
    if (nrhs > 16 || (lda & 3) || (((uintptr_t)A) & 31) || (((uintptr_t)Xp) & 31)) {
        set_error("panel16: needs <= 16 right-hand sides and 32-byte aligned rows");
        return SELLA_E_INVALID;
    }
    prof_begin(c, PROF_GEMV, 8.0 * rows * (double)cols, 2.0 * rows * (double)cols * nrhs);
    // Rows per workgroup by size: the kernel is fastest with ONE workgroup per CU (256 of them) — n = 12288: 48 rows
    // 210 us = 5.76 TB/s against 281 (16 rows, 768 workgroups) and 286 (32 rows, 384); n = 8192: 32 rows 106 us against 130
    // (16) and 128 (48); n = 3072: 16 rows (192 workgroups) 26 us against 36 (32): tools/panel_bench.py.  A variant that
    // staged the matrix chunk through LDS to fetch whole rows per wavefront was built and measured slower (383 us).
    long rt = c->opt.panel_rows;
    if (rt == 0) {
        const int per = (rows + 4095) / 4096;
        rt = 16 * (per < 1 ? 1 : (per > 4 ? 4 : per));
    }
    if (rt == 48)
        SELLA_LAUNCHB_PROF(c, HIP_KERNEL_NAME(panel16_mfma_kernel<3>), SELLA_BODY(panel16_mfma_vb<3>), 256, dim3((rows + 47) / 48), dim3(256), 0, A, rows, lda, Xp, nrhs, Y, ldy);
    else if (rt == 64)
        SELLA_LAUNCHB_PROF(c, HIP_KERNEL_NAME(panel16_mfma_kernel<4>), SELLA_BODY(panel16_mfma_vb<4>), 256, dim3((rows + 63) / 64), dim3(256), 0, A, rows, lda, Xp, nrhs, Y, ldy);
    else if (rt == 32)
        SELLA_LAUNCHB_PROF(c, HIP_KERNEL_NAME(panel16_mfma_kernel<2>), SELLA_BODY(panel16_mfma_vb<2>), 256, dim3((rows + 31) / 32), dim3(256), 0, A, rows, lda, Xp, nrhs, Y, ldy);
    else
        SELLA_LAUNCHB_PROF(c, HIP_KERNEL_NAME(panel16_mfma_kernel<1>), SELLA_BODY(panel16_mfma_vb<1>), 256, dim3((rows + 15) / 16), dim3(256), 0, A, rows, lda, Xp, nrhs, Y, ldy);
    prof_end(c);
    HIPCHK(hipGetLastError());
    return SELLA_OK;
}

}  // namespace sella
