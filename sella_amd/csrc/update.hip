// update.hip — secant symmetrisation and multi-secant quasi-Newton updates of the resident
// approximate Hessian (sella/hessian_update.py:12-157; torch formulation :160-203).
//
// Every update of the reference family can be written, after the final symmetrisation
// B+ = (B + Delta + (B + Delta)^T)/2 (hessian_update.py:104-109), as
//       B+ = sym(B) + sum_a (U_a Z_a^T + Z_a U_a^T)
// with two vector-major panels U, Z of kk <= 2k rows (k = number of secant pairs):
//   TS-BFGS / PSB / DFP / Greenstadt : Delta = U J^T + J U^T - U (J^T S) U^T
//                                      ->  Z = J - 1/2 sym(J^T S) U
//   SR1                              : Delta = J (J^T S)^-1 J^T      -> U = J, Z = 1/2 sym(C^-1) J
//   BFGS                             : two such terms (Y and B S panels).
// The n x n work is therefore: k-column panel products B S, Q^T S, Q (|lam| Q^T S) (row-panel
// matvecs) and ONE fused pass over B that symmetrises and applies the rank-2kk update
// (reads and writes B once: 16 n^2 bytes).  All k x k algebra is host code (host_math.h).
#include "internal.h"
#include <chrono>
#include <cstdlib>
#include <cstdint>
#include "host_math.h"

namespace sella {

// ------------------------------------------------------------------------------------------
// fused symmetrise + symmetric rank-2kk update.  Workgroup (bx >= by) owns the 32x32 tile
// T1 = B[by*32.., bx*32..] and its mirror T2; both are read once and written once.
// ------------------------------------------------------------------------------------------
constexpr int R2K_KT = 16;

__device__ __forceinline__ void sym_rank2k_vb(const VB vb, double* __restrict__ B, int n, int ld,
                                                         const double* __restrict__ Up,
                                                         const double* __restrict__ Zp, int ldp, int kk,
                                                         double alpha) {
    if (vb.x < vb.y) return;
    __shared__ double t1[32][33];
    __shared__ double t2[32][33];
    __shared__ double ur[R2K_KT][32], zr[R2K_KT][32], uc[R2K_KT][32], zc[R2K_KT][32];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int r0 = vb.y * 32, c0 = vb.x * 32;
    for (int k = ty; k < 32; k += 8) {
        int r = r0 + k, cc = c0 + tx;
        t1[k][tx] = (r < n && cc < n) ? B[(size_t)r * ld + cc] : 0.0;
        r = c0 + k; cc = r0 + tx;
        t2[k][tx] = (r < n && cc < n) ? B[(size_t)r * ld + cc] : 0.0;
    }
    double acc[4] = {0.0, 0.0, 0.0, 0.0};     // rows ty, ty+8, ty+16, ty+24 ; column tx
    for (int a0 = 0; a0 < kk; a0 += R2K_KT) {
        const int at = (kk - a0 < R2K_KT) ? (kk - a0) : R2K_KT;
        __syncthreads();
        for (int t = threadIdx.x; t < R2K_KT * 32; t += 256) {
            const int a = t >> 5, i = t & 31;
            const bool ok = a < at;
            const int gr = r0 + i, gc = c0 + i;
            ur[a][i] = (ok && gr < n) ? Up[(size_t)(a0 + a) * ldp + gr] : 0.0;
            zr[a][i] = (ok && gr < n) ? Zp[(size_t)(a0 + a) * ldp + gr] : 0.0;
            uc[a][i] = (ok && gc < n) ? Up[(size_t)(a0 + a) * ldp + gc] : 0.0;
            zc[a][i] = (ok && gc < n) ? Zp[(size_t)(a0 + a) * ldp + gc] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < R2K_KT; ++a) {
            const double ucx = uc[a][tx], zcx = zc[a][tx];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += ur[a][ty + 8 * q] * zcx + zr[a][ty + 8 * q] * ucx;
        }
    }
    __syncthreads();
    // new value of element (r0 + k, c0 + tx), k = ty + 8q; mirror element gets the same value
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = ty + 8 * q;
        const double v = 0.5 * (t1[k][tx] + t2[tx][k]) + alpha * acc[q];
        t1[k][tx] = v;
    }
    __syncthreads();
    const bool diag = (vb.x == vb.y);
    for (int k = ty; k < 32; k += 8) {
        int r = r0 + k, cc = c0 + tx;
        // inside a diagonal tile both (k, tx) and (tx, k) are computed (in different summation
        // orders): publish the upper-triangle value for both so B+ is exactly symmetric
        const double v = (diag && k > tx) ? t1[tx][k] : t1[k][tx];
        if (r < n && cc < n) B[(size_t)r * ld + cc] = v;
        if (!diag) {
            r = c0 + k; cc = r0 + tx;
            if (r < n && cc < n) B[(size_t)r * ld + cc] = t1[tx][k];
        }
    }
}
__global__ __launch_bounds__(256) void sym_rank2k_kernel(double* __restrict__ B, int n, int ld,
                                                         const double* __restrict__ Up,
                                                         const double* __restrict__ Zp, int ldp, int kk,
                                                         double alpha) { sym_rank2k_vb(vb_hw(), B, n, ld, Up, Zp, ldp, kk, alpha); }

// ------------------------------------------------------------------------------------------
// Streaming rank-2kk update of an ALREADY SYMMETRIC block (the trailing update of the tridiagonalisation,
// eigh.hip):  C <- C + alpha (U^T Z + Z^T U)  with U, Z vector-major (kk rows).  No mirror tile is read: every
// element is updated in place from its own row and column of the panels, so the pass is a pure stream — 16-byte
// loads and stores along rows (a workgroup owns a 32 x 128 tile: 1 KiB contiguous per row), the 2 kk-deep products
// on the matrix cores (v_mfma_f64_16x16x4: X = [U; Z]^T, Y = [Z; U]^T, K = 2 kk), the MFMA fragments passed
// through LDS once to reach the row-major order of the global accesses.  The two triangles receive the same
// products in a different summation order, i.e. equal to roundoff, not bitwise: fine for a block whose rows are
// the only thing read afterwards; the quasi-Newton path keeps the exactly symmetric kernel above.
// Traffic: 16 m^2 bytes (read + write once), 4 kk m^2 flop.
// ------------------------------------------------------------------------------------------
typedef double upd_f64x4 __attribute__((ext_vector_type(4)));
constexpr int RS_TR = 32, RS_TC = 128;

__global__ __launch_bounds__(256) void rank2k_stream_kernel(double* __restrict__ C, int m, int ld,
                                                            const double* __restrict__ Up,
                                                            const double* __restrict__ Zp, int ldp, int kk,
                                                            double alpha) {
    __shared__ double dl[RS_TR][RS_TC + 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int r0 = blockIdx.y * RS_TR, c0 = blockIdx.x * RS_TC;
    const int wr = r0 + 16 * (wave & 1);                 // this wave's 16 rows
    const int wc = c0 + 64 * (wave >> 1);                // ... and 64 columns (4 MFMA tiles)
    upd_f64x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = upd_f64x4{0.0, 0.0, 0.0, 0.0};
    const int rr = (wr + li < m) ? wr + li : m - 1;
    const int K = 2 * kk;
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int k = k0 + lq;                           // summation index of this lane
        const bool ok = k < K;
        const int kc = ok ? k : 0;
        // X[r][k] = (k < kk ? U[k][r] : Z[k - kk][r]);  Y[c][k] = (k < kk ? Z[k][c] : U[k - kk][c])
        const double* xrow = (kc < kk) ? Up + (size_t)kc * ldp : Zp + (size_t)(kc - kk) * ldp;
        const double* yrow = (kc < kk) ? Zp + (size_t)kc * ldp : Up + (size_t)(kc - kk) * ldp;
        const double a = ok ? xrow[rr] : 0.0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int cc = wc + 16 * t + li;
            const double b = (ok && cc < m) ? yrow[cc] : 0.0;
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
    }
    // fragments -> LDS (C/D layout of the f64 MFMA: row = (lane >> 4) + 4 reg, col = lane & 15)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) dl[16 * (wave & 1) + lq + 4 * q][64 * (wave >> 1) + 16 * t + li] = acc[t][q];
    __syncthreads();
    // row-major pass: thread -> (row, 16-byte piece): 64 pieces per row, 4 rows per sweep
    const int pc = tid & 63, pr = tid >> 6;
#pragma unroll
    for (int s8 = 0; s8 < RS_TR; s8 += 4) {
        const int r = r0 + s8 + pr, cidx = c0 + 2 * pc;
        if (r < m && cidx < m) {
            double* p = C + (size_t)r * ld + cidx;
            const double d0 = dl[s8 + pr][2 * pc], d1 = dl[s8 + pr][2 * pc + 1];
            if (cidx + 1 < m) {
                double2 v = *reinterpret_cast<double2*>(p);
                v.x += alpha * d0;
                v.y += alpha * d1;
                *reinterpret_cast<double2*>(p) = v;
            } else {
                p[0] += alpha * d0;
            }
        }
    }
}

// The same update for a compile-time panel depth (KK columns: the eigensolver's block size): every load of the workgroup is
// issued before the first MFMA — the C pieces each thread will update (they do not depend on the panels) and all
// 2 KK / 4 operand fragments — instead of 2 KK / 4 dependent rounds of "five loads from L2, four MFMAs".  PMC of round 3
// had shown the generic kernel at 2.2–2.8 TB/s where a plain in-place stream runs at 6.5: latency of the operand loads,
// not bandwidth (profiles/r03_pmc.md, r03_rmw_lab.log).  Round 5: the loads are straight-line code (clamped addresses,
// masked afterwards) with the operands in front — 21.6 -> 17.3 us per launch at 3N = 3072, the largest block (151 MB)
// 38.2 -> 29.9 us = 5.0 TB/s, bit-identical results.  Two other layouts were measured and dropped (sessions r05k, r05l,
// r05u): 64 x 128 tiles with the column operand staged in LDS (0.375 operand bytes per matrix byte instead of 1.25:
// SLOWER, 25.1 us — two workgroups per CU), and pairs of 64 x 64 tiles sharing one MFMA product, the lower tile taking
// it transposed (half the MFMA work: 16.2-16.8 us, but the two triangles then differ from this kernel's by rounding,
// which moves the Davidson trajectories for 0.05 ms per eigh).
template <int KK>
__global__ __launch_bounds__(256) void rank2k_stream_fixed_kernel(double* __restrict__ C, int m, int ld,
                                                                  const double* __restrict__ Up,
                                                                  const double* __restrict__ Zp, int ldp, double alpha,
                                                                  int upper_only) {
    __shared__ double dl[RS_TR][RS_TC + 2];
    constexpr int KS = 2 * KK / 4;
    // upper_only: the caller reads the upper triangle only (symmetric-aware matvec of the tridiagonalisation): tiles
    // entirely below the diagonal leave at once
    if (upper_only && (int)(blockIdx.x * RS_TC + RS_TC - 1) < (int)(blockIdx.y * RS_TR)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int r0 = blockIdx.y * RS_TR, c0 = blockIdx.x * RS_TC;
    const int wr = r0 + 16 * (wave & 1);
    const int wc = c0 + 64 * (wave >> 1);
    // operands first, then this thread's pieces of the C tile (row-major pass below): the counter of outstanding loads
    // retires in order, so the MFMA phase starts on the operands while the pieces of C (needed only at the end) are
    // still on their way.  Every load is unconditional, from a clamped address (ld is even: the pair behind an even
    // column of any row exists) — a load behind a condition costs a full wait for everything issued before it.
    const int rr = (wr + li < m) ? wr + li : m - 1;
    double av[KS], bv[KS][4];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int k = 4 * s + lq;                                        // < 2 KK always
        const double* xrow = (k < KK) ? Up + (size_t)k * ldp : Zp + (size_t)(k - KK) * ldp;
        const double* yrow = (k < KK) ? Zp + (size_t)k * ldp : Up + (size_t)(k - KK) * ldp;
        av[s] = xrow[rr];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int cc = wc + 16 * t + li;
            bv[s][t] = yrow[cc < m ? cc : m - 1];
        }
    }
    const int pc = tid & 63, pr = tid >> 6;
    const int cidx = c0 + 2 * pc;
    double2 cv[RS_TR / 4];
    {
        const int cl = (cidx < m) ? cidx : 0;
#pragma unroll
        for (int q = 0; q < RS_TR / 4; ++q) {
            const int r = (r0 + 4 * q + pr < m) ? r0 + 4 * q + pr : m - 1;
            cv[q] = *reinterpret_cast<const double2*>(C + (size_t)r * ld + cl);
        }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) bv[s][t] = (wc + 16 * t + li < m) ? bv[s][t] : 0.0;
    upd_f64x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = upd_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[s][t], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) dl[16 * (wave & 1) + lq + 4 * q][64 * (wave >> 1) + 16 * t + li] = acc[t][q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RS_TR / 4; ++q) {
        const int r = r0 + 4 * q + pr;
        if (r < m && cidx < m) {
            double* p = C + (size_t)r * ld + cidx;
            const double d0 = dl[4 * q + pr][2 * pc], d1 = dl[4 * q + pr][2 * pc + 1];
            if (cidx + 1 < m) {
                double2 v = cv[q];
                v.x += alpha * d0;
                v.y += alpha * d1;
                *reinterpret_cast<double2*>(p) = v;
            } else {
                p[0] = cv[q].x + alpha * d0;
            }
        }
    }
}

// lower triangle of the m x m block C <- transpose of its upper triangle (32 x 32 tiles through LDS)
__device__ __forceinline__ void mirror_upper_vb(const VB vb, double* __restrict__ C, int m, int ld) {
    __shared__ double t[32][33];
    const int bi = vb.y, bj = vb.x;              // source tile (rows bi, columns bj), bj >= bi
    if (bj < bi) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int q = 0; q < 4; ++q) {
        const int r = bi * 32 + ty + 8 * q, cc = bj * 32 + tx;
        t[ty + 8 * q][tx] = (r < m && cc < m) ? C[(size_t)r * ld + cc] : 0.0;
    }
    __syncthreads();
    for (int q = 0; q < 4; ++q) {
        const int r = bj * 32 + ty + 8 * q, cc = bi * 32 + tx;   // destination (r, cc) = source (cc, r)
        if (r < m && cc < m && r > cc) C[(size_t)r * ld + cc] = t[tx][ty + 8 * q];
    }
}
__global__ __launch_bounds__(256) void mirror_upper_kernel(double* __restrict__ C, int m, int ld) { mirror_upper_vb(vb_hw(), C, m, ld); }

int launch_mirror_upper(sella_ctx* c, double* C, int m, int ld) {
    if (m <= 1) return SELLA_OK;
    const int nt = (m + 31) / 32;
    prof_begin(c, PROF_OTHER, 8.0 * m * (double)m, 0.0);
    SELLA_LAUNCHB_PROF(c, mirror_upper_kernel, mirror_upper_vb, 256, dim3(nt, nt), dim3(256), 0, C, m, ld);
    prof_end(c);
    HIPCHK(hipGetLastError());
    return SELLA_OK;
}

int launch_rank2k_stream(sella_ctx* c, double* C, int m, int ld, const double* Up, const double* Zp, int ldp, int kk,
                         double alpha, bool upper_only) {
    if (m <= 0 || kk <= 0) return SELLA_OK;
    if ((ld & 1) || (reinterpret_cast<uintptr_t>(C) & 15)) return launch_sym_rank2k(c, C, m, ld, Up, Zp, ldp, kk, alpha);
    const double part = upper_only ? 0.5 : 1.0;
    prof_begin(c, PROF_UPDATE, part * 16.0 * m * (double)m, part * 4.0 * kk * (double)m * m);
    const dim3 grid((m + RS_TC - 1) / RS_TC, (m + RS_TR - 1) / RS_TR);
    const int uo = upper_only ? 1 : 0;
    if (kk == 16 && c->opt.rank2k_fixed)
        SELLA_LAUNCH(c, rank2k_stream_fixed_kernel<16>, grid, dim3(256), 0, C, m, ld, Up, Zp, ldp, alpha, uo);
    else if (kk == 32 && c->opt.rank2k_fixed)
        SELLA_LAUNCH(c, rank2k_stream_fixed_kernel<32>, grid, dim3(256), 0, C, m, ld, Up, Zp, ldp, alpha, uo);
    else
    SELLA_LAUNCH(c, rank2k_stream_kernel, grid, dim3(256), 0, C, m, ld, Up,
                 Zp, ldp, kk, alpha);
    prof_end(c);
    HIPCHK(hipGetLastError());
    return SELLA_OK;
}

int launch_sym_rank2k(sella_ctx* c, double* B, int n, int ld, const double* Up, const double* Zp, int ldp,
                      int kk, double alpha) {
    prof_begin(c, PROF_UPDATE, 16.0 * n * (double)n, 4.0 * kk * (double)n * n);
    const int nb = (n + 31) / 32;
    SELLA_LAUNCHB_PROF(c, sym_rank2k_kernel, sym_rank2k_vb, 256, dim3(nb, nb), dim3(256), 0, B, n, ld, Up, Zp, ldp, kk, alpha);
    prof_end(c);
    HIPCHK(hipGetLastError());
    return SELLA_OK;
}

namespace {

using hostm::vec;

// G[a][b] = P_a . Q_b for two vector-major panels (ka x kb), one sync
int gram(sella_ctx* c, const double* P, int ka, const double* Q, int kb, int n, int ld, vec& G) {
    G.assign((size_t)ka * kb, 0.0);
    if (ka * kb > DS_STAGE - DS_CVEC) { set_error("update: too many secant pairs"); return SELLA_E_UNSUPPORTED; }
    double* d = c->dscal + DS_CVEC;
    // Y[h*ldy + i] = P_i . Q_h
    SCHK(launch_gemv_rows(c, P, ka, n, ld, Q, ld, kb, d, ka, GemvEpi()));
    SCHK(read_scalars(c, DS_CVEC, ka * kb));
    for (int a = 0; a < ka; ++a)
        for (int b = 0; b < kb; ++b) G[(size_t)a * kb + b] = c->hscal[DS_CVEC + (size_t)b * ka + a];
    return SELLA_OK;
}

// device copy of a small host matrix in the SCR_W scratch (stream-ordered, no wait; k x k only)
int put_k(sella_ctx* c, const vec& h, size_t offset, double** d) {
    double* base;
    SCHK(scratch_get(c, SCR_W, (size_t)(1 << 16) * sizeof(double), &base));
    if (offset + h.size() > (size_t)(1 << 16)) { set_error("update: coefficient buffer overflow"); return SELLA_E_UNSUPPORTED; }
    SCHK(h2d_async(c, base + offset, h.data(), h.size() * sizeof(double)));
    *d = base + offset;
    return SELLA_OK;
}

// Ytilde panel from S, Y panels (symmetrize_Y, hessian_update.py:27-37)
int symmetrize_panels(sella_ctx* c, const double* Sp, const double* Yp, int n, int k, int ld, int symm,
                      double* Ytp) {
    SCHK(launch_axpby2d(c, k, n, 1.0, Yp, ld, 0.0, nullptr, 0, Ytp, ld));
    if (symm < 0 || k == 1) return SELLA_OK;
    if (symm > 2) { set_error("Unknown symmetrization method %d", symm); return SELLA_E_INVALID; }
    vec STS, STY, X((size_t)k * k);
    SCHK(gram(c, Sp, k, Sp, k, n, ld, STS));
    SCHK(gram(c, Sp, k, Yp, k, n, ld, STY));
    const int which = hostm::symm_coeffs(k, STS.data(), STY.data(), symm, X.data());
    if (which < 0) return SELLA_OK;
    double* dX;
    SCHK(put_k(c, X, 0, &dX));
    return launch_lincomb(c, n, k, which == 0 ? Sp : Yp, ld, k, dX, k, nullptr, 0, 0, nullptr, 0, 1.0, Ytp, ld);
}

// inverse of a general k x k matrix (LU); returns false if singular
bool invert(int k, const vec& M, vec& Minv) {
    vec LU(M);
    std::vector<int> piv(k);
    if (small::lu_factor(k, LU.data(), k, piv.data()) != 0) return false;
    Minv.assign((size_t)k * k, 0.0);
    for (int i = 0; i < k; ++i) Minv[(size_t)i * k + i] = 1.0;
    small::lu_solve(k, LU.data(), k, piv.data(), Minv.data(), k, k);
    return true;
}

// symmetric pseudo-inverse (minimum norm, like lstsq) of a symmetric k x k matrix
void sym_pinv(int k, const vec& M, vec& Mp) {
    // column by column hostm::sym_pinv_solve(M, e_col) — with the eigendecomposition taken ONCE (it was k times: 1.5 ms
    // of host arithmetic for the 30 secant pairs of a Davidson run); same operations in the same order
    Mp.assign((size_t)k * k, 0.0);
    if (k == 0) return;
    vec w(k), Z((size_t)k * k), work(k);
    small::sym_eig(k, M.data(), k, w.data(), Z.data(), k, work.data());
    double wmax = 0.0;
    for (int i = 0; i < k; ++i) wmax = std::max(wmax, fabs(w[i]));
    const double cut = 2.220446049250313e-16 * k * wmax;
    for (int col = 0; col < k; ++col)
        for (int j = 0; j < k; ++j) {
            if (fabs(w[j]) <= cut) continue;
            const double p = Z[(size_t)col * k + j] / w[j];
            for (int i = 0; i < k; ++i) Mp[(size_t)i * k + col] += Z[(size_t)i * k + j] * p;
        }
}

}  // namespace
}  // namespace sella

using namespace sella;

extern "C" int sella_symmetrize_y(sella_ctx* c, const double* S, const double* Y, int n, int k, int symm,
                                  double* out) {
    if (!c || !S || !Y || !out || n <= 0 || k <= 0) return SELLA_E_INVALID;
    const int ld = round_up(n, 8);
    double *Sp, *Yp, *Ytp;
    SCHK(scratch_get(c, SCR_UPD0, (size_t)k * ld * sizeof(double), &Sp));
    SCHK(scratch_get(c, SCR_UPD1, (size_t)k * ld * sizeof(double), &Yp));
    SCHK(scratch_get(c, SCR_UPD2, (size_t)k * ld * sizeof(double), &Ytp));
    SCHK(upload_panel(c, S, n, k, Sp, ld));
    SCHK(upload_panel(c, Y, n, k, Yp, ld));
    SCHK(symmetrize_panels(c, Sp, Yp, n, k, ld, symm, Ytp));
    return download_panel(c, Ytp, ld, n, k, out);
}

// Principal submatrix B[idx][idx] kept in step with B (the Hessian projected on the free coordinates of a
// constraint set that pins single coordinates, peswrapper.py:363-386): its matrix and, when given, its
// eigendecomposition receive the same update restricted to idx.
struct SubView {
    sella_mat B, V, Vt;
    double* evals;
    const int* idx;
    int m;
    int* nrank1;
    // structured eigendecomposition of the view (Vt = explicit rows, V unused) when lr_r != nullptr
    int* lr_r = nullptr;
    double lr_lam0 = 0.0;
};

// Structured eigendecomposition of B (eigh.hip, lr_lowrank_update): r explicit eigenpairs (mu ascending, rows of Wt)
// and the eigenvalue lam0 on the orthogonal complement of their span.
struct LrRef {
    sella_mat Wt;
    int* r;
    double* mu;
    double lam0;
    int* nrank1;
};

// |B| S for a structured B:  |lam0| S + W^T [(|mu| - |lam0|) (W S)]   (k rows, vector-major panels)
static int lr_abs_times(sella_ctx* c, const LrRef& lr, const double* Sp, int n, int k, int ld, double* mid, double* out) {
    SCHK(launch_axpby2d(c, k, n, fabs(lr.lam0), Sp, ld, 0.0, nullptr, 0, out, ld));
    const int r = *lr.r;
    if (r <= 0) return SELLA_OK;
    Mat* Wm = mat_get(c, lr.Wt);
    if (!Wm || Wm->cols != n || Wm->rows < r) { set_error("update_H: bad structured eigenvector handle"); return SELLA_E_INVALID; }
    std::vector<double> dv(r);
    for (int i = 0; i < r; ++i) dv[i] = fabs(lr.mu[i]) - fabs(lr.lam0);
    double* dev;
    SCHK(scratch_get(c, SCR_C, (size_t)std::max(ld, round_up(r, 8)) * sizeof(double), &dev));
    SCHK(h2d_async(c, dev, dv.data(), (size_t)r * sizeof(double)));
    GemvEpi e;
    e.mode = 4;
    e.dvec = dev;
    SCHK(launch_gemv_rows(c, Wm->d, r, n, Wm->ld, Sp, ld, k, mid, ld, e));
    for (int h = 0; h < k; ++h)
        SCHK(launch_lincomb(c, n, 1, Wm->d, Wm->ld, r, mid + (size_t)h * ld, 1, nullptr, 0, 0, nullptr, 0, 1.0,
                            out + (size_t)h * ld, ld));
    return SELLA_OK;
}

__device__ __forceinline__ void gather_cols_vb(const VB vb, const double* __restrict__ P, int ldp, int rows,
                                                          const int* __restrict__ idx, int m,
                                                          double* __restrict__ out, int ldo) {
    const int i = vb.x * 256 + threadIdx.x;
    const int r = vb.y;
    if (i < m && r < rows) out[(size_t)r * ldo + i] = P[(size_t)r * ldp + idx[i]];
}
__global__ __launch_bounds__(256) void gather_cols_kernel(const double* __restrict__ P, int ldp, int rows,
                                                          const int* __restrict__ idx, int m,
                                                          double* __restrict__ out, int ldo) { gather_cols_vb(vb_hw(), P, ldp, rows, idx, m, out, ldo); }

// k = 1 TS-BFGS (the per-step quasi-Newton update): the three scalars of hessian_update.py:120-126 stay on the
// device.  in: d[0] = s.ytilde, d[1] = s.|B|s, d[2] = j.s;  out: coef[0], coef[1] = pinv(G) [M1, M2] with
// G = M1^2 + M2^2, coef[2] = -1/2 sym(J^T S).  Same operations in the same order as the host k x k code.
__device__ __forceinline__ void tsbfgs_k1_coef_vb(const VB vb, const double* __restrict__ d, double* __restrict__ coef) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    if (threadIdx.x != 0 || vb.x != 0) return;
    const double m1 = d[0], m2 = d[1];
    const double g = m1 * m1 + m2 * m2;
    const double gp = (fabs(g) <= 2.220446049250313e-16 * fabs(g)) ? 0.0 : 1.0 / g;      // sym_pinv_solve's cut, m = 1
    coef[0] = gp * m1;
    coef[1] = gp * m2;
    coef[2] = -0.25 * (d[2] + d[2]);
}
__global__ __launch_bounds__(1024) void tsbfgs_k1_coef_kernel(const double* __restrict__ d, double* __restrict__ coef) { tsbfgs_k1_coef_vb(vb_hw(), d, coef); }

static int update_h_core(sella_ctx* c, sella_mat hB, sella_mat hV, sella_mat hVt, const double* evals,
                         const double* S, const double* Y, int n, int k, int method, int symm,
                         double* evals_io, int max_rank, int* nrank1, const SubView* sv = nullptr,
                         const LrRef* lr = nullptr) {
    if (nrank1) *nrank1 = -1;
    if (sv && sv->nrank1) *sv->nrank1 = -1;
    const bool dbg_time = getenv("SELLA_DEBUG_TIMING") != nullptr;
    auto now = [&] { if (dbg_time) (void)hipStreamSynchronize(c->stream); return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_u0 = now();
    Mat* B = mat_get(c, hB);
    if (!B || !S || !Y || k <= 0) return SELLA_E_INVALID;
    if (B->rows != n || B->cols != n) { set_error("update_H: B must be %d x %d", n, n); return SELLA_E_INVALID; }
    if (method < SELLA_UPD_TS_BFGS || method > SELLA_UPD_BFGS_AUTO) {
        set_error("Unknown update method %d", method);
        return SELLA_E_INVALID;
    }
    const int ld = round_up(n, 8);
    double *Sp, *Yp, *Ytp, *BSp, *wk;
    SCHK(scratch_get(c, SCR_UPD0, (size_t)k * ld * sizeof(double), &Sp));
    SCHK(scratch_get(c, SCR_UPD1, (size_t)k * ld * sizeof(double), &Yp));
    SCHK(scratch_get(c, SCR_UPD2, (size_t)k * ld * sizeof(double), &Ytp));
    SCHK(scratch_get(c, SCR_UPD3, (size_t)6 * k * ld * sizeof(double), &wk));
    BSp = wk;                                    // k rows
    double* Jp = wk + (size_t)k * ld;            // k rows
    double* Up = wk + 2 * (size_t)k * ld;        // up to 2k rows
    double* Zp = wk + 4 * (size_t)k * ld;        // up to 2k rows
    SCHK(upload_panel(c, S, n, k, Sp, ld));
    SCHK(upload_panel(c, Y, n, k, Yp, ld));
    SCHK(symmetrize_panels(c, Sp, Yp, n, k, ld, symm, Ytp));
    B = mat_get(c, hB);
    // B S (k right-hand sides through the row-panel matvec) and J = Ytilde - B S
    SCHK(launch_gemv_rows(c, B->d, n, n, B->ld, Sp, ld, k, BSp, ld, GemvEpi()));
    SCHK(launch_axpby2d(c, k, n, 1.0, Ytp, ld, -1.0, BSp, ld, Jp, ld));

    hostm::vec STY, STS, C, tmp, Minv;
    if (method == SELLA_UPD_BFGS_AUTO) {                                   // hessian_update.py:80-87
        method = SELLA_UPD_TS_BFGS;
        bool pd = evals != nullptr || lr != nullptr;
        if (pd && lr) {
            pd = lr->lam0 > 0.0;
            for (int i = 0; i < *lr->r && pd; ++i) pd = lr->mu[i] > 0.0;
        } else if (pd) {
            if (hV == SELLA_NO_MAT) pd = evals[0] > 0.0;
            else for (int i = 0; i < n; ++i) if (!(evals[i] > 0.0)) { pd = false; break; }
        }
        if (pd) {
            SCHK(gram(c, Sp, k, Ytp, k, n, ld, STY));
            SCHK(gram(c, Sp, k, Sp, k, n, ld, STS));
            hostm::vec lam(k), Wv((size_t)k * k);
            if (hostm::gen_sym_eig(k, STY.data(), STS.data(), lam.data(), Wv.data()) == 0) {
                bool allpos = true;
                for (int i = 0; i < k; ++i) allpos = allpos && lam[i] > 0.0;
                if (allpos) method = SELLA_UPD_BFGS;
            }
        }
    }

    int kk = k;
    bool uz_done = false;                    // U and Z already formed on the device (k = 1 TS-BFGS)
    auto panel_times = [&](const hostm::vec& M, const double* P, double* out, size_t off) -> int {
        // out_a = sum_b M[a][b] P_b   (k x k times a k-row panel)
        hostm::vec Mt((size_t)k * k);
        for (int a = 0; a < k; ++a)
            for (int b = 0; b < k; ++b) Mt[(size_t)b * k + a] = M[(size_t)a * k + b];
        double* dM;
        SCHK(put_k(c, Mt, off, &dM));
        return launch_lincomb(c, n, k, P, ld, k, dM, k, nullptr, 0, 0, nullptr, 0, 0.0, out, ld);
    };

    if (method == SELLA_UPD_SR1) {
        // Delta = J (J^T S)^-1 J^T ;  U = J, Z = 1/2 sym(C^-1) J
        SCHK(gram(c, Jp, k, Sp, k, n, ld, C));
        if (!invert(k, C, Minv)) { set_error("SR1 update: singular J^T S"); return SELLA_E_NOCONV; }
        for (int a = 0; a < k; ++a)
            for (int b = 0; b <= a; ++b) {
                const double v = 0.25 * (Minv[(size_t)a * k + b] + Minv[(size_t)b * k + a]);
                Minv[(size_t)a * k + b] = Minv[(size_t)b * k + a] = v;
            }
        SCHK(launch_axpby2d(c, k, n, 1.0, Jp, ld, 0.0, nullptr, 0, Up, ld));
        SCHK(panel_times(Minv, Jp, Zp, 0));
    } else if (method == SELLA_UPD_BFGS) {
        // Delta = Y (Y^T S)^-1 Y^T - BS (S^T B S)^-1 (BS)^T
        hostm::vec YTS, SBS, I1, I2;
        SCHK(gram(c, Ytp, k, Sp, k, n, ld, YTS));
        SCHK(gram(c, Sp, k, BSp, k, n, ld, SBS));
        if (!invert(k, YTS, I1) || !invert(k, SBS, I2)) { set_error("BFGS update: singular curvature matrix"); return SELLA_E_NOCONV; }
        for (int a = 0; a < k; ++a)
            for (int b = 0; b <= a; ++b) {
                double v = 0.25 * (I1[(size_t)a * k + b] + I1[(size_t)b * k + a]);
                I1[(size_t)a * k + b] = I1[(size_t)b * k + a] = v;
                v = -0.25 * (I2[(size_t)a * k + b] + I2[(size_t)b * k + a]);
                I2[(size_t)a * k + b] = I2[(size_t)b * k + a] = v;
            }
        SCHK(launch_axpby2d(c, k, n, 1.0, Ytp, ld, 0.0, nullptr, 0, Up, ld));
        SCHK(launch_axpby2d(c, k, n, 1.0, BSp, ld, 0.0, nullptr, 0, Up + (size_t)k * ld, ld));
        SCHK(panel_times(I1, Ytp, Zp, 0));
        SCHK(panel_times(I2, BSp, Zp + (size_t)k * ld, (size_t)k * k));
        kk = 2 * k;
    } else {
        // U-family: Delta = U J^T + J U^T - U (J^T S) U^T
        if (method == SELLA_UPD_TS_BFGS) {
            // |B| S = Q (|lam| * (Q^T S))                                      hessian_update.py:121
            double* absBS = Zp;                  // temporary use of the Z rows
            double* mid = Up;
            if (lr) {
                SCHK(lr_abs_times(c, *lr, Sp, n, k, ld, mid, absBS));
            } else if (hV == SELLA_NO_MAT) {
                if (!evals) { set_error("TS-BFGS needs the eigendecomposition of B"); return SELLA_E_INVALID; }
                SCHK(launch_axpby2d(c, k, n, fabs(evals[0]), Sp, ld, 0.0, nullptr, 0, absBS, ld));
            } else {
                Mat *V = mat_get(c, hV), *Vt = mat_get(c, hVt);
                if (!V || !Vt || !evals) { set_error("TS-BFGS needs evecs, evecsT and evals"); return SELLA_E_INVALID; }
                double* dev;
                SCHK(scratch_get(c, SCR_C, (size_t)ld * sizeof(double), &dev));
                SCHK(h2d_async(c, dev, evals, (size_t)n * sizeof(double)));
                GemvEpi e;
                e.mode = 3;
                e.dvec = dev;
                SCHK(launch_gemv_rows(c, Vt->d, n, n, Vt->ld, Sp, ld, k, mid, ld, e));
                SCHK(launch_gemv_rows(c, V->d, n, n, V->ld, mid, ld, k, absBS, ld, GemvEpi()));
            }
            // X = X1 + X2 = M1 Ytilde^T + M2 absBS^T with M1 = S^T Ytilde, M2 = S^T absBS  (k x n)
            // G = X S = M1 M1^T + M2 M2^T ;  U^T = pinv(G) X                  hessian_update.py:120-123
            if (k == 1) {
                // one secant pair: no host round trip (three dots, one single-thread kernel, two combinations)
                double* dd = c->dscal + DS_CVEC;
                double* dcoef;
                SCHK(scratch_get(c, SCR_W, (size_t)(1 << 16) * sizeof(double), &dcoef));
                SCHK(launch_gemv_rows(c, Ytp, 1, n, ld, Sp, ld, 1, dd, 1, GemvEpi()));
                SCHK(launch_gemv_rows(c, absBS, 1, n, ld, Sp, ld, 1, dd + 1, 1, GemvEpi()));
                SCHK(launch_gemv_rows(c, Jp, 1, n, ld, Sp, ld, 1, dd + 2, 1, GemvEpi()));
                SELLA_LAUNCHB(c, tsbfgs_k1_coef_kernel, tsbfgs_k1_coef_vb, 1024, dim3(1), dim3(64), 0, dd, dcoef);
                HIPCHK(hipGetLastError());
                SCHK(launch_lincomb(c, n, 1, Ytp, ld, 1, dcoef, 1, absBS, ld, 1, dcoef + 1, 1, 0.0, Up, ld));
                SCHK(launch_axpby2d(c, 1, n, 1.0, Jp, ld, 0.0, nullptr, 0, Zp, ld));
                SCHK(launch_lincomb(c, n, 1, Up, ld, 1, dcoef + 2, 1, nullptr, 0, 0, nullptr, 0, 1.0, Zp, ld));
                uz_done = true;
            }
            hostm::vec M1, M2, G((size_t)k * k, 0.0), Gp, C1((size_t)k * k), C2((size_t)k * k);
            if (!uz_done) {
            SCHK(gram(c, Sp, k, Ytp, k, n, ld, M1));
            SCHK(gram(c, Sp, k, absBS, k, n, ld, M2));
            for (int a = 0; a < k; ++a)
                for (int b = 0; b < k; ++b) {
                    double s = 0.0;
                    for (int l = 0; l < k; ++l)
                        s += M1[(size_t)a * k + l] * M1[(size_t)b * k + l] + M2[(size_t)a * k + l] * M2[(size_t)b * k + l];
                    G[(size_t)a * k + b] = s;
                }
            sym_pinv(k, G, Gp);
            for (int a = 0; a < k; ++a)
                for (int b = 0; b < k; ++b) {
                    double s1 = 0.0, s2 = 0.0;
                    for (int l = 0; l < k; ++l) {
                        s1 += Gp[(size_t)a * k + l] * M1[(size_t)l * k + b];
                        s2 += Gp[(size_t)a * k + l] * M2[(size_t)l * k + b];
                    }
                    // lincomb wants W[j][c] = coefficient of panel row j in output row c
                    C1[(size_t)b * k + a] = s1;
                    C2[(size_t)b * k + a] = s2;
                }
            double *d1, *d2;
            SCHK(put_k(c, C1, 0, &d1));
            SCHK(put_k(c, C2, (size_t)k * k, &d2));
            // Up cannot alias its inputs: absBS lives in Zp, Ytp is separate; write U into Up
            SCHK(launch_lincomb(c, n, k, Ytp, ld, k, d1, k, absBS, ld, k, d2, k, 0.0, Up, ld));
            }
        } else if (method == SELLA_UPD_PSB) {                                   // U = S (S^T S)^-1
            SCHK(gram(c, Sp, k, Sp, k, n, ld, STS));
            if (!invert(k, STS, Minv)) { set_error("PSB update: singular S^T S"); return SELLA_E_NOCONV; }
            SCHK(panel_times(Minv, Sp, Up, 0));
        } else if (method == SELLA_UPD_DFP) {                                   // U^T = (S^T Y)^-1 Y^T
            SCHK(gram(c, Sp, k, Ytp, k, n, ld, STY));
            if (!invert(k, STY, Minv)) { set_error("DFP update: singular S^T Y"); return SELLA_E_NOCONV; }
            SCHK(panel_times(Minv, Ytp, Up, 0));
        } else {                                                                // Greenstadt
            hostm::vec SBS;
            SCHK(gram(c, Sp, k, BSp, k, n, ld, SBS));
            if (!invert(k, SBS, Minv)) { set_error("Greenstadt update: singular S^T B S"); return SELLA_E_NOCONV; }
            SCHK(panel_times(Minv, BSp, Up, 0));
        }
        // Z = J - 1/2 sym(J^T S) U
        if (!uz_done) {
        SCHK(gram(c, Jp, k, Sp, k, n, ld, C));
        hostm::vec Cs((size_t)k * k);
        for (int a = 0; a < k; ++a)
            for (int b = 0; b < k; ++b) Cs[(size_t)a * k + b] = -0.25 * (C[(size_t)a * k + b] + C[(size_t)b * k + a]);
        // Z_a = J_a + sum_b Cs[a][b] U_b
        SCHK(launch_axpby2d(c, k, n, 1.0, Jp, ld, 0.0, nullptr, 0, Zp, ld));
        hostm::vec CsT((size_t)k * k);
        for (int a = 0; a < k; ++a)
            for (int b = 0; b < k; ++b) CsT[(size_t)b * k + a] = Cs[(size_t)a * k + b];
        double* dC;
        SCHK(put_k(c, CsT, 2 * (size_t)k * k, &dC));
        SCHK(launch_lincomb(c, n, k, Up, ld, k, dC, k, nullptr, 0, 0, nullptr, 0, 1.0, Zp, ld));
        }
    }
    const double t_u1 = now();
    B = mat_get(c, hB);
    SCHK(launch_sym_rank2k(c, B->d, n, B->ld, Up, Zp, ld, kk));
    double *Us = nullptr, *Zs = nullptr;
    int lds = 0;
    if (sv) {
        // the update vectors restricted to the view's coordinates (before the panels are consumed below)
        Mat* Bs = mat_get(c, sv->B);
        const int m = sv->m;
        if (!Bs || Bs->rows != m || Bs->cols != m || m <= 0 || m > n || !sv->idx) {
            set_error("update_H: bad principal-submatrix view");
            return SELLA_E_INVALID;
        }
        lds = round_up(m, 8);
        double* wks;
        SCHK(scratch_get(c, SCR_UPD4, ((size_t)2 * kk * lds + (size_t)m / 2 + 8) * sizeof(double), &wks));
        Us = wks;
        Zs = wks + (size_t)kk * lds;
        int* didx = reinterpret_cast<int*>(wks + 2 * (size_t)kk * lds);
        SCHK(h2d_async(c, didx, sv->idx, (size_t)m * sizeof(int)));
        HIPCHK(s_memset0(c, wks, (size_t)2 * kk * lds * sizeof(double)));
        const dim3 gg((m + 255) / 256, kk);
        SELLA_LAUNCHB(c, gather_cols_kernel, gather_cols_vb, 256, gg, dim3(256), 0, Up, ld, kk, didx, m, Us, lds);
        SELLA_LAUNCHB(c, gather_cols_kernel, gather_cols_vb, 256, gg, dim3(256), 0, Zp, ld, kk, didx, m, Zs, lds);
        HIPCHK(hipGetLastError());
        SCHK(launch_sym_rank2k(c, Bs->d, m, Bs->ld, Us, Zs, lds, kk));
    }
    const double t_u2 = now();
    if (lr) {
        // structured eigendecomposition: every rank is carried (O(n r) per rank-one term), 32 pairs per call
        Mat* Wm = mat_get(c, lr->Wt);
        if (!Wm || Wm->cols != n) { set_error("update_H: bad structured eigenvector handle"); return SELLA_E_INVALID; }
        int total = 0;
        for (int a0 = 0; a0 < kk; a0 += 32) {
            const int kc = std::min(32, kk - a0);
            int nr1 = 0;
            SCHK(lr_lowrank_update(c, n, lr->r, lr->mu, lr->lam0, Wm, Up + (size_t)a0 * ld, Zp + (size_t)a0 * ld, ld, kc, &nr1));
            total += nr1;
        }
        if (lr->nrank1) *lr->nrank1 = total;
    } else if (evals_io && hV != SELLA_NO_MAT && 2 * kk <= max_rank) {
        Mat *V = mat_get(c, hV), *Vt = mat_get(c, hVt);
        if (!V || !Vt || V->rows != n || Vt->rows != n) { set_error("update_H: bad eigenvector handles"); return SELLA_E_INVALID; }
        SCHK(eig_lowrank_update(c, n, evals_io, V, Vt, Up, Zp, ld, kk, nrank1));
    }
    const double t_u3 = now();
    if (sv && sv->lr_r) {
        Mat* Wm = mat_get(c, sv->Vt);
        if (!Wm || Wm->cols != sv->m) { set_error("update_H: bad structured eigenvector handle of the view"); return SELLA_E_INVALID; }
        int total = 0;
        for (int a0 = 0; a0 < kk; a0 += 32) {
            const int kc = std::min(32, kk - a0);
            int nr1 = 0;
            SCHK(lr_lowrank_update(c, sv->m, sv->lr_r, sv->evals, sv->lr_lam0, Wm, Us + (size_t)a0 * lds, Zs + (size_t)a0 * lds,
                                   lds, kc, &nr1));
            total += nr1;
        }
        if (sv->nrank1) *sv->nrank1 = total;
    } else if (sv && sv->evals && sv->V != SELLA_NO_MAT && 2 * kk <= max_rank) {
        Mat *V = mat_get(c, sv->V), *Vt = mat_get(c, sv->Vt);
        if (!V || !Vt || V->rows != sv->m || Vt->rows != sv->m) { set_error("update_H: bad eigenvector handles of the view"); return SELLA_E_INVALID; }
        SCHK(eig_lowrank_update(c, sv->m, sv->evals, V, Vt, Us, Zs, lds, kk, sv->nrank1));
    }
    SCHK(stream_wait(c));
    if (dbg_time)
        fprintf(stderr, "update_H n=%d k=%d: vectors %.3f ms, rank-2k (+view gather) %.3f ms, eigen-update %.3f ms (%d rank-one), view %.3f ms (%d)\n",
                n, k, 1e3 * (t_u1 - t_u0), 1e3 * (t_u2 - t_u1), 1e3 * (t_u3 - t_u2), nrank1 ? *nrank1 : -1,
                1e3 * (now() - t_u3), (sv && sv->nrank1) ? *sv->nrank1 : -1);
    return SELLA_OK;
}

extern "C" int sella_update_h_eig_view(sella_ctx* c, sella_mat hB, sella_mat hV, sella_mat hVt, double* evals,
                                       const double* S, const double* Y, int n, int k, int method, int symm,
                                       int max_rank, int* nrank1, sella_mat hBsub, sella_mat hVsub, sella_mat hVtsub,
                                       double* evals_sub, const int* idx, int m, int* nrank1_sub) {
    if (!evals || !nrank1 || !idx || !nrank1_sub) { set_error("update_H (view): evals, nrank1, idx and nrank1_sub are required"); return SELLA_E_INVALID; }
    for (int i = 0; i < m; ++i)
        if (idx[i] < 0 || idx[i] >= n || (i && idx[i] <= idx[i - 1])) { set_error("update_H (view): idx must be ascending in [0, n)"); return SELLA_E_INVALID; }
    SubView sv{hBsub, hVsub, hVtsub, evals_sub, idx, m, nrank1_sub};
    return update_h_core(c, hB, hV, hVt, evals, S, Y, n, k, method, symm, evals, max_rank, nrank1, &sv);
}

extern "C" int sella_update_h(sella_ctx* c, sella_mat hB, sella_mat hV, sella_mat hVt, const double* evals,
                              const double* S, const double* Y, int n, int k, int method, int symm) {
    return update_h_core(c, hB, hV, hVt, evals, S, Y, n, k, method, symm, nullptr, 0, nullptr);
}

extern "C" int sella_update_h_eig(sella_ctx* c, sella_mat hB, sella_mat hV, sella_mat hVt, double* evals,
                                  const double* S, const double* Y, int n, int k, int method, int symm,
                                  int max_rank, int* nrank1) {
    if (!evals || !nrank1) { set_error("update_H (eig): evals and nrank1 are required"); return SELLA_E_INVALID; }
    return update_h_core(c, hB, hV, hVt, evals, S, Y, n, k, method, symm, evals, max_rank, nrank1);
}

// Quasi-Newton update of a matrix held as dense B PLUS a structured eigendecomposition (lam0 * I + rank r): B updated in
// place as in sella_update_h; the r explicit eigenpairs (mu, rows of Wt) are carried by rank-one merges on r + 1 rows —
// O(n r) traffic instead of the O(n^2) passes of sella_update_h_eig.  Optionally a principal-submatrix view with a
// structured eigendecomposition of its own (lam0 the same) is kept in step.
extern "C" int sella_update_h_lr(sella_ctx* c, sella_mat hB, sella_mat hWt, int* r, double* mu, double lam0,
                                 const double* S, const double* Y, int n, int k, int method, int symm, int* nrank1,
                                 sella_mat hBsub, sella_mat hWtsub, int* rsub, double* musub, const int* idx, int m,
                                 int* nrank1_sub) {
    if (!c || !r || !mu || !nrank1) { set_error("update_H (structured): r, mu and nrank1 are required"); return SELLA_E_INVALID; }
    Mat* Wm = mat_get(c, hWt);
    if (!Wm || *r < 0 || *r > Wm->rows) { set_error("update_H (structured): bad eigenvector handle or rank"); return SELLA_E_INVALID; }
    LrRef lr{hWt, r, mu, lam0, nrank1};
    if (hBsub == SELLA_NO_MAT)
        return update_h_core(c, hB, SELLA_NO_MAT, SELLA_NO_MAT, nullptr, S, Y, n, k, method, symm, nullptr, 0, nullptr, nullptr, &lr);
    if (!idx || !nrank1_sub) { set_error("update_H (structured view): idx and nrank1_sub are required"); return SELLA_E_INVALID; }
    for (int i = 0; i < m; ++i)
        if (idx[i] < 0 || idx[i] >= n || (i && idx[i] <= idx[i - 1])) { set_error("update_H (view): idx must be ascending in [0, n)"); return SELLA_E_INVALID; }
    SubView sv{hBsub, SELLA_NO_MAT, hWtsub, musub, idx, m, nrank1_sub};
    if (hWtsub != SELLA_NO_MAT && rsub && musub) { sv.lr_r = rsub; sv.lr_lam0 = lam0; }
    return update_h_core(c, hB, SELLA_NO_MAT, SELLA_NO_MAT, nullptr, S, Y, n, k, method, symm, nullptr, 0, nullptr, &sv, &lr);
}

// Structured eigendecomposition of a PRINCIPAL SUBMATRIX: B = lam0 I + W^T diag(mu - lam0) W restricted to the ascending
// coordinates idx (m of them) is lam0 I_m + Wf^T diag(mu - lam0) Wf with Wf = W[:, idx] — rank <= r again, but the rows
// of Wf are no longer orthonormal.  They are orthonormalised on the device (Gram-Schmidt with the reference's
// accept / drop rules, gs.hip), Wf = R Qf, the r' x r' core R^T diag(mu - lam0) R is diagonalised on the host, and
// the explicit eigenvectors of the submatrix are F^T Qf.  What `get_HL_projected` needs when the constraints pin single
// coordinates (peswrapper.py:363-386) without an m x m eigh.   Wt_sub: capacity rows x m; r_sub, mu_sub outputs.
extern "C" int sella_lr_restrict(sella_ctx* c, sella_mat hWt, int r, const double* mu, double lam0, const int* idx, int m,
                                 sella_mat hWt_sub, int* r_sub, double* mu_sub) {
    if (!c || !r_sub || !mu_sub || r < 0 || m <= 0 || !idx || (r > 0 && !mu)) {
        set_error("lr_restrict: invalid arguments");
        return SELLA_E_INVALID;
    }
    Mat *Wm = mat_get(c, hWt), *Ws = mat_get(c, hWt_sub);
    if (!Wm || !Ws || Wm->rows < r || Ws->cols != m || Ws->rows < std::min(r, m)) {
        set_error("lr_restrict: bad handles (need %d rows of capacity)", std::min(r, m));
        return SELLA_E_INVALID;
    }
    *r_sub = 0;
    if (r == 0) return SELLA_OK;
    const int n = Wm->cols;
    for (int i = 0; i < m; ++i)
        if (idx[i] < 0 || idx[i] >= n || (i && idx[i] <= idx[i - 1])) { set_error("lr_restrict: idx must be ascending in [0, n)"); return SELLA_E_INVALID; }
    if (r > 256) { set_error("lr_restrict: rank %d too large for the host-side core", r); return SELLA_E_UNSUPPORTED; }
    const int lds = Ws->ld;
    double* wk;
    SCHK(scratch_get(c, SCR_UPD4, ((size_t)2 * r * lds + (size_t)m / 2 + 8 + (size_t)r * round_up(r, 8)) * sizeof(double), &wk));
    double* Wf = wk;                               // r x m   (restricted rows)
    double* Qf = wk + (size_t)r * lds;             // up to r x m (orthonormalised)
    int* didx = reinterpret_cast<int*>(wk + 2 * (size_t)r * lds);
    double* Rd = wk + 2 * (size_t)r * lds + (size_t)m / 2 + 8;      // r' x r coordinates (device)
    SCHK(h2d_async(c, didx, idx, (size_t)m * sizeof(int)));
    HIPCHK(s_memset0(c, wk, (size_t)2 * r * lds * sizeof(double)));
    Wm = mat_get(c, hWt);
    SELLA_LAUNCHB(c, gather_cols_kernel, gather_cols_vb, 256, dim3((m + 255) / 256, r), dim3(256), 0, Wm->d, Wm->ld, r, didx, m, Wf, lds);
    HIPCHK(hipGetLastError());
    int rq = 0;
    for (int v = 0; v < r && rq < m; ++v) {
        double* slot = Qf + (size_t)rq * lds;
        HIPCHK(s_memcpy(c, slot, Wf + (size_t)v * lds, (size_t)lds * sizeof(double), hipMemcpyDeviceToDevice));
        int kept = 0;
        SCHK(gs_orthonormalise(c, Qf, lds, rq, slot, m, 1e-15, 1e-13, 100, &kept, nullptr));
        if (kept) ++rq;
    }
    if (rq == 0) return SELLA_OK;
    // R[j][v] = Qf_j . Wf_v   (rq x r): Wf_v = sum_j R[j][v] Qf_j
    const int ldr = round_up(r, 8);
    SCHK(launch_gemm(c, 0, 1, rq, r, m, 1.0, Qf, lds, Wf, lds, 0.0, Rd, ldr));
    std::vector<double> R((size_t)rq * r);
    SCHK(d2h_async_2d(c, R.data(), Rd, (size_t)ldr * sizeof(double), (size_t)r * sizeof(double), rq));
    SCHK(stream_wait(c));
    // core C = R diag(mu - lam0) R^T (rq x rq), eigenpairs C = F Sigma F^T
    std::vector<double> C((size_t)rq * rq), sig(rq), F((size_t)rq * rq), work(rq);
    for (int i = 0; i < rq; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = 0.0;
            for (int v = 0; v < r; ++v) s += R[(size_t)i * r + v] * (mu[v] - lam0) * R[(size_t)j * r + v];
            C[(size_t)i * rq + j] = C[(size_t)j * rq + i] = s;
        }
    if (small::sym_eig(rq, C.data(), rq, sig.data(), F.data(), rq, work.data()) != 0) {
        set_error("lr_restrict: small eigenproblem did not converge");
        return SELLA_E_NOCONV;
    }
    // W_sub row t = sum_j F[j][t] Qf_j  -> lincomb with coefficient matrix (rq x rq), W1[j*ldw + t] = F[j][t]
    double* dF;
    SCHK(put_k(c, F, 0, &dF));
    Ws = mat_get(c, hWt_sub);
    SCHK(launch_lincomb(c, m, rq, Qf, lds, rq, dF, rq, nullptr, 0, 0, nullptr, 0, 0.0, Ws->d, Ws->ld));
    for (int t = 0; t < rq; ++t) mu_sub[t] = lam0 + sig[t];
    *r_sub = rq;
    return stream_wait(c);
}
