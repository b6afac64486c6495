// davidson_block.hip — block Davidson for the lowest eigenpairs of a large dense symmetric operator
// (BASELINE.json configs[4]: 3N = 12288, 16 new vectors per iteration, H.V panel on the matrix cores,
// rows of H optionally sharded over the GPUs of one node).
//
// The reference has no block method: rayleigh_ritz adds ONE vector per iteration
// (sella/eigensolvers.py:111-112), i.e. one full stream of the matrix per vector.  At 3N = 12288 the matrix
// is 1.2 GB, so the MI355X-shaped formulation streams it ONCE for 16 right-hand sides
// (`panel16_mfma_kernel`, kernels.hip) and expands the subspace by a block.  Parity target: the converged
// eigenpairs equal those of exact() (sella/eigensolvers.py:9-28) — there is no trajectory to match.
//
// Structure of one iteration (all panels vector-major, k rows x n, same leading dimension as the matrix):
//   Rayleigh-Ritz on the host (k x k, G = V^T A V kept incrementally);
//   residuals of the lowest `nev` Ritz pairs  R = (AV) W - (V W) diag(theta)        [2 combine launches]
//   correction  T = (P - theta)^-1 R  through the eigenbasis of P (two 16-RHS panel products), a diagonal,
//   or none; block Gram-Schmidt against V (two passes) with an SVQB step inside the block (16 x 16 Gram
//   matrix on the host, drops numerically dependent directions: the block analogue of mgs' eps2 rule,
//   sella/utilities/math.pyx:112-117); A T on the matrix cores (+ ONE all-gather when the rows are sharded);
//   one panel product for the new Gram rows; thick restart when the basis is full.
#include "internal.h"
#include <chrono>
#include "host_math.h"

namespace sella {
namespace {

using hostm::vec;
constexpr int BD_NB = 16;      // block width = right-hand sides of one panel16 pass
constexpr int BD_HG = 4;       // outputs per thread of the combine kernel

// out[h][i] = beta out[h][i] + alpha sum_a C[h ldc + a] P[a ldp + i],  h < nh (<= 16), a < k.
// Thread i owns element i of BD_HG outputs (blockIdx.y picks the group): coalesced panel reads, coefficients
// broadcast from LDS.
// rowscale (device, nh entries, may be null): output h is additionally scaled by rowscale[h].  base (may be null): the
// beta term is read from base (same layout as out) instead of out — an out-of-place update.
__global__ __launch_bounds__(256) void bd_combine_kernel(int n, int nh, int k, const double* __restrict__ C, int ldc,
                                                         const double* __restrict__ P, int ldp, double alpha, double beta,
                                                         double* __restrict__ out, int ldo,
                                                         const double* __restrict__ rowscale,
                                                         const double* __restrict__ base) {
    __shared__ double cs[128][BD_HG];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int h0 = blockIdx.y * BD_HG;
    double acc[BD_HG];
#pragma unroll
    for (int h = 0; h < BD_HG; ++h) acc[h] = 0.0;
    for (int a0 = 0; a0 < k; a0 += 128) {
        const int jt = (k - a0 < 128) ? (k - a0) : 128;
        __syncthreads();
        for (int t = threadIdx.x; t < jt * BD_HG; t += 256) {
            const int a = t / BD_HG, h = t % BD_HG;
            cs[a][h] = (h0 + h < nh) ? C[(size_t)(h0 + h) * ldc + a0 + a] : 0.0;
        }
        __syncthreads();
        if (i < n) {
#pragma unroll 4
            for (int a = 0; a < jt; ++a) {
                const double p = P[(size_t)(a0 + a) * ldp + i];
#pragma unroll
                for (int h = 0; h < BD_HG; ++h) acc[h] += cs[a][h] * p;
            }
        }
    }
    if (i < n) {
#pragma unroll
        for (int h = 0; h < BD_HG; ++h)
            if (h0 + h < nh) {
                double* o = out + (size_t)(h0 + h) * ldo + i;
                const double al = rowscale ? alpha * rowscale[h0 + h] : alpha;
                const double b0 = (beta == 0.0) ? 0.0 : (base ? base[(size_t)(h0 + h) * ldo + i] : *o);
                *o = (beta == 0.0) ? al * acc[h] : (beta * b0 + al * acc[h]);
            }
    }
}

struct Theta16 {
    double v[BD_NB];
};

// X[h][i] <- X[h][i] / (d[i] - theta_h), the denominator kept away from zero (|.| >= guard, sign preserved):
// the eigenbasis form of (P - theta)^-1 would otherwise inject inf when a Ritz value hits an eigenvalue of P.
__global__ __launch_bounds__(256) void bd_shift_scale_kernel(int n, int nh, double* __restrict__ X, int ld,
                                                             const double* __restrict__ d, Theta16 th, double guard) {
    const int i = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y;
    if (i >= n || h >= nh) return;
    double den = d[i] - th.v[h];
    if (fabs(den) < guard) den = (den < 0.0) ? -guard : guard;
    X[(size_t)h * ld + i] /= den;
}

// recv[(r * 16 + h) * m_max + i]  ->  Y[h ldy + r m_max + i]   (row-sharded block product after the all-gather)
__global__ __launch_bounds__(256) void bd_unpack_kernel(const double* __restrict__ recv, int world, int m_max, int n,
                                                        double* __restrict__ Y, int ldy) {
    const int g = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y;
    if (g >= n) return;
    const int r = g / m_max, i = g - r * m_max;
    Y[(size_t)h * ldy + g] = recv[((size_t)r * BD_NB + h) * m_max + i];
}

// ---- kernels of the pipelined iteration (run_pipelined below) ------------------------------------------------------
// out[z][h][i] = sum_a C[h ldc + a] P_z[a ldp + i] for TWO panels in one launch (blockIdx.z): the thick restart
// (V, AV) <- (W^T V, W^T AV) and the block's final transformation (T, A T) <- (S T', S (A T)'), written straight into
// their slots of the basis.  Columns [n, ld) of the outputs are left alone (zero since allocation).
__global__ __launch_bounds__(256) void bd_combine_pair_kernel(int n, int nh, int k, const double* __restrict__ C, int ldc,
                                                              const double* __restrict__ P0, const double* __restrict__ P1,
                                                              int ldp, double* __restrict__ out0, double* __restrict__ out1,
                                                              int ldo) {
    __shared__ double cs[128][BD_HG];
    const double* __restrict__ P = blockIdx.z ? P1 : P0;
    double* __restrict__ out = blockIdx.z ? out1 : out0;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int h0 = blockIdx.y * BD_HG;
    double acc[BD_HG];
#pragma unroll
    for (int h = 0; h < BD_HG; ++h) acc[h] = 0.0;
    for (int a0 = 0; a0 < k; a0 += 128) {
        const int jt = (k - a0 < 128) ? (k - a0) : 128;
        __syncthreads();
        for (int t = threadIdx.x; t < jt * BD_HG; t += 256) {
            const int a = t / BD_HG, h = t % BD_HG;
            cs[a][h] = (h0 + h < nh) ? C[(size_t)(h0 + h) * ldc + a0 + a] : 0.0;
        }
        __syncthreads();
        if (i < n) {
#pragma unroll 4
            for (int a = 0; a < jt; ++a) {
                const double p = P[(size_t)(a0 + a) * ldp + i];
#pragma unroll
                for (int h = 0; h < BD_HG; ++h) acc[h] += cs[a][h] * p;
            }
        }
    }
    if (i < n) {
#pragma unroll
        for (int h = 0; h < BD_HG; ++h)
            if (h0 + h < nh) out[(size_t)(h0 + h) * ldo + i] = acc[h];
    }
}

// Residuals of the lowest Ritz pairs when the basis IS the Ritz basis (right after a thick restart):
// R_h = (AV)_h - theta_h V_h, and the diagonally preconditioned correction T_h = R_h / (d - theta_h) beside it
// (d null: T = R), the latter twice: into the rows behind the basis and into scratch rows (T2, may be null).  Rows
// [nh, 16) of all of them are zero-filled: T is the operand of a 16-row panel product.  npart (may be null, pinned host
// memory): npart[h * gridDim.x + b] = sum of R_h^2 over the columns of workgroup b — the host adds them up in order.
__device__ __forceinline__ void bd_block_sumsq(double v, double* __restrict__ dst) {
    __shared__ double red[4];
    const double w = wave_sum64(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) *dst = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
}

__global__ __launch_bounds__(256) void bd_resid_ritz_kernel(int n, int nh, const double* __restrict__ V,
                                                            const double* __restrict__ AV, int ld, Theta16 th,
                                                            const double* __restrict__ d, double guard,
                                                            double* __restrict__ R, double* __restrict__ T,
                                                            double* __restrict__ T2, double* __restrict__ npart) {
    const int i = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y;
    double r = 0.0, t = 0.0;
    if (i < n && h < nh) {
        r = AV[(size_t)h * ld + i] - th.v[h] * V[(size_t)h * ld + i];
        t = r;
        if (d) {
            double den = d[i] - th.v[h];
            if (fabs(den) < guard) den = (den < 0.0) ? -guard : guard;
            t = r / den;
        }
    }
    if (i < n) {
        R[(size_t)h * ld + i] = r;
        T[(size_t)h * ld + i] = t;
        if (T2) T2[(size_t)h * ld + i] = t;
    }
    if (npart) bd_block_sumsq(r * r, npart + (size_t)h * gridDim.x + blockIdx.x);
}

// The same from general Ritz coefficients (before the first restart): R_h = sum_a C[h][a] (AV_a - theta_h V_a).
__global__ __launch_bounds__(256) void bd_resid_coef_kernel(int n, int nh, int k, const double* __restrict__ C, int ldc,
                                                            const double* __restrict__ V, const double* __restrict__ AV,
                                                            int ld, Theta16 th, const double* __restrict__ d, double guard,
                                                            double* __restrict__ R, double* __restrict__ T,
                                                            double* __restrict__ T2, double* __restrict__ npart) {
    __shared__ double cs[128][BD_HG];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int h0 = blockIdx.y * BD_HG;
    double av[BD_HG], vv[BD_HG];
#pragma unroll
    for (int h = 0; h < BD_HG; ++h) av[h] = vv[h] = 0.0;
    for (int a0 = 0; a0 < k; a0 += 128) {
        const int jt = (k - a0 < 128) ? (k - a0) : 128;
        __syncthreads();
        for (int t = threadIdx.x; t < jt * BD_HG; t += 256) {
            const int a = t / BD_HG, h = t % BD_HG;
            cs[a][h] = (h0 + h < nh) ? C[(size_t)(h0 + h) * ldc + a0 + a] : 0.0;
        }
        __syncthreads();
        if (i < n) {
#pragma unroll 4
            for (int a = 0; a < jt; ++a) {
                const double pa = AV[(size_t)(a0 + a) * ld + i], pv = V[(size_t)(a0 + a) * ld + i];
#pragma unroll
                for (int h = 0; h < BD_HG; ++h) { av[h] += cs[a][h] * pa; vv[h] += cs[a][h] * pv; }
            }
        }
    }
#pragma unroll
    for (int h = 0; h < BD_HG; ++h) {
        const int hh = h0 + h;                            // (grid.y = 4: hh < 16)
        double r = 0.0, t = 0.0;
        if (i < n && hh < nh) {
            r = av[h] - th.v[hh] * vv[h];
            t = r;
            if (d) {
                double den = d[i] - th.v[hh];
                if (fabs(den) < guard) den = (den < 0.0) ? -guard : guard;
                t = r / den;
            }
        }
        if (i < n) {
            R[(size_t)hh * ld + i] = r;
            T[(size_t)hh * ld + i] = t;
            if (T2) T2[(size_t)hh * ld + i] = t;
        }
        if (npart) bd_block_sumsq(r * r, npart + (size_t)hh * gridDim.x + blockIdx.x);
    }
}

// (T, A T) = [-S X | S] applied to [basis rows; raw block] for both panels in one launch (blockIdx.z), out of place:
//   out_z[h] = sum_{a < k} C[h ldc + a] P_z[a] + sum_{j < 16} C[h ldc + k + j] B_z[j],   h < 16,
// P_z the basis panel (V / AV), B_z the raw block kept in scratch rows (T' / A T'), out_z the 16 rows behind the basis
// (cp_z, optional: the same rows once more in scratch, the B_z of a second pass).  grid.z = 1: the V panel only.
// Workgroup (x, y, z): 256 columns, outputs 4 y .. 4 y + 3.
__global__ __launch_bounds__(256) void bd_transform2_kernel(int n, int k, const double* __restrict__ C, int ldc,
                                                            const double* __restrict__ P0, const double* __restrict__ P1,
                                                            const double* __restrict__ B0, const double* __restrict__ B1, int ldp,
                                                            double* __restrict__ out0, double* __restrict__ out1,
                                                            double* __restrict__ cp0, double* __restrict__ cp1) {
    __shared__ double cs[128][BD_HG];
    const double* __restrict__ P = blockIdx.z ? P1 : P0;
    const double* __restrict__ B = blockIdx.z ? B1 : B0;
    double* __restrict__ out = blockIdx.z ? out1 : out0;
    double* __restrict__ cp = blockIdx.z ? cp1 : cp0;              // (may be null: a second copy for a pass that follows)
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int h0 = blockIdx.y * BD_HG;
    const int kt = k + BD_NB;
    double acc[BD_HG];
#pragma unroll
    for (int h = 0; h < BD_HG; ++h) acc[h] = 0.0;
    for (int a0 = 0; a0 < kt; a0 += 128) {
        const int jt = (kt - a0 < 128) ? (kt - a0) : 128;
        __syncthreads();
        for (int t = threadIdx.x; t < jt * BD_HG; t += 256) {
            const int a = t / BD_HG, h = t % BD_HG;
            cs[a][h] = C[(size_t)(h0 + h) * ldc + a0 + a];
        }
        __syncthreads();
        if (i < n) {
#pragma unroll 4
            for (int a = 0; a < jt; ++a) {
                const int ga = a0 + a;
                const double p = (ga < k) ? P[(size_t)ga * ldp + i] : B[(size_t)(ga - k) * ldp + i];
#pragma unroll
                for (int h = 0; h < BD_HG; ++h) acc[h] += cs[a][h] * p;
            }
        }
    }
    if (i < n) {
#pragma unroll
        for (int h = 0; h < BD_HG; ++h) {
            out[(size_t)(h0 + h) * ldp + i] = acc[h];
            if (cp) cp[(size_t)(h0 + h) * ldp + i] = acc[h];
        }
    }
}

// unit vectors e_{idx[j]} as the rows of a zeroed panel (Davidson's classic start block)
struct Idx16 {
    int v[BD_NB];
};
__global__ void bd_unit_rows_kernel(double* __restrict__ T, int ld, int nt, Idx16 idx) {
    const int j = threadIdx.x;
    if (j < nt) T[(size_t)j * ld + idx.v[j]] = 1.0;
}

struct Blk {
    sella_ctx* c = nullptr;
    int n = 0, ld = 0, maxvec = 0, k = 0;
    const Mat* A = nullptr;
    int row0 = 0, world = 1, m_max = 0;
    sella_allgather_fn gather = nullptr;
    void* user = nullptr;
    const Mat *Q = nullptr, *Qt = nullptr;
    double* dP = nullptr;          // eigenvalues of P or diag(A) on the device (n), or null
    double guard = 1e-10;
    double *V = nullptr, *AV = nullptr, *Vt = nullptr;   // (maxvec + 16) rows each (Vt: restart scratch)
    double *R = nullptr, *T = nullptr, *T2 = nullptr, *MID = nullptr, *AT = nullptr;   // 16 rows each
    double *send = nullptr, *recv = nullptr;
    double* dC = nullptr;          // device coefficients, 16 x (maxvec + 16)
    size_t bytesV = 0, bytes16 = 0, bytesS = 0, bytesR = 0, bytesC = 0;
    int nmatvec = 0;
    vec G;                         // (maxvec + 16)^2 host Gram matrix V^T A V, leading dimension kcap
    int kcap = 0;
    // pipelined iteration (run_pipelined): restart targets, restart coefficients W (32 x kcap)
    double *Valt = nullptr, *AValt = nullptr, *dW = nullptr;
    long n_clean = 0, n_second = 0, n_direct = 0, n_refresh = 0;
    double anorm = 0.0;            // scale of the operator as far as known (largest |diagonal entry| / |eigenvalue of P|)
    bool early_ok = true;          // the last block's transformed A T stayed inside the error budget: apply A to the raw block again
    long n_late = 0;               // blocks whose matrix pass waited for the final T (A T exact)
    long n_lanczos = 0;            // blocks that fell back to the residuals themselves
    double av_err = 1.0;           // largest error estimate over the rows of AV, units of eps |A| (run_pipelined)
    vec err;                       // ... row by row
};

int blk_alloc(Blk& s) {
    sella_ctx* c = s.c;
    s.kcap = s.maxvec + BD_NB;
    s.bytesV = (size_t)s.kcap * s.ld * sizeof(double);
    s.bytes16 = (size_t)BD_NB * s.ld * sizeof(double);
    s.bytesC = (size_t)BD_NB * s.kcap * sizeof(double);
    double** big[3] = {&s.V, &s.AV, &s.Vt};
    for (auto p : big) {
        SCHK(dev_alloc(c, s.bytesV, p));
        HIPCHK(s_memset0(c, *p, s.bytesV));
    }
    double** small16[5] = {&s.R, &s.T, &s.T2, &s.MID, &s.AT};
    for (auto p : small16) {
        SCHK(dev_alloc(c, s.bytes16, p));
        HIPCHK(s_memset0(c, *p, s.bytes16));
    }
    SCHK(dev_alloc(c, s.bytesC, &s.dC));
    if (s.gather) {
        s.bytesS = (size_t)BD_NB * s.m_max * sizeof(double);
        s.bytesR = s.bytesS * s.world;
        SCHK(dev_alloc(c, s.bytesS, &s.send));
        SCHK(dev_alloc(c, s.bytesR, &s.recv));
        HIPCHK(s_memset0(c, s.send, s.bytesS));
    }
    s.G.assign((size_t)s.kcap * s.kcap, 0.0);
    return SELLA_OK;
}

void blk_free(Blk& s) {
    sella_ctx* c = s.c;
    if (!c) return;
    (void)hipStreamSynchronize(c->stream);
    if (s.V) dev_free(c, s.V, s.bytesV);
    if (s.AV) dev_free(c, s.AV, s.bytesV);
    if (s.Vt) dev_free(c, s.Vt, s.bytesV);
    double* small16[5] = {s.R, s.T, s.T2, s.MID, s.AT};
    for (double* p : small16)
        if (p) dev_free(c, p, s.bytes16);
    if (s.dC) dev_free(c, s.dC, s.bytesC);
    if (s.Valt) dev_free(c, s.Valt, s.bytesV);
    if (s.AValt) dev_free(c, s.AValt, s.bytesV);
    if (s.dW) dev_free(c, s.dW, 2 * s.bytesC);
    if (s.send) dev_free(c, s.send, s.bytesS);
    if (s.recv) dev_free(c, s.recv, s.bytesR);
    s.V = nullptr;
}

int combine(Blk& s, int nh, int k, const double* dC, int ldc, const double* P, double alpha, double beta, double* out,
            const double* rowscale = nullptr, const double* base = nullptr) {
    if (nh <= 0) return SELLA_OK;
    if (k <= 0) {
        if (beta == 0.0) HIPCHK(hipMemsetAsync(out, 0, (size_t)nh * s.ld * sizeof(double), s.c->stream));
        return SELLA_OK;
    }
    dim3 grid((s.n + 255) / 256, (nh + BD_HG - 1) / BD_HG);
    hipLaunchKernelGGL(bd_combine_kernel, grid, dim3(256), 0, s.c->stream, s.n, nh, k, dC, ldc, P, s.ld, alpha, beta, out,
                       s.ld, rowscale, base);
    HIPCHK(hipGetLastError());
    return SELLA_OK;
}

// host coefficients (nh x k, row h = output h) -> device buffer dC (ld kcap)
int put_coeffs(Blk& s, const vec& Ch, int nh, int k) {
    // rows of stride kcap on the device; through the pinned ring (context.hip h2d_async): no wait, and the caller's
    // temporary may go at once
    // (ONE transfer: the rows are re-strided on the host first; the columns behind k are never read)
    if (k == s.kcap) return h2d_async(s.c, s.dC, Ch.data(), (size_t)nh * k * sizeof(double));
    static thread_local vec packed;
    packed.resize((size_t)nh * s.kcap);
    for (int h = 0; h < nh; ++h) memcpy(packed.data() + (size_t)h * s.kcap, Ch.data() + (size_t)h * k, (size_t)k * sizeof(double));
    return h2d_async(s.c, s.dC, packed.data(), ((size_t)(nh - 1) * s.kcap + k) * sizeof(double));
}

// Y (nh rows) = A X^T for the 16-row panel X (rows >= nh zero): local panel product (+ all-gather)
int apply_A(Blk& s, const double* X, int nh, double* Y, bool count = true) {
    sella_ctx* c = s.c;
    if (count) s.nmatvec += nh;
    if (!s.gather) return launch_panel16(c, s.A->d, s.n, s.n, s.ld, X, nh, Y, s.ld);
    SCHK(launch_panel16(c, s.A->d, s.A->rows, s.n, s.ld, X, nh, s.send, s.m_max));
    {
        CallbackScope scope(c);                  // (stream-ordered: entering and leaving the scope does not wait)
        SCHK(scope.status);
        if (s.gather(s.user, s.send, s.recv, s.bytesS, (void*)c->stream) != 0) {
            set_error("davidson_block: all-gather callback failed");
            return SELLA_E_CALLBACK;
        }
    }
    hipLaunchKernelGGL(bd_unpack_kernel, dim3((s.n + 255) / 256, nh), dim3(256), 0, c->stream, s.recv, s.world, s.m_max, s.n,
                       Y, s.ld);
    HIPCHK(hipGetLastError());
    return SELLA_OK;
}

// T (nt rows, rows >= nt zero) <- T - V^T (V T^T): one classical Gram-Schmidt pass of the block against V[0:k)
int project_out(Blk& s, double* T, int nt, int k) {
    if (k <= 0 || nt <= 0) return SELLA_OK;
    SCHK(launch_panel16(s.c, s.V, k, s.n, s.ld, T, nt, s.dC, s.kcap));          // dC[h][a] = V_a . T_h
    return combine(s, nt, k, s.dC, s.kcap, s.V, -1.0, 1.0, T);
}

// SVQB inside the block: T (nt rows) -> orthonormal rows (count *kept), written back to T.  has_pre: dscal holds, behind
// the Gram matrix, the squared norms of the rows BEFORE they were projected against V — a row that lost more than `drop` of its norm
// there is discarded (math.pyx:112-117, eps2), as are directions whose singular value falls below `drop` of the
// largest one.
// *clean (optional) = the pass lost little: every surviving row kept at least a quarter of its squared norm in the
// projection against V and the block's Gram matrix has a condition number below 1e3 — one classical Gram-Schmidt
// pass then already leaves orthogonality at the 1e-13 level and the second pass can be skipped.
// The host half: S = Gram matrix of the nt rows (16 x 16 layout), pre = their squared norms before the projection (or
// null), skip[h] != 0 = row h does not take part (a converged pair's correction).  Ch (mk x nt, row jj = coefficients of
// new row jj) and *mk come back.
// *amp (optional): by how much the pass amplifies rounding errors of the rows it transforms — the largest norm loss in
// the projection times the square root of the block's condition number; *gain (optional): by how much it amplifies
// errors of the BASIS rows that the projection subtracts (|X_h| / |T_h - X_h V| for orthonormal V, times the same root).
int svqb_host(const double* S, const double* pre, const char* skip, int nt, double drop, vec& Ch, int* mk_out, bool* clean,
              double* amp = nullptr, double* gain = nullptr) {
    *mk_out = 0;
    if (clean) *clean = false;
    if (amp) *amp = 1.0;
    if (gain) *gain = 0.0;
    std::vector<int> live;
    int nskip = 0;
    for (int h = 0; h < nt; ++h) {
        if (skip && skip[h]) { ++nskip; continue; }
        const double d = S[h * BD_NB + h];
        if (!(d > 0.0) || d != d) continue;
        if (pre && !(d >= drop * drop * pre[h])) continue;
        live.push_back(h);
    }
    const int m = (int)live.size();
    if (m == 0) return SELLA_OK;
    vec Ss((size_t)m * m), sig(m), U((size_t)m * m), work(m), dinv(m);
    for (int a = 0; a < m; ++a) dinv[a] = 1.0 / sqrt(S[live[a] * BD_NB + live[a]]);
    for (int a = 0; a < m; ++a)
        for (int b = 0; b < m; ++b)
            Ss[(size_t)a * m + b] = 0.5 * (S[live[a] * BD_NB + live[b]] + S[live[b] * BD_NB + live[a]]) * dinv[a] * dinv[b];
    if (small::sym_eig(m, Ss.data(), m, sig.data(), U.data(), m, work.data()) != 0) {
        set_error("davidson_block: 16 x 16 eigenproblem of the block Gram matrix failed");
        return SELLA_E_NOCONV;
    }
    const double smax = sig[m - 1];
    std::vector<int> good;
    for (int j = m - 1; j >= 0; --j)                 // largest singular directions first
        if (sig[j] > drop * drop * smax) good.push_back(j);
    const int mk = (int)good.size();
    if (mk == 0) return SELLA_OK;
    if (amp) {
        double loss = 1.0;
        if (pre)
            for (int a = 0; a < m; ++a) loss = std::max(loss, pre[live[a]] / S[live[a] * BD_NB + live[a]]);
        *amp = sqrt(loss * smax / sig[good[mk - 1]]);
        if (gain) *gain = sqrt((loss - 1.0) * smax / sig[good[mk - 1]]);
    }
    if (clean && pre) {
        bool ok = mk == m && (int)live.size() == nt - nskip && sig[0] > 1e-3 * smax;
        // (a quarter of the squared norm kept: |X| <= sqrt(3) |T' - X V|.  One Gram-Schmidt pass multiplies whatever
        // non-orthogonality the basis has by |X| / |T' - X V| in the new rows — a looser rule, cancellation x conditioning
        // <= 100, was tried in round 6 and lost the basis within ten blocks of gain ~50: 2e-14 -> 1)
        for (int a = 0; a < m && ok; ++a) ok = S[live[a] * BD_NB + live[a]] >= 0.25 * pre[live[a]];
        *clean = ok;
    }
    // T_new[jj] = sum_a dinv_a U[a][j] / sqrt(sig_j) T[live_a]
    Ch.assign((size_t)mk * nt, 0.0);
    for (int jj = 0; jj < mk; ++jj) {
        const int j = good[jj];
        const double f = 1.0 / sqrt(sig[j]);
        for (int a = 0; a < m; ++a) Ch[(size_t)jj * nt + live[a]] = dinv[a] * U[(size_t)a * m + j] * f;
    }
    *mk_out = mk;
    return SELLA_OK;
}

int svqb(Blk& s, double*& T, double*& T2, int nt, bool has_pre, double drop, int* kept, bool* clean = nullptr) {
    sella_ctx* c = s.c;
    *kept = 0;
    if (clean) *clean = false;
    if (nt <= 0) return SELLA_OK;
    double* dS = c->dscal + DS_GRAM;
    SCHK(launch_panel16(c, T, nt, s.n, s.ld, T, nt, dS, BD_NB));                  // dS[h * 16 + r] = T_r . T_h
    SCHK(read_scalars(c, DS_GRAM, BD_NB * BD_NB + BD_NB));                        // ... + the 16 pre-projection norms
    vec Ch;
    int mk = 0;
    SCHK(svqb_host(c->hscal + DS_GRAM, has_pre ? c->hscal + DS_GRAM + BD_NB * BD_NB : nullptr, nullptr, nt, drop, Ch, &mk, clean));
    if (mk == 0) return SELLA_OK;
    SCHK(put_coeffs(s, Ch, mk, nt));
    HIPCHK(s_memset0(c, T2, s.bytes16));
    SCHK(combine(s, mk, nt, s.dC, s.kcap, T, 1.0, 0.0, T2));
    std::swap(T, T2);
    *kept = mk;
    return SELLA_OK;
}

// Orthonormalise the block T (nt rows) against V[0:k) and within itself; *kept rows survive (in s.T).
int orthonormalise_block(Blk& s, int nt, int k, int* kept) {
    sella_ctx* c = s.c;
    *kept = 0;
    if (nt <= 0) return SELLA_OK;
    // squared norms before the projection (read together with the first Gram matrix)
    SCHK(launch_rows_sumsq(c, s.T, s.ld, nt, s.n, c->dscal + DS_GRAM + BD_NB * BD_NB));
    SCHK(project_out(s, s.T, nt, k));
    int m1 = 0;
    bool clean = false;
    SCHK(svqb(s, s.T, s.T2, nt, true, 1e-6, &m1, &clean));
    if (m1 == 0) return SELLA_OK;
    if (clean && k > 0) { *kept = m1; return SELLA_OK; }
    // second pass: re-project (classical Gram-Schmidt twice) and re-orthonormalise; nothing is dropped here
    // unless the block collapsed to roundoff
    SCHK(project_out(s, s.T, m1, k));
    int m2 = 0;
    SCHK(svqb(s, s.T, s.T2, m1, false, 1e-6, &m2));
    *kept = m2;
    return SELLA_OK;
}

// ---- the pipelined iteration ------------------------------------------------------------------------------------------
// State shared by the two iteration drivers and the result stage.
struct BlkRun {
    vec theta, W, rn;              // Ritz values, Ritz coefficients (k x k, eigenvectors as COLUMNS), residual norms
    std::vector<char> conv;
    int iter = 0, nconv = 0, kept = 0;
    bool done = false;
};

inline double bd_now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// One block iteration used to be a chain of ~60 stream operations with three or four host waits between the matrix
// passes: Rayleigh-Ritz, residuals (wait), corrections, Gram-Schmidt against V (wait, second pass: wait), A T (the one
// n^2 pass), Gram rows (wait).  Here the chain is cut to two waits and the matrix pass covers the host's share of the
// orthonormalisation:
//   * ONE panel product over V's k + 16 rows (the raw corrections T' sit in the 16 rows behind the basis) gives the
//     projection coefficients X = T' V^T and T' T'^T together — the Gram matrix of the projected block is T' T'^T - X X^T —
//     and ONE launch with the coefficient rows [-S X | S] (S from the host's SVQB step) turns those rows into
//     T = S (T' - X V).  A `clean` pass (the projection kept at least a quarter of every row, block condition below 1e3)
//     ends there; otherwise a second pass of the same two launches follows (n_second).  Then the matrix pass A T.
//   * option bd_early_matvec (on): A is applied to the RAW block T' while the host does the SVQB step, and the same
//     coefficients give A T = S (A T' - X AV) — the matrix pass covers the host's share, 0.44 instead of 0.47 ms per
//     iteration at 3N = 12288.  The transformed A T inherits the errors of the AV rows through its coefficients: an error
//     estimate per row of AV is carried (Blk::err), the budget is 5 % of tol |theta| (4 .. 1e4 eps |A|), past it the
//     matrix pass waits for the final T again and, when the basis rows are the cause, AV and V^T AV are recomputed
//     (n_direct / n_late / n_refresh).  Measured: the error follows amp eps when corrections project onto old rows and
//     compounds x10 per block when they project onto rows just added; the estimate tracks both within 20 x
//     (SELLA_BD_CHECK).  (For half of round 6 this was off: runs from random start blocks stagnated with it — and, it
//     turned out, without it too, only less often: the thick restart kept 32 of 37 vectors when nev = 5, block = 16; see
//     the restart below.)
//   * the residual norms are read together with the Gram matrix (convergence is decided one wait later, the corrections
//     of converged pairs are dropped from the block by the SVQB coefficients), so the iteration has no wait between the
//     Rayleigh-Ritz step and the matrix pass;
//   * the thick restart runs BEFORE the residuals (the restarted basis is the Ritz basis: R_h = AV_h - theta_h V_h, one
//     launch for both panels into alternate buffers, no copies back);
//   * results the host consumes are written by their kernels into pinned host memory and waited for by polling
//     (context.hip poll_mark / poll_wait); the k x k eigenproblem runs in hostm::sym_eig_rows (row-oriented, vectorised).
// Applies when every wanted pair can be corrected in every iteration: nev <= block and the start block gave at least nev
// vectors (k >= nev from then on).  Everything else (nev > 16, narrow blocks) stays with the general driver.
int run_pipelined(Blk& s, BlkRun& r, int nev, int block, double tol, int maxiter) {
    sella_ctx* c = s.c;
    const int n = s.n, ld = s.ld, kcap = s.kcap;
    double** bV[2] = {&s.Valt, &s.AValt};
    for (auto p : bV) { SCHK(dev_alloc(c, s.bytesV, p)); HIPCHK(s_memset0(c, *p, s.bytesV)); }
    SCHK(dev_alloc(c, 2 * s.bytesC, &s.dW));
    double* const hN = c->hscal + DS_GRAM;              // |R_h|^2 (16)
    double* const hY = hN + 512;                        // projections + block Gram matrix, then Gram rows V_a . (A T)_h: 16 x kcap
    const int nblk = (n + 255) / 256;                   // workgroups of the residual kernels = partial sums per residual norm
    double* const hpart = (512 + (size_t)BD_NB * kcap + (size_t)BD_NB * nblk <= (size_t)(DS_TOTAL - DS_GRAM))
                              ? hY + (size_t)BD_NB * kcap : nullptr;
    if (512 + (size_t)BD_NB * kcap > (size_t)(DS_TOTAL - DS_GRAM)) { set_error("davidson_block: basis too large for the exchange buffer"); return SELLA_E_UNSUPPORTED; }
    static const bool timing = getenv("SELLA_BD_TIMING") != nullptr;
    // measurement / debugging aids (environment, read once): second pass always, checks of the new block against V and of the
    // transformed A T against A T, row norms of the basis per iteration, the error budget by hand
    static const bool aid_always2 = getenv("SELLA_BD_ALWAYS2") != nullptr, aid_check2 = getenv("SELLA_BD_CHECK2") != nullptr,
                      aid_check = getenv("SELLA_BD_CHECK") != nullptr, aid_norms = getenv("SELLA_BD_TRACE_NORMS") != nullptr;
    static const char* const aid_limit = getenv("SELLA_BD_LIMIT");
    static const char* const aid_keep = getenv("SELLA_BD_KEEP");
    const double* dprec = s.Q ? nullptr : s.dP;
    vec Gk, Wt, theta, Ch, pk;
    std::vector<char> skip(BD_NB, 0);
    r.conv.assign(nev, 0);
    r.rn.assign(std::max(nev, BD_NB), 0.0);
    auto wait_mark = [&]() -> int { SCHK(poll_mark(c)); return poll_wait(c); };
    auto gram_rows = [&](int k0, int nbk) {
        for (int h = 0; h < nbk; ++h)
            for (int a = 0; a < k0 + nbk; ++a) {
                const double v = hY[(size_t)h * kcap + a];
                s.G[(size_t)a * kcap + k0 + h] = v;
                s.G[(size_t)(k0 + h) * kcap + a] = v;
            }
        for (int h = 0; h < nbk; ++h)
            for (int g = 0; g < h; ++g) {
                const double v = 0.5 * (hY[(size_t)h * kcap + k0 + g] + hY[(size_t)g * kcap + k0 + h]);
                s.G[(size_t)(k0 + h) * kcap + k0 + g] = s.G[(size_t)(k0 + g) * kcap + k0 + h] = v;
            }
    };
    // rows [0, nrows) of a row-major (nrows x ncols) host block -> device rows of stride lddev, ONE transfer
    auto put_rows = [&](double* dev, int lddev, const double* h, int nrows, int ncols, int ldh) -> int {
        pk.assign((size_t)nrows * lddev, 0.0);
        for (int a = 0; a < nrows; ++a) memcpy(pk.data() + (size_t)a * lddev, h + (size_t)a * ldh, (size_t)ncols * sizeof(double));
        return h2d_async(c, dev, pk.data(), ((size_t)(nrows - 1) * lddev + ncols) * sizeof(double));
    };
    // Ritz coefficients as columns for the result stage (and the general conventions)
    auto export_W = [&](int k) {
        r.W.assign((size_t)k * k, 0.0);
        for (int j = 0; j < k; ++j)
            for (int a = 0; a < k; ++a) r.W[(size_t)a * k + j] = Wt[(size_t)j * k + a];
    };

    // ---- the start block: V[0:kept) = T, AV = A T, first Gram block ---------------------------------------------------
    {
        const int nbk = r.kept;
        HIPCHK(s_memcpy(c, s.V, s.T, (size_t)nbk * ld * sizeof(double), hipMemcpyDeviceToDevice));
        if (nbk < BD_NB) HIPCHK(s_memset0(c, s.T + (size_t)nbk * ld, (size_t)(BD_NB - nbk) * ld * sizeof(double)));
        SCHK(apply_A(s, s.T, nbk, s.AV));
        s.k = nbk;
        SCHK(launch_panel16_marked(c, s.V, s.k, n, ld, s.AV, nbk, hY, kcap));
        SCHK(poll_wait(c));
        gram_rows(0, nbk);
    }
    const int nwant = nev;
    s.early_ok = c->opt.bd_early_matvec != 0;
    s.err.assign(kcap + BD_NB, 1.0);
    vec Sg((size_t)BD_NB * BD_NB), pre(BD_NB), Cf;
    while (true) {
        int k = s.k;
        const double t0 = timing ? bd_now_us() : 0.0;
        // ---- Rayleigh-Ritz ------------------------------------------------------------------------------------------------
        Gk.resize((size_t)k * k);
        for (int a = 0; a < k; ++a) memcpy(Gk.data() + (size_t)a * k, s.G.data() + (size_t)a * kcap, (size_t)k * sizeof(double));
        theta.assign(k, 0.0);
        Wt.assign((size_t)k * k, 0.0);
        if (hostm::sym_eig_rows(k, Gk.data(), k, theta.data(), Wt.data()) != 0) {
            set_error("davidson_block: Rayleigh-Ritz eigenproblem failed");
            return SELLA_E_NOCONV;
        }
        const double t1 = timing ? bd_now_us() : 0.0;
        Theta16 th;
        for (int h = 0; h < BD_NB; ++h) th.v[h] = (h < nwant) ? theta[h] : 0.0;
        // ---- thick restart when the next block would overflow the basis: (V, AV) <- Ritz vectors, alternate buffers ---------
        // (as many new vectors as pairs were unconverged at the last verdict; should a pair fall back the basis may pass maxvec
        // by a few rows for one iteration — the panels hold maxvec + 16)
        bool ritz_basis = false;
        if (k + std::max(1, nwant - r.nconv) > s.maxvec) {
            // Vectors kept: twice the wanted pairs (at least eight beyond them), leaving room for the next block.  The
            // general loop's max(nev + block, 2 block) presumes `block` new vectors per iteration; here an iteration adds at
            // most nev <= block, and with nev = 5, block = 16 that rule keeps 32 of 37 and restarts EVERY iteration: the
            // iteration then crawls (theta_0 0.9 -> 0.65 in 70 iterations with residuals ~2) until enough pairs converge for
            // two blocks to fit between restarts — or never does (what looked like a defect of bd_early_matvec in round 6).
            int keep = std::min(k, std::max(nev, std::min(std::max(2 * nev, nev + 8), s.maxvec - nev)));
            if (aid_keep) keep = std::min(k, std::max(nev, atoi(aid_keep)));                  // (measurement aid)
            SCHK(put_rows(s.dW, kcap, Wt.data(), keep, k, k));
            hipLaunchKernelGGL(bd_combine_pair_kernel, dim3((n + 255) / 256, (keep + BD_HG - 1) / BD_HG, 2), dim3(256), 0, c->stream, n,
                               keep, k, s.dW, kcap, s.V, s.AV, ld, s.Valt, s.AValt, ld);
            HIPCHK(hipGetLastError());
            std::swap(s.V, s.Valt);
            std::swap(s.AV, s.AValt);
            std::fill(s.G.begin(), s.G.end(), 0.0);
            for (int a = 0; a < keep; ++a) s.G[(size_t)a * kcap + a] = theta[a];
            {
                vec en(keep);
                for (int a = 0; a < keep; ++a) {
                    double acc = 0.0;
                    for (int b = 0; b < k; ++b) acc += Wt[(size_t)a * k + b] * Wt[(size_t)a * k + b] * s.err[b] * s.err[b];
                    en[a] = std::max(1.0, sqrt(acc));
                }
                for (int a = 0; a < keep; ++a) s.err[a] = en[a];
            }
            theta.resize(keep);
            Wt.assign((size_t)keep * keep, 0.0);
            for (int a = 0; a < keep; ++a) Wt[(size_t)a * keep + a] = 1.0;
            s.k = k = keep;
            ritz_basis = true;
        }
        r.theta = theta;
        // ---- residuals (16 rows of their own) and raw corrections T' — into the 16 rows BEHIND the basis, rows [nwant, 16)
        // zero: one panel product over V's k + 16 rows then gives the projections and the block's Gram matrix together ------
        double* Ts = s.V + (size_t)k * ld;
        double* ATs = s.AV + (size_t)k * ld;
        const int kt = k + BD_NB;
        // (T' a second time in scratch rows: the final transformation reads it there and writes the rows behind the basis;
        // the squared residual norms as per-workgroup partial sums straight into pinned host memory)
        double* T2 = s.Q ? nullptr : s.T;
        if (ritz_basis) {
            hipLaunchKernelGGL(bd_resid_ritz_kernel, dim3(nblk, BD_NB), dim3(256), 0, c->stream, n, nwant, s.V, s.AV, ld, th,
                               dprec, s.guard, s.R, Ts, T2, hpart);
        } else {
            SCHK(put_rows(s.dC, kcap, Wt.data(), nwant, k, k));
            hipLaunchKernelGGL(bd_resid_coef_kernel, dim3(nblk, BD_NB / BD_HG), dim3(256), 0, c->stream, n, nwant, k, s.dC,
                               kcap, s.V, s.AV, ld, th, dprec, s.guard, s.R, Ts, T2, hpart);
        }
        HIPCHK(hipGetLastError());
        if (s.Q) {
            // T' = Q diag(1 / (d - theta_h)) Q^T R   (eigensolvers.py:119-121 'gd', in the eigenbasis of P)
            SCHK(launch_panel16(c, s.Qt->d, n, n, ld, s.R, nwant, s.MID, ld));
            hipLaunchKernelGGL(bd_shift_scale_kernel, dim3((n + 255) / 256, nwant), dim3(256), 0, c->stream, n, nwant, s.MID, ld, s.dP,
                               th, s.guard);
            HIPCHK(hipGetLastError());
            SCHK(launch_panel16(c, s.Q->d, n, n, ld, s.MID, nwant, Ts, ld));
            HIPCHK(s_memcpy(c, s.T, Ts, s.bytes16, hipMemcpyDeviceToDevice));
        }
        if (!hpart) SCHK(launch_rows_sumsq(c, s.R, ld, nwant, n, hN));
        auto verdict = [&]() {
            r.nconv = 0;
            for (int h = 0; h < nwant; ++h) {
                if (hpart) {
                    double acc = 0.0;
                    for (int b = 0; b < nblk; ++b) acc += hpart[(size_t)h * nblk + b];
                    hN[h] = acc;
                }
                r.rn[h] = sqrt(hN[h]);
                const bool ok = r.rn[h] <= tol * std::max(fabs(theta[h]), 1e-300);
                r.conv[h] = ok ? 1 : 0;
                skip[h] = r.conv[h];
                r.nconv += ok ? 1 : 0;
            }
        };
        if (r.iter >= maxiter) {                        // out of iterations: the norms of the final pairs, nothing else
            SCHK(wait_mark());
            verdict();
            r.done = (r.nconv == nev);
            export_W(k);
            break;
        }
        // hY[h][a] = V_a . T'_h (a < k: the projection coefficients X) and T'_{a-k} . T'_h (the block's Gram matrix): marked
        SCHK(launch_panel16_marked(c, s.V, kt, n, ld, Ts, nwant, hY, kcap));
        // ---- the matrix pass on the raw block (into scratch rows: the final transformation writes the rows behind AV) — unless
        // the last block's transformed A T left the error budget: then A waits for the final T (27 us later, exact) --------
        const bool early = s.early_ok;
        if (early) SCHK(apply_A(s, Ts, nwant, s.AT, false));     // (counted below: the rows of unconverged pairs)
        const double t2 = timing ? bd_now_us() : 0.0;
        SCHK(poll_wait(c));
        const double t3 = timing ? bd_now_us() : 0.0;
        verdict();
        if (r.nconv == nev) { r.done = true; export_W(k); break; }
        s.nmatvec += nwant - r.nconv;
        // ---- the host's share: SVQB on the Gram matrix of the projected block, (T' - X V)(T' - X V)^T = T' T'^T - X X^T for an
        // orthonormal V (a projection that cancels digits here is not `clean`, and the second pass measures its Gram matrix
        // afresh); coefficient rows [-S X | S] over the k + 16 rows transform T' and A T' where they stand ----------------
        static const bool force2 = getenv("SELLA_BD_FORCE2") != nullptr;       // (test aid: second pass and A T from T, always)
        int mk = nwant;
        bool clean = false, direct = false, stop = false, refreshed = false, fragile = false;
        double amp = 1.0, gain = 0.0, inherited = 0.0;
        // Error budget of the transformed A T, ROW BY ROW (units of eps |A|; `err`: what every row of AV carries).  Row h of
        // the new block inherits sqrt(sum_a c_ha^2 err_a^2) through its coefficients c = [-S X | S] over the basis rows and
        // adds the fresh product's error times the cancellation (`amp`).  Measured (tools/block_iter_hash.py, SELLA_BD_CHECK):
        // with corrections that project onto OLD, accurate rows the error of A T follows amp eps for forty iterations whatever
        // |X| is; it compounds — exponentially — only when a block projects onto the rows just added (a start block of random
        // vectors under a diagonal preconditioner).  The per-row sum tells the two apart; a scalar bound |X| max(err) did not.
        double errb[BD_NB];
        bool lanczos = false;
        for (;;) {
        for (int pass = 0; pass < 2; ++pass) {
            const int nt = mk;
            for (int h = 0; h < nt; ++h) {
                pre[h] = hY[(size_t)h * kcap + k + h];
                for (int g = 0; g < nt; ++g) {
                    double xg = 0.0;
                    for (int a = 0; a < k; ++a) xg += hY[(size_t)h * kcap + a] * hY[(size_t)g * kcap + a];
                    Sg[(size_t)h * BD_NB + g] = hY[(size_t)h * kcap + k + g] - xg;
                }
            }
            int m2 = 0;
            if (pass == 0) {
                // T' T'^T - X X^T loses its digits when the projection cancels more than ~eight of them (a correction that lies
                // in the basis but for roundoff: spikes of a diagonal preconditioner at a start from random vectors): such a
                // row must not be judged — or dropped — by that difference.  The first pass then only projects and scales
                // (T_h = (T'_h - X_h V) / |T'_h|), and the second pass measures the block as it really is.
                fragile = false;
                for (int h = 0; h < nt; ++h)
                    if (!skip[h] && pre[h] > 0.0 && !(Sg[(size_t)h * BD_NB + h] >= 1e-8 * pre[h])) fragile = true;
                if (fragile) {
                    Ch.assign((size_t)nt * nt, 0.0);
                    for (int h = 0; h < nt; ++h)
                        if (!skip[h] && pre[h] > 0.0 && pre[h] == pre[h]) {
                            Ch[(size_t)m2 * nt + h] = 1.0 / sqrt(pre[h]);
                            ++m2;
                        }
                    clean = false;
                    amp = 1e8;
                    gain = 1e8;
                } else {
                    SCHK(svqb_host(Sg.data(), pre.data(), skip.data(), nt, 1e-6, Ch, &m2, &clean, &amp, &gain));
                }
            } else if (fragile) {
                // (rows of norm 1 / loss after the scaled projection: the drop rule of the first pass, applied to what was measured)
                for (int h = 0; h < nt; ++h) pre[h] = 1.0;
                SCHK(svqb_host(Sg.data(), pre.data(), nullptr, nt, 1e-6, Ch, &m2, nullptr));
            } else {
                SCHK(svqb_host(Sg.data(), nullptr, nullptr, nt, 1e-6, Ch, &m2, nullptr));
            }
            if (m2 == 0) { stop = true; break; }
            if (pass == 0 && (force2 || aid_always2)) clean = false;
            Cf.assign((size_t)BD_NB * kcap, 0.0);
            for (int jj = 0; jj < m2; ++jj) {
                double* cf = Cf.data() + (size_t)jj * kcap;
                for (int j = 0; j < nt; ++j) {
                    const double sj = Ch[(size_t)jj * nt + j];
                    if (sj == 0.0) continue;
                    cf[k + j] = sj;
                    const double* xj = hY + (size_t)j * kcap;
                    for (int a = 0; a < k; ++a) cf[a] -= sj * xj[a];
                }
            }
            SCHK(h2d_async(c, s.dC, Cf.data(), ((size_t)(BD_NB - 1) * kcap + kt) * sizeof(double)));
            // (both panels only with the early matrix pass; a block that is not clean leaves a second copy of its rows in
            // scratch, which the second pass reads while it writes the rows behind the basis)
            const unsigned nz = early ? 2u : 1u;
            if (pass == 0)
                hipLaunchKernelGGL(bd_transform2_kernel, dim3(nblk, BD_NB / BD_HG, nz), dim3(256), 0, c->stream, n, k, s.dC, kcap, s.V, s.AV,
                                   s.T, s.AT, ld, Ts, ATs, clean ? nullptr : s.T2, clean ? nullptr : s.MID);
            else
                hipLaunchKernelGGL(bd_transform2_kernel, dim3(nblk, BD_NB / BD_HG, nz), dim3(256), 0, c->stream, n, k, s.dC, kcap, s.V, s.AV,
                                   s.T2, s.MID, ld, Ts, ATs, (double*)nullptr, (double*)nullptr);
            HIPCHK(hipGetLastError());
            mk = m2;
            {
                double eb[BD_NB];
                for (int jj = 0; jj < BD_NB; ++jj) {
                    const double* cf = Cf.data() + (size_t)jj * kcap;
                    double acc = 0.0, fresh = 0.0;
                    for (int a = 0; a < k; ++a) acc += cf[a] * cf[a] * s.err[a] * s.err[a];
                    if (pass == 0) {
                        inherited = std::max(inherited, sqrt(acc));
                        fresh = amp;
                    } else {
                        for (int j = 0; j < nt; ++j) acc += cf[k + j] * cf[k + j] * errb[j] * errb[j];
                    }
                    eb[jj] = (jj < m2) ? sqrt(acc) + fresh : 0.0;
                }
                for (int jj = 0; jj < BD_NB; ++jj) errb[jj] = eb[jj];
            }
            if (pass == 0) {
                if (clean) { ++s.n_clean; break; }
                ++s.n_second;
                SCHK(launch_panel16_marked(c, s.V, kt, n, ld, Ts, mk, hY, kcap));
                SCHK(poll_wait(c));
            }
        }
        if (stop && !lanczos && (dprec || s.Q)) {
            // Every preconditioned correction lies in the basis to working precision (a Ritz value inside the spectrum of a
            // diagonal preconditioner: the correction is a spike along a vector the basis already holds).  The reference
            // falls back to the residual itself there ("Do Lanczos instead", sella/eigensolvers.py:93-95): so does the block.
            lanczos = true;
            stop = clean = direct = fragile = false;
            mk = nwant;
            amp = 1.0;
            gain = inherited = 0.0;
            HIPCHK(s_memcpy(c, Ts, s.R, s.bytes16, hipMemcpyDeviceToDevice));
            HIPCHK(s_memcpy(c, s.T, s.R, s.bytes16, hipMemcpyDeviceToDevice));
            SCHK(launch_panel16_marked(c, s.V, kt, n, ld, Ts, nwant, hY, kcap));
            if (early) SCHK(apply_A(s, Ts, nwant, s.AT, false));
            SCHK(poll_wait(c));
            ++s.n_lanczos;
            continue;
        }
        break;
        }
        if (stop) { ++r.iter; export_W(k); break; }              // the corrections are in span(V): nothing left to add
        if (aid_check2) {
            SCHK(launch_panel16_marked(c, s.V, kt, n, ld, Ts, mk, hY, kcap));
            SCHK(poll_wait(c));
            double ov = 0.0, ot = 0.0, vv = 0.0;
            for (int h = 0; h < mk; ++h) {
                for (int a = 0; a < k; ++a) ov = std::max(ov, fabs(hY[(size_t)h * kcap + a]));
                for (int g = 0; g < mk; ++g) ot = std::max(ot, fabs(hY[(size_t)h * kcap + k + g] - (g == h ? 1.0 : 0.0)));
            }
            fprintf(stderr, "    CHECK2 iteration %d: new block: max |V . T| %.2e, max |T T^T - I| %.2e (rows %d, clean %d, amp %.2e)", r.iter, ov, ot, mk, (int)clean, amp);
            for (int h = 0; h < nwant; ++h) fprintf(stderr, " [pre %.2e S %.2e skip %d]", pre[h], Sg[(size_t)h * BD_NB + h], (int)skip[h]);
            fprintf(stderr, "\n");
            (void)vv;
        }
        // The budget: an error delta in AV puts a floor of about delta under the residual norms, and the caller asks for
        // tol |theta|: 5 % of that, between 4 and 1e4 units (below 4 no path delivers).  Past it the block's A T is
        // computed from T itself (one more matrix pass if the raw block's was already taken), and the next block's matrix
        // pass waits for its final T; when it is what the basis rows carry that put it there, AV = A V is recomputed too.
        double errmax = 0.0;
        for (int h = 0; h < mk; ++h) errmax = std::max(errmax, errb[h]);
        double anorm = s.anorm, thref = 0.0;
        for (int a = 0; a < k; ++a) anorm = std::max(anorm, fabs(theta[a]));
        for (int h = 0; h < nwant; ++h) { const double th_h = std::max(fabs(theta[h]), 1e-2 * anorm); thref = (h == 0) ? th_h : std::min(thref, th_h); }
        double limit = std::min(1e4, std::max(4.0, 0.05 * tol * thref / (2.220446049250313e-16 * std::max(anorm, 1e-300))));
        if (aid_limit) limit = atof(aid_limit);
        direct = !(errmax <= limit) || force2 || !early;
        s.early_ok = c->opt.bd_early_matvec && errmax <= 0.5 * limit;
        static const bool never_direct = getenv("SELLA_BD_NEVER_DIRECT") != nullptr;      // (measurement aid)
        if (never_direct) direct = false;
        if (aid_check) {
            // measurement aid: the transformed A T against A T computed from T itself
            SCHK(apply_A(s, Ts, mk, s.MID, false));
            vec a1((size_t)n * mk), a2((size_t)n * mk);
            SCHK(download_panel(c, ATs, ld, n, mk, a1.data()));
            SCHK(download_panel(c, s.MID, ld, n, mk, a2.data()));
            double e = 0.0, m = 0.0;
            for (size_t q = 0; q < a1.size(); ++q) { e = std::max(e, fabs(a1[q] - a2[q])); m = std::max(m, fabs(a2[q])); }
            fprintf(stderr, "    CHECK iteration %d: transformed A T against A T: max diff %.3e (max entry %.3e), gain %.2e amp %.2e bound %.2e%s\n", r.iter, e, m,
                    gain, amp, errmax, direct ? " (direct)" : "");
        }
        if (direct) {
            if (early) ++s.n_direct; else ++s.n_late;
            SCHK(apply_A(s, Ts, mk, ATs, false));
            for (int h = 0; h < BD_NB; ++h) errb[h] = 1.0;
            if (inherited > 0.5 * limit) {
                // the basis rows are near the limit themselves: AV = A V afresh, 16 rows per matrix pass
                for (int a0 = 0; a0 < k; a0 += BD_NB) {
                    // (a last chunk shorter than 16 rows multiplies rows of the new block too: their products are what stands there)
                    SCHK(apply_A(s, s.V + (size_t)a0 * ld, std::min(BD_NB, kt - a0), s.AV + (size_t)a0 * ld, false));
                }
                ++s.n_refresh;
                refreshed = true;
                for (int a = 0; a < k; ++a) s.err[a] = 1.0;
            }
        }
        for (int h = 0; h < mk; ++h) s.err[k + h] = std::max(1.0, errb[h]);
        s.av_err = 0.0;
        for (int a = 0; a < k + mk; ++a) s.av_err = std::max(s.av_err, s.err[a]);
        // ---- new rows of the projected matrix ------------------------------------------------------------------------------------
        SCHK(launch_panel16_marked(c, s.V, k + mk, n, ld, ATs, mk, hY, kcap));
        SCHK(poll_wait(c));
        gram_rows(k, mk);
        if (refreshed) {
            // AV was recomputed: so is the projected matrix V^T AV (what it held was measured against the old rows)
            const int kn = k + mk;
            for (int a0 = 0; a0 < k; a0 += BD_NB) {
                const int nh = std::min(BD_NB, kn - a0);
                SCHK(launch_panel16_marked(c, s.V, kn, n, ld, s.AV + (size_t)a0 * ld, nh, hY, kcap));
                SCHK(poll_wait(c));
                for (int h = 0; h < nh; ++h)
                    for (int a = 0; a < kn; ++a) s.G[(size_t)a * kcap + a0 + h] = hY[(size_t)h * kcap + a];
            }
            for (int a = 0; a < kn; ++a)
                for (int b = 0; b < a; ++b) {
                    const double v = 0.5 * (s.G[(size_t)a * kcap + b] + s.G[(size_t)b * kcap + a]);
                    s.G[(size_t)a * kcap + b] = s.G[(size_t)b * kcap + a] = v;
                }
        }
        s.k = k + mk;
        ++r.iter;
        if (aid_norms) {
            SCHK(launch_rows_sumsq(c, s.V, ld, s.k, n, c->dscal + DS_MISC));
            SCHK(read_scalars(c, DS_MISC, s.k));
            double mn = 1e300; int arg = -1;
            for (int a = 0; a < s.k; ++a) if (c->hscal[DS_MISC + a] < mn) { mn = c->hscal[DS_MISC + a]; arg = a; }
            fprintf(stderr, "    NORMS iteration %d: k %d -> %d (restart %d, early %d, clean %d, direct %d, refreshed %d), smallest |V_a|^2 = %.3e at row %d\n", r.iter, k, s.k,
                    (int)ritz_basis, (int)early, (int)clean, (int)direct, (int)refreshed, mn, arg);
        }
        if (timing) {
            fprintf(stderr, "block davidson (pipelined): k = %d: Rayleigh-Ritz %.1f us, queueing %.1f us, wait behind the projection %.1f us, "
                            "SVQB + final stage + wait for the Gram rows %.1f us%s\n", k, t1 - t0, t2 - t1, t3 - t2, bd_now_us() - t3,
                    clean ? "" : " (second pass)");
            fprintf(stderr, "    theta0 %.6e rn0 %.2e nconv %d new rows %d amplification %.2e gain %.2e%s\n", theta[0], r.rn[0], r.nconv, mk,
                    amp, gain, direct ? " (A T recomputed)" : "");
        }
    }
    if (const char* dump = getenv("SELLA_BD_DUMP")) {          // (debugging aid: the basis and its images as raw doubles)
        const int k = s.k;
        vec hv((size_t)n * k), hav((size_t)n * k);
        SCHK(download_panel(c, s.V, ld, n, k, hv.data()));
        SCHK(download_panel(c, s.AV, ld, n, k, hav.data()));
        if (FILE* f = fopen(dump, "wb")) {
            fwrite(&n, sizeof(int), 1, f); fwrite(&k, sizeof(int), 1, f);
            fwrite(hv.data(), sizeof(double), hv.size(), f); fwrite(hav.data(), sizeof(double), hav.size(), f);
            for (int a = 0; a < k; ++a) fwrite(s.G.data() + (size_t)a * kcap, sizeof(double), k, f);
            fclose(f);
        }
    }
    return SELLA_OK;
}

}  // namespace
}  // namespace sella

using namespace sella;

extern "C" int sella_davidson_block(sella_ctx* c, sella_mat hA, int n, int row0, int world,
                                    sella_allgather_fn gather, void* user, sella_mat hPvecs, sella_mat hPvecsT,
                                    const double* pevals, const double* diag, const double* V0, int nv0, int nev,
                                    int block, int maxvec, double tol, int maxiter, double* lams_out, double* V_out,
                                    double* res_out, int* niter_out, int* nmatvec_out, int* nconv_out) {
    if (!c || n <= 0 || nev <= 0 || !lams_out || !V_out) {
        set_error("davidson_block: invalid arguments");
        return SELLA_E_INVALID;
    }
    if (block <= 0 || block > BD_NB) block = BD_NB;
    if (nev > n) nev = n;
    if (block > n) block = n;
    // default basis limit: the Ritz vectors kept at a restart plus TWO new blocks.  One block of room (nev + 2 block, the
    // default until round 6) means a thick restart in every iteration: at 3N = 12288, nev = block = 16, tol 1e-9 that takes
    // 439 iterations / 185 ms to converge against 220 / 106 ms with a second block between restarts, and 149 / 108 ms with
    // four (tools/block_restart_sweep.py; the k x k Rayleigh-Ritz problem on the host is O(k^3): 0.1 ms at k = 48, 0.25 at 64)
    if (maxvec <= 0) maxvec = std::max(nev + 3 * block, 24);
    if (maxvec < nev + 2 * block) maxvec = nev + 2 * block;
    if (maxvec > n) maxvec = n;
    if (maxvec + BD_NB > 2048) {
        set_error("davidson_block: basis of %d vectors exceeds the exchange layout (2032)", maxvec);
        return SELLA_E_UNSUPPORTED;
    }
    if (!(tol > 0.0)) tol = 1e-8;
    if (maxiter <= 0) maxiter = 1000;
    Blk s;
    s.c = c;
    s.n = n;
    s.A = mat_get(c, hA);
    if (!s.A) return SELLA_E_INVALID;
    s.ld = s.A->ld;
    if (s.A->cols != n) { set_error("davidson_block: the operator must have %d columns", n); return SELLA_E_INVALID; }
    s.gather = gather;
    s.user = user;
    s.row0 = row0;
    s.world = (world > 0) ? world : 1;
    if (gather) {
        s.m_max = (n + s.world - 1) / s.world;
        if (s.A->rows > s.m_max || row0 < 0 || row0 + s.A->rows > n || row0 % s.m_max != 0) {
            set_error("davidson_block: row panel [%d, %d) does not fit the sharding %d x %d", row0, row0 + s.A->rows,
                      s.world, s.m_max);
            return SELLA_E_INVALID;
        }
    } else if (s.A->rows != n || row0 != 0) {
        set_error("davidson_block: without an all-gather the operator must hold all %d rows", n);
        return SELLA_E_INVALID;
    }
    if (hPvecs != SELLA_NO_MAT) {
        s.Q = mat_get(c, hPvecs);
        s.Qt = mat_get(c, hPvecsT);
        if (!s.Q || !s.Qt || !pevals || s.Q->rows != n || s.Q->cols != n || s.Qt->rows != n || s.Qt->cols != n) {
            set_error("davidson_block: the preconditioner needs Pvecs, PvecsT (%d x %d) and pevals", n, n);
            return SELLA_E_INVALID;
        }
    }
    s.maxvec = maxvec;
    const bool stage_timing = getenv("SELLA_BD_TIMING") != nullptr;
    const double ts0 = bd_now_us();
    int st = blk_alloc(s);
    auto fail = [&](int code) { blk_free(s); return code; };
#define BCHK(expr) do { int s__ = (expr); if (s__ != SELLA_OK) return fail(s__); } while (0)
#define BHIP(expr) do { if ((expr) != hipSuccess) { set_error("%s failed (%s:%d)", #expr, __FILE__, __LINE__); return fail(SELLA_E_HIP); } } while (0)
    if (st != SELLA_OK) return fail(st);
    const double* pdiag = s.Q ? pevals : diag;
    if (pdiag) {
        BCHK(scratch_get(c, SCR_C, (size_t)s.ld * sizeof(double), &s.dP));
        BCHK(h2d_async(c, s.dP, pdiag, (size_t)n * sizeof(double)));
        double amax = 0.0;
        for (int i = 0; i < n; ++i) amax = std::max(amax, fabs(pdiag[i]));
        s.guard = std::max(1e-300, 1e-10 * amax);
        s.anorm = amax;
    }

    // ---- start block ------------------------------------------------------------------------------------
    int nt = 0;
    {
        vec X0;
        if (V0 && nv0 > 0) {
            nt = std::min(nv0, BD_NB);
            X0.assign((size_t)n * nt, 0.0);
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < nt; ++j) X0[(size_t)i * nt + j] = V0[(size_t)i * nv0 + j];
        } else if (pdiag && !s.Q) {
            // unit vectors at the `nt` smallest diagonal entries (Davidson's classic start), set on the device: s.T is zero
            nt = std::min(block, n);
            std::vector<int> idx(n);
            for (int i = 0; i < n; ++i) idx[i] = i;
            std::partial_sort(idx.begin(), idx.begin() + nt, idx.end(),
                              [&](int a, int b) { return pdiag[a] < pdiag[b] || (pdiag[a] == pdiag[b] && a < b); });
            Idx16 ix;
            for (int j = 0; j < BD_NB; ++j) ix.v[j] = (j < nt) ? idx[j] : 0;
            hipLaunchKernelGGL(bd_unit_rows_kernel, dim3(1), dim3(64), 0, c->stream, s.T, s.ld, nt, ix);
            BHIP(hipGetLastError());
        } else if (s.Q) {
            nt = std::min(block, n);          // (the lowest eigenvectors of P, copied below)
        } else {
            nt = std::min(block, n);
            X0.assign((size_t)n * nt, 0.0);
            {
                unsigned long long lcg = 0x9E3779B97F4A7C15ull;          // deterministic pseudo-random start
                for (size_t e = 0; e < X0.size(); ++e) {
                    lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
                    X0[e] = (double)(lcg >> 11) * (1.0 / 9007199254740992.0) - 0.5;
                }
            }
        }
        if (!X0.empty()) BCHK(upload_panel(c, X0.data(), n, nt, s.T, s.ld));
        if (s.Q && !(V0 && nv0 > 0)) {
            // with an eigenbasis preconditioner the natural start is its lowest eigenvectors
            BHIP(hipMemcpy2DAsync(s.T, (size_t)s.ld * sizeof(double), s.Qt->d, (size_t)s.Qt->ld * sizeof(double),
                                  (size_t)n * sizeof(double), nt, hipMemcpyDeviceToDevice, c->stream));
        }
    }
    int kept = 0;
    const double ts1 = bd_now_us();
    BCHK(orthonormalise_block(s, nt, 0, &kept));
    const double ts2 = bd_now_us();
    if (kept == 0) { set_error("davidson_block: the start block is numerically zero"); return fail(SELLA_E_INVALID); }

    BlkRun run;
    vec &theta = run.theta, &W = run.W, &rn = run.rn;
    std::vector<char>& conv = run.conv;
    int &iter = run.iter, &nconv = run.nconv;
    bool& done = run.done;
    vec Gk, work, Ch;
    rn.assign(std::max(nev, BD_NB), 0.0);
    conv.assign(nev, 0);
    unsigned long long lcg = 0xD1B54A32D192ED03ull;
    // every wanted pair corrected in every iteration, two waits per iteration, the matrix pass over the host's share of the
    // orthonormalisation: run_pipelined above; narrow blocks and nev > block stay with the general loop below
    const bool pipelined = c->opt.bd_pipeline && nev <= block && kept >= nev;
    if (pipelined) {
        run.kept = kept;
        BCHK(run_pipelined(s, run, nev, block, tol, maxiter));
        if (getenv("SELLA_BD_TIMING"))
            fprintf(stderr, "block davidson (pipelined): %d iterations, %ld clean blocks, %ld with a second pass, %ld blocks with A T recomputed, %ld with the matrix pass behind the final T, %ld refreshes of AV, %ld blocks of plain residuals, error estimate %.1e eps |A|\n", iter, s.n_clean, s.n_second, s.n_direct, s.n_late, s.n_refresh, s.n_lanczos, s.av_err);
    }
    while (!pipelined) {
        // ---- append the orthonormal block in s.T: V, AV, Gram rows ------------------------------------------
        const int k0 = s.k, nbk = kept;
        BHIP(s_memcpy(c, s.V + (size_t)k0 * s.ld, s.T, (size_t)nbk * s.ld * sizeof(double), hipMemcpyDeviceToDevice));
        if (nbk < BD_NB)      // rows [nbk, 16) of the panel operand must be zero
            BHIP(s_memset0(c, s.T + (size_t)nbk * s.ld, (size_t)(BD_NB - nbk) * s.ld * sizeof(double)));
        BCHK(apply_A(s, s.T, nbk, s.AT));
        BHIP(s_memcpy(c, s.AV + (size_t)k0 * s.ld, s.AT, (size_t)nbk * s.ld * sizeof(double), hipMemcpyDeviceToDevice));
        s.k = k0 + nbk;
        {
            double* dY = c->dscal + DS_GRAM;                                   // dY[h * kcap + a] = V_a . (A T)_h
            if (nbk < BD_NB)
                BHIP(s_memset0(c, s.AT + (size_t)nbk * s.ld, (size_t)(BD_NB - nbk) * s.ld * sizeof(double)));
            BCHK(launch_panel16(c, s.V, s.k, n, s.ld, s.AT, nbk, dY, s.kcap));
            BCHK(read_scalars(c, DS_GRAM, BD_NB * s.kcap));
            const double* Y = c->hscal + DS_GRAM;
            for (int h = 0; h < nbk; ++h)
                for (int a = 0; a < s.k; ++a) {
                    const double v = Y[(size_t)h * s.kcap + a];
                    s.G[(size_t)a * s.kcap + k0 + h] = v;
                    s.G[(size_t)(k0 + h) * s.kcap + a] = v;
                }
            for (int h = 0; h < nbk; ++h)                                        // exact symmetry inside the new block
                for (int g = 0; g < h; ++g) {
                    const double v = 0.5 * (s.G[(size_t)(k0 + h) * s.kcap + k0 + g] + s.G[(size_t)(k0 + g) * s.kcap + k0 + h]);
                    s.G[(size_t)(k0 + h) * s.kcap + k0 + g] = s.G[(size_t)(k0 + g) * s.kcap + k0 + h] = v;
                }
        }
        // ---- Rayleigh-Ritz ------------------------------------------------------------------------------------
        const int k = s.k;
        theta.assign(k, 0.0);
        Gk.resize((size_t)k * k);
        for (int a = 0; a < k; ++a)
            for (int b = 0; b < k; ++b) Gk[(size_t)a * k + b] = s.G[(size_t)a * s.kcap + b];
        W.assign((size_t)k * k, 0.0);
        work.resize(k);
        static const bool rr_timing = getenv("SELLA_BD_TIMING") != nullptr;      // host Rayleigh-Ritz, per iteration (stderr)
        const auto trr0 = std::chrono::steady_clock::now();
        if (small::sym_eig(k, Gk.data(), k, theta.data(), W.data(), k, work.data()) != 0) {
            set_error("davidson_block: Rayleigh-Ritz eigenproblem failed");
            return fail(SELLA_E_NOCONV);
        }
        if (rr_timing)
            fprintf(stderr, "block davidson: host Rayleigh-Ritz k = %d: %.1f us\n", k,
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - trr0).count());
        // ---- residuals of the lowest nev Ritz pairs, 16 at a time; the first chunk with unconverged pairs feeds
        // the correction block s.T (its rows are copied out before s.R is reused) --------------------------------
        const int nwant = std::min(nev, k);
        int na = 0;
        Theta16 th;
        for (int h = 0; h < BD_NB; ++h) th.v[h] = 0.0;
        BHIP(s_memset0(c, s.T, s.bytes16));
        nconv = 0;
        for (int j0 = 0; j0 < nwant; j0 += BD_NB) {
            const int nh = std::min(BD_NB, nwant - j0);
            Ch.assign((size_t)nh * k, 0.0);
            for (int h = 0; h < nh; ++h)
                for (int a = 0; a < k; ++a) Ch[(size_t)h * k + a] = W[(size_t)a * k + j0 + h];
            BCHK(put_coeffs(s, Ch, nh, k));
            BCHK(combine(s, nh, k, s.dC, s.kcap, s.AV, 1.0, 0.0, s.R));
            for (int h = 0; h < nh; ++h)
                for (int a = 0; a < k; ++a) Ch[(size_t)h * k + a] *= -theta[j0 + h];
            BCHK(put_coeffs(s, Ch, nh, k));
            BCHK(combine(s, nh, k, s.dC, s.kcap, s.V, 1.0, 1.0, s.R));
            BCHK(launch_rows_sumsq(c, s.R, s.ld, nh, n, c->dscal + DS_MISC));
            BCHK(read_scalars(c, DS_MISC, nh));
            const bool feed = (na == 0);
            int run_src = -1, run_dst = 0, run_len = 0;              // consecutive residual rows travel as one copy
            auto flush_run = [&]() -> int {
                if (run_len > 0)
                    HIPCHK(s_memcpy(c, s.T + (size_t)run_dst * s.ld, s.R + (size_t)run_src * s.ld,
                                          (size_t)run_len * s.ld * sizeof(double), hipMemcpyDeviceToDevice));
                run_len = 0;
                return SELLA_OK;
            };
            for (int h = 0; h < nh; ++h) {
                rn[j0 + h] = sqrt(c->hscal[DS_MISC + h]);
                const bool ok = rn[j0 + h] <= tol * std::max(fabs(theta[j0 + h]), 1e-300);
                conv[j0 + h] = ok ? 1 : 0;
                nconv += ok ? 1 : 0;
                if (!ok && feed && na < block) {
                    if (run_len > 0 && h == run_src + run_len) {
                        ++run_len;
                    } else {
                        BCHK(flush_run());
                        run_src = h; run_dst = na; run_len = 1;
                    }
                    th.v[na++] = theta[j0 + h];
                }
            }
            BCHK(flush_run());
        }
        if (nconv == nwant && nwant == nev) { done = true; break; }
        if (iter >= maxiter) break;
        bool precondition = true;
        if (na == 0) {
            // every pair the basis can offer is converged but there are fewer than nev of them (a start block
            // smaller than nev): expand with deterministic pseudo-random directions
            na = std::min(block, nev - nwant);
            vec rnd((size_t)n * na);
            for (auto& x : rnd) {
                lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
                x = (double)(lcg >> 11) * (1.0 / 9007199254740992.0) - 0.5;
            }
            BCHK(upload_panel(c, rnd.data(), n, na, s.T, s.ld));
            precondition = false;
        }
        // ---- corrections in s.T ----------------------------------------------------------------------------------
        if (!precondition) {
        } else if (s.Q) {
            // T <- Q diag(1 / (d - theta_h)) Q^T T   (eigensolvers.py:119-121 'gd', in the eigenbasis of P)
            BCHK(launch_panel16(c, s.Qt->d, n, n, s.ld, s.T, na, s.MID, s.ld));
            hipLaunchKernelGGL(bd_shift_scale_kernel, dim3((n + 255) / 256, na), dim3(256), 0, c->stream, n, na, s.MID, s.ld,
                               s.dP, th, s.guard);
            if (na < BD_NB) BHIP(s_memset0(c, s.MID + (size_t)na * s.ld, (size_t)(BD_NB - na) * s.ld * sizeof(double)));
            BCHK(launch_panel16(c, s.Q->d, n, n, s.ld, s.MID, na, s.T, s.ld));
        } else if (s.dP) {
            hipLaunchKernelGGL(bd_shift_scale_kernel, dim3((n + 255) / 256, na), dim3(256), 0, c->stream, n, na, s.T, s.ld, s.dP,
                               th, s.guard);
        }
        BHIP(hipGetLastError());
        // ---- thick restart before the basis overflows ---------------------------------------------------------------
        if (s.k + na > s.maxvec) {
            // (nev <= block: an iteration adds at most nev vectors — the rule of run_pipelined, see there; else nev + block)
            const int keep = (nev <= block) ? std::min(s.k, std::max(nev, std::min(std::max(2 * nev, nev + 8), s.maxvec - nev)))
                                            : std::min(s.k, std::max(nev + block, 2 * block));
            for (int pass = 0; pass < 2; ++pass) {
                double* src = pass ? s.AV : s.V;
                for (int j0 = 0; j0 < keep; j0 += BD_NB) {
                    const int nh = std::min(BD_NB, keep - j0);
                    Ch.assign((size_t)nh * s.k, 0.0);
                    for (int h = 0; h < nh; ++h)
                        for (int a = 0; a < s.k; ++a) Ch[(size_t)h * s.k + a] = W[(size_t)a * s.k + j0 + h];
                    BCHK(put_coeffs(s, Ch, nh, s.k));
                    BCHK(combine(s, nh, s.k, s.dC, s.kcap, src, 1.0, 0.0, s.Vt + (size_t)j0 * s.ld));
                }
                BHIP(s_memcpy(c, src, s.Vt, (size_t)keep * s.ld * sizeof(double), hipMemcpyDeviceToDevice));
                BHIP(s_memset0(c, src + (size_t)keep * s.ld, (size_t)(s.kcap - keep) * s.ld * sizeof(double)));
            }
            std::fill(s.G.begin(), s.G.end(), 0.0);
            for (int a = 0; a < keep; ++a) s.G[(size_t)a * s.kcap + a] = theta[a];
            s.k = keep;
            theta.resize(keep);                       // the restarted basis IS the Ritz basis: W = I until the next RR
            W.assign((size_t)keep * keep, 0.0);
            for (int a = 0; a < keep; ++a) W[(size_t)a * keep + a] = 1.0;
        }
        BCHK(orthonormalise_block(s, na, s.k, &kept));
        ++iter;
        if (kept == 0) break;                  // the corrections are in span(V): nothing left to add
    }

    // ---- results: lowest nev Ritz pairs --------------------------------------------------------------------------
    const double ts3 = bd_now_us();
    const int k = s.k, nout = std::min(nev, k);
    for (int j = 0; j < nev; ++j) lams_out[j] = (j < nout) ? theta[j] : 0.0;
    if (res_out)
        for (int j = 0; j < nev; ++j) res_out[j] = (j < nout) ? rn[j] : -1.0;
    static thread_local vec Xh;                  // (kept across calls: fresh pages cost more than the transfer)
    Xh.resize((size_t)n * BD_NB);
    if (nout < nev)
        for (size_t e = 0; e < (size_t)n * nev; ++e) V_out[e] = 0.0;
    for (int j0 = 0; j0 < nout; j0 += BD_NB) {
        const int nh = std::min(BD_NB, nout - j0);
        Ch.assign((size_t)nh * k, 0.0);
        for (int h = 0; h < nh; ++h)
            for (int a = 0; a < k; ++a) Ch[(size_t)h * k + a] = W[(size_t)a * k + j0 + h];
        BCHK(put_coeffs(s, Ch, nh, k));
        BCHK(combine(s, nh, k, s.dC, s.kcap, s.V, 1.0, 0.0, s.R));
        BCHK(d2h_async_2d(c, Xh.data(), s.R, (size_t)s.ld * sizeof(double), (size_t)n * sizeof(double), nh));   // vector-major
        BCHK(stream_wait(c));
        for (int i = 0; i < n; ++i)
            for (int h = 0; h < nh; ++h) V_out[(size_t)i * nev + j0 + h] = Xh[(size_t)h * n + i];
    }
    if (niter_out) *niter_out = iter;
    if (nmatvec_out) *nmatvec_out = s.nmatvec;
    if (nconv_out) *nconv_out = done ? nev : nconv;
    const double ts4 = bd_now_us();
    blk_free(s);
    if (stage_timing)
        fprintf(stderr, "block davidson: allocation + start vectors %.1f us, start block orthonormalised %.1f us, iterations %.1f us, "
                        "Ritz vectors to the host %.1f us, release %.1f us\n", ts1 - ts0, ts2 - ts1, ts3 - ts2, ts4 - ts3, bd_now_us() - ts4);
    return SELLA_OK;
#undef BCHK
#undef BHIP
}
