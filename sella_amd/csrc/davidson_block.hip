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
// rowscale (device, nh entries, may be null): output h is additionally scaled by rowscale[h] — the -theta_h of the
// residual when the Ritz values never leave the device.
__global__ __launch_bounds__(256) void bd_combine_kernel(int n, int nh, int k, const double* __restrict__ C, int ldc,
                                                         const double* __restrict__ P, int ldp, double alpha, double beta,
                                                         double* __restrict__ out, int ldo,
                                                         const double* __restrict__ rowscale) {
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
                *o = (beta == 0.0) ? al * acc[h] : (beta * (*o) + al * acc[h]);
            }
    }
}

struct Theta16 {
    double v[BD_NB];
};

// ---- Rayleigh-Ritz on the device ---------------------------------------------------------------------------------
// The k x k projected eigenproblem (k <= 64: nev + 2 blocks) used to be solved on the host — 0.3-0.5 ms of scalar
// tred2 / tql2 per block iteration at k = 48, more than the sharded panel product it sits behind.  One workgroup does it
// by parallel cyclic Jacobi instead: the matrix in LDS, k / 2 disjoint rotations per step in round-robin order, k - 1
// steps per sweep, ~6-8 sweeps to full accuracy (Jacobi is backward stable and at least as accurate as QL); the
// eigenvectors accumulate as ROWS of Wt in global memory (L2 resident), sorted ascending with the host routine's sign
// convention, ready to be read as the coefficient rows of bd_combine_kernel — so neither the Ritz vectors' coefficients
// nor the Gram matrix ever travel to the host.
constexpr int BD_JMAX = 56;      // both k x k matrices of the Jacobi kernel live in LDS: 2 x 56 x 56 doubles = 49 KB

__global__ __launch_bounds__(256) void bd_jacobi_eig_kernel(int k, const double* __restrict__ G, int ldg,
                                                            double* __restrict__ theta, double* __restrict__ Wt, int ldw,
                                                            double* __restrict__ /*unused*/, int* __restrict__ info) {
    __shared__ double Af[BD_JMAX * BD_JMAX], Wf[BD_JMAX * BD_JMAX];
    __shared__ double cc[BD_JMAX / 2], ss[BD_JMAX / 2], red[256];
    __shared__ int pp[BD_JMAX / 2], qq[BD_JMAX / 2], rankof[BD_JMAX];
    __shared__ double dsort[BD_JMAX];
    __shared__ int again;
    const int tid = threadIdx.x;
    const int k2 = k + (k & 1);                       // even size; the pad index carries a decoupled zero
    for (int e = tid; e < k2 * k2; e += 256) {
        const int i = e / k2, j = e % k2;
        double v = 0.0;
        if (i < k && j < k) v = 0.5 * (G[(size_t)i * ldg + j] + G[(size_t)j * ldg + i]);
        Af[(i) * k2 + (j)] = v;
    }
    for (int e = tid; e < k2 * k2; e += 256) {
        const int i = e / k2, j = e % k2;
        Wf[(i) * k2 + (j)] = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();
    const int half = k2 / 2, m1 = k2 - 1;
    int sweep = 0;
    for (; sweep < 40; ++sweep) {
        // convergence: off-diagonal mass against the diagonal's
        double off = 0.0, dia = 0.0;
        for (int e = tid; e < k2 * k2; e += 256) {
            const int i = e / k2, j = e % k2;
            const double v = Af[(i) * k2 + (j)];
            if (i == j) dia += v * v; else off += v * v;
        }
        red[tid] = off;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) { if (tid < st) red[tid] += red[tid + st]; __syncthreads(); }
        const double offt = red[0];
        __syncthreads();
        red[tid] = dia;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) { if (tid < st) red[tid] += red[tid + st]; __syncthreads(); }
        const double diat = red[0];
        __syncthreads();
        // rounding floor of the rotations: (4 eps)^2 per entry against the diagonal's mass
        if (tid == 0) again = (offt > 7.9e-31 * k2 * diat && offt > 0.0) ? 1 : 0;
        __syncthreads();
        if (!again) break;
        for (int step = 0; step < m1; ++step) {
            if (tid < half) {
                int p, q;
                if (tid == 0) { p = m1; q = step; }
                else {
                    p = step + tid; if (p >= m1) p -= m1;
                    q = step + m1 - tid; if (q >= m1) q -= m1;
                }
                if (p > q) { const int t = p; p = q; q = t; }
                const double apq = Af[p * k2 + q];
                double c = 1.0, sn = 0.0;
                if (fabs(apq) > 1e-300) {
                    const double tau = (Af[q * k2 + q] - Af[p * k2 + p]) / (2.0 * apq);
                    const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    c = 1.0 / sqrt(1.0 + t * t);
                    sn = t * c;
                }
                pp[tid] = p; qq[tid] = q; cc[tid] = c; ss[tid] = sn;
            }
            __syncthreads();
            // A <- J^T A J in ONE phase: the index pairs partition the matrix into disjoint 2 x 2 blocks
            // (row pair tr) x (column pair tc), each read and written by one thread; thread (tr, tc) = (tid / 32, tid % 32)
            const int tc = tid & 31;
            for (int tr = tid >> 5; tr < half; tr += 8) {
                if (tc < half) {
                    const int p = pp[tr], q = qq[tr], u = pp[tc], v = qq[tc];
                    const double cr = cc[tr], sr = ss[tr], c2 = cc[tc], s2 = ss[tc];
                    const double apu = Af[p * k2 + u], apv = Af[p * k2 + v], aqu = Af[q * k2 + u], aqv = Af[q * k2 + v];
                    // rows: (p, q) <- (c p - s q, s p + c q)
                    const double bpu = cr * apu - sr * aqu, bpv = cr * apv - sr * aqv;
                    const double bqu = sr * apu + cr * aqu, bqv = sr * apv + cr * aqv;
                    // columns: (u, v) <- (c u - s v, s u + c v)
                    Af[p * k2 + u] = c2 * bpu - s2 * bpv;
                    Af[p * k2 + v] = s2 * bpu + c2 * bpv;
                    Af[q * k2 + u] = c2 * bqu - s2 * bqv;
                    Af[q * k2 + v] = s2 * bqu + c2 * bqv;
                }
            }
            // eigenvector rows p, q of W: thread (tr, column j), 64 columns per pass
            {
                const int j = tid & 63;
                if (j < k2)
                    for (int tr = tid >> 6; tr < half; tr += 4) {
                        const int p = pp[tr], q = qq[tr];
                        const double c = cc[tr], sn = ss[tr];
                        const double wp = Wf[p * k2 + j], wq = Wf[q * k2 + j];
                        Wf[p * k2 + j] = c * wp - sn * wq;
                        Wf[q * k2 + j] = sn * wp + c * wq;
                    }
            }
            __syncthreads();
        }
    }
    if (tid == 0 && sweep >= 40) info[0] = 1;
    // ascending order (ties: lower index first), sign: largest-magnitude component positive
    if (tid < k) dsort[tid] = Af[(tid) * k2 + (tid)];
    __syncthreads();
    if (tid < k) {
        const double d = dsort[tid];
        int r = 0;
        for (int j = 0; j < k; ++j) r += (dsort[j] < d || (dsort[j] == d && j < tid)) ? 1 : 0;
        rankof[tid] = r;
        theta[r] = d;
    }
    __syncthreads();
    for (int row = tid >> 6; row < k; row += 4) {                 // one wavefront per eigenvector row
        const int lane = tid & 63;
        double best = 0.0;
        int bi = 0;
        for (int j = lane; j < k; j += 64) {
            const double v = fabs(Wf[(row) * k2 + (j)]);
            if (v > best) { best = v; bi = j; }
        }
        for (int mz = 32; mz > 0; mz >>= 1) {
            const double ob = __shfl_xor(best, mz, 64);
            const int oi = __shfl_xor(bi, mz, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        const double sgn = (Wf[(row) * k2 + (bi)] < 0.0) ? -1.0 : 1.0;
        const int r = rankof[row];
        for (int j = lane; j < k; j += 64) Wt[(size_t)r * ldw + j] = sgn * Wf[(row) * k2 + (j)];
    }
}

// New rows / columns of the projected matrix from the panel product dY[h * ldy + a] = V_a . (A T)_h, h < nbk, a < k:
// G[a][k0 + h] = G[k0 + h][a], the new diagonal block exactly symmetric.
__global__ __launch_bounds__(256) void bd_gram_rows_kernel(const double* __restrict__ dY, int ldy, int k0, int nbk, int k,
                                                           double* __restrict__ G, int ldg) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nbk * k) return;
    const int h = e / k, a = e % k;
    double v = dY[(size_t)h * ldy + a];
    if (a >= k0) v = 0.5 * (v + dY[(size_t)(a - k0) * ldy + k0 + h]);
    G[(size_t)a * ldg + k0 + h] = v;
    G[(size_t)(k0 + h) * ldg + a] = v;
}

// G <- diag(theta[0 .. keep)) after a thick restart (the restarted basis is the Ritz basis)
__global__ __launch_bounds__(256) void bd_gram_reset_kernel(double* __restrict__ G, int ldg, int kcap, int keep,
                                                            const double* __restrict__ theta) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= kcap * kcap) return;
    const int i = e / kcap, j = e % kcap;
    G[(size_t)i * ldg + j] = (i == j && i < keep) ? theta[i] : 0.0;
}

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
    // Rayleigh-Ritz on the device (kcap <= 64): projected matrix, Ritz values, Ritz coefficient rows
    bool dev_rr = false;
    double *dG = nullptr, *dWt = nullptr, *dWtmp = nullptr, *dtheta = nullptr;
    int* dinfo = nullptr;
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
    s.dev_rr = s.maxvec <= BD_JMAX && c->opt.bd_dev_rr;      // the basis never holds more than maxvec vectors
    if (s.dev_rr) {
        const size_t kk2 = (size_t)s.kcap * s.kcap;
        SCHK(dev_alloc(c, (3 * kk2 + 2 * (size_t)s.kcap + 16) * sizeof(double), &s.dG));
        HIPCHK(s_memset0(c, s.dG, (3 * kk2 + 2 * (size_t)s.kcap + 16) * sizeof(double)));
        s.dWt = s.dG + kk2;
        s.dWtmp = s.dWt + kk2;
        s.dtheta = s.dWtmp + kk2;
        s.dinfo = reinterpret_cast<int*>(s.dtheta + 2 * (size_t)s.kcap);
    }
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
    if (s.dG) dev_free(c, s.dG, (3 * (size_t)s.kcap * s.kcap + 2 * (size_t)s.kcap + 16) * sizeof(double));
    if (s.send) dev_free(c, s.send, s.bytesS);
    if (s.recv) dev_free(c, s.recv, s.bytesR);
    s.V = nullptr;
}

int combine(Blk& s, int nh, int k, const double* dC, int ldc, const double* P, double alpha, double beta, double* out,
            const double* rowscale = nullptr) {
    if (nh <= 0) return SELLA_OK;
    if (k <= 0) {
        if (beta == 0.0) HIPCHK(hipMemsetAsync(out, 0, (size_t)nh * s.ld * sizeof(double), s.c->stream));
        return SELLA_OK;
    }
    dim3 grid((s.n + 255) / 256, (nh + BD_HG - 1) / BD_HG);
    hipLaunchKernelGGL(bd_combine_kernel, grid, dim3(256), 0, s.c->stream, s.n, nh, k, dC, ldc, P, s.ld, alpha, beta, out,
                       s.ld, rowscale);
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
int apply_A(Blk& s, const double* X, int nh, double* Y) {
    sella_ctx* c = s.c;
    s.nmatvec += nh;
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
int svqb(Blk& s, double*& T, double*& T2, int nt, bool has_pre, double drop, int* kept, bool* clean = nullptr) {
    sella_ctx* c = s.c;
    *kept = 0;
    if (clean) *clean = false;
    if (nt <= 0) return SELLA_OK;
    double* dS = c->dscal + DS_GRAM;
    SCHK(launch_panel16(c, T, nt, s.n, s.ld, T, nt, dS, BD_NB));                  // dS[h * 16 + r] = T_r . T_h
    SCHK(read_scalars(c, DS_GRAM, BD_NB * BD_NB + BD_NB));                        // ... + the 16 pre-projection norms
    const double* S = c->hscal + DS_GRAM;
    const double* pre = has_pre ? c->hscal + DS_GRAM + BD_NB * BD_NB : nullptr;
    std::vector<int> live;
    for (int h = 0; h < nt; ++h) {
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
    if (clean && pre) {
        bool ok = mk == m && (int)live.size() == nt && sig[0] > 1e-3 * smax;
        for (int a = 0; a < m && ok; ++a) ok = S[live[a] * BD_NB + live[a]] >= 0.25 * pre[live[a]];
        *clean = ok;
    }
    // T_new[jj] = sum_a dinv_a U[a][j] / sqrt(sig_j) T[live_a]
    vec Ch((size_t)mk * nt, 0.0);
    for (int jj = 0; jj < mk; ++jj) {
        const int j = good[jj];
        const double f = 1.0 / sqrt(sig[j]);
        for (int a = 0; a < m; ++a) Ch[(size_t)jj * nt + live[a]] = dinv[a] * U[(size_t)a * m + j] * f;
    }
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
    // default basis limit: the nev + block lowest Ritz vectors kept at a restart plus one new block — the k x k
    // Rayleigh-Ritz problem on the host is O(k^3) in scalar code (0.5 ms at k = 48, 3 ms at k = 96: more than the
    // whole device side of an iteration at 3N = 12288), so a larger history has to be asked for explicitly
    if (maxvec <= 0) maxvec = std::max(nev + 2 * block, 24);
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
        } else {
            nt = std::min(block, n);
            X0.assign((size_t)n * nt, 0.0);
            if (pdiag && !s.Q) {
                // unit vectors at the `nt` smallest diagonal entries (Davidson's classic start)
                std::vector<int> idx(n);
                for (int i = 0; i < n; ++i) idx[i] = i;
                std::partial_sort(idx.begin(), idx.begin() + nt, idx.end(),
                                  [&](int a, int b) { return pdiag[a] < pdiag[b] || (pdiag[a] == pdiag[b] && a < b); });
                for (int j = 0; j < nt; ++j) X0[(size_t)idx[j] * nt + j] = 1.0;
            } else {
                unsigned long long lcg = 0x9E3779B97F4A7C15ull;          // deterministic pseudo-random start
                for (size_t e = 0; e < X0.size(); ++e) {
                    lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
                    X0[e] = (double)(lcg >> 11) * (1.0 / 9007199254740992.0) - 0.5;
                }
            }
        }
        BCHK(upload_panel(c, X0.data(), n, nt, s.T, s.ld));
        if (s.Q && !(V0 && nv0 > 0)) {
            // with an eigenbasis preconditioner the natural start is its lowest eigenvectors
            BHIP(hipMemcpy2DAsync(s.T, (size_t)s.ld * sizeof(double), s.Qt->d, (size_t)s.Qt->ld * sizeof(double),
                                  (size_t)n * sizeof(double), nt, hipMemcpyDeviceToDevice, c->stream));
        }
    }
    int kept = 0;
    BCHK(orthonormalise_block(s, nt, 0, &kept));
    if (kept == 0) { set_error("davidson_block: the start block is numerically zero"); return fail(SELLA_E_INVALID); }

    vec theta, W, Gk, work, rn(std::max(nev, BD_NB), 0.0), Ch;
    unsigned long long lcg = 0xD1B54A32D192ED03ull;
    std::vector<char> conv(nev, 0);
    int iter = 0, nconv = 0;
    bool done = false;
    while (true) {
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
            if (s.dev_rr) {
                hipLaunchKernelGGL(bd_gram_rows_kernel, dim3((nbk * s.k + 255) / 256), dim3(256), 0, c->stream, dY, s.kcap, k0,
                                   nbk, s.k, s.dG, s.kcap);
                BHIP(hipGetLastError());
            } else {
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
        }
        // ---- Rayleigh-Ritz ------------------------------------------------------------------------------------
        const int k = s.k;
        theta.assign(k, 0.0);
        if (s.dev_rr) {
            // one workgroup: Ritz values -> dtheta, Ritz coefficient rows -> dWt; theta comes back with the residual norms
            hipLaunchKernelGGL(bd_jacobi_eig_kernel, dim3(1), dim3(256), 0, c->stream, k, s.dG, s.kcap, s.dtheta, s.dWt, s.kcap,
                               s.dWtmp, s.dinfo);
            BHIP(hipGetLastError());
        } else {
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
        }
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
            if (s.dev_rr) {
                // R = (AV) W - theta (V W) with the coefficient rows and -theta read from the device
                const double* Crow = s.dWt + (size_t)j0 * s.kcap;
                BCHK(combine(s, nh, k, Crow, s.kcap, s.AV, 1.0, 0.0, s.R));
                BCHK(combine(s, nh, k, Crow, s.kcap, s.V, -1.0, 1.0, s.R, s.dtheta + j0));
                BCHK(launch_rows_sumsq(c, s.R, s.ld, nh, n, c->dscal + DS_MISC));
                if (j0 == 0) {
                    BCHK(d2h_async(c, theta.data(), s.dtheta, (size_t)k * sizeof(double)));
                    BHIP(hipMemcpyAsync(c->hscal + DS_MISC + 32, s.dinfo, sizeof(int), hipMemcpyDeviceToHost, c->stream));
                }
                BCHK(read_scalars(c, DS_MISC, nh));
                if (j0 == 0 && *reinterpret_cast<const int*>(c->hscal + DS_MISC + 32) != 0) {
                    set_error("davidson_block: Rayleigh-Ritz eigenproblem failed (Jacobi sweeps exhausted)");
                    return fail(SELLA_E_NOCONV);
                }
            } else {
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
            }
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
            const int keep = std::min(s.k, std::max(nev + block, 2 * block));
            for (int pass = 0; pass < 2; ++pass) {
                double* src = pass ? s.AV : s.V;
                for (int j0 = 0; j0 < keep; j0 += BD_NB) {
                    const int nh = std::min(BD_NB, keep - j0);
                    if (s.dev_rr) {
                        BCHK(combine(s, nh, s.k, s.dWt + (size_t)j0 * s.kcap, s.kcap, src, 1.0, 0.0, s.Vt + (size_t)j0 * s.ld));
                        continue;
                    }
                    Ch.assign((size_t)nh * s.k, 0.0);
                    for (int h = 0; h < nh; ++h)
                        for (int a = 0; a < s.k; ++a) Ch[(size_t)h * s.k + a] = W[(size_t)a * s.k + j0 + h];
                    BCHK(put_coeffs(s, Ch, nh, s.k));
                    BCHK(combine(s, nh, s.k, s.dC, s.kcap, src, 1.0, 0.0, s.Vt + (size_t)j0 * s.ld));
                }
                BHIP(s_memcpy(c, src, s.Vt, (size_t)keep * s.ld * sizeof(double), hipMemcpyDeviceToDevice));
                BHIP(s_memset0(c, src + (size_t)keep * s.ld, (size_t)(s.kcap - keep) * s.ld * sizeof(double)));
            }
            if (s.dev_rr) {
                hipLaunchKernelGGL(bd_gram_reset_kernel, dim3((s.kcap * s.kcap + 255) / 256), dim3(256), 0, c->stream, s.dG,
                                   s.kcap, s.kcap, keep, s.dtheta);
                BHIP(hipGetLastError());
            } else {
                std::fill(s.G.begin(), s.G.end(), 0.0);
                for (int a = 0; a < keep; ++a) s.G[(size_t)a * s.kcap + a] = theta[a];
            }
            s.k = keep;
            theta.resize(keep);                       // the restarted basis IS the Ritz basis: W = I until the next RR
            if (!s.dev_rr) {
                W.assign((size_t)keep * keep, 0.0);
                for (int a = 0; a < keep; ++a) W[(size_t)a * keep + a] = 1.0;
            }
        }
        BCHK(orthonormalise_block(s, na, s.k, &kept));
        ++iter;
        if (kept == 0) break;                  // the corrections are in span(V): nothing left to add
    }

    // ---- results: lowest nev Ritz pairs --------------------------------------------------------------------------
    const int k = s.k, nout = std::min(nev, k);
    for (int j = 0; j < nev; ++j) lams_out[j] = (j < nout) ? theta[j] : 0.0;
    if (res_out)
        for (int j = 0; j < nev; ++j) res_out[j] = (j < nout) ? rn[j] : -1.0;
    vec Xh((size_t)n * BD_NB);
    for (size_t e = 0; e < (size_t)n * nev; ++e) V_out[e] = 0.0;
    for (int j0 = 0; j0 < nout; j0 += BD_NB) {
        const int nh = std::min(BD_NB, nout - j0);
        if (s.dev_rr) {
            BCHK(combine(s, nh, k, s.dWt + (size_t)j0 * s.kcap, s.kcap, s.V, 1.0, 0.0, s.R));
        } else {
        Ch.assign((size_t)nh * k, 0.0);
        for (int h = 0; h < nh; ++h)
            for (int a = 0; a < k; ++a) Ch[(size_t)h * k + a] = W[(size_t)a * k + j0 + h];
        BCHK(put_coeffs(s, Ch, nh, k));
        BCHK(combine(s, nh, k, s.dC, s.kcap, s.V, 1.0, 0.0, s.R));
        }
        BCHK(download_panel(c, s.R, s.ld, n, nh, Xh.data()));
        for (int i = 0; i < n; ++i)
            for (int h = 0; h < nh; ++h) V_out[(size_t)i * nev + j0 + h] = Xh[(size_t)i * nh + h];
    }
    if (niter_out) *niter_out = iter;
    if (nmatvec_out) *nmatvec_out = s.nmatvec;
    if (nconv_out) *nconv_out = done ? nev : nconv;
    blk_free(s);
    return SELLA_OK;
#undef BCHK
#undef BHIP
}
