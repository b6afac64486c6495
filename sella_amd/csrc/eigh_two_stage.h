// eigh_two_stage.h — two-stage tridiagonalisation for sella_eigh (included by eigh.hip, which owns the divide & conquer
// stage and the compact-WY back-transformation this file reuses).
//
//   stage 1  dense -> band (bandwidth b = 32): per panel of b columns a Householder QR of the block below the band
//            (ts_panel_qr_kernel, on the ROWS of the symmetric matrix, so every access is along a row), its compact-WY
//            factor, and the two-sided update of the trailing block as one 32-right-hand-side pass over it (two
//            panel16 MFMA passes), four small GEMMs and the streaming rank-2b update (rank2k_stream) — level-3 work:
//            the trailing matrix is read n/b times instead of once per column.
//   stage 2  band -> tridiagonal by bulge chasing (ts_chase_kernel): sweep s annihilates column s below the sub-diagonal
//            with a reflector of length b and chases the bulge down the band, one b x b block pair per task.  Task
//            (s, k) depends on (s, k - 1) and (s - 1, k + 1): all tasks with 2 s + k = t are independent, and THIS
//            version runs one launch per t (2 n + n / b launches: correct under the host emulation, and what the
//            persistent point-to-point version — sweeps handed over through device flags, DESIGN.md section 8 — has to
//            reproduce task for task).
//   back     X <- X Q2^T Q1^T on the rows of X (eigenvectors of the tridiagonal matrix as rows, as the one-stage path
//            keeps them).  Q2: the reflectors of G consecutive sweeps at the same k form one compact-WY block on
//            b + G - 1 consecutive columns; groups descending, k ascending inside a group (the order in which they
//            commute into blocks: see ts_q2_apply_kernel).  Q1: the panels' blocks through wy_apply_mfma_kernel as is.
//
// Where it pays: the one-stage path reads the trailing matrix once per column (8 n^3 / 3 bytes, bandwidth bound from
// 3N ~ 6000 on); here stage 1 is compute bound and the chain of stage 2 is 2 n launches whatever b is.  At 3N = 3072 the
// chain alone is as long as the one-stage factorisation (option eigh_two_stage, by size: eigh2_min).
#pragma once

namespace sella {
namespace {

constexpr int TS_BMAX = 32;                 // bandwidth (and reflectors per block of either back-transformation)

__device__ __forceinline__ double ts_block_sum(double v, double* red, int nwaves) {
    v = wave_sum64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < nwaves; ++w) s += red[w];
    return s;
}

// LAPACK dlarfg on a vector whose head is alpha and whose tail has squared norm xn2: tau, beta and the scale of the tail
__device__ __forceinline__ void ts_larfg(double alpha, double xn2, double* tau, double* beta, double* scale) {
    if (xn2 == 0.0) { *tau = 0.0; *beta = alpha; *scale = 0.0; return; }
    const double nrm = sqrt(alpha * alpha + xn2);
    const double be = alpha >= 0.0 ? -nrm : nrm;
    *beta = be;
    *tau = (be - alpha) / be;
    *scale = 1.0 / (alpha - be);
}

// ---- stage 1 ---------------------------------------------------------------------------------------------------------
// Householder QR of the panel P = A[r0:, p : p + b] (m x b), taken on its transpose Pt = A[p : p + b, r0:] (b rows of m
// contiguous entries; A is symmetric).  One workgroup of 16 wavefronts: per column a block-wide norm, then every wavefront
// updates "its" remaining rows (row j' belongs to wavefront j' mod 16: a dot and an axpy along the row, no block-wide
// exchange).  Yt row j (absolute column index, zero outside the reflector) <- v_j; taus[j]; R stays in Pt, the reflector
// tails there are zeroed (they are the entries the reduction annihilates).
__global__ __launch_bounds__(1024) void ts_panel_qr_kernel(double* __restrict__ A, int ld, int n, int p, int b,
                                                           double* __restrict__ Yt, double* __restrict__ taus) {
    __shared__ double red[16];
    __shared__ double sc[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = p + b, m = n - r0;
    for (int j = 0; j < b; ++j) {
        double* row = A + (size_t)(p + j) * ld + r0;
        double* yrow = Yt + (size_t)j * ld + r0;
        if (j >= m - 1 || j >= m) {
            // no sub-column left: identity reflector (the unit entry only where the row exists)
            if (tid == 0) { taus[j] = 0.0; if (j < m) yrow[j] = 1.0; }
            __syncthreads();
            continue;
        }
        double part = 0.0;
        for (int i = j + 1 + tid; i < m; i += 1024) part += row[i] * row[i];
        const double xn2 = ts_block_sum(part, red, 16);
        if (tid == 0) {
            double tau, beta, scale;
            ts_larfg(row[j], xn2, &tau, &beta, &scale);
            sc[0] = tau; sc[1] = beta; sc[2] = scale;
            taus[j] = tau;
        }
        __syncthreads();
        const double tau = sc[0], beta = sc[1], scale = sc[2];
        for (int i = j + tid; i < m; i += 1024) {
            if (i == j) { yrow[i] = 1.0; row[i] = beta; }
            else { yrow[i] = row[i] * scale; row[i] = 0.0; }
        }
        __syncthreads();
        if (tau != 0.0) {
            for (int jp = j + 1 + wave; jp < b; jp += 16) {
                double* r2 = A + (size_t)(p + jp) * ld + r0;
                double d = 0.0;
                for (int i = j + lane; i < m; i += 64) d += yrow[i] * r2[i];
                d = wave_sum64(d) * tau;
                for (int i = j + lane; i < m; i += 64) r2[i] -= d * yrow[i];
            }
        }
        __syncthreads();
    }
}

// Compact-WY factor of a block of nb <= 32 reflectors from their Gram matrix Gm (row-major nb x nb, leading dimension ldg)
// and scalars: T upper triangular, forward columnwise (dlarft): T[0:j, j] = -tau_j T[0:j, 0:j] Gm[0:j, j].  Writes T
// (row-major, ld 32) and C = T^T (what wy_apply_* take).  One workgroup of 64 threads.
__global__ __launch_bounds__(64) void ts_tfactor_kernel(const double* __restrict__ Gm, int ldg, const double* __restrict__ taus,
                                                        int nb, double* __restrict__ T, double* __restrict__ C) {
    __shared__ double sT[TS_BMAX][TS_BMAX + 1];
    const int tid = threadIdx.x;
    for (int e = tid; e < TS_BMAX * TS_BMAX; e += 64) sT[e / TS_BMAX][e % TS_BMAX] = 0.0;
    __syncthreads();
    for (int j = 0; j < nb; ++j) {
        const double tj = taus[j];
        double val = 0.0;
        if (tid < j) {
            double s = 0.0;
            for (int l = tid; l < j; ++l) s += sT[tid][l] * Gm[(size_t)l * ldg + j];
            val = -tj * s;
        }
        __syncthreads();
        if (tid < j) sT[tid][j] = val;
        if (tid == j) sT[j][j] = tj;
        __syncthreads();
    }
    for (int e = tid; e < TS_BMAX * TS_BMAX; e += 64) {
        const int i = e / TS_BMAX, j = e % TS_BMAX;
        T[e] = sT[i][j];
        C[e] = sT[j][i];
    }
}

// S = (T^T G2 + (T^T G2)^T) / 4  (= T^T Y^T A Y T / 2, symmetric up to roundoff), b x b with leading dimension 32
__global__ __launch_bounds__(256) void ts_smat_kernel(const double* __restrict__ T, const double* __restrict__ G2, int nb,
                                                      double* __restrict__ S) {
    __shared__ double sM[TS_BMAX][TS_BMAX + 1];
    for (int e = threadIdx.x; e < TS_BMAX * TS_BMAX; e += 256) {
        const int i = e / TS_BMAX, j = e % TS_BMAX;
        double s = 0.0;
        if (i < nb && j < nb)
            for (int l = 0; l <= i; ++l) s += T[l * TS_BMAX + i] * G2[l * TS_BMAX + j];      // (T^T)[i][l] = T[l][i], l <= i
        sM[i][j] = s;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < TS_BMAX * TS_BMAX; e += 256) {
        const int i = e / TS_BMAX, j = e % TS_BMAX;
        S[e] = 0.25 * (sM[i][j] + sM[j][i]);
    }
}

// upper band of the reduced matrix (row form) -> lower band storage with room for the bulge: AB[j * LDB + d] = B(j + d, j),
// d <= b from A[j][j + d], zero above (LDB = 2 b)
__global__ __launch_bounds__(256) void ts_band_extract_kernel(const double* __restrict__ A, int ld, int n, int b,
                                                              double* __restrict__ AB, int LDB) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n * LDB) return;
    const int j = e / LDB, d = e % LDB;
    AB[e] = (d <= b && j + d < n) ? A[(size_t)j * ld + j + d] : 0.0;
}

// ---- stage 2 ---------------------------------------------------------------------------------------------------------
struct ChaseArgs {
    double* AB; int LDB, n, b;
    int t, smin, count;                        // tasks (s, k = t - 2 s), s = smin .. smin + count - 1
    double* Vst; double* taus; int KMAX;       // reflector of task (s, k): Vst[(s * KMAX + k) * b ..], taus[s * KMAX + k]
};

// One task of the bulge chase (see the file header).  r = s + 1 + k b, rows / columns J = [r, r + L), L = min(b, n - r).
//   k = 0: reflector from column s below the diagonal; k > 0: E = B[J, J - b] <- E H_{k-1}, reflector from its first column,
//   E <- H E;  then D = B[J, J] <- H D H.   One workgroup, the two blocks in LDS.
__global__ __launch_bounds__(256) void ts_chase_kernel(ChaseArgs a) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    __shared__ double E[TS_BMAX][TS_BMAX + 1], D[TS_BMAX][TS_BMAX + 1];
    __shared__ double vp[TS_BMAX], v[TS_BMAX], wv[TS_BMAX], sc[4];
    const int tid = threadIdx.x;
    const int s = a.smin + blockIdx.x, k = a.t - 2 * s;
    const int b = a.b, n = a.n, LDB = a.LDB;
    const int r = s + 1 + k * b;
    const int L = (n - r < b) ? n - r : b;
    double* AB = a.AB;
    // ---- load
    for (int e = tid; e < L * L; e += 256) {
        const int ii = e / L, jj = e % L;
        if (ii >= jj) {
            const double x = AB[(size_t)(r + jj) * LDB + (ii - jj)];
            D[ii][jj] = x;
            D[jj][ii] = x;
        }
    }
    const int rp = r - b;
    if (k > 0) {
        for (int e = tid; e < L * b; e += 256) {
            const int ii = e / b, jj = e % b;
            E[ii][jj] = AB[(size_t)(rp + jj) * LDB + (r + ii - rp - jj)];
        }
        if (tid < b) vp[tid] = a.Vst[((size_t)s * a.KMAX + (k - 1)) * b + tid];
        if (tid == 0) sc[3] = a.taus[(size_t)s * a.KMAX + (k - 1)];
    } else if (tid < L) {
        wv[tid] = AB[(size_t)s * LDB + 1 + tid];                  // column s below the diagonal
    }
    __syncthreads();
    if (k > 0) {
        // (a) E <- E (I - taup vp vp^T)
        const double taup = sc[3];
        if (tid < L) {
            double acc = 0.0;
            for (int jj = 0; jj < b; ++jj) acc += E[tid][jj] * vp[jj];
            wv[tid] = taup * acc;
        }
        __syncthreads();
        for (int e = tid; e < L * b; e += 256) {
            const int ii = e / b, jj = e % b;
            E[ii][jj] -= wv[ii] * vp[jj];
        }
        __syncthreads();
        if (tid < L) wv[tid] = E[tid][0];                          // the column the new reflector annihilates
        __syncthreads();
    }
    // ---- reflector from wv[0:L]
    if (tid == 0) {
        double xn2 = 0.0;
        for (int i = 1; i < L; ++i) xn2 += wv[i] * wv[i];
        double tau, beta, scale;
        ts_larfg(wv[0], xn2, &tau, &beta, &scale);
        sc[0] = tau; sc[1] = beta; sc[2] = scale;
    }
    __syncthreads();
    const double tau = sc[0], beta = sc[1], scale = sc[2];
    if (tid < b) v[tid] = (tid == 0) ? 1.0 : (tid < L ? wv[tid] * scale : 0.0);
    __syncthreads();
    if (k > 0) {
        // (b) E <- H E, first column set exactly
        if (tau != 0.0) {
            if (tid < b) {
                double acc = 0.0;
                for (int ii = 0; ii < L; ++ii) acc += v[ii] * E[ii][tid];
                wv[tid] = tau * acc;
            }
            __syncthreads();
            for (int e = tid; e < L * b; e += 256) {
                const int ii = e / b, jj = e % b;
                E[ii][jj] -= v[ii] * wv[jj];
            }
            __syncthreads();
        }
        if (tid < L) E[tid][0] = (tid == 0) ? beta : 0.0;
        __syncthreads();
        for (int e = tid; e < L * b; e += 256) {
            const int ii = e / b, jj = e % b;
            AB[(size_t)(rp + jj) * LDB + (r + ii - rp - jj)] = E[ii][jj];
        }
    } else if (tid < L) {
        AB[(size_t)s * LDB + 1 + tid] = (tid == 0) ? beta : 0.0;
    }
    // (c) D <- H D H
    if (tau != 0.0) {
        if (tid < L) {
            double acc = 0.0;
            for (int jj = 0; jj < L; ++jj) acc += D[tid][jj] * v[jj];
            wv[tid] = tau * acc;
        }
        __syncthreads();
        if (tid == 0) {
            double pv = 0.0;
            for (int i = 0; i < L; ++i) pv += wv[i] * v[i];
            sc[3] = -0.5 * tau * pv;
        }
        __syncthreads();
        const double al = sc[3];
        if (tid < L) wv[tid] += al * v[tid];
        __syncthreads();
        for (int e = tid; e < L * L; e += 256) {
            const int ii = e / L, jj = e % L;
            if (ii >= jj) AB[(size_t)(r + jj) * LDB + (ii - jj)] = D[ii][jj] - (v[ii] * wv[jj] + wv[ii] * v[jj]);
        }
    }
    if (tid < b) a.Vst[((size_t)s * a.KMAX + k) * b + tid] = v[tid];
    if (tid == 0) a.taus[(size_t)s * a.KMAX + k] = tau;
}

// d, e of the tridiagonal matrix out of the band storage
__global__ __launch_bounds__(256) void ts_diag_kernel(const double* __restrict__ AB, int LDB, int n, double* __restrict__ d,
                                                      double* __restrict__ e) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    d[j] = AB[(size_t)j * LDB];
    e[j] = (j + 1 < n) ? AB[(size_t)j * LDB + 1] : 0.0;
}

// ---- back-transformation with the reflectors of stage 2 --------------------------------------------------------------
// Block (g, k): reflectors of the sweeps s0 .. s0 + G - 1 (s0 = g G) at step k; column i of V is v(s0 + i, k), shifted down
// by i rows; rows of the block = columns c0 .. c0 + b + G - 2 of X, c0 = s0 + 1 + k b.  T (G x G) as for dlarft.
__global__ __launch_bounds__(64) void ts_q2_tfactor_kernel(const double* __restrict__ Vst, const double* __restrict__ taus, int n,
                                                           int b, int G, int KMAX, double* __restrict__ Tst) {
    __shared__ double sV[2 * TS_BMAX][TS_BMAX + 1];          // rows of the block x reflectors
    __shared__ double sG[TS_BMAX][TS_BMAX + 1], sT[TS_BMAX][TS_BMAX + 1], st[TS_BMAX];
    const int g = blockIdx.x, k = blockIdx.y, tid = threadIdx.x;
    const int s0 = g * G;
    for (int e = tid; e < 2 * TS_BMAX * TS_BMAX; e += 64) sV[e / TS_BMAX][e % TS_BMAX] = 0.0;
    for (int e = tid; e < TS_BMAX * TS_BMAX; e += 64) sT[e / TS_BMAX][e % TS_BMAX] = 0.0;
    __syncthreads();
    for (int i = 0; i < G; ++i) {
        const int s = s0 + i, r = s + 1 + k * b;
        const bool on = s <= n - 3 && r <= n - 1;
        if (tid < b) sV[i + tid][i] = on ? Vst[((size_t)s * KMAX + k) * b + tid] : 0.0;
        if (tid == 0) st[i] = on ? taus[(size_t)s * KMAX + k] : 0.0;
    }
    __syncthreads();
    for (int e = tid; e < G * G; e += 64) {
        const int i = e / G, j = e % G;
        double acc = 0.0;
        for (int c = 0; c < b + G - 1; ++c) acc += sV[c][i] * sV[c][j];
        sG[i][j] = acc;
    }
    __syncthreads();
    for (int j = 0; j < G; ++j) {
        const double tj = st[j];
        double val = 0.0;
        if (tid < j) {
            double acc = 0.0;
            for (int l = tid; l < j; ++l) acc += sT[tid][l] * sG[l][j];
            val = -tj * acc;
        }
        __syncthreads();
        if (tid < j) sT[tid][j] = val;
        if (tid == j) sT[j][j] = tj;
        __syncthreads();
    }
    double* out = Tst + ((size_t)g * KMAX + k) * TS_BMAX * TS_BMAX;
    for (int e = tid; e < TS_BMAX * TS_BMAX; e += 64) out[e] = sT[e / TS_BMAX][e % TS_BMAX];
}

// X (rows = eigenvectors) <- X Q2^T.  Q2 = prod over sweeps s ascending, steps k ascending, of H(s, k).  Inside a group of G
// consecutive sweeps the factors commute into  prod_{k descending} W_k,  W_k = prod_{s ascending} H(s, k) = I - V T V^T
// (H(s, k) and H(s', k + 1) overlap only for s' < s, and then the generation order already has (s', k + 1) first), so
// X Q2^T = X prod_{g descending} prod_{k ascending} W_{g,k}^T, W^T = I - V T^T V^T: each workgroup takes 16 rows of X through
// all blocks in that order; a block touches b + G - 1 <= 63 consecutive columns.
__global__ __launch_bounds__(256) void ts_q2_apply_kernel(double* __restrict__ X, int ldx, int n, int b, int G, int ngroups,
                                                          const double* __restrict__ Vst, const double* __restrict__ taus,
                                                          int KMAX, const double* __restrict__ Tst) {
    __shared__ double sX[16][2 * TS_BMAX + 1];
    __shared__ double sV[2 * TS_BMAX][TS_BMAX + 1];
    __shared__ double sT[TS_BMAX][TS_BMAX + 1];
    __shared__ double sM[16][TS_BMAX + 1], sM2[16][TS_BMAX + 1];
    const int tid = threadIdx.x;
    const int row0 = blockIdx.x * 16;
    const int W = b + G - 1;                                   // columns of a block
    for (int g = ngroups - 1; g >= 0; --g) {
        const int s0 = g * G;
        // steps this group has: the first sweep of the group goes furthest
        const int kcount = (n - 2 - s0) / b + 1;               // r = s0 + 1 + k b <= n - 1
        for (int k = 0; k < kcount; ++k) {
            const int c0 = s0 + 1 + k * b;
            __syncthreads();
            for (int e = tid; e < 2 * TS_BMAX * TS_BMAX; e += 256) sV[e / TS_BMAX][e % TS_BMAX] = 0.0;
            __syncthreads();
            for (int e = tid; e < G * b; e += 256) {
                const int i = e / b, q = e % b;
                const int s = s0 + i, r = s + 1 + k * b;
                if (s <= n - 3 && r <= n - 1) sV[i + q][i] = Vst[((size_t)s * KMAX + k) * b + q];
            }
            const double* Tb = Tst + ((size_t)g * KMAX + k) * TS_BMAX * TS_BMAX;
            for (int e = tid; e < TS_BMAX * TS_BMAX; e += 256) sT[e / TS_BMAX][e % TS_BMAX] = Tb[e];
            for (int e = tid; e < 16 * W; e += 256) {
                const int rr = e / W, cc = e % W;
                sX[rr][cc] = (row0 + rr < n && c0 + cc < n) ? X[(size_t)(row0 + rr) * ldx + c0 + cc] : 0.0;
            }
            __syncthreads();
            // M = Xc V (16 x G)
            for (int e = tid; e < 16 * G; e += 256) {
                const int rr = e / G, i = e % G;
                double acc = 0.0;
                for (int cc = i; cc < i + b && cc < W; ++cc) acc += sX[rr][cc] * sV[cc][i];
                sM[rr][i] = acc;
            }
            __syncthreads();
            // M2 = M T^T: M2[rr][i] = sum_{j >= i} M[rr][j] T[i][j]
            for (int e = tid; e < 16 * G; e += 256) {
                const int rr = e / G, i = e % G;
                double acc = 0.0;
                for (int j = i; j < G; ++j) acc += sM[rr][j] * sT[i][j];
                sM2[rr][i] = acc;
            }
            __syncthreads();
            // Xc -= M2 V^T
            for (int e = tid; e < 16 * W; e += 256) {
                const int rr = e / W, cc = e % W;
                if (row0 + rr < n && c0 + cc < n) {
                    double acc = 0.0;
                    const int ilo = cc - b + 1 > 0 ? cc - b + 1 : 0, ihi = cc < G - 1 ? cc : G - 1;
                    for (int i = ilo; i <= ihi; ++i) acc += sM2[rr][i] * sV[cc][i];
                    X[(size_t)(row0 + rr) * ldx + c0 + cc] = sX[rr][cc] - acc;
                }
            }
        }
    }
}

}  // namespace

// The two-stage factorisation of W.A (destroyed): d, e (host) of the tridiagonal matrix, and what the back-transformation
// needs, kept in scratch slots of the context.
struct TwoStage {
    int n = 0, ld = 0, b = 0, G = 0, npanels = 0, KMAX = 0, ngroups = 0;
    double *Ystore = nullptr, *Cstore = nullptr;             // stage 1: reflector rows (npanels * b x ld), C = T^T blocks
    double *taus1 = nullptr;
    double *Vst = nullptr, *taus2 = nullptr, *Tst = nullptr; // stage 2
};

static int two_stage_reduce(EighWork& W, TwoStage& ts, std::vector<double>& d, std::vector<double>& e, bool want_vectors) {
    sella_ctx* c = W.c;
    const int n = W.n, ld = W.ld, b = TS_BMAX;
    ts.n = n; ts.ld = ld; ts.b = b; ts.G = TS_BMAX;
    ts.npanels = 0;
    for (int p = 0; p + b <= n - 2; p += b) ++ts.npanels;
    ts.KMAX = (n - 1 + b - 1) / b;
    ts.ngroups = (n - 2 + ts.G - 1) / ts.G;
    const int LDB = 2 * b;
    const size_t nrefl1 = (size_t)std::max(1, ts.npanels) * b;
    // scratch: reflector rows of stage 1 (+ 64 spare rows: the 64-reflector kernels are not used here, but wy_apply reads whole
    // blocks), small matrices, work panels of stage 1
    SCHK(scratch_get(c, SCR_V, (nrefl1 + 2) * ld * sizeof(double), &ts.Ystore));
    double* small;
    SCHK(scratch_get(c, SCR_AV, ((size_t)(ts.npanels + 1) * (2 * TS_BMAX * TS_BMAX + TS_BMAX) + 8 * TS_BMAX * TS_BMAX + 3 * (size_t)b * ld)
                                    * sizeof(double), &small));
    ts.Cstore = small;                                          // npanels x 32 x 32
    double* Tall = ts.Cstore + (size_t)(ts.npanels + 1) * TS_BMAX * TS_BMAX;
    ts.taus1 = Tall + (size_t)(ts.npanels + 1) * TS_BMAX * TS_BMAX;
    double* Gm = ts.taus1 + (size_t)(ts.npanels + 1) * TS_BMAX;
    double* G2 = Gm + TS_BMAX * TS_BMAX;
    double* Sm = G2 + TS_BMAX * TS_BMAX;
    double* Zt = Sm + 5 * TS_BMAX * TS_BMAX;                    // b x ld work rows
    double* Pt = Zt + (size_t)b * ld;
    double* Wt = Pt + (size_t)b * ld;
    HIPCHK(hipMemsetAsync(ts.Ystore, 0, (nrefl1 + 2) * ld * sizeof(double), c->stream));
    HIPCHK(hipMemsetAsync(small, 0, ((size_t)(ts.npanels + 1) * (2 * TS_BMAX * TS_BMAX + TS_BMAX)) * sizeof(double), c->stream));
    // ---- stage 1 ----------------------------------------------------------------------------------------------------
    for (int ip = 0; ip < ts.npanels; ++ip) {
        const int p = ip * b, r0 = p + b, m = n - r0;
        double* Yt = ts.Ystore + (size_t)ip * b * ld;
        double* taus = ts.taus1 + (size_t)ip * TS_BMAX;
        double* T = Tall + (size_t)ip * TS_BMAX * TS_BMAX;
        double* C = ts.Cstore + (size_t)ip * TS_BMAX * TS_BMAX;
        hipLaunchKernelGGL(ts_panel_qr_kernel, dim3(1), dim3(1024), 0, c->stream, W.A, ld, n, p, b, Yt, taus);
        HIPCHK(hipGetLastError());
        // Gram of the reflectors, T and C = T^T
        SCHK(launch_gemm(c, 0, 1, b, b, m, 1.0, Yt + r0, ld, Yt + r0, ld, 0.0, Gm, TS_BMAX));
        hipLaunchKernelGGL(ts_tfactor_kernel, dim3(1), dim3(64), 0, c->stream, Gm, TS_BMAX, taus, b, T, C);
        HIPCHK(hipGetLastError());
        if (m <= 1) continue;
        // Zt = Yt A22 (rows): two 16-row passes over the trailing block
        double* A22 = W.A + (size_t)r0 * ld + r0;
        HIPCHK(hipMemsetAsync(Zt, 0, (size_t)3 * b * ld * sizeof(double), c->stream));
        for (int h = 0; h < b; h += 16)
            SCHK(launch_panel16(c, A22, m, m, ld, Yt + (size_t)h * ld + r0, 16, Zt + (size_t)h * ld + r0, ld));
        // Pt = T^T Zt;  G2 = Yt Pt^T;  S;  Wt = Pt - S^T Yt  (S symmetric)
        SCHK(launch_gemm(c, 1, 0, b, m, b, 1.0, T, TS_BMAX, Zt + r0, ld, 0.0, Pt + r0, ld));
        SCHK(launch_gemm(c, 0, 1, b, b, m, 1.0, Yt + r0, ld, Pt + r0, ld, 0.0, G2, TS_BMAX));
        hipLaunchKernelGGL(ts_smat_kernel, dim3(1), dim3(256), 0, c->stream, T, G2, b, Sm);
        HIPCHK(hipGetLastError());
        SCHK(launch_axpby2d(c, b, m, 1.0, Pt + r0, ld, 0.0, nullptr, 0, Wt + r0, ld));
        SCHK(launch_gemm(c, 0, 0, b, m, b, -1.0, Sm, TS_BMAX, Yt + r0, ld, 1.0, Wt + r0, ld));
        // A22 <- A22 - Y W^T - W Y^T
        SCHK(launch_rank2k_stream(c, A22, m, ld, Yt + r0, Wt + r0, ld, b, -1.0));
    }
    // ---- band storage, stage 2 --------------------------------------------------------------------------------------------
    double* AB;
    SCHK(scratch_get(c, SCR_V2, ((size_t)n * LDB + 4 * LDB) * sizeof(double), &AB));
    hipLaunchKernelGGL(ts_band_extract_kernel, dim3((n * LDB + 255) / 256), dim3(256), 0, c->stream, W.A, ld, n, b, AB, LDB);
    HIPCHK(hipGetLastError());
    const size_t nv = (size_t)n * ts.KMAX;
    SCHK(scratch_get(c, SCR_AV2, (nv * b + nv + 64) * sizeof(double), &ts.Vst));
    ts.taus2 = ts.Vst + nv * b;
    HIPCHK(hipMemsetAsync(ts.Vst, 0, (nv * b + nv) * sizeof(double), c->stream));
    ChaseArgs ca;
    ca.AB = AB; ca.LDB = LDB; ca.n = n; ca.b = b; ca.Vst = ts.Vst; ca.taus = ts.taus2; ca.KMAX = ts.KMAX;
    const int tmax = 2 * (n - 3) + ts.KMAX;
    for (int t = 0; t <= tmax && n >= 3; ++t) {
        // s <= t / 2, s <= n - 3, s + 1 + (t - 2 s) b <= n - 1
        int smax = std::min(n - 3, t / 2);
        long num = (long)t * b - n + 2;
        int smin = num > 0 ? (int)((num + 2 * b - 2) / (2 * b - 1)) : 0;
        if (smin > smax) continue;
        ca.t = t; ca.smin = smin; ca.count = smax - smin + 1;
        hipLaunchKernelGGL(ts_chase_kernel, dim3(ca.count), dim3(256), 0, c->stream, ca);
    }
    HIPCHK(hipGetLastError());
    double* dvec = W.vec + (size_t)V_D * ld;
    double* evec = W.vec + (size_t)V_E * ld;
    hipLaunchKernelGGL(ts_diag_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, AB, LDB, n, dvec, evec);
    HIPCHK(hipGetLastError());
    if (want_vectors) {
        SCHK(scratch_get(c, SCR_R, ((size_t)ts.ngroups * ts.KMAX * TS_BMAX * TS_BMAX + 64) * sizeof(double), &ts.Tst));
        hipLaunchKernelGGL(ts_q2_tfactor_kernel, dim3(ts.ngroups, ts.KMAX), dim3(64), 0, c->stream, ts.Vst, ts.taus2, n, b, ts.G,
                           ts.KMAX, ts.Tst);
        HIPCHK(hipGetLastError());
    }
    d.assign(n, 0.0);
    e.assign(n, 0.0);
    {
        static_assert(V_E == V_D + 1, "slot order");
        void* st;
        SCHK(host_stage(c, ((size_t)ld + n) * sizeof(double), &st));
        const double* hd = static_cast<const double*>(st);
        HIPCHK(hipMemcpyAsync(st, dvec, ((size_t)ld + n) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        SCHK(stream_wait(c));
        std::copy(hd, hd + n, d.begin());
        std::copy(hd + ld, hd + ld + n, e.begin());
    }
    return SELLA_OK;
}

// X (rows: eigenvectors of the tridiagonal matrix) <- X Q2^T Q1^T
static int two_stage_back(EighWork& W, TwoStage& ts, double* X) {
    sella_ctx* c = W.c;
    const int n = ts.n, ld = ts.ld;
    if (n >= 3) {
        hipLaunchKernelGGL(ts_q2_apply_kernel, dim3((n + 15) / 16), dim3(256), 0, c->stream, X, ld, n, ts.b, ts.G, ts.ngroups, ts.Vst,
                           ts.taus2, ts.KMAX, ts.Tst);
        HIPCHK(hipGetLastError());
    }
    if (ts.npanels > 0) {
        prof_begin(c, PROF_OTHER, 0.0, 2.0 * n * (double)n * n);
        SELLA_LAUNCH(c, wy_apply_mfma_kernel<4>, dim3((n + 15) / 16), dim3(256), 0, X, ld, n, ts.Ystore, ts.Cstore, ts.npanels);
        prof_end(c);
        HIPCHK(hipGetLastError());
    }
    return SELLA_OK;
}

}  // namespace sella
