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
            if (tid == 0) taus[j] = 0.0;                        // (the row of Yt stays zero: wy_tinv64_kernel expects that)
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
            if (i == j) { yrow[i] = tau != 0.0 ? 1.0 : 0.0; row[i] = beta; }
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

// ---- the same factorisation with the panel held in registers -----------------------------------------------------------
// One workgroup cannot stream the panel from L2 once per column fast enough (a CU draws ~150 GB/s: 32 x 31 row passes of
// 8 m bytes each).  Here thread t owns the entries i = t + 512 v of every row; SB rows at a time live in registers and are
// factored there (one block-wide sum per column: norm, head and the cross products with the other SB - 1 rows at once),
// then the SB reflectors go over the remaining rows as one block (I - Y T^T Y^T, T of order SB from the Gram products of
// the sub-panel), RB rows per block-wide sum.  Memory traffic per panel: (b / SB) / 2 row passes instead of b.
template <int CTRL>
__device__ __forceinline__ double ts_dpp_get(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// One halving step of a transposing wavefront reduction: H values per lane -> H / 2, each the sum of the lane's and its
// partner's value; the lower lane of a pair (SIDE bit clear) keeps the first half.
template <int CTRL, int H>
__device__ __forceinline__ void ts_halve(double* v, bool side) {
#pragma unroll
    for (int k = 0; k < H / 2; ++k) {
        const double keep = side ? v[k + H / 2] : v[k];
        const double send = side ? v[k] : v[k + H / 2];
        v[k] = keep + ts_dpp_get<CTRL>(send);
    }
}

// Block-wide sums of K values per thread (K = 4, 8 or 16; all wavefronts of the workgroup, 64 lanes each): K log-steps of
// DPP moves would be 6 K of them; transposed, the values are dealt to the lanes while they are added — K - 1 moves inside
// the rows of 16 and two exchanges across them.  Lane l ends up with value ts_sum_index<K>(l).
// (The steps run from the widest exchange to the narrowest: the mirrors flip every lower lane bit as well, so they can only
// pair lanes that still hold the same set of values.)
template <int K>
__device__ __forceinline__ int ts_sum_index(int lane) {
    const int b3 = (lane >> 3) & 1, b2 = (lane >> 2) & 1, b1 = (lane >> 1) & 1, b0 = lane & 1;
    return K == 16 ? 8 * b3 + 4 * b2 + 2 * b1 + b0 : (K == 8 ? 4 * b3 + 2 * b2 + b1 : 2 * b3 + b2);
}

template <int K>
__device__ __forceinline__ void ts_block_sums(double (&v)[K], double (*red)[16], double* tot, int nwaves) {
    static_assert(K == 4 || K == 8 || K == 16, "ts_block_sums: 4, 8 or 16 values");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double t;
    if constexpr (K == 16) {
        ts_halve<0x140, 16>(v, lane & 8);
        ts_halve<0x141, 8>(v, lane & 4);
        ts_halve<0x4E, 4>(v, lane & 2);
        ts_halve<0xB1, 2>(v, lane & 1);
        t = v[0];
    } else if constexpr (K == 8) {
        ts_halve<0x140, 8>(v, lane & 8);
        ts_halve<0x141, 4>(v, lane & 4);
        ts_halve<0x4E, 2>(v, lane & 2);
        t = v[0];
        t += ts_dpp_get<0xB1>(t);
    } else {
        ts_halve<0x140, 4>(v, lane & 8);
        ts_halve<0x141, 2>(v, lane & 4);
        t = v[0];
        t += ts_dpp_get<0x4E>(t);
        t += ts_dpp_get<0xB1>(t);
    }
    t += __shfl_xor(t, 16);
    t += __shfl_xor(t, 32);
    if (lane < 16) red[wave][ts_sum_index<K>(lane)] = t;
    __syncthreads();
    if (threadIdx.x < K) {
        double sum = 0.0;
        for (int w = 0; w < nwaves; ++w) sum += red[w][threadIdx.x];
        tot[threadIdx.x] = sum;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = tot[k];
}

#if defined(__HIP_DEVICE_COMPILE__)
#define TS_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define TS_SCHED_FENCE() ((void)0)
#endif

template <int VPT, int SB, bool KEEP>
__global__ __launch_bounds__(512) void ts_panel_qr_reg_kernel(double* __restrict__ A, int ld, int n, int p, int b,
                                                              double* __restrict__ Yt, double* __restrict__ taus) {
    constexpr int NT = 512, NW = 8;
    constexpr int NV = SB == 4 ? 8 : 4;                       // norm, head and (cross product, head) of the other SB - 1 rows
    __shared__ double red[NW][16], tot[16];
    const int tid = threadIdx.x;
    const int r0 = p + b, m = n - r0;
    double x[SB][VPT];                                          // the sub-panel: loaded once, then handed on by the block update
#pragma unroll
    for (int q = 0; q < SB; ++q) {
        const double* row = A + (size_t)(p + q) * ld + r0;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const int i = tid + NT * v;
            x[q][v] = i < m ? row[i] : 0.0;
        }
    }
    for (int j0 = 0; j0 < b; j0 += SB) {
        double tau[SB];
        // ---- factor the sub-panel in registers
#pragma unroll
        for (int jj = 0; jj < SB; ++jj) {
            const int j = j0 + jj;
            double vals[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) vals[k] = 0.0;
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const int i = tid + NT * v;
                const double xj = x[jj][v];
                if (i > j) {
                    vals[0] += xj * xj;
#pragma unroll
                    for (int q = jj + 1; q < SB; ++q) vals[2 + 2 * (q - jj - 1)] += xj * x[q][v];
                } else if (i == j) {
                    vals[1] = xj;
#pragma unroll
                    for (int q = jj + 1; q < SB; ++q) vals[3 + 2 * (q - jj - 1)] = x[q][v];
                }
            }
            ts_block_sums<NV>(vals, red, tot, NW);
            double tj, beta, scale;
            ts_larfg(vals[1], vals[0], &tj, &beta, &scale);
            if (j >= m) tj = 0.0;
            tau[jj] = tj;
            double dq[SB];
#pragma unroll
            for (int q = jj + 1; q < SB; ++q) dq[q] = tj * (vals[3 + 2 * (q - jj - 1)] + scale * vals[2 + 2 * (q - jj - 1)]);
            double* row = A + (size_t)(p + j) * ld + r0;
            double* yrow = Yt + (size_t)j * ld + r0;
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const int i = tid + NT * v;
                const double y = i > j ? x[jj][v] * scale : ((i == j && tj != 0.0) ? 1.0 : 0.0);
#pragma unroll
                for (int q = jj + 1; q < SB; ++q) x[q][v] -= dq[q] * y;
                if (i < m) {
                    row[i] = i < j ? x[jj][v] : (i == j ? beta : 0.0);  // left of the diagonal: R, final after this sub-panel's reflectors
                    yrow[i] = y;
                }
                x[jj][v] = y;
            }
            if (tid == 0) taus[j] = tj;
        }
        if (j0 + SB >= b) break;
        // ---- T of the sub-panel (dlarft) from its Gram products, by every thread
        double T[SB][SB];
        {
            constexpr int NG = SB == 4 ? 8 : 4;               // SB (SB - 1) / 2 products, padded to the reduction's sizes
            double g[NG];
#pragma unroll
            for (int k = 0; k < NG; ++k) g[k] = 0.0;
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                int k = 0;
#pragma unroll
                for (int l = 0; l < SB; ++l)
#pragma unroll
                    for (int q = l + 1; q < SB; ++q) g[k++] += x[l][v] * x[q][v];
            }
            ts_block_sums<NG>(g, red, tot, NW);
            double G[SB][SB];
            {
                int k = 0;
#pragma unroll
                for (int l = 0; l < SB; ++l)
#pragma unroll
                    for (int q = l + 1; q < SB; ++q) G[l][q] = g[k++];
            }
#pragma unroll
            for (int q = 0; q < SB; ++q) {
#pragma unroll
                for (int l = 0; l < SB; ++l) T[l][q] = 0.0;
                T[q][q] = tau[q];
#pragma unroll
                for (int l = 0; l < q; ++l) {
                    double acc = 0.0;
#pragma unroll
                    for (int l2 = l; l2 < q; ++l2) acc += T[l][l2] * G[l2][q];
                    T[l][q] = -tau[q] * acc;
                }
            }
        }
        // ---- the block of SB reflectors over the remaining rows, SB rows per block-wide sum.  The first SB of them are the
        // next sub-panel: they stay in registers (their memory image is rewritten when their own reflectors are formed); the
        // rows of the following batch are fetched while the current one is being reduced (PF: when the registers allow).
        // KEEP = false (the largest panels): the rows are not held between the two passes but read again for the update.
        constexpr bool PF = KEEP && VPT * SB <= 16;
        constexpr int XS = KEEP ? SB : 1, XV = KEEP ? VPT : 1;
        double xr[XS][XV], xn[PF ? SB : 1][PF ? VPT : 1];
        auto load_rows = [&](int jp, double (&dst)[XS][XV]) {
            if constexpr (KEEP) {
#pragma unroll
                for (int q = 0; q < SB; ++q) {
                    const double* row = A + (size_t)(p + jp + q) * ld + r0;
#pragma unroll
                    for (int v = 0; v < VPT; ++v) {
                        const int i = tid + NT * v;
                        dst[q][v] = i < m ? row[i] : 0.0;
                    }
                }
            }
        };
        // batches in the order 2nd, 3rd, ..., last, 1st: the result of the 1st goes straight into x, which nobody needs any more
        const int nbatch = (b - j0 - SB) / SB;
        auto batch_row = [&](int bi) { return bi < nbatch ? j0 + SB + bi * SB : j0 + SB; };
        // one batch: products with the SB reflectors, block-wide sums, update; LAST (the 1st batch, taken last): into x
        auto batch = [&](int jp, auto last_tag) {
            constexpr bool LAST = decltype(last_tag)::value;
            double d[SB * SB];
#pragma unroll
            for (int k = 0; k < SB * SB; ++k) d[k] = 0.0;
#pragma unroll
            for (int q = 0; q < SB; ++q) {
                const double* row = A + (size_t)(p + jp + q) * ld + r0;
#pragma unroll
                for (int v = 0; v < VPT; ++v) {
                    const int i = tid + NT * v;
                    double xv;
                    if constexpr (KEEP) xv = xr[q][v];
                    else xv = i < m ? row[i] : 0.0;
#pragma unroll
                    for (int l = 0; l < SB; ++l) d[q * SB + l] += x[l][v] * xv;
                }
            }
            ts_block_sums<SB * SB>(d, red, tot, NW);
            double w[SB][SB];
#pragma unroll
            for (int q = 0; q < SB; ++q)
#pragma unroll
                for (int l = 0; l < SB; ++l) {
                    double acc = 0.0;
#pragma unroll
                    for (int l2 = 0; l2 <= l; ++l2) acc += T[l2][l] * d[q * SB + l2];
                    w[q][l] = acc;
                }
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const int i = tid + NT * v;
                double acc[SB];
#pragma unroll
                for (int q = 0; q < SB; ++q) {
                    if constexpr (KEEP) acc[q] = xr[q][v];
                    else acc[q] = i < m ? A[(size_t)(p + jp + q) * ld + r0 + i] : 0.0;
#pragma unroll
                    for (int l = 0; l < SB; ++l) acc[q] -= x[l][v] * w[q][l];
                }
#pragma unroll
                for (int q = 0; q < SB; ++q) {
                    if constexpr (LAST) x[q][v] = acc[q];
                    else if (i < m) A[(size_t)(p + jp + q) * ld + r0 + i] = acc[q];
                }
            }
        };
        load_rows(batch_row(1), xr);
        for (int bi = 1; bi < nbatch; ++bi) {
            if constexpr (PF) load_rows(batch_row(bi + 1), reinterpret_cast<double (&)[XS][XV]>(xn));
            batch(batch_row(bi), std::false_type());
            if constexpr (PF) {
#pragma unroll
                for (int q = 0; q < SB; ++q)
#pragma unroll
                    for (int v = 0; v < VPT; ++v) xr[q][v] = xn[q][v];
            } else {
                TS_SCHED_FENCE();                                // keep the next batch's loads behind this batch's stores (registers)
                load_rows(batch_row(bi + 1), xr);
            }
        }
        batch(j0 + SB, std::true_type());
    }
}

// Partial products over 64-column chunks of the trailing index: part[chunk][0] = Yt Yt^T, part[chunk][1] = Yt Zt^T (32 x 32
// each).  grid = chunks.
__global__ __launch_bounds__(256) void ts_gram2_kernel(const double* __restrict__ Yt, const double* __restrict__ Zt, int ld, int m,
                                                       double* __restrict__ part) {
    __shared__ double sY[TS_BMAX][65], sZ[TS_BMAX][65];
    const int tid = threadIdx.x, c0 = blockIdx.x * 64;
    for (int e = tid; e < TS_BMAX * 64; e += 256) {
        const int l = e >> 6, cc = e & 63;
        const bool ok = c0 + cc < m;
        sY[l][cc] = ok ? Yt[(size_t)l * ld + c0 + cc] : 0.0;
        sZ[l][cc] = ok ? Zt[(size_t)l * ld + c0 + cc] : 0.0;
    }
    __syncthreads();
    const int l = tid >> 3, q0 = (tid & 7) * 4;
    double g[4] = {0.0, 0.0, 0.0, 0.0}, kk[4] = {0.0, 0.0, 0.0, 0.0};
    for (int cc = 0; cc < 64; ++cc) {
        const double yl = sY[l][cc];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            g[q] += yl * sY[q0 + q][cc];
            kk[q] += yl * sZ[q0 + q][cc];
        }
    }
    double* out = part + (size_t)blockIdx.x * 2 * TS_BMAX * TS_BMAX;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        out[l * TS_BMAX + q0 + q] = g[q];
        out[TS_BMAX * TS_BMAX + l * TS_BMAX + q0 + q] = kk[q];
    }
}

// One workgroup: sums the partial products, T (dlarft) and C = T^T of the panel, S = (T^T K T + its transpose) / 4
// (K = Y^T A Y: the symmetric core of the two-sided update, W = A Y T - Y S).
__global__ __launch_bounds__(1024) void ts_panel_small_kernel(const double* __restrict__ part, int nparts, const double* __restrict__ taus,
                                                              double* __restrict__ T, double* __restrict__ C, double* __restrict__ S) {
    __shared__ double sG[TS_BMAX][TS_BMAX + 1], sK[TS_BMAX][TS_BMAX + 1], sT[TS_BMAX][TS_BMAX + 1], sM[TS_BMAX][TS_BMAX + 1];
    const int tid = threadIdx.x, i = tid >> 5, j = tid & 31;
    {
        double g = 0.0, k = 0.0;
        for (int pp = 0; pp < nparts; ++pp) {
            g += part[(size_t)pp * 2 * TS_BMAX * TS_BMAX + tid];
            k += part[(size_t)pp * 2 * TS_BMAX * TS_BMAX + TS_BMAX * TS_BMAX + tid];
        }
        sG[i][j] = g;
        sK[i][j] = k;
        sT[i][j] = 0.0;
    }
    __syncthreads();
    // dlarft: T[r][q] = -tau_q sum_{l = r}^{q - 1} T[r][l] G[l][q] — row r of T depends on row r only: a thread per row, no exchange
    if (tid < TS_BMAX) {
        const int r = tid;
        double trow[TS_BMAX];
#pragma unroll
        for (int q = 0; q < TS_BMAX; ++q) {
            double acc = 0.0;
#pragma unroll
            for (int l = 0; l < q; ++l) acc += (l >= r ? trow[l] : 0.0) * sG[l][q];
            trow[q] = q < r ? 0.0 : (q == r ? taus[q] : -taus[q] * acc);
        }
#pragma unroll
        for (int q = 0; q < TS_BMAX; ++q) sT[r][q] = trow[q];
    }
    __syncthreads();
    T[tid] = sT[i][j];
    C[tid] = sT[j][i];
    {
        double acc = 0.0;                                       // M = K T
        for (int l = 0; l <= j; ++l) acc += sK[i][l] * sT[l][j];
        sM[i][j] = acc;
    }
    __syncthreads();
    double s0;
    {
        double acc = 0.0;                                       // T^T M
        for (int l = 0; l <= i; ++l) acc += sT[l][i] * sM[l][j];
        s0 = acc;
    }
    __syncthreads();
    sG[i][j] = s0;
    __syncthreads();
    S[tid] = 0.25 * (sG[i][j] + sG[j][i]);
}

// Wt = T^T Zt - S Yt on the trailing columns: a thread per column and quarter of the 32 output rows.  grid = ceil(m / 64).
__global__ __launch_bounds__(256) void ts_wt_kernel(const double* __restrict__ T, const double* __restrict__ S,
                                                    const double* __restrict__ Yt, const double* __restrict__ Zt, int ld, int m,
                                                    double* __restrict__ Wt) {
    __shared__ double sT[TS_BMAX][TS_BMAX], sS[TS_BMAX][TS_BMAX];
    const int tid = threadIdx.x;
    for (int e = tid; e < TS_BMAX * TS_BMAX; e += 256) { sT[e >> 5][e & 31] = T[e]; sS[e >> 5][e & 31] = S[e]; }
    __syncthreads();
    const int cc = blockIdx.x * 64 + (tid & 63), jq = (tid >> 6) * 8;
    if (cc >= m) return;
    double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int l = 0; l < TS_BMAX; ++l) {
        const double z = Zt[(size_t)l * ld + cc], y = Yt[(size_t)l * ld + cc];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += sT[l][jq + q] * z - sS[jq + q][l] * y;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) Wt[(size_t)(jq + q) * ld + cc] = acc[q];
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

// Partial Gram matrices of 64 consecutive reflector rows of stage 1 (two panels) for the 64-reflector blocks of the
// back-transformation (wy_tinv64_kernel / wy_apply_mfma64_kernel of the one-stage path): grid (blocks, S) as wy_gram64_kernel.
__global__ __launch_bounds__(256) void ts_gram64_kernel(const double* __restrict__ Yf, int ld, int n, double* __restrict__ G) {
    __shared__ double Ys[64][65];
    const int b = blockIdx.x, j0 = b * 64;
    const int S = gridDim.y, sl = blockIdx.y;
    const int p = threadIdx.x >> 2, q0 = (threadIdx.x & 3) * 16;
    double acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.0;
    for (int ct = j0 + 64 * sl; ct < n; ct += 64 * S) {
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * 64; e += 256) {
            const int r = e >> 6, cc = e & 63;
            Ys[r][cc] = ct + cc < n ? Yf[(size_t)(j0 + r) * ld + ct + cc] : 0.0;
        }
        __syncthreads();
#pragma unroll 4
        for (int cc = 0; cc < 64; ++cc) {
            const double yp = Ys[p][cc];
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] += yp * Ys[q0 + k][cc];
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) G[(((size_t)b * S + sl) * 64 + p) * 64 + q0 + k] = acc[k];
}

// ---- stage 2 ---------------------------------------------------------------------------------------------------------
struct ChaseArgs {
    double* AB; int LDB, n, b;
    int t, smin, count;                        // tasks (s, k = t - 2 s), s = smin .. smin + count - 1
    double* Vst; double* taus; int KMAX;       // reflector of task (s, k): Vst[(s * KMAX + k) * b ..], taus[s * KMAX + k]
};

// One task of the bulge chase (see the file header).  r = s + 1 + k b, rows / columns J = [r, r + L), L = min(b, n - r).
//   k = 0: reflector from column s below the diagonal; k > 0: E = B[J, J - b] <- E H_{k-1}, reflector from its first column,
//   E <- H E;  then D = B[J, J] <- H D H.
// One workgroup of 256 threads, both blocks in registers: thread (ii = tid / 8, cg = tid % 8) holds columns 4 cg .. 4 cg + 3
// of row ii of E and of D.  Sums along a row stay inside eight neighbouring lanes; sums down a column are three exchanges
// inside the wavefront (eight rows) and one through LDS (four wavefronts).  Two workgroup barriers per task: the task is the
// unit of the critical path (2 n of them in a row), so its own depth is what counts.
__device__ __forceinline__ double ts_sum8(double v) {
    v = wave_dpp_add<0xB1>(v);
    v = wave_dpp_add<0x4E>(v);
    return wave_dpp_add<0x141>(v);
}

__global__ __launch_bounds__(256) void ts_chase_kernel(ChaseArgs a) {
    __shared__ double sx[TS_BMAX], sw[TS_BMAX], part[4][TS_BMAX];
    const int tid = threadIdx.x, ii = tid >> 3, cg = tid & 7, wave = tid >> 6;
    const int s = a.smin + blockIdx.x, k = a.t - 2 * s;
    const int b = a.b, n = a.n, LDB = a.LDB;
    const int r = s + 1 + k * b, rp = r - b;
    const int L = (n - r < b) ? n - r : b;
    double* AB = a.AB;
    const bool rowok = ii < L;
    double e[4], dd[4], vpq[4];
    double taup = 0.0, x0 = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = 4 * cg + q;
        e[q] = (k > 0 && rowok) ? AB[(size_t)(rp + c) * LDB + b + ii - c] : 0.0;
        dd[q] = (rowok && c < L) ? (ii >= c ? AB[(size_t)(r + c) * LDB + ii - c] : AB[(size_t)(r + ii) * LDB + c - ii]) : 0.0;
        vpq[q] = k > 0 ? a.Vst[((size_t)s * a.KMAX + (k - 1)) * b + c] : 0.0;
    }
    if (k > 0) {
        taup = a.taus[(size_t)s * a.KMAX + (k - 1)];
        // (a) E <- E (I - taup vp vp^T)
        const double wv = taup * ts_sum8(e[0] * vpq[0] + e[1] * vpq[1] + e[2] * vpq[2] + e[3] * vpq[3]);
#pragma unroll
        for (int q = 0; q < 4; ++q) e[q] -= wv * vpq[q];
        x0 = e[0];                                                // cg == 0: the column the new reflector annihilates
    } else if (cg == 0 && rowok) {
        x0 = AB[(size_t)s * LDB + 1 + ii];                        // column s below the diagonal
    }
    if (cg == 0) sx[ii] = x0;
    __syncthreads();
    // ---- reflector: every thread from the same 32 numbers in the same order
    double xn2 = 0.0;
    for (int i = 1; i < TS_BMAX; ++i) xn2 += sx[i] * sx[i];
    double tau, beta, scale;
    ts_larfg(sx[0], xn2, &tau, &beta, &scale);
    const double vi = ii == 0 ? 1.0 : sx[ii] * scale;
    double vc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) vc[q] = (4 * cg + q == 0) ? 1.0 : sx[4 * cg + q] * scale;
    // (b) column sums of v^T E, (c) w = tau D v
    double pc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double t = vi * e[q];
        t += __shfl_xor(t, 8);
        t += __shfl_xor(t, 16);
        t += __shfl_xor(t, 32);
        pc[q] = t;
    }
    if ((tid & 63) < 8) {
#pragma unroll
        for (int q = 0; q < 4; ++q) part[wave][4 * cg + q] = pc[q];
    }
    double wi = tau * ts_sum8(dd[0] * vc[0] + dd[1] * vc[1] + dd[2] * vc[2] + dd[3] * vc[3]);
    if (cg == 0) sw[ii] = wi;
    __syncthreads();
    if (k > 0) {
        if (tau != 0.0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = 4 * cg + q;
                const double cs = tau * ((part[0][c] + part[1][c]) + (part[2][c] + part[3][c]));
                e[q] -= vi * cs;
            }
        }
        if (cg == 0) e[0] = ii == 0 ? beta : 0.0;                 // first column set exactly
        if (rowok) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = 4 * cg + q;
                AB[(size_t)(rp + c) * LDB + b + ii - c] = e[q];
            }
        }
    } else if (cg == 0 && rowok) {
        AB[(size_t)s * LDB + 1 + ii] = ii == 0 ? beta : 0.0;
    }
    // (c) D <- H D H = D - v w^T - w v^T,  w <- w - (tau (w . v) / 2) v
    if (tau != 0.0) {
        double pv = sw[0];
        for (int i = 1; i < TS_BMAX; ++i) pv += sw[i] * (sx[i] * scale);
        const double al = -0.5 * tau * pv;
        wi += al * vi;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = 4 * cg + q;
            const double wc = sw[c] + al * vc[q];
            if (rowok && c <= ii) AB[(size_t)(r + c) * LDB + ii - c] = dd[q] - (vi * wc + wi * vc[q]);
        }
    }
    if (cg == 0) a.Vst[((size_t)s * a.KMAX + k) * b + ii] = rowok ? vi : 0.0;
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
// X (rows = eigenvectors) <- X Q2^T.  Q2 = prod over sweeps s ascending, steps k ascending, of H(s, k).  Inside a group of
// G = b consecutive sweeps (s0 = g G) the factors commute into  prod_{k descending} W_k,  W_k = prod_{s ascending} H(s, k)
// = I - V T V^T  (H(s, k) and H(s', k + 1) overlap only for s' < s, and then the generation order already has (s', k + 1)
// first), so  X Q2^T = X prod_{g descending} prod_{k ascending} W_{g,k}^T,  W^T = I - V T^T V^T = I - V (V T)^T.
// Block (g, k) lives on the 64-column window starting at column 32 (g + k) (its first column is untouched): row cc of V
// holds entry cc - i - 1 of v(s0 + i, k).
//
// Matrix-core form, on the TRANSPOSE of a 16-row slab of X:   M^T = V^T Xc^T (32 x 16),   Xc^T -= (V T) M^T (64 x 16).
// A tile of Xc^T in the accumulator layout of v_mfma_f64_16x16x4 is at the same time the B operand of the first product
// (the summation index of a tile product may be permuted: slot lg of step r <-> the tile row the lane holds in register r),
// and so is M^T for the second: the slab goes through a block without leaving the registers, and the only loads are the A
// operands — V^T and -(V T), which ts_q2_pack_kernel lays out once, fragment by fragment in exactly the lane order the
// products consume them (13 non-zero 16 x 16 tiles per block: 6 of V^T, 7 of V T; one 32-byte load per lane and tile).
constexpr int TS_Q2_TILES = 13;
//                                   V^T tiles (mt, ct)                            V T tiles (ct, mt)
__device__ const signed char ts_q2_t1[6][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 1}, {1, 2}, {1, 3}};
__device__ const signed char ts_q2_t3[7][2] = {{0, 0}, {1, 0}, {2, 0}, {0, 1}, {1, 1}, {2, 1}, {3, 1}};

// index of block (g, k) in the packed stream: groups in order, kcount(g) = K0 - g blocks each
__host__ __device__ inline size_t ts_q2_block(int g, int k, int K0) { return (size_t)g * K0 - (size_t)g * (g - 1) / 2 + k; }

// grid (K0, ngroups): block (g, k = blockIdx.x) if k < K0 - g.  T as for dlarft from the Gram matrix of the block's columns.
__global__ __launch_bounds__(256) void ts_q2_pack_kernel(const double* __restrict__ Vst, const double* __restrict__ taus, int n,
                                                         int KMAX, int K0, double* __restrict__ F) {
    __shared__ double sV[2 * TS_BMAX][TS_BMAX + 1];          // window rows x reflectors
    __shared__ double sU[2 * TS_BMAX][TS_BMAX + 1];          // V T
    __shared__ double sG[TS_BMAX][TS_BMAX + 1], sT[TS_BMAX][TS_BMAX + 1], st[TS_BMAX];
    constexpr int b = TS_BMAX;
    const int g = blockIdx.y, k = blockIdx.x, tid = threadIdx.x;
    if (k >= K0 - g) return;
    const int s0 = g * b;
    for (int e = tid; e < 2 * b * b; e += 256) sV[e / b][e % b] = 0.0;
    for (int e = tid; e < b * b; e += 256) sT[e / b][e % b] = 0.0;
    __syncthreads();
    for (int e = tid; e < b * b; e += 256) {
        const int i = e / b, q = e % b;
        const int s = s0 + i, r = s + 1 + k * b;
        if (s <= n - 3 && r <= n - 1) sV[i + 1 + q][i] = Vst[((size_t)s * KMAX + k) * b + q];
    }
    if (tid < b) {
        const int s = s0 + tid, r = s + 1 + k * b;
        st[tid] = (s <= n - 3 && r <= n - 1) ? taus[(size_t)s * KMAX + k] : 0.0;
    }
    __syncthreads();
    for (int e = tid; e < b * b; e += 256) {
        const int i = e / b, j = e % b;
        double acc = 0.0;
        for (int cc = 0; cc < 2 * b; ++cc) acc += sV[cc][i] * sV[cc][j];
        sG[i][j] = acc;
    }
    __syncthreads();
    for (int j = 0; j < b; ++j) {
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
    for (int e = tid; e < 2 * b * b; e += 256) {
        const int cc = e / b, j = e % b;
        double acc = 0.0;
        for (int i = 0; i <= j; ++i) acc += sV[cc][i] * sT[i][j];
        sU[cc][j] = acc;
    }
    __syncthreads();
    double* out = F + ts_q2_block(g, k, K0) * (TS_Q2_TILES * 256);
    for (int e = tid; e < TS_Q2_TILES * 256; e += 256) {
        const int tile = e >> 8, lane = (e >> 2) & 63, r = e & 3;
        const int li = lane & 15, lg = lane >> 4;
        double val;
        if (tile < 6) {
            const int mt = ts_q2_t1[tile][0], ct = ts_q2_t1[tile][1];
            val = sV[16 * ct + 4 * lg + r][16 * mt + li];
        } else {
            const int ct = ts_q2_t3[tile - 6][0], mt = ts_q2_t3[tile - 6][1];
            val = -sU[16 * ct + 4 * (li & 3) + (li >> 2)][16 * mt + lg + 4 * r];
        }
        out[e] = val;
    }
}

typedef double ts_f64x4 __attribute__((ext_vector_type(4)));

// four consecutive entries of a row of X: fetched from a clamped address with no branch (the load stays in flight), masked
// to the matrix when they are consumed
__device__ __forceinline__ double4 ts_q2_fetch(const double* xrow, int col, int n) {
    return *reinterpret_cast<const double4*>(xrow + (col < n ? col : 0));   // col is a multiple of 4, rows are padded to 8
}
__device__ __forceinline__ ts_f64x4 ts_q2_mask(double4 t, int col, int n) {
    ts_f64x4 v;
    v[0] = col < n ? t.x : 0.0;
    v[1] = col + 1 < n ? t.y : 0.0;
    v[2] = col + 2 < n ? t.z : 0.0;
    v[3] = col + 3 < n ? t.w : 0.0;
    return v;
}
__device__ __forceinline__ ts_f64x4 ts_q2_load(const double* xrow, int col, int n) { return ts_q2_mask(ts_q2_fetch(xrow, col, n), col, n); }

// One wavefront per 16 rows of X, through every block in the order above; the window slides by two tiles per block, so every
// entry of the slab is loaded and stored once per group.
__global__ __launch_bounds__(64) void ts_q2_apply_mfma_kernel(double* __restrict__ X, int ldx, int n, int ngroups, int K0,
                                                              const double* __restrict__ F) {
    const int lane = threadIdx.x, li = lane & 15, lg = lane >> 4;
    const int row0 = blockIdx.x * 16;
    const bool live = row0 + li < n;
    double* xrow = X + (size_t)(live ? row0 + li : n - 1) * ldx;
    const int c4 = 4 * lg;
    for (int g = ngroups - 1; g >= 0; --g) {
        const int kcount = K0 - g;
        if (kcount <= 0) continue;
        const int base = TS_BMAX * g;
        ts_f64x4 x[4];
        x[0] = ts_q2_load(xrow, base + c4, n);
        x[1] = ts_q2_load(xrow, base + 16 + c4, n);
        x[2] = ts_q2_load(xrow, base + 32 + c4, n);
        x[3] = ts_q2_load(xrow, base + 48 + c4, n);
        const double4* f = reinterpret_cast<const double4*>(F + ts_q2_block(g, 0, K0) * (TS_Q2_TILES * 256)) + lane;
        double4 a[TS_Q2_TILES];
#pragma unroll
        for (int t = 0; t < TS_Q2_TILES; ++t) a[t] = f[t * 64];
        for (int k = 0; k < kcount; ++k) {
            const int cb = base + TS_BMAX * k;
            // operands of the next block and the two tiles the window gains there: in flight under this block's products
            f += TS_Q2_TILES * 64;
            double4 an[TS_Q2_TILES];
            const double4 xn2 = ts_q2_fetch(xrow, cb + 64 + c4, n), xn3 = ts_q2_fetch(xrow, cb + 80 + c4, n);
            if (k + 1 < kcount) {
#pragma unroll
                for (int t = 0; t < TS_Q2_TILES; ++t) an[t] = f[t * 64];
            } else {
#pragma unroll
                for (int t = 0; t < TS_Q2_TILES; ++t) an[t] = a[t];
            }
            ts_f64x4 m[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
#define TS_Q2_P1(T, MT, CT)                                                                 \
            m[MT] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[T].x, x[CT][0], m[MT], 0, 0, 0);     \
            m[MT] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[T].y, x[CT][1], m[MT], 0, 0, 0);     \
            m[MT] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[T].z, x[CT][2], m[MT], 0, 0, 0);     \
            m[MT] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[T].w, x[CT][3], m[MT], 0, 0, 0);
            TS_Q2_P1(0, 0, 0) TS_Q2_P1(3, 1, 1) TS_Q2_P1(1, 0, 1) TS_Q2_P1(4, 1, 2) TS_Q2_P1(2, 0, 2) TS_Q2_P1(5, 1, 3)
#undef TS_Q2_P1
#define TS_Q2_P3(T, CT, MT)                                                                 \
            x[CT] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[T].x, m[MT][0], x[CT], 0, 0, 0);     \
            x[CT] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[T].y, m[MT][1], x[CT], 0, 0, 0);     \
            x[CT] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[T].z, m[MT][2], x[CT], 0, 0, 0);     \
            x[CT] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[T].w, m[MT][3], x[CT], 0, 0, 0);
            TS_Q2_P3(6, 0, 0) TS_Q2_P3(7, 1, 0) TS_Q2_P3(8, 2, 0) TS_Q2_P3(12, 3, 1)
            TS_Q2_P3(9, 0, 1) TS_Q2_P3(10, 1, 1) TS_Q2_P3(11, 2, 1)
#undef TS_Q2_P3
            if (live) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const int col = cb + 16 * ct + c4;
                    if (col < n) *reinterpret_cast<double4*>(xrow + col) = make_double4(x[ct][0], x[ct][1], x[ct][2], x[ct][3]);
                }
            }
            x[0] = x[2];
            x[1] = x[3];
            x[2] = ts_q2_mask(xn2, cb + 64 + c4, n);
            x[3] = ts_q2_mask(xn3, cb + 80 + c4, n);
#pragma unroll
            for (int t = 0; t < TS_Q2_TILES; ++t) a[t] = an[t];
        }
        if (live) {
            const int cb = base + TS_BMAX * kcount;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int col = cb + 16 * ct + c4;
                if (col < n) *reinterpret_cast<double4*>(xrow + col) = make_double4(x[ct][0], x[ct][1], x[ct][2], x[ct][3]);
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
    int K0 = 0;                                              // steps of the first sweep group
    double *Vst = nullptr, *taus2 = nullptr, *Fq2 = nullptr; // stage 2: reflectors, packed blocks of the back-transformation
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
    SCHK(scratch_get(c, SCR_V, (nrefl1 + 64) * ld * sizeof(double), &ts.Ystore));
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
    HIPCHK(hipMemsetAsync(ts.Ystore, 0, (nrefl1 + 64) * ld * sizeof(double), c->stream));
    HIPCHK(hipMemsetAsync(small, 0, ((size_t)(ts.npanels + 1) * (2 * TS_BMAX * TS_BMAX + TS_BMAX)) * sizeof(double), c->stream));
    // ---- stage 1 ----------------------------------------------------------------------------------------------------
    double* gpart;
    const int maxparts = (n + 63) / 64;
    SCHK(scratch_get(c, SCR_T, ((size_t)maxparts * 2 * TS_BMAX * TS_BMAX + 64) * sizeof(double), &gpart));
    for (int ip = 0; ip < ts.npanels; ++ip) {
        const int p = ip * b, r0 = p + b, m = n - r0;
        double* Yt = ts.Ystore + (size_t)ip * b * ld;
        double* taus = ts.taus1 + (size_t)ip * TS_BMAX;
        double* T = Tall + (size_t)ip * TS_BMAX * TS_BMAX;
        double* C = ts.Cstore + (size_t)ip * TS_BMAX * TS_BMAX;
        const int vpt = (m + 511) / 512;
#define TS_QR(VPT, SB, KEEP) hipLaunchKernelGGL(HIP_KERNEL_NAME(ts_panel_qr_reg_kernel<VPT, SB, KEEP>), dim3(1), dim3(512), 0, c->stream, W.A, ld, n, p, b, Yt, taus)
        if (c->opt.eigh2_qr_reg == 0 || vpt > 24) hipLaunchKernelGGL(ts_panel_qr_kernel, dim3(1), dim3(1024), 0, c->stream, W.A, ld, n, p, b, Yt, taus);
        else if (c->opt.eigh2_qr_reg == 2) TS_QR(24, 2, false);       // tests: the two-row variant of the large sizes at any size
        else if (vpt <= 2) TS_QR(2, 4, true);
        else if (vpt <= 4) TS_QR(4, 4, true);
        else if (vpt <= 8) TS_QR(8, 4, true);
        else if (vpt <= 12) TS_QR(12, 2, true);
        else if (vpt <= 16) TS_QR(16, 2, true);
        else TS_QR(24, 2, false);
#undef TS_QR
        HIPCHK(hipGetLastError());
        // Zt = Yt A22 (rows): two 16-row passes over the trailing block
        double* A22 = W.A + (size_t)r0 * ld + r0;
        if (m > 1)
            for (int h = 0; h < b; h += 16)
                SCHK(launch_panel16(c, A22, m, m, ld, Yt + (size_t)h * ld + r0, 16, Zt + (size_t)h * ld + r0, ld));
        // Gram products, T, C = T^T and the symmetric core S; Wt = T^T Zt - S Yt; A22 <- A22 - Y W^T - W Y^T
        const int nparts = (m + 63) / 64;
        hipLaunchKernelGGL(ts_gram2_kernel, dim3(nparts), dim3(256), 0, c->stream, Yt + r0, m > 1 ? Zt + r0 : Yt + r0, ld, m, gpart);
        hipLaunchKernelGGL(ts_panel_small_kernel, dim3(1), dim3(1024), 0, c->stream, gpart, nparts, taus, T, C, Sm);
        HIPCHK(hipGetLastError());
        if (m <= 1) continue;
        hipLaunchKernelGGL(ts_wt_kernel, dim3(nparts), dim3(256), 0, c->stream, T, Sm, Yt + r0, Zt + r0, ld, m, Wt + r0);
        HIPCHK(hipGetLastError());
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
        ts.K0 = (n - 2) / b + 1;
        const size_t nblocks = ts_q2_block(ts.ngroups, 0, ts.K0);
        SCHK(scratch_get(c, SCR_R, (nblocks * TS_Q2_TILES * 256 + 64) * sizeof(double), &ts.Fq2));
        hipLaunchKernelGGL(ts_q2_pack_kernel, dim3(ts.K0, ts.ngroups), dim3(256), 0, c->stream, ts.Vst, ts.taus2, n, ts.KMAX, ts.K0,
                           ts.Fq2);
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
        hipLaunchKernelGGL(ts_q2_apply_mfma_kernel, dim3((n + 15) / 16), dim3(64), 0, c->stream, X, ld, n, ts.ngroups, ts.K0, ts.Fq2);
        HIPCHK(hipGetLastError());
    }
    if (ts.npanels > 0) {
        const int nrefl = ts.npanels * ts.b;
        const bool wy64 = c->opt.eigh_wy_nb64_min > 0 && n >= c->opt.eigh_wy_nb64_min;
        prof_begin(c, PROF_OTHER, 0.0, 2.0 * n * (double)n * n);
        if (wy64) {
            // two panels per compact-WY block: half the passes over the rows of X (see wy_apply_mfma64_kernel)
            const int nblk = (nrefl + 63) / 64;
            const int gsl = std::max(1, std::min(8, (256 + nblk - 1) / nblk));
            double* Gd;
            SCHK(scratch_get(c, SCR_EIG6, (size_t)nblk * (1 + gsl) * 64 * 64 * sizeof(double), &Gd));
            double* Gpart = Gd + (size_t)nblk * 64 * 64;
            hipLaunchKernelGGL(ts_gram64_kernel, dim3(nblk, gsl), dim3(256), 0, c->stream, ts.Ystore, ld, n, Gpart);
            hipLaunchKernelGGL(wy_tinv64_kernel, dim3(nblk), dim3(64), 0, c->stream, Gd, Gpart, gsl, nrefl, ts.taus1);
            SELLA_LAUNCH(c, wy_apply_mfma64_kernel<4>, dim3((n + 15) / 16), dim3(256), 0, X, ld, n, ts.Ystore, Gd, nblk);
        } else {
            SELLA_LAUNCH(c, wy_apply_mfma_kernel<4>, dim3((n + 15) / 16), dim3(256), 0, X, ld, n, ts.Ystore, ts.Cstore, ts.npanels);
        }
        prof_end(c);
        HIPCHK(hipGetLastError());
    }
    return SELLA_OK;
}

}  // namespace sella
