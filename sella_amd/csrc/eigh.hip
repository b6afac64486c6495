// eigh.hip — fp64 symmetric eigensolver on the device (replaces torch.linalg.eigh behind
// gpu_eigh / gpu_eigh_t, sella/_gpu.py:70-97; consumers: ApproximateHessian.evals/evecs
// linalg.py:174-231, the TS-BFGS |B| term hessian_update.py:121, the P-RFO split stepper.py:163).
//
// Three stages, all device-resident:
//   1. Blocked Householder tridiagonalisation  Q^T A Q = T  (panels of nb columns, LAPACK
//      dlatrd's algebra).  The full symmetric trailing matrix is kept, so the matrix-vector
//      product of each column is the row-panel matvec of the Davidson loop (coalesced 16-byte
//      streams, wave64 reductions).  Per column exactly TWO launches: `trd_row_kernel`
//      (finish the previous w, form the updated row, partial norms / panel dots) and
//      `trd_gemv_kernel` (reflector scalars from the partials, v staged on the fly, A22 v);
//      the rank-2nb trailing update is two GEMMs per panel.
//   2. Divide and conquer on T (Cuppen, with Gu/Eisenstat's stable eigenvectors): leaves by
//      implicit QL (one wavefront per leaf), merges = deflation (host, O(N log N)) + secular
//      equation (one wavefront per root) + one GEMM per merge; two host synchronisations per
//      tree level.
//   3. Back-transformation X <- X (I - Y T Y^T)^T block by block inside ONE kernel: every
//      workgroup owns 16 eigenvector rows and sweeps all reflector blocks (rows are independent).
// Eigenvectors are handled as ROWS of a row-major matrix throughout (vector-major, like the
// Krylov panels), so rotations, gathers and the final Q^T x products are all coalesced.
#include "internal.h"
#include "host_math.h"
#include "secular.h"

#include <algorithm>
#include <chrono>
#include <numeric>

namespace sella {
namespace {

constexpr int TRD_NBMAX = 64;              // max panel width
constexpr int TRD_PA = 1;                  // doubles per block in the K1 partial buffer (sum u^2)
constexpr int WY_NB = 32;                  // reflectors per compact-WY block

__device__ __forceinline__ double wave_sum_e(double v) { return wave_sum64(v); }

// block-wide sum for 256 threads; every thread gets the result
__device__ __forceinline__ double block_sum_256(double v, double* red /* >= 4 doubles LDS */) {
    v = wave_sum_e(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// ---------------------------------------------------------------------------------------
// stage 1 kernels
// ---------------------------------------------------------------------------------------
struct TrdRowArgs {
    double* A; int ld, n;
    int j;              // global column (row) index handled now
    int i;              // its local index in the panel; i == kb: only finish the last w
    int do_row;
    double* Vp; double* Wp; int ldp;
    const double* u_prev; double* u_cur;        // row buffers, absolute column index
    const double* wraw;                         // A22 v of column j-1, absolute row index
    const double* partA_prev; int nblkA_prev;   // K1 partials of column j-1
    double* partA_cur;
    const double* partB; int nblkB;             // v.wraw partials of column j-1
    const double* colscal;                      // {tau, scale} of column j-1
    const double* cdots;                        // W_p.v (p < i-1) at [p], V_p.v at [TRD_NBMAX + p] (from K2)
    double* dvec;
};

// K1.  Thread owns absolute column c = j + blockIdx.x*256 + tid.
//   (1) i > 0: finish w_{i-1} = tau (wraw - V c1 - W c2) + alpha2 v   (dlatrd), store in Wp
//   (2) u[c] = A[j][c] - sum_{p<i} (V_p[c] W_p[j] + W_p[c] V_p[j])    (pending rank-2i update)
//   (3) per-block partial of sum u^2 (c >= j+2)
// The kernel sits on the critical path of the factorisation (n dependent launches), so it is laid
// out for latency: every global load is issued before the first barrier, the panel is read once for
// both sums, and the only block-wide exchanges are the reduction of the v.wraw partials and of u^2.
// IPC >= 0: number of finished panel columns (i - 1) as a compile-time constant, so the panel loop
// is fully unrolled with all its loads in flight together; IPC < 0: generic loop (wide panels).
template <int IPC>
__global__ __launch_bounds__(256) void trd_row_kernel(TrdRowArgs a) {
    __shared__ double red[4];
    // all arguments in one batch of scalar loads (left to itself the compiler fetches them in four dependent stages, each
    // behind the branch that first needs it: -0.5 us per launch, 1.1 ms per eigh at 3N = 3072, session r05p; the same in
    // the matvec and in trd_upd_kernel measured equal / 0.2 us SLOWER per launch — their first loads start later)
    SELLA_ARG(a.A); SELLA_ARG(a.ld); SELLA_ARG(a.n); SELLA_ARG(a.j); SELLA_ARG(a.i); SELLA_ARG(a.do_row); SELLA_ARG(a.Vp);
    SELLA_ARG(a.Wp); SELLA_ARG(a.ldp); SELLA_ARG(a.u_cur); SELLA_ARG(a.wraw); SELLA_ARG(a.partA_cur); SELLA_ARG(a.partB);
    SELLA_ARG(a.nblkB); SELLA_ARG(a.colscal); SELLA_ARG(a.cdots); SELLA_ARG(a.dvec);
    const int tid = threadIdx.x;
    const int j = a.j, i = a.i, ldp = a.ldp;
    const int c = j + blockIdx.x * 256 + tid;
    const bool valid = c < a.n;
    const int cl = valid ? c : a.n - 1;                         // clamped: loads are always in range
    const int ip = i - 1;
    // ---- loads that do not depend on anything computed here.  Every batch is written as "load all,
    // then use" with clamped indices: a plain loop makes the compiler wait for each load in turn, and
    // this kernel is nothing but a chain of memory round trips.
    // the first 2048 partials of v.wraw: loaded unconditionally (clamped) and summed only where the sum is needed — a loop
    // around them would wait for them before anything else is issued (one more dependent round trip per launch)
    double pv0[8];
    {
        const int last = (a.nblkB > 0) ? a.nblkB - 1 : 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int b = tid + 256 * k;
            pv0[k] = a.partB[b < a.nblkB ? b : last];
        }
    }
    auto sum_partials = [&]() -> double {
        double vw = 0.0;
        if (IPC > 0 || i > 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) vw += (tid + 256 * k < a.nblkB) ? pv0[k] : 0.0;
            for (int b0 = tid + 256 * 8; b0 < a.nblkB; b0 += 256 * 8) {          // (more than 2048 workgroups in the matvec only)
                double pv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int b = b0 + 256 * k;
                    pv[k] = a.partB[b < a.nblkB ? b : a.nblkB - 1];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) vw += (b0 + 256 * k < a.nblkB) ? pv[k] : 0.0;
            }
        }
        return vw;
    };
    double vw = 0.0;
    const double arow = a.do_row ? a.A[(size_t)j * a.ld + cl] : 0.0;
    const double wrawc = (i > 0) ? a.wraw[cl] : 0.0;
    const double wrawj = (i > 0) ? a.wraw[j] : 0.0;
    const double vprev = (i > 0) ? a.Vp[(size_t)ip * ldp + cl] : 0.0;
    const double tau = (i > 0) ? a.colscal[0] : 0.0;
    // ---- one pass over the finished panel columns p < i-1
    double s = 0.0, q = 0.0, cc = 0.0, t = 0.0;
    const int np = (IPC >= 0) ? IPC : ip;
    bool vw_done = false;
    if (IPC > 3) {
        // Deep panels: the 4 uniform scalars per panel column (W_p.v, V_p.v, V_p[j], W_p[j]) no longer fit the
        // scalar register file (58 SGPRs spilled at depth 15, and the row kernel went from 4.4 to 7.9 us): they
        // are fetched once by 4 IPC lanes into LDS and read back as broadcasts, behind the barrier the v.wraw
        // reduction needs anyway; the per-column panel loads are issued before it.
        __shared__ double uni[4 * 32];
        constexpr int NPC = IPC > 0 ? IPC : 1;
        double vcs[NPC], wcs[NPC];
#pragma unroll
        for (int p = 0; p < NPC; ++p) {
            vcs[p] = a.Vp[(size_t)p * ldp + cl];
            wcs[p] = a.Wp[(size_t)p * ldp + cl];
        }
        if (tid < 4 * NPC) {
            const int p = tid >> 2, k = tid & 3;
            uni[tid] = (k == 0) ? a.cdots[p] : (k == 1) ? a.cdots[TRD_NBMAX + p]
                     : (k == 2) ? a.Vp[(size_t)p * ldp + j] : a.Wp[(size_t)p * ldp + j];
        }
        vw = block_sum_256(sum_partials(), red);                            // (i > 0 always holds here)
        vw_done = true;
#pragma unroll
        for (int p = 0; p < NPC; ++p) {
            const double c1 = uni[4 * p], c2 = uni[4 * p + 1], vj = uni[4 * p + 2], wj_p = uni[4 * p + 3];
            s += vcs[p] * c1 + wcs[p] * c2;
            q += vcs[p] * wj_p + wcs[p] * vj;
            cc += c1 * c2;
            t += vj * c1 + wj_p * c2;
        }
    } else {
#pragma unroll
        for (int p = 0; p < np; ++p) {
            const double c1 = a.cdots[p], c2 = a.cdots[TRD_NBMAX + p];          // W_p.v, V_p.v (uniform)
            const double vj = a.Vp[(size_t)p * ldp + j], wj_p = a.Wp[(size_t)p * ldp + j];
            const double vc = a.Vp[(size_t)p * ldp + cl], wcp = a.Wp[(size_t)p * ldp + cl];
            s += vc * c1 + wcp * c2;
            q += vc * wj_p + wcp * vj;
            cc += c1 * c2;
            t += vj * c1 + wj_p * c2;
        }
    }
    double wc = 0.0, u = 0.0;
    if (i > 0) {
        if (!vw_done) vw = block_sum_256(sum_partials(), red);
        const double alpha2 = -0.5 * tau * tau * (vw - 2.0 * cc);
        const double wj = tau * (wrawj - t) + alpha2;           // w_{i-1}[j], v_{i-1}[j] = 1
        wc = tau * (wrawc - s) + alpha2 * vprev;
        if (valid) a.Wp[(size_t)ip * ldp + c] = wc;
        u = arow - q - (vprev * wj + wc);                       // p = i-1: W[j] = wj, V[j] = 1
    } else {
        u = arow;
    }
    if (!a.do_row) return;
    if (valid) {
        a.u_cur[c] = u;
        if (c == j) a.dvec[j] = u;
    }
    double ss = (valid && c >= j + 2) ? u * u : 0.0;
    ss = block_sum_256(ss, red);
    if (tid == 0) a.partA_cur[(size_t)blockIdx.x * TRD_PA] = ss;
}

__device__ __forceinline__ double2 ldg2(const double* p) { return *reinterpret_cast<const double2*>(p); }

struct TrdGemvArgs {
    const double* A22;          // A + o*ld + oc   (oc = o rounded down to even: aligned 16-byte rows)
    int ld, m, shift, o, n, j;
    int pad;                    // dummy rows in front (see the launch: keeps row -> XCD fixed across columns)
    const double* ubuf;         // updated row j (absolute column index)
    const double* partA; int nblkA;
    double* wraw;               // absolute row index
    double* partB;
    double* Vrow;               // Vp + i*ldp
    double* Arow;               // A + j*ld (reflector tail stored for the back-transformation)
    double* taus; double* evec; double* colscal;
    const double* Wp; const double* Vp; int ldp, i;   // panel rows appended to the matvec: exact W_p.v, V_p.v
    double* cdots;
};

// K2.  wraw = A22 v for the reflector v = [1, scale*u] of column j (same block-cooperative streaming
// structure as gemv_rows_kernel<1, 2>).  The product is taken with the UNSCALED row u, which the
// previous kernel left in memory, and the reflector scalars (norm -> beta, tau, scale) are applied
// afterwards:  A22 v = scale * (A22 u') + A22[:, 0]  with u' = u without its leading entry — so the
// matrix stream starts at once instead of waiting for the norm.  The 2i panel rows W_p, V_p (p < i)
// are appended as extra rows of the same launch: a wavefront streams a whole row, so the dots the
// next column needs (dlatrd's W^T v, V^T v) come out exact, without partial buffers.
// NCH > 0: the rows are at most NCH chunks of 256 16-byte pieces long and EVERY load of the workgroup is issued before the
// first wait — clamped addresses, masked afterwards.  The loop form (NCH = 0) waits for its loads once per pass (the
// compiler cannot keep conditional loads in flight across the back edge): three dependent memory round trips per launch at
// 3N = 3072 on top of the one for the partials, in a kernel whose whole life is 5 - 12 us.  Same expressions in the same
// order: bit-identical results.
template <int NCH>
__global__ __launch_bounds__(256) void trd_gemv_kernel(TrdGemvArgs a) {
    __shared__ double red[4][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = blockIdx.x * 2 - a.pad;              // local row of this workgroup's first row (may be < 0)
    const int mtot = a.m + 2 * a.i;
    const int oc = a.o - a.shift;                     // even absolute column of local column 0
    const double2* arow[2];
    double lead[2];                                   // entry of each row in column o (v[o] = 1)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        int rr = row0 + r;
        if (rr < 0) rr = 0;
        if (rr > mtot - 1) rr = mtot - 1;
        const double* base;
        if (rr < a.m) base = a.A22 + (size_t)rr * a.ld;
        else if (rr < a.m + a.i) base = a.Wp + (size_t)(rr - a.m) * a.ldp + oc;
        else base = a.Vp + (size_t)(rr - a.m - a.i) * a.ldp + oc;
        arow[r] = reinterpret_cast<const double2*>(base);
        lead[r] = base[a.shift];
    }
    // K1 partials of sum u^2 (one per lane, reduced after the stream) and the entries of u this
    // workgroup's epilogue needs: issued now, consumed at the end
    double ssl0 = a.partA[(size_t)(lane < a.nblkA ? lane : a.nblkA - 1) * TRD_PA];     // consumed after the stream
    double ssl = 0.0;
    for (int b = lane + 64; b < a.nblkA; b += 64) ssl += a.partA[(size_t)b * TRD_PA];  // (n > 16384 only)
    const double alpha = a.ubuf[a.o];
    double urow[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int rr = row0 + r;
        urow[r] = a.ubuf[(rr >= 0 && rr < a.m) ? a.o + rr : a.o];
    }
    double acc[2] = {0.0, 0.0};
    const int n2 = (a.m + a.shift + 1) >> 1;
    // u' on the fly: u behind column o, 0 at column o, in the alignment pad and beyond n
    auto uval = [&](int cabs) -> double { return (cabs > a.o && cabs < a.n) ? a.ubuf[cabs] : 0.0; };
    if constexpr (NCH > 0) {
        double2 av[2][NCH], xv[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int jk = threadIdx.x + 256 * k;
            const int jc = (jk < n2) ? jk : n2 - 1;
            av[0][k] = arow[0][jc];
            av[1][k] = arow[1][jc];
            xv[k] = ldg2(a.ubuf + oc + 2 * jc);
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            SELLA_PIN(av[0][k].x); SELLA_PIN(av[0][k].y); SELLA_PIN(av[1][k].x); SELLA_PIN(av[1][k].y);
            SELLA_PIN(xv[k].x); SELLA_PIN(xv[k].y);
        }
        const double2 zero2 = make_double2(0.0, 0.0);
#pragma unroll
        for (int k = 0; k < NCH; k += 2) {
            const int j0 = threadIdx.x + 256 * k, j1 = j0 + 256;
            const bool has0 = j0 < n2, has1 = j1 < n2;
            const int c0 = oc + 2 * j0, c1 = oc + 2 * j1;
            const double x00 = (has0 && c0 > a.o && c0 < a.n) ? xv[k].x : 0.0;
            const double x01 = (has0 && c0 + 1 > a.o && c0 + 1 < a.n) ? xv[k].y : 0.0;
            const double x10 = (has1 && c1 > a.o && c1 < a.n) ? xv[k + 1].x : 0.0;
            const double x11 = (has1 && c1 + 1 > a.o && c1 + 1 < a.n) ? xv[k + 1].y : 0.0;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                // a chunk beyond the row adds exact zeros (the accumulator starts at +0 and can never be -0)
                const double2 a0 = has0 ? av[r][k] : zero2, a1 = has1 ? av[r][k + 1] : zero2;
                acc[r] += a0.x * x00 + a0.y * x01 + a1.x * x10 + a1.y * x11;
            }
        }
    } else
    for (int j0 = threadIdx.x; j0 < n2; j0 += 512) {
        const int j1 = j0 + 256;
        const bool has1 = j1 < n2;
        double2 a0[2], a1[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            a0[r] = arow[r][j0];
            a1[r] = has1 ? arow[r][j1] : make_double2(0.0, 0.0);
        }
        const double x00 = uval(oc + 2 * j0), x01 = uval(oc + 2 * j0 + 1);
        const double x10 = has1 ? uval(oc + 2 * j1) : 0.0, x11 = has1 ? uval(oc + 2 * j1 + 1) : 0.0;
#pragma unroll
        for (int r = 0; r < 2; ++r) acc[r] += a0[r].x * x00 + a0[r].y * x01 + a1[r].x * x10 + a1[r].y * x11;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const double v = wave_sum_e(acc[r]);
        if (lane == 0) red[wave][r] = v;
    }
    const double ss = wave_sum_e(ssl + ((lane < a.nblkA) ? ssl0 : 0.0));
    __syncthreads();
    if (threadIdx.x == 0) {
        // reflector scalars from the K1 partials (dlarfg)
        double beta, tau, scale;
        if (ss == 0.0) { beta = alpha; tau = 0.0; scale = 0.0; }
        else {
            const double nrm = sqrt(alpha * alpha + ss);
            beta = (alpha >= 0.0) ? -nrm : nrm;
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        double p = 0.0;
        for (int r = 0; r < 2; ++r) {
            const int rr = row0 + r;
            const double res = scale * (red[0][r] + red[1][r] + red[2][r] + red[3][r]) + lead[r];
            if (rr < 0) {
                // dummy row of the alignment pad
            } else if (rr < a.m) {
                const int rabs = a.o + rr;
                const double vr = (rr == 0) ? 1.0 : scale * urow[r];
                a.wraw[rabs] = res;
                a.Vrow[rabs] = vr;
                if (rr > 0) a.Arow[rabs] = vr;
                p += vr * res;
            } else if (rr < a.m + a.i) {
                a.cdots[rr - a.m] = res;
            } else if (rr < mtot) {
                a.cdots[TRD_NBMAX + rr - a.m - a.i] = res;
            }
        }
        a.partB[blockIdx.x] = p;
        if (blockIdx.x == 0) {
            a.taus[a.j] = tau;
            a.evec[a.j] = beta;
            a.colscal[0] = tau;
            a.colscal[1] = scale;
        }
    }
}

// ---- one launch per column, trailing block kept up to date (round 5) ------------------------------------------------
// Below ~1800 trailing rows the block (8 m^2 <= 26 MB) is served by the L2s and the Infinity Cache, and every column of
// the blocked chain costs two launches on their floors (row kernel 3.4-4.4 us + matvec 3.5-6 us) plus its share of the
// rank-2nb update.  There the chain switches to dsytd2's algebra with ONE launch per column: the launch for column j
//   (1) finishes  w = tau A v + alpha2 v  of column j-1 from the raw product, its per-workgroup v.(A v) partials and tau
//       (each wavefront reduces the partials for itself: no barrier in front of the stream),
//   (2) applies   A22 -= v w^T + w v^T   to the rows it owns — so the trailing block is always current and there is no
//       panel, no W^T v / V^T v correction and no separate trailing update —
//   (3) forms row j of the updated block on the fly, u = A[j] - w - w[j] v (three vector loads per column, against
//       2 i + 3 for a lazy panel: a first version that formed the lazily updated row inside the matvec of the BLOCKED
//       chain lost to the two-launch pair at every size, profiles/r05_trd_col_onthefly_sweep.log — the panel reads of all
//       workgroups together exceeded what the L2s deliver),
//   (4) multiplies its updated rows with u (reflector scalars applied afterwards, as in trd_gemv_kernel).
// Cost: the block is written back once per column (16 m^2 instead of 8 m^2 + 16 m^2 / nb bytes) — which is why this is
// the path of the SMALL trailing blocks only.  Everything a launch reads that it also writes is double-buffered by the
// host (v, A v, partials, column scalars); reflector j cannot go to row j of A while other workgroups still read that
// row, so row j-1 is written here (v is streamed anyway) and the last one by trd_upd_finish_kernel, which also applies
// the last rank-2 update before the LDS tail takes over.
struct TrdUpdArgs {
    double* A; int ld, n;
    int j;                      // column
    int o, m, oc, shift;        // o = j + 1, m = n - o, oc = o rounded down to even, shift = o - oc
    int pad;                    // dummy rows in front: row group -> workgroup id mod 8 (XCD) fixed across columns
    const double* v_prev; double* v_cur;            // reflector of column j-1 / j, absolute index (v[o] = 1)
    const double* wraw_prev; double* wraw_cur;      // A22 v, absolute row index
    const double* partB_prev; int nblkB_prev; double* partB_cur;
    const double* colscal_prev; double* colscal_cur;
    double* taus; double* evec; double* dvec;
};

constexpr int TRD_UPD_MAXGRID = 1024;   // workgroups per launch: their v.wraw partials are summed by every wavefront, 16 per lane

template <int R, int NT, bool FIRST>
__global__ __launch_bounds__(NT) void trd_upd_kernel(TrdUpdArgs a) {
    constexpr int NW = NT / 64;
    __shared__ double red[NW][R + 1];
    __shared__ double ush[R];
    __shared__ double bc[3];                                // u[o], w[o], v[o]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bw = a.pad / R;                               // first workgroup that owns a real row
    const int b = blockIdx.x;
    if (b < bw) {
        if (tid == 0) a.partB_cur[b] = 0.0;
        return;
    }
    const int row0 = b * R - a.pad;
    const int j = a.j, o = a.o, n = a.n;
    double* rbase[R];
    int rrow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int rr = row0 + r;
        if (rr < 0) rr = 0;
        if (rr > a.m - 1) rr = a.m - 1;
        rrow[r] = o + rr;
        rbase[r] = a.A + (size_t)(o + rr) * a.ld + a.oc;
    }
    const int n2 = (a.m + a.shift + 1) >> 1;
    const double* Arowj = a.A + (size_t)j * a.ld;
    // Every load of the first chunk goes out before the first use, in the order of use.
    // ---- (1) scalars: partial sums of v.wraw (lane-strided), tau, the entries of v and A v at this workgroup's rows
    double pv[TRD_UPD_MAXGRID / 64], tau = 0.0, wrawj = 0.0, vrow[R], wrow[R];
    const double ajj = Arowj[j];
    if (!FIRST) {
#pragma unroll
        for (int k = 0; k < TRD_UPD_MAXGRID / 64; ++k) {
            const int bb = lane + 64 * k;
            pv[k] = a.partB_prev[bb < a.nblkB_prev ? bb : a.nblkB_prev - 1];
        }
        tau = a.colscal_prev[0];
        wrawj = a.wraw_prev[j];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            vrow[r] = a.v_prev[rrow[r]];
            wrow[r] = a.wraw_prev[rrow[r]];
        }
    }
    // ---- (2) row j, A v and v at this thread's columns, (3) the matrix rows
    int k = tid;
    bool has = k < n2;
    double2 ajv, wrv = make_double2(0.0, 0.0), vpv = wrv, mat[R];
    double lead[R];
    {
        // (lanes without a chunk load chunk 0 again: a branch here would make every wait below cover both paths)
        const int kc = has ? k : 0, ca = a.oc + 2 * kc;
        ajv = ldg2(Arowj + ca);
        if (!FIRST) {
            wrv = ldg2(a.wraw_prev + ca);
            vpv = ldg2(a.v_prev + ca);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) mat[r] = ldg2(rbase[r] + 2 * kc);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) lead[r] = rbase[r][a.shift];
    // ---- uniform scalars: every wavefront reduces for itself (same order everywhere: no barrier, identical results)
    double alpha2 = 0.0, wj = 0.0, dj = ajj;
    if (!FIRST) {
        double vw = 0.0;
#pragma unroll
        for (int q = 0; q < TRD_UPD_MAXGRID / 64; ++q) vw += (lane + 64 * q < a.nblkB_prev) ? pv[q] : 0.0;
        vw = wave_sum_e(vw);
        alpha2 = -0.5 * tau * tau * vw;
        wj = tau * wrawj + alpha2;                          // w[j]   (v[j] = 1)
        dj = ajj - 2.0 * wj;                                // u[j]: the diagonal entry of T
#pragma unroll
        for (int r = 0; r < R; ++r) wrow[r] = tau * wrow[r] + alpha2 * vrow[r];       // w at this workgroup's rows
    }
    // ---- the stream
    double acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0;
    double ss = 0.0;
    double* Aprev = FIRST ? a.A : a.A + (size_t)(j - 1) * a.ld;     // row of the previous reflector (unused in the first launch)
    const int nown = (int)gridDim.x - bw;                   // workgroups that run the stream
    const int nset = (n2 + NT - 1) / NT;
    for (int t = 0; t < nset; ++t) {
        if (has) {
            const int ca = a.oc + 2 * k;
            double ue[2] = {ajv.x, ajv.y};
            double wce[2] = {0.0, 0.0};
            const double vpe[2] = {vpv.x, vpv.y};
            if (!FIRST) {
                wce[0] = tau * wrv.x + alpha2 * vpv.x;
                wce[1] = tau * wrv.y + alpha2 * vpv.y;
                ue[0] = ajv.x - wce[0] - wj * vpv.x;
                ue[1] = ajv.y - wce[1] - wj * vpv.y;
            }
            const bool store = t % nown == b - bw;
            double x[2];
            bool inw[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int c = ca + e;
                inw[e] = c >= o && c < n;
                const bool inu = c > o && c < n;
                if (c == o) { bc[0] = ue[e]; bc[1] = wce[e]; bc[2] = vpe[e]; }
                const int own = c - o - row0;
                if (inw[e] && own >= 0 && own < R) ush[own] = ue[e];
                if (!FIRST && store && inw[e]) Aprev[c] = vpe[e];
                x[e] = inu ? ue[e] : 0.0;
                ss += x[e] * x[e];
                if (!inw[e]) wce[e] = 0.0;                  // (outside the block: no update, and nothing undefined
            }                                               //  from the padding of the vectors reaches the sums)
            const double vm[2] = {inw[0] ? vpe[0] : 0.0, inw[1] ? vpe[1] : 0.0};
#pragma unroll
            for (int r = 0; r < R; ++r) {
                double m0 = mat[r].x, m1 = mat[r].y;
                if (!FIRST) {
                    // the rank-2 update of column j-1 on this row, written back (rows beyond the block are clamped copies
                    // of its last row: they compute, they do not store)
                    m0 -= vrow[r] * wce[0] + wrow[r] * vm[0];
                    m1 -= vrow[r] * wce[1] + wrow[r] * vm[1];
                    const int rr = row0 + r;
                    if (rr >= 0 && rr < a.m) {
                        double* dst = rbase[r] + 2 * k;
                        if (inw[0] && inw[1]) *reinterpret_cast<double2*>(dst) = make_double2(m0, m1);
                        else {
                            if (inw[0]) dst[0] = m0;
                            if (inw[1]) dst[1] = m1;
                        }
                    }
                }
                acc[r] += m0 * x[0] + m1 * x[1];
            }
        }
        k += NT;
        has = k < n2;
        if (t + 1 < nset) {
            const int kc = has ? k : 0, ca = a.oc + 2 * kc;
            ajv = ldg2(Arowj + ca);
            if (!FIRST) {
                wrv = ldg2(a.wraw_prev + ca);
                vpv = ldg2(a.v_prev + ca);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) mat[r] = ldg2(rbase[r] + 2 * kc);
        }
    }
    // ---- one exchange: row sums and |u'|^2
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const double v = wave_sum_e(acc[r]);
        if (lane == 0) red[wave][r] = v;
    }
    {
        const double v0 = wave_sum_e(ss);
        if (lane == 0) red[wave][R] = v0;
    }
    __syncthreads();
    if (wave != 0) return;
    auto total = [&](int q) -> double {
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) sum += red[w][q];
        return sum;
    };
    const double sst = total(R);
    const double alpha = bc[0];
    double beta, taun, scale;
    if (sst == 0.0) { beta = alpha; taun = 0.0; scale = 0.0; }
    else {
        const double nrm = sqrt(alpha * alpha + sst);
        beta = (alpha >= 0.0) ? -nrm : nrm;
        taun = (beta - alpha) / beta;
        scale = 1.0 / (alpha - beta);
    }
    double p = 0.0;
    // per-row registers are indexed by r: unrolled select instead of a dynamic index
    double res = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (lane == r) {
            // entry of the UPDATED row in column o (v[o] = 1 multiplies it)
            const double ld_new = FIRST ? lead[r] : lead[r] - (vrow[r] * bc[1] + wrow[r] * bc[2]);
            res = scale * total(r) + ld_new;
        }
    if (lane < R) {
        const int rr = row0 + lane;
        if (rr >= 0 && rr < a.m) {
            const int rabs = o + rr;
            const double vr = (rr == 0) ? 1.0 : scale * ush[lane];
            a.wraw_cur[rabs] = res;
            a.v_cur[rabs] = vr;
            p = vr * res;
        }
    }
    p = wave_sum_e(p);
    if (lane == 0) {
        a.partB_cur[b] = p;
        if (b == bw) {
            a.taus[j] = taun;
            a.evec[j] = beta;
            a.colscal_cur[0] = taun;
            a.colscal_cur[1] = scale;
            a.dvec[j] = dj;
        }
    }
}

// The last rank-2 update of the one-launch-per-column chain (column jl = o - 1) on the rows >= o it leaves behind, and
// reflector jl into its row.  One workgroup per 4 rows (the block has at most ~130 rows here when the LDS tail follows,
// more when it is switched off); launched with 256 threads.
struct TrdUpdFinishArgs {
    double* A; int ld, n, o;
    const double* v; const double* wraw;
    const double* partB; int nblkB;
    const double* colscal;
};

__global__ __launch_bounds__(256) void trd_upd_finish_kernel(TrdUpdFinishArgs a) {
    const int tid = threadIdx.x, lane = tid & 63;
    double vw = 0.0;
    for (int b0 = lane; b0 < a.nblkB; b0 += 64) vw += a.partB[b0];
    vw = wave_sum_e(vw);
    const double tau = a.colscal[0];
    const double alpha2 = -0.5 * tau * tau * vw;
    const int m = a.n - a.o;
    for (int rr = blockIdx.x * 4; rr < blockIdx.x * 4 + 4 && rr < m; ++rr) {
        const int r = a.o + rr;
        const double vr = a.v[r], wr = tau * a.wraw[r] + alpha2 * vr;
        double* row = a.A + (size_t)r * a.ld;
        for (int c = a.o + tid; c < a.n; c += 256) {
            const double vc = a.v[c], wc = tau * a.wraw[c] + alpha2 * vc;
            row[c] -= vr * wc + wr * vc;
        }
    }
    if (blockIdx.x == 0) {
        double* refl = a.A + (size_t)(a.o - 1) * a.ld;
        for (int c = a.o + 1 + tid; c < a.n; c += 256) refl[c] = a.v[c];
    }
}

// ---- symmetric-aware trailing matvec (large trailing blocks: the stage is bandwidth bound there) ----------------
// A22 u' from the UPPER triangle only: every stored entry A[r][c], c >= r, serves row r (a dot along the row) and —
// transposed — row c.  The matrix is cut into tiles of SYMV_TR rows x SYMV_TC columns on a grid that is fixed in
// ABSOLUTE coordinates (a workgroup keeps its tile, and its XCD, from one column of the factorisation to the next);
// tiles below the diagonal or above the trailing block leave at once.  A workgroup writes, for its tile (I, J),
//     Prow[J][r] = sum_{c in tile, c >= r} A[r][c] u'[c]          Pcol[I][c] = sum_{r in tile, r < c} A[r][c] u'[r]
// and `trd_symv_finish_kernel` adds them up in a fixed order — deterministic, unlike atomics — and does what the
// epilogue of trd_gemv_kernel does (reflector scalars, v, A22 v, the v . A22 v partials, the panel dots).
// Bytes per column: 4 m^2 instead of 8 m^2, plus 16 m (m / SYMV_TR + m / SYMV_TC) for the partial sums.
constexpr int SYMV_TC = 256;            // tile columns; tile rows (64 or 128) are a launch parameter: option eigh_symv_tr

struct TrdSymvArgs {
    const double* A; int ld, n, o;
    const double* ubuf;
    double* Prow; double* Pcol; int ldp;
};

template <int SYMV_TR>
__device__ __forceinline__ void trd_symv_tile(const TrdSymvArgs& a, int I, int J, int NI, int NJ,
                                              double (*colred)[SYMV_TC], double* rowres) {
    const int r0 = I * SYMV_TR, c0 = J * SYMV_TC;
    if (r0 + SYMV_TR - 1 < a.o || c0 + SYMV_TC - 1 < r0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool crossing = c0 < r0 + SYMV_TR;                 // the diagonal runs through this tile
    // the four columns of this lane: c0 + 2 lane (+1), c0 + 128 + 2 lane (+1); loads are clamped into the row
    const int ca = c0 + 2 * lane, cb = ca + 128;
    const int lim = a.ld - 2;
    const int cal = ca <= lim ? ca : lim, cbl = cb <= lim ? cb : lim;
    auto uval = [&](int cabs) -> double { return (cabs > a.o && cabs < a.n) ? a.ubuf[cabs < a.n ? cabs : a.n - 1] : 0.0; };
    const double x0 = uval(ca), x1 = uval(ca + 1), x2 = uval(cb), x3 = uval(cb + 1);
    double col[4] = {0.0, 0.0, 0.0, 0.0};
    constexpr int RPW = SYMV_TR / 4;                         // rows per wavefront, eight at a time
    const int rw = r0 + RPW * wave;
#pragma unroll
    for (int half = 0; half < RPW / 8; ++half) {
        double2 va[8], vb[8];
        double xr[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int r = rw + 8 * half + k;
            const int rl = r < a.n ? r : a.n - 1;
            const double* row = a.A + (size_t)rl * a.ld;
            va[k] = *reinterpret_cast<const double2*>(row + cal);
            vb[k] = *reinterpret_cast<const double2*>(row + cbl);
            xr[k] = uval(r);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int r = rw + 8 * half + k;
            double e0 = va[k].x, e1 = va[k].y, e2 = vb[k].x, e3 = vb[k].y;
            if (crossing) {                                   // entries left of the diagonal belong to other tiles
                if (ca < r) e0 = 0.0;
                if (ca + 1 < r) e1 = 0.0;
                if (cb < r) e2 = 0.0;
                if (cb + 1 < r) e3 = 0.0;
            }
            double dot = e0 * x0 + e1 * x1 + e2 * x2 + e3 * x3;
            if (crossing) {                                   // the diagonal entry itself counts once (in the row dot)
                if (ca == r) e0 = 0.0;
                if (ca + 1 == r) e1 = 0.0;
                if (cb == r) e2 = 0.0;
                if (cb + 1 == r) e3 = 0.0;
            }
            col[0] += e0 * xr[k]; col[1] += e1 * xr[k]; col[2] += e2 * xr[k]; col[3] += e3 * xr[k];
            dot = wave_sum64(dot);
            if (lane == 0) rowres[RPW * wave + 8 * half + k] = dot;
        }
    }
    colred[wave][2 * lane] = col[0];
    colred[wave][2 * lane + 1] = col[1];
    colred[wave][128 + 2 * lane] = col[2];
    colred[wave][128 + 2 * lane + 1] = col[3];
    __syncthreads();
    const int t = threadIdx.x;
    // partial sums are stored by blocks of 64 entries, all partial rows of a block next to each other: the finish kernel
    // streams one contiguous region per block (rows of a plain [tile][entry] array are n * 8 bytes apart — one DRAM page
    // and, for n a multiple of a large power of two, one channel each; measured 0.85 TB/s)
    if (c0 + t < a.n) {
        const int cabs = c0 + t;
        a.Pcol[((size_t)(cabs >> 6) * NI + I) * 64 + (cabs & 63)] = (colred[0][t] + colred[1][t]) + (colred[2][t] + colred[3][t]);
    }
    if (t < SYMV_TR && r0 + t < a.n && r0 + t >= a.o) {
        const int rabs = r0 + t;
        a.Prow[((size_t)(rabs >> 6) * NJ + J) * 64 + (rabs & 63)] = rowres[t];
    }
}

// one workgroup per tile of the absolute grid (NJ, NI).  (A fixed number of workgroups walking the tiles round-robin —
// 256 ... 2048 of them — was measured equal at best: 942 ms per eigh at n = 12288 either way, 1050 with 256.)
template <int SYMV_TR>
__global__ __launch_bounds__(256) void trd_symv_kernel(TrdSymvArgs a) {
    __shared__ double colred[4][SYMV_TC];
    __shared__ double rowres[SYMV_TR];
    trd_symv_tile<SYMV_TR>(a, blockIdx.y, blockIdx.x, gridDim.y, gridDim.x, colred, rowres);
}

struct TrdSymvFinishArgs {
    const double* A; int ld, n, o, j;
    const double* ubuf;
    const double* partA; int nblkA;
    const double* Prow; const double* Pcol; int ldp;
    double* wraw; double* partB;
    double* Vrow; double* Arow;
    double* taus; double* evec; double* colscal;
    const double* Wp; const double* Vp; int ldpan, i;
    double* cdots;
    int tr;                     // rows per tile of the matvec that wrote the partial sums
    int nelem;                  // workgroups that own 64 entries of the result; the 2 i behind them one panel row each
};

__global__ __launch_bounds__(256) void trd_symv_finish_kernel(TrdSymvFinishArgs a) {
    __shared__ double red[4];
    const int tid = threadIdx.x, lane = tid & 63;
    double ssl = 0.0;
    for (int b = lane; b < a.nblkA; b += 64) ssl += a.partA[(size_t)b * TRD_PA];
    const double ss = wave_sum_e(ssl);
    const double alpha = a.ubuf[a.o];
    double beta, tau, scale;
    if (ss == 0.0) { beta = alpha; tau = 0.0; scale = 0.0; }
    else {
        const double nrm = sqrt(alpha * alpha + ss);
        beta = (alpha >= 0.0) ? -nrm : nrm;
        tau = (beta - alpha) / beta;
        scale = 1.0 / (alpha - beta);
    }
    if ((int)blockIdx.x >= a.nelem) {
        // panel row q against v = e_o + scale u'
        const int q = blockIdx.x - a.nelem;
        const double* row = (q < a.i) ? a.Wp + (size_t)q * a.ldpan : a.Vp + (size_t)(q - a.i) * a.ldpan;
        // one workgroup streams a whole row (the dots come out exact, as in trd_gemv_kernel): sixteen entries per thread
        // in flight at a time — a plain loop waits for every load in turn (24 us at m = 12000)
        double acc = 0.0;
        for (int k0 = a.o + tid; k0 < a.n; k0 += 256 * 16) {
            double rv[16], uv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int k = k0 + 256 * u;
                const int kc = k < a.n ? k : a.n - 1;
                rv[u] = row[kc];
                uv[u] = a.ubuf[kc];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int k = k0 + 256 * u;
                if (k < a.n) acc += rv[u] * ((k == a.o) ? 1.0 : scale * uv[u]);
            }
        }
        acc = block_sum_256(acc, red);
        if (tid == 0) a.cdots[(q < a.i) ? q : TRD_NBMAX + q - a.i] = acc;
        return;
    }
    // 64 entries per workgroup; wavefront w adds every fourth partial row (row- and column-side lists taken as one),
    // sixteen loads in flight at a time; the four sums meet in LDS in a fixed order
    __shared__ double quarter[4][64];
    const int wave = tid >> 6;
    // blocks of 64 entries at absolute multiples of 64: a wavefront's entries share their tiles, so the loop bounds and the
    // row pointers below are uniform
    const int k = (a.o & ~63) + blockIdx.x * 64 + lane;
    const int NJ = (a.n + SYMV_TC - 1) / SYMV_TC;
    // every entry of this workgroup needs row-side tiles J >= kmax / TC ... hence per-entry bounds below
    const int kb0 = (a.o & ~63) + blockIdx.x * 64;          // < n: first entry of the block (64 | tile sizes)
    const int J0 = kb0 / SYMV_TC, nrow = NJ - J0;
    const int I0 = a.o / a.tr, ncol = kb0 / a.tr - I0 + 1;
    const int NI = (a.n + a.tr - 1) / a.tr, eb = kb0 >> 6;
    double acc = 0.0;
    for (int q0 = wave; q0 < nrow + ncol; q0 += 64) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int q = q0 + 4 * u;
            const int qq = q < nrow + ncol ? q : nrow + ncol - 1;
            const double* src = (qq < nrow) ? a.Prow + ((size_t)eb * NJ + J0 + qq) * 64
                                            : a.Pcol + ((size_t)eb * NI + I0 + qq - nrow) * 64;
            v[u] = src[lane];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += (q0 + 4 * u < nrow + ncol) ? v[u] : 0.0;
    }
    quarter[wave][lane] = acc;
    __syncthreads();
    double p = 0.0;
    if (wave == 0 && k < a.n && k >= a.o) {
        const double sum = (quarter[0][lane] + quarter[1][lane]) + (quarter[2][lane] + quarter[3][lane]);
        const double res = scale * sum + a.A[(size_t)a.o * a.ld + k];
        const double vk = (k == a.o) ? 1.0 : scale * a.ubuf[k];
        a.wraw[k] = res;
        a.Vrow[k] = vk;
        if (k > a.o) a.Arow[k] = vk;
        p = vk * res;
    }
    p = block_sum_256(p, red);
    if (tid == 0) {
        a.partB[blockIdx.x] = p;
        if (blockIdx.x == 0) {
            a.taus[a.j] = tau;
            a.evec[a.j] = beta;
            a.colscal[0] = tau;
            a.colscal[1] = scale;
        }
    }
}

// ---- the last columns in LDS --------------------------------------------------------------------------------------------
// Below ~1500 trailing rows every column of the factorisation costs two launches on their floors (row kernel 3.3 us,
// matvec 2.2-3.6 us) whatever the block size.  Once the trailing block has m <= TRD_TAIL rows it fits the 160 KB of LDS of
// one CU: ONE workgroup of 1024 threads finishes the factorisation there (unblocked Householder steps, dsytd2's algebra:
// v, p = tau A v, w = p - tau/2 (p.v) v, A -= v w^T + w v^T), five barriers per column instead of two launches.  Row r of
// the block belongs to the 8 threads tid / 8 == r (columns interleaved), so the 8-lane sums of the matvec are three DPP
// steps.  The upper triangle of the block is the authoritative one on entry; reflectors go to the rows of A as in
// trd_gemv_kernel.
constexpr int TRD_TAIL = 128, TRD_TAIL_LD = 136;

__global__ __launch_bounds__(1024) void trd_tail_lds_kernel(double* __restrict__ A, int ld, int n, int j0,
                                                            double* __restrict__ dvec, double* __restrict__ evec,
                                                            double* __restrict__ taus) {
    __shared__ double S[TRD_TAIL][TRD_TAIL_LD];
    __shared__ double v[TRD_TAIL], w[TRD_TAIL], pr[TRD_TAIL];
    __shared__ double red[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = n - j0;
    const int r = tid >> 3, q = tid & 7;
    for (int e = tid; e < m * m; e += 1024) {
        const int rr = e / m, cc = e - rr * m;
        const int lo = rr < cc ? rr : cc, hi = rr < cc ? cc : rr;
        S[rr][cc] = A[(size_t)(j0 + lo) * ld + j0 + hi];
    }
    __syncthreads();
    for (int k = 0; k + 2 < m; ++k) {
        const int o = k + 1;
        // (a) sum of squares behind the leading entry of row k
        if (tid < 128) {
            const double x = (tid > o && tid < m) ? S[k][tid] : 0.0;
            const double ss = wave_sum64(x * x);
            if (lane == 0) red[wave] = ss;
        }
        __syncthreads();
        // (b) reflector (dlarfg), v, the tridiagonal entries
        const double ss = red[0] + red[1];
        const double alpha = S[k][o];
        double beta, tau, scale;
        if (ss == 0.0) { beta = alpha; tau = 0.0; scale = 0.0; }
        else {
            const double nrm = sqrt(alpha * alpha + ss);
            beta = (alpha >= 0.0) ? -nrm : nrm;
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        if (tid < m) {
            const double vc = (tid < o) ? 0.0 : (tid == o ? 1.0 : scale * S[k][tid]);
            v[tid] = vc;
            if (tid > o) A[(size_t)(j0 + k) * ld + j0 + tid] = vc;
        }
        if (tid == 0) { dvec[j0 + k] = S[k][k]; evec[j0 + k] = beta; taus[j0 + k] = tau; }
        __syncthreads();
        // (c) p = tau S22 v
        {
            double acc = 0.0;
            if (r >= o && r < m)
                for (int c = o + q; c < m; c += 8) acc += S[r][c] * v[c];
            acc = wave_dpp_add<0xB1>(acc);
            acc = wave_dpp_add<0x4E>(acc);
            acc = wave_dpp_add<0x141>(acc);
            if (q == 0 && r < m) pr[r] = (r >= o) ? tau * acc : 0.0;
        }
        __syncthreads();
        // (d) w = p - tau/2 (p.v) v
        if (tid < 128) {
            const double x = (tid >= o && tid < m) ? pr[tid] * v[tid] : 0.0;
            const double pv = wave_sum64(x);
            if (lane == 0) red[wave] = pv;
        }
        __syncthreads();
        const double alpha2 = -0.5 * tau * (red[0] + red[1]);
        if (tid >= o && tid < m) w[tid] = pr[tid] + alpha2 * v[tid];
        __syncthreads();
        // (e) S22 -= v w^T + w v^T
        if (r >= o && r < m) {
            const double vr = v[r], wr = w[r];
            for (int c = o + q; c < m; c += 8) S[r][c] -= vr * w[c] + wr * v[c];
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (m >= 2) {
            dvec[j0 + m - 2] = S[m - 2][m - 2];
            evec[j0 + m - 2] = S[m - 2][m - 1];
            taus[j0 + m - 2] = 0.0;
        }
        dvec[j0 + m - 1] = S[m - 1][m - 1];
        evec[j0 + m - 1] = 0.0;
        taus[j0 + m - 1] = 0.0;
    }
}

__global__ void tridiag_tail_kernel(const double* __restrict__ A, int ld, int n, double* dvec,
                                    double* evec, double* taus) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (n >= 2) {
        dvec[n - 2] = A[(size_t)(n - 2) * ld + n - 2];
        evec[n - 2] = A[(size_t)(n - 2) * ld + n - 1];
        taus[n - 2] = 0.0;
    }
    dvec[n - 1] = A[(size_t)(n - 1) * ld + n - 1];
    evec[n - 1] = 0.0;
    taus[n - 1] = 0.0;
}

// ---------------------------------------------------------------------------------------
// stage 2 kernels
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void set_identity_kernel(double* __restrict__ Z, int ld, int n) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j < n) Z[(size_t)i * ld + j] = (i == j) ? 1.0 : 0.0;
}

// one wavefront per leaf; lane r owns "row r" of the leaf's eigenvector matrix, which is stored
// transposed (eigenvectors as rows of Zt): element (r, i) lives at Zt[(lo+i)*ld + lo + r]
__global__ __launch_bounds__(64) void leaf_ql_kernel(const double* __restrict__ dvec,
                                                     const double* __restrict__ evec,
                                                     double* __restrict__ wout,
                                                     const int* __restrict__ ranges,
                                                     double* __restrict__ Zt, int ld,
                                                     int* __restrict__ info) {
    const int lo = ranges[2 * blockIdx.x], hi = ranges[2 * blockIdx.x + 1];
    const int m = hi - lo;
    double dl[64], el[64];
    for (int i = 0; i < m; ++i) {
        dl[i] = dvec[lo + i];
        el[i] = (i < m - 1) ? evec[lo + i] : 0.0;
    }
    const int lane = threadIdx.x;
    int st = small::tridiag_ql(m, dl, el, Zt + (size_t)lo * ld + lo, 1, m, lane, 64, ld);
    if (lane == 0) {
        for (int i = 0; i < m; ++i) wout[lo + i] = dl[i];
        if (st != 0) info[0] = st;
    }
}


// Givens rotations on pairs of rows, in order:  x' = c x + s y ; y' = c y - s x   (BLAS drot), one
// column per lane.  The deflation of a merge produces CHAINS: the row left as y by one rotation is
// the x of the next, and no row is touched again once it has been an x (plan_deflation visits the poles
// in ascending order).  The chained value therefore stays in a register, every row is read once and
// written once, and — since the indices do not depend on the data — the loads of a whole batch of
// rotations are issued before the (sequential) arithmetic on it, which turns a chain of nrot memory
// round trips into nrot/16 of them.
constexpr int ROT_BATCH = 16;
__device__ __forceinline__ void rot_rows_body(double* __restrict__ Zb, int ld, int N, int nrot,
                                              const int* __restrict__ i1, const int* __restrict__ i2,
                                              const double* __restrict__ cs) {
    const int col = blockIdx.x * 64 + threadIdx.x;
    if (col >= N) return;
    double carry = 0.0;
    int crow = -1;                                              // row whose new value sits in `carry`
    for (int r0 = 0; r0 < nrot; r0 += ROT_BATCH) {
        double y[ROT_BATCH], xf[ROT_BATCH], c[ROT_BATCH], sn[ROT_BATCH];
        int ra[ROT_BATCH], rb[ROT_BATCH];
        bool fresh[ROT_BATCH];
#pragma unroll
        for (int k = 0; k < ROT_BATCH; ++k) {
            const int r = (r0 + k < nrot) ? r0 + k : nrot - 1;
            ra[k] = i1[r];
            rb[k] = i2[r];
            c[k] = cs[2 * r];
            sn[k] = cs[2 * r + 1];
            fresh[k] = (k == 0) ? (ra[0] != crow) : (ra[k] != rb[k - 1]);
            y[k] = Zb[(size_t)rb[k] * ld + col];
            xf[k] = fresh[k] ? Zb[(size_t)ra[k] * ld + col] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < ROT_BATCH; ++k) {
            if (r0 + k < nrot) {
                double x = carry;
                if (fresh[k]) {
                    if (crow >= 0) Zb[(size_t)crow * ld + col] = carry;
                    x = xf[k];
                }
                Zb[(size_t)ra[k] * ld + col] = c[k] * x + sn[k] * y[k];
                carry = c[k] * y[k] - sn[k] * x;
                crow = rb[k];
            }
        }
    }
    if (crow >= 0) Zb[(size_t)crow * ld + col] = carry;
}

__global__ __launch_bounds__(64) void rot_rows_kernel(double* __restrict__ Zb, int ld, int N, int nrot,
                                                      const int* __restrict__ i1,
                                                      const int* __restrict__ i2,
                                                      const double* __restrict__ cs) {
    rot_rows_body(Zb, ld, N, nrot, i1, i2, cs);
}

// R[i][c] -= tau v[i] w[c] on a block of rows (Householder reflection applied to eigenvector rows)
__global__ __launch_bounds__(256) void rows_ger_kernel(double* __restrict__ R, int ld, int nrows, int ncols,
                                                       const double* __restrict__ v,
                                                       const double* __restrict__ wv, double tau) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (col < ncols && i < nrows) R[(size_t)i * ld + col] -= tau * v[i] * wv[col];
}

struct WaveSum {
    __device__ double operator()(double v) const { return wave_sum_e(v); }
};
struct WaveProd {
    __device__ double operator()(double v) const {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) v *= __shfl_xor(v, m, 64);
        return v;
    }
};

// one WAVEFRONT per root: the 64 lanes split every O(K) sum of the iteration
__device__ __forceinline__ void secular_body(int K, const double* __restrict__ D, const double* __restrict__ w,
                                             double rho, double* __restrict__ tau, int* __restrict__ org,
                                             double* __restrict__ lam, int* __restrict__ info) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (j >= K) return;            // whole wavefront leaves together
    int o;
    double t;
    const int it = secular::solve_root(K, D, w, rho, j, &o, &t, lane, 64, WaveSum());
    if (lane == 0) {
        tau[j] = t;
        org[j] = o;
        lam[j] = D[o] + t;
        if (it < 0) info[1] = j + 1;
    }
}

__global__ __launch_bounds__(256) void secular_kernel(int K, const double* __restrict__ D,
                                                      const double* __restrict__ w, double rho,
                                                      double* __restrict__ tau, int* __restrict__ org,
                                                      double* __restrict__ lam, int* __restrict__ info) {
    secular_body(K, D, w, rho, tau, org, lam, info);
}

__device__ __forceinline__ void zhat_body(int K, const double* __restrict__ D, const double* __restrict__ w,
                                          const double* __restrict__ tau, const int* __restrict__ org,
                                          double* __restrict__ zh) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= K) return;
    const double z = secular::zhat(K, D, w, tau, org, i, lane, 64, WaveProd());
    if (lane == 0) zh[i] = z;
}

__global__ __launch_bounds__(256) void zhat_kernel(int K, const double* __restrict__ D,
                                                   const double* __restrict__ w,
                                                   const double* __restrict__ tau,
                                                   const int* __restrict__ org, double* __restrict__ zh) {
    zhat_body(K, D, w, tau, org, zh);
}

// Ut[j][i] = zhat_i / ((D_i - D_org_j) - tau_j), row j normalised
__device__ __forceinline__ void build_u_body(int K, const double* __restrict__ D, const double* __restrict__ zh,
                                             const double* __restrict__ tau, const int* __restrict__ org,
                                             double* __restrict__ Ut, int ldu) {
    __shared__ double red[4];
    const int j = blockIdx.x;
    const double Do = D[org[j]], tj = tau[j];
    double ss = 0.0;
    for (int i = threadIdx.x; i < K; i += 256) {
        const double u = zh[i] / ((D[i] - Do) - tj);
        Ut[(size_t)j * ldu + i] = u;
        ss += u * u;
    }
    ss = block_sum_256(ss, red);
    const double inv = 1.0 / sqrt(ss);
    for (int i = threadIdx.x; i < K; i += 256) Ut[(size_t)j * ldu + i] *= inv;
}

__global__ __launch_bounds__(256) void build_u_kernel(int K, const double* __restrict__ D,
                                                      const double* __restrict__ zh,
                                                      const double* __restrict__ tau,
                                                      const int* __restrict__ org,
                                                      double* __restrict__ Ut, int ldu) {
    build_u_body(K, D, zh, tau, org, Ut, ldu);
}

// ---- one launch per tree level: blockIdx.y (or .z) selects the merge ----------------------------------
// Per-merge parameters on the device.  The per-row arrays (D, w, tau, origin, lambda, zhat, rotation
// lists, gather indices) of all merges of a level share n-length buffers at offset lo.
struct MergeDev {
    int lo, N, K, nrot;     // first row/column, size, non-deflated count, rotations
    int n1, mid;            // rows of the left child, first column of the right child
    double rho, sgn;
};

__global__ __launch_bounds__(256) void gather_z_batched_kernel(const MergeDev* __restrict__ md,
                                                               const double* __restrict__ Zt, int ld,
                                                               double* __restrict__ z) {
    const MergeDev m = md[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m.N) return;
    z[m.lo + i] = (i < m.n1) ? Zt[(size_t)(m.lo + i) * ld + m.mid - 1] : m.sgn * Zt[(size_t)(m.lo + i) * ld + m.mid];
}

__global__ __launch_bounds__(64) void rot_rows_batched_kernel(const MergeDev* __restrict__ md, double* __restrict__ Zb,
                                                              int ld, const int* __restrict__ i1,
                                                              const int* __restrict__ i2,
                                                              const double* __restrict__ cs) {
    const MergeDev m = md[blockIdx.y];
    if (m.nrot == 0 || (int)(blockIdx.x * 64) >= m.N) return;
    rot_rows_body(Zb + (size_t)m.lo * ld + m.lo, ld, m.N, m.nrot, i1 + m.lo, i2 + m.lo, cs + 2 * (size_t)m.lo);
}

__global__ __launch_bounds__(256) void secular_batched_kernel(const MergeDev* __restrict__ md,
                                                              const double* __restrict__ D,
                                                              const double* __restrict__ w,
                                                              double* __restrict__ tau, int* __restrict__ org,
                                                              double* __restrict__ lam, int* __restrict__ info) {
    const MergeDev m = md[blockIdx.y];
    secular_body(m.K, D + m.lo, w + m.lo, m.rho, tau + m.lo, org + m.lo, lam + m.lo, info);
}

__global__ __launch_bounds__(256) void zhat_batched_kernel(const MergeDev* __restrict__ md,
                                                           const double* __restrict__ D,
                                                           const double* __restrict__ w,
                                                           const double* __restrict__ tau,
                                                           const int* __restrict__ org, double* __restrict__ zh) {
    const MergeDev m = md[blockIdx.y];
    zhat_body(m.K, D + m.lo, w + m.lo, tau + m.lo, org + m.lo, zh + m.lo);
}

__global__ __launch_bounds__(256) void build_u_batched_kernel(const MergeDev* __restrict__ md,
                                                              const double* __restrict__ D,
                                                              const double* __restrict__ zh,
                                                              const double* __restrict__ tau,
                                                              const int* __restrict__ org,
                                                              double* __restrict__ Ut, int ldu) {
    const MergeDev m = md[blockIdx.y];
    if ((int)blockIdx.x >= m.K) return;
    build_u_body(m.K, D + m.lo, zh + m.lo, tau + m.lo, org + m.lo, Ut + (size_t)m.lo * ldu + m.lo, ldu);
}

// Row p of the level's output layout takes row idx[p] of `cur` restricted to its merge's columns: the
// first K rows of a merge (non-deflated, secular order) go to Zc for the GEMM, the others straight to nxt.
__global__ __launch_bounds__(256) void gather_level_kernel(const MergeDev* __restrict__ md,
                                                           const int* __restrict__ merge_of_row,
                                                           const int* __restrict__ idx,
                                                           const double* __restrict__ cur,
                                                           double* __restrict__ Zc, double* __restrict__ nxt,
                                                           int ld) {
    const int p = blockIdx.y;
    const MergeDev m = md[merge_of_row[p]];
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= m.N) return;
    const double v = cur[(size_t)idx[p] * ld + m.lo + col];
    double* dst = (p - m.lo < m.K) ? Zc : nxt;
    dst[(size_t)p * ld + m.lo + col] = v;
}

// ---------------------------------------------------------------------------------------
// stage 3 kernels
// ---------------------------------------------------------------------------------------
// component c of reflector j as stored by stage 1 (unit head at j+1, tail in A[j][j+2..])
__device__ __forceinline__ double yval(const double* __restrict__ A, int ld, const double* __restrict__ taus,
                                       int j, int c) {
    if (c <= j || taus[j] == 0.0) return 0.0;
    return (c == j + 1) ? 1.0 : A[(size_t)j * ld + c];
}

// G[b][p][q] = y_p . y_q for the reflectors of block b  (one workgroup per block)
__global__ __launch_bounds__(256) void wy_gram_kernel(const double* __restrict__ A, int ld, int n, int nrefl,
                                                      const double* __restrict__ taus,
                                                      double* __restrict__ G) {
    __shared__ double Ys[WY_NB][65];
    const int b = blockIdx.x;
    const int j0 = b * WY_NB;
    const int kb = (nrefl - j0 < WY_NB) ? (nrefl - j0) : WY_NB;
    const int p = threadIdx.x >> 3, q0 = (threadIdx.x & 7) * 4;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int ct = j0 + 1; ct < n; ct += 64) {
        __syncthreads();
        for (int e = threadIdx.x; e < WY_NB * 64; e += 256) {
            const int r = e >> 6, cc = e & 63;
            Ys[r][cc] = (r < kb && ct + cc < n) ? yval(A, ld, taus, j0 + r, ct + cc) : 0.0;
        }
        __syncthreads();
#pragma unroll 8
        for (int cc = 0; cc < 64; ++cc) {
            const double yp = Ys[p][cc];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += yp * Ys[q0 + k][cc];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) G[((size_t)b * WY_NB + p) * WY_NB + q0 + k] = acc[k];
}

// X <- X (H_{nrefl-1} ... H_0): every workgroup owns 16 rows of X and sweeps the compact-WY blocks
// from the last to the first:  M = X Y_b^T ; M2 = M C_b ; X -= M2 Y_b   with C_b = T_b^T.
// 64-column tiles go through LDS; the global loads of tile t+1 are issued into registers before
// the arithmetic on tile t starts (software pipelining), so HBM/L2 latency overlaps the FMAs.
__global__ __launch_bounds__(256) void wy_apply_kernel(double* __restrict__ X, int ldx, int n,
                                                       const double* __restrict__ A, int ld, int nrefl,
                                                       const double* __restrict__ taus,
                                                       const double* __restrict__ Call, int nblk) {
    __shared__ double Xs[16][65];
    __shared__ double Ys[WY_NB][65];
    __shared__ double Ms[16][WY_NB + 1];
    __shared__ double M2s[16][WY_NB + 1];
    __shared__ double Cs[WY_NB][WY_NB + 1];
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * 16;
    const int r = tid >> 4, h = tid & 15;
    const bool rowok = r0 + r < n;
    double* xrow = X + (size_t)(rowok ? r0 + r : 0) * ldx;
    for (int b = nblk - 1; b >= 0; --b) {
        const int j0 = b * WY_NB;
        const int kb = (nrefl - j0 < WY_NB) ? (nrefl - j0) : WY_NB;
        const int c0 = j0 + 1;
        double xr[4], yr[8];
        // the prefetch below reads X entries other threads updated in the previous block's phase 3
        __syncthreads();
        // ---- phase 1: M = X Y^T --------------------------------------------------------------
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = tid + k * 256, rr = e >> 6, cc = e & 63;
            xr[k] = (r0 + rr < n && c0 + cc < n) ? X[(size_t)(r0 + rr) * ldx + c0 + cc] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = tid + k * 256, pr = e >> 6, cc = e & 63;
            yr[k] = (pr < kb && c0 + cc < n) ? yval(A, ld, taus, j0 + pr, c0 + cc) : 0.0;
        }
        double a0 = 0.0, a1 = 0.0;
        for (int ct = c0; ct < n; ct += 64) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int e = tid + k * 256; Xs[e >> 6][e & 63] = xr[k]; }
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int e = tid + k * 256; Ys[e >> 6][e & 63] = yr[k]; }
            __syncthreads();
            const int cn = ct + 64;
            if (cn < n) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e = tid + k * 256, rr = e >> 6, cc = e & 63;
                    xr[k] = (r0 + rr < n && cn + cc < n) ? X[(size_t)(r0 + rr) * ldx + cn + cc] : 0.0;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int e = tid + k * 256, pr = e >> 6, cc = e & 63;
                    yr[k] = (pr < kb && cn + cc < n) ? yval(A, ld, taus, j0 + pr, cn + cc) : 0.0;
                }
            }
#pragma unroll 8
            for (int cc = 0; cc < 64; ++cc) {
                const double x = Xs[r][cc];
                a0 += x * Ys[2 * h][cc];
                a1 += x * Ys[2 * h + 1][cc];
            }
        }
        __syncthreads();
        Ms[r][2 * h] = a0;
        Ms[r][2 * h + 1] = a1;
        for (int e = tid; e < WY_NB * WY_NB; e += 256) Cs[e >> 5][e & 31] = Call[(size_t)b * WY_NB * WY_NB + e];
        __syncthreads();
        // ---- phase 2: M2 = M C ----------------------------------------------------------------
        double m0 = 0.0, m1 = 0.0;
#pragma unroll 8
        for (int p = 0; p < WY_NB; ++p) {
            const double mv = Ms[r][p];
            m0 += mv * Cs[p][2 * h];
            m1 += mv * Cs[p][2 * h + 1];
        }
        M2s[r][2 * h] = m0;
        M2s[r][2 * h + 1] = m1;
        // ---- phase 3: X -= M2 Y  (thread: row r, columns 4h..4h+3 of every tile) -----------------
        double xv[4], xn[4];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = tid + k * 256, pr = e >> 6, cc = e & 63;
            yr[k] = (pr < kb && c0 + cc < n) ? yval(A, ld, taus, j0 + pr, c0 + cc) : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cabs = c0 + 4 * h + k;
            xv[k] = (rowok && cabs < n) ? xrow[cabs] : 0.0;
        }
        for (int ct = c0; ct < n; ct += 64) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int e = tid + k * 256; Ys[e >> 6][e & 63] = yr[k]; }
            __syncthreads();
            const int cn = ct + 64;
            if (cn < n) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int e = tid + k * 256, pr = e >> 6, cc = e & 63;
                    yr[k] = (pr < kb && cn + cc < n) ? yval(A, ld, taus, j0 + pr, cn + cc) : 0.0;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int cabs = cn + 4 * h + k;
                    xn[k] = (rowok && cabs < n) ? xrow[cabs] : 0.0;
                }
            }
            double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
            for (int q = 0; q < WY_NB; ++q) {
                const double mq = M2s[r][q];
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k] += mq * Ys[q][4 * h + k];
            }
            if (rowok) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int cabs = ct + 4 * h + k;
                    if (cabs < n) xrow[cabs] = xv[k] - s[k];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) xv[k] = xn[k];
        }
    }
}

// C_b = T_b^T from the block's Gram matrix, in place: T^-1 = striu(Y Y^T) + diag(1/tau) (forward, columnwise
// compact WY), inverted by back-substitution — one workgroup per block, thread `col` owns column col of T.
// (This was host code between two synchronisations, ~1 ms with the queue empty at n = 3072.)
__global__ __launch_bounds__(64) void wy_tinv_kernel(double* __restrict__ G, int nrefl,
                                                     const double* __restrict__ taus) {
    __shared__ double S[WY_NB][WY_NB + 1], Tm[WY_NB][WY_NB + 1];
    const int b = blockIdx.x, j0 = b * WY_NB;
    const int kb = (nrefl - j0 < WY_NB) ? (nrefl - j0) : WY_NB;
    double* Gb = G + (size_t)b * WY_NB * WY_NB;
    for (int e = threadIdx.x; e < WY_NB * WY_NB; e += 64) {
        const int i = e / WY_NB, k2 = e % WY_NB;
        double v = 0.0;
        if (i < kb && k2 < kb) {
            if (k2 > i) v = Gb[e];
            else if (k2 == i) { const double t = taus[j0 + i]; v = (t != 0.0) ? 1.0 / t : 1.0; }
        }
        S[i][k2] = v;
        Tm[i][k2] = 0.0;
    }
    __syncthreads();
    const int col = threadIdx.x;
    if (col < kb) {
        for (int i = col; i >= 0; --i) {                       // entries below the diagonal of T stay zero
            double s = (i == col) ? 1.0 : 0.0;
            for (int k2 = i + 1; k2 <= col; ++k2) s -= S[i][k2] * Tm[k2][col];
            Tm[i][col] = s / S[i][i];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < WY_NB * WY_NB; e += 64) {
        const int p = e / WY_NB, q = e % WY_NB;                  // C[p][q] = T[q][p]
        Gb[e] = (p < kb && q < kb) ? Tm[q][p] : 0.0;
    }
}

// ---- MFMA back-transformation -------------------------------------------------------------------
typedef double wy_f64x4 __attribute__((ext_vector_type(4)));

// Yf[j][c] = explicit reflector j (0 up to column j, 1 at j+1, stored tail beyond), zero in the row
// padding and in the rows that pad the last block to WY_NB.  grid (ceil(ld/256), nrows).
__global__ __launch_bounds__(256) void wy_expand_kernel(const double* __restrict__ A, int ld, int n, int nrefl,
                                                        const double* __restrict__ taus,
                                                        double* __restrict__ Yf) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int j = blockIdx.y;
    if (c >= ld) return;
    Yf[(size_t)j * ld + c] = (j < nrefl && c < n) ? yval(A, ld, taus, j, c) : 0.0;
}

// X <- X (H_{nrefl-1} ... H_0) on the matrix cores.  A workgroup (4 wavefronts) owns 16 rows of X and
// sweeps the compact-WY blocks from the last to the first; for each block
//     M = X Y_b^T (16 x 32, the wavefronts split the columns, partials meet in LDS),
//     M2 = M C_b,   X -= M2 Y_b
// with v_mfma_f64_16x16x4_f64.  Operands are fetched straight from global memory as 32-byte vectors:
// the summation index of a tile product may be permuted freely, so lane (i, g) takes the four
// consecutive columns 4g..4g+3 of a 16-column group for four successive MFMAs; in the update the
// 16 tile columns are the strided set {4 nn + q}, which again makes every access a 32-byte vector.
// Columns are dealt to wavefronts by absolute 64-column chunk index, so a wavefront only ever
// re-reads X entries it wrote itself.
template <int NW>
__global__ __launch_bounds__(64 * NW) void wy_apply_mfma_kernel(double* __restrict__ X, int ldx, int n,
                                                            const double* __restrict__ Yf,
                                                            const double* __restrict__ Call, int nblk) {
    __shared__ double Mp[NW][16][33];
    __shared__ double Ms[16][33];
    __shared__ double M2s[16][33];
    __shared__ double Cs[WY_NB][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int r0 = blockIdx.x * 16;
    const int rowA = (r0 + li < n) ? r0 + li : n - 1;            // A-operand row of this lane (phase 1)
    int rowD[4];                                                // C/D rows of this lane (phase 3)
#pragma unroll
    for (int r = 0; r < 4; ++r) rowD[r] = (r0 + lg + 4 * r < n) ? r0 + lg + 4 * r : n - 1;
    const double4 zero4 = make_double4(0.0, 0.0, 0.0, 0.0);
    for (int b = nblk - 1; b >= 0; --b) {
        const int j0 = b * WY_NB;
        const int cs = ((j0 + 1) >> 6) << 6;                    // Y_b vanishes left of column j0 + 1
        const int cw = cs + 64 * ((wave - (cs >> 6)) & (NW - 1));   // first chunk of this wavefront
        for (int e = tid; e < WY_NB * WY_NB; e += 64 * NW) Cs[e >> 5][e & 31] = Call[(size_t)b * WY_NB * WY_NB + e];
        // ---- phase 1: M = X Y^T -------------------------------------------------------------------
        wy_f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        const double* xrow = X + (size_t)rowA * ldx;
        const double* y0row = Yf + (size_t)(j0 + li) * ldx;
        const double* y1row = y0row + (size_t)16 * ldx;
        for (int cc = cw; cc < ldx; cc += 64 * NW) {
#pragma unroll
            for (int sg = 0; sg < 4; ++sg) {
                const int col = cc + 16 * sg + 4 * lg;
                const bool ok = col < ldx;
                const int colc = ok ? col : 0;
                double4 xa = *reinterpret_cast<const double4*>(xrow + colc);
                const double4 ya = *reinterpret_cast<const double4*>(y0row + colc);
                const double4 yb = *reinterpret_cast<const double4*>(y1row + colc);
                if (!ok) xa = zero4;
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.x, ya.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.x, yb.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.y, ya.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.y, yb.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.z, ya.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.z, yb.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.w, ya.w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.w, yb.w, acc1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            Mp[wave][lg + 4 * r][li] = acc0[r];
            Mp[wave][lg + 4 * r][16 + li] = acc1[r];
        }
        __syncthreads();
        if (tid < 256) {
            const int row = tid >> 4, pc = tid & 15;
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int wv = 0; wv < NW; ++wv) { s0 += Mp[wv][row][pc]; s1 += Mp[wv][row][pc + 16]; }
            Ms[row][pc] = s0;
            Ms[row][pc + 16] = s1;
        }
        __syncthreads();
        // ---- phase 2: M2 = M C  (negated: the update is an accumulate) --------------------------------
        if (tid < 256) {
            const int row = tid >> 4, qc = tid & 15;
            double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
            for (int pp = 0; pp < WY_NB; ++pp) {
                const double m = Ms[row][pp];
                s0 += m * Cs[pp][qc];
                s1 += m * Cs[pp][qc + 16];
            }
            M2s[row][qc] = -s0;
            M2s[row][qc + 16] = -s1;
        }
        __syncthreads();
        // ---- phase 3: X += (-M2) Y ----------------------------------------------------------------
        double m2a[8];
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) m2a[kt] = M2s[li][4 * kt + lg];
        const double* ybase = Yf + (size_t)(j0 + lg) * ldx;
        for (int cc = cw; cc < ldx; cc += 64 * NW) {
            const int col = cc + 4 * li;
            const bool ok = col < ldx;
            const int colc = ok ? col : 0;
            double4 yv[8], xv[4];
#pragma unroll
            for (int kt = 0; kt < 8; ++kt)
                yv[kt] = *reinterpret_cast<const double4*>(ybase + (size_t)(4 * kt) * ldx + colc);
#pragma unroll
            for (int r = 0; r < 4; ++r) xv[r] = *reinterpret_cast<const double4*>(X + (size_t)rowD[r] * ldx + colc);
            wy_f64x4 d0, d1, d2, d3;
#pragma unroll
            for (int r = 0; r < 4; ++r) { d0[r] = xv[r].x; d1[r] = xv[r].y; d2[r] = xv[r].z; d3[r] = xv[r].w; }
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) {
                d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(m2a[kt], yv[kt].x, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(m2a[kt], yv[kt].y, d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(m2a[kt], yv[kt].z, d2, 0, 0, 0);
                d3 = __builtin_amdgcn_mfma_f64_16x16x4f64(m2a[kt], yv[kt].w, d3, 0, 0, 0);
            }
            if (ok) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (r0 + lg + 4 * r < n)
                        *reinterpret_cast<double4*>(X + (size_t)rowD[r] * ldx + col) = make_double4(d0[r], d1[r], d2[r], d3[r]);
            }
        }
        __syncthreads();                                        // LDS tiles are reused by the next block
    }
}

// ---- 64 reflectors per compact-WY block ---------------------------------------------------------------------------------
// Every block of the sweep above is one pass over the workgroup's rows of X (read twice, written once).  For n beyond a
// few thousand those rows no longer stay in L2 between blocks (16 x 12288 doubles = 1.5 MB per workgroup, 768 workgroups)
// and the sweep is bound by that traffic: n / 32 passes over X.  Twice the reflectors per block = half the passes; the
// reflector stream and the MFMA work are unchanged.
constexpr int WY_NB2 = 64;

// grid (blocks, S): slice s of a block takes every S-th 64-column chunk and writes its own partial Gram matrix
// (G[(b S + s)]); wy_tinv64_kernel adds the S partials in a fixed order.
__global__ __launch_bounds__(256) void wy_gram64_kernel(const double* __restrict__ A, int ld, int n, int nrefl,
                                                        const double* __restrict__ taus, double* __restrict__ G) {
    __shared__ double Ys[WY_NB2][65];
    const int b = blockIdx.x, j0 = b * WY_NB2;
    const int S = gridDim.y, sl = blockIdx.y;
    const int kb = (nrefl - j0 < WY_NB2) ? (nrefl - j0) : WY_NB2;
    const int p = threadIdx.x >> 2, q0 = (threadIdx.x & 3) * 16;
    double acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.0;
    for (int ct = j0 + 1 + 64 * sl; ct < n; ct += 64 * S) {
        __syncthreads();
        for (int e = threadIdx.x; e < WY_NB2 * 64; e += 256) {
            const int r = e >> 6, cc = e & 63;
            Ys[r][cc] = (r < kb && ct + cc < n) ? yval(A, ld, taus, j0 + r, ct + cc) : 0.0;
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
    for (int k = 0; k < 16; ++k) G[(((size_t)b * S + sl) * WY_NB2 + p) * WY_NB2 + q0 + k] = acc[k];
}

// C_b = T_b^T for a 64-block, in place (see wy_tinv_kernel).  One LDS array serves both triangular matrices: T^-1 (from
// the Gram matrix) in the upper triangle with its diagonal, the transpose of T — which is the output — growing in the
// strictly lower one, its diagonal beside it.
__global__ __launch_bounds__(64) void wy_tinv64_kernel(double* __restrict__ G, const double* __restrict__ Gpart, int nsl,
                                                       int nrefl, const double* __restrict__ taus) {
    __shared__ double S[WY_NB2][WY_NB2 + 1];
    __shared__ double dg[WY_NB2];
    const int b = blockIdx.x, j0 = b * WY_NB2;
    const int kb = (nrefl - j0 < WY_NB2) ? (nrefl - j0) : WY_NB2;
    double* Gb = G + (size_t)b * WY_NB2 * WY_NB2;
    for (int e = threadIdx.x; e < WY_NB2 * WY_NB2; e += 64) {
        const int i = e >> 6, k2 = e & 63;
        double v = 0.0;
        if (i < kb && k2 < kb) {
            if (k2 > i) {
                for (int sl = 0; sl < nsl; ++sl) v += Gpart[((size_t)b * nsl + sl) * WY_NB2 * WY_NB2 + e];
            } else if (k2 == i) { const double t = taus[j0 + i]; v = (t != 0.0) ? 1.0 / t : 1.0; }
        }
        S[i][k2] = v;
    }
    dg[threadIdx.x] = 0.0;
    __syncthreads();
    const int col = threadIdx.x;
    if (col < kb) {
        for (int i = col; i >= 0; --i) {                       // column `col` of T, bottom up; T[k2][col] lives at S[col][k2]
            double s = (i == col) ? 1.0 : 0.0;
            for (int k2 = i + 1; k2 <= col; ++k2) s -= S[i][k2] * ((k2 == col) ? dg[col] : S[col][k2]);
            const double val = s / S[i][i];
            if (i == col) dg[col] = val;
            else S[col][i] = val;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < WY_NB2 * WY_NB2; e += 64) {
        const int p = e >> 6, q = e & 63;                        // C[p][q] = T[q][p]
        Gb[e] = (p < kb && q < kb) ? ((q < p) ? S[p][q] : (q == p ? dg[p] : 0.0)) : 0.0;
    }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void wy_apply_mfma64_kernel(double* __restrict__ X, int ldx, int n,
                                                              const double* __restrict__ Yf,
                                                              const double* __restrict__ Call, int nblk) {
    constexpr int MPC = (NW * 16 > WY_NB2 ? NW * 16 : WY_NB2) * 65;
    __shared__ double MpC[MPC];                                  // phase-1 partials of the wavefronts, then C_b
    __shared__ double Ms[16][65];
    __shared__ double M2s[16][65];
    double (*Mp)[16][65] = reinterpret_cast<double (*)[16][65]>(MpC);
    double (*Cs)[65] = reinterpret_cast<double (*)[65]>(MpC);
    constexpr int CR = WY_NB2 * WY_NB2 / (64 * NW);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int r0 = blockIdx.x * 16;
    const int rowA = (r0 + li < n) ? r0 + li : n - 1;
    int rowD[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rowD[r] = (r0 + lg + 4 * r < n) ? r0 + lg + 4 * r : n - 1;
    const double4 zero4 = make_double4(0.0, 0.0, 0.0, 0.0);
    for (int b = nblk - 1; b >= 0; --b) {
        const int j0 = b * WY_NB2;
        const int cs = ((j0 + 1) >> 6) << 6;
        const int cw = cs + 64 * ((wave - (cs >> 6)) & (NW - 1));
        double creg[CR];                                         // C_b: in flight now, into LDS once the partials are consumed
#pragma unroll
        for (int u = 0; u < CR; ++u) creg[u] = Call[(size_t)b * WY_NB2 * WY_NB2 + tid + u * 64 * NW];
        // ---- phase 1: M = X Y^T (16 x 64) ---------------------------------------------------------
        wy_f64x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = wy_f64x4{0.0, 0.0, 0.0, 0.0};
        const double* xrow = X + (size_t)rowA * ldx;
        const double* yrow0 = Yf + (size_t)(j0 + li) * ldx;
        for (int cc = cw; cc < ldx; cc += 64 * NW) {
#pragma unroll
            for (int sg = 0; sg < 4; ++sg) {
                const int col = cc + 16 * sg + 4 * lg;
                const bool ok = col < ldx;
                const int colc = ok ? col : 0;
                double4 xa = *reinterpret_cast<const double4*>(xrow + colc);
                double4 yv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) yv[q] = *reinterpret_cast<const double4*>(yrow0 + (size_t)(16 * q) * ldx + colc);
                if (!ok) xa = zero4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.x, yv[q].x, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.y, yv[q].y, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.z, yv[q].z, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.w, yv[q].w, acc[q], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) Mp[wave][lg + 4 * r][16 * q + li] = acc[q][r];
        __syncthreads();
        double sreg[4] = {0.0, 0.0, 0.0, 0.0};
        if (tid < 256) {
            const int row = tid >> 4, pc = tid & 15;
#pragma unroll
            for (int wv = 0; wv < NW; ++wv)
#pragma unroll
                for (int q = 0; q < 4; ++q) sreg[q] += Mp[wv][row][pc + 16 * q];
        }
        __syncthreads();                                          // the partials are consumed: their space takes C_b
        if (tid < 256) {
            const int row = tid >> 4, pc = tid & 15;
#pragma unroll
            for (int q = 0; q < 4; ++q) Ms[row][pc + 16 * q] = sreg[q];
        }
#pragma unroll
        for (int u = 0; u < CR; ++u) { const int e = tid + u * 64 * NW; Cs[e >> 6][e & 63] = creg[u]; }
        __syncthreads();
        // ---- phase 2: M2 = -M C -----------------------------------------------------------------------
        if (tid < 256) {
            const int row = tid >> 4, qc = tid & 15;
            double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
            for (int pp = 0; pp < WY_NB2; ++pp) {
                const double m = Ms[row][pp];
#pragma unroll
                for (int q = 0; q < 4; ++q) s4[q] += m * Cs[pp][qc + 16 * q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) M2s[row][qc + 16 * q] = -s4[q];
        }
        __syncthreads();
        // ---- phase 3: X += (-M2) Y, the 64 reflectors in two halves -----------------------------------
        double m2a[16];
#pragma unroll
        for (int kt = 0; kt < 16; ++kt) m2a[kt] = M2s[li][4 * kt + lg];
        const double* ybase = Yf + (size_t)(j0 + lg) * ldx;
        for (int cc = cw; cc < ldx; cc += 64 * NW) {
            const int col = cc + 4 * li;
            const bool ok = col < ldx;
            const int colc = ok ? col : 0;
            double4 xv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) xv[r] = *reinterpret_cast<const double4*>(X + (size_t)rowD[r] * ldx + colc);
            wy_f64x4 d0, d1, d2, d3;
#pragma unroll
            for (int r = 0; r < 4; ++r) { d0[r] = xv[r].x; d1[r] = xv[r].y; d2[r] = xv[r].z; d3[r] = xv[r].w; }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double4 yv[8];
#pragma unroll
                for (int kt = 0; kt < 8; ++kt)
                    yv[kt] = *reinterpret_cast<const double4*>(ybase + (size_t)(32 * h + 4 * kt) * ldx + colc);
#pragma unroll
                for (int kt = 0; kt < 8; ++kt) {
                    d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(m2a[8 * h + kt], yv[kt].x, d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(m2a[8 * h + kt], yv[kt].y, d1, 0, 0, 0);
                    d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(m2a[8 * h + kt], yv[kt].z, d2, 0, 0, 0);
                    d3 = __builtin_amdgcn_mfma_f64_16x16x4f64(m2a[8 * h + kt], yv[kt].w, d3, 0, 0, 0);
                }
            }
            if (ok) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (r0 + lg + 4 * r < n)
                        *reinterpret_cast<double4*>(X + (size_t)rowD[r] * ldx + col) = make_double4(d0[r], d1[r], d2[r], d3[r]);
            }
        }
        __syncthreads();
    }
}

// ---- the strip of X in registers for the whole sweep (round 5) -------------------------------------------------------------
// wy_apply_mfma64_kernel is bound by what the L2s deliver: per block a workgroup streams its 16 rows of X twice and writes
// them once, and streams the reflector block twice (19.9 GB through the L2s at n = 3072, of which 5.4 GB are X).  Here the
// strip never leaves the CU: 8 wavefronts hold the 16 x n strip in their accumulator registers — wavefront w the 64-column
// chunks w, w + 8, ... (dealt cyclically: the active columns of a block are a suffix, so every wavefront keeps about the
// same share) — for ALL blocks; only the reflectors stream.  What makes that possible is the TRANSPOSED tile: a chunk is
// kept as four 16 x 16 tiles of X^T in the MFMA C/D layout (lane (i, g), register r: X[row i][64 ch + 16 r + 4 g + c] for
// tile c), updated as  X^T -= Y_b^T M2^T  (A operand: 32-byte vectors of a reflector row, columns 4 i .. 4 i + 3 — the same
// loads as above), and the very same registers are the B operand of the first product  M^T = Y_b X^T  (columns
// {16 r + 4 g + c : g} of a chunk are the four summation slots of MFMA (r, c); the A operand is again a 32-byte vector of a
// reflector row).  X is read once and written once per eigh.  Needs n = ld = a multiple of 64 with at most 8 NCH chunks.
template <int NCH, int NW>
__global__ __launch_bounds__(64 * NW) void wy_apply_strip_kernel(double* __restrict__ X, int ldx, int n,
                                                                 const double* __restrict__ Yf,
                                                                 const double* __restrict__ Call, int nblk) {
    __shared__ double Mp[NW][WY_NB2][17];                        // partial M^T of every wavefront: 64 reflectors x 16 rows
    __shared__ double Ms[WY_NB2][17];
    __shared__ double M2s[WY_NB2][17];
    __shared__ double Cs[WY_NB2][WY_NB2 + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int r0 = blockIdx.x * 16;
    const int row = (r0 + li < n) ? r0 + li : n - 1;
    const int nchunks = ldx >> 6;
    // ---- the strip: xr[i][c][r] = X[row][64 (wave + 8 i) + 16 r + 4 lg + c]
    wy_f64x4 xr[NCH][4];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = wave + NW * i;
        const bool has = ch < nchunks;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double4 v = has ? *reinterpret_cast<const double4*>(X + (size_t)row * ldx + 64 * ch + 16 * r + 4 * lg)
                                  : make_double4(0.0, 0.0, 0.0, 0.0);
            xr[i][0][r] = v.x; xr[i][1][r] = v.y; xr[i][2][r] = v.z; xr[i][3][r] = v.w;
        }
    }
    for (int b = nblk - 1; b >= 0; --b) {
        const int j0 = b * WY_NB2;
        const int ch0 = (j0 + 1) >> 6;                           // first chunk with a non-zero reflector entry
        // C_b into LDS (consumed in phase 2, behind two barriers)
        for (int e = tid; e < WY_NB2 * WY_NB2; e += 64 * NW) Cs[e >> 6][e & 63] = Call[(size_t)b * WY_NB2 * WY_NB2 + e];
        // ---- phase 1: M^T = Y_b X^T (64 x 16), this wavefront's chunks; the 64 reflectors in two halves (the strip leaves
        // 64 registers for everything else: two accumulator tiles and two operand vectors at a time)
        // (operand addresses = uniform base + a 32-bit lane offset: the base stays in scalar registers)
        const unsigned offA = (unsigned)(li * ldx + 4 * lg);
#pragma unroll
        for (int qh = 0; qh < 2; ++qh) {
            wy_f64x4 accM0 = wy_f64x4{0.0, 0.0, 0.0, 0.0}, accM1 = wy_f64x4{0.0, 0.0, 0.0, 0.0};
            const double* yq = Yf + (size_t)(j0 + 32 * qh) * ldx;               // uniform
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int ch = wave + NW * i;
                if (ch >= ch0 && ch < nchunks) {
                    const double* yc = yq + 64 * ch;                             // uniform
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double4 y0 = *reinterpret_cast<const double4*>(yc + 16 * r + offA);
                        const double4 y1 = *reinterpret_cast<const double4*>(yc + (size_t)16 * ldx + 16 * r + offA);
                        accM0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y0.x, xr[i][0][r], accM0, 0, 0, 0);
                        accM1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y1.x, xr[i][0][r], accM1, 0, 0, 0);
                        accM0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y0.y, xr[i][1][r], accM0, 0, 0, 0);
                        accM1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y1.y, xr[i][1][r], accM1, 0, 0, 0);
                        accM0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y0.z, xr[i][2][r], accM0, 0, 0, 0);
                        accM1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y1.z, xr[i][2][r], accM1, 0, 0, 0);
                        accM0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y0.w, xr[i][3][r], accM0, 0, 0, 0);
                        accM1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y1.w, xr[i][3][r], accM1, 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                Mp[wave][32 * qh + lg + 4 * r][li] = accM0[r];
                Mp[wave][32 * qh + 16 + lg + 4 * r][li] = accM1[r];
            }
        }
        __syncthreads();
        // the eight partials in a fixed order: 1024 entries, two per thread
#pragma unroll
        for (int u = 0; u < 1024 / (64 * NW); ++u) {
            const int e = tid + 64 * NW * u, j = e >> 4, rr = e & 15;
            double sacc = 0.0;
#pragma unroll
            for (int wv = 0; wv < NW; ++wv) sacc += Mp[wv][j][rr];
            Ms[j][rr] = sacc;
        }
        __syncthreads();
        // ---- phase 2: M2^T = -C_b^T M^T
#pragma unroll
        for (int u = 0; u < 1024 / (64 * NW); ++u) {
            const int e = tid + 64 * NW * u, qc = e >> 4, rr = e & 15;
            double sacc = 0.0;
#pragma unroll 8
            for (int pp = 0; pp < WY_NB2; ++pp) sacc += Ms[pp][rr] * Cs[pp][qc];
            M2s[qc][rr] = -sacc;
        }
        __syncthreads();
        // ---- phase 3: X^T += Y_b^T M2^T on this wavefront's chunks
        const unsigned offB = (unsigned)(lg * ldx + 4 * li);
        const double* yb = Yf + (size_t)j0 * ldx;                                // uniform
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ch = wave + NW * i;
            if (ch >= ch0 && ch < nchunks) {
                const double* yc = yb + 64 * ch;                                 // uniform
#pragma unroll
                for (int kp = 0; kp < 8; ++kp) {
                    const double4 y0 = *reinterpret_cast<const double4*>(yc + (size_t)(8 * kp) * ldx + offB);
                    const double4 y1 = *reinterpret_cast<const double4*>(yc + (size_t)(8 * kp + 4) * ldx + offB);
                    const double m20 = M2s[8 * kp + lg][li], m21 = M2s[8 * kp + 4 + lg][li];
                    xr[i][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(y0.x, m20, xr[i][0], 0, 0, 0);
                    xr[i][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(y0.y, m20, xr[i][1], 0, 0, 0);
                    xr[i][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(y0.z, m20, xr[i][2], 0, 0, 0);
                    xr[i][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(y0.w, m20, xr[i][3], 0, 0, 0);
                    xr[i][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(y1.x, m21, xr[i][0], 0, 0, 0);
                    xr[i][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(y1.y, m21, xr[i][1], 0, 0, 0);
                    xr[i][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(y1.z, m21, xr[i][2], 0, 0, 0);
                    xr[i][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(y1.w, m21, xr[i][3], 0, 0, 0);
                }
            }
        }
        // (the next block's first barrier separates this block's reads of M2s / Cs from their next writes: Cs is rewritten
        // at the top of the loop, so one barrier here)
        __syncthreads();
    }
    if (r0 + li < n) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ch = wave + NW * i;
            if (ch < nchunks) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    *reinterpret_cast<double4*>(X + (size_t)row * ldx + 64 * ch + 16 * r + 4 * lg) =
                        make_double4(xr[i][0][r], xr[i][1][r], xr[i][2][r], xr[i][3][r]);
            }
        }
    }
}

// The same sweep with 32 rows of X per workgroup (two MFMA row tiles sharing every reflector fetch).  The kernel above
// is bound by L2 bandwidth — 22.7 GB of requests in 3.16 ms at n = 3072, two thirds of them the reflector blocks Y_b, which
// every workgroup streams in full twice per block (PMC pass in profiles/, 4 / 8 / 16 wavefronts per workgroup measured
// equal) — so the lever is fewer workgroups per reflector byte, not more wavefronts: with two row tiles the Y traffic
// per row of X halves.  n / 32 workgroups of NW wavefronts; the MFMA work of a workgroup doubles, which is where the
// two limits meet at n = 3072 (96 CUs busy).
template <int NW>
__global__ __launch_bounds__(64 * NW) void wy_apply_mfma2_kernel(double* __restrict__ X, int ldx, int n,
                                                             const double* __restrict__ Yf,
                                                             const double* __restrict__ Call, int nblk) {
    __shared__ double Mp[NW][32][33];
    __shared__ double Ms[32][33];
    __shared__ double M2s[32][33];
    __shared__ double Cs[WY_NB][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int r0 = blockIdx.x * 32;
    int rowA[2], rowD[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        rowA[t] = (r0 + 16 * t + li < n) ? r0 + 16 * t + li : n - 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) rowD[t][r] = (r0 + 16 * t + lg + 4 * r < n) ? r0 + 16 * t + lg + 4 * r : n - 1;
    }
    const double4 zero4 = make_double4(0.0, 0.0, 0.0, 0.0);
    for (int b = nblk - 1; b >= 0; --b) {
        const int j0 = b * WY_NB;
        const int cs = ((j0 + 1) >> 6) << 6;
        const int cw = cs + 64 * ((wave - (cs >> 6)) & (NW - 1));
        for (int e = tid; e < WY_NB * WY_NB; e += 64 * NW) Cs[e >> 5][e & 31] = Call[(size_t)b * WY_NB * WY_NB + e];
        // ---- phase 1: M = X Y^T (32 x 32) ---------------------------------------------------------
        wy_f64x4 acc[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) { acc[t][0] = wy_f64x4{0.0, 0.0, 0.0, 0.0}; acc[t][1] = wy_f64x4{0.0, 0.0, 0.0, 0.0}; }
        const double* xrow0 = X + (size_t)rowA[0] * ldx;
        const double* xrow1 = X + (size_t)rowA[1] * ldx;
        const double* y0row = Yf + (size_t)(j0 + li) * ldx;
        const double* y1row = y0row + (size_t)16 * ldx;
        for (int cc = cw; cc < ldx; cc += 64 * NW) {
#pragma unroll
            for (int sg = 0; sg < 4; ++sg) {
                const int col = cc + 16 * sg + 4 * lg;
                const bool ok = col < ldx;
                const int colc = ok ? col : 0;
                double4 xa = *reinterpret_cast<const double4*>(xrow0 + colc);
                double4 xb = *reinterpret_cast<const double4*>(xrow1 + colc);
                const double4 ya = *reinterpret_cast<const double4*>(y0row + colc);
                const double4 yb = *reinterpret_cast<const double4*>(y1row + colc);
                if (!ok) { xa = zero4; xb = zero4; }
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.x, ya.x, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.x, yb.x, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(xb.x, ya.x, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(xb.x, yb.x, acc[1][1], 0, 0, 0);
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.y, ya.y, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.y, yb.y, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(xb.y, ya.y, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(xb.y, yb.y, acc[1][1], 0, 0, 0);
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.z, ya.z, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.z, yb.z, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(xb.z, ya.z, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(xb.z, yb.z, acc[1][1], 0, 0, 0);
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.w, ya.w, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.w, yb.w, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(xb.w, ya.w, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(xb.w, yb.w, acc[1][1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                Mp[wave][16 * t + lg + 4 * r][li] = acc[t][0][r];
                Mp[wave][16 * t + lg + 4 * r][16 + li] = acc[t][1][r];
            }
        __syncthreads();
        for (int e = tid; e < 32 * 32; e += 64 * NW) {
            const int row = e >> 5, pc = e & 31;
            double s0 = 0.0;
#pragma unroll
            for (int wv = 0; wv < NW; ++wv) s0 += Mp[wv][row][pc];
            Ms[row][pc] = s0;
        }
        __syncthreads();
        // ---- phase 2: M2 = -M C ---------------------------------------------------------------------
        for (int e = tid; e < 32 * 32; e += 64 * NW) {
            const int row = e >> 5, qc = e & 31;
            double s0 = 0.0;
#pragma unroll 8
            for (int pp = 0; pp < WY_NB; ++pp) s0 += Ms[row][pp] * Cs[pp][qc];
            M2s[row][qc] = -s0;
        }
        __syncthreads();
        // ---- phase 3: X += (-M2) Y ------------------------------------------------------------------
        double m2a[2][8];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) m2a[t][kt] = M2s[16 * t + li][4 * kt + lg];
        const double* ybase = Yf + (size_t)(j0 + lg) * ldx;
        for (int cc = cw; cc < ldx; cc += 64 * NW) {
            const int col = cc + 4 * li;
            const bool ok = col < ldx;
            const int colc = ok ? col : 0;
            double4 yv[8];
#pragma unroll
            for (int kt = 0; kt < 8; ++kt)
                yv[kt] = *reinterpret_cast<const double4*>(ybase + (size_t)(4 * kt) * ldx + colc);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                double4 xv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) xv[r] = *reinterpret_cast<const double4*>(X + (size_t)rowD[t][r] * ldx + colc);
                wy_f64x4 d0, d1, d2, d3;
#pragma unroll
                for (int r = 0; r < 4; ++r) { d0[r] = xv[r].x; d1[r] = xv[r].y; d2[r] = xv[r].z; d3[r] = xv[r].w; }
#pragma unroll
                for (int kt = 0; kt < 8; ++kt) {
                    d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(m2a[t][kt], yv[kt].x, d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(m2a[t][kt], yv[kt].y, d1, 0, 0, 0);
                    d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(m2a[t][kt], yv[kt].z, d2, 0, 0, 0);
                    d3 = __builtin_amdgcn_mfma_f64_16x16x4f64(m2a[t][kt], yv[kt].w, d3, 0, 0, 0);
                }
                if (ok) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (r0 + 16 * t + lg + 4 * r < n)
                            *reinterpret_cast<double4*>(X + (size_t)rowD[t][r] * ldx + col) = make_double4(d0[r], d1[r], d2[r], d3[r]);
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
struct Node {
    int lo, hi, mid;      // mid < 0 for a leaf
    int left, right;
};

void build_tree(int lo, int hi, int depth, int maxdepth, std::vector<Node>& nodes,
                std::vector<std::vector<int>>& by_height, int* self) {
    Node nd;
    nd.lo = lo; nd.hi = hi; nd.mid = -1; nd.left = nd.right = -1;
    const int idx = (int)nodes.size();
    nodes.push_back(nd);
    *self = idx;
    if (depth < maxdepth) {
        const int mid = lo + (hi - lo) / 2;
        int l, r;
        build_tree(lo, mid, depth + 1, maxdepth, nodes, by_height, &l);
        build_tree(mid, hi, depth + 1, maxdepth, nodes, by_height, &r);
        nodes[idx].mid = mid;
        nodes[idx].left = l;
        nodes[idx].right = r;
    }
    by_height[maxdepth - depth].push_back(idx);
}

struct EighWork {
    sella_ctx* c;
    int n, ld;
    double *A;                 // working copy (n x n), destroyed
    double *Za, *Zb;           // eigenvector rows, ping-pong
    double *Zc, *Ut;           // compacted rows / inner eigenvectors
    double *vec;               // vectors of length ld (slots)
    int* ibuf;                 // device ints
};

// vec slots (each ld doubles)
enum { V_D = 0, V_E, V_W, V_Z, V_CS0, V_CS1, V_DD, V_WD, V_TAU, V_ZH, V_LAM, V_TAUS, V_U0, V_U1, V_WRAW, V_COL, V_WRAW2,
       V_NSLOTS };

struct MergePlan {
    int lo, N, K, nrot;
    double rho;
    std::vector<int> nondef, defl, r1, r2;
    std::vector<double> cs;
};

}  // namespace

// Deflation of one rank-one modification diag(D) + rho z z^T (LAPACK dlaed2's rules): entries with a
// negligible weight are set aside, and of two (nearly) equal poles one is rotated out.  D and zz are
// modified (rotations); pl.rho, pl.lo, pl.N must be set by the caller.  Indices are local (0..N-1).
// Stable ascending order of v[0..N): the inputs here are sorted already (eigenvalues carried by a rank-one
// update) or a few sorted runs back to back (the halves of a merge; new roots followed by deflated values, with a
// rotated-out entry or two out of place), so the order is the identity or a couple of linear merges; anything
// else falls back to a sort.  Same result as std::stable_sort on the indices, at a fraction of its ~100 us for
// N = 3072.
// (order: N ints; tmp: N ints of scratch for the merges — nothing is allocated here: at the lowest levels of divide &
// conquer this runs hundreds of times per level on a dozen entries each)
static void ascending_order(const double* v, int N, int* order, int* tmp) {
    for (int i = 0; i < N; ++i) order[i] = i;
    constexpr int MAXCUTS = 32;
    int bounds[MAXCUTS + 3], nb = 0;                    // starts of the ascending runs, then N
    bounds[nb++] = 0;
    for (int i = 1; i < N && nb <= MAXCUTS + 1; ++i)
        if (v[i] < v[i - 1]) bounds[nb++] = i;
    if (nb == 1) return;
    auto less = [&](int a, int b) { return v[a] < v[b]; };
    if (nb > MAXCUTS + 1) { std::stable_sort(order, order + N, less); return; }
    bounds[nb++] = N;
    // natural merge sort over the few runs (a deflation rotation or two perturbs an otherwise ordered list); merges are
    // stable: of equal values the one from the earlier run comes first, as std::stable_sort would leave them
    while (nb > 2) {
        int k = 0, nn = 0;
        for (; k + 2 < nb; k += 2) {
            int* lo = order + bounds[k];
            const int n1 = bounds[k + 1] - bounds[k], n2 = bounds[k + 2] - bounds[k + 1];
            std::copy(lo, lo + n1, tmp);                                  // first run aside, merged back in place
            int i = 0, j = 0, o = 0;
            const int* second = lo + n1;
            while (i < n1 && j < n2) {
                if (less(second[j], tmp[i])) lo[o++] = second[j++];
                else lo[o++] = tmp[i++];
            }
            while (i < n1) lo[o++] = tmp[i++];
            bounds[nn++] = bounds[k];
        }
        for (; k < nb; ++k) bounds[nn++] = bounds[k];
        if (bounds[nn - 1] != N) bounds[nn++] = N;
        nb = nn;
    }
}

// One deflation plan written into caller-provided arrays (each of N entries, cs of 2 N): nothing allocated.
struct PlanOut {
    int *nondef, *defl, *r1, *r2;
    double* cs;
    int K, ndefl, nrot;
};

static void plan_deflation_core(int N, double* D, double* zz, double rho, int* order, int* tmp, PlanOut& out) {
    const double eps = 2.220446049250313e-16;
    double zmax = 0.0, dmax = 0.0;
    for (int i = 0; i < N; ++i) {
        zmax = std::max(zmax, fabs(zz[i]));
        dmax = std::max(dmax, fabs(D[i]));
    }
    const double tol = 8.0 * eps * std::max(dmax, zmax);
    ascending_order(D, N, order, tmp);
    int K = 0, nd = 0, nrot = 0;
    if (rho * zmax <= tol) {
        for (int i = 0; i < N; ++i) out.defl[nd++] = order[i];
    } else {
        int pj = -1;
        for (int jj = 0; jj < N; ++jj) {
            const int nj = order[jj];
            if (rho * fabs(zz[nj]) <= tol) { out.defl[nd++] = nj; continue; }
            if (pj < 0) { pj = nj; continue; }
            double s = zz[pj], cc = zz[nj];
            const double tau = hypot(cc, s);
            const double t = D[nj] - D[pj];
            cc /= tau;
            s = -s / tau;
            if (fabs(t * cc * s) <= tol) {
                zz[nj] = tau;
                zz[pj] = 0.0;
                out.r1[nrot] = pj; out.r2[nrot] = nj; out.cs[2 * nrot] = cc; out.cs[2 * nrot + 1] = s;
                ++nrot;
                const double tt = D[pj] * cc * cc + D[nj] * s * s;
                D[nj] = D[pj] * s * s + D[nj] * cc * cc;
                D[pj] = tt;
                out.defl[nd++] = pj;
                pj = nj;
            } else {
                out.nondef[K++] = pj;
                pj = nj;
            }
        }
        if (pj >= 0) out.nondef[K++] = pj;
    }
    out.K = K;
    out.ndefl = nd;
    out.nrot = nrot;
}

static void plan_deflation(int N, double* D, double* zz, MergePlan& pl) {
    std::vector<int> work(2 * (size_t)N);
    pl.nondef.assign(N, 0); pl.defl.assign(N, 0); pl.r1.assign(N, 0); pl.r2.assign(N, 0); pl.cs.assign(2 * (size_t)N, 0.0);
    PlanOut out;
    out.nondef = pl.nondef.data(); out.defl = pl.defl.data(); out.r1 = pl.r1.data(); out.r2 = pl.r2.data(); out.cs = pl.cs.data();
    plan_deflation_core(N, D, zz, pl.rho, work.data(), work.data() + N, out);
    pl.nondef.resize(out.K); pl.defl.resize(out.ndefl); pl.r1.resize(out.nrot); pl.r2.resize(out.nrot); pl.cs.resize(2 * (size_t)out.nrot);
    pl.K = out.K;
    pl.nrot = out.nrot;
}

// Divide and conquer on the tridiagonal (d, e) (host copies, modified).  On exit wout holds the
// ascending eigenvalues and W.Za the eigenvectors as rows in matching order.
static int dc_solve(EighWork& W, std::vector<double>& d, std::vector<double>& e, double* wout) {
    sella_ctx* c = W.c;
    const int n = W.n, ld = W.ld;
    const int leaf = (int)c->opt.eigh_leaf;
    int maxdepth = 0;
    while (((n + (1 << maxdepth) - 1) >> maxdepth) > leaf) ++maxdepth;
    std::vector<Node> nodes;
    std::vector<std::vector<int>> by_height(maxdepth + 1);
    int root;
    build_tree(0, n, 0, maxdepth, nodes, by_height, &root);

    // tear: T = diag(T1', T2') + |e_m| u u^T at every internal node
    for (const Node& nd : nodes)
        if (nd.mid >= 0) {
            const double em = fabs(e[nd.mid - 1]);
            d[nd.mid - 1] -= em;
            d[nd.mid] -= em;
        }
    double* ddev = W.vec + (size_t)V_D * ld;
    double* edev = W.vec + (size_t)V_E * ld;
    double* wdev = W.vec + (size_t)V_W * ld;
    const std::vector<int>& leaves = by_height[0];
    // pinned staging (see the level loop below): torn d | e in the layout of the device slots, leaf ranges behind
    void* stage;
    const size_t nmmax = (size_t)n / 2 + 8;
    const size_t stage_doubles = 4 * (size_t)ld + 2 * (size_t)n + 8;
    const size_t stage_ints = 5 * (size_t)n + 4 * nmmax + 16;
    SCHK(host_stage(c, stage_doubles * sizeof(double) + stage_ints * sizeof(int) + 2 * nmmax * sizeof(MergeDev), &stage));
    {
        double* hde = static_cast<double*>(stage);
        std::copy(d.begin(), d.end(), hde);
        std::copy(e.begin(), e.end(), hde + ld);
        SCHK(h2d_pinned(c, ddev, hde, ((size_t)ld + n) * sizeof(double)));
    }
    int* ranges = reinterpret_cast<int*>(static_cast<double*>(stage) + 2 * (size_t)ld);
    const size_t nranges = 2 * leaves.size();
    for (size_t q = 0; q < leaves.size(); ++q) { ranges[2 * q] = nodes[leaves[q]].lo; ranges[2 * q + 1] = nodes[leaves[q]].hi; }
    // device int layout: [0,8) info | [8, 8+2*nleaves) leaf ranges | then 4 arrays of n ints
    int* info = W.ibuf;
    int* rdev = W.ibuf + 8;
    int* ibase = W.ibuf + 8 + 2 * (int)leaves.size() + 8;
    int* i1d = ibase;
    int* i2d = ibase + n;
    int* idxd = ibase + 2 * n;
    int* orgd = ibase + 3 * n;
    int* mrowd = ibase + 4 * n;                       // merge index of every row (per level)
    int* gdescd = ibase + 5 * n;                      // GEMM descriptors, 4 ints per merge
    double* mdraw;                                    // per-merge parameters (the stage-1 partial buffer is free now)
    SCHK(scratch_get(c, SCR_MISC1, ((size_t)n / 2 + 8) * sizeof(MergeDev), &mdraw));
    MergeDev* mdd = reinterpret_cast<MergeDev*>(mdraw);
    HIPCHK(s_memset0(c, info, 8 * sizeof(int)));
    HIPCHK(hipMemcpyAsync(rdev, ranges, nranges * sizeof(int), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(set_identity_kernel, dim3((n + 255) / 256, n), dim3(256), 0, c->stream, W.Za, ld, n);
    hipLaunchKernelGGL(leaf_ql_kernel, dim3((unsigned)leaves.size()), dim3(64), 0, c->stream, ddev, edev, wdev,
                       rdev, W.Za, ld, info);
    HIPCHK(hipGetLastError());
    std::vector<double> vals(n);
    int hinfo[2];
    {
        double* hv = static_cast<double*>(stage) + 3 * (size_t)ld;
        int* hinfo_p = reinterpret_cast<int*>(hv + n);
        HIPCHK(hipMemcpyAsync(hv, wdev, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(hinfo_p, info, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        SCHK(stream_wait(c));
        std::copy(hv, hv + n, vals.begin());
        hinfo[0] = hinfo_p[0];
        hinfo[1] = hinfo_p[1];
    }
    if (hinfo[0] != 0) {
        set_error("eigh: QL iteration did not converge in a leaf (eigenvalue %d)", hinfo[0]);
        return SELLA_E_NOCONV;
    }

    double* cur = W.Za;
    double* nxt = W.Zb;
    double* zdev = W.vec + (size_t)V_Z * ld;
    double* csd = W.vec + (size_t)V_CS0 * ld;          // 2 slots: (c, s) pairs
    double* Dd = W.vec + (size_t)V_DD * ld;
    double* wd = W.vec + (size_t)V_WD * ld;
    double* taud = W.vec + (size_t)V_TAU * ld;
    double* zhd = W.vec + (size_t)V_ZH * ld;
    double* lamd = W.vec + (size_t)V_LAM * ld;
    // Per-level host <-> device traffic goes through one pinned staging buffer laid out like the device side
    // (the (c,s) | D | w slots and the int arrays are contiguous there), so a level is 4 asynchronous uploads
    // and 2 downloads instead of 13 pageable — i.e. synchronous, ~20 us each — copies.
    double* hcs = static_cast<double*>(stage);                 // mirrors csd (2 slots), Dd, wd
    double* hD = hcs + 2 * (size_t)ld;
    double* hw = hcs + 3 * (size_t)ld;
    double* z = hcs + 4 * (size_t)ld;
    double* lam = z + n;
    int* hint = reinterpret_cast<int*>(lam + n + 8);           // [hi2 (8)] then mirrors i1d | i2d | idxd
    int* hi2 = hint;
    int* hr1 = hint + 8;
    int* hr2 = hr1 + n;
    int* hidx = hr1 + 2 * (size_t)n;
    int* hmrow = hr1 + 3 * (size_t)n;                          // mirrors mrowd | gdescd (device: + n for orgd in between)
    int* hgd = hmrow + n;
    MergeDev* hmd2[2];                                         // per-merge parameters, double-buffered by level parity
    hmd2[0] = reinterpret_cast<MergeDev*>(hgd + 4 * nmmax + 8 - ((4 * nmmax + 8) & 1));
    hmd2[1] = hmd2[0] + nmmax;
    std::vector<int> order;
    // Eigenvector rows of a node are supported on its own column range only, so everything outside the
    // diagonal blocks must read as zero.  One clearing of the second buffer suffices: level h overwrites
    // its diagonal blocks completely (K updated + N-K deflated rows of N columns each), and the blocks
    // this buffer held two levels earlier lie inside them.
    HIPCHK(s_memset0(c, nxt, (size_t)n * ld * sizeof(double)));
    // (1) of a level — the rank-one vectors of all its merges: one launch and one download, queued (not waited for) by
    // the level BEFORE it, right behind that level's last kernel: the vectors are rows of the eigenvector blocks just
    // written, nothing the host has to decide first.  So a level costs ONE synchronisation, which hands over the previous
    // level's eigenvalues and this level's vectors together (it was two, with the launch of this small kernel and its
    // download exposed in between: about 50 us per level, eight levels at 3N = 3072).
    const bool pipelined = c->opt.eigh_dc_pipeline != 0;
    const bool dc_timing = getenv("SELLA_DC_TIMING") != nullptr;
    auto queue_level_vectors = [&](int h, const double* rows) -> int {
        const std::vector<int>& lvl = by_height[h];
        MergeDev* hmd = hmd2[h & 1];
        const int nm = (int)lvl.size();
        int maxN = 0;
        for (int mi = 0; mi < nm; ++mi) {
            const Node& nd = nodes[lvl[mi]];
            MergeDev& m = hmd[mi];
            m.lo = nd.lo; m.N = nd.hi - nd.lo; m.K = 0; m.nrot = 0;
            m.n1 = nd.mid - nd.lo; m.mid = nd.mid;
            m.rho = 0.0; m.sgn = e[nd.mid - 1] < 0 ? -1.0 : 1.0;
            maxN = std::max(maxN, m.N);
        }
        HIPCHK(hipMemcpyAsync(mdd, hmd, (size_t)nm * sizeof(MergeDev), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(gather_z_batched_kernel, dim3((maxN + 255) / 256, nm), dim3(256), 0, c->stream, mdd, rows, ld, zdev);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(z, zdev, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        return SELLA_OK;
    };
    // Plans of a level in flat arrays indexed by matrix position (the blocks of a level's merges are disjoint): nothing is
    // allocated per merge — at the lowest levels that was 1,500 small allocations per level, a third of the planning time.
    // The deflated indices are kept for two levels (the values they select are taken after the NEXT wait).
    struct LevelPlan { int lo, N, K, nrot; double rho; };
    std::vector<LevelPlan> plans, plans_prev;
    std::vector<int> nondef_all(n), defl_all[2] = {std::vector<int>(n), std::vector<int>(n)}, order_all(2 * (size_t)n);
    std::vector<double> nv(n);
    // eigenvalues of a finished level into `vals` (new roots, then the deflated values), after its download has arrived
    auto take_level_values = [&](const std::vector<LevelPlan>& done, const std::vector<int>& defl) -> int {
        if (hi2[1] != 0) {
            if (getenv("SELLA_DEBUG")) {
                for (const LevelPlan& pl : done) {
                    if (hi2[1] - 1 >= pl.K) continue;
                    fprintf(stderr, "secular fail? merge lo=%d N=%d K=%d rho=%.17g root=%d\n", pl.lo, pl.N, pl.K, pl.rho, hi2[1] - 1);
                }
            }
            set_error("eigh: secular equation solver hit its iteration cap (root %d)", hi2[1] - 1);
            return SELLA_E_NOCONV;
        }
        for (const LevelPlan& pl : done) {
            double* D = vals.data() + pl.lo;
            const int* df = defl.data() + pl.lo;
            for (int p = 0; p < pl.K; ++p) nv[p] = lam[pl.lo + p];
            for (int p = 0; p < pl.N - pl.K; ++p) nv[pl.K + p] = D[df[p]];
            for (int p = 0; p < pl.N; ++p) D[p] = nv[p];
        }
        return SELLA_OK;
    };
    if (pipelined && maxdepth >= 1) SCHK(queue_level_vectors(1, cur));
    for (int h = 1; h <= maxdepth; ++h) {
        const std::vector<int>& lvl = by_height[h];
        MergeDev* hmd = hmd2[h & 1];
        const int nm = (int)lvl.size();
        int maxN = 0;
        for (int mi = 0; mi < nm; ++mi) maxN = std::max(maxN, nodes[lvl[mi]].hi - nodes[lvl[mi]].lo);
        if (!pipelined) SCHK(queue_level_vectors(h, cur));
        const auto tw0 = std::chrono::steady_clock::now();
        SCHK(stream_wait(c));
        const auto tw1 = std::chrono::steady_clock::now();
        // (only now: the staging arrays below are the sources of the previous level's uploads, which have been executed
        // for certain only behind this wait)
        for (int mi = 0; mi < nm; ++mi) {
            const Node& nd = nodes[lvl[mi]];
            for (int p = nd.lo; p < nd.hi; ++p) hmrow[p] = mi;
        }
        if (pipelined && h > 1) SCHK(take_level_values(plans_prev, defl_all[(h - 1) & 1]));
        // ---- (2) deflation of every merge on the host (dlaed2 logic) ------------------------------
        plans.resize(lvl.size());
        std::vector<int>& defl_lvl = defl_all[h & 1];
        for (size_t mi = 0; mi < lvl.size(); ++mi) {
            const Node& nd = nodes[lvl[mi]];
            LevelPlan& pl = plans[mi];
            const int lo = nd.lo, N = nd.hi - nd.lo;
            double* D = vals.data() + lo;
            double* zz = z + lo;
            pl.lo = lo;
            pl.N = N;
            pl.rho = fabs(2.0 * e[nd.mid - 1]);
            for (int i = 0; i < N; ++i) zz[i] *= 0.7071067811865476;
            // rotations go straight into the staging arrays (offset lo inside n-length host arrays; blocks of different
            // merges are disjoint), the index lists into the level-wide arrays
            PlanOut out;
            out.nondef = nondef_all.data() + lo; out.defl = defl_lvl.data() + lo;
            out.r1 = hr1 + lo; out.r2 = hr2 + lo; out.cs = hcs + 2 * (size_t)lo;
            plan_deflation_core(N, D, zz, pl.rho, order_all.data() + lo, order_all.data() + n + lo, out);
            pl.K = out.K;
            pl.nrot = out.nrot;
            for (int p = 0; p < pl.K; ++p) {
                hD[lo + p] = D[out.nondef[p]];
                hw[lo + p] = zz[out.nondef[p]];
                hidx[lo + p] = lo + out.nondef[p];
            }
            for (int p = 0; p < N - pl.K; ++p) hidx[lo + pl.K + p] = lo + out.defl[p];
        }
        const auto tw2 = std::chrono::steady_clock::now();
        // (c,s) pairs | D | w in one piece (device slots V_CS0, V_CS1, V_DD, V_WD are consecutive), rotation and
        // gather indices in another
        static_assert(V_CS1 == V_CS0 + 1 && V_DD == V_CS0 + 2 && V_WD == V_CS0 + 3, "slot order");
        // (by kernel from 16 KB on: the runtime's copy takes 12 - 15 us at these sizes, h2d_kernel_min)
        SCHK(h2d_pinned(c, csd, hcs, (3 * (size_t)ld + n) * sizeof(double)));
        SCHK(h2d_pinned(c, i1d, hr1, 3 * (size_t)n * sizeof(int)));
        // ---- (3) device work of the whole level: one launch per kernel, blockIdx.y = merge -------------
        {
            int maxK = 0, maxrot = 0;
            for (int mi = 0; mi < nm; ++mi) {
                const LevelPlan& pl = plans[mi];
                hmd[mi].K = pl.K;
                hmd[mi].nrot = pl.nrot;
                hmd[mi].rho = pl.rho;
                maxK = std::max(maxK, pl.K);
                maxrot = std::max(maxrot, pl.nrot);
                hgd[4 * mi] = pl.lo; hgd[4 * mi + 1] = pl.N; hgd[4 * mi + 2] = pl.K; hgd[4 * mi + 3] = 0;
            }
            HIPCHK(hipMemcpyAsync(mdd, hmd, (size_t)nm * sizeof(MergeDev), hipMemcpyHostToDevice, c->stream));
            SCHK(h2d_pinned(c, mrowd, hmrow, ((size_t)n + 4 * (size_t)nm) * sizeof(int)));
            if (maxrot > 0)
                hipLaunchKernelGGL(rot_rows_batched_kernel, dim3((maxN + 63) / 64, nm), dim3(64), 0, c->stream, mdd, cur, ld,
                                   i1d, i2d, csd);
            if (maxK > 0) {
                hipLaunchKernelGGL(secular_batched_kernel, dim3((maxK + 3) / 4, nm), dim3(256), 0, c->stream, mdd, Dd, wd,
                                   taud, orgd, lamd, info);
                hipLaunchKernelGGL(zhat_batched_kernel, dim3((maxK + 3) / 4, nm), dim3(256), 0, c->stream, mdd, Dd, wd, taud,
                                   orgd, zhd);
                hipLaunchKernelGGL(build_u_batched_kernel, dim3(maxK, nm), dim3(256), 0, c->stream, mdd, Dd, zhd, taud, orgd,
                                   W.Ut, ld);
            }
            hipLaunchKernelGGL(gather_level_kernel, dim3((maxN + 255) / 256, n), dim3(256), 0, c->stream, mdd, mrowd, idxd, cur,
                               W.Zc, nxt, ld);
            HIPCHK(hipGetLastError());
            SCHK(launch_gemm_merge_batched(c, nm, gdescd, maxN, maxK, W.Ut, W.Zc, nxt, ld));
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(lam, lamd, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(hi2, info, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        std::swap(cur, nxt);
        if (dc_timing) {
            const auto tw3 = std::chrono::steady_clock::now();
            auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
            fprintf(stderr, "dc level %d: wait %.1f us, values + plan + staging %.1f us, uploads + launches %.1f us\n", h, us(tw0, tw1),
                    us(tw1, tw2), us(tw2, tw3));
        }
        if (pipelined) {
            plans_prev.swap(plans);
            if (h < maxdepth) SCHK(queue_level_vectors(h + 1, cur));
        } else {
            SCHK(stream_wait(c));
            SCHK(take_level_values(plans, defl_all[h & 1]));
        }
    }
    if (pipelined && maxdepth >= 1) {
        SCHK(stream_wait(c));
        SCHK(take_level_values(plans_prev, defl_all[maxdepth & 1]));
    }
    // final ascending order
    // (after the last merge: its new roots ascending, then its deflated values ascending — two runs, one linear merge;
    // the same order std::stable_sort gives, at a tenth of its time)
    order.resize(n);
    ascending_order(vals.data(), n, order.data(), order_all.data());
    for (int i = 0; i < n; ++i) wout[i] = vals[order[i]];
    std::copy(order.begin(), order.end(), hidx);
    HIPCHK(hipMemcpyAsync(idxd, hidx, (size_t)n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    SCHK(launch_gather_rows(c, cur, ld, idxd, n, n, nxt, ld));
    SCHK(stream_wait(c));
    if (nxt != W.Za) std::swap(W.Za, W.Zb);
    return SELLA_OK;
}

// Blocked tridiagonalisation of W.A (destroyed; reflector tails are left in its rows).
static int tridiagonalise(EighWork& W, double* taus, double* dvec, double* evec) {
    sella_ctx* c = W.c;
    const int n = W.n, ld = W.ld;
    int nb = (int)c->opt.eigh_nb;
    if (nb > TRD_NBMAX) nb = TRD_NBMAX;
    const int nrefl = n - 2;
    double *Vp, *Wp, *part;
    SCHK(scratch_get(c, SCR_MISC0, (size_t)2 * TRD_NBMAX * ld * sizeof(double), &Vp));
    Wp = Vp + (size_t)nb * ld;
    // Cleared ONCE per factorisation (a memset is ~5 us whatever its size: per panel it was 1 ms of every eigh).  Within
    // a panel every entry of V_p, W_p that meets a non-zero factor has been written by this panel (indices >= j0 + p + 1);
    // the entries in front of it — left over from the previous panel — only ever meet the zeros of the masked row u',
    // so they must be finite, nothing more: the shared scratch may hold anything before the first panel.
    HIPCHK(s_memset0(c, Vp, (size_t)2 * nb * ld * sizeof(double)));
    const int maxblkA = (n + 255) / 256 + 1, maxblkB = (n + 16 + 2 * TRD_NBMAX + 1) / 2 + 1;
    SCHK(scratch_get(c, SCR_MISC1, ((size_t)2 * maxblkA * TRD_PA + 2 * (size_t)maxblkB + 4 * TRD_NBMAX + 128) * sizeof(double), &part));
    double* partA[2] = {part, part + (size_t)maxblkA * TRD_PA};
    double* partB = part + 2 * (size_t)maxblkA * TRD_PA;
    double* ub[2] = {W.vec + (size_t)V_U0 * ld, W.vec + (size_t)V_U1 * ld};
    double* wraw = W.vec + (size_t)V_WRAW * ld;
    double* colscal = W.vec + (size_t)V_COL * ld;
    double* cdots = partB + maxblkB + 8;               // 2 * TRD_NBMAX doubles
    // one-launch-per-column chain (trd_upd_kernel): second set of everything a launch both reads (previous column) and writes
    double* partB2[2] = {partB, cdots + 2 * TRD_NBMAX + 8};
    double* colscal2[2] = {colscal, partB2[1] + maxblkB + 8};
    double* wraw2[2] = {wraw, W.vec + (size_t)V_WRAW2 * ld};
    const int upd_max = (int)c->opt.eigh_upd_max;
    int cur = 0, nblkA_prev = 0, nblkB_prev = 0;
    // symmetric-aware matvec for trailing blocks of at least `eigh_symv_min` rows (0: never)
    const int symv_min = (int)c->opt.eigh_symv_min;
    const int symTR = c->opt.eigh_symv_tr == 128 ? 128 : 64;
    const int symNJ = (n + SYMV_TC - 1) / SYMV_TC, symNI = (n + symTR - 1) / symTR;
    const int ldP = (n + 255) / 256 * 256;
    double *Prow = nullptr, *Pcol = nullptr;
    if (symv_min > 0 && n - 1 >= symv_min) {
        SCHK(scratch_get(c, SCR_SYMV, (size_t)(symNJ + symNI) * ldP * sizeof(double), &Prow));
        Pcol = Prow + (size_t)symNJ * ldP;
    }
    const int tail_lds = (int)std::min<long>(c->opt.eigh_tail_lds, TRD_TAIL);
    bool lower_stale = false;                          // the trailing update has been writing the upper triangle only
    const bool can_tri = c->opt.rank2k_stream && c->opt.eigh_symv_tri;
    for (int j0 = 0; j0 < nrefl; j0 += nb) {
        const int kb = std::min(nb, nrefl - j0);
        if (upd_max > 0 && !lower_stale && n - j0 - 1 <= upd_max && !(tail_lds > 0 && n - j0 <= tail_lds) &&
            !(symv_min > 0 && n - j0 - 1 >= symv_min)) {
            // ---- small trailing block: one launch per column from here to the LDS tail (or to the end)
            const int jend = tail_lds > 0 ? std::min(nrefl, n - tail_lds) : nrefl;
            int fc = 0, nblk_last = 0;
            for (int j = j0; j < jend; ++j) {
                const int o = j + 1, m = n - o;
                TrdUpdArgs ua;
                ua.A = W.A; ua.ld = ld; ua.n = n; ua.j = j;
                ua.o = o; ua.m = m; ua.oc = o & ~1; ua.shift = o - ua.oc;
                ua.v_prev = ub[1 - fc]; ua.v_cur = ub[fc];
                ua.wraw_prev = wraw2[1 - fc]; ua.wraw_cur = wraw2[fc];
                ua.partB_prev = partB2[1 - fc]; ua.nblkB_prev = nblk_last; ua.partB_cur = partB2[fc];
                ua.colscal_prev = colscal2[1 - fc]; ua.colscal_cur = colscal2[fc];
                ua.taus = taus; ua.evec = evec; ua.dvec = dvec;
                int R = (int)c->opt.eigh_upd_rows;
                if (R != 2 && R != 4 && R != 8) R = m >= c->opt.eigh_upd_r8_min ? 8 : m >= c->opt.eigh_upd_r4_min ? 4 : 2;
                while (R < 8 && (8 * R + m + R - 1) / R > TRD_UPD_MAXGRID) R *= 2;
                ua.pad = o % (8 * R);
                const int grid = (ua.pad + m + R - 1) / R;
                if (grid > TRD_UPD_MAXGRID) { set_error("eigh_upd_max too large for the one-launch-per-column chain"); return SELLA_E_INVALID; }
                const int n2 = (m + ua.shift + 1) >> 1;     // 16-byte chunks per row: one per thread up to 1024 columns
                const int nt_max = (int)c->opt.eigh_upd_nt;
                const bool first = j == j0;
                const bool prof_all = c->prof;
                if (prof_all && (j & 3)) c->prof = false;
                if (c->prof) SCHK(stream_wait(c));
                prof_begin(c, PROF_OTHER, (first ? 8.0 : 16.0) * m * (double)m, (first ? 2.0 : 6.0) * m * (double)m);   // (PROF_TRD_GEMV stays the matvec of the blocked chain alone)
#define SELLA_TRD_UPD(RR, NT)                                                                                             \
    do {                                                                                                                  \
        if (first) SELLA_LAUNCH(c, HIP_KERNEL_NAME(trd_upd_kernel<RR, NT, true>), dim3(grid), dim3(NT), 0, ua);           \
        else SELLA_LAUNCH(c, HIP_KERNEL_NAME(trd_upd_kernel<RR, NT, false>), dim3(grid), dim3(NT), 0, ua);                \
    } while (0)
#define SELLA_TRD_UPD_NT(RR)                                                                                              \
    do {                                                                                                                  \
        if (nt_max <= 128 || n2 <= 128) SELLA_TRD_UPD(RR, 128);                                                           \
        else if (nt_max <= 256 || n2 <= 256) SELLA_TRD_UPD(RR, 256);                                                      \
        else SELLA_TRD_UPD(RR, 512);                                                                                      \
    } while (0)
                switch (R) {
                    case 2: SELLA_TRD_UPD_NT(2); break;
                    case 4: SELLA_TRD_UPD_NT(4); break;
                    default: SELLA_TRD_UPD_NT(8); break;
                }
#undef SELLA_TRD_UPD_NT
#undef SELLA_TRD_UPD
                prof_end(c);
                c->prof = prof_all;
                nblk_last = grid;
                fc = 1 - fc;
            }
            if (jend > j0) {
                TrdUpdFinishArgs fa;
                fa.A = W.A; fa.ld = ld; fa.n = n; fa.o = jend;
                fa.v = ub[1 - fc]; fa.wraw = wraw2[1 - fc];
                fa.partB = partB2[1 - fc]; fa.nblkB = nblk_last;
                fa.colscal = colscal2[1 - fc];
                SELLA_LAUNCH(c, trd_upd_finish_kernel, dim3((n - jend + 3) / 4), dim3(256), 0, fa);
            }
            HIPCHK(hipGetLastError());
            j0 = jend - nb;                                 // (the loop adds nb: the next pass starts at jend)
            continue;
        }
        if (tail_lds > 0 && n - j0 <= tail_lds) {
            // the rest of the factorisation inside one workgroup (the trailing block is up to date at a panel boundary)
            prof_begin(c, PROF_OTHER, 8.0 * (n - j0) * (double)(n - j0), 0.0);
            SELLA_LAUNCH(c, trd_tail_lds_kernel, dim3(1), dim3(1024), 0, W.A, ld, n, j0, dvec, evec, taus);
            prof_end(c);
            HIPCHK(hipGetLastError());
            return SELLA_OK;
        }
        // one decision per panel: the symmetric-aware matvec reads the upper triangle, the streaming one the full block
        const bool panel_symv = symv_min > 0 && n - j0 - 1 >= symv_min;
        for (int i = 0; i <= kb; ++i) {
            const int j = j0 + i;
            const bool do_row = i < kb;
            TrdRowArgs ra;
            ra.A = W.A; ra.ld = ld; ra.n = n; ra.j = j; ra.i = i; ra.do_row = do_row ? 1 : 0;
            ra.Vp = Vp; ra.Wp = Wp; ra.ldp = ld;
            ra.u_prev = ub[1 - cur]; ra.u_cur = ub[cur];
            ra.wraw = wraw;
            ra.partA_prev = partA[1 - cur]; ra.nblkA_prev = nblkA_prev;
            ra.partA_cur = partA[cur];
            ra.partB = partB; ra.nblkB = nblkB_prev;
            ra.colscal = colscal;
            ra.cdots = cdots;
            ra.dvec = dvec;
            const int nblkA = (n - j + 255) / 256;
            const dim3 gA(nblkA), bA(256);
            // Profiling samples every 4th column.  The event pair is attached to the dispatch packet, whose
            // start stamp is taken when the packet is picked up — behind a backlog of earlier launches that
            // would include queueing time.  So the queue is drained first and then BOTH kernels of the column
            // are launched with events: the matvec starts right behind its row kernel exactly as in the
            // untraced run, and nothing else is ahead of it.
            const bool prof_all = c->prof;
            if (prof_all && (j & 3)) c->prof = false;
            if (c->prof) SCHK(stream_wait(c));
            prof_begin(c, PROF_OTHER, 8.0 * (2.0 * i + 3.0) * (n - j), 0.0);
            switch (i - 1) {
#define SELLA_TRD_ROW_CASE(IP) case IP: SELLA_LAUNCH(c, HIP_KERNEL_NAME(trd_row_kernel<IP>), gA, bA, 0, ra); break;
                case -1:
                SELLA_TRD_ROW_CASE(0) SELLA_TRD_ROW_CASE(1) SELLA_TRD_ROW_CASE(2) SELLA_TRD_ROW_CASE(3)
                SELLA_TRD_ROW_CASE(4) SELLA_TRD_ROW_CASE(5) SELLA_TRD_ROW_CASE(6) SELLA_TRD_ROW_CASE(7)
                SELLA_TRD_ROW_CASE(8) SELLA_TRD_ROW_CASE(9) SELLA_TRD_ROW_CASE(10) SELLA_TRD_ROW_CASE(11)
                SELLA_TRD_ROW_CASE(12) SELLA_TRD_ROW_CASE(13) SELLA_TRD_ROW_CASE(14) SELLA_TRD_ROW_CASE(15)
                SELLA_TRD_ROW_CASE(16) SELLA_TRD_ROW_CASE(17) SELLA_TRD_ROW_CASE(18) SELLA_TRD_ROW_CASE(19)
                SELLA_TRD_ROW_CASE(20) SELLA_TRD_ROW_CASE(21) SELLA_TRD_ROW_CASE(22) SELLA_TRD_ROW_CASE(23)
                SELLA_TRD_ROW_CASE(24) SELLA_TRD_ROW_CASE(25) SELLA_TRD_ROW_CASE(26) SELLA_TRD_ROW_CASE(27)
                SELLA_TRD_ROW_CASE(28) SELLA_TRD_ROW_CASE(29) SELLA_TRD_ROW_CASE(30) SELLA_TRD_ROW_CASE(31)
#undef SELLA_TRD_ROW_CASE
                default: SELLA_LAUNCH(c, HIP_KERNEL_NAME(trd_row_kernel<-1>), gA, bA, 0, ra);
            }
            prof_end(c);
            if (!do_row) { c->prof = prof_all; break; }
            const int o = j + 1, m = n - o, oc = o & ~1;
            TrdGemvArgs ga;
            ga.A22 = W.A + (size_t)o * ld + oc; ga.ld = ld; ga.m = m; ga.shift = o - oc; ga.o = o; ga.n = n; ga.j = j;
            ga.ubuf = ub[cur];
            ga.partA = partA[cur]; ga.nblkA = nblkA;
            ga.wraw = wraw; ga.partB = partB;
            ga.Vrow = Vp + (size_t)i * ld;
            ga.Arow = W.A + (size_t)j * ld;
            ga.taus = taus; ga.evec = evec; ga.colscal = colscal;
            ga.Wp = Wp; ga.Vp = Vp; ga.ldp = ld; ga.i = i; ga.cdots = cdots;
            // Row pair P = (absolute row)/2 is always handled by workgroup P - P0 with P0 a multiple of 8, so a
            // given row stays on the same XCD (workgroup id mod 8) from one column to the next and is served
            // from that XCD's L2 once the trailing block fits (m <~ 1800); the <= 15 rows between 2 P0 and o
            // are dummies.
            ga.pad = o - (o / 16) * 16;
            int nblkB = (ga.pad + m + 2 * i + 1) / 2;
            if (panel_symv) {
                // large trailing block: upper triangle only, partial sums added up by a second (small) kernel
                TrdSymvArgs sa;
                sa.A = W.A; sa.ld = ld; sa.n = n; sa.o = o; sa.ubuf = ub[cur];
                sa.Prow = Prow; sa.Pcol = Pcol; sa.ldp = ldP;
                prof_begin(c, PROF_TRD_GEMV, 4.0 * m * (double)m, 2.0 * m * (double)m);
                if (symTR == 128) SELLA_LAUNCH(c, trd_symv_kernel<128>, dim3(symNJ, symNI), dim3(256), 0, sa);
                else SELLA_LAUNCH(c, trd_symv_kernel<64>, dim3(symNJ, symNI), dim3(256), 0, sa);
                prof_end(c);
                TrdSymvFinishArgs fa;
                fa.A = W.A; fa.ld = ld; fa.n = n; fa.o = o; fa.j = j; fa.ubuf = ub[cur];
                fa.partA = partA[cur]; fa.nblkA = nblkA;
                fa.Prow = Prow; fa.Pcol = Pcol; fa.ldp = ldP;
                fa.wraw = wraw; fa.partB = partB; fa.Vrow = ga.Vrow; fa.Arow = ga.Arow;
                fa.taus = taus; fa.evec = evec; fa.colscal = colscal;
                fa.Wp = Wp; fa.Vp = Vp; fa.ldpan = ld; fa.i = i; fa.cdots = cdots;
                fa.tr = symTR;
                fa.nelem = (n - (o & ~63) + 63) / 64;
                nblkB = fa.nelem;
                prof_begin(c, PROF_OTHER, 16.0 * m * (double)(m / symTR + m / SYMV_TC), 0.0);
                SELLA_LAUNCH(c, trd_symv_finish_kernel, dim3(fa.nelem + 2 * i), dim3(256), 0, fa);
                prof_end(c);
            } else {
                prof_begin(c, PROF_TRD_GEMV, 8.0 * m * (double)m, 2.0 * m * (double)m);
                const int n2g = (m + ga.shift + 1) >> 1;                // 16-byte pieces per row (as in the kernel)
                const int nch = c->opt.eigh_gemv_flat ? (n2g + 255) / 256 : 1 << 30;
                if (nch <= 2) SELLA_LAUNCH(c, trd_gemv_kernel<2>, dim3(nblkB), dim3(256), 0, ga);
                else if (nch <= 4) SELLA_LAUNCH(c, trd_gemv_kernel<4>, dim3(nblkB), dim3(256), 0, ga);
                else if (nch <= 6) SELLA_LAUNCH(c, trd_gemv_kernel<6>, dim3(nblkB), dim3(256), 0, ga);
                else if (nch <= 8) SELLA_LAUNCH(c, trd_gemv_kernel<8>, dim3(nblkB), dim3(256), 0, ga);
                else if (nch <= 10) SELLA_LAUNCH(c, trd_gemv_kernel<10>, dim3(nblkB), dim3(256), 0, ga);
                else SELLA_LAUNCH(c, trd_gemv_kernel<0>, dim3(nblkB), dim3(256), 0, ga);
                prof_end(c);
            }
            c->prof = prof_all;
            nblkA_prev = nblkA;
            nblkB_prev = nblkB;
            cur = 1 - cur;
        }
        HIPCHK(hipGetLastError());
        // trailing update A22 -= V^T W + W^T V over rows/columns >= j0 + kb
        const int r0 = j0 + kb, mt = n - r0;
        if (mt > 0) {
            double* At = W.A + (size_t)r0 * ld + r0;
            // one fused pass (update.hip): every tile pair is read and written once
            const bool next_symv = symv_min > 0 && n - r0 - 1 >= symv_min;
            // (the streaming kernel needs 16-byte aligned rows; the kernel that stands in otherwise averages the two
            // triangles, so it must see a full block)
            const bool aligned = !(ld & 1) && !(reinterpret_cast<uintptr_t>(At) & 15);
            if (lower_stale && !(can_tri && aligned)) {
                SCHK(launch_mirror_upper(c, At, mt, ld));
                lower_stale = false;
            }
            if (can_tri && aligned && (next_symv || lower_stale)) {
                SCHK(launch_rank2k_stream(c, At, mt, ld, Vp + r0, Wp + r0, ld, kb, -1.0, true));
                lower_stale = true;
                if (!next_symv) {                          // the streaming matvec takes over: full block again
                    SCHK(launch_mirror_upper(c, At, mt, ld));
                    lower_stale = false;
                }
            } else if (c->opt.rank2k_stream) SCHK(launch_rank2k_stream(c, At, mt, ld, Vp + r0, Wp + r0, ld, kb, -1.0));
            else SCHK(launch_sym_rank2k(c, At, mt, ld, Vp + r0, Wp + r0, ld, kb, -1.0));
        }
    }
    hipLaunchKernelGGL(tridiag_tail_kernel, dim3(1), dim3(64), 0, c->stream, W.A, ld, n, dvec, evec, taus);
    HIPCHK(hipGetLastError());
    return SELLA_OK;
}

// ---------------------------------------------------------------------------------------
// Low-rank update of a full eigendecomposition.
//
// Given B = V diag(w) V^T (rows of Vt are the eigenvectors) and the symmetric modification
//     B+ = B + sum_a (U_a Z_a^T + Z_a U_a^T)        (what sella_update_h applies to B, update.hip),
// the eigendecomposition of B+ is obtained without touching B: the modification is diagonalised in
// the span of its 2kk vectors (Gram-Schmidt on the device, a 2kk x 2kk eigenproblem on the host),
//     B+ = B + sum_r sigma_r q_r q_r^T,
// and every term is one rank-one modification of a diagonal matrix in the current eigenbasis,
//     V^T B+ V = diag(w) + sigma z z^T,  z = V^T q,
// i.e. exactly the merge step of the divide-and-conquer solver above (deflation on the host, secular
// equation, Gu/Eisenstat vectors, ONE K x K x n GEMM on the matrix cores).  A quasi-Newton step thus
// costs O(n^2 K) MFMA flops per rank instead of a fresh latency-bound tridiagonalisation.
// ---------------------------------------------------------------------------------------
// Vt holds nr eigenvectors of length n as rows (nr == n: a full eigendecomposition; nr < n: the explicit part of a
// structured one, see lr_lowrank_update below); w (host, nr) ascending on entry and on exit.
//
// append_lam0 != nullptr (structured decompositions): the component of q outside the span of the nr rows — an
// eigenvector for *append_lam0 — is orthonormalised into row nr SPECULATIVELY (two Gram-Schmidt sweeps, no host round
// trip of their own), z is taken over nr + 1 rows, and the accept / drop rules of gs_orthonormalise are applied to the
// sweep norms that come back with z; *nr_out = rows held on exit (nr or nr + 1).  The row re-ordering at the end is left
// queued (the host-side eigenvalues are final before it).
static int eig_rank1_update(sella_ctx* c, EighWork& W, int nr, int n, int ld, double* w, double* Vt, const double* q,
                            double sigma, const double* append_lam0 = nullptr, int* nr_out = nullptr) {
    double* zdev = W.vec + (size_t)V_Z * ld;
    double* csd = W.vec + (size_t)V_CS0 * ld;
    double* Dd = W.vec + (size_t)V_DD * ld;
    double* wd = W.vec + (size_t)V_WD * ld;
    double* taud = W.vec + (size_t)V_TAU * ld;
    double* zhd = W.vec + (size_t)V_ZH * ld;
    double* lamd = W.vec + (size_t)V_LAM * ld;
    int* info = W.ibuf;
    int* i1d = W.ibuf + 16;
    int* i2d = i1d + n;
    int* idxd = i1d + 2 * n;
    int* orgd = i1d + 3 * n;                 // (strides of n >= nr entries)
    static const bool dbg_time = getenv("SELLA_DEBUG_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tt0 = now();
    const int nr_in = nr;
    if (append_lam0) {
        double* slot = Vt + (size_t)nr * ld;
        HIPCHK(s_memcpy(c, slot, q, (size_t)ld * sizeof(double), hipMemcpyDeviceToDevice));
        SCHK(gs_project_twice(c, Vt, ld, nr, slot, n));          // sweep norms -> scalar slots 8, 9, 10
        ++nr;                                                     // z includes the speculative row
    }
    // z = Vt q
    SCHK(launch_gemv_rows(c, Vt, nr, n, ld, q, ld, 1, zdev, ld, GemvEpi()));
    // small transfers through the pinned staging buffer, laid out like the device side (see dc_solve)
    void* stage;
    SCHK(host_stage(c, (4 * (size_t)ld + 2 * (size_t)n + 8) * sizeof(double) + (3 * (size_t)n + 16) * sizeof(int), &stage));
    double* hcs = static_cast<double*>(stage);                 // mirrors csd (2 slots) | Dd | wd
    double* hD = hcs + 2 * (size_t)ld;
    double* hw = hcs + 3 * (size_t)ld;
    double* z = hcs + 4 * (size_t)ld;
    double* lam = z + n;
    int* hinfo = reinterpret_cast<int*>(lam + n + 8);
    int* hr1 = hinfo + 8;                                      // mirrors i1d | i2d | idxd
    int* hr2 = hr1 + n;
    int* hidx = hr1 + 2 * (size_t)n;
    HIPCHK(hipMemcpyAsync(z, zdev, (size_t)nr * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(s_memset0(c, info, 8 * sizeof(int)));
    if (append_lam0) SCHK(sync_scalars(c, 8, 3));
    else SCHK(stream_wait(c));
    if (append_lam0) {
        // gs_orthonormalise's verdict on the speculative row (math.pyx:112-129 thresholds as used there: dropped when
        // a sweep leaves less than 1e-13 of it, accepted when the second sweep's norm is one to 1e-15)
        const double n0sq = c->hscal[8], n1 = sqrt(c->hscal[9]), n2 = sqrt(c->hscal[10]);
        bool kept = (n0sq > 0.0) && (n1 == n1) && !(n1 < 1e-13) && (n2 == n2) && !(n2 < 1e-13);
        if (kept && fabs(1.0 - n2) > 1e-15) {
            // rare: a third sweep is needed — finish the row synchronously and take its z entry again
            int k2 = 0;
            double* slot = Vt + (size_t)nr_in * ld;
            SCHK(gs_orthonormalise(c, Vt, ld, nr_in, slot, n, 1e-15, 1e-13, 100, &k2, nullptr));
            kept = k2 != 0;
            if (kept) {
                SCHK(launch_gemv_rows(c, slot, 1, n, ld, q, ld, 1, zdev + nr_in, ld, GemvEpi()));
                HIPCHK(hipMemcpyAsync(z + nr_in, zdev + nr_in, sizeof(double), hipMemcpyDeviceToHost, c->stream));
                SCHK(stream_wait(c));
            }
        }
        if (kept) w[nr_in] = *append_lam0;
        else nr = nr_in;
        if (nr_out) *nr_out = nr;
        if (nr == 0) return SELLA_OK;
    }
    const double tt1 = now();
    // a negative weight is handled on the negated, reversed spectrum: primed index i' <-> row n-1-i'
    const bool neg = sigma < 0.0;
    auto rowof = [&](int ip) { return neg ? nr - 1 - ip : ip; };
    std::vector<double> D(nr), zz(nr);
    double znorm2 = 0.0;
    for (int ip = 0; ip < nr; ++ip) {
        D[ip] = neg ? -w[rowof(ip)] : w[ip];
        zz[ip] = z[rowof(ip)];
        znorm2 += zz[ip] * zz[ip];
    }
    if (!(znorm2 > 0.0)) return SELLA_OK;
    const double zn = sqrt(znorm2);
    for (int ip = 0; ip < nr; ++ip) zz[ip] /= zn;
    MergePlan pl;
    pl.lo = 0;
    pl.N = nr;
    pl.rho = fabs(sigma) * znorm2;
    // ---- clusters of (numerically) equal eigenvalues: one Householder reflection per cluster --------
    // An approximate Hessian starts as lam0*I + low rank, so most of its spectrum is one value repeated
    // ~n times.  The pairwise Givens deflation below would walk through such a cluster with a chain of
    // ~n dependent row rotations; a single reflection H = I - tau v v^T with H z_cluster = beta e_last
    // does the same job (all weight moved onto one row, the others deflate with z = 0) as one
    // transposed matvec and one rank-one update over the cluster's rows, which are contiguous because
    // the rows are kept in eigenvalue order.
    {
        const double eps = 2.220446049250313e-16;
        double dmax = 0.0;
        for (int ip = 0; ip < nr; ++ip) dmax = std::max(dmax, fabs(D[ip]));
        const double spread = 4.0 * eps * dmax;
        double* vdev = W.vec + (size_t)V_U1 * ld;
        double* wvd = W.vec + (size_t)V_WRAW * ld;
        std::vector<double> vh;
        int ip = 0;
        while (ip < nr) {
            int j = ip;
            while (j + 1 < nr && fabs(D[j + 1] - D[ip]) <= spread) ++j;       // (fabs: a freshly appended row of a
                                                                              // structured update sits out of order)
            const int len = j - ip + 1;
            if (len >= 8) {
                double nrm2 = 0.0;
                for (int t = ip; t <= j; ++t) nrm2 += zz[t] * zz[t];
                const double xl = zz[j];
                const double beta = (xl >= 0.0) ? -sqrt(nrm2) : sqrt(nrm2);
                // v = x - beta e_last ; v^T v = 2 (nrm2 - beta x_last)
                const double vtv = 2.0 * (nrm2 - beta * xl);
                if (nrm2 > 0.0 && vtv > 0.0 && nrm2 > zz[j] * zz[j]) {
                    vh.assign(len, 0.0);
                    const int rlo = std::min(rowof(ip), rowof(j));
                    for (int t = ip; t <= j; ++t) vh[rowof(t) - rlo] = zz[t] - ((t == j) ? beta : 0.0);
                    const double tau = 2.0 / vtv;
                    SCHK(h2d_async(c, vdev, vh.data(), (size_t)len * sizeof(double)));
                    double* R = Vt + (size_t)rlo * ld;
                    SCHK(launch_gemv_cols(c, R, len, n, ld, vdev, ld, 1, wvd, ld));
                    hipLaunchKernelGGL(rows_ger_kernel, dim3((n + 255) / 256, len), dim3(256), 0, c->stream, R, ld, len, n,
                                       vdev, wvd, tau);
                    HIPCHK(hipGetLastError());
                    for (int t = ip; t < j; ++t) zz[t] = 0.0;
                    zz[j] = beta;
                }
            }
            ip = j + 1;
        }
    }
    plan_deflation(nr, D.data(), zz.data(), pl);
    const int K = pl.K;
    for (int p = 0; p < K; ++p) {
        hD[p] = D[pl.nondef[p]];
        hw[p] = zz[pl.nondef[p]];
        hidx[p] = rowof(pl.nondef[p]);
    }
    for (int p = 0; p < nr - K; ++p) hidx[K + p] = rowof(pl.defl[p]);
    for (int r = 0; r < pl.nrot; ++r) {
        hr1[r] = rowof(pl.r1[r]);
        hr2[r] = rowof(pl.r2[r]);
        hcs[2 * (size_t)r] = pl.cs[2 * r];
        hcs[2 * (size_t)r + 1] = pl.cs[2 * r + 1];
    }
    const double tt2 = now();
    static_assert(V_CS1 == V_CS0 + 1 && V_DD == V_CS0 + 2 && V_WD == V_CS0 + 3, "slot order");
    SCHK(h2d_pinned(c, i1d, hr1, 3 * (size_t)n * sizeof(int)));
    SCHK(h2d_pinned(c, csd, hcs, (3 * (size_t)ld + n) * sizeof(double)));
    double* nxt = W.Zb;
    if (pl.nrot > 0) {
        hipLaunchKernelGGL(rot_rows_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, Vt, ld, n, pl.nrot, i1d, i2d, csd);
    }
    if (K > 0) {
        hipLaunchKernelGGL(secular_kernel, dim3((K + 3) / 4), dim3(256), 0, c->stream, K, Dd, wd, pl.rho, taud, orgd, lamd, info);
        hipLaunchKernelGGL(zhat_kernel, dim3((K + 3) / 4), dim3(256), 0, c->stream, K, Dd, wd, taud, orgd, zhd);
        hipLaunchKernelGGL(build_u_kernel, dim3(K), dim3(256), 0, c->stream, K, Dd, zhd, taud, orgd, W.Ut, ld);
        HIPCHK(hipGetLastError());
        SCHK(launch_gather_rows(c, Vt, ld, idxd, K, n, W.Zc, ld));
        SCHK(launch_gemm(c, 0, 0, K, n, K, 1.0, W.Ut, ld, W.Zc, ld, 0.0, nxt, ld));
        HIPCHK(hipMemcpyAsync(lam, lamd, (size_t)K * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    if (nr - K > 0) SCHK(launch_gather_rows(c, Vt, ld, idxd + K, nr - K, n, nxt + (size_t)K * ld, ld));
    HIPCHK(hipMemcpyAsync(hinfo, info, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    SCHK(stream_wait(c));
    if (hinfo[1] != 0) {
        set_error("eigh update: secular equation solver hit its iteration cap (root %d)", hinfo[1] - 1);
        return SELLA_E_NOCONV;
    }
    const double tt3 = now();
    // new spectrum (rows of nxt: K updated vectors, then the deflated ones), back to ascending order
    // (ordered in the primed spectrum, where both pieces ascend; a negative weight reverses the result)
    std::vector<double> nv(nr);
    for (int p = 0; p < K; ++p) nv[p] = lam[p];
    for (int p = 0; p < nr - K; ++p) nv[K + p] = D[pl.defl[p]];
    std::vector<int> order(nr), order_tmp(nr);
    ascending_order(nv.data(), nr, order.data(), order_tmp.data());
    if (neg) std::reverse(order.begin(), order.end());
    for (int i = 0; i < nr; ++i) w[i] = neg ? -nv[order[i]] : nv[order[i]];
    std::copy(order.begin(), order.end(), hidx);
    HIPCHK(hipMemcpyAsync(idxd, hidx, (size_t)n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    const double tt4 = now();
    SCHK(launch_gather_rows(c, nxt, ld, idxd, nr, n, Vt, ld));
    if (!append_lam0) SCHK(stream_wait(c));
    if (dbg_time)
        fprintf(stderr, "rank-one eigen-update n=%d rows=%d K=%d rot=%d: z %.0f us, plan %.0f us, device %.0f us, order %.0f us, final gather %.0f us\n",
                n, nr, K, pl.nrot, 1e6 * (tt1 - tt0), 1e6 * (tt2 - tt1), 1e6 * (tt3 - tt2), 1e6 * (tt4 - tt3), 1e6 * (now() - tt4));
    return SELLA_OK;
}

// One secant pair (kk = 1, the per-step quasi-Newton update): Delta = u z^T + z u^T has rank two and its two terms
// sigma_t q_t q_t^T follow in closed form from the three dots u.u, u.z, z.z — ONE round trip instead of two
// Gram-Schmidt passes and a coordinate read-back.  With e1 = u / |u|, e2 the normalised part of z orthogonal to u,
// Delta = [e1 e2] [[2b, s], [s, 0]] [e1 e2]^T, b = u.z, s = |u| |z_perp|: sigma = b +- sqrt(b^2 + s^2), eigenvectors
// (sigma, s).  src: rows u, z; Q: two output rows.  *ok = false when z is too nearly parallel to u for the Gram form
// to keep 1e-13 (|z_perp|^2 < 1e-3 |z|^2): the caller takes the Gram-Schmidt path.
static int pair_terms(sella_ctx* c, const double* src, int n, int ld, double* Q, double sig[2], int* nterms, bool* ok) {
    *nterms = 0;
    *ok = true;
    SCHK(launch_gemv_rows(c, src, 2, n, ld, src, ld, 2, c->dscal + DS_CVEC, 2, GemvEpi()));
    SCHK(read_scalars(c, DS_CVEC, 4));
    const double a = c->hscal[DS_CVEC], b = c->hscal[DS_CVEC + 1], cc = c->hscal[DS_CVEC + 3];
    if (!(a > 0.0) || !(cc > 0.0)) return SELLA_OK;                 // one factor vanishes: Delta = 0
    const double rr = cc - b * b / a;
    if (!(rr >= 1e-3 * cc)) { *ok = false; return SELLA_OK; }
    const double s = sqrt(a * rr), root = sqrt(b * b + s * s);
    const double lam[2] = {b + root, b - root};
    // q = (lam e1 + s e2) / norm = alpha u + beta z
    double* st = c->hscal + DS_STAGE;
    for (int t = 0; t < 2; ++t) {
        const double nrm = sqrt(lam[t] * lam[t] + s * s);
        const double beta = s / (nrm * sqrt(rr));
        const double alpha = lam[t] / (nrm * sqrt(a)) - beta * b / a;
        st[0 * 2 + t] = alpha;                                        // W1[j * nout + t]: coefficient of src row j
        st[1 * 2 + t] = beta;
        sig[t] = lam[t];
    }
    double* coef = c->dscal + DS_STAGE;
    HIPCHK(hipMemcpyAsync(coef, st, 4 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    SCHK(launch_lincomb(c, n, 2, src, ld, 2, coef, 2, nullptr, 0, 0, nullptr, 0, 0.0, Q, ld));
    *nterms = 2;
    return SELLA_OK;
}

int eig_lowrank_update(sella_ctx* c, int n, double* w, Mat* V, Mat* Vt, const double* Up, const double* Zp,
                       int ldp, int kk, int* nrank1) {
    const int ld = Vt->ld;
    const int m = 2 * kk;
    if (nrank1) *nrank1 = 0;
    if (ldp != ld) { set_error("eigh update: panel stride mismatch"); return SELLA_E_INVALID; }
    EighWork W;
    W.c = c; W.n = n; W.ld = ld;
    const size_t mbytes = ((size_t)std::max(n, 64) + WY_NB2 + 2) * std::max(ld, 64) * sizeof(double);
    SCHK(scratch_get(c, SCR_EIG1, mbytes, &W.Za));
    SCHK(scratch_get(c, SCR_EIG2, mbytes, &W.Zb));
    SCHK(scratch_get(c, SCR_EIG3, mbytes, &W.Zc));
    SCHK(scratch_get(c, SCR_EIG4, mbytes, &W.Ut));
    SCHK(scratch_get(c, SCR_EIG5, (size_t)V_NSLOTS * ld * sizeof(double) + (size_t)(8 * n + 512) * sizeof(int), &W.vec));
    W.ibuf = reinterpret_cast<int*>(W.vec + (size_t)V_NSLOTS * ld);
    W.A = nullptr;
    // ---- orthonormal basis of span{U_a, Z_a} (rows of Qb) and the coordinates of U, Z in it ------
    double* Qb = W.Za;                       // up to m rows
    double* src = Qb + (size_t)m * ld;       // [U; Z] copied next to it
    SCHK(launch_axpby2d(c, kk, n, 1.0, Up, ldp, 0.0, nullptr, 0, src, ld));
    SCHK(launch_axpby2d(c, kk, n, 1.0, Zp, ldp, 0.0, nullptr, 0, src + (size_t)kk * ld, ld));
    if (kk == 1) {
        double sg[2];
        int nt = 0;
        bool ok = true;
        SCHK(pair_terms(c, src, n, ld, Qb, sg, &nt, &ok));
        if (ok) {
            double wm = 0.0;
            for (int i = 0; i < n; ++i) wm = std::max(wm, fabs(w[i]));
            const double dropk = 4.0 * 2.220446049250313e-16 * std::max(wm, std::max(fabs(sg[0]), fabs(sg[1])));
            const int first = fabs(sg[0]) >= fabs(sg[1]) ? 0 : 1;
            for (int t = 0; t < nt; ++t) {
                const int rr = t == 0 ? first : 1 - first;
                if (fabs(sg[rr]) <= dropk) continue;
                SCHK(eig_rank1_update(c, W, n, n, ld, w, Vt->d, Qb + (size_t)rr * ld, sg[rr]));
                if (nrank1) ++*nrank1;
            }
            if (V) SCHK(launch_transpose(c, Vt->d, n, n, Vt->ld, V->d, V->ld));
            SCHK(stream_wait(c));
            return SELLA_OK;
        }
    }
    int mb = 0;
    for (int v = 0; v < m; ++v) {
        double* slot = Qb + (size_t)mb * ld;
        HIPCHK(s_memcpy(c, slot, src + (size_t)v * ld, (size_t)ld * sizeof(double), hipMemcpyDeviceToDevice));
        int kept = 0;
        // a vector is dropped only when nothing but rounding noise is left of it
        SCHK(gs_orthonormalise(c, Qb, ld, mb, slot, n, 1e-15, 1e-13, 100, &kept, nullptr));
        if (kept) ++mb;
    }
    if (mb == 0) return SELLA_OK;
    // R[i][v] = Qb_i . src_v  (mb x m), column-chunks of at most 8 right-hand sides
    std::vector<double> R((size_t)mb * m);
    for (int v0 = 0; v0 < m; v0 += 8) {
        const int nv = std::min(8, m - v0);
        SCHK(launch_gemv_rows(c, Qb, mb, n, ld, src + (size_t)v0 * ld, ld, nv, c->dscal + DS_CVEC, mb, GemvEpi()));
        SCHK(read_scalars(c, DS_CVEC, mb * nv));
        for (int h = 0; h < nv; ++h)
            for (int i = 0; i < mb; ++i) R[(size_t)i * m + v0 + h] = c->hscal[DS_CVEC + (size_t)h * mb + i];
    }
    // core C = R_U R_Z^T + R_Z R_U^T
    std::vector<double> C((size_t)mb * mb, 0.0), sig(mb), F((size_t)mb * mb), work(mb);
    for (int i = 0; i < mb; ++i)
        for (int j = 0; j < mb; ++j) {
            double sum = 0.0;
            for (int a = 0; a < kk; ++a)
                sum += R[(size_t)i * m + a] * R[(size_t)j * m + kk + a] + R[(size_t)i * m + kk + a] * R[(size_t)j * m + a];
            C[(size_t)i * mb + j] = sum;
        }
    if (small::sym_eig(mb, C.data(), mb, sig.data(), F.data(), mb, work.data()) != 0) {
        set_error("eigh update: small eigenproblem did not converge");
        return SELLA_E_NOCONV;
    }
    double wmax = 0.0, smax = 0.0;
    for (int i = 0; i < n; ++i) wmax = std::max(wmax, fabs(w[i]));
    for (int r = 0; r < mb; ++r) smax = std::max(smax, fabs(sig[r]));
    const double drop = 4.0 * 2.220446049250313e-16 * std::max(wmax, smax);
    double* qv = W.vec + (size_t)V_U0 * ld;
    double* coef = c->dscal + DS_STAGE;
    // largest terms first
    std::vector<int> ord(mb);
    std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return fabs(sig[a]) > fabs(sig[b]); });
    for (int t = 0; t < mb; ++t) {
        const int r = ord[t];
        if (fabs(sig[r]) <= drop) continue;
        double* st = c->hscal + DS_STAGE + (size_t)t * 64;
        for (int i = 0; i < mb; ++i) st[i] = F[(size_t)i * mb + r];
        HIPCHK(hipMemcpyAsync(coef + (size_t)t * 64, st, (size_t)mb * sizeof(double), hipMemcpyHostToDevice, c->stream));
        SCHK(launch_lincomb(c, n, 1, Qb, ld, mb, coef + (size_t)t * 64, 1, nullptr, 0, 0, nullptr, 0, 0.0, qv, ld));
        SCHK(eig_rank1_update(c, W, n, n, ld, w, Vt->d, qv, sig[r]));
        if (nrank1) ++*nrank1;
    }
    if (V) SCHK(launch_transpose(c, Vt->d, n, n, Vt->ld, V->d, V->ld));
    SCHK(stream_wait(c));
    return SELLA_OK;
}


// ---------------------------------------------------------------------------------------
// Low-rank update of a STRUCTURED eigendecomposition.
//
// An approximate Hessian that started as lam0 * I (linalg.py:274-289: the first update of an uninitialised
// Hessian) is, after any number of quasi-Newton updates, lam0 * I plus a matrix of rank r << n:
//     B = lam0 (I - W^T W) + W^T diag(mu) W,      W (r x n) orthonormal rows,
// i.e. r explicit eigenpairs (mu_i, W_i) and the eigenvalue lam0 on the orthogonal complement of span(W),
// multiplicity n - r, with no basis stored for it.  The dense form above carries that cluster as ~n explicit rows
// and pays an n x n pass to reflect / reorder them at every rank-one modification.  Here a modification
// B+ = B + sigma q q^T touches span{W, q} only: the component of q outside span(W), normalised, joins W as a new row
// with eigenvalue lam0 (it IS an eigenvector of B for lam0), and the modification becomes the SAME rank-one merge
// as above on r + 1 rows of length n — O(n r) memory traffic instead of O(n^2).  Identical eigenpairs, to roundoff,
// as the dense update (tests/test_lr_eig.py).
//   Wt: rows [0, *r) valid, capacity Wt->rows; mu (host, capacity >= Wt->rows) ascending; both updated.
// ---------------------------------------------------------------------------------------
int lr_lowrank_update(sella_ctx* c, int n, int* r_io, double* mu, double lam0, Mat* Wt, const double* Up, const double* Zp,
                      int ldp, int kk, int* nrank1) {
    const int ld = Wt->ld;
    const int m = 2 * kk;
    if (nrank1) *nrank1 = 0;
    if (ldp != ld || Wt->cols != n) { set_error("structured eigen-update: panel stride mismatch"); return SELLA_E_INVALID; }
    if (m > 64) { set_error("structured eigen-update: at most 32 pairs per call (%d given)", kk); return SELLA_E_INVALID; }
    EighWork W;
    W.c = c; W.n = n; W.ld = ld;
    const int cap = Wt->rows;
    const size_t rows_max = (size_t)std::max(cap + m + 2, 64) + 2;
    const size_t mbytes = rows_max * std::max(ld, 64) * sizeof(double);
    SCHK(scratch_get(c, SCR_EIG1, mbytes, &W.Za));
    SCHK(scratch_get(c, SCR_EIG2, mbytes, &W.Zb));
    SCHK(scratch_get(c, SCR_EIG3, mbytes, &W.Zc));
    SCHK(scratch_get(c, SCR_EIG4, mbytes, &W.Ut));
    SCHK(scratch_get(c, SCR_EIG5, (size_t)V_NSLOTS * ld * sizeof(double) + (size_t)(8 * n + 512) * sizeof(int), &W.vec));
    W.ibuf = reinterpret_cast<int*>(W.vec + (size_t)V_NSLOTS * ld);
    W.A = nullptr;
    // ---- orthonormal basis of span{U_a, Z_a} and the 2kk x 2kk core, as in eig_lowrank_update ------------------
    double* Qb = W.Za;
    double* src = Qb + (size_t)m * ld;
    SCHK(launch_axpby2d(c, kk, n, 1.0, Up, ldp, 0.0, nullptr, 0, src, ld));
    SCHK(launch_axpby2d(c, kk, n, 1.0, Zp, ldp, 0.0, nullptr, 0, src + (size_t)kk * ld, ld));
    int r = *r_io;
    double* qv = W.vec + (size_t)V_U0 * ld;
    bool done = false;
    if (kk == 1) {
        double sg[2];
        int nt = 0;
        bool ok = true;
        SCHK(pair_terms(c, src, n, ld, Qb, sg, &nt, &ok));
        if (ok) {
            double wmax = fabs(lam0);
            for (int i = 0; i < r; ++i) wmax = std::max(wmax, fabs(mu[i]));
            const double drop = 4.0 * 2.220446049250313e-16 * std::max(wmax, std::max(fabs(sg[0]), fabs(sg[1])));
            const int first = fabs(sg[0]) >= fabs(sg[1]) ? 0 : 1;          // largest term first
            for (int t = 0; t < nt; ++t) {
                const int rr = t == 0 ? first : 1 - first;
                if (fabs(sg[rr]) <= drop) continue;
                if (r >= cap) { set_error("structured eigen-update: capacity of %d rows exhausted", cap); return SELLA_E_INVALID; }
                SCHK(eig_rank1_update(c, W, r, n, ld, mu, Wt->d, Qb + (size_t)rr * ld, sg[rr], &lam0, &r));
                if (nrank1) ++*nrank1;
            }
            done = true;
        }
    }
    if (!done) {
    int mb = 0;
    // Orthonormal basis of the 2 kk update vectors.  A block of them (the secant pairs of a Davidson run: 2 kk ~ 60)
    // by Cholesky-QR twice — Gram matrix on the matrix cores, its Cholesky factor on the host, the rows recombined, and
    // once more on the result: three round trips instead of two per vector.  The second pass sees a Gram matrix within
    // 0.1 of the identity or the block goes through the vector-by-vector Gram-Schmidt below (rank-deficient or wildly
    // scaled input: a non-positive pivot, a first pass that left more than that).
    bool blocked = false;
    if (m >= 8 && c->opt.lr_cholqr) {
        double* Gd = W.Ut;                                   // Gram matrices, then the recombination coefficients
        double* Q1 = W.Zb;
        std::vector<double> G((size_t)m * m), L, T, Wc;
        bool ok = true;
        // pass 1: PIVOTED Cholesky of the scaled Gram matrix — the secant pairs of a Krylov run span about half of
        // 2 kk dimensions (U and Z are combinations of S and Y = A S), so the factorisation has to reveal the rank: rows
        // enter in order of their remaining norm until that falls below 1e-6 of the row (1e-12 on the Gram scale, four
        // digits above its rounding noise); what the rejected rows still carry outside the basis is measured below
        SCHK(launch_gemm(c, 0, 1, m, m, n, 1.0, src, ld, src, ld, 0.0, Gd, m));
        SCHK(d2h_async(c, G.data(), Gd, (size_t)m * m * sizeof(double)));
        SCHK(stream_wait(c));
        std::vector<double> sc(m, 0.0), diag(m), rown2(m);
        std::vector<int> piv;
        for (int i = 0; i < m; ++i) {
            const double d = G[(size_t)i * m + i];
            rown2[i] = d;
            if (d == d && d > 0.0) sc[i] = 1.0 / sqrt(d);
            else if (!(d == 0.0)) ok = false;
            diag[i] = sc[i] > 0.0 ? 1.0 : 0.0;
        }
        std::vector<double> Lf((size_t)m * m, 0.0);            // Lf[i][q]: column q of the pivoted factor, all rows i
        if (ok) {
            std::vector<char> used(m, 0);
            for (int q = 0; q < m; ++q) {
                int best = -1;
                for (int i = 0; i < m; ++i)
                    if (!used[i] && (best < 0 || diag[i] > diag[best])) best = i;
                if (best < 0 || !(diag[best] > 1e-12)) break;
                const double lqq = sqrt(diag[best]);
                used[best] = 1;
                piv.push_back(best);
                for (int i = 0; i < m; ++i) {
                    if (used[i] && i != best) continue;
                    double t = 0.5 * (G[(size_t)i * m + best] + G[(size_t)best * m + i]) * sc[i] * sc[best];
                    for (int kq = 0; kq < q; ++kq) t -= Lf[(size_t)i * m + kq] * Lf[(size_t)best * m + kq];
                    Lf[(size_t)i * m + q] = (i == best) ? lqq : t / lqq;
                    if (i != best) diag[i] -= Lf[(size_t)i * m + q] * Lf[(size_t)i * m + q];
                }
            }
        }
        const int mp = (int)piv.size();
        ok = ok && mp > 0;
        if (ok) {
            // T = L11^-1 (mp x mp, L11 = rows piv of Lf): Q1_c = sum_q T[c][q] sc[piv q] src[piv q]
            T.assign((size_t)mp * mp, 0.0);
            for (int col = 0; col < mp; ++col)
                for (int i = col; i < mp; ++i) {
                    double acc = (i == col) ? 1.0 : 0.0;
                    for (int kq = col; kq < i; ++kq) acc -= Lf[(size_t)piv[i] * m + kq] * T[(size_t)kq * mp + col];
                    T[(size_t)i * mp + col] = acc / Lf[(size_t)piv[i] * m + i];
                }
            Wc.assign((size_t)m * mp, 0.0);                        // lincomb: Wc[j][c] = coefficient of src row j in output c
            for (int cq = 0; cq < mp; ++cq)
                for (int q = 0; q <= cq; ++q) Wc[(size_t)piv[q] * mp + cq] = T[(size_t)cq * mp + q] * sc[piv[q]];
            SCHK(h2d_async(c, Gd, Wc.data(), Wc.size() * sizeof(double)));
            SCHK(launch_lincomb(c, n, mp, src, ld, m, Gd, mp, nullptr, 0, 0, nullptr, 0, 0.0, Q1, ld));
            // pass 2: plain Cholesky-QR of the mp rows (now well conditioned): Gram matrix within 0.1 of the identity
            G.assign((size_t)mp * mp, 0.0);
            SCHK(launch_gemm(c, 0, 1, mp, mp, n, 1.0, Q1, ld, Q1, ld, 0.0, Gd, mp));
            SCHK(d2h_async(c, G.data(), Gd, (size_t)mp * mp * sizeof(double)));
            SCHK(stream_wait(c));
            double dev = 0.0;
            for (int i = 0; i < mp; ++i)
                for (int j = 0; j < mp; ++j) dev = std::max(dev, fabs(G[(size_t)i * mp + j] - (i == j ? 1.0 : 0.0)));
            L.assign((size_t)mp * mp, 0.0);
            std::vector<double> Gs((size_t)mp * mp);
            for (int i = 0; i < mp; ++i)
                for (int j = 0; j < mp; ++j) Gs[(size_t)i * mp + j] = 0.5 * (G[(size_t)i * mp + j] + G[(size_t)j * mp + i]);
            ok = dev < 0.1 && small::cholesky(mp, Gs.data(), mp, L.data(), mp) == 0;
        }
        if (ok) {
            T.assign((size_t)mp * mp, 0.0);
            for (int col = 0; col < mp; ++col)
                for (int i = col; i < mp; ++i) {
                    double acc = (i == col) ? 1.0 : 0.0;
                    for (int kq = col; kq < i; ++kq) acc -= L[(size_t)i * mp + kq] * T[(size_t)kq * mp + col];
                    T[(size_t)i * mp + col] = acc / L[(size_t)i * mp + i];
                }
            Wc.assign((size_t)mp * mp, 0.0);
            for (int cq = 0; cq < mp; ++cq)
                for (int q = 0; q <= cq; ++q) Wc[(size_t)q * mp + cq] = T[(size_t)cq * mp + q];
            SCHK(h2d_async(c, Gd, Wc.data(), Wc.size() * sizeof(double)));
            SCHK(launch_lincomb(c, n, mp, Q1, ld, mp, Gd, mp, nullptr, 0, 0, nullptr, 0, 0.0, Qb, ld));
            mb = mp;
            // what the update vectors carry outside the basis: rows above 1e-13 of their norm (the drop threshold of the
            // vector-by-vector path, math.pyx:112-117) are orthonormalised into it one by one
            double* Rd = W.Ut;
            double* Res = W.Zb;
            SCHK(launch_gemm(c, 0, 1, mb, m, n, -1.0, Qb, ld, src, ld, 0.0, Rd, m));             // -(Qb src^T), mb x m
            SCHK(launch_axpby2d(c, m, n, 1.0, src, ld, 0.0, nullptr, 0, Res, ld));
            SCHK(launch_lincomb(c, n, m, Qb, ld, mb, Rd, m, nullptr, 0, 0, nullptr, 0, 1.0, Res, ld));    // src - (src Qb^T) Qb
            double* nd = c->dscal + DS_CVEC;
            SCHK(launch_rows_sumsq(c, Res, ld, m, n, nd));
            SCHK(read_scalars(c, DS_CVEC, m));
            for (int h = 0; h < m; ++h) {
                if (!(c->hscal[DS_CVEC + h] > 1e-26 * rown2[h])) continue;
                double* slot = Qb + (size_t)mb * ld;
                HIPCHK(s_memcpy(c, slot, Res + (size_t)h * ld, (size_t)ld * sizeof(double), hipMemcpyDeviceToDevice));
                int kept = 0;
                SCHK(gs_orthonormalise(c, Qb, ld, mb, slot, n, 1e-15, 1e-13, 100, &kept, nullptr));
                if (kept) ++mb;
            }
            blocked = true;
        }
    }
    if (!blocked) {
    for (int v = 0; v < m; ++v) {
        double* slot = Qb + (size_t)mb * ld;
        HIPCHK(s_memcpy(c, slot, src + (size_t)v * ld, (size_t)ld * sizeof(double), hipMemcpyDeviceToDevice));
        int kept = 0;
        SCHK(gs_orthonormalise(c, Qb, ld, mb, slot, n, 1e-15, 1e-13, 100, &kept, nullptr));
        if (kept) ++mb;
    }
    }
    if (getenv("SELLA_DEBUG_TIMING"))
        fprintf(stderr, "structured eigen-update: %d update vectors, basis of %d by %s\n", m, mb,
                blocked ? "Cholesky-QR twice" : "vector-by-vector Gram-Schmidt");
    if (mb == 0) return SELLA_OK;
    // coordinates of the update vectors in the basis: R = Qb src^T in one product and one read-back
    std::vector<double> R((size_t)mb * m);
    {
        double* Rd = W.Ut;
        SCHK(launch_gemm(c, 0, 1, mb, m, n, 1.0, Qb, ld, src, ld, 0.0, Rd, m));
        SCHK(d2h_async(c, R.data(), Rd, (size_t)mb * m * sizeof(double)));
        SCHK(stream_wait(c));
    }
    std::vector<double> C((size_t)mb * mb, 0.0), sig(mb), F((size_t)mb * mb), work(mb);
    for (int i = 0; i < mb; ++i)
        for (int j = 0; j < mb; ++j) {
            double sum = 0.0;
            for (int a = 0; a < kk; ++a)
                sum += R[(size_t)i * m + a] * R[(size_t)j * m + kk + a] + R[(size_t)i * m + kk + a] * R[(size_t)j * m + a];
            C[(size_t)i * mb + j] = sum;
        }
    if (small::sym_eig(mb, C.data(), mb, sig.data(), F.data(), mb, work.data()) != 0) {
        set_error("structured eigen-update: small eigenproblem did not converge");
        return SELLA_E_NOCONV;
    }
    double wmax = fabs(lam0), smax = 0.0;
    for (int i = 0; i < r; ++i) wmax = std::max(wmax, fabs(mu[i]));
    for (int t = 0; t < mb; ++t) smax = std::max(smax, fabs(sig[t]));
    const double drop = 4.0 * 2.220446049250313e-16 * std::max(wmax, smax);
    if (r == 0) {
        // no explicit pairs yet (the first update of lam0 * I, i.e. the block of secant pairs of the initial
        // diagonalisation): the eigenpairs ARE those of the core — W = F^T Qb, mu = lam0 + sigma — no merges at all
        std::vector<double> Fk;
        int keep = 0;
        std::vector<int> cols;
        for (int t = 0; t < mb; ++t)
            if (fabs(sig[t]) > drop) cols.push_back(t);
        keep = (int)cols.size();
        if (keep > cap) { set_error("structured eigen-update: capacity of %d rows exhausted", cap); return SELLA_E_INVALID; }
        if (keep > 0) {
            Fk.resize((size_t)mb * keep);
            for (int j = 0; j < mb; ++j)
                for (int t = 0; t < keep; ++t) Fk[(size_t)j * keep + t] = F[(size_t)j * mb + cols[t]];
            SCHK(h2d_async(c, W.Ut, Fk.data(), Fk.size() * sizeof(double)));
            SCHK(launch_lincomb(c, n, keep, Qb, ld, mb, W.Ut, keep, nullptr, 0, 0, nullptr, 0, 0.0, Wt->d, ld));
            for (int t = 0; t < keep; ++t) mu[t] = lam0 + sig[cols[t]];          // sym_eig: ascending
        }
        *r_io = keep;
        if (nrank1) *nrank1 = keep;
        return stream_wait(c);
    }
    double* coef = c->dscal + DS_STAGE;
    std::vector<int> ord(mb);
    std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return fabs(sig[a]) > fabs(sig[b]); });
    for (int t = 0; t < mb; ++t) {
        const int rr = ord[t];
        if (fabs(sig[rr]) <= drop) continue;
        double* st = c->hscal + DS_STAGE + (size_t)t * 64;
        for (int i = 0; i < mb; ++i) st[i] = F[(size_t)i * mb + rr];
        HIPCHK(hipMemcpyAsync(coef + (size_t)t * 64, st, (size_t)mb * sizeof(double), hipMemcpyHostToDevice, c->stream));
        SCHK(launch_lincomb(c, n, 1, Qb, ld, mb, coef + (size_t)t * 64, 1, nullptr, 0, 0, nullptr, 0, 0.0, qv, ld));
        // the part of q outside span(W) is an eigenvector of B for lam0: it becomes an explicit row (speculatively,
        // decided with the same round trip that brings z back)
        if (r >= cap) { set_error("structured eigen-update: capacity of %d rows exhausted", cap); return SELLA_E_INVALID; }
        SCHK(eig_rank1_update(c, W, r, n, ld, mu, Wt->d, qv, sig[rr], &lam0, &r));
        if (nrank1) ++*nrank1;
    }
    }
    // explicit rows whose eigenvalue is (still) exactly lam0 belong to the cluster again: drop them
    {
        int keep = 0;
        std::vector<int> idx;
        for (int i = 0; i < r; ++i)
            if (mu[i] != lam0) idx.push_back(i);
        keep = (int)idx.size();
        if (keep != r) {
            int* idxd = W.ibuf + 16;
            SCHK(h2d_async(c, idxd, idx.data(), (size_t)keep * sizeof(int)));
            SCHK(launch_gather_rows(c, Wt->d, ld, idxd, keep, n, W.Zb, ld));
            if (keep) SCHK(launch_axpby2d(c, keep, n, 1.0, W.Zb, ld, 0.0, nullptr, 0, Wt->d, ld));
            for (int i = 0; i < keep; ++i) mu[i] = mu[idx[i]];
            r = keep;
        }
    }
    *r_io = r;
    SCHK(stream_wait(c));
    return SELLA_OK;
}

}  // namespace sella

using namespace sella;

// Eigendecomposition of diag(D) + rho w w^T through the merge kernels of the divide-and-conquer
// stage (secular roots, Gu/Eisenstat weights, eigenvector rows).  No deflation is performed: the
// caller guarantees D strictly ascending, w_i != 0 and rho > 0.
extern "C" int sella_rank1_eig(sella_ctx* c, int K, const double* D, const double* w, double rho,
                               double* lam, double* Ut) {
    if (!c || !D || !w || !lam || K <= 0 || !(rho > 0.0)) {
        set_error("rank1_eig: invalid arguments");
        return SELLA_E_INVALID;
    }
    for (int i = 0; i < K; ++i)
        if (w[i] == 0.0 || (i > 0 && !(D[i] > D[i - 1]))) {
            set_error("rank1_eig: D must be strictly ascending and w nonzero (entry %d)", i);
            return SELLA_E_INVALID;
        }
    const int ldu = round_up(K, 8);
    double* buf;
    SCHK(scratch_get(c, SCR_EIG5, ((size_t)6 * ldu + 64) * sizeof(double) + (size_t)K * ldu * sizeof(double), &buf));
    double *Dd = buf, *wd = buf + ldu, *taud = buf + 2 * ldu, *lamd = buf + 3 * ldu, *zhd = buf + 4 * ldu;
    int* orgd = reinterpret_cast<int*>(buf + 5 * ldu);
    int* info = reinterpret_cast<int*>(buf + 6 * ldu);
    double* Ud = buf + 6 * ldu + 64;
    HIPCHK(hipMemcpyAsync(Dd, D, (size_t)K * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(wd, w, (size_t)K * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(s_memset0(c, info, 2 * sizeof(int)));
    hipLaunchKernelGGL(secular_kernel, dim3((K + 3) / 4), dim3(256), 0, c->stream, K, Dd, wd, rho, taud, orgd, lamd, info);
    hipLaunchKernelGGL(zhat_kernel, dim3((K + 3) / 4), dim3(256), 0, c->stream, K, Dd, wd, taud, orgd, zhd);
    hipLaunchKernelGGL(build_u_kernel, dim3(K), dim3(256), 0, c->stream, K, Dd, zhd, taud, orgd, Ud, ldu);
    HIPCHK(hipGetLastError());
    int hinfo[2];
    HIPCHK(hipMemcpyAsync(hinfo, info, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(lam, lamd, (size_t)K * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (Ut)
        HIPCHK(hipMemcpy2DAsync(Ut, (size_t)K * sizeof(double), Ud, (size_t)ldu * sizeof(double),
                                (size_t)K * sizeof(double), K, hipMemcpyDeviceToHost, c->stream));
    SCHK(stream_wait(c));
    if (hinfo[1] != 0) {
        set_error("rank1_eig: secular equation solver hit its iteration cap (root %d)", hinfo[1] - 1);
        return SELLA_E_NOCONV;
    }
    return SELLA_OK;
}

extern "C" int sella_eigh(sella_ctx* c, sella_mat hA, double* w, sella_mat* hV, sella_mat* hVt) {
    Mat* a = mat_get(c, hA);
    if (!a || !w) return SELLA_E_INVALID;
    const int n = a->rows;
    if (a->cols != n || n <= 0) { set_error("eigh: matrix must be square"); return SELLA_E_INVALID; }
    const int ld = round_up(n, 8);
    EighWork W;
    W.c = c; W.n = n; W.ld = ld;
    const size_t mbytes = ((size_t)std::max(n, 64) + WY_NB2 + 2) * std::max(ld, 64) * sizeof(double);
    SCHK(scratch_get(c, SCR_EIG0, mbytes, &W.A));
    SCHK(scratch_get(c, SCR_EIG1, mbytes, &W.Za));
    SCHK(scratch_get(c, SCR_EIG2, mbytes, &W.Zb));
    SCHK(scratch_get(c, SCR_EIG3, mbytes, &W.Zc));
    SCHK(scratch_get(c, SCR_EIG4, mbytes, &W.Ut));
    SCHK(scratch_get(c, SCR_EIG5, (size_t)V_NSLOTS * ld * sizeof(double) + (size_t)(8 * n + 512) * sizeof(int), &W.vec));
    W.ibuf = reinterpret_cast<int*>(W.vec + (size_t)V_NSLOTS * ld);
    a = mat_get(c, hA);
    SCHK(launch_axpby2d(c, n, n, 1.0, a->d, a->ld, 0.0, nullptr, 0, W.A, ld));

    const bool dbg_time = getenv("SELLA_DEBUG_TIMING") != nullptr;
    auto now = [&] { if (dbg_time) (void)hipStreamSynchronize(c->stream); return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_s0 = now();
    // eigenvector rows X -> the two output matrices (shared by both factorisations)
    auto outputs = [&](double* X) -> int {
        sella_mat vt = SELLA_NO_MAT, v = SELLA_NO_MAT;
        SCHK(mat_new(c, n, n, &vt));
        Mat* mvt = mat_get(c, vt);
        SCHK(launch_axpby2d(c, n, n, 1.0, X, ld, 0.0, nullptr, 0, mvt->d, mvt->ld));
        if (hV) {
            SCHK(mat_new(c, n, n, &v));
            mvt = mat_get(c, vt);
            Mat* mv = mat_get(c, v);
            SCHK(launch_transpose(c, mvt->d, n, n, mvt->ld, mv->d, mv->ld));
            *hV = v;
        }
        SCHK(stream_wait(c));
        if (hVt) *hVt = vt;
        else sella_mat_free(c, vt);
        return SELLA_OK;
    };
    // ---- stage 1 ----------------------------------------------------------------------------
    double* dvec = W.vec + (size_t)V_D * ld;
    double* evec = W.vec + (size_t)V_E * ld;
    double* taus = W.vec + (size_t)V_TAUS * ld;
    HIPCHK(s_memset0(c, W.vec, (size_t)V_NSLOTS * ld * sizeof(double)));
    SCHK(tridiagonalise(W, taus, dvec, evec));
    std::vector<double> d(n), e(n);
    {
        static_assert(V_E == V_D + 1, "slot order");
        void* st;
        SCHK(host_stage(c, ((size_t)ld + n) * sizeof(double), &st));       // pinned: one asynchronous download
        const double* hd = static_cast<const double*>(st);
        HIPCHK(hipMemcpyAsync(st, dvec, ((size_t)ld + n) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        SCHK(stream_wait(c));
        std::copy(hd, hd + n, d.begin());
        std::copy(hd + ld, hd + ld + n, e.begin());
    }
    // The compact-WY factors of the back-transformation depend on the reflectors only: their Gram matrices
    // and triangular factors are enqueued now and run while the host plans the divide & conquer stage.
    const int nrefl = n - 2;
    // 64 reflectors per block from `eigh_wy_nb64_min` rows on (MFMA path with 16 rows and 4 wavefronts per workgroup)
    const bool wy64 = c->opt.eigh_wy_mfma && c->opt.eigh_wy_rows != 32 && c->opt.eigh_wy_waves < 8 &&
                      c->opt.eigh_wy_nb64_min > 0 && n >= c->opt.eigh_wy_nb64_min;
    const int wynb = wy64 ? WY_NB2 : WY_NB;
    const int nblk = nrefl > 0 ? (nrefl + wynb - 1) / wynb : 0;
    double* Gd = nullptr;                                // nblk x nb x nb: Gram matrices, then C = T^T
    hipStream_t wy_stream = c->stream;
    if ((hV || hVt) && nblk > 0) {
        // 64-blocks: the Gram matrix of a block in `gsl` column slices (one workgroup per block would leave most of the
        // chip idle: 48 blocks at n = 3072, 1.27 ms), partial matrices behind the C array
        const int gsl = wy64 ? std::max(1, std::min(8, (256 + nblk - 1) / nblk)) : 0;
        SCHK(scratch_get(c, SCR_EIG6, (size_t)nblk * (1 + gsl) * wynb * wynb * sizeof(double), &Gd));
        // On a second stream (round 5): 0.45 ms of Gram / triangular-factor kernels at n = 3072 (and the explicit
        // reflectors, 0.03 ms) used to sit on the main stream in FRONT of the divide & conquer stage, whose levels are
        // small kernels between host round trips — the chip has room for both.  Joined in front of the back-transformation.
        if (c->opt.eigh_wy_overlap && !c->prof) {
            if (!c->stream2) {
                if (hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess ||
                    hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
                    set_error("second stream for the compact-WY factors could not be created");
                    return SELLA_E_HIP;
                }
            }
            HIPCHK(hipEventRecord(c->ev_fork, c->stream));
            HIPCHK(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
            wy_stream = c->stream2;
        }
        if (wy64) {
            double* Gpart = Gd + (size_t)nblk * wynb * wynb;
            hipLaunchKernelGGL(wy_gram64_kernel, dim3(nblk, gsl), dim3(256), 0, wy_stream, W.A, ld, n, nrefl, taus, Gpart);
            hipLaunchKernelGGL(wy_tinv64_kernel, dim3(nblk), dim3(64), 0, wy_stream, Gd, Gpart, gsl, nrefl, taus);
        } else {
            hipLaunchKernelGGL(wy_gram_kernel, dim3(nblk), dim3(256), 0, wy_stream, W.A, ld, n, nrefl, taus, Gd);
            hipLaunchKernelGGL(wy_tinv_kernel, dim3(nblk), dim3(64), 0, wy_stream, Gd, nrefl, taus);
        }
        if (wy_stream != c->stream) HIPCHK(hipEventRecord(c->ev_join, wy_stream));
        HIPCHK(hipGetLastError());
    }

    const double t_s1 = now();
    // ---- stage 2 ----------------------------------------------------------------------------
    // The waits of divide & conquer cover the main stream only while the factor kernels run beside it (they read the
    // reflectors and write SCR_EIG6, nothing the levels touch, and use neither ring) — otherwise the first level's wait
    // would block on them and the overlap would end there.  Joined again whatever the outcome.
    const bool detached = wy_stream != c->stream;
    c->stream2_detached = detached;
    const int dc_status = dc_solve(W, d, e, w);
    c->stream2_detached = false;
    if (dc_status != SELLA_OK) {
        if (detached) (void)hipStreamWaitEvent(c->stream, c->ev_join, 0);
        return dc_status;
    }
    if (!hV && !hVt) return SELLA_OK;

    const double t_s2 = now();
    // ---- stage 3: X = Z H_{n-3} ... H_0 (rows) -------------------------------------------------
    double* X = W.Za;
    if (nrefl > 0) {
        const int yrows = nblk * wynb;
        double* Yf = W.Zc;                               // explicit reflectors (yrows x ld)
        if (wy_stream != c->stream) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
        hipLaunchKernelGGL(wy_expand_kernel, dim3((ld + 255) / 256, yrows), dim3(256), 0, c->stream, W.A, ld, n, nrefl,
                           taus, Yf);
        prof_begin(c, PROF_OTHER, 0.0, 2.0 * n * (double)n * n);
        if (c->opt.eigh_wy_mfma)
        {
            // wavefronts per workgroup: a workgroup owns 16 rows of X, so there are only n / 16 of them (one per CU at
            // n = 3072) — more wavefronts splitting the columns is what hides the L2 latency of the operand streams
            const long nw = c->opt.eigh_wy_waves;
            if (wy64 && c->opt.eigh_wy_strip && ld == n && (n & 63) == 0 && n <= 64 * 8 * 6 && (n > 64 * 8 * 4 || c->opt.eigh_wy_strip == 2))
                SELLA_LAUNCH(c, HIP_KERNEL_NAME(wy_apply_strip_kernel<6, 8>), dim3(n / 16), dim3(512), 0, X, ld, n, Yf, Gd, nblk);
            else if (wy64)
                SELLA_LAUNCH(c, wy_apply_mfma64_kernel<4>, dim3((n + 15) / 16), dim3(256), 0, X, ld, n, Yf, Gd, nblk);
            else if (c->opt.eigh_wy_rows == 32 && nw >= 8)
                SELLA_LAUNCH(c, wy_apply_mfma2_kernel<8>, dim3((n + 31) / 32), dim3(512), 0, X, ld, n, Yf, Gd, nblk);
            else if (c->opt.eigh_wy_rows == 32)
                SELLA_LAUNCH(c, wy_apply_mfma2_kernel<4>, dim3((n + 31) / 32), dim3(256), 0, X, ld, n, Yf, Gd, nblk);
            else if (nw >= 16) SELLA_LAUNCH(c, wy_apply_mfma_kernel<16>, dim3((n + 15) / 16), dim3(1024), 0, X, ld, n, Yf, Gd, nblk);
            else if (nw >= 8) SELLA_LAUNCH(c, wy_apply_mfma_kernel<8>, dim3((n + 15) / 16), dim3(512), 0, X, ld, n, Yf, Gd, nblk);
            else SELLA_LAUNCH(c, wy_apply_mfma_kernel<4>, dim3((n + 15) / 16), dim3(256), 0, X, ld, n, Yf, Gd, nblk);
        }
        else
            SELLA_LAUNCH(c, wy_apply_kernel, dim3((n + 15) / 16), dim3(256), 0, X, ld, n, W.A, ld, nrefl, taus, Gd, nblk);
        prof_end(c);
        HIPCHK(hipGetLastError());
    }
    const double t_s3 = now();
    // ---- outputs -------------------------------------------------------------------------------
    SCHK(outputs(X));
    if (dbg_time)
        fprintf(stderr, "eigh n=%d: tridiagonalise %.2f ms, divide&conquer %.2f ms, back-transform %.2f ms, outputs %.2f ms\n", n,
                1e3 * (t_s1 - t_s0), 1e3 * (t_s2 - t_s1), 1e3 * (t_s3 - t_s2), 1e3 * (now() - t_s3));
    return SELLA_OK;
}
