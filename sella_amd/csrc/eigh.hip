// eigh.hip — fp64 symmetric eigensolver on the device (replaces torch.linalg.eigh behind
// gpu_eigh / gpu_eigh_t, sella/_gpu.py:70-97; consumers: ApproximateHessian.evals/evecs
// linalg.py:174-231, the TS-BFGS |B| term hessian_update.py:121, the P-RFO split stepper.py:163).
//
// Three stages, all device-resident:
//   1. Householder tridiagonalisation  Q^T A Q = T.  The full symmetric trailing matrix is
//      kept up to date, so the matrix-vector product of each step is the same row-panel matvec
//      that drives the Davidson loop (coalesced 16-byte streams, wave64 reductions).
//   2. Divide and conquer on T (Cuppen, with Gu/Eisenstat's stable eigenvectors): leaves by
//      implicit QL (one wavefront per leaf), merges = deflation (host, O(N log N)) + secular
//      equation (one thread per root) + one GEMM per merge (MFMA f64).
//   3. Back-transformation with compact-WY blocks: X <- X (I - Y T Y^T)^T, three GEMMs per block.
// Eigenvectors are handled as ROWS of a row-major matrix throughout (vector-major, like the
// Krylov panels), so rotations, gathers and the final Q^T x products are all coalesced.
#include "internal.h"
#include "host_math.h"
#include "secular.h"

#include <algorithm>
#include <numeric>

namespace sella {
namespace {

__device__ __forceinline__ double wave_sum_e(double v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

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
// Reflector for step j from row j of the (symmetric, fully updated) matrix.
// vpad[0] = 0, vpad[1 + i] = v_i (v_0 = 1); the tail v_1.. is also stored in A[j][j+2..]
__global__ __launch_bounds__(256) void house_gen_kernel(double* __restrict__ A, int ld, int n, int j,
                                                        double* __restrict__ vpad,
                                                        double* __restrict__ taus,
                                                        double* __restrict__ dvec,
                                                        double* __restrict__ evec) {
    __shared__ double red[4];
    const int m = n - j - 1;
    double* x = A + (size_t)j * ld + j + 1;
    double ss = 0.0;
    for (int i = 1 + threadIdx.x; i < m; i += 256) ss += x[i] * x[i];
    ss = block_sum_256(ss, red);
    const double alpha = x[0];
    double beta, tau, scale;
    if (ss == 0.0) {
        beta = alpha; tau = 0.0; scale = 0.0;
    } else {
        const double nrm = sqrt(alpha * alpha + ss);
        beta = (alpha >= 0.0) ? -nrm : nrm;
        tau = (beta - alpha) / beta;
        scale = 1.0 / (alpha - beta);
    }
    __syncthreads();
    for (int i = 1 + threadIdx.x; i < m; i += 256) {
        const double v = x[i] * scale;
        x[i] = v;
        vpad[1 + i] = v;
    }
    if (threadIdx.x == 0) {
        vpad[0] = 0.0;
        vpad[1] = 1.0;
        taus[j] = tau;
        evec[j] = beta;
        dvec[j] = A[(size_t)j * ld + j];
    }
}

// w = tau*q - (tau^2/2)(q.v) v
__global__ __launch_bounds__(256) void house_w_kernel(const double* __restrict__ q,
                                                      const double* __restrict__ v,
                                                      const double* __restrict__ taup, int m,
                                                      double* __restrict__ w) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < m; i += 256) s += q[i] * v[i];
    s = block_sum_256(s, red);
    const double tau = taup[0];
    const double c2 = 0.5 * tau * tau * s;
    for (int i = threadIdx.x; i < m; i += 256) w[i] = tau * q[i] - c2 * v[i];
}

// A22[i][k] -= v[i] w[k] + w[i] v[k]; tile = 8 rows x 256 columns per workgroup
__global__ __launch_bounds__(256) void rank2_kernel(double* __restrict__ A, int ld, int m,
                                                    const double* __restrict__ v,
                                                    const double* __restrict__ w) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int i0 = blockIdx.y * 8;
    if (k >= m) return;
    const double vk = v[k], wk = w[k];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int i = i0 + r;
        if (i < m) A[(size_t)i * ld + k] -= v[i] * wk + w[i] * vk;
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

// z_i = Zt[rows[i]][col_i] * sign_i with col = mid-1 for the left child, mid for the right one
__global__ __launch_bounds__(256) void gather_z_kernel(const double* __restrict__ Zt, int ld, int lo,
                                                       int n1, int N, int mid, double sgn,
                                                       double* __restrict__ z) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    z[i] = (i < n1) ? Zt[(size_t)(lo + i) * ld + mid - 1] : sgn * Zt[(size_t)(lo + i) * ld + mid];
}

// apply a chain of Givens rotations to pairs of rows of the block Zb (N columns); rotation r:
// x = row i1, y = row i2:  x' = c x + s y ; y' = c y - s x     (BLAS drot convention)
__global__ __launch_bounds__(256) void rot_rows_kernel(double* __restrict__ Zb, int ld, int N, int nrot,
                                                       const int* __restrict__ i1,
                                                       const int* __restrict__ i2,
                                                       const double* __restrict__ cs) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= N) return;
    for (int r = 0; r < nrot; ++r) {
        double* xp = Zb + (size_t)i1[r] * ld + col;
        double* yp = Zb + (size_t)i2[r] * ld + col;
        const double c = cs[2 * r], s = cs[2 * r + 1];
        const double x = *xp, y = *yp;
        *xp = c * x + s * y;
        *yp = c * y - s * x;
    }
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

// one WAVEFRONT per root: the 64 lanes split every O(K) sum of the iteration (K wavefronts in
// flight instead of K/64), all lanes follow the same control flow on wave-reduced values
__global__ __launch_bounds__(256) void secular_kernel(int K, const double* __restrict__ D,
                                                      const double* __restrict__ w, double rho,
                                                      double* __restrict__ tau, int* __restrict__ org,
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

__global__ __launch_bounds__(256) void zhat_kernel(int K, const double* __restrict__ D,
                                                   const double* __restrict__ w,
                                                   const double* __restrict__ tau,
                                                   const int* __restrict__ org, double* __restrict__ zh) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= K) return;
    const double z = secular::zhat(K, D, w, tau, org, i, lane, 64, WaveProd());
    if (lane == 0) zh[i] = z;
}

// Ut[j][i] = zhat_i / ((D_i - D_org_j) - tau_j), row j normalised
__global__ __launch_bounds__(256) void build_u_kernel(int K, const double* __restrict__ D,
                                                      const double* __restrict__ zh,
                                                      const double* __restrict__ tau,
                                                      const int* __restrict__ org,
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

// ---------------------------------------------------------------------------------------
// stage 3 kernels
// ---------------------------------------------------------------------------------------
// Yt[r][c] = r-th reflector of the block (zero-padded, unit at c = j+1); zero row if tau == 0
__global__ __launch_bounds__(256) void build_y_kernel(const double* __restrict__ A, int ld, int n, int j0,
                                                      int nb, const double* __restrict__ taus,
                                                      double* __restrict__ Yt, int ldy) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    if (c >= n || r >= nb) return;
    const int j = j0 + r;
    double v = 0.0;
    if (taus[j] != 0.0) {
        if (c == j + 1) v = 1.0;
        else if (c > j + 1) v = A[(size_t)j * ld + c];
    }
    Yt[(size_t)r * ldy + c] = v;
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
    double *vec;               // d, e, tau, vpad, q, w, z, D, w, tau, zhat, lam ... (device)
    int* ibuf;                 // device ints
};

}  // namespace

// Divide and conquer on the tridiagonal (d, e) (host copies, modified).  On exit w holds the
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
    double* ddev = W.vec;               // n
    double* edev = W.vec + W.ld;        // n
    double* wdev = W.vec + 2 * (size_t)W.ld;   // leaf eigenvalues (n)
    HIPCHK(hipMemcpyAsync(ddev, d.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(edev, e.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    // leaves
    const std::vector<int>& leaves = by_height[0];
    std::vector<int> ranges;
    for (int li : leaves) { ranges.push_back(nodes[li].lo); ranges.push_back(nodes[li].hi); }
    int* info = W.ibuf;                 // 2 ints
    int* rdev = W.ibuf + 8;
    HIPCHK(hipMemsetAsync(info, 0, 8 * sizeof(int), c->stream));
    HIPCHK(hipMemcpyAsync(rdev, ranges.data(), ranges.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(set_identity_kernel, dim3((n + 255) / 256, n), dim3(256), 0, c->stream, W.Za, ld, n);
    hipLaunchKernelGGL(leaf_ql_kernel, dim3((unsigned)leaves.size()), dim3(64), 0, c->stream, ddev, edev, wdev,
                       rdev, W.Za, ld, info);
    HIPCHK(hipGetLastError());
    std::vector<double> vals(n);
    int hinfo[2];
    HIPCHK(hipMemcpyAsync(vals.data(), wdev, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(hinfo, info, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (hinfo[0] != 0) {
        set_error("eigh: QL iteration did not converge in a leaf (eigenvalue %d)", hinfo[0]);
        return SELLA_E_NOCONV;
    }

    const double eps = 2.220446049250313e-16;
    double* cur = W.Za;
    double* nxt = W.Zb;
    std::vector<double> z, Dn, wn, cs, lam;
    std::vector<int> order, nondef, defl, r1, r2, idx;
    for (int h = 1; h <= maxdepth; ++h) {
        // eigenvector rows of a node are supported on its own column range only: the blocks of
        // `nxt` outside the diagonal blocks written below must read as zero at the next level
        HIPCHK(hipMemsetAsync(nxt, 0, (size_t)n * ld * sizeof(double), c->stream));
        for (int ni : by_height[h]) {
            const Node& nd = nodes[ni];
            const int lo = nd.lo, hi = nd.hi, mid = nd.mid, N = hi - lo, n1 = mid - lo;
            const double em = e[mid - 1];
            double* zdev = W.vec + 2 * (size_t)W.ld;
            hipLaunchKernelGGL(gather_z_kernel, dim3((N + 255) / 256), dim3(256), 0, c->stream, cur, ld, lo, n1, N,
                               mid, em < 0 ? -1.0 : 1.0, zdev);
            HIPCHK(hipGetLastError());
            z.resize(N);
            HIPCHK(hipMemcpyAsync(z.data(), zdev, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            double* D = vals.data() + lo;       // eigenvalues of the two children, row order
            // ---- deflation (host) ------------------------------------------------------------
            const double rho = fabs(2.0 * em);
            double zmax = 0.0, dmax = 0.0;
            for (int i = 0; i < N; ++i) {
                z[i] *= 0.7071067811865476;
                zmax = std::max(zmax, fabs(z[i]));
                dmax = std::max(dmax, fabs(D[i]));
            }
            const double tol = 8.0 * eps * std::max(dmax, zmax);
            order.resize(N);
            std::iota(order.begin(), order.end(), 0);
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return D[a] < D[b]; });
            nondef.clear(); defl.clear(); r1.clear(); r2.clear(); cs.clear();
            if (rho * zmax <= tol) {
                defl = order;
            } else {
                int pj = -1;
                for (int jj = 0; jj < N; ++jj) {
                    const int nj = order[jj];
                    if (rho * fabs(z[nj]) <= tol) { defl.push_back(nj); continue; }
                    if (pj < 0) { pj = nj; continue; }
                    double s = z[pj], cc = z[nj];
                    const double tau = hypot(cc, s);
                    const double t = D[nj] - D[pj];
                    cc /= tau;
                    s = -s / tau;
                    if (fabs(t * cc * s) <= tol) {
                        z[nj] = tau;
                        z[pj] = 0.0;
                        r1.push_back(pj); r2.push_back(nj); cs.push_back(cc); cs.push_back(s);
                        const double tt = D[pj] * cc * cc + D[nj] * s * s;
                        D[nj] = D[pj] * s * s + D[nj] * cc * cc;
                        D[pj] = tt;
                        defl.push_back(pj);
                        pj = nj;
                    } else {
                        nondef.push_back(pj);
                        pj = nj;
                    }
                }
                if (pj >= 0) nondef.push_back(pj);
            }
            const int K = (int)nondef.size();
            const int nrot = (int)r1.size();
            if (getenv("SELLA_DEBUG")) {
                fprintf(stderr, "merge [%d,%d) mid %d em %g rho %g K %d nrot %d\n  D:", lo, hi, mid, em, rho, K, nrot);
                for (int i = 0; i < N; ++i) fprintf(stderr, " %.6f", D[i]);
                fprintf(stderr, "\n  z:");
                for (int i = 0; i < N; ++i) fprintf(stderr, " %.6f", z[i]);
                fprintf(stderr, "\n");
            }
            if (nrot > 0) {
                int* i1d = W.ibuf + 16 + 2 * (int)leaves.size() + 16;
                int* i2d = i1d + N;
                double* csd = W.vec + 3 * (size_t)W.ld;    // 2*N doubles (two ld slots)
                HIPCHK(hipMemcpyAsync(i1d, r1.data(), (size_t)nrot * sizeof(int), hipMemcpyHostToDevice, c->stream));
                HIPCHK(hipMemcpyAsync(i2d, r2.data(), (size_t)nrot * sizeof(int), hipMemcpyHostToDevice, c->stream));
                HIPCHK(hipMemcpyAsync(csd, cs.data(), (size_t)2 * nrot * sizeof(double), hipMemcpyHostToDevice, c->stream));
                hipLaunchKernelGGL(rot_rows_kernel, dim3((N + 255) / 256), dim3(256), 0, c->stream,
                                   cur + (size_t)lo * ld + lo, ld, N, nrot, i1d, i2d, csd);
                HIPCHK(hipGetLastError());
                HIPCHK(hipStreamSynchronize(c->stream));   // host vectors are re-used below
            }
            // the nondeflated D may have lost strict ordering by a rounding; keep as produced
            // ---- secular equation + inner eigenvectors (device) ----------------------------------
            idx.clear();
            for (int i : nondef) idx.push_back(lo + i);
            for (int i : defl) idx.push_back(lo + i);
            int* idxd = W.ibuf + 16 + 2 * (int)leaves.size() + 16 + 2 * n;
            HIPCHK(hipMemcpyAsync(idxd, idx.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice, c->stream));
            lam.assign(N, 0.0);
            if (K > 0) {
                Dn.resize(K); wn.resize(K);
                for (int p = 0; p < K; ++p) { Dn[p] = D[nondef[p]]; wn[p] = z[nondef[p]]; }
                double* Dd = W.vec + 5 * (size_t)W.ld;
                double* wd = W.vec + 6 * (size_t)W.ld;
                double* taud = W.vec + 7 * (size_t)W.ld;
                double* zhd = W.vec + 8 * (size_t)W.ld;
                double* lamd = W.vec + 9 * (size_t)W.ld;
                int* orgd = W.ibuf + 16 + 2 * (int)leaves.size() + 16 + 3 * n;
                HIPCHK(hipMemcpyAsync(Dd, Dn.data(), (size_t)K * sizeof(double), hipMemcpyHostToDevice, c->stream));
                HIPCHK(hipMemcpyAsync(wd, wn.data(), (size_t)K * sizeof(double), hipMemcpyHostToDevice, c->stream));
                hipLaunchKernelGGL(secular_kernel, dim3((K + 3) / 4), dim3(256), 0, c->stream, K, Dd, wd, rho, taud,
                                   orgd, lamd, info);
                hipLaunchKernelGGL(zhat_kernel, dim3((K + 3) / 4), dim3(256), 0, c->stream, K, Dd, wd, taud, orgd, zhd);
                const int ldu = round_up(K, 8);
                hipLaunchKernelGGL(build_u_kernel, dim3(K), dim3(256), 0, c->stream, K, Dd, zhd, taud, orgd, W.Ut, ldu);
                HIPCHK(hipGetLastError());
                // compact the non-deflated rows (sorted) and multiply
                SCHK(launch_gather_rows(c, cur + lo, ld, idxd, K, N, W.Zc, ld));
                SCHK(launch_gemm(c, 0, 0, K, N, K, 1.0, W.Ut, ldu, W.Zc, ld, 0.0, nxt + (size_t)lo * ld + lo, ld));
                HIPCHK(hipMemcpyAsync(lam.data(), lamd, (size_t)K * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            }
            if (N - K > 0)
                SCHK(launch_gather_rows(c, cur + lo, ld, idxd + K, N - K, N, nxt + (size_t)(lo + K) * ld + lo, ld));
            HIPCHK(hipStreamSynchronize(c->stream));
            for (int p = 0; p < N - K; ++p) lam[K + p] = D[defl[p]];
            for (int p = 0; p < N; ++p) D[p] = lam[p];
        }
        std::swap(cur, nxt);
        int hi2[2];
        HIPCHK(hipMemcpyAsync(hi2, info, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (hi2[1] != 0) {
            set_error("eigh: secular equation solver hit its iteration cap (root %d)", hi2[1] - 1);
            return SELLA_E_NOCONV;
        }
    }
    // final ascending order
    order.resize(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return vals[a] < vals[b]; });
    for (int i = 0; i < n; ++i) wout[i] = vals[order[i]];
    int* idxd = W.ibuf + 16 + 2 * (int)leaves.size() + 16 + 2 * n;
    HIPCHK(hipMemcpyAsync(idxd, order.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    SCHK(launch_gather_rows(c, cur, ld, idxd, n, n, nxt, ld));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (nxt != W.Za) std::swap(W.Za, W.Zb);
    return SELLA_OK;
}

}  // namespace sella

using namespace sella;

extern "C" int sella_eigh(sella_ctx* c, sella_mat hA, double* w, sella_mat* hV, sella_mat* hVt) {
    Mat* a = mat_get(c, hA);
    if (!a || !w) return SELLA_E_INVALID;
    const int n = a->rows;
    if (a->cols != n || n <= 0) { set_error("eigh: matrix must be square"); return SELLA_E_INVALID; }
    const int ld = round_up(n, 8);
    EighWork W;
    W.c = c; W.n = n; W.ld = ld;
    // (the compact-WY work arrays of stage 3 need at least 64 rows / columns)
    const size_t mbytes = ((size_t)std::max(n, 64) + 2) * std::max(ld, 64) * sizeof(double);
    SCHK(scratch_get(c, SCR_EIG0, mbytes, &W.A));
    SCHK(scratch_get(c, SCR_EIG1, mbytes, &W.Za));
    SCHK(scratch_get(c, SCR_EIG2, mbytes, &W.Zb));
    SCHK(scratch_get(c, SCR_EIG3, mbytes, &W.Zc));
    SCHK(scratch_get(c, SCR_EIG4, mbytes, &W.Ut));
    SCHK(scratch_get(c, SCR_EIG5, (size_t)16 * ld * sizeof(double) + (size_t)(6 * n + 256) * sizeof(int), &W.vec));
    W.ibuf = reinterpret_cast<int*>(W.vec + 16 * (size_t)ld);
    a = mat_get(c, hA);
    SCHK(launch_axpby2d(c, n, n, 1.0, a->d, a->ld, 0.0, nullptr, 0, W.A, ld));

    // ---- stage 1: tridiagonalisation --------------------------------------------------------
    double* dvec = W.vec;
    double* evec = W.vec + ld;
    double* taus = W.vec + 10 * (size_t)ld;
    double* vpad = W.vec + 11 * (size_t)ld;     // 1 + m values (+ slack)
    double* qv = W.vec + 13 * (size_t)ld;
    double* wv = W.vec + 14 * (size_t)ld;
    HIPCHK(hipMemsetAsync(vpad, 0, 2 * (size_t)ld * sizeof(double), c->stream));
    for (int j = 0; j + 2 < n; ++j) {
        const int m = n - j - 1, o = j + 1;
        hipLaunchKernelGGL(house_gen_kernel, dim3(1), dim3(256), 0, c->stream, W.A, ld, n, j, vpad, taus, dvec, evec);
        HIPCHK(hipGetLastError());
        // q = A22 v ; start the row stream at an even column so 16-byte loads stay aligned
        const int oc = o & ~1;
        const double* x = vpad + 1 - (o - oc);
        SCHK(launch_gemv_rows(c, W.A + (size_t)o * ld + oc, m, m + (o - oc), ld, x, ld, 1, qv, ld, GemvEpi()));
        hipLaunchKernelGGL(house_w_kernel, dim3(1), dim3(256), 0, c->stream, qv, vpad + 1, taus + j, m, wv);
        hipLaunchKernelGGL(rank2_kernel, dim3((m + 255) / 256, (m + 7) / 8), dim3(256), 0, c->stream,
                           W.A + (size_t)o * ld + o, ld, m, vpad + 1, wv);
        HIPCHK(hipGetLastError());
    }
    hipLaunchKernelGGL(tridiag_tail_kernel, dim3(1), dim3(64), 0, c->stream, W.A, ld, n, dvec, evec, taus);
    HIPCHK(hipGetLastError());
    std::vector<double> d(n), e(n), tauh(n);
    HIPCHK(hipMemcpyAsync(d.data(), dvec, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(e.data(), evec, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(tauh.data(), taus, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));

    // ---- stage 2: divide and conquer on T ------------------------------------------------------
    SCHK(dc_solve(W, d, e, w));
    if (!hV && !hVt) return SELLA_OK;

    // ---- stage 3: X = Z H_{n-3} ... H_0 (rows), compact-WY blocks from the last to the first ----
    double* X = W.Za;
    const int nrefl = n - 2;
    if (nrefl > 0) {
        const int nb = 32;
        double* Yt = W.Zb;                       // nb x n
        double* Wt = W.Zb + (size_t)nb * ld;     // nb x n
        double* Mx = W.Zc;                       // n x nb (ld = nbp)
        double* Gd = W.Ut;                       // nb x nb
        const int nbp = round_up(nb, 8);
        std::vector<double> G((size_t)nb * nb), S((size_t)nb * nb), Tm((size_t)nb * nb);
        const int nblk = (nrefl + nb - 1) / nb;
        for (int b = nblk - 1; b >= 0; --b) {
            const int j0 = b * nb;
            const int kb = std::min(nb, nrefl - j0);
            const int c0 = j0 + 1;               // first column touched by this block
            const int nc = n - c0;
            hipLaunchKernelGGL(build_y_kernel, dim3((n + 255) / 256, kb), dim3(256), 0, c->stream, W.A, ld, n, j0, kb,
                               taus, Yt, ld);
            HIPCHK(hipGetLastError());
            SCHK(launch_gemm(c, 0, 1, kb, kb, nc, 1.0, Yt + c0, ld, Yt + c0, ld, 0.0, Gd, nbp));
            HIPCHK(hipMemcpy2DAsync(G.data(), (size_t)kb * sizeof(double), Gd, (size_t)nbp * sizeof(double),
                                    (size_t)kb * sizeof(double), kb, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            // S = striu(G) + diag(1/tau) ; T = S^-1 (upper) ; we need C = T^T
            for (int i = 0; i < kb; ++i)
                for (int k2 = 0; k2 < kb; ++k2) {
                    double v = 0.0;
                    if (k2 > i) v = G[(size_t)i * kb + k2];
                    else if (k2 == i) v = (tauh[j0 + i] != 0.0) ? 1.0 / tauh[j0 + i] : 1.0;
                    S[(size_t)i * kb + k2] = v;
                }
            // invert upper-triangular S by back substitution, column by column
            for (int col = 0; col < kb; ++col) {
                for (int i = kb - 1; i >= 0; --i) {
                    double s = (i == col) ? 1.0 : 0.0;
                    for (int k2 = i + 1; k2 < kb; ++k2) s -= S[(size_t)i * kb + k2] * Tm[(size_t)k2 * kb + col];
                    Tm[(size_t)i * kb + col] = (i <= col) ? s / S[(size_t)i * kb + i] : 0.0;
                }
            }
            // upload C = T^T (kb x kb) into Gd
            std::vector<double> Cm((size_t)kb * kb);
            for (int i = 0; i < kb; ++i)
                for (int k2 = 0; k2 < kb; ++k2) Cm[(size_t)i * kb + k2] = Tm[(size_t)k2 * kb + i];
            HIPCHK(hipMemcpy2DAsync(Gd, (size_t)nbp * sizeof(double), Cm.data(), (size_t)kb * sizeof(double),
                                    (size_t)kb * sizeof(double), kb, hipMemcpyHostToDevice, c->stream));
            // Wt = C Yt ; Mx = X Yt^T ; X -= Mx Wt      (columns c0..n only)
            SCHK(launch_gemm(c, 0, 0, kb, nc, kb, 1.0, Gd, nbp, Yt + c0, ld, 0.0, Wt + c0, ld));
            SCHK(launch_gemm(c, 0, 1, n, kb, nc, 1.0, X + c0, ld, Yt + c0, ld, 0.0, Mx, nbp));
            SCHK(launch_gemm(c, 0, 0, n, nc, kb, -1.0, Mx, nbp, Wt + c0, ld, 1.0, X + c0, ld));
            HIPCHK(hipStreamSynchronize(c->stream));   // Cm goes out of scope
        }
    }
    // ---- outputs ---------------------------------------------------------------------------------
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
    HIPCHK(hipStreamSynchronize(c->stream));
    if (hVt) *hVt = vt;
    else sella_mat_free(c, vt);
    return SELLA_OK;
}
