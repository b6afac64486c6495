// qr.hip — economy Householder QR on the device (replaces torch.linalg.qr(mode='reduced') behind
// gpu_qr, sella/_gpu.py:100-111; used for the internal-coordinate Jacobian, peswrapper.py:674-709).
//
// The matrix is held transposed as a vector-major panel (row j = column j of A, contiguous), so
// the reflector of step j is a contiguous vector, its application to the remaining columns is
// the row-panel matvec (dots) plus a coalesced rank-1 update, and Q is formed the same way from
// the identity.  Same sign conventions as LAPACK dgeqrf/dorgqr (R_jj = -sign(a_jj) |a_j|).
#include "internal.h"

namespace sella {
namespace {

__device__ __forceinline__ double wave_sum_q(double v) { return wave_sum64(v); }

// Householder vector of x (len): x <- [beta, v_1, ...]; vpad[0] = 0, vpad[1..len] = [1, v_1, ...]
__global__ __launch_bounds__(256) void house_vec_kernel(double* __restrict__ x, int len,
                                                        double* __restrict__ vpad,
                                                        double* __restrict__ tau_out) {
    __shared__ double red[4];
    double ss = 0.0;
    for (int i = 1 + threadIdx.x; i < len; i += 256) ss += x[i] * x[i];
    ss = wave_sum_q(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    ss = red[0] + red[1] + red[2] + red[3];
    const double alpha = x[0];
    double beta, tau, scale;
    if (ss == 0.0) { beta = alpha; tau = 0.0; scale = 0.0; }
    else {
        const double nrm = sqrt(alpha * alpha + ss);
        beta = (alpha >= 0.0) ? -nrm : nrm;
        tau = (beta - alpha) / beta;
        scale = 1.0 / (alpha - beta);
    }
    __syncthreads();
    for (int i = 1 + threadIdx.x; i < len; i += 256) {
        const double v = x[i] * scale;
        x[i] = v;
        vpad[1 + i] = v;
    }
    if (threadIdx.x == 0) {
        vpad[0] = 0.0;
        vpad[1] = 1.0;
        x[0] = beta;
        tau_out[0] = tau;
    }
}

// P[l][i] -= tau * dots[l] * v[i]   (l < nrows, i < ncols)
__global__ __launch_bounds__(256) void rows_rank1_kernel(double* __restrict__ P, int ldp, int nrows,
                                                         int ncols, const double* __restrict__ dots,
                                                         const double* __restrict__ taup,
                                                         const double* __restrict__ v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int l0 = blockIdx.y * 8;
    if (i >= ncols) return;
    const double tv = taup[0] * v[i];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int l = l0 + r;
        if (l < nrows) P[(size_t)l * ldp + i] -= dots[l] * tv;
    }
}

// apply H = I - tau v v^T (v = vpad+1, support [j, m)) to rows [r0, r1) of the panel
int apply_reflector(sella_ctx* c, double* P, int ld, int r0, int r1, int j, int m, const double* vpad,
                    const double* taup, double* dots) {
    const int nrows = r1 - r0;
    if (nrows <= 0) return SELLA_OK;
    const int len = m - j;
    const int jc = j & ~1;                      // even start column for aligned 16-byte loads
    const double* x = vpad + 1 - (j - jc);      // x[0] = 0 pad when j is odd
    SCHK(launch_gemv_rows(c, P + (size_t)r0 * ld + jc, nrows, len + (j - jc), ld, x, ld, 1, dots, nrows, GemvEpi()));
    hipLaunchKernelGGL(rows_rank1_kernel, dim3((len + 255) / 256, (nrows + 7) / 8), dim3(256), 0, c->stream,
                       P + (size_t)r0 * ld + j, ld, nrows, len, dots, taup, vpad + 1);
    HIPCHK(hipGetLastError());
    return SELLA_OK;
}

}  // namespace
}  // namespace sella

using namespace sella;

extern "C" int sella_qr_thin(sella_ctx* c, const double* A, int m, int n, double* Q, double* R) {
    if (!c || !A || !Q || !R || m <= 0 || n <= 0 || m < n) {
        set_error("qr_thin: need m >= n >= 1");
        return SELLA_E_INVALID;
    }
    const int ld = round_up(m, 8) + 8;
    double *At, *Qt, *wk;
    SCHK(scratch_get(c, SCR_QR0, ((size_t)n + 2) * ld * sizeof(double), &At));
    SCHK(scratch_get(c, SCR_QR1, ((size_t)n + 2) * ld * sizeof(double), &Qt));
    SCHK(scratch_get(c, SCR_MISC1, (size_t)(3 * ld + 2 * n + 64) * sizeof(double), &wk));
    double* vpad = wk;                       // 1 + m
    double* taus = wk + 2 * (size_t)ld;      // n
    double* dots = taus + n + 8;             // n
    HIPCHK(s_memset0(c, wk, (size_t)(3 * ld + 2 * n + 64) * sizeof(double)));
    SCHK(upload_panel(c, A, m, n, At, ld));
    // factorisation
    for (int j = 0; j < n; ++j) {
        hipLaunchKernelGGL(house_vec_kernel, dim3(1), dim3(256), 0, c->stream, At + (size_t)j * ld + j, m - j, vpad,
                           taus + j);
        HIPCHK(hipGetLastError());
        SCHK(apply_reflector(c, At, ld, j + 1, n, j, m, vpad, taus + j, dots));
    }
    // R: upper triangle lives in At[l][j] for j <= l
    {
        std::vector<double> at((size_t)n * n);
        SCHK(d2h_async_2d(c, at.data(), At, (size_t)ld * sizeof(double), (size_t)n * sizeof(double), n));
        SCHK(stream_wait(c));
        for (int j = 0; j < n; ++j)
            for (int l = 0; l < n; ++l) R[(size_t)j * n + l] = (l >= j) ? at[(size_t)l * n + j] : 0.0;
    }
    // Q = H_0 ... H_{n-1} [I; 0]: rows of Qt start as unit vectors, reflectors applied last to first
    HIPCHK(s_memset0(c, Qt, ((size_t)n + 2) * ld * sizeof(double)));
    {
        std::vector<double> ones(n, 1.0);
        HIPCHK(hipMemcpy2DAsync(Qt, ((size_t)ld + 1) * sizeof(double), ones.data(), sizeof(double), sizeof(double), n,
                                hipMemcpyHostToDevice, c->stream));
        // a device-resident 1.0 in the spare row (source of the unit head of every reflector)
        HIPCHK(hipMemcpyAsync(Qt + (size_t)(n + 1) * ld, ones.data(), sizeof(double), hipMemcpyHostToDevice, c->stream));
        SCHK(stream_wait(c));
    }
    for (int j = n - 1; j >= 0; --j) {
        // rebuild vpad from the stored reflector: vpad[1] = 1, vpad[2..] = At[j][j+1..]
        const int len = m - j;
        HIPCHK(s_memset0(c, vpad, 2 * sizeof(double)));
        if (len > 1)
            HIPCHK(s_memcpy(c, vpad + 2, At + (size_t)j * ld + j + 1, (size_t)(len - 1) * sizeof(double), hipMemcpyDeviceToDevice));
        HIPCHK(s_memcpy(c, vpad + 1, Qt + (size_t)(n + 1) * ld, sizeof(double), hipMemcpyDeviceToDevice));
        SCHK(apply_reflector(c, Qt, ld, j, n, j, m, vpad, taus + j, dots));
    }
    return download_panel(c, Qt, ld, m, n, Q);
}
