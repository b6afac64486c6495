// gemv_lab.hip — standalone bandwidth experiments for the row-panel matvec (not part of the library).
// hipcc --offload-arch=gfx950 -O3 -o gemv_lab gemv_lab.hip && ./gemv_lab [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// variant A: wave owns RW rows, x in LDS (tile TC), unroll U
template <int NRHS, int RW, int U, int TC>
__global__ __launch_bounds__(256) void gemv_a(const double* __restrict__ A, int rows, int cols, int lda,
                                              const double* __restrict__ X, int ldx, double* __restrict__ Y, int ldy) {
    extern __shared__ double xs[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * 4 + wave) * RW;
    const double* arow[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) { int rr = row0 + r; if (rr > rows - 1) rr = rows - 1; arow[r] = A + (size_t)rr * lda; }
    double acc[RW][NRHS];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int h = 0; h < NRHS; ++h) acc[r][h] = 0.0;
    for (int c0 = 0; c0 < cols; c0 += TC) {
        const int tc = (cols - c0 < TC) ? (cols - c0) : TC;
#pragma unroll
        for (int h = 0; h < NRHS; ++h)
            for (int j = threadIdx.x; j < tc; j += 256) xs[h * TC + j] = X[(size_t)h * ldx + c0 + j];
        __syncthreads();
        const int tc2 = tc >> 1;
        const double2* xs2 = reinterpret_cast<const double2*>(xs);
#pragma unroll U
        for (int j = lane; j < tc2; j += 64) {
            double2 av[RW];
#pragma unroll
            for (int r = 0; r < RW; ++r) av[r] = *reinterpret_cast<const double2*>(arow[r] + c0 + 2 * j);
#pragma unroll
            for (int h = 0; h < NRHS; ++h) {
                const double2 xv = xs2[h * (TC / 2) + j];
#pragma unroll
                for (int r = 0; r < RW; ++r) acc[r][h] += av[r].x * xv.x + av[r].y * xv.y;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int h = 0; h < NRHS; ++h) acc[r][h] = wave_sum(acc[r][h]);
    if (lane == 0)
#pragma unroll
        for (int r = 0; r < RW; ++r)
            if (row0 + r < rows)
#pragma unroll
                for (int h = 0; h < NRHS; ++h) Y[(size_t)h * ldy + row0 + r] = acc[r][h];
}

// variant B: x straight from global memory (L1/L2), explicit register blocking of U loads
template <int NRHS, int RW, int U>
__global__ __launch_bounds__(256) void gemv_b(const double* __restrict__ A, int rows, int cols, int lda,
                                              const double* __restrict__ X, int ldx, double* __restrict__ Y, int ldy) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * 4 + wave) * RW;
    const double2* arow[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) { int rr = row0 + r; if (rr > rows - 1) rr = rows - 1; arow[r] = reinterpret_cast<const double2*>(A + (size_t)rr * lda); }
    double acc[RW][NRHS];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int h = 0; h < NRHS; ++h) acc[r][h] = 0.0;
    const int n2 = cols >> 1;
    for (int j0 = lane; j0 < n2; j0 += 64 * U) {
        double2 av[U][RW];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * 64;
#pragma unroll
            for (int r = 0; r < RW; ++r) av[u][r] = (j < n2) ? arow[r][j] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * 64;
            if (j < n2) {
#pragma unroll
                for (int h = 0; h < NRHS; ++h) {
                    const double2 xv = reinterpret_cast<const double2*>(X + (size_t)h * ldx)[j];
#pragma unroll
                    for (int r = 0; r < RW; ++r) acc[r][h] += av[u][r].x * xv.x + av[u][r].y * xv.y;
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int h = 0; h < NRHS; ++h) acc[r][h] = wave_sum(acc[r][h]);
    if (lane == 0)
#pragma unroll
        for (int r = 0; r < RW; ++r)
            if (row0 + r < rows)
#pragma unroll
                for (int h = 0; h < NRHS; ++h) Y[(size_t)h * ldy + row0 + r] = acc[r][h];
}

// variant C: like B but the whole 256-thread block works on RB rows at once (each wave a quarter of the row)
template <int NRHS, int RB, int U>
__global__ __launch_bounds__(256) void gemv_c(const double* __restrict__ A, int rows, int cols, int lda,
                                              const double* __restrict__ X, int ldx, double* __restrict__ Y, int ldy) {
    __shared__ double red[4][RB][NRHS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = blockIdx.x * RB;
    const double2* arow[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) { int rr = row0 + r; if (rr > rows - 1) rr = rows - 1; arow[r] = reinterpret_cast<const double2*>(A + (size_t)rr * lda); }
    double acc[RB][NRHS];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int h = 0; h < NRHS; ++h) acc[r][h] = 0.0;
    const int n2 = cols >> 1;
    for (int j0 = threadIdx.x; j0 < n2; j0 += 256 * U) {
        double2 av[U][RB];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * 256;
#pragma unroll
            for (int r = 0; r < RB; ++r) av[u][r] = (j < n2) ? arow[r][j] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * 256;
            if (j < n2) {
#pragma unroll
                for (int h = 0; h < NRHS; ++h) {
                    const double2 xv = reinterpret_cast<const double2*>(X + (size_t)h * ldx)[j];
#pragma unroll
                    for (int r = 0; r < RB; ++r) acc[r][h] += av[u][r].x * xv.x + av[u][r].y * xv.y;
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int h = 0; h < NRHS; ++h) { double v = wave_sum(acc[r][h]); if (lane == 0) red[wave][r][h] = v; }
    __syncthreads();
    if (threadIdx.x < RB * NRHS) {
        const int r = threadIdx.x / NRHS, h = threadIdx.x % NRHS;
        if (row0 + r < rows) Y[(size_t)h * ldy + row0 + r] = red[0][r][h] + red[1][r][h] + red[2][r][h] + red[3][r][h];
    }
}

template <class F> float timeit(F f, int reps = 50) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) f();
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 3072;
    const int ld = (n + 7) / 8 * 8;
    double *A, *X, *Y;
    CK(hipMalloc(&A, (size_t)(n + 2) * ld * 8)); CK(hipMalloc(&X, (size_t)8 * ld * 8)); CK(hipMalloc(&Y, (size_t)8 * ld * 8));
    std::vector<double> h((size_t)n * ld);
    for (auto& v : h) v = (double)rand() / RAND_MAX - 0.5;
    CK(hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(X, h.data(), (size_t)8 * ld * 8, hipMemcpyHostToDevice));
    const double bytes = 8.0 * n * n;
#define RUN_A(NR, RW, U, TC) { float ms = timeit([&] { hipLaunchKernelGGL((gemv_a<NR, RW, U, TC>), dim3((n + 4 * RW - 1) / (4 * RW)), dim3(256), NR * TC * 8, 0, A, n, n, ld, X, ld, Y, ld); }); \
    printf("A  nrhs=%d rw=%d u=%-2d tc=%-4d : %7.2f us  %7.1f GB/s\n", NR, RW, U, TC, ms * 1e3, bytes / ms / 1e6); }
#define RUN_B(NR, RW, U) { float ms = timeit([&] { hipLaunchKernelGGL((gemv_b<NR, RW, U>), dim3((n + 4 * RW - 1) / (4 * RW)), dim3(256), 0, 0, A, n, n, ld, X, ld, Y, ld); }); \
    printf("B  nrhs=%d rw=%d u=%-2d         : %7.2f us  %7.1f GB/s\n", NR, RW, U, ms * 1e3, bytes / ms / 1e6); }
#define RUN_C(NR, RB, U) { float ms = timeit([&] { hipLaunchKernelGGL((gemv_c<NR, RB, U>), dim3((n + RB - 1) / RB), dim3(256), 0, 0, A, n, n, ld, X, ld, Y, ld); }); \
    printf("C  nrhs=%d rb=%d u=%-2d         : %7.2f us  %7.1f GB/s\n", NR, RB, U, ms * 1e3, bytes / ms / 1e6); }
    printf("n = %d  (%.1f MB)\n", n, bytes / 1e6);
    RUN_A(1, 1, 4, 2048) RUN_A(1, 2, 4, 2048) RUN_A(1, 1, 8, 2048) RUN_A(1, 2, 8, 2048) RUN_A(1, 1, 12, 4096) RUN_A(1, 2, 12, 4096) RUN_A(1, 4, 8, 4096)
    RUN_A(2, 1, 4, 2048) RUN_A(2, 2, 4, 2048) RUN_A(2, 1, 8, 2048) RUN_A(2, 2, 8, 2048) RUN_A(2, 1, 12, 2048)
    RUN_B(1, 1, 4) RUN_B(1, 1, 8) RUN_B(1, 2, 4) RUN_B(1, 2, 8) RUN_B(1, 1, 12) RUN_B(1, 4, 4)
    RUN_B(2, 1, 4) RUN_B(2, 1, 8) RUN_B(2, 2, 4) RUN_B(2, 2, 8)
    RUN_C(1, 1, 2) RUN_C(1, 1, 3) RUN_C(1, 2, 2) RUN_C(1, 2, 3) RUN_C(1, 4, 2) RUN_C(1, 4, 3) RUN_C(1, 2, 6)
    RUN_C(2, 1, 3) RUN_C(2, 2, 2) RUN_C(2, 2, 3) RUN_C(2, 4, 2) RUN_C(2, 4, 3)
    // plain copy-rate probe: read-only sum over the matrix
    return 0;
}
