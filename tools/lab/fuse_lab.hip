// Lab: cost of "last-arriving workgroups do the dependent tail" inside one kernel, against a second launch.
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ double wsum(double v) {
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// phase A: every workgroup streams 2 rows of an m x m matrix against x (like trd_gemv) and writes y[row]
// phase B (tail): the last T workgroups to arrive each process a 256-column slice: out[c] = y[c] * 2 + 1
__global__ __launch_bounds__(256) void fused(const double2* __restrict__ A, int m, int ld2, const double* __restrict__ x,
                                            double* __restrict__ y, double* __restrict__ out, unsigned* counter, int T,
                                            int mode) {
    __shared__ double red[4][2];
    __shared__ unsigned tk;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = blockIdx.x * 2;
    double acc[2] = {0.0, 0.0};
    const int n2 = m >> 1;
    for (int j = threadIdx.x; j < n2; j += 256) {
        const double x0 = x[2 * j], x1 = x[2 * j + 1];
        for (int r = 0; r < 2; ++r) {
            const int rr = (row0 + r < m) ? row0 + r : m - 1;
            const double2 a = A[(size_t)rr * ld2 + j];
            acc[r] += a.x * x0 + a.y * x1;
        }
    }
    for (int r = 0; r < 2; ++r) { double v = wsum(acc[r]); if (lane == 0) red[wave][r] = v; }
    __syncthreads();
    if (threadIdx.x < 2 && row0 + (int)threadIdx.x < m)
        y[row0 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (mode == 0) return;
    // ---- release: make this workgroup's y visible, then take a ticket
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        tk = atomicAdd(counter, 1u);
    }
    __syncthreads();
    const unsigned nblk = gridDim.x;
    if (tk + (unsigned)T < nblk) return;
    const int slice = (int)(tk - (nblk - (unsigned)T));
    if (threadIdx.x == 0) {
        long spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nblk && spins < 2000000) { __builtin_amdgcn_s_sleep(1); ++spins; }
        __threadfence();
    }
    __syncthreads();
    const int c = slice * 256 + threadIdx.x;
    if (c < m) out[c] = y[c] * 2.0 + 1.0;
}
__global__ __launch_bounds__(256) void tail_only(const double* __restrict__ y, double* __restrict__ out, int m) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < m) out[c] = y[c] * 2.0 + 1.0;
}
int main() {
    const int M[] = {3072, 2048, 1024, 512};
    double *A, *x, *y, *out; unsigned* cnt;
    hipMalloc(&A, (size_t)3072 * 3072 * 8); hipMalloc(&x, 3072 * 8 * 2); hipMalloc(&y, 3072 * 8); hipMalloc(&out, 3072 * 8);
    hipMalloc(&cnt, 4096 * sizeof(unsigned));
    hipMemset(A, 0, (size_t)3072 * 3072 * 8); hipMemset(x, 0, 3072 * 16); hipMemset(y, 0, 3072 * 8);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int m : M) {
        const int nblk = (m + 1) / 2, T = (m + 255) / 256;
        const int reps = 1000;
        float ms2 = 0, ms1 = 0;
        for (int w = 0; w < 2; ++w) {
            hipEventRecord(e0, st);
            for (int r = 0; r < reps; ++r) {
                hipLaunchKernelGGL(fused, dim3(nblk), dim3(256), 0, st, (const double2*)A, m, m / 2, (r & 1) ? x : out, y, (r & 1) ? out : x, cnt, T, 0);
                hipLaunchKernelGGL(tail_only, dim3(T), dim3(256), 0, st, y, (r & 1) ? out : x, m);
            }
            hipEventRecord(e1, st); hipEventSynchronize(e1); hipEventElapsedTime(&ms2, e0, e1);
        }
        for (int w = 0; w < 2; ++w) {
            hipMemsetAsync(cnt, 0, 4096 * sizeof(unsigned), st);
            hipEventRecord(e0, st);
            for (int r = 0; r < reps; ++r)
                hipLaunchKernelGGL(fused, dim3(nblk), dim3(256), 0, st, (const double2*)A, m, m / 2, (r & 1) ? x : out, y, (r & 1) ? out : x, cnt + r, T, 1);
            hipEventRecord(e1, st); hipEventSynchronize(e1); hipEventElapsedTime(&ms1, e0, e1);
        }
        printf("m=%4d: two launches %.2f us/column, fused %.2f us/column\n", m, 1e3 * ms2 / reps, 1e3 * ms1 / reps);
    }
    return 0;
}
