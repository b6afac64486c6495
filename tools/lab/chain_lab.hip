// Lab: does the duration of a tiny dependent kernel depend on the footprint of the kernel before it?
// hipcc -O3 --offload-arch=gfx950 tools/lab/chain_lab.hip -o /tmp/chain_lab && rocprofv3 --kernel-trace --stats ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void stream_read(const double2* __restrict__ a, size_t n2, double* out, int nt) {
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        const double* ap = reinterpret_cast<const double*>(a + i);
        double vx, vy;
        if (nt) { vx = __builtin_nontemporal_load(ap); vy = __builtin_nontemporal_load(ap + 1); }
        else { double2 v = a[i]; vx = v.x; vy = v.y; }
        s += vx + vy;
    }
    if (s == 12345.678) out[0] = s;
}
__global__ __launch_bounds__(256) void tiny(const double* __restrict__ in, double* __restrict__ out, int n) {
    __shared__ double red[256];
    int i = blockIdx.x * 256 + threadIdx.x;
    double v = (i < n) ? in[i] : 0.0;
    red[threadIdx.x] = v * v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (i < n) out[i] = v + 1.0;
    if (threadIdx.x == 0) out[n + blockIdx.x] = red[0];
}
int main() {
    const size_t maxb = (size_t)75 << 20;
    double *a, *x, *y;
    hipMalloc(&a, maxb); hipMalloc(&x, 1 << 20); hipMalloc(&y, 1 << 20);
    hipMemset(a, 0, maxb); hipMemset(x, 0, 1 << 20); hipMemset(y, 0, 1 << 20);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t sizes[] = {(size_t)1 << 20, (size_t)8 << 20, (size_t)32 << 20, (size_t)75 << 20};
    for (int nt = 0; nt < 2; ++nt)
    for (size_t sz : sizes) {
        const size_t n2 = sz / 16;
        const int reps = 400;
        for (int w = 0; w < 2; ++w) {
            hipEventRecord(e0, st);
            for (int r = 0; r < reps; ++r) {
                hipLaunchKernelGGL(stream_read, dim3(1536), dim3(256), 0, st, (const double2*)a, n2, y, nt);
                hipLaunchKernelGGL(tiny, dim3(12), dim3(256), 0, st, (r & 1) ? x : y, (r & 1) ? y : x, 3072);
            }
            hipEventRecord(e1, st);
            hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("nt=%d footprint %3zu MB: pair %.2f us (stream at %.2f TB/s would be %.2f us)\n", nt, sz >> 20, 1e3 * ms / reps,
               5.4, sz / 5.4e6);
    }
    // tiny alone
    {
        const int reps = 2000;
        hipEventRecord(e0, st);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(tiny, dim3(12), dim3(256), 0, st, (r & 1) ? x : y, (r & 1) ? y : x, 3072);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("tiny alone: %.2f us per launch\n", 1e3 * ms / reps);
    }
    return 0;
}
