// Lab: what does an in-place read-modify-write stream sustain on this part, against a copy and a read-only stream?
// (the trailing rank-2k update of the tridiagonalisation is an in-place pass over an m x m block)
// hipcc -O3 --offload-arch=gfx950 tools/lab/rmw_lab.hip -o tools/lab/bin/rmw_lab
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void rmw(double2* __restrict__ a, size_t n2, double s) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        double2 v = a[i]; v.x = v.x * s + 1e-9; v.y = v.y * s + 1e-9; a[i] = v;
    }
}
__global__ __launch_bounds__(256) void rmw_tile(double* __restrict__ a, int m, int ld, double s) {
    // 32 x 128 tiles like rank2k_stream_kernel: thread = (row r in 0..31 step 4, 16-byte column pair)
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 128 + 2 * tx, r0 = blockIdx.y * 32;
    if (c0 >= m) return;
    for (int r = ty; r < 32; r += 4) {
        if (r0 + r >= m) break;
        double2* p = reinterpret_cast<double2*>(a + (size_t)(r0 + r) * ld + c0);
        double2 v = *p; v.x = v.x * s + 1e-9; v.y = v.y * s + 1e-9; *p = v;
    }
}
__global__ __launch_bounds__(256) void copyk(const double2* __restrict__ a, double2* __restrict__ b, size_t n2) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void readk(const double2* __restrict__ a, size_t n2, double* out) {
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) { double2 v = a[i]; s += v.x + v.y; }
    if (s == 1.2345e300) out[0] = s;
}
int main() {
    for (int m : {3072, 2048, 1024, 12288}) {
        const int ld = m;
        const size_t n = (size_t)m * ld, n2 = n / 2;
        double *a, *b, *o;
        hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&o, 8);
        hipMemset(a, 0, n * 8); hipMemset(b, 0, n * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto time = [&](auto launch, const char* name, double bytes) {
            for (int i = 0; i < 3; ++i) launch();
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            const int reps = 20;
            for (int i = 0; i < reps; ++i) launch();
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("m=%5d %-10s %7.1f us  %6.2f TB/s (read + written)\n", m, name, 1e3 * ms / reps, bytes / (ms / reps * 1e-3) / 1e12);
        };
        const int grid = 256 * 8;
        time([&] { hipLaunchKernelGGL(rmw, dim3(grid), dim3(256), 0, 0, (double2*)a, n2, 1.0000001); }, "rmw", 16.0 * n);
        time([&] { hipLaunchKernelGGL(rmw_tile, dim3((m + 127) / 128, (m + 31) / 32), dim3(256), 0, 0, a, m, ld, 1.0000001); }, "rmw_tile", 16.0 * n);
        time([&] { hipLaunchKernelGGL(copyk, dim3(grid), dim3(256), 0, 0, (const double2*)a, (double2*)b, n2); }, "copy", 16.0 * n);
        time([&] { hipLaunchKernelGGL(readk, dim3(grid), dim3(256), 0, 0, (const double2*)a, n2, o); }, "read", 8.0 * n);
        hipFree(a); hipFree(b); hipFree(o);
    }
    return 0;
}
