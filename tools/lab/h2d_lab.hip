// What does a small host-to-device transfer cost on this stack (round 5)?  N dependent operations on one stream, host time
// per call (the call returning) and stream time per operation (N calls + one synchronisation):
//   hipMemcpyAsync from pinned memory | a copy kernel reading the pinned memory directly (zero-copy) | the same alternating
//   with a consumer kernel (the pattern of the optimizer step: upload, launch, upload, launch, ...)
//   hipcc --offload-arch=gfx950 -O3 -o tools/lab/bin/h2d_lab tools/lab/h2d_lab.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void copy_kernel(double* __restrict__ dst, const double* __restrict__ src, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
__global__ void consume_kernel(double* __restrict__ acc, const double* __restrict__ x, int n) {
    const int i = threadIdx.x;
    if (i < n) acc[i] += x[i];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const int N = 2000;
    double *h, *d, *acc;
    CHK(hipHostMalloc(&h, 1 << 20, hipHostMallocDefault));
    CHK(hipMalloc(&d, 1 << 20)); CHK(hipMalloc(&acc, 1 << 16));
    CHK(hipMemset(acc, 0, 1 << 16));
    hipStream_t s; CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int n : {128, 512, 1024, 2048, 3072, 15360}) {
        for (int mode = 0; mode < 2; ++mode) {
            for (int i = 0; i < n; ++i) h[i] = i;
            CHK(hipStreamSynchronize(s));
            const double t0 = now();
            for (int it = 0; it < N; ++it) {
                if (mode == 0 || mode == 2) CHK(hipMemcpyAsync(d, h, n * sizeof(double), hipMemcpyHostToDevice, s));
                else hipLaunchKernelGGL(copy_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d, h, n);
                if (mode >= 2) hipLaunchKernelGGL(consume_kernel, dim3(1), dim3(256), 0, s, acc, d, n < 256 ? n : 256);
            }
            const double t1 = now();
            CHK(hipStreamSynchronize(s));
            const double t2 = now();
            const char* names[4] = {"hipMemcpyAsync (pinned)            ", "copy kernel (zero-copy read)       ", "hipMemcpyAsync + consumer kernel   ",
                                    "copy kernel + consumer kernel      "};
            printf("%5d doubles, %s: host %6.2f us per iteration, stream %6.2f us per iteration\n", n, names[mode], 1e6 * (t1 - t0) / N, 1e6 * (t2 - t0) / N);
        }
    }
    // device -> host: hipMemcpyAsync into pinned memory against a kernel storing into it (followed by the wait the host needs)
    for (int n : {1, 128, 1024, 3072}) {
        for (int mode = 0; mode < 2; ++mode) {
            CHK(hipStreamSynchronize(s));
            const double t0 = now();
            for (int it = 0; it < N; ++it) {
                if (mode == 0) CHK(hipMemcpyAsync(h, d, n * sizeof(double), hipMemcpyDeviceToHost, s));
                else hipLaunchKernelGGL(copy_kernel, dim3((n + 255) / 256), dim3(256), 0, s, h, d, n);
                CHK(hipStreamSynchronize(s));
            }
            const double t2 = now();
            printf("%5d doubles device -> host + wait, %s: %6.2f us per iteration\n", n, mode == 0 ? "hipMemcpyAsync into pinned memory" : "kernel storing into pinned memory ", 1e6 * (t2 - t0) / N);
        }
    }
    return 0;
}
