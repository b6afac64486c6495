// Lab: does replaying a dependent chain of small kernels from a captured hipGraph shorten the per-launch
// floor of the tridiagonalisation loop (two dependent launches per column, eigh.hip)?
// hipcc -O3 --offload-arch=gfx950 tools/lab/graph_lab.hip -o /tmp/graph_lab && /tmp/graph_lab
#include <hip/hip_runtime.h>
#include <cstdio>

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
__global__ __launch_bounds__(256) void wide(const double2* __restrict__ a, size_t n2, const double* __restrict__ x,
                                            double* __restrict__ out) {
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        double2 v = a[i];
        s += v.x * x[0] + v.y;
    }
    if (s == 12345.678) out[0] = s;
}

int main() {
    double *a, *x, *y;
    const size_t bytes = (size_t)24 << 20;           // the mean trailing-matrix footprint at n = 3072
    hipMalloc(&a, bytes); hipMalloc(&x, 1 << 20); hipMalloc(&y, 1 << 20);
    hipMemset(a, 0, bytes); hipMemset(x, 0, 1 << 20); hipMemset(y, 0, 1 << 20);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int chain = 2048;
    for (int mode = 0; mode < 2; ++mode) {           // 0: tiny only, 1: tiny + 24 MB streaming kernel (the K1/K2 pair)
        auto enqueue = [&]() {
            for (int r = 0; r < chain; ++r) {
                hipLaunchKernelGGL(tiny, dim3(12), dim3(256), 0, st, (r & 1) ? x : y, (r & 1) ? y : x, 3072);
                if (mode) hipLaunchKernelGGL(wide, dim3(1536), dim3(256), 0, st, (const double2*)a, bytes / 16, x, y);
            }
        };
        float ms_stream = 0, ms_graph = 0;
        for (int w = 0; w < 2; ++w) {
            hipEventRecord(e0, st); enqueue(); hipEventRecord(e1, st); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms_stream, e0, e1);
        }
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        enqueue();
        hipStreamEndCapture(st, &g);
        if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("instantiate failed\n"); return 1; }
        for (int w = 0; w < 3; ++w) {
            hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms_graph, e0, e1);
        }
        const int launches = chain * (mode ? 2 : 1);
        printf("mode %d (%s): stream %.2f us per launch, graph replay %.2f us per launch (%d launches)\n", mode,
               mode ? "tiny + 24 MB read" : "tiny only", 1e3 * ms_stream / launches, 1e3 * ms_graph / launches, launches);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
