// How much LDS may one workgroup of a plain hipLaunchKernelGGL launch ask for on gfx950 (static + dynamic), with and
// without hipFuncSetAttribute(hipFuncAttributeMaxDynamicSharedMemorySize)?   hipcc --offload-arch=gfx950 lds_lab.hip -o lds_lab
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void k(double* out, int ndyn) {
    __shared__ double st[4096];                       // 32 KB static
    extern __shared__ double dyn[];
    for (int i = threadIdx.x; i < 4096; i += 1024) st[i] = i;
    for (int i = threadIdx.x; i < ndyn; i += 1024) dyn[i] = 2.0 * i;
    __syncthreads();
    double s = 0.0;
    for (int i = threadIdx.x; i < 4096; i += 1024) s += st[4095 - i];
    for (int i = threadIdx.x; i < ndyn; i += 1024) s += dyn[ndyn - 1 - i];
    atomicAdd(out, s);
}
int main() {
    double* d; hipMalloc(&d, 8);
    for (int attr = 0; attr < 2; ++attr)
        for (int kb : {16, 31, 33, 64, 96, 120, 128}) {
            const int ndyn = kb * 128;
            if (attr) {
                hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
                if (e != hipSuccess) { printf("attr %d KB: %s\n", kb, hipGetErrorString(e)); (void)hipGetLastError(); }
            }
            hipMemset(d, 0, 8);
            hipLaunchKernelGGL(k, dim3(1), dim3(1024), (size_t)kb * 1024, 0, d, ndyn);
            hipError_t e1 = hipGetLastError();
            hipError_t e2 = hipDeviceSynchronize();
            double h = -1; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
            double want = 4095.0 * 4096 / 2 + 2.0 * (double)(ndyn - 1) * ndyn / 2;
            printf("attribute %d, 32 KB static + %3d KB dynamic: launch %s, sync %s, result %s\n", attr, kb, hipGetErrorString(e1), hipGetErrorString(e2), h == want ? "ok" : "WRONG");
            (void)hipGetLastError();
        }
    return 0;
}
