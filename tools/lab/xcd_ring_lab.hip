// One-XCD cooperative lab (round 5): how much does a COLUMN of a tridiagonalisation cost when the trailing block lives in
// the LDS of P workgroups of ONE XCD and the workgroups exchange two m-vectors per column through memory instead of ending a
// launch?  (The verdict's route (b): "grow trd_tail_lds_kernel past 128 columns, multi-workgroup inside one XCD".)
// A launch of 8 P workgroups; those with id % 8 != 0 leave at once (workgroups go to the XCDs round-robin), the P others
// each own S = m / P rows of an m x m block in LDS (S m 8 bytes <= 128 KiB) and loop over NCOL columns:
//   (1) all-gather of v  (each workgroup publishes its S entries + a sequence flag, everybody polls the P flags and reads m)
//   (2) y = A_loc v  over the LDS block (S rows)
//   (3) all-gather of y
//   (4) rank-2 update of the LDS block with (v, y)
// Stores / loads of the exchange are agent-scope atomics (no fences: profiles/r05_trd_handoff.md).  Reported: microseconds
// per column, the XCC ids the active workgroups ran on, stale reads.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_ring_lab tools/lab/xcd_ring_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Ctl { int abort, stale, xcc_mask, pad; };

__device__ __forceinline__ void publish(double* buf, unsigned long long* flags, int wg, int S, const double* mine, unsigned long long seq) {
    for (int i = threadIdx.x; i < S; i += blockDim.x) __hip_atomic_store(&buf[wg * S + i], mine[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&flags[wg * 16], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ bool collect(const double* buf, const unsigned long long* flags, int P, int m, double* dst, unsigned long long seq, Ctl* ctl) {
    __shared__ int okflag;
    if (threadIdx.x < 64) {
        bool ok = false;
        for (int spin = 0; spin < (1 << 18); ++spin) {
            const bool mine = threadIdx.x >= P || __hip_atomic_load(&flags[threadIdx.x * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= seq;
            if (__ballot(mine) == ~0ull) { ok = true; break; }
        }
        if (threadIdx.x == 0) okflag = ok;
    }
    __syncthreads();
    if (!okflag) { if (threadIdx.x == 0) ctl->abort = 1; return false; }
    for (int i = threadIdx.x; i < m; i += blockDim.x) dst[i] = __hip_atomic_load(&buf[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return true;
}

template <int NT>
__global__ __launch_bounds__(NT) void ring(Ctl* ctl, double* vbuf0, double* vbuf1, double* ybuf0, double* ybuf1, unsigned long long* vflags, unsigned long long* yflags,
                                           int P, int m, int ncol, int compute) {
    extern __shared__ double lds[];                       // S x (m + 1) block, then v (m), y (m), mine (S)
    if (blockIdx.x % 8 != 0) return;
    const int wg = blockIdx.x / 8, S = m / P, tid = threadIdx.x;
    if (tid == 0) {
        int xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        atomicOr(&ctl->xcc_mask, 1 << (xcc & 15));
    }
    double* A = lds;
    double* v = lds + (size_t)S * (m + 1);
    double* y = v + m;
    double* mine = y + m;
    for (int e = tid; e < S * (m + 1); e += NT) A[e] = 1e-3 * ((e * 7 + wg) % 13);
    __syncthreads();
    for (int j = 1; j <= ncol; ++j) {
        double* vb = (j & 1) ? vbuf1 : vbuf0;
        double* yb = (j & 1) ? ybuf1 : ybuf0;
        // (1) v: S entries of this workgroup (a function of j so that stale data is detectable)
        for (int i = tid; i < S; i += NT) mine[i] = (double)j + 1e-3 * (wg * S + i);
        __syncthreads();
        publish(vb, vflags, wg, S, mine, (unsigned long long)j);
        if (!collect(vb, vflags, P, m, v, (unsigned long long)j, ctl)) return;
        int bad = 0;
        for (int i = tid; i < m; i += NT) bad += (v[i] != (double)j + 1e-3 * i);
        if (bad) atomicAdd(&ctl->stale, bad);
        // (2) y_loc = A_loc v: a wavefront per row
        if (compute) {
            const int lane = tid & 63, wave = tid >> 6;
            for (int r = wave; r < S; r += NT / 64) {
                double acc = 0.0;
                for (int c = lane; c < m; c += 64) acc += A[r * (m + 1) + c] * v[c];
                for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
                if (lane == 0) mine[r] = acc;
            }
        }
        __syncthreads();
        // (3) y
        publish(yb, yflags, wg, S, mine, (unsigned long long)j);
        if (!collect(yb, yflags, P, m, y, (unsigned long long)j, ctl)) return;
        // (4) rank-2 update of the local rows
        if (compute) {
            for (int e = tid; e < S * m; e += NT) {
                const int r = e / m, c = e - r * m;
                A[r * (m + 1) + c] -= 1e-9 * (v[wg * S + r] * y[c] + y[wg * S + r] * v[c]);
            }
        }
        __syncthreads();
    }
    if (A[tid] == 12345.6789) vbuf0[0] = A[tid];
}

int main() {
    Ctl* ctl; double *vb0, *vb1, *yb0, *yb1; unsigned long long *vf, *yf;
    CHK(hipMalloc(&ctl, sizeof(Ctl)));
    CHK(hipMalloc(&vb0, 1 << 16)); CHK(hipMalloc(&vb1, 1 << 16)); CHK(hipMalloc(&yb0, 1 << 16)); CHK(hipMalloc(&yb1, 1 << 16));
    CHK(hipMalloc(&vf, 64 * 16 * 8)); CHK(hipMalloc(&yf, 64 * 16 * 8));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int ncol = 1000;
    for (int compute : {0, 1})
        for (int cfg = 0; cfg < 5; ++cfg) {
            const int Ps[5] = {2, 4, 8, 16, 32}, ms[5] = {128, 256, 384, 512, 704};
            const int P = Ps[cfg], m = ms[cfg], S = m / P;
            const size_t lds = ((size_t)S * (m + 1) + 2 * m + S) * sizeof(double);
            CHK(hipFuncSetAttribute((const void*)ring<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            for (int rep = 0; rep < 2; ++rep) {
                CHK(hipMemset(ctl, 0, sizeof(Ctl))); CHK(hipMemset(vf, 0, 64 * 16 * 8)); CHK(hipMemset(yf, 0, 64 * 16 * 8));
                CHK(hipEventRecord(e0));
                hipLaunchKernelGGL(ring<512>, dim3(8 * P), dim3(512), lds, 0, ctl, vb0, vb1, yb0, yb1, vf, yf, P, m, ncol, compute);
                CHK(hipEventRecord(e1));
                CHK(hipEventSynchronize(e1));
                float ms_ = 0; CHK(hipEventElapsedTime(&ms_, e0, e1));
                Ctl h; CHK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
                if (rep == 1)
                    printf("P %2d workgroups, block %3d x %3d (%5.1f KiB LDS each), %s: %6.2f us per column, XCC mask 0x%x, %d stale%s\n", P, m, m, lds / 1024.0,
                           compute ? "exchange + LDS matvec + rank-2 update" : "two all-gathers only               ", 1e3 * ms_ / ncol, h.xcc_mask, h.stale,
                           h.abort ? "  (ABORTED)" : "");
            }
        }
    return 0;
}
