// Barrier lab: what a barrier among ALL workgroups of a persistent kernel costs on MI355X (one workgroup per CU, 256 of
// them) — flat (every workgroup adds to one counter) and hierarchical (one counter per XCD, the last arrival of an XCD adds
// to a global one, the last of those publishes the generation).  Counters only grow (no reset race); every spin is bounded
// and watches an abort flag, so the kernel ends even if the workgroups are not all resident.
//   hipcc --offload-arch=gfx950 -O3 -o tools/lab/bin/barrier_lab tools/lab/barrier_lab.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Bar {
    unsigned long long flags[256][16];   // mode 2: one cache line per workgroup, plain release stores, no atomics
    unsigned long long xcd[8][16];       // one cache line (128 B) per XCD counter
    unsigned long long global[16];
    unsigned long long gen[16];
    int abort, pad[31];
};

__device__ bool spin_until(unsigned long long* p, unsigned long long want, int* abort_flag) {
    for (int spin = 0; spin < (1 << 22); ++spin) {
        if (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
        if ((spin & 1023) == 1023 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
    }
    __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}

// mode 0: flat; mode 1: hierarchical by blockIdx.x % 8; mode 2: flag array (no atomics); `work` doubles are written by every workgroup before each barrier
// and one word of a NEIGHBOUR's slice is read after it (a data dependency through the barrier: a value older than this
// round's would be a stale read)
__global__ __launch_bounds__(256) void barrier_loop(Bar* b, int mode, int rounds, double* data, int work, int* bad) {
    const int nwg = gridDim.x, me = blockIdx.x, tid = threadIdx.x;
    const int per_xcd = nwg / 8;
    for (int r = 1; r <= rounds; ++r) {
        for (int w = tid; w < work; w += 256) data[(size_t)me * work + w] = (double)r;
        if (mode != 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        bool ok = true;
        if (mode == 2) {
            // flag array: every workgroup publishes its round number in its own line; wavefront 0 of workgroup 0 polls all
            // of them (four per lane) and publishes the generation; nobody performs a read-modify-write
            if (tid == 0) __hip_atomic_store(&b->flags[me][0], (unsigned long long)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (me == 0 && tid < 64) {
                bool all = false;
                for (int spin = 0; spin < (1 << 20) && !all; ++spin) {
                    bool mine = true;
                    for (int w = tid; w < nwg; w += 64)
                        mine = mine && (__hip_atomic_load(&b->flags[w][0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned long long)r);
                    all = __all(mine);
                }
                if (tid == 0) {
                    if (all) __hip_atomic_store(&b->gen[0], (unsigned long long)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    else __hip_atomic_store(&b->abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (tid == 0) ok = spin_until(&b->gen[0], r, &b->abort);
        } else if (mode == 3) {
            // arrivals alone: relaxed read-modify-writes on one counter, relaxed polling, no fence anywhere — what the
            // synchronisation costs when nobody writes back or invalidates an L2 (NOT a barrier that carries data)
            if (tid == 0) {
                const unsigned long long old = __hip_atomic_fetch_add(&b->global[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == (unsigned long long)nwg * r - 1) __hip_atomic_store(&b->gen[0], (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int spin = 0; spin < (1 << 22); ++spin) {
                    if (__hip_atomic_load(&b->gen[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned long long)r) break;
                    if (spin == (1 << 22) - 1) { __hip_atomic_store(&b->abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; }
                }
            }
        } else if (tid == 0) {
            if (mode == 0) {
                const unsigned long long old = __hip_atomic_fetch_add(&b->global[0], 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                if (old == (unsigned long long)nwg * r - 1) __hip_atomic_store(&b->gen[0], (unsigned long long)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                const int x = me & 7;
                const unsigned long long old = __hip_atomic_fetch_add(&b->xcd[x][0], 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                if (old == (unsigned long long)per_xcd * r - 1) {
                    const unsigned long long o2 = __hip_atomic_fetch_add(&b->global[0], 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                    if (o2 == 8ull * r - 1) __hip_atomic_store(&b->gen[0], (unsigned long long)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            ok = spin_until(&b->gen[0], r, &b->abort);
        }
        __syncthreads();
        ok = __shfl(ok ? 1 : 0, 0) != 0;       // (wave 0 holds the verdict; the others read it through LDS below)
        __shared__ int verdict;
        if (tid == 0) verdict = ok ? 1 : 0;
        __syncthreads();
        if (!verdict) return;
        if (mode != 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (work > 0 && tid == 0) {
            const int nb = (me + 37) % nwg;
            // the neighbour may already be writing round r + 1 (there is no second barrier): only an OLDER value is stale
            if (data[(size_t)nb * work + (r % work)] < (double)r) atomicAdd(bad, 1);
        }
        __syncthreads();
    }
}

int main() {
    Bar* b;
    double* data;
    int* bad;
    CHK(hipMalloc(&b, sizeof(Bar)));
    CHK(hipMalloc(&data, (size_t)256 * 512 * sizeof(double)));
    CHK(hipMalloc(&bad, sizeof(int)));
    const int rounds = 2000;
    for (int nwg : {256, 64, 16})
        for (int mode = 0; mode < 4; ++mode)
            for (int work : {0, 64, 512}) {
                CHK(hipMemset(b, 0, sizeof(Bar)));
                CHK(hipMemset(bad, 0, sizeof(int)));
                CHK(hipDeviceSynchronize());
                const auto t0 = std::chrono::steady_clock::now();
                hipLaunchKernelGGL(barrier_loop, dim3(nwg), dim3(256), 0, 0, b, mode, rounds, data, work, bad);
                CHK(hipDeviceSynchronize());
                const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                Bar h;
                int hb;
                CHK(hipMemcpy(&h, b, sizeof(Bar), hipMemcpyDeviceToHost));
                CHK(hipMemcpy(&hb, bad, sizeof(int), hipMemcpyDeviceToHost));
                printf("%3d workgroups, %-12s, %4d B written per workgroup and round: %8.1f ns per barrier round, %d stale reads%s\n", nwg,
                       mode == 3 ? "no fences" : mode == 2 ? "flag array" : mode ? "hierarchical" : "flat", work * 8, 1e9 * dt / rounds, hb, h.abort ? "  [ABORTED]" : "");
            }
    return 0;
}
