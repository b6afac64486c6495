// Hand-off lab: what does it cost to pass freshly written data from one workgroup to another INSIDE a kernel on MI355X —
// the primitive a persistent (one-launch) tridiagonalisation or bulge chase would be built from.
// Two workgroups of a 256-workgroup grid play ping-pong through flags in device memory: the producer writes a payload
// (plain stores), publishes a sequence number with an agent-scope RELEASE store, the consumer spins on an agent-scope
// ACQUIRE load, checks the payload, answers.  Reported: nanoseconds per round trip and payload mismatches, for partners
// on the same XCD (workgroup ids 0 and 8: ids are dealt to the 8 XCDs round-robin) and on different XCDs (0 and 1, 0 and 4),
// for payloads of 0, 64 B, 4 KiB and 64 KiB.  Every spin is bounded and both sides watch an abort flag: the kernel ends
// even if a hand-off never arrives.
//   hipcc --offload-arch=gfx950 -O3 -o tools/lab/bin/handoff_lab tools/lab/handoff_lab.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Shared {
    unsigned long long flagA, pad0[15], flagB, pad1[15];
    int abort, mismatches, pad2[30];
};

__device__ bool wait_for(unsigned long long* flag, unsigned long long want, int* abort_flag) {
    for (int spin = 0; spin < (1 << 22); ++spin) {
        if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
        if ((spin & 1023) == 1023 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
    }
    __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}

__global__ __launch_bounds__(64) void pingpong(Shared* sh, double* payload, int words, int rounds, int ida, int idb) {
    const int me = blockIdx.x;
    if (me != ida && me != idb) return;
    const int lane = threadIdx.x;
    for (int it = 1; it <= rounds; ++it) {
        if (me == ida) {
            for (int w = lane; w < words; w += 64) payload[w] = (double)it;          // plain stores
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                       // every lane publishes its own stores
            __syncthreads();
            if (lane == 0) __hip_atomic_store(&sh->flagA, (unsigned long long)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            bool ok = true;
            if (lane == 0) ok = wait_for(&sh->flagB, it, &sh->abort);
            ok = __shfl(ok ? 1 : 0, 0) != 0;
            if (!ok) return;
        } else {
            bool ok = true;
            if (lane == 0) ok = wait_for(&sh->flagA, it, &sh->abort);
            ok = __shfl(ok ? 1 : 0, 0) != 0;
            if (!ok) return;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                       // every lane reads behind the hand-off
            int bad = 0;
            for (int w = lane; w < words; w += 64) bad += (payload[w] != (double)it) ? 1 : 0;
            if (bad) atomicAdd(&sh->mismatches, bad);
            __syncthreads();
            if (lane == 0) __hip_atomic_store(&sh->flagB, (unsigned long long)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main() {
    Shared* sh;
    double* payload;
    CHK(hipMalloc(&sh, sizeof(Shared)));
    CHK(hipMalloc(&payload, 1 << 20));
    const int rounds = 2000;
    const int pairs[3][2] = {{0, 8}, {0, 1}, {0, 4}};
    const char* names[3] = {"same XCD (ids 0, 8)", "XCD 0 -> 1 (ids 0, 1)", "XCD 0 -> 4 (ids 0, 4)"};
    const int sizes[4] = {0, 8, 512, 8192};
    for (int p = 0; p < 3; ++p)
        for (int s = 0; s < 4; ++s) {
            CHK(hipMemset(sh, 0, sizeof(Shared)));
            CHK(hipMemset(payload, 0, 1 << 20));
            CHK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(pingpong, dim3(256), dim3(64), 0, 0, sh, payload, sizes[s], rounds, pairs[p][0], pairs[p][1]);
            CHK(hipDeviceSynchronize());
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            Shared h;
            CHK(hipMemcpy(&h, sh, sizeof(Shared), hipMemcpyDeviceToHost));
            printf("%-24s payload %6d B: %8.1f ns per round trip (two hand-offs), %d stale words in %d rounds%s\n", names[p],
                   sizes[s] * 8, 1e9 * dt / rounds, h.mismatches, rounds, h.abort ? "  [ABORTED: a hand-off never arrived]" : "");
        }
    return 0;
}
