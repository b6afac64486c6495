// Broadcast hand-off lab (round 5): a few PRODUCER workgroups of a launch write a vector, a thousand CONSUMER workgroups of
// the SAME launch wait for it and read all of it — the shape a column launch of the tridiagonalisation would have if the row
// kernel (12 workgroups forming the updated row u) ran inside the matvec launch (1,500 workgroups that need all of u) while
// the consumers' matrix loads are already in flight.  Per launch: producers (lowest block ids: dispatched first) write their
// 256-entry slices of u = f(launch number), publish, add to a counter; consumers spin on the counter (bounded), read all of
// u and count stale entries.  Variants of the protocol:
//   0  no hand-off at all (u written by the previous launch): the floor of the launch itself
//   1  plain stores + agent release fence | agent acquire fence + plain loads        (the textbook protocol)
//   2  plain stores + agent release fence | NO acquire fence, u read with agent-scope atomic loads (sc1: past the L2)
//   3  agent-scope atomic stores, no fence, s_waitcnt only | agent-scope atomic loads
// Reported: mean launch time over 2000 dependent launches and stale entries.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/bcast_lab tools/lab/bcast_lab.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Ctl { unsigned long long counter; unsigned long long pad[15]; int abort, stale, pad2[30]; };

template <int MODE>
__global__ __launch_bounds__(256) void col(Ctl* ctl, double* u, const double* uprev, int m, int nprod, int it, double* sink) {
    const int tid = threadIdx.x, b = blockIdx.x;
    if (b < nprod) {
        if (MODE == 0) return;
        const int c = b * 256 + tid;
        if (c < m) {
            const double v = (double)it + 1e-3 * c;
            if (MODE == 3) __hip_atomic_store(&u[c], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else u[c] = v;
        }
        if (MODE == 3) __builtin_amdgcn_s_waitcnt(0);
        else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(&ctl->counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const double* src = MODE == 0 ? uprev : u;
    if (MODE != 0) {
        const unsigned long long want = (unsigned long long)nprod * it;
        bool ok = false;
        for (int spin = 0; spin < (1 << 20); ++spin) {
            if (__hip_atomic_load(&ctl->counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) { ok = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok) { if (tid == 0) ctl->abort = 1; return; }
        if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    double acc = 0.0;
    int bad = 0;
    for (int c = tid; c < m; c += 256) {
        double v;
        if (MODE >= 2) v = __hip_atomic_load(&src[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else v = src[c];
        const double want = (double)(MODE == 0 ? it - 1 : it) + 1e-3 * c;
        bad += (MODE != 0 && v != want) ? 1 : 0;
        acc += v;
    }
    if (bad) atomicAdd(&ctl->stale, bad);
    if (acc == 12345.678) sink[b] = acc;
}

template <int MODE>
int run(Ctl* ctl, double* u0, double* u1, double* sink, int m, int ncons, const char* name) {
    const int nprod = (m + 255) / 256, rounds = 2000;
    CHK(hipMemset(ctl, 0, sizeof(Ctl)));
    CHK(hipMemset(u0, 0, m * sizeof(double)));
    CHK(hipMemset(u1, 0, m * sizeof(double)));
    CHK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    for (int it = 1; it <= rounds; ++it)
        hipLaunchKernelGGL(col<MODE>, dim3(nprod + ncons), dim3(256), 0, 0, ctl, (it & 1) ? u1 : u0, (it & 1) ? u0 : u1, m, nprod, it, sink);
    CHK(hipDeviceSynchronize());
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    Ctl h;
    CHK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
    printf("m %5d, %2d producers + %4d consumers, %-62s: %7.2f us per launch, %d stale entries%s\n", m, nprod, ncons, name,
           1e6 * dt / rounds, h.stale, h.abort ? "  (ABORTED: a hand-off never arrived)" : "");
    return 0;
}

int main() {
    Ctl* ctl; double *u0, *u1, *sink;
    CHK(hipMalloc(&ctl, sizeof(Ctl)));
    CHK(hipMalloc(&u0, 1 << 16)); CHK(hipMalloc(&u1, 1 << 16)); CHK(hipMalloc(&sink, 1 << 16));
    for (int m : {3072, 1024})
        for (int ncons : {1536, 512}) {
            if (run<0>(ctl, u0, u1, sink, m, ncons, "no hand-off (vector from the previous launch)")) return 1;
            if (run<1>(ctl, u0, u1, sink, m, ncons, "release fence | acquire fence + plain loads")) return 1;
            if (run<2>(ctl, u0, u1, sink, m, ncons, "release fence | atomic (sc1) loads, no acquire fence")) return 1;
            if (run<3>(ctl, u0, u1, sink, m, ncons, "atomic stores + s_waitcnt | atomic loads, no fences")) return 1;
        }
    return 0;
}
