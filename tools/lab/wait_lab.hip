// Lab: how should several host threads (one stream each) wait for their streams?
//   mode 0: hipStreamSynchronize                      (what the library does)
//   mode 1: a one-thread kernel writes a sequence number into pinned host memory; the host spins on it (no HIP call)
//   mode 2: as 1, and all HIP calls of all threads are serialised by ONE process-wide mutex (released while spinning)
//   mode 3: hipStreamWriteValue32 into pinned host memory instead of the flag kernel, host spins
// Each round = `work` launches of a small kernel (64 workgroups, ~3 us) + a 1 KB device-to-host copy + wait.
// hipcc -O3 --offload-arch=gfx950 tools/lab/wait_lab.hip -o /tmp/wait_lab -lpthread && /tmp/wait_lab
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <mutex>
#include <thread>
#include <vector>
#include <immintrin.h>

__global__ __launch_bounds__(256) void small(double* __restrict__ x, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = x[i] * 1.0000001 + 1e-9;
}
__global__ void flag(volatile unsigned* f, unsigned v) { if (threadIdx.x == 0) { __threadfence_system(); *f = v; } }

static std::mutex g_api;

static void run(int mode, int work, int rounds, int slot, double* secs, std::atomic<int>* go, int T) {
    hipSetDevice(0);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    double* d;
    hipMalloc(&d, 16384 * sizeof(double));
    hipMemset(d, 0, 16384 * sizeof(double));
    double* h;
    hipHostMalloc(&h, 1024, hipHostMallocDefault);
    unsigned* f;
    hipHostMalloc(&f, 64, hipHostMallocDefault);
    *f = 0;
    hipDeviceSynchronize();
    auto body = [&](int r) {
        for (int k = 0; k < work; ++k) hipLaunchKernelGGL(small, dim3(64), dim3(256), 0, s, d, 16384);
        hipMemcpyAsync(h, d, 1024, hipMemcpyDeviceToHost, s);
        if (mode == 1 || mode == 2) hipLaunchKernelGGL(flag, dim3(1), dim3(64), 0, s, f, (unsigned)(r + 1));
        if (mode == 3) hipStreamWriteValue32(s, f, (unsigned)(r + 1), 0);
    };
    auto wait = [&](int r) {
        if (mode == 0) { hipStreamSynchronize(s); return; }
        volatile unsigned* vf = f;
        while (*vf != (unsigned)(r + 1)) _mm_pause();
    };
    for (int r = 0; r < 20; ++r) {
        if (mode == 2) { std::lock_guard<std::mutex> lk(g_api); body(r); } else body(r);
        wait(r);
    }
    go->fetch_add(1);
    while (go->load() < T) _mm_pause();
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 20; r < 20 + rounds; ++r) {
        if (mode == 2) { std::lock_guard<std::mutex> lk(g_api); body(r); } else body(r);
        wait(r);
    }
    secs[slot] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    hipStreamSynchronize(s);
}

int main() {
    const int rounds = 2000;
    for (int work : {1, 4}) {
        for (int mode = 0; mode < 4; ++mode) {
            for (int T : {1, 2, 4, 8}) {
                std::vector<double> secs(T, 0.0);
                std::atomic<int> go{0};
                std::vector<std::thread> th;
                for (int k = 0; k < T; ++k) th.emplace_back(run, mode, work, rounds, k, secs.data(), &go, T);
                for (auto& t : th) t.join();
                double mx = 0;
                for (double v : secs) mx = v > mx ? v : mx;
                printf("work=%d mode=%d threads=%d: %8.0f rounds/s total, %6.1f us per round per thread\n", work, mode, T,
                       T * rounds / mx, 1e6 * mx / rounds);
                fflush(stdout);
            }
        }
    }
    return 0;
}
