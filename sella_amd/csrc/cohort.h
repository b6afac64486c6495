// cohort.h — the replica dimension of the ensemble path ("batched kernels over the local replicas").
//
// The reference's ensemble members are independent `Sella` objects (sella/optimize/optimize.py:42-81, 359-440): nothing
// couples them, so on one GPU they were host threads with a context and a stream each — and sat at the runtime's launch
// rate (≈ 63 launches per member step, 224 k launches/s across 8 threads).  A COHORT advances W members in lockstep
// instead, with ONE launch per kernel of the step for all of them:
//
//   * every member is the unchanged host code of a search (search.hip → optstep / lrstep / stepper / davidson / emt ...)
//     running on a FIBER of the one issuing thread; where that code launches a batchable kernel it parks with the launch
//     it wants (kernel body, virtual grid, argument pack), where it waits for the stream it parks at the wait;
//   * once every member is parked the scheduler merges the parked launches kernel by kernel: `batched_kernel<Body>` takes
//     the members' argument packs as ONE by-value descriptor array in its kernel arguments, `blockIdx.z` is the member,
//     and every workgroup runs the member's own body on the member's own virtual grid (workgroups beyond it return at
//     once) — the per-problem descriptor pattern of the divide & conquer merges (eigh.hip, MergeDev).  Members that took
//     another branch (a further round of the root search, a deflation) simply form a group of their own;
//   * the members' waits become one stream synchronisation per phase; barriers at the step boundaries let members that
//     needed fewer rounds idle until the others arrive, so the cohort reconverges.
//
// A kernel body is a `__device__` function whose first parameter is the virtual block index / grid `VB`; the plain
// `__global__` kernel of the same name calls it with the hardware's indices, so one source is compiled for both forms and
// the per-member arithmetic is the same instruction sequence — results are bit-identical to a search run on its own
// (tests/test_library_search.py, tests/test_multi.py).
#pragma once
#include <hip/hip_runtime.h>

#include <string.h>

struct sella_ctx;

namespace sella {

struct VB {
    unsigned x, y, gx, gy;      // block index and grid extent as the kernel body sees them
};
__device__ __forceinline__ VB vb_hw() { return VB{(unsigned)blockIdx.x, (unsigned)blockIdx.y, (unsigned)gridDim.x, (unsigned)gridDim.y}; }

constexpr int COHORT_MAX = 16;                 // members per cohort (and per batched launch)
constexpr size_t COHORT_PACK_BYTES = 640;      // largest argument pack of a batchable kernel

template <class... A> struct Pack;
template <> struct Pack<> {};
template <class H, class... T> struct Pack<H, T...> {
    H h;
    Pack<T...> t;
};
template <class... A> struct PackMaker;
template <> struct PackMaker<> {
    static Pack<> make() { return {}; }
};
template <class H, class... T> struct PackMaker<H, T...> {
    template <class X, class... Y> static Pack<H, T...> make(X&& x, Y&&... y) {
        return Pack<H, T...>{static_cast<H>(x), PackMaker<T...>::make(y...)};
    }
};
template <class F> struct BodyArgs;
template <class... A> struct BodyArgs<void (*)(VB, A...)> {
    using pack = Pack<A...>;
    template <class... B> static pack make(B&&... b) { return PackMaker<A...>::make(b...); }
};

// members per launch so that the descriptor array stays inside the 4 KB of kernel arguments
template <class P> constexpr int batch_cap() {
    return sizeof(P) <= 232 ? 16 : sizeof(P) <= 488 ? 8 : 4;
}
template <class P, int NB> struct BatchArgs {
    unsigned gx[NB], gy[NB];
    P a[NB];
};

template <auto Body, class... Done>
__device__ __forceinline__ void unpack_call(const VB& vb, const Pack<>&, const Done&... d) { Body(vb, d...); }
template <auto Body, class H, class... T, class... Done>
__device__ __forceinline__ void unpack_call(const VB& vb, const Pack<H, T...>& p, const Done&... d) {
    unpack_call<Body>(vb, p.t, d..., p.h);
}

template <auto Body, int LB, int NB, class P>
__global__ void __launch_bounds__(LB) batched_kernel(BatchArgs<P, NB> b) {
    const unsigned m = blockIdx.z;
    const VB vb{(unsigned)blockIdx.x, (unsigned)blockIdx.y, b.gx[m], b.gy[m]};
    if (vb.x >= vb.gx || vb.y >= vb.gy) return;
    unpack_call<Body>(vb, b.a[m]);
}

// one group of parked launches of the same body: `packs[i]` / `grids[i]` belong to member i of the group
typedef void (*BatchLauncher)(hipStream_t st, int nm, const void* const* packs, const dim3* grids, dim3 block, size_t shmem);

// one launch for `cnt` (<= NB) members: the descriptor array has NB slots, so a small group travels in a small argument
// segment (the runtime copies it per launch)
template <auto Body, int LB, int NB, class P>
void batched_launch_n(hipStream_t st, int cnt, const void* const* packs, const dim3* grids, dim3 block, size_t shmem) {
    static_assert(sizeof(BatchArgs<P, NB>) <= 4096, "descriptor array beyond the kernel-argument segment");
    BatchArgs<P, NB> b;
    unsigned mx = 1, my = 1;
    for (int i = 0; i < cnt; ++i) {
        b.gx[i] = grids[i].x;
        b.gy[i] = grids[i].y;
        memcpy(&b.a[i], packs[i], sizeof(P));
        mx = b.gx[i] > mx ? b.gx[i] : mx;
        my = b.gy[i] > my ? b.gy[i] : my;
    }
    for (int i = cnt; i < NB; ++i) {                           // (unused slots: defined bytes in the argument segment)
        b.gx[i] = b.gy[i] = 0;
        memcpy(&b.a[i], packs[0], sizeof(P));
    }
    hipLaunchKernelGGL((batched_kernel<Body, LB, NB, P>), dim3(mx, my, (unsigned)cnt), block, shmem, st, b);
}

template <auto Body, int LB, class P>
void batched_launcher(hipStream_t st, int nm, const void* const* packs, const dim3* grids, dim3 block, size_t shmem) {
    constexpr int CAP = batch_cap<P>();
    for (int lo = 0; lo < nm; lo += CAP) {
        const int cnt = nm - lo < CAP ? nm - lo : CAP;
        if constexpr (CAP >= 16) {
            if (cnt > 8) { batched_launch_n<Body, LB, 16, P>(st, cnt, packs + lo, grids + lo, block, shmem); continue; }
        }
        if constexpr (CAP >= 8) {
            if (cnt > 4) { batched_launch_n<Body, LB, 8, P>(st, cnt, packs + lo, grids + lo, block, shmem); continue; }
        }
        if (cnt > 1) batched_launch_n<Body, LB, 4, P>(st, cnt, packs + lo, grids + lo, block, shmem);
        else batched_launch_n<Body, LB, 1, P>(st, cnt, packs + lo, grids + lo, block, shmem);
    }
}

// ---- runtime (cohort.hip) -------------------------------------------------------------------------------------------
bool cohort_in_fiber();                        // the calling code runs on a member fiber of a cohort being advanced
// park the calling member at a launch / at a wait for the stream / at a reconvergence point
void cohort_park_launch(sella_ctx* c, BatchLauncher fn, const void* pack, size_t pack_bytes, dim3 grid, dim3 block, size_t shmem,
                        const char* name);
void cohort_park_wait(sella_ctx* c);
// Reconvergence points.  Members whose control flow differs (a further round of the root search, another Davidson
// iteration, a deflation) drift apart and would never again park at the same kernel; a barrier holds a member until
// every other live member is parked at a barrier too, and then releases the members that are FURTHEST BEHIND — those with
// the smallest key (phase, iteration, sub-stage), the others stay — so the cohort closes up at the next common point.
// cohort_set_phase: (optimizer step, stage within the step), set by the search loop; cohort_barrier adds the position
// inside the stage.  Both are no-ops outside a cohort.
void cohort_set_phase(sella_ctx* c, long epoch, int stage);
void cohort_barrier(sella_ctx* c, unsigned iter = 0, unsigned sub = 0);

template <auto Body, int LB, class... B>
void cohort_launch(sella_ctx* c, const char* name, dim3 grid, dim3 block, size_t shmem, B&&... b) {
    using Args = BodyArgs<decltype(Body)>;
    typename Args::pack p = Args::make(b...);
    using P = typename Args::pack;
    static_assert(sizeof(P) <= COHORT_PACK_BYTES, "argument pack larger than a member's launch slot");
    cohort_park_launch(c, &batched_launcher<Body, LB, P>, &p, sizeof(P), grid, block, shmem, name);
}

}  // namespace sella

// Launch of a kernel that has a batchable body: on a member fiber the launch is parked and merged with the other
// members' (cohort.hip); everywhere else it is the plain launch on the context's stream.  LB = the body's launch bound.
#define SELLA_LAUNCHB(c, kernel, body, LB, grid, block, shmem, ...)                                          \
    do {                                                                                                     \
        if ((c)->cohort && sella::cohort_in_fiber())                                                         \
            sella::cohort_launch<body, LB>((c), #body, dim3(grid), dim3(block), (size_t)(shmem), __VA_ARGS__); \
        else                                                                                                 \
            hipLaunchKernelGGL((kernel), grid, block, shmem, (c)->stream, __VA_ARGS__);                      \
    } while (0)
