// TEST INFRASTRUCTURE ONLY — fake HIP runtime header for CPU-side CI.
// See tests/hostemu/README.md.  Compiled with clang++ -x c++ (host only).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

// ----------------------------------------------------------------- basics --
struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef int hipError_t;
enum : int {
    hipSuccess = 0,
    hipErrorInvalidValue = 1,
    hipErrorOutOfMemory = 2,
    hipErrorNotReady = 600,
    hipErrorNoDevice = 100,
};
typedef struct hipemu_stream_t* hipStream_t;
typedef struct hipemu_event_t* hipEvent_t;
enum hipMemcpyKind {
    hipMemcpyHostToHost = 0,
    hipMemcpyHostToDevice = 1,
    hipMemcpyDeviceToHost = 2,
    hipMemcpyDeviceToDevice = 3,
    hipMemcpyDefault = 4
};
struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    size_t totalGlobalMem;
    int multiProcessorCount;
    int clockRate;
    size_t sharedMemPerBlock;
    int warpSize;
    int l2CacheSize;
};
#define hipHostMallocDefault 0
#define hipStreamNonBlocking 1
#define hipEventDefault 0

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define HIP_KERNEL_NAME(...) __VA_ARGS__
#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)hipemu::dyn_smem();

struct double2 { double x, y; };
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
struct double4 { double x, y, z, w; };
static inline double4 make_double4(double x, double y, double z, double w) { return double4{x, y, z, w}; }
struct int2 { int x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

// ---------------------------------------------------------------- runtime --
namespace hipemu {
struct Idx { unsigned x, y, z; };
extern Idx g_threadIdx, g_blockIdx;
extern dim3 g_blockDim, g_gridDim;
void* dyn_smem();
void run_grid(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
void block_barrier();
void wave_exchange(const void* in, void* out, size_t bytes, int src_lane);
void quad_exchange(const void* in, void* out, size_t bytes, int src_lane);  // the four lanes of a quad rendezvous on their own
void wave_gather64(const void* in, size_t bytes, void* all64);  // every lane gets all 64 values
int lane_id();
long launches();

template <class K, class... A>
static inline void launch(K kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t, A... args) {
    std::function<void()> body = [=]() { kernel(args...); };
    run_grid(grid, block, shmem, body);
}
}  // namespace hipemu

#define threadIdx hipemu::g_threadIdx
#define blockIdx hipemu::g_blockIdx
#define blockDim hipemu::g_blockDim
#define gridDim hipemu::g_gridDim

void hipemu_note_launch(const char* kernel_text);      // HIPEMU_LAUNCH_LOG=<file>: histogram of launched kernels at exit
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    (hipemu_note_launch(#kernel), hipemu::launch(kernel, dim3(grid), dim3(block), (size_t)(shmem), stream, ##__VA_ARGS__))

static inline void __syncthreads() { hipemu::block_barrier(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}
static inline void __threadfence_system() {}

template <class T>
static inline T __shfl(T v, int src, int width = 64) {
    int lane = hipemu::lane_id();
    int base = (lane / width) * width;
    T out;
    hipemu::wave_exchange(&v, &out, sizeof(T), base + (src % width + width) % width);
    return out;
}
template <class T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
    int lane = hipemu::lane_id();
    int src = lane ^ mask;
    if (src / width != lane / width) src = lane;
    T out;
    hipemu::wave_exchange(&v, &out, sizeof(T), src);
    return out;
}
template <class T>
static inline T __shfl_down(T v, unsigned delta, int width = 64) {
    int lane = hipemu::lane_id();
    int src = lane + (int)delta;
    if (src / width != lane / width) src = lane;
    T out;
    hipemu::wave_exchange(&v, &out, sizeof(T), src);
    return out;
}
template <class T>
static inline T __shfl_up(T v, unsigned delta, int width = 64) {
    int lane = hipemu::lane_id();
    int src = lane - (int)delta;
    if (src < 0 || src / width != lane / width) src = lane;
    T out;
    hipemu::wave_exchange(&v, &out, sizeof(T), src);
    return out;
}

// DPP / readlane subset used by the wave reductions (quad_perm, row_mirror 0x140, row_half_mirror 0x141)
static inline int hipemu_update_dpp(int old, int src, int ctrl, int, int, bool) {
    (void)old;
    const int l = hipemu::lane_id(), base = l & ~15, idx = l & 15;
    int from;
    if (ctrl < 0x100) from = base + (idx & ~3) + ((ctrl >> (2 * (idx & 3))) & 3);
    else if (ctrl == 0x140) from = base + (15 - idx);
    else if (ctrl == 0x141) from = base + (idx & 8) + (7 - (idx & 7));
    else { fprintf(stderr, "hipemu: unsupported DPP control 0x%x\n", ctrl); abort(); }
    int out;
    if (ctrl < 0x100) hipemu::quad_exchange(&src, &out, sizeof(int), from);   // quad permutes stay inside the quad
    else hipemu::wave_exchange(&src, &out, sizeof(int), from);
    return out;
}
#define __builtin_amdgcn_update_dpp hipemu_update_dpp
static inline int hipemu_readlane(int v, int lane) {
    int out;
    hipemu::wave_exchange(&v, &out, sizeof(int), lane);
    return out;
}
#define __builtin_amdgcn_readlane hipemu_readlane
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
static inline long long wall_clock64() { return 0; }
// lanes of a wavefront are fibers here: where the hardware's lockstep orders "all lanes wrote LDS, then any lane reads it",
// the kernels say so with a wave barrier (a compiler-only barrier on the device, a rendezvous of the 64 fibers here)
static inline void hipemu_wave_barrier() { int x = 0, y; hipemu::wave_exchange(&x, &y, sizeof(int), hipemu::lane_id()); (void)y; }
#define __builtin_amdgcn_wave_barrier hipemu_wave_barrier
static inline int __double2loint(double d) { long long b; memcpy(&b, &d, 8); return (int)(b & 0xffffffffll); }
static inline int __double2hiint(double d) { long long b; memcpy(&b, &d, 8); return (int)(b >> 32); }
static inline double __hiloint2double(int hi, int lo) {
    unsigned long long b = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
    double d; memcpy(&d, &b, 8); return d;
}

static inline double atomicAdd(double* p, double v) { double o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline int atomicMax(int* p, int v) { int o = *p; if (v > o) *p = v; return o; }
static inline int atomicMin(int* p, int v) { int o = *p; if (v < o) *p = v; return o; }

// f64 MFMA 16x16x4, gfx950 layout (cdna_hip_programming.md §3):
//   A: lane l holds A[i = l & 15][k = l >> 4];  B: lane l holds B[k = l >> 4][j = l & 15]
//   C/D: 4 values per lane, col = l & 15, row = (l >> 4) + 4 * reg
typedef double hipemu_f64x4 __attribute__((ext_vector_type(4)));
static inline hipemu_f64x4 hipemu_mfma_f64_16x16x4(double a, double b, hipemu_f64x4 c, int, int, int) {
    double A[64], B[64];
    hipemu::wave_gather64(&a, sizeof(double), A);
    hipemu::wave_gather64(&b, sizeof(double), B);
    int l = hipemu::lane_id();
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) + 4 * r;
        double acc = c[r];
        for (int k = 0; k < 4; ++k) acc = std::fma(A[k * 16 + row], B[k * 16 + col], acc);
        c[r] = acc;
    }
    return c;
}
#define __builtin_amdgcn_mfma_f64_16x16x4f64 hipemu_mfma_f64_16x16x4

// ------------------------------------------------------------- host "API" --
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d);
hipError_t hipMalloc(void** p, size_t bytes);
template <class T> static inline hipError_t hipMalloc(T** p, size_t bytes) { return hipMalloc((void**)p, bytes); }
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned flags = 0);
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t bytes, unsigned flags = 0) { return hipHostMalloc((void**)p, bytes, flags); }
hipError_t hipHostFree(void* p);
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s = nullptr);
hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width,
                            size_t height, hipMemcpyKind kind, hipStream_t s = nullptr);
hipError_t hipMemset(void* dst, int value, size_t bytes);
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t s = nullptr);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize();
hipError_t hipEventCreate(hipEvent_t* e);
#define hipEventDisableTiming 2u
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags = 0);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipGetLastError();
hipError_t hipPeekAtLastError();
const char* hipGetErrorString(hipError_t e);
hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b);
// Graph capture is not emulated: hipStreamBeginCapture reports failure and the library issues the
// launches directly (the branch it also takes on a device whose stream cannot capture).
typedef struct hipemu_graph_t* hipGraph_t;
typedef struct hipemu_graphexec_t* hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };
inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorInvalidValue; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorInvalidValue; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, void*, void*, size_t) { *e = nullptr; return hipErrorInvalidValue; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorInvalidValue; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
