// TEST INFRASTRUCTURE ONLY — runtime for the fake hip_runtime.h (tests/hostemu/README.md).
// One OS thread; every HIP thread of the running block is a fiber.  On x86-64 the fibers switch
// with a dozen instructions (callee-saved registers + stack pointer); elsewhere ucontext is used
// (correct but slow: glibc's swapcontext makes two sigprocmask system calls per switch).
#include <hip/hip_runtime.h>
#undef threadIdx
#undef blockIdx
#undef blockDim
#undef gridDim

#if !defined(__x86_64__)
#include <ucontext.h>
#endif
#include <chrono>
#include <mutex>
#include <vector>

// AddressSanitizer build (HOSTEMU_SANITIZE=1, build_emu.py): a fibre that finishes never returns from its frames, so
// the redzones they poisoned on its (heap-allocated, recycled) stack must be cleared before the stack is used again.
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#include <sanitizer/asan_interface.h>
#define HIPEMU_ASAN 1
#endif
#endif

namespace hipemu {

Idx g_threadIdx, g_blockIdx;
dim3 g_blockDim, g_gridDim;

namespace {
constexpr size_t kStack = 256 * 1024;

#if defined(__x86_64__)
extern "C" void hipemu_switch(void** save_sp, void* new_sp);
asm(R"(
    .text
    .globl hipemu_switch
    .type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hipemu_switch,.-hipemu_switch
)");
struct Ctx {
    void* sp = nullptr;
};
inline void ctx_switch(Ctx* from, Ctx* to) { hipemu_switch(&from->sp, to->sp); }
inline void ctx_make(Ctx* c, char* stack, size_t size, void (*entry)()) {
    uintptr_t top = ((uintptr_t)stack + size) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;                 // fake return address of entry (keeps the ABI stack alignment)
    *--sp = (void*)entry;            // popped by the `ret` of the first switch
    for (int i = 0; i < 6; ++i) *--sp = nullptr;
    c->sp = sp;
}
#else
struct Ctx {
    ucontext_t uc;
};
inline void ctx_switch(Ctx* from, Ctx* to) { swapcontext(&from->uc, &to->uc); }
inline void ctx_make(Ctx* c, char* stack, size_t size, void (*entry)()) {
    getcontext(&c->uc);
    c->uc.uc_stack.ss_sp = stack;
    c->uc.uc_stack.ss_size = size;
    c->uc.uc_link = nullptr;
    makecontext(&c->uc, entry, 0);
}
#endif

struct Fiber {
    Ctx ctx;
    char* stack = nullptr;
    bool done = false;
    Idx tid;
    int lin = 0;
};

Ctx g_sched;
std::vector<Fiber> g_fibers;
Fiber* g_cur = nullptr;
const std::function<void()>* g_body = nullptr;
std::vector<char> g_dyn;
void* g_dyn_aligned = nullptr;
long g_launches = 0;

int g_nthreads = 0;      // threads per block of the running launch
int g_live = 0;          // fibers of the block not yet finished
int g_bar_count = 0;
unsigned g_bar_gen = 0;

struct WaveState {
    int live = 0;
    int count = 0;
    unsigned gen = 0;
    // quads (4 consecutive lanes) rendezvous on their own for quad-permute DPP: code in which the quads of a wavefront
    // follow different control flow (one small problem per quad) only ever exchanges data inside a quad
    int qlive[16] = {0};
    int qcount[16] = {0};
    unsigned qgen[16] = {0};
    alignas(64) unsigned char xchg[64][64];
};
std::vector<WaveState> g_waves;

void yield() { ctx_switch(&g_cur->ctx, &g_sched); }

void trampoline() {
    (*g_body)();
    Fiber* f = g_cur;
    f->done = true;
    --g_live;
    WaveState& w = g_waves[f->lin >> 6];
    --w.live;
    const int q = (f->lin & 63) >> 2;
    --w.qlive[q];
    // a finished thread must not strand its peers at a barrier
    if (g_bar_count > 0 && g_bar_count == g_live) { g_bar_count = 0; ++g_bar_gen; }
    if (w.count > 0 && w.count == w.live) { w.count = 0; ++w.gen; }
    if (w.qcount[q] > 0 && w.qcount[q] == w.qlive[q]) { w.qcount[q] = 0; ++w.qgen[q]; }
    ctx_switch(&f->ctx, &g_sched);
    abort();                         // a finished fiber is never resumed
}

void wave_barrier(WaveState& w) {
    unsigned gen = w.gen;
    if (++w.count == w.live) { w.count = 0; ++w.gen; return; }
    while (w.gen == gen) yield();
}

void quad_barrier(WaveState& w, int q) {
    unsigned gen = w.qgen[q];
    if (++w.qcount[q] == w.qlive[q]) { w.qcount[q] = 0; ++w.qgen[q]; return; }
    while (w.qgen[q] == gen) yield();
}
}  // namespace

void* dyn_smem() { return g_dyn_aligned; }
int lane_id() { return g_cur->lin & 63; }
long launches() { return g_launches; }

void block_barrier() {
    unsigned gen = g_bar_gen;
    if (++g_bar_count == g_live) { g_bar_count = 0; ++g_bar_gen; return; }
    while (g_bar_gen == gen) yield();
}

void wave_exchange(const void* in, void* out, size_t bytes, int src_lane) {
    if (bytes > 64) { fprintf(stderr, "hipemu: shuffle payload too large\n"); abort(); }
    WaveState& w = g_waves[g_cur->lin >> 6];
    int lane = g_cur->lin & 63;
    memcpy(w.xchg[lane], in, bytes);
    wave_barrier(w);
    int nlanes = 64;
    int total = g_nthreads;
    int wave_base = (g_cur->lin >> 6) << 6;
    if (wave_base + nlanes > total) nlanes = total - wave_base;
    if (src_lane < 0 || src_lane >= nlanes) src_lane = lane;
    memcpy(out, w.xchg[src_lane], bytes);
    wave_barrier(w);
}

void quad_exchange(const void* in, void* out, size_t bytes, int src_lane) {
    if (bytes > 64) { fprintf(stderr, "hipemu: shuffle payload too large\n"); abort(); }
    WaveState& w = g_waves[g_cur->lin >> 6];
    const int lane = g_cur->lin & 63, q = lane >> 2;
    if ((src_lane >> 2) != q) { fprintf(stderr, "hipemu: quad exchange across quads\n"); abort(); }
    memcpy(w.xchg[lane], in, bytes);
    quad_barrier(w, q);
    int total = g_nthreads;
    int wave_base = (g_cur->lin >> 6) << 6;
    if (wave_base + src_lane >= total) src_lane = lane;
    memcpy(out, w.xchg[src_lane], bytes);
    quad_barrier(w, q);
}

void wave_gather64(const void* in, size_t bytes, void* all64) {
    WaveState& w = g_waves[g_cur->lin >> 6];
    int lane = g_cur->lin & 63;
    memcpy(w.xchg[lane], in, bytes);
    wave_barrier(w);
    for (int l = 0; l < 64; ++l) memcpy((char*)all64 + l * bytes, w.xchg[l], bytes);
    wave_barrier(w);
}

void run_grid(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    // the emulated device is one global machine: launches from several host threads take turns
    static std::mutex mu;
    std::lock_guard<std::mutex> hold(mu);
    ++g_launches;
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads <= 0 || nthreads > 1024) { fprintf(stderr, "hipemu: bad block size %d\n", nthreads); abort(); }
    if (shmem > 160 * 1024) { fprintf(stderr, "hipemu: %zu B of LDS requested (> 160 KiB)\n", shmem); abort(); }
    g_nthreads = nthreads;
    g_blockDim = block;
    g_gridDim = grid;
    g_body = &body;
    g_dyn.assign(shmem + 64, 0);
    g_dyn_aligned = (void*)(((uintptr_t)g_dyn.data() + 63) & ~(uintptr_t)63);
    if ((int)g_fibers.size() < nthreads) {
        size_t old = g_fibers.size();
        g_fibers.resize(nthreads);
        for (size_t i = old; i < g_fibers.size(); ++i) g_fibers[i].stack = (char*)malloc(kStack);
    }
    const int nwaves = (nthreads + 63) / 64;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                g_blockIdx = Idx{bx, by, bz};
                g_waves.assign(nwaves, WaveState());
                g_live = nthreads;
                g_bar_count = 0;
                for (int t = 0; t < nthreads; ++t) {
                    Fiber& f = g_fibers[t];
                    f.done = false;
                    f.lin = t;
                    f.tid = Idx{(unsigned)(t % block.x), (unsigned)((t / block.x) % block.y),
                                (unsigned)(t / (block.x * block.y))};
#ifdef HIPEMU_ASAN
                    __asan_unpoison_memory_region(f.stack, kStack);
#endif
                    ctx_make(&f.ctx, f.stack, kStack, trampoline);
                    g_waves[t >> 6].live++;
                    g_waves[t >> 6].qlive[(t & 63) >> 2]++;
                }
                int remaining = nthreads;
                while (remaining > 0) {
                    int progressed = 0;
                    for (int t = 0; t < nthreads; ++t) {
                        Fiber& f = g_fibers[t];
                        if (f.done) continue;
                        g_cur = &f;
                        g_threadIdx = f.tid;
                        ctx_switch(&g_sched, &f.ctx);
                        ++progressed;
                        if (f.done) --remaining;
                    }
                    if (!progressed) break;
                }
            }
    g_body = nullptr;
}
}  // namespace hipemu

// ---- launch log (which kernels does a path use, and how often): HIPEMU_LAUNCH_LOG=<file> ------------------------------
#include <map>
#include <string>
namespace {
struct LaunchLog {
    std::map<std::string, long> count;
    const char* path = getenv("HIPEMU_LAUNCH_LOG");
    ~LaunchLog() {
        if (!path) return;
        FILE* f = fopen(path, "w");
        if (!f) return;
        for (const auto& kv : count) fprintf(f, "%8ld  %s\n", kv.second, kv.first.c_str());
        fclose(f);
    }
};
LaunchLog g_launch_log;
std::mutex g_launch_log_mu;
}  // namespace
void hipemu_note_launch(const char* kernel_text) {
    if (!g_launch_log.path) return;
    std::lock_guard<std::mutex> hold(g_launch_log_mu);
    ++g_launch_log.count[kernel_text];
}

// ----------------------------------------------------------- host API shim --
struct hipemu_event_t { std::chrono::steady_clock::time_point t; };
struct hipemu_stream_t { int dummy; };

hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof(*p));
    snprintf(p->name, sizeof(p->name), "hostemu (CPU fibers, tests only)");
    snprintf(p->gcnArchName, sizeof(p->gcnArchName), "hostemu");
    p->totalGlobalMem = (size_t)8 << 30;
    p->multiProcessorCount = 256;
    p->clockRate = 2400000;
    p->sharedMemPerBlock = 160 * 1024;
    p->warpSize = 64;
    return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t bytes) {
    // guard bytes on both sides so small overruns are caught by the tests' canaries
    *p = aligned_alloc(256, ((bytes + 255) / 256 + 1) * 256);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t b, hipMemcpyKind) { memmove(d, s, b); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t b, hipMemcpyKind k, hipStream_t) {
    hipemu_note_launch(k == hipMemcpyHostToDevice ? "[hipMemcpyAsync H2D]" : k == hipMemcpyDeviceToHost ? "[hipMemcpyAsync D2H]" : "[hipMemcpyAsync D2D]");
    memmove(d, s, b);
    return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width,
                            size_t height, hipMemcpyKind, hipStream_t) {
    hipemu_note_launch("[hipMemcpy2DAsync]");
    for (size_t r = 0; r < height; ++r) memmove((char*)dst + r * dpitch, (const char*)src + r * spitch, width);
    return hipSuccess;
}
hipError_t hipMemset(void* d, int v, size_t b) { memset(d, v, b); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t b, hipStream_t) { hipemu_note_launch("[hipMemsetAsync]"); memset(d, v, b); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t* s) { *s = new hipemu_stream_t{0}; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { hipemu_note_launch("[hipStreamSynchronize]"); return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemu_event_t; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }     // (every launch is synchronous here)
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipPeekAtLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)4 << 30; *t = (size_t)8 << 30; return hipSuccess; }
