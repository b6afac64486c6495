// cohort.hip — W ensemble members advanced in lockstep by one issuing thread, one batched launch per kernel of the step
// (design: cohort.h).  This file is the host runtime: member fibers, the scheduler that merges their parked launches, and
// the C ABI (`sella_cohort_*`).  Reference semantics: independent `Sella` objects (sella/optimize/optimize.py:42-81,
// 359-440) — a member's results do not depend on who else is in the cohort.
#include "internal.h"

#include <sys/mman.h>

#include <sched.h>

#include <chrono>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#include <sanitizer/asan_interface.h>
#define SELLA_COHORT_ASAN 1
#endif
#endif

using namespace sella;

// ---- fibers: callee-saved registers + stack pointer (x86-64 System V), a dozen instructions per switch ----------------
// (glibc's swapcontext makes two sigprocmask system calls per switch, ~0.5 us: a member parks ~90 times per step.)
#if !defined(__HIP_DEVICE_COMPILE__)
#if !defined(__x86_64__)
#error "cohort.hip: the member fibers are written for x86-64 hosts"
#endif
extern "C" void sella_fiber_switch(void** save_sp, void* new_sp);
asm(R"(
    .text
    .globl sella_fiber_switch
    .type sella_fiber_switch,@function
sella_fiber_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    subq $8, %rsp
    stmxcsr (%rsp)
    fnstcw 4(%rsp)
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    ldmxcsr (%rsp)
    fldcw 4(%rsp)
    addq $8, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size sella_fiber_switch,.-sella_fiber_switch
)");
#else
extern "C" void sella_fiber_switch(void** save_sp, void* new_sp);
#endif

namespace {

constexpr size_t FIBER_STACK = (size_t)4 << 20;      // virtual: touched pages only

enum FiberState { F_IDLE = 0, F_RUNNABLE, F_AT_LAUNCH, F_AT_WAIT, F_AT_BARRIER, F_DONE };

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    int state = F_IDLE;                            // FiberState; with member threads: stored / loaded with release / acquire
    std::function<int()> job;
    int status = SELLA_OK;
    std::string error;
    long n_launch = 0, n_wait = 0, n_barrier = 0;  // parks of this member (since creation of the cohort)
    // parked launch
    unsigned long long key = 0;                    // parked at a barrier: its position (smaller = further behind)
    BatchLauncher fn = nullptr;
    const char* name = "";
    dim3 grid, block;
    size_t shmem = 0;
    alignas(16) char pack[COHORT_PACK_BYTES];
    // member threads (sella_cohort_member_threads): the member's host code runs on a thread of its own
    std::thread worker;
    long job_seq = 0, job_seen = 0;                // guarded by the cohort's mutex
};

inline int load_state(const Fiber& f) { return __atomic_load_n(&f.state, __ATOMIC_ACQUIRE); }
inline void store_state(Fiber& f, int st) { __atomic_store_n(&f.state, st, __ATOMIC_RELEASE); }
inline void cpu_relax(long& spins) {
    __builtin_ia32_pause();
    if ((++spins & 0x3ff) == 0) sched_yield();     // (a preempted partner gets the core when there are fewer cores than spinners)
}

}  // namespace

struct sella_cohort {
    int device = 0;
    hipStream_t stream = nullptr;                 // the cohort's stream: member 0's own
    std::vector<sella_ctx*> members;
    std::vector<hipStream_t> own_stream;          // the members' own streams, put back after a run
    std::vector<Fiber> fibers;
    void* sched_sp = nullptr;
    int current = -1;                             // member whose fiber is running (-1: the scheduler)
    // Member threads instead of fibers: the members' host code between their parks — serial on the fibers of the one
    // issuing thread — runs in parallel, the launches are still merged and issued by the thread that advances the cohort.
    // Every member thread and the issuing thread spin while they wait for each other: width + 1 busy cores per cohort.
    bool member_threads = false, quit = false;
    bool broken = false;                          // a run ended on a HIP error with member threads still parked inside their searches
    std::mutex mu;
    std::condition_variable cv;
    // statistics of the last run / since creation
    long rounds = 0, launches_issued = 0, syncs = 0;
    std::map<std::string, double> host_by_park;             // ... and seconds of member host code in front of a park at that body / wait
    std::map<std::string, std::pair<long, long>> by_name;   // SELLA_COHORT_TRACE=2: body -> (launches asked, launches issued)
    double t_members = 0.0, t_issue = 0.0, t_sync = 0.0;      // seconds: member host code, issuing merged launches, stream synchronisations
};

namespace {

thread_local sella_cohort* g_running = nullptr;   // the cohort this thread is advancing
thread_local sella_cohort* g_member_of = nullptr; // member thread: its cohort ...
thread_local int g_member_idx = -1;               // ... its slot ...
thread_local bool g_member_busy = false;          // ... and whether it is inside a job

inline Fiber& self_fiber(sella_cohort*& co) {
    if (g_member_busy) { co = g_member_of; return co->fibers[g_member_idx]; }
    co = g_running;
    return co->fibers[co->current];
}

void fiber_main();

void fiber_prepare(Fiber& f) {
    if (!f.stack) {
        void* p = mmap(nullptr, FIBER_STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        f.stack = p == MAP_FAILED ? nullptr : static_cast<char*>(p);
    }
#ifdef SELLA_COHORT_ASAN
    if (f.stack) __asan_unpoison_memory_region(f.stack, FIBER_STACK);
#endif
    if (!f.stack) return;
    uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + FIBER_STACK) & ~static_cast<uintptr_t>(15);
    void** sp = reinterpret_cast<void**>(top);
    *--sp = nullptr;                               // fake return address of fiber_main (keeps the ABI's stack alignment)
    *--sp = reinterpret_cast<void*>(&fiber_main);  // popped by the `ret` of the first switch
    for (int i = 0; i < 6; ++i) *--sp = nullptr;   // rbp, rbx, r12 .. r15
    // control words of the creating thread (round to nearest, exceptions masked)
    unsigned int mxcsr = 0;
    unsigned short fcw = 0;
    asm volatile("stmxcsr %0" : "=m"(mxcsr));
    asm volatile("fnstcw %0" : "=m"(fcw));
    unsigned int words[2] = {mxcsr, fcw};
    --sp;
    memcpy(sp, words, 8);
    f.sp = sp;
}

// from a member back to the scheduler: a fiber switches stacks, a member thread publishes its state and spins until the
// scheduler makes it runnable again (what it parked with — launch slot, barrier key — is written before the state)
void park(sella_cohort* co, Fiber& f, FiberState st) {
    if (co->member_threads) {
        store_state(f, st);
        if (st == F_DONE) return;
        long spins = 0;
        while (load_state(f) != F_RUNNABLE) cpu_relax(spins);
        return;
    }
    f.state = st;
    sella_fiber_switch(&f.sp, co->sched_sp);
}

int run_job(Fiber& f) {
    int st;
    try {
        st = f.job();
    } catch (const std::bad_alloc&) {
        set_error("cohort member: out of host memory");
        st = SELLA_E_NOMEM;
    } catch (...) {
        set_error("cohort member: unexpected exception");
        st = SELLA_E_INVALID;
    }
    f.status = st;
    if (st != SELLA_OK) f.error = sella_last_error();
    return st;
}

void member_thread_main(sella_cohort* co, int idx) {
    (void)hipSetDevice(co->device);
    g_member_of = co;
    g_member_idx = idx;
    Fiber& f = co->fibers[idx];
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(co->mu);
            co->cv.wait(lk, [&] { return co->quit || f.job_seq != f.job_seen; });
            if (co->quit) return;
            f.job_seen = f.job_seq;
        }
        g_member_busy = true;
        run_job(f);
        g_member_busy = false;
        store_state(f, F_DONE);
    }
}

void fiber_main() {
    sella_cohort* co = g_running;
    Fiber& f = co->fibers[co->current];
    run_job(f);
    park(co, f, F_DONE);
    abort();                                       // a finished fiber is never resumed
}

void resume(sella_cohort* co, int i) {
    co->current = i;
    co->fibers[i].state = F_RUNNABLE;
    sella_fiber_switch(&co->sched_sp, co->fibers[i].sp);
    co->current = -1;
}

// Advance the members until all of them are done.  A round: run every runnable member until it parks; then merge the
// parked launches body by body (members parked at the same kernel form one launch, in member order); with no launch
// parked, one synchronisation of the stream releases every member parked at a wait; with nobody at a wait either, the
// members at the barrier go on together.
int advance(sella_cohort* co) {
    const int n = (int)co->fibers.size();
    const void* packs[COHORT_MAX];
    dim3 grids[COHORT_MAX];
    int who[COHORT_MAX];
    static const char* trace_env = getenv("SELLA_COHORT_TRACE");
    static const bool trace = trace_env && trace_env[0] == '1';             // 1: one line per scheduler round on stderr
    static const bool count_names = trace_env && trace_env[0] == '2';       // 2: launches per kernel body, printed at destruction
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (;;) {
        ++co->rounds;
        const double t0 = now();
        if (co->member_threads) {
            // the runnable members ARE running, each on its thread: wait until every one of them has parked again
            long spins = 0;
            for (int i = 0; i < n; ++i)
                while (load_state(co->fibers[i]) == F_RUNNABLE) cpu_relax(spins);
        } else
        for (int i = 0; i < n; ++i)
            if (co->fibers[i].state == F_RUNNABLE) {
                if (count_names) {
                    // host time of the member up to its next park, booked on what it parks at
                    const double ta = now();
                    resume(co, i);
                    const Fiber& f = co->fibers[i];
                    co->host_by_park[f.state == F_AT_LAUNCH ? f.name : f.state == F_AT_WAIT ? "(wait)" : f.state == F_AT_BARRIER ? "(barrier)" : "(done)"] += now() - ta;
                } else {
                    resume(co, i);
                }
            }
        const double t1 = now();
        co->t_members += t1 - t0;
        int nl = 0, nw = 0, nb = 0;
        for (int i = 0; i < n; ++i) {
            const int s = load_state(co->fibers[i]);
            nl += s == F_AT_LAUNCH;
            nw += s == F_AT_WAIT;
            nb += s == F_AT_BARRIER;
        }
        if (trace) {
            fprintf(stderr, "cohort round %ld:", co->rounds);
            for (int i = 0; i < n; ++i) {
                const Fiber& f = co->fibers[i];
                fprintf(stderr, " %s", f.state == F_AT_LAUNCH ? f.name : f.state == F_AT_WAIT ? "WAIT" : f.state == F_AT_BARRIER ? "BARRIER" :
                                       f.state == F_DONE ? "done" : "-");
            }
            fprintf(stderr, "\n");
        }
        if (nl + nw + nb == 0) return SELLA_OK;
        if (nl > 0) {
            for (int i = 0; i < n; ++i) {
                Fiber& f = co->fibers[i];
                if (f.state != F_AT_LAUNCH) continue;
                int cnt = 0;
                for (int j = i; j < n; ++j) {
                    Fiber& g = co->fibers[j];
                    if (g.state != F_AT_LAUNCH || g.fn != f.fn || g.shmem != f.shmem || g.block.x != f.block.x || g.block.y != f.block.y ||
                        g.block.z != f.block.z)
                        continue;
                    packs[cnt] = g.pack;
                    grids[cnt] = g.grid;
                    who[cnt++] = j;
                }
                f.fn(co->stream, cnt, packs, grids, f.block, f.shmem);
                ++co->launches_issued;
                if (count_names) { auto& e = co->by_name[f.name]; e.first += cnt; e.second += 1; }
                for (int q = 0; q < cnt; ++q) store_state(co->fibers[who[q]], F_RUNNABLE);
            }
            HIPCHK(hipGetLastError());
            co->t_issue += now() - t1;
            continue;
        }
        if (nw > 0) {
            static const bool spin = getenv("SELLA_COHORT_SPIN") != nullptr;
            if (spin) {
                hipError_t q;
                while ((q = hipStreamQuery(co->stream)) == hipErrorNotReady) {}
                HIPCHK(q);
            } else {
                HIPCHK(hipStreamSynchronize(co->stream));
            }
            co->t_sync += now() - t1;
            ++co->syncs;
            for (int i = 0; i < n; ++i)
                if (co->fibers[i].state == F_AT_WAIT) store_state(co->fibers[i], F_RUNNABLE);
            continue;
        }
        unsigned long long lowest = ~0ull;
        for (int i = 0; i < n; ++i)
            if (co->fibers[i].state == F_AT_BARRIER && co->fibers[i].key < lowest) lowest = co->fibers[i].key;
        for (int i = 0; i < n; ++i)
            if (co->fibers[i].state == F_AT_BARRIER && co->fibers[i].key == lowest) store_state(co->fibers[i], F_RUNNABLE);
    }
}

}  // namespace

namespace sella {

bool cohort_in_fiber() { return g_member_busy || (g_running != nullptr && g_running->current >= 0); }

void cohort_park_launch(sella_ctx* c, BatchLauncher fn, const void* pack, size_t pack_bytes, dim3 grid, dim3 block, size_t shmem,
                        const char* name) {
    (void)c;
    sella_cohort* co;
    Fiber& f = self_fiber(co);
    f.fn = fn;
    f.name = name;
    f.grid = grid;
    f.block = block;
    f.shmem = shmem;
    memcpy(f.pack, pack, pack_bytes);
    ++f.n_launch;
    park(co, f, F_AT_LAUNCH);
}

void cohort_park_wait(sella_ctx* c) {
    (void)c;
    sella_cohort* co;
    Fiber& f = self_fiber(co);
    ++f.n_wait;
    park(co, f, F_AT_WAIT);
}

void cohort_set_phase(sella_ctx* c, long epoch, int stage) {
    if (c) c->cohort_phase = ((unsigned long long)(epoch < 0 ? 0 : epoch) << 8) | (unsigned)(stage & 0xff);
}

void cohort_barrier(sella_ctx* c, unsigned iter, unsigned sub) {
    if (!c || !c->cohort || !cohort_in_fiber()) return;
    sella_cohort* co;
    Fiber& f = self_fiber(co);
    ++f.n_barrier;
    f.key = (c->cohort_phase << 32) | ((unsigned long long)(iter & 0xffffffu) << 8) | (sub & 0xffu);
    park(co, f, F_AT_BARRIER);
}

}  // namespace sella

extern "C" {

int sella_cohort_create(sella_ctx* const* members, int n, sella_cohort** out) {
    if (!members || !out || n < 1 || n > COHORT_MAX) {
        set_error("cohort: between 1 and %d member contexts", COHORT_MAX);
        return SELLA_E_INVALID;
    }
    for (int i = 0; i < n; ++i) {
        if (!members[i] || members[i]->cohort || members[i]->device != members[0]->device) {
            set_error("cohort: member contexts must be distinct, on one device, and in no other cohort");
            return SELLA_E_INVALID;
        }
        for (int j = 0; j < i; ++j)
            if (members[j] == members[i]) { set_error("cohort: member context given twice"); return SELLA_E_INVALID; }
    }
    sella_cohort* co = new sella_cohort();
    co->device = members[0]->device;
    co->stream = members[0]->stream;
    co->members.assign(members, members + n);
    co->own_stream.resize(n);
    co->fibers.resize(n);
    for (int i = 0; i < n; ++i) members[i]->cohort = co;
    *out = co;
    return SELLA_OK;
}

int sella_cohort_destroy(sella_cohort* co) {
    if (!co) return SELLA_OK;
    if (!co->by_name.empty()) {
        fprintf(stderr, "cohort of %d: launches asked / issued by kernel body\n", (int)co->members.size());
        for (const auto& kv : co->by_name) fprintf(stderr, "  %8ld %8ld  %s\n", kv.second.first, kv.second.second, kv.first.c_str());
        fprintf(stderr, "microseconds of member host code in front of a park at\n");
        for (const auto& kv : co->host_by_park) fprintf(stderr, "  %10.1f  host-before %s\n", 1e6 * kv.second, kv.first.c_str());
    }
    {
        std::lock_guard<std::mutex> lk(co->mu);
        co->quit = true;
    }
    co->cv.notify_all();
    for (Fiber& f : co->fibers)
        if (f.worker.joinable()) {
            if (co->broken) f.worker.detach();
            else f.worker.join();
        }
    if (co->broken) return SELLA_OK;                                  // (leaked on purpose: parked threads still point into it)
    for (sella_ctx* c : co->members) c->cohort = nullptr;
    for (Fiber& f : co->fibers)
        if (f.stack) munmap(f.stack, FIBER_STACK);
    delete co;
    return SELLA_OK;
}

// on != 0: the members' host code runs on a worker thread each (created here) instead of on fibers of the advancing thread:
// parallel host code between the parks, the launches still merged by the advancing thread.  Width + 1 cores spin per cohort
// while it runs — for a GPU's share of an ensemble on a host with cores to spare, not for many cohorts per GPU.
int sella_cohort_member_threads(sella_cohort* co, int on) {
    if (!co) return SELLA_E_INVALID;
    if (g_running == co) { set_error("cohort: cannot change its members while it is being advanced"); return SELLA_E_INVALID; }
    if (on && !co->member_threads) {
        for (int i = 0; i < (int)co->fibers.size(); ++i)
            if (!co->fibers[i].worker.joinable()) co->fibers[i].worker = std::thread(member_thread_main, co, i);
    }
    co->member_threads = on != 0;
    return SELLA_OK;
}

int sella_cohort_size(sella_cohort* co) { return co ? (int)co->members.size() : 0; }

// counters[8]: scheduler rounds, launches parked by the members, launches issued (merged), waits parked, stream
// synchronisations, barrier arrivals, microseconds inside the members' host code, microseconds issuing the merged
// launches — accumulated since creation
int sella_cohort_stats(sella_cohort* co, long* counters) {
    if (!co || !counters) return SELLA_E_INVALID;
    long nl = 0, nw = 0, nb = 0;
    for (const Fiber& f : co->fibers) { nl += f.n_launch; nw += f.n_wait; nb += f.n_barrier; }
    const long v[8] = {co->rounds, nl, co->launches_issued, nw, co->syncs, nb, (long)(1e6 * co->t_members), (long)(1e6 * co->t_issue)};
    memcpy(counters, v, sizeof(v));
    return SELLA_OK;
}

const char* sella_cohort_error(sella_cohort* co, int member) {
    if (!co || member < 0 || member >= (int)co->fibers.size()) return "";
    return co->fibers[member].error.c_str();
}

// `Optimizer.irun` of up to W searches at once: search i must live on member context i (NULL: the slot stays empty).
// status[i] / converged[i] are what sella_search_run would have returned for that member on its own.
int sella_cohort_run_searches(sella_cohort* co, sella_search* const* searches, int n, double fmax, long steps, int* converged,
                              int* status) {
    if (!co || !searches || !converged || !status || n < 1 || n > (int)co->members.size()) {
        set_error("cohort: invalid arguments");
        return SELLA_E_INVALID;
    }
    if (g_running) { set_error("cohort: a cohort is already being advanced on this thread"); return SELLA_E_INVALID; }
    if (co->broken) { set_error("cohort: an earlier run ended on a device error"); return SELLA_E_INVALID; }
    for (int i = 0; i < n; ++i)
        if (searches[i] && sella_search_ctx(searches[i]) != co->members[i]) {
            set_error("cohort: search %d does not live on member context %d", i, i);
            return SELLA_E_INVALID;
        }
    HIPCHK(hipSetDevice(co->device));
    // everything the members queued on their own streams so far is complete before they share one
    const int W = (int)co->members.size();
    for (int i = 0; i < W; ++i) {
        sella_ctx* c = co->members[i];
        SCHK(stream_wait(c));
        co->own_stream[i] = c->stream;
        c->stream = co->stream;
        c->stream_main = co->stream;
    }
    for (int i = 0; i < W; ++i) {
        Fiber& f = co->fibers[i];
        f.state = F_IDLE;
        f.status = SELLA_OK;
        f.error.clear();
        if (i >= n || !searches[i]) continue;
        converged[i] = 0;
        sella_search* S = searches[i];
        int* conv = &converged[i];
        f.job = [S, fmax, steps, conv]() { return sella_search_run(S, fmax, steps, conv); };
        if (co->member_threads) {
            store_state(f, F_RUNNABLE);
            std::lock_guard<std::mutex> lk(co->mu);
            ++f.job_seq;
            continue;
        }
        fiber_prepare(f);
        if (!f.stack) {
            for (int j = 0; j < W; ++j) { co->members[j]->stream = co->own_stream[j]; co->members[j]->stream_main = co->own_stream[j]; }
            set_error("cohort: no memory for a member's stack");
            return SELLA_E_NOMEM;
        }
        f.state = F_RUNNABLE;
    }
    if (co->member_threads) co->cv.notify_all();
    g_running = co;
    const int st = advance(co);
    g_running = nullptr;
    if (st != SELLA_OK && co->member_threads) co->broken = true;      // (their searches cannot be unwound: the threads are left behind)
    (void)hipStreamSynchronize(co->stream);
    for (int i = 0; i < W; ++i) {
        sella_ctx* c = co->members[i];
        c->stream = co->own_stream[i];
        c->stream_main = co->own_stream[i];
    }
    for (int i = 0; i < n; ++i) status[i] = searches[i] ? co->fibers[i].status : SELLA_OK;
    return st;
}

}  // extern "C"
