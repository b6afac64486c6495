// internal.h — shared declarations of libsella_hip (context, device matrices, kernel launchers).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <deque>
#include <map>
#include <string>
#include <vector>

#include "../../include/sella_hip.h"

#define SELLA_HD __host__ __device__
#include "small_linalg.h"
#include "cohort.h"

// "this value is in a register from here on": an empty statement the compiler cannot look through.  Placed behind a batch
// of loads it keeps them unconditional and in flight together — otherwise a load whose only use sits behind a condition is
// sunk behind that condition (and the wait for it with everything issued before it).
// SELLA_ARG(x): the same for a kernel argument (uniform): all arguments named at the top of a kernel are fetched by one
// batch of scalar loads instead of one dependent fetch per branch that first needs them.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SELLA_NO_ARG_BATCH)
#define SELLA_PIN(x) asm volatile("" : "+v"(x))
#define SELLA_ARG(x) asm volatile("" : : "s"(x))
#elif defined(__HIP_DEVICE_COMPILE__)
#define SELLA_PIN(x) asm volatile("" : "+v"(x))
#define SELLA_ARG(x) ((void)0)
#else
#define SELLA_PIN(x) ((void)0)
#define SELLA_ARG(x) ((void)0)
#endif

namespace sella {

// ---- wave64 sum, every lane gets the result -------------------------------------------------------
// Four DPP steps (two quad permutes, half-row mirror, row mirror: VALU-rate cross-lane moves inside
// rows of 16 lanes) and four v_readlane for the rows.  `__shfl_xor` on a double lowers to a pair of
// ds_bpermute_b32 per step, six dependent LDS round trips — the reductions were a visible part of the
// 4.5 us floor of the latency-bound kernels.  All 64 lanes must be active.
template <int CTRL>
__device__ __forceinline__ double wave_dpp_add(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_readlane(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                            __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_sum64(double v) {
    v = wave_dpp_add<0xB1>(v);          // quad_perm [1,0,3,2]
    v = wave_dpp_add<0x4E>(v);          // quad_perm [2,3,0,1]
    v = wave_dpp_add<0x141>(v);         // row_half_mirror
    v = wave_dpp_add<0x140>(v);         // row_mirror
    return (wave_readlane(v, 0) + wave_readlane(v, 16)) + (wave_readlane(v, 32) + wave_readlane(v, 48));
}

void set_error(const char* fmt, ...);

#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            sella::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),        \
                             __FILE__, __LINE__);                                          \
            return SELLA_E_HIP;                                                            \
        }                                                                                  \
    } while (0)

#define SCHK(expr)                                                                         \
    do {                                                                                   \
        int s_ = (expr);                                                                   \
        if (s_ != SELLA_OK) return s_;                                                     \
    } while (0)

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline long round_up_l(long x, long m) { return (x + m - 1) / m * m; }

// Device matrix, row-major, leading dimension padded to a multiple of 8 doubles (64 B) with
// the padding kept at zero, so 16-byte vector loads never straddle a row and never see NaNs.
struct Mat {
    double* d = nullptr;
    int rows = 0, cols = 0, ld = 0;
    bool live = false;
};

enum ProfKind { PROF_GEMV = 0, PROF_GEMM = 1, PROF_UPDATE = 2, PROF_OTHER = 3, PROF_GEMV_SMALL = 4, PROF_TRD_GEMV = 5,
                PROF_NKIND = 6 };

struct ProfPending {
    hipEvent_t a, b;
    int kind;
    double bytes, flops;
};

struct ProfSlot {
    long launches = 0;
    double ms = 0, bytes = 0, flops = 0;
};

struct Options {
    long gemv_rw = 0;        // rows per workgroup in the row-panel matvec (1, 2, 4; 0 = by size: 4 from 4096 rows on, else 2)
    long gemm_mfma = 1;      // 1: MFMA f64 16x16x4 GEMM tiles, 0: VALU register tiles
    long host_scalars = 0;   // 1: host-consumed scalars are written straight into pinned host memory (measured: no gain)
    long gemm_tile128 = 1;   // 1: 128x128 double-buffered tiles for large NN/TN products
    long eigh_leaf = 16;     // leaf size of the divide-and-conquer tree (16: 39.7 ms per eigh at n = 3072, 32: 40.2, 64: 41.5)
    long eigh_symv_tri = 1;      // ... and the trailing update then writes the upper triangle only (mirrored once, when the
                                 // trailing block drops below eigh_symv_min)
    long eigh_symv_tr = 64;      // rows per tile of that matvec (64 or 128)
    long eigh_symv_min = 5120;   // trailing blocks of at least this many rows use the symmetric-aware matvec of the
                                 // tridiagonalisation (upper triangle only, eigh.hip); 0: never
    long eigh_nb = 16;       // panel width of the blocked tridiagonalisation (tools/eigh_tune.py)
    long panel_mfma = 1;     // 1: products with more than 8 right-hand sides stream the matrix once (MFMA panel kernel)
    long panel_rows = 0;     // rows per workgroup of the panel kernel: 16, 32, 48, 64, or 0 = by size (one workgroup per CU)
    long eigh_wy_mfma = 1;   // 1: back-transformation on the matrix cores, 0: VALU/LDS variant
    long dav_fuse_scale = 1;     // Davidson chain: (d - theta)^-1 scaling in the epilogue of the residual kernel (0: own kernel)
    long dav_poll = 1;           // ... 1: the fused iteration's one wait polls a sequence word in pinned host memory, written by a
                                 //    one-thread kernel behind the iteration's last kernel (the scalars of the iteration are then
                                 //    stored there by their kernels for the length of the call), instead of sleeping on an event — the wake-up of an interrupt-driven wait is
                                 //    ~10 us of a ~95 us iteration
    long dav_zero_copy = 0;      // ... its coefficients read from pinned host memory instead of a copy launch (measured equal or slower)
    long eigh_tail_lds = 128;    // trailing blocks of at most this many rows (<= 128) are tridiagonalised by one workgroup in LDS (0: never)
    long eigh_wy_nb64_min = 2560; // 64 instead of 32 reflectors per block of the back-transformation from this many rows on (0: never)
    long eigh_wy_rows = 16;  // rows of X per workgroup of the MFMA back-transformation (16, or 32: two row tiles)
    long eigh_wy_waves = 4;  // wavefronts per workgroup of the MFMA back-transformation (4, 8 or 16: measured equal at n = 3072 and 12288 — the kernel is bound by L2 bandwidth, 22.7 GB in 3.16 ms, not by latency)
    long lr_cholqr = 1;      // 1: block of update vectors orthonormalised by Cholesky-QR twice (eigh.hip, lr_lowrank_update)
    long h2d_kernel_min = 16384; // host-to-device payloads of at least this many bytes are copied by a kernel reading the pinned ring (0: never)
    long eigh_dc_pipeline = 1; // 1: divide & conquer queues the next level's rank-one vectors behind the current level (one wait per level)
    long eigh_gemv_flat = 1; // 1: trailing matvec of the tridiagonalisation with every load issued before the first wait (eigh.hip)
    long eigh_wy_strip = 1;  // 1: back-transformation with the strip of X in registers for the whole sweep (n = ld a multiple of 64,
                             //    2048 < n <= 3072: 2.49 -> 2.19 ms, L2 traffic 19.9 -> 14.5 GB); 2: any such n <= 3072 (tests); 0: never
    long rank2k_fixed = 1;   // 1: trailing update with all loads issued up front for the panel depths 16 / 32 (update.hip)
    long rs_fast = 1;        // 1: sella_opt_step searches the restricted step by interpolating batches (stepper.hip)
    long lr_dev = 1;         // 1: sella_opt_step updates structured decompositions in coordinates, all decisions on the device (lrstep.hip)
    long rank2k_stream = 1;  // 1: trailing update of the tridiagonalisation as a mirror-free MFMA stream (update.hip)
    long panel_small = 2048; // panel products with <= 64 rows and at least this many columns split the long index over the
                             // chip (kernels.hip); 0: never
    long dav_rotate_fused = 1; // sella_davidson's result stage: rotation into the Ritz basis and the caller's layout in one launch (k <= 32)
    long bd_early_matvec = 1; // pipelined block Davidson: 1 = A applied to the raw correction block while the host orthonormalises it
                             // (A T by the same coefficients as T, error budget; see davidson_block.hip), 0 = A applied to the final T
    long bd_pipeline = 1;    // block Davidson: pipelined iteration (davidson_block.hip run_pipelined: A applied to the raw correction
                             // block while the host does the SVQB step, two polled waits per iteration); 0: the general loop
    long eigh_wy_overlap = 1; // 1: Gram matrices / triangular factors of the compact-WY blocks on a second stream, beside divide & conquer
    long eigh_upd_max = 1024; // trailing blocks of at most this many rows are tridiagonalised with ONE launch per column, the block
                             //    kept up to date by the launch itself (trd_upd_kernel, eigh.hip); 0: never.  eigh at n = 3072 by
                             //    switch-over size (session r05b): 0: 39.85 ms, 512: 38.78, 1024: 38.28, 1536: 39.07, 2048: 40.19,
                             //    3072: 47.9 — the block is written back once per column, which only pays while it is small
    long eigh_upd_rows = 0;  // rows per workgroup of that kernel (2, 4, 8); 0: by trailing size, thresholds below
    long eigh_upd_nt = 512;  // most threads per workgroup of that kernel (128, 256, 512; tests force several chunks per thread with 128)
    long eigh_upd_r4_min = 1 << 30, eigh_upd_r8_min = 1 << 30;   // (2 rows per workgroup measured best at every size up to 1024)
    long emt_hcap = 8;       // neighbour-list slots per thread of the EMT kernels (tests: 1 forces the overflow path)
    long lr_overlap = 0;     // 1: the view job of the one-call step is queued on a second stream, beside the coordinate kernels of the
                             //    full-space job.  Measured (session r04k): EMT-slab step 0.59-0.62 ms either way, and the ensemble of EMT
                             //    members DROPS from 211 to 172-182 searches/s with 8 threads x 2 streams: off
    long rs_batch_result = 1; // 1: on an expected boundary step the start value rides in the first batch and the final step is read from the
                              //    batch that produced it (stepper.hip)
    long lr_pipe = 1;        // 1: the library search queues the force call in front of the update that consumes it: one wait for both (search.hip)
    long lr_chain = 1;       // 1: the O(n r) passes of the one-call step as five fused launches, merged coordinate kernels (lrstep.hip)
    long rs_hint = 1;        // 1: the batched root search interpolates quadratically through three evaluated points and, given the alpha the
                             //    previous root search of the same saddle search ended at (sella_opt_step_t::alpha_hint), looks around it first:
                             //    4.9 -> 3.5 rounds per boundary step on the EMT slab
    long rs_poll = 0;        // 1: the rounds of the batched root search wait by polling a pinned sequence word (context.hip, poll_wait); measured equal on the EMT slab (0.576 vs 0.577 ms per step: the mark kernel costs what the wake-up saves): off
    long gs_small = 2048;    // Gram-Schmidt of vectors of at most this many entries (<= 2048) in ONE launch of one workgroup, sweeps,
                             //    norms and accept / drop decisions included (gs.hip); 0: always the sweep-by-sweep launches
    long rs_batch = 1;       // 1: bisection phase of the restricted-step root find evaluates 15 trial alphas per round trip (stepper.hip)
};

}  // namespace sella

struct sella_cohort;
struct sella_search;
extern "C" sella_ctx* sella_search_ctx(sella_search* search);

struct sella_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // member of a cohort (cohort.h): while the cohort is being advanced, `stream` is the cohort's stream, batchable
    // launches are parked and merged across the members, and every wait is one synchronisation for all of them
    sella_cohort* cohort = nullptr;
    unsigned long long cohort_phase = 0;       // (optimizer step << 8 | stage) of the member, the high part of its barrier keys
    // Handle table.  A deque: push_back never moves existing elements, so a `Mat*` obtained from mat_get stays valid
    // while the handle is live, even when a host callback (sella_matvec_fn / sella_allgather_fn) re-enters the library
    // and creates matrices.  (It was a std::vector until round 3: a reallocation inside the Davidson callback left
    // the solver reading a freed Mat.)
    std::deque<sella::Mat> mats;
    // small exchange buffers: device scalars + pinned host mirror
    double* dscal = nullptr;
    double* hscal = nullptr;
    int nscal = 0;
    // Device allocator: blocks are carved out of a few large arenas and recycled through per-size free
    // lists.  A single hipMalloc was measured at 40-80 ms on this stack whenever the driver has to map new
    // memory, and hipFree synchronises the device; an optimizer loop (or every new member of an ensemble)
    // allocates the same few sizes over and over.  One stream per context, so a recycled block is always
    // used after its last reader.  Arenas are only returned when the context is destroyed.
    struct Arena {
        char* base;
        size_t size, used;
    };
    std::vector<Arena> arenas;
    std::map<size_t, std::vector<void*>> pool;
    // pooled scratch (grown on demand, never shrunk)
    std::vector<std::pair<double*, size_t>> scratch;   // slot -> (ptr, bytes)
    sella::Options opt;
    bool prof = false;
    hipEvent_t prof_a = nullptr, prof_b = nullptr;   // events of the launch being profiled
    std::vector<sella::ProfPending> pending;
    sella::ProfSlot slots[sella::PROF_NKIND];
    char name[256] = {0};
    int num_cu = 256;
    // pinned host staging for bursts of small transfers (divide & conquer levels): pageable copies are
    // synchronous staged copies, ~20 us each with the queue empty
    void* hstage = nullptr;
    size_t hstage_bytes = 0;
    char* hring = nullptr;          // pinned ring behind h2d_async
    size_t hring_bytes = 0, hring_pos = 0;
    // pinned ring behind d2h_async: device -> CALLER memory without a pageable (runtime-staged, process-serialising)
    // copy; the payload lands in the ring and is handed to its destination by stream_wait()
    char* dring = nullptr;
    size_t dring_bytes = 0, dring_pos = 0;
    struct PendingD2H { void* dst; const char* slot; size_t bytes; size_t dpitch, width, rows; };
    std::vector<PendingD2H> d2h_pending;
    std::vector<double> hbuf_a, hbuf_b;             // host work vectors of sella_opt_step (optstep.hip)
    // Re-entrancy: the working state a library call may leave LIVE across a host callback (scalar exchange buffers,
    // pooled scratch slots, pinned staging, host work vectors) exists once per call depth.  callback_enter parks the
    // caller's set and installs the set of the next depth (created on first use); callback_leave restores it.  So a
    // callback may call ANY entry point of the library on the same context without touching what the interrupted
    // call still needs.
    struct Frame {
        double* dscal = nullptr;
        double* hscal = nullptr;
        std::vector<std::pair<double*, size_t>> scratch;
        void* hstage = nullptr;
        size_t hstage_bytes = 0;
        std::vector<double> hbuf_a, hbuf_b;
    };
    // second stream for work that is independent of the main chain for a while (the view job of the one-call optimizer
    // step, lrstep.hip): forked and joined with events, so every wait on `stream` still covers it
    hipStream_t stream2 = nullptr;
    hipStream_t stream_main = nullptr;         // == stream except while a job is being queued on stream2
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    unsigned long long* poll_word = nullptr;   // pinned sequence word of polled waits (davidson.hip, dav_poll) and its last value
    unsigned long long poll_seq = 0;
    unsigned* poll_count = nullptr;            // device counter of poll_arm (zero between uses)
    bool stream2_detached = false;             // stream2 runs work no wait of the main chain has to cover (eigh.hip, WY factors)
    std::deque<Frame> frames;      // frames[d] = parked state of depth d (d != depth)
    int depth = 0;
};

namespace sella {

// ---- context helpers (context.hip) ------------------------------------------------------
int mat_new(sella_ctx* c, int rows, int cols, sella_mat* h);       // zero-initialised
// Bracket EVERY invocation of a host callback with these (see sella_ctx::Frame); CallbackScope does it by scope.
int callback_enter(sella_ctx* c);
void callback_leave(sella_ctx* c);
struct CallbackScope {
    sella_ctx* c;
    int status;
    explicit CallbackScope(sella_ctx* ctx) : c(ctx), status(callback_enter(ctx)) {}
    ~CallbackScope() { if (status == SELLA_OK) callback_leave(c); }
    CallbackScope(const CallbackScope&) = delete;
    CallbackScope& operator=(const CallbackScope&) = delete;
};
Mat* mat_get(sella_ctx* c, sella_mat h);
int scratch_get(sella_ctx* c, int slot, size_t bytes, double** p); // persistent scratch slot
int host_stage(sella_ctx* c, size_t bytes, void** p);               // pinned host staging buffer (grown on demand)
int dev_alloc(sella_ctx* c, size_t bytes, double** p);             // caching allocator (contents undefined)
void dev_free(sella_ctx* c, double* p, size_t bytes);
int upload_panel(sella_ctx* c, const double* X, int n, int k, double* dpanel, int ldp);   // (n x k) host -> k rows
int download_panel(sella_ctx* c, const double* dpanel, int ldp, int n, int k, double* X); // k rows -> (n x k) host
int h2d_async(sella_ctx* c, void* dst, const void* src, size_t bytes);   // caller memory -> device, no wait (pinned ring)
int h2d_pinned(sella_ctx* c, void* dst, const void* pinned_src, size_t bytes);   // pinned source of the caller: by kernel from 16 KB on (h2d_kernel_min)
int h2d_begin(sella_ctx* c, size_t bytes, void** slot);                   // pinned slot (zeroed) for the caller to compose in ...
int h2d_end(sella_ctx* c, void* dst, const void* slot, size_t bytes);     // ... and its transfer queued
// device -> caller memory through the pinned ring: `dst` is valid after the next stream_wait(c).  The 2-D form copies
// `rows` rows of `width` bytes from a device pitch `spitch` into a dense destination.
int d2h_async(sella_ctx* c, void* dst, const void* src_dev, size_t bytes);
int d2h_async_2d(sella_ctx* c, void* dst, const void* src_dev, size_t spitch, size_t width, size_t rows);
// THE wait of the library: stream synchronisation, then the queued device-to-host payloads are delivered and both
// pinned rings rewound.  Every host-side wait goes through here (never hipStreamSynchronize directly).
// Stream-ordered copies and zero-fills of the library.  Outside a cohort they ARE the runtime's hipMemcpyAsync /
// hipMemsetAsync on the context's stream; on a member fiber of a cohort they are batchable kernels like every other launch
// of the step (one launch for all members instead of one runtime blit per member) whenever both sides are visible to the
// device — `host_pinned` says that the host side of an H2D / D2H copy is pinned (hipHostMalloc) memory.
hipError_t s_memcpy(sella_ctx* c, void* dst, const void* src, size_t bytes, hipMemcpyKind kind, bool host_pinned = false);
hipError_t s_memcpy2d(sella_ctx* c, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows,
                      hipMemcpyKind kind, bool host_pinned = false);
hipError_t s_memset0(sella_ctx* c, void* dst, size_t bytes, const char* file = __builtin_FILE(), int line = __builtin_LINE());
int stream_wait(sella_ctx* c);
int stream_sync_raw(sella_ctx* c);                                  // the synchronisation alone (nothing delivered, rings kept)
int event_wait(sella_ctx* c, hipEvent_t ev);
// Polled wait: poll_mark queues a one-thread kernel that stores the next sequence number into a pinned word behind
// everything queued so far; poll_wait spins on that word (a cohort member parks instead).  What kernels in front of the mark
// stored into pinned host memory is visible when the word is.
int poll_mark(sella_ctx* c);
// the same without the extra kernel: the LAST workgroup of the caller's kernel publishes `seq` to `*word` (pinned) after
// counting itself on `*count` (device, zero between uses); on a member fiber of a cohort all three come back null / 0
int poll_arm(sella_ctx* c, unsigned long long** word, unsigned long long* seq, unsigned** count);
int poll_wait(sella_ctx* c);                        // wait for an event recorded on the context's stream
int read_scalars(sella_ctx* c, int offset, int count);             // dscal -> hscal (sync)
// Where a kernel should put scalars that only the HOST consumes next: with `host_scalars` on, the pinned,
// device-visible host mirror itself (zero-copy: the readback is then just the stream synchronisation and the
// 4 us copy launch disappears from the critical path); otherwise the device buffer.
double* scal_out(sella_ctx* c, int offset);
int sync_scalars(sella_ctx* c, int offset, int count);             // makes hscal[offset, offset+count) valid
// layout of the scalar exchange buffer (doubles)
enum { DS_MISC = 0, DS_CVEC = 16384, DS_STAGE = 32768, DS_GRAM = 65536, DS_TOTAL = 131072 };
// Profiling (bench.py roofline leg): prof_begin creates an event pair, the launch between begin and
// end must go through SELLA_LAUNCH, which attaches the pair to the kernel's own dispatch packet
// (hipExtLaunchKernelGGL) so that the elapsed time is the kernel's execution time — the same
// start/end timestamps rocprofv3 reports — and not the distance between two stream markers.
void prof_begin(sella_ctx* c, int kind, double bytes, double flops);
void prof_end(sella_ctx* c);
int prof_flush(sella_ctx* c);
#define SELLA_LAUNCH(c, kernel, grid, block, shmem, ...)                                             \
    do {                                                                                             \
        if ((c)->prof_a)                                                                             \
            hipExtLaunchKernelGGL((kernel), grid, block, shmem, (c)->stream, (c)->prof_a,            \
                                  (c)->prof_b, 0, __VA_ARGS__);                                                   \
        else                                                                                         \
            hipLaunchKernelGGL((kernel), grid, block, shmem, (c)->stream, __VA_ARGS__);              \
    } while (0)

// the same for a kernel with a batchable body (cohort.h): parked and merged on a member fiber of a cohort
#define SELLA_BODY(...) __VA_ARGS__
#define SELLA_LAUNCHB_PROF(c, kernel, body, LB, grid, block, shmem, ...)                                     \
    do {                                                                                                     \
        if ((c)->cohort && sella::cohort_in_fiber())                                                         \
            sella::cohort_launch<body, LB>((c), #body, dim3(grid), dim3(block), (size_t)(shmem), __VA_ARGS__); \
        else                                                                                                 \
            SELLA_LAUNCH(c, (kernel), grid, block, shmem, __VA_ARGS__);                                      \
    } while (0)

enum ScratchSlot {
    SCR_X = 0, SCR_Y, SCR_PART, SCR_V, SCR_AV, SCR_V2, SCR_AV2, SCR_R, SCR_T, SCR_W, SCR_C,
    SCR_EIG0, SCR_EIG1, SCR_EIG2, SCR_EIG3, SCR_EIG4, SCR_EIG5, SCR_EIG6, SCR_UPD0, SCR_UPD1, SCR_UPD2,
    SCR_UPD3, SCR_UPD4, SCR_QR0, SCR_QR1, SCR_STEP0, SCR_STEP1, SCR_STEP2, SCR_MISC0, SCR_MISC1, SCR_PSMALL, SCR_SYMV, SCR_NSLOTS
};

// ---- kernel launchers (kernels.hip) -------------------------------------------------------
// Epilogue of the row-panel matvec.
struct GemvEpi {
    int mode = 0;            // 0: y = alpha*acc; 1: y = alpha*acc/(dvec[i]-theta); 2: y = alpha*acc + beta*y;
                             // 3: * |dvec[i]|; 4: * dvec[i]; 5: * (1/(dvec[i]-theta) - beta)
    const double* dvec = nullptr;
    double theta = 0, alpha = 1, beta = 0;
};
// right-hand sides of the row-panel matvec as up to 8 independent vectors
struct GemvX {
    const double* p[8];
};
// Y[h*ldy + i] = epi(sum_j A[i*lda + j] * X[h*ldx + j]),  i < rows, j < cols, h < nrhs (<= 8).
// A must be 16-byte aligned with even lda; X rows must be readable up to round_up(cols, 2).
int launch_gemv_rows(sella_ctx* c, const double* A, int rows, int cols, int lda, const double* X,
                     int ldx, int nrhs, double* Y, int ldy, const GemvEpi& epi);
// same with the right-hand sides given as separate pointers (no contiguity requirement)
int launch_gemv_rows_xp(sella_ctx* c, const double* A, int rows, int cols, int lda,
                        const double* const* xs, int nrhs, double* Y, int ldy, const GemvEpi& epi);
// y = A x and y2 = A2 x in ONE launch (two row sources, one right-hand side): the speculative A t and the panel dots
// V t of the Davidson chain share a launch boundary
int launch_gemv_rows2(sella_ctx* c, const double* A, int rows, int lda, const double* A2, int rows2, int lda2, int cols,
                      const double* x, double* y, double* y2);
// Y (nrhs <= 16 rows, vector-major) = A X^T on the matrix cores; Xp: 16-row panel with leading dimension lda,
// rows beyond nrhs and the row padding zero
int launch_panel16(sella_ctx* c, const double* A, int rows, int cols, int lda, const double* Xp, int nrhs,
                   double* Y, int ldy);
int launch_panel16_marked(sella_ctx* c, const double* A, int rows, int cols, int lda, const double* Xp, int nrhs, double* Y,
                          int ldy);                  // ... and publishes the sequence word of a polled wait behind it
// |x|^2 -> out[0]; x <- x/|x|   (one single-workgroup launch)
int launch_normalize(sella_ctx* c, double* x, int n, double* out);
// Y[h*ldy + j] = sum_i A[i*lda + j] * X[h*ldx + i]   (transposed product, deterministic 2-pass)
int launch_gemv_cols(sella_ctx* c, const double* A, int rows, int cols, int lda, const double* X,
                     int ldx, int nrhs, double* Y, int ldy);
// out[cidx*ldo + i] = beta * out[..] + sum_j W1[j*ldw1 + cidx] P1[j*ldp1 + i]
//                                   + sum_j W2[j*ldw2 + cidx] P2[j*ldp2 + i]
// W1/W2 are DEVICE pointers (k1 x nout, k2 x nout); P2 may be null.
int launch_lincomb(sella_ctx* c, int n, int nout, const double* P1, int ldp1, int k1,
                   const double* W1, int ldw1, const double* P2, int ldp2, int k2,
                   const double* W2, int ldw2, double beta, double* out, int ldo);
// out[r] = sum_i P[r*ldp + i]^2
int launch_rows_sumsq(sella_ctx* c, const double* P, int ldp, int nrows, int n, double* out);
// x[i] *= f(scal[0]) : mode 0 -> 1/sqrt(s), 1 -> 1/s, 2 -> s
int launch_scale_by(sella_ctx* c, double* x, int n, const double* scal, int mode);
// t = -x + y * (dots[0]/dots[1])  (|dots[1]| < 1e-12 -> t = x), eigensolvers.py:123-139
int launch_jd_combine(sella_ctx* c, const double* x, const double* y, const double* dots,
                      double* t, int n);
// generic elementwise z = a*x + b*y (y may be null)
int launch_axpby(sella_ctx* c, int n, double a, const double* x, double b, const double* y,
                 double* z);
// 2-D: C[i*ldc + j] = a*A[i*lda + j] + b*B[i*ldb + j]
int launch_axpby2d(sella_ctx* c, int rows, int cols, double a, const double* A, int lda, double b,
                   const double* B, int ldb, double* C, int ldc);
int launch_transpose(sella_ctx* c, const double* A, int rows, int cols, int lda, double* At,
                     int ldat);
// C = alpha * op(A) op(B) + beta * C ; op(X) = X or X^T; M,N,K are the product dimensions.
int launch_gemm(sella_ctx* c, int transA, int transB, int M, int N, int K, double alpha,
                const double* A, int lda, const double* B, int ldb, double beta, double* C,
                int ldc);
// B[i][j] = 0.5*(B[i][j] + B[j][i]) in place (square)
int launch_symmetrize(sella_ctx* c, double* B, int n, int ld);
// gs.hip: orthonormalise t (n) against k orthonormal rows of a vector-major panel
int gs_orthonormalise(sella_ctx* c, const double* basis, int ldb, int k, double* t, int n,
                      double eps1, double eps2, int maxiter, int* kept, double* first_norm);
// two sweeps + normalisation without a host round trip: |t|^2 before / after sweep 1 / after sweep 2 -> scalar slots 8..10
int gs_project_twice(sella_ctx* c, const double* basis, int ldb, int k, double* t, int n);
// B <- (B + B^T)/2 + alpha * sum_a (U_a Z_a^T + Z_a U_a^T), U/Z vector-major panels with kk rows
int launch_sym_rank2k(sella_ctx* c, double* B, int n, int ld, const double* Up, const double* Zp,
                      int ldp, int kk, double alpha = 1.0);
// C <- C + alpha sum_a (U_a Z_a^T + Z_a U_a^T) for an ALREADY symmetric block, as a pure stream (no mirror tile; the two
// triangles agree to roundoff, not bitwise): the trailing update of the tridiagonalisation
int launch_rank2k_stream(sella_ctx* c, double* C, int m, int ld, const double* Up, const double* Zp, int ldp, int kk,
                         double alpha, bool upper_only = false);
int launch_mirror_upper(sella_ctx* c, double* C, int m, int ld);      // lower triangle <- transpose of the upper one
// eigh.hip: eigendecomposition (w host ascending, Vt rows / V columns, both updated in place) of
// B + sum_a (U_a Z_a^T + Z_a U_a^T) from that of B; *nrank1 = rank-one modifications applied
int eig_lowrank_update(sella_ctx* c, int n, double* w, Mat* V, Mat* Vt, const double* Up, const double* Zp,
                       int ldp, int kk, int* nrank1);
// the same for a STRUCTURED eigendecomposition: r explicit eigenpairs (mu ascending, rows of Wt) + the eigenvalue lam0
// on the orthogonal complement of their span; *r_io and mu are updated (r grows by at most one per rank-one term)
int lr_lowrank_update(sella_ctx* c, int n, int* r_io, double* mu, double lam0, Mat* Wt, const double* Up, const double* Zp,
                      int ldp, int kk, int* nrank1);
// emt.hip: sella_emt_eval with the parameter table and shift vectors optionally resident (dconst: 9 n + 3 nshift doubles)
int emt_eval_resident(sella_ctx* c, int n, const double* pos, const double* par, int nshift, const double* shifts,
                      const double* dconst, double rc, double acut, double cutoff, double beta, double* energy, double* grad);
int emt_queue(sella_ctx* c, int n, const double* pos, const double* par, int nshift, const double* shifts, const double* dconst,
              double rc, double acut, double cutoff, double beta, double** eatom, double** grad);
// calc.hip: a force call of a library calculator in two halves, so that what consumes the gradient can be queued behind
// it without a wait in between.  calc_queue: x (host) uploaded, kernels queued; *g_dev = gradient on the device (n),
// *aux_dev / *naux = what the energy is assembled from.  calc_finish: energy from the read-back aux values, after the wait.
int calc_queue(sella_calc* k, const double* x, double** g_dev, double** aux_dev, int* naux);
double calc_finish(sella_calc* k, const double* x, const double* aux_host);
// stepper.hip: step family on m modes = rows idx[0..m) of a device panel (gathered into matrices the stepper owns)
int stepper_from_panel(sella_ctx* c, int kind, const double* src, int ld, const int* idx, int m, int n, const double* ev,
                       const double* gh, int order, sella_stepper** out);
// the same without copies: the family reads the rows where they are (trust-region measure only; stepper.hip)
int stepper_on_panel(sella_ctx* c, int kind, const double* src, int ld, const int* idx, int m, int n, const double* ev,
                     const double* gh, int order, sella_stepper** out);
void stepper_panel_scale(sella_stepper* st, int mode, double factor);      // mode's row is stored unnormalised: row * factor
// the interpolating batched root search of sella_restricted_step instead of the reference's alpha schedule (sella_opt_step)
void stepper_set_fast_search(sella_stepper* st, bool on, bool boundary_hint = false);
// where the previous root search of the same saddle search ended (its first round looks there) / where this one ended
void stepper_set_alpha_hint(sella_stepper* st, double hint);
double stepper_alpha_found(const sella_stepper* st);
// lrstep.hip: the learn / adapt / propose step on structured decompositions with every decision on the device
// `pipe` (optional): the force call at the new geometry has NOT been made yet — the step queues it on the stream in front
// of the update that consumes its gradient and waits once for both (csrc/search.hip: no host round trip between force call
// and update).  g_out (n) and f receive gradient and energy; a->g_new must point at g_out, a->f_new is set here.
// done == false on return: the step did not qualify before anything was queued, the caller makes the force call itself.
struct CalcPipe {
    sella_calc* calc = nullptr;
    const double* x = nullptr;
    double* g_out = nullptr;
    double f = 0.0;
    bool done = false;
};
int lr_fused_step(sella_ctx* c, sella_opt_step_t* a, bool* handled, CalcPipe* pipe = nullptr);
// optstep.hip: sella_opt_step with the force call inside (see CalcPipe); *f_new receives the energy
int opt_step_with_calc(sella_ctx* c, sella_opt_step_t* a, sella_calc* calc, const double* x, double* g_new, double* f_new,
                       bool* force_call_made = nullptr);
// batched NN GEMM for the merges of one divide-and-conquer level: batch b multiplies the diagonal blocks
// at offset lo_b:  C[lo.., lo..] (K_b x N_b) = A[lo.., lo..] (K_b x K_b) * B[lo.., lo..] (K_b x N_b), all with
// leading dimension ld.  desc (device): 4 ints per batch {lo, N, K, unused}.
int launch_gemm_merge_batched(sella_ctx* c, int nbatch, const int* desc, int maxN, int maxK, const double* A,
                              const double* B, double* C, int ld);
// gather rows: out[r*ldo + j] = in[idx[r]*ldi + j] (idx device int array)
int launch_gather_rows(sella_ctx* c, const double* in, int ldi, const int* idx, int nrows,
                       int ncols, double* out, int ldo);

}  // namespace sella
