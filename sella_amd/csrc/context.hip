// context.hip — context, device matrices, scratch pool, profiling hooks and the simple
// C-ABI entry points (uploads, products, projection).  See include/sella_hip.h.
#include "internal.h"

namespace sella {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int mat_new(sella_ctx* c, int rows, int cols, sella_mat* h) {
    if (rows < 0 || cols < 0) {
        set_error("mat_new: negative shape %d x %d", rows, cols);
        return SELLA_E_INVALID;
    }
    Mat m;
    m.rows = rows;
    m.cols = cols;
    m.ld = round_up(cols > 0 ? cols : 1, 8);
    // two spare rows so 16-byte reads that start inside the last row never leave the buffer
    const size_t bytes = ((size_t)(rows > 0 ? rows : 1) + 2) * m.ld * sizeof(double);
    SCHK(dev_alloc(c, bytes, &m.d));
    HIPCHK(s_memset0(c, m.d, bytes));
    m.live = true;
    for (size_t i = 0; i < c->mats.size(); ++i)
        if (!c->mats[i].live) {
            c->mats[i] = m;
            *h = (int)i;
            return SELLA_OK;
        }
    c->mats.push_back(m);
    *h = (int)c->mats.size() - 1;
    return SELLA_OK;
}

static void frame_swap(sella_ctx* c, sella_ctx::Frame& f) {
    std::swap(c->dscal, f.dscal);
    std::swap(c->hscal, f.hscal);
    c->scratch.swap(f.scratch);
    std::swap(c->hstage, f.hstage);
    std::swap(c->hstage_bytes, f.hstage_bytes);
    c->hbuf_a.swap(f.hbuf_a);
    c->hbuf_b.swap(f.hbuf_b);
}

int callback_enter(sella_ctx* c) {
    // No wait here: one stream per context, so whatever the callback queues through the library is ordered behind the
    // interrupted call's work, and recycled device blocks (dev_free -> dev_alloc) are reused in stream order.
    if ((int)c->frames.size() < c->depth + 2) c->frames.resize(c->depth + 2);
    frame_swap(c, c->frames[c->depth]);          // park the caller's working set ...
    c->depth += 1;
    frame_swap(c, c->frames[c->depth]);          // ... and install the one of the next depth
    if (!c->dscal) {
        if (hipMalloc((void**)&c->dscal, (size_t)c->nscal * sizeof(double)) != hipSuccess ||
            hipHostMalloc((void**)&c->hscal, (size_t)c->nscal * sizeof(double), hipHostMallocDefault) != hipSuccess) {
            set_error("allocating the scalar exchange buffers of call depth %d failed", c->depth);
            callback_leave(c);
            return SELLA_E_NOMEM;
        }
        (void)s_memset0(c, c->dscal, (size_t)c->nscal * sizeof(double));
        c->scratch.assign(SCR_NSLOTS, {nullptr, 0});
    }
    return SELLA_OK;
}

void callback_leave(sella_ctx* c) {
    // device-to-host payloads the nested calls queued but never waited for are delivered while their destinations
    // (the callback's own memory) are still alive
    if (!c->d2h_pending.empty()) (void)stream_wait(c);
    frame_swap(c, c->frames[c->depth]);
    c->depth -= 1;
    frame_swap(c, c->frames[c->depth]);
}

static size_t pool_class(size_t bytes) { return (size_t)round_up_l((long)(bytes ? bytes : 1), 4096); }

int dev_alloc(sella_ctx* c, size_t bytes, double** p) {
    const size_t cls = pool_class(bytes);
    auto it = c->pool.find(cls);
    if (it != c->pool.end() && !it->second.empty()) {
        *p = (double*)it->second.back();
        it->second.pop_back();
        return SELLA_OK;
    }
    for (auto& a : c->arenas)
        if (a.size - a.used >= cls) {
            *p = (double*)(a.base + a.used);
            a.used += cls;
            return SELLA_OK;
        }
    // new arena: 512 MiB, doubling with every arena up to 8 GiB, or the request itself if larger
    size_t want = (size_t)512 << 20;
    for (size_t i = 0; i < c->arenas.size() && want < ((size_t)8 << 30); ++i) want *= 2;
    if (want < cls) want = cls;
    char* base = nullptr;
    hipError_t e = hipMalloc((void**)&base, want);
    if (e != hipSuccess && want > cls) {
        want = cls;
        e = hipMalloc((void**)&base, want);
    }
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
        return SELLA_E_NOMEM;
    }
    c->arenas.push_back(sella_ctx::Arena{base, want, cls});
    *p = (double*)base;
    return SELLA_OK;
}

void dev_free(sella_ctx* c, double* p, size_t bytes) {
    if (!p) return;
    c->pool[pool_class(bytes)].push_back(p);
}

Mat* mat_get(sella_ctx* c, sella_mat h) {
    if (c == nullptr || h < 0 || h >= (int)c->mats.size() || !c->mats[h].live) {
        set_error("invalid matrix handle %d", h);
        return nullptr;
    }
    return &c->mats[h];
}

int scratch_get(sella_ctx* c, int slot, size_t bytes, double** p) {
    if (slot < 0 || slot >= SCR_NSLOTS) {
        set_error("bad scratch slot %d", slot);
        return SELLA_E_INVALID;
    }
    if ((int)c->scratch.size() < SCR_NSLOTS) c->scratch.resize(SCR_NSLOTS, {nullptr, 0});
    auto& s = c->scratch[slot];
    bytes = round_up_l((long)bytes + 64, 256);
    if (s.second < bytes) {
        if (s.first) {
            dev_free(c, s.first, s.second);
            s.first = nullptr;
            s.second = 0;
        }
        size_t want = bytes + bytes / 4;
        SCHK(dev_alloc(c, want, &s.first));
        s.second = want;
        HIPCHK(s_memset0(c, s.first, want));
    }
    *p = s.first;
    return SELLA_OK;
}

// Host (n x k) row-major  <->  device panel of k rows (vector-major).  The transposition is
// done on the host side of the copy (n*k doubles, negligible next to the n^2 streams).
// Host-to-device copy of CALLER memory that does not wait: a hipMemcpyAsync from pageable memory is staged
// synchronously by the runtime (20-100 us each on the optimizer's per-step paths), so the payload is copied into a
// pinned ring first and the DMA reads from there.  The source may be reused at once.  The ring is only rewound behind a
// stream synchronisation, so a slot is never overwritten while a copy that reads it is still queued.
static constexpr size_t H2D_RING_BYTES = (size_t)8 << 20;

// From 16 KB on the runtime's host-to-device copy leaves its fast path: 2.8 us per copy up to 2048 doubles, 15.2 us at
// 3072 (24 KB: every n-vector of the 1024-atom configurations), 11.8 us for the 120 KB staging block of an optimizer step
// (tools/lab/h2d_lab.hip, profiles/r05_h2d_lab.log) — where a kernel that reads the pinned slot directly (the ring is
// device-visible host memory) takes 3.5 - 5.2 us.  Payloads of at least `h2d_kernel_min` bytes go that way.
__device__ __forceinline__ void h2d_copy_vb(const VB vb, double2* __restrict__ dst, const double2* __restrict__ src, size_t n2,
                                                       double* __restrict__ dst_tail, const double* __restrict__ src_tail) {
    const size_t i = (size_t)vb.x * 256 + threadIdx.x;
    if (i < n2) dst[i] = src[i];
    if (i == 0 && dst_tail) *dst_tail = *src_tail;
}
__global__ __launch_bounds__(256) void h2d_copy_kernel(double2* __restrict__ dst, const double2* __restrict__ src, size_t n2,
                                                       double* __restrict__ dst_tail, const double* __restrict__ src_tail) { h2d_copy_vb(vb_hw(), dst, src, n2, dst_tail, src_tail); }

static int h2d_queue(sella_ctx* c, void* dst, const void* slot, size_t bytes) {
    const long kmin = c->opt.h2d_kernel_min;
    if (kmin > 0 && bytes >= (size_t)kmin && (bytes & 7) == 0 && ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(slot)) & 15) == 0) {
        const size_t n2 = bytes / 16;
        const bool tail = (bytes & 15) != 0;
        SELLA_LAUNCHB(c, h2d_copy_kernel, h2d_copy_vb, 256, dim3((unsigned)((n2 + 255) / 256 + (n2 == 0))), dim3(256), 0,
                           static_cast<double2*>(dst), static_cast<const double2*>(slot), n2,
                           tail ? static_cast<double*>(dst) + 2 * n2 : nullptr, tail ? static_cast<const double*>(slot) + 2 * n2 : nullptr);
        HIPCHK(hipGetLastError());
        return SELLA_OK;
    }
    HIPCHK(s_memcpy(c, dst, slot, bytes, hipMemcpyHostToDevice, true));
    return SELLA_OK;
}

int h2d_async(sella_ctx* c, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return SELLA_OK;
    if (!c->hring) {
        void* p = nullptr;
        HIPCHK(hipHostMalloc(&p, H2D_RING_BYTES, hipHostMallocDefault));
        c->hring = static_cast<char*>(p);
        c->hring_bytes = H2D_RING_BYTES;
        c->hring_pos = 0;
    }
    if (bytes > c->hring_bytes / 2) {                       // large payloads: the runtime's own staging
        HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
        return stream_wait(c);                              // the source may be a temporary of the caller
    }
    const size_t need = (bytes + 63) & ~(size_t)63;
    if (c->hring_pos + need > c->hring_bytes) SCHK(stream_wait(c));           // rewinds the ring
    char* slot = c->hring + c->hring_pos;
    memcpy(slot, src, bytes);
    SCHK(h2d_queue(c, dst, slot, bytes));
    c->hring_pos += need;
    return SELLA_OK;
}

// The same in two halves for payloads the caller composes itself: *slot is pinned memory to fill (zeroed here), h2d_end
// queues the transfer.  Saves the intermediate buffer and its copy (the staging block of the one-call optimizer step is
// 120 KB per step).  Payloads beyond half the ring are refused (SELLA_E_INVALID): the caller takes h2d_async.
int h2d_begin(sella_ctx* c, size_t bytes, void** slot) {
    if (!c->hring) {
        void* p = nullptr;
        HIPCHK(hipHostMalloc(&p, H2D_RING_BYTES, hipHostMallocDefault));
        c->hring = static_cast<char*>(p);
        c->hring_bytes = H2D_RING_BYTES;
        c->hring_pos = 0;
    }
    if (bytes == 0 || bytes > c->hring_bytes / 2) return SELLA_E_INVALID;
    const size_t need = (bytes + 63) & ~(size_t)63;
    if (c->hring_pos + need > c->hring_bytes) SCHK(stream_wait(c));
    *slot = c->hring + c->hring_pos;
    c->hring_pos += need;
    memset(*slot, 0, bytes);
    return SELLA_OK;
}

int h2d_end(sella_ctx* c, void* dst, const void* slot, size_t bytes) { return h2d_queue(c, dst, slot, bytes); }

// the same for any pinned (hipHostMalloc) source the caller owns and keeps unchanged until the stream has passed the copy
int h2d_pinned(sella_ctx* c, void* dst, const void* pinned_src, size_t bytes) {
    if (bytes == 0) return SELLA_OK;
    return h2d_queue(c, dst, pinned_src, bytes);
}

static constexpr size_t D2H_RING_BYTES = (size_t)8 << 20;

int d2h_async_2d(sella_ctx* c, void* dst, const void* src_dev, size_t spitch, size_t width, size_t rows) {
    const size_t bytes = width * rows;
    if (bytes == 0) return SELLA_OK;
    static const bool direct = getenv("SELLA_D2H_DIRECT") != nullptr;      // measurement knob: the runtime's own staging
    if (direct) {
        if (rows == 1 || spitch == width) HIPCHK(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
        else HIPCHK(hipMemcpy2DAsync(dst, width, src_dev, spitch, width, rows, hipMemcpyDeviceToHost, c->stream));
        return SELLA_OK;
    }
    if (!c->dring) {
        void* p = nullptr;
        HIPCHK(hipHostMalloc(&p, D2H_RING_BYTES, hipHostMallocDefault));
        c->dring = static_cast<char*>(p);
        c->dring_bytes = D2H_RING_BYTES;
        c->dring_pos = 0;
    }
    const size_t need = (bytes + 63) & ~(size_t)63;
    if (need > c->dring_bytes - c->dring_pos) {
        if (need > c->dring_bytes) {                        // a whole matrix: the runtime's own staging, in place
            if (rows == 1 || spitch == width)
                HIPCHK(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
            else
                HIPCHK(hipMemcpy2DAsync(dst, width, src_dev, spitch, width, rows, hipMemcpyDeviceToHost, c->stream));
            return SELLA_OK;
        }
        SCHK(stream_wait(c));                                // delivers what is queued and rewinds the ring
    }
    char* slot = c->dring + c->dring_pos;
    if (rows == 1 || spitch == width)
        HIPCHK(s_memcpy(c, slot, src_dev, bytes, hipMemcpyDeviceToHost, true));
    else
        HIPCHK(s_memcpy2d(c, slot, width, src_dev, spitch, width, rows, hipMemcpyDeviceToHost, true));
    c->d2h_pending.push_back({dst, slot, bytes, width, width, rows});
    c->dring_pos += need;
    return SELLA_OK;
}

int d2h_async(sella_ctx* c, void* dst, const void* src_dev, size_t bytes) {
    return d2h_async_2d(c, dst, src_dev, bytes, bytes, 1);
}

// ---- stream-ordered copies / fills (see internal.h) -------------------------------------------------------------------
// 4-byte words: every payload of the library is doubles or ints.  grid.y = row of a 2-D copy (pitches in words).
__device__ __forceinline__ void copy_words_vb(const VB vb, unsigned* __restrict__ dst, const unsigned* __restrict__ src, size_t nwords,
                                              size_t dpitch, size_t spitch) {
    const size_t i = ((size_t)vb.x * 256 + threadIdx.x) * 4;
    unsigned* d = dst + (size_t)vb.y * dpitch;
    const unsigned* s = src + (size_t)vb.y * spitch;
    if (i + 4 <= nwords && ((reinterpret_cast<uintptr_t>(d + i) | reinterpret_cast<uintptr_t>(s + i)) & 15) == 0) {
        *reinterpret_cast<uint4*>(d + i) = *reinterpret_cast<const uint4*>(s + i);
    } else {
        for (size_t q = i; q < i + 4 && q < nwords; ++q) d[q] = s[q];
    }
}
__global__ __launch_bounds__(256) void copy_words_kernel(unsigned* __restrict__ dst, const unsigned* __restrict__ src, size_t nwords,
                                                         size_t dpitch, size_t spitch) { copy_words_vb(vb_hw(), dst, src, nwords, dpitch, spitch); }
__device__ __forceinline__ void zero_words_vb(const VB vb, unsigned* __restrict__ dst, size_t nwords) {
    const size_t i = ((size_t)vb.x * 256 + threadIdx.x) * 4;
    if (i + 4 <= nwords && (reinterpret_cast<uintptr_t>(dst + i) & 15) == 0) {
        *reinterpret_cast<uint4*>(dst + i) = make_uint4(0u, 0u, 0u, 0u);
    } else {
        for (size_t q = i; q < i + 4 && q < nwords; ++q) dst[q] = 0u;
    }
}
__global__ __launch_bounds__(256) void zero_words_kernel(unsigned* __restrict__ dst, size_t nwords) { zero_words_vb(vb_hw(), dst, nwords); }

static bool cohort_copy_ok(sella_ctx* c, const void* a, const void* b, size_t bytes, size_t rows) {
    return c->cohort && cohort_in_fiber() && bytes > 0 && (bytes & 3) == 0 && bytes / 16 + 1 < ((size_t)1 << 30) && rows < 65536 &&
           ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 3) == 0;
}

hipError_t s_memcpy(sella_ctx* c, void* dst, const void* src, size_t bytes, hipMemcpyKind kind, bool host_pinned) {
    if (bytes == 0) return hipSuccess;
    if ((kind == hipMemcpyDeviceToDevice || host_pinned) && cohort_copy_ok(c, dst, src, bytes, 1)) {
        const size_t nw = bytes / 4;
        SELLA_LAUNCHB(c, copy_words_kernel, copy_words_vb, 256, dim3((unsigned)((nw + 1023) / 1024)), dim3(256), 0,
                      static_cast<unsigned*>(dst), static_cast<const unsigned*>(src), nw, (size_t)0, (size_t)0);
        return hipGetLastError();
    }
    return hipMemcpyAsync(dst, src, bytes, kind, c->stream);
}

hipError_t s_memcpy2d(sella_ctx* c, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows,
                      hipMemcpyKind kind, bool host_pinned) {
    if (width == 0 || rows == 0) return hipSuccess;
    if ((kind == hipMemcpyDeviceToDevice || host_pinned) && cohort_copy_ok(c, dst, src, width, rows) && (dpitch & 3) == 0 && (spitch & 3) == 0) {
        const size_t nw = width / 4;
        SELLA_LAUNCHB(c, copy_words_kernel, copy_words_vb, 256, dim3((unsigned)((nw + 1023) / 1024), (unsigned)rows), dim3(256), 0,
                      static_cast<unsigned*>(dst), static_cast<const unsigned*>(src), nw, dpitch / 4, spitch / 4);
        return hipGetLastError();
    }
    return hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, kind, c->stream);
}

hipError_t s_memset0(sella_ctx* c, void* dst, size_t bytes, const char* file, int line) {
    if (bytes == 0) return hipSuccess;
    if (cohort_copy_ok(c, dst, dst, bytes, 1)) {
        const size_t nw = bytes / 4;
        static const bool sites = getenv("SELLA_COHORT_TRACE") && getenv("SELLA_COHORT_TRACE")[0] == '2';
        if (sites) {                                   // (profiling aid: the call site as the launch's name)
            static thread_local std::map<std::pair<const char*, int>, std::string> names;
            std::string& nm = names[{file, line}];
            if (nm.empty()) { const char* b = strrchr(file, '/'); nm = std::string("zero_words_vb@") + (b ? b + 1 : file) + ":" + std::to_string(line); }
            cohort_launch<zero_words_vb, 256>(c, nm.c_str(), dim3((unsigned)((nw + 1023) / 1024)), dim3(256), 0, static_cast<unsigned*>(dst), nw);
            return hipGetLastError();
        }
        SELLA_LAUNCHB(c, zero_words_kernel, zero_words_vb, 256, dim3((unsigned)((nw + 1023) / 1024)), dim3(256), 0,
                      static_cast<unsigned*>(dst), nw);
        return hipGetLastError();
    }
    return hipMemsetAsync(dst, 0, bytes, c->stream);
}

// On a member fiber of a cohort a wait parks the member: the scheduler synchronises the shared stream once for everybody
// who waits (cohort.hip), which covers everything this member has queued.
int stream_sync_raw(sella_ctx* c) {
    if (c->cohort && cohort_in_fiber()) { cohort_park_wait(c); return SELLA_OK; }
    HIPCHK(hipStreamSynchronize(c->stream));
    return SELLA_OK;
}

int event_wait(sella_ctx* c, hipEvent_t ev) {
    if (c->cohort && cohort_in_fiber()) { cohort_park_wait(c); return SELLA_OK; }
    HIPCHK(hipEventSynchronize(ev));
    return SELLA_OK;
}

__global__ void poll_mark_kernel(unsigned long long* word, unsigned long long seq) {
    __threadfence_system();
    *reinterpret_cast<volatile unsigned long long*>(word) = seq;
}

int poll_mark(sella_ctx* c) {
    if (!c->poll_word) {
        void* p = nullptr;
        HIPCHK(hipHostMalloc(&p, 64, hipHostMallocDefault));
        c->poll_word = static_cast<unsigned long long*>(p);
        *c->poll_word = 0;
    }
    ++c->poll_seq;
    if (c->cohort && cohort_in_fiber()) return SELLA_OK;             // (the member parks at the wait: one synchronisation for all)
    hipLaunchKernelGGL(poll_mark_kernel, dim3(1), dim3(1), 0, c->stream, c->poll_word, c->poll_seq);
    HIPCHK(hipGetLastError());
    return SELLA_OK;
}

int poll_arm(sella_ctx* c, unsigned long long** word, unsigned long long* seq, unsigned** count) {
    *word = nullptr; *seq = 0; *count = nullptr;
    if (!c->poll_word) {
        void* p = nullptr;
        HIPCHK(hipHostMalloc(&p, 64, hipHostMallocDefault));
        c->poll_word = static_cast<unsigned long long*>(p);
        *c->poll_word = 0;
    }
    if (!c->poll_count) {
        HIPCHK(hipMalloc((void**)&c->poll_count, 64));
        HIPCHK(hipMemsetAsync(c->poll_count, 0, 64, c->stream));
    }
    ++c->poll_seq;
    if (c->cohort && cohort_in_fiber()) return SELLA_OK;
    *word = c->poll_word;
    *seq = c->poll_seq;
    *count = c->poll_count;
    return SELLA_OK;
}

int poll_wait(sella_ctx* c) {
    if (c->cohort && cohort_in_fiber()) { cohort_park_wait(c); return SELLA_OK; }
    const unsigned long long want = c->poll_seq;
    long spins = 0;
    while (__atomic_load_n(c->poll_word, __ATOMIC_ACQUIRE) != want) {
        if ((++spins & 0xfffff) == 0 && hipStreamQuery(c->stream) != hipErrorNotReady) {
            // the stream has drained (or failed) without the word arriving: fall back to the synchronisation's verdict
            HIPCHK(hipStreamSynchronize(c->stream));
            if (__atomic_load_n(c->poll_word, __ATOMIC_ACQUIRE) == want) break;
            set_error("polled wait: the stream finished without the sequence word");
            return SELLA_E_HIP;
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    return SELLA_OK;
}

int stream_wait(sella_ctx* c) {
    SCHK(stream_sync_raw(c));
    // (both streams: the rings below are shared, and a wait issued while a job is being queued on the second stream must
    // not rewind them under transfers the main stream still has queued)
    if (c->stream_main && c->stream_main != c->stream) HIPCHK(hipStreamSynchronize(c->stream_main));
    if (c->stream2 && c->stream2 != c->stream && !c->stream2_detached) HIPCHK(hipStreamSynchronize(c->stream2));
    for (const auto& p : c->d2h_pending) memcpy(p.dst, p.slot, p.bytes);
    c->d2h_pending.clear();
    c->dring_pos = 0;
    c->hring_pos = 0;                                        // nothing that reads the upload ring is still queued
    return SELLA_OK;
}

// X is n x k row-major (k vectors as columns); the device panel holds them as k rows of stride ldp.  Stream-ordered:
// returns without waiting.
int upload_panel(sella_ctx* c, const double* X, int n, int k, double* dpanel, int ldp) {
    if (k == 1) return h2d_async(c, dpanel, X, (size_t)n * sizeof(double));
    std::vector<double> tmp((size_t)k * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < k; ++j) tmp[(size_t)j * n + i] = X[(size_t)i * k + j];
    if (ldp == n) return h2d_async(c, dpanel, tmp.data(), tmp.size() * sizeof(double));
    for (int j = 0; j < k; ++j) SCHK(h2d_async(c, dpanel + (size_t)j * ldp, tmp.data() + (size_t)j * n, (size_t)n * sizeof(double)));
    return SELLA_OK;
}

int download_panel(sella_ctx* c, const double* dpanel, int ldp, int n, int k, double* X) {
    if (k == 1) {
        SCHK(d2h_async(c, X, dpanel, (size_t)n * sizeof(double)));
        return stream_wait(c);
    }
    std::vector<double> tmp((size_t)k * n);
    SCHK(d2h_async_2d(c, tmp.data(), dpanel, (size_t)ldp * sizeof(double), (size_t)n * sizeof(double), k));
    SCHK(stream_wait(c));
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < k; ++j) X[(size_t)i * k + j] = tmp[(size_t)j * n + i];
    return SELLA_OK;
}

int read_scalars(sella_ctx* c, int offset, int count) {
    if (offset < 0 || offset + count > c->nscal) {
        set_error("read_scalars: range [%d, %d) outside the exchange buffer", offset, offset + count);
        return SELLA_E_INVALID;
    }
    HIPCHK(s_memcpy(c, c->hscal + offset, c->dscal + offset, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, true));
    SCHK(stream_wait(c));
    return SELLA_OK;
}

int host_stage(sella_ctx* c, size_t bytes, void** p) {
    if (bytes > c->hstage_bytes) {
        SCHK(stream_wait(c));
        if (c->hstage) (void)hipHostFree(c->hstage);
        c->hstage = nullptr;
        c->hstage_bytes = 0;
        const size_t want = std::max(bytes + bytes / 2, (size_t)1 << 20);
        HIPCHK(hipHostMalloc(&c->hstage, want, hipHostMallocDefault));
        c->hstage_bytes = want;
    }
    *p = c->hstage;
    return SELLA_OK;
}

double* scal_out(sella_ctx* c, int offset) { return (c->opt.host_scalars ? c->hscal : c->dscal) + offset; }

int sync_scalars(sella_ctx* c, int offset, int count) {
    if (!c->opt.host_scalars) return read_scalars(c, offset, count);
    SCHK(stream_wait(c));
    return SELLA_OK;
}

void prof_begin(sella_ctx* c, int kind, double bytes, double flops) {
    if (!c->prof) return;
    ProfPending p;
    p.kind = kind;
    p.bytes = bytes;
    p.flops = flops;
    if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
    c->prof_a = p.a;
    c->prof_b = p.b;
    c->pending.push_back(p);
}

void prof_end(sella_ctx* c) {
    if (!c->prof || !c->prof_a) return;
    c->prof_a = c->prof_b = nullptr;
    if (c->pending.size() >= 4096) (void)prof_flush(c);
}

int prof_flush(sella_ctx* c) {
    if (c->pending.empty()) return SELLA_OK;
    SCHK(stream_wait(c));
    for (auto& p : c->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            ProfSlot& s = c->slots[p.kind];
            s.launches += 1;
            s.ms += ms;
            s.bytes += p.bytes;
            s.flops += p.flops;
        }
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    c->pending.clear();
    return SELLA_OK;
}

}  // namespace sella

using namespace sella;

extern "C" {

const char* sella_last_error(void) { return g_err; }
const char* sella_version(void) { return "sella_hip 0.1 (gfx950, fp64)"; }

int sella_device_count(int* count) {
    if (!count) return SELLA_E_INVALID;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return SELLA_E_NODEVICE;
    }
    *count = n;
    return SELLA_OK;
}

int sella_ctx_create(int device, sella_ctx** out) {
    if (!out) return SELLA_E_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device visible: libsella_hip has no CPU fallback");
        return SELLA_E_NODEVICE;
    }
    if (device < 0 || device >= n) {
        set_error("device %d out of range (%d visible)", device, n);
        return SELLA_E_INVALID;
    }
    HIPCHK(hipSetDevice(device));
    sella_ctx* c = new sella_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        // (the marketing name is empty on some driver stacks: the architecture string always identifies the part)
        snprintf(c->name, sizeof(c->name), "%s (%s)", prop.name[0] ? prop.name : "AMD Instinct", prop.gcnArchName);
        c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    // A stream of its own per context, NOT synchronising with the legacy default stream: with several contexts per
    // process (ensemble members on host threads) a blocking stream would serialise behind every other context's
    // implicit default-stream work.  SELLA_STREAM_BLOCKING=1 restores hipStreamCreate's default for comparison.
    const char* sb = getenv("SELLA_STREAM_BLOCKING");
    hipError_t e = (sb && sb[0] == '1') ? hipStreamCreate(&c->stream)
                                        : hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        delete c;
        return SELLA_E_HIP;
    }
    c->stream_main = c->stream;
    c->nscal = DS_TOTAL;
    if (hipMalloc((void**)&c->dscal, (size_t)c->nscal * sizeof(double)) != hipSuccess ||
        hipHostMalloc((void**)&c->hscal, (size_t)c->nscal * sizeof(double), hipHostMallocDefault) != hipSuccess) {
        set_error("allocating the scalar exchange buffers failed");
        delete c;
        return SELLA_E_NOMEM;
    }
    (void)s_memset0(c, c->dscal, (size_t)c->nscal * sizeof(double));
    c->scratch.resize(SCR_NSLOTS, {nullptr, 0});
    *out = c;
    return SELLA_OK;
}

int sella_ctx_destroy(sella_ctx* c) {
    if (!c) return SELLA_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)prof_flush(c);
    for (auto& a : c->arenas) (void)hipFree(a.base);       // matrices, panels and scratch all live in the arenas
    if (c->dscal) (void)hipFree(c->dscal);
    if (c->hscal) (void)hipHostFree(c->hscal);
    if (c->hstage) (void)hipHostFree(c->hstage);
    for (auto& f : c->frames) {                            // parked working sets of deeper call levels
        if (f.dscal) (void)hipFree(f.dscal);
        if (f.hscal) (void)hipHostFree(f.hscal);
        if (f.hstage) (void)hipHostFree(f.hstage);
    }
    if (c->hring) (void)hipHostFree(c->hring);
    if (c->dring) (void)hipHostFree(c->dring);
    if (c->poll_word) (void)hipHostFree(c->poll_word);
    if (c->poll_count) (void)hipFree(c->poll_count);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    (void)hipStreamDestroy(c->stream_main ? c->stream_main : c->stream);
    delete c;
    return SELLA_OK;
}

int sella_ctx_sync(sella_ctx* c) {
    if (!c) return SELLA_E_INVALID;
    SCHK(stream_wait(c));
    return SELLA_OK;
}

int sella_ctx_device_name(sella_ctx* c, char* buf, int buflen) {
    if (!c || !buf || buflen <= 0) return SELLA_E_INVALID;
    snprintf(buf, buflen, "%s", c->name);
    return SELLA_OK;
}

int sella_ctx_set_option(sella_ctx* c, const char* key, long value) {
    if (!c || !key) return SELLA_E_INVALID;
    if (!strcmp(key, "gemv_rw")) {
        if (value != 0 && value != 1 && value != 2 && value != 4) { set_error("gemv_rw must be 0, 1, 2 or 4"); return SELLA_E_INVALID; }
        c->opt.gemv_rw = value;
    } else if (!strcmp(key, "gemm_mfma")) c->opt.gemm_mfma = value ? 1 : 0;
    else if (!strcmp(key, "rank2k_stream")) c->opt.rank2k_stream = value ? 1 : 0;
    else if (!strcmp(key, "eigh_wy_rows")) c->opt.eigh_wy_rows = value;
    else if (!strcmp(key, "eigh_wy_nb64_min")) c->opt.eigh_wy_nb64_min = value;
    else if (!strcmp(key, "dav_fuse_scale")) c->opt.dav_fuse_scale = value ? 1 : 0;
    else if (!strcmp(key, "dav_zero_copy")) c->opt.dav_zero_copy = value ? 1 : 0;
    else if (!strcmp(key, "dav_poll")) c->opt.dav_poll = value ? 1 : 0;
    else if (!strcmp(key, "rs_poll")) c->opt.rs_poll = value ? 1 : 0;
    else if (!strcmp(key, "rs_hint")) c->opt.rs_hint = value ? 1 : 0;
    else if (!strcmp(key, "eigh_tail_lds")) c->opt.eigh_tail_lds = value < 0 ? 0 : value;
    else if (!strcmp(key, "eigh_wy_waves")) c->opt.eigh_wy_waves = value;
    else if (!strcmp(key, "lr_cholqr")) c->opt.lr_cholqr = value ? 1 : 0;
    else if (!strcmp(key, "rank2k_fixed")) c->opt.rank2k_fixed = value ? 1 : 0;
    else if (!strcmp(key, "eigh_wy_strip")) c->opt.eigh_wy_strip = value;
    else if (!strcmp(key, "gs_small")) c->opt.gs_small = value < 0 ? 0 : (value > 2048 ? 2048 : value);
    else if (!strcmp(key, "h2d_kernel_min")) c->opt.h2d_kernel_min = value < 0 ? 0 : value;
    else if (!strcmp(key, "eigh_dc_pipeline")) c->opt.eigh_dc_pipeline = value ? 1 : 0;
    else if (!strcmp(key, "eigh_gemv_flat")) c->opt.eigh_gemv_flat = value ? 1 : 0;
    else if (!strcmp(key, "lr_dev")) c->opt.lr_dev = value ? 1 : 0;
    else if (!strcmp(key, "lr_chain")) c->opt.lr_chain = value ? 1 : 0;
    else if (!strcmp(key, "lr_pipe")) c->opt.lr_pipe = value ? 1 : 0;
    else if (!strcmp(key, "eigh_wy_overlap")) c->opt.eigh_wy_overlap = value ? 1 : 0;
    // (one launch per column: at most 1024 workgroups of at most 8 rows, eigh.hip TRD_UPD_MAXGRID — larger blocks stay
    //  with the blocked chain instead of failing in the middle of a factorisation)
    else if (!strcmp(key, "eigh_upd_max")) c->opt.eigh_upd_max = value < 0 ? 0 : (value > 8 * 1024 - 64 ? 8 * 1024 - 64 : value);
    else if (!strcmp(key, "eigh_upd_rows")) {
        if (value != 0 && value != 2 && value != 4 && value != 8) { set_error("eigh_upd_rows must be 0, 2, 4 or 8"); return SELLA_E_INVALID; }
        c->opt.eigh_upd_rows = value;
    }
    else if (!strcmp(key, "eigh_upd_nt")) {
        if (value != 128 && value != 256 && value != 512) { set_error("eigh_upd_nt must be 128, 256 or 512"); return SELLA_E_INVALID; }
        c->opt.eigh_upd_nt = value;
    }
    else if (!strcmp(key, "eigh_upd_r4_min")) c->opt.eigh_upd_r4_min = value < 0 ? 0 : value;
    else if (!strcmp(key, "eigh_upd_r8_min")) c->opt.eigh_upd_r8_min = value < 0 ? 0 : value;
    else if (!strcmp(key, "emt_hcap")) c->opt.emt_hcap = value;
    else if (!strcmp(key, "lr_overlap")) c->opt.lr_overlap = value ? 1 : 0;
    else if (!strcmp(key, "rs_batch_result")) c->opt.rs_batch_result = value ? 1 : 0;
    else if (!strcmp(key, "rs_fast")) c->opt.rs_fast = value ? 1 : 0;
    else if (!strcmp(key, "rs_batch")) c->opt.rs_batch = value ? 1 : 0;
    else if (!strcmp(key, "bd_pipeline")) c->opt.bd_pipeline = value ? 1 : 0;
    else if (!strcmp(key, "bd_early_matvec")) c->opt.bd_early_matvec = value ? 1 : 0;
    else if (!strcmp(key, "dav_rotate_fused")) c->opt.dav_rotate_fused = value ? 1 : 0;
    else if (!strcmp(key, "panel_small")) c->opt.panel_small = value > 0 ? value : 0;
    else if (!strcmp(key, "eigh_leaf")) {
        if (value < 2 || value > 64) { set_error("eigh_leaf must be in [2, 64]"); return SELLA_E_INVALID; }
        c->opt.eigh_leaf = value;
    } else if (!strcmp(key, "host_scalars")) {
        c->opt.host_scalars = value ? 1 : 0;
    } else if (!strcmp(key, "gemm_tile128")) {
        c->opt.gemm_tile128 = value ? 1 : 0;
    } else if (!strcmp(key, "panel_mfma")) {
        c->opt.panel_mfma = value ? 1 : 0;
    } else if (!strcmp(key, "panel_rows")) {
        if (value != 0 && value != 16 && value != 32 && value != 48 && value != 64) { set_error("panel_rows must be 0, 16, 32, 48 or 64"); return SELLA_E_INVALID; }
        c->opt.panel_rows = value;
    } else if (!strcmp(key, "eigh_wy_mfma")) {
        c->opt.eigh_wy_mfma = value ? 1 : 0;
    } else if (!strcmp(key, "eigh_symv_tri")) {
        c->opt.eigh_symv_tri = value ? 1 : 0;
    } else if (!strcmp(key, "eigh_symv_tr")) {
        if (value != 64 && value != 128) { set_error("eigh_symv_tr must be 64 or 128"); return SELLA_E_INVALID; }
        c->opt.eigh_symv_tr = value;
    } else if (!strcmp(key, "eigh_symv_min")) {
        if (value < 0) { set_error("eigh_symv_min must be >= 0"); return SELLA_E_INVALID; }
        c->opt.eigh_symv_min = value;
    } else if (!strcmp(key, "eigh_nb")) {
        if (value < 1 || value > 64) { set_error("eigh_nb must be in [1, 64]"); return SELLA_E_INVALID; }
        c->opt.eigh_nb = value;
    } else {
        set_error("unknown option '%s'", key);
        return SELLA_E_INVALID;
    }
    return SELLA_OK;
}

// ---- matrices ---------------------------------------------------------------------------
int sella_mat_alloc(sella_ctx* c, int rows, int cols, sella_mat* h) {
    if (!c || !h) return SELLA_E_INVALID;
    return mat_new(c, rows, cols, h);
}

int sella_mat_set(sella_ctx* c, sella_mat h, const double* A) {
    Mat* m = mat_get(c, h);
    if (!m || !A) return SELLA_E_INVALID;
    if (m->rows == 0 || m->cols == 0) return SELLA_OK;
    HIPCHK(hipMemcpy2DAsync(m->d, (size_t)m->ld * sizeof(double), A, (size_t)m->cols * sizeof(double),
                            (size_t)m->cols * sizeof(double), m->rows, hipMemcpyHostToDevice, c->stream));
    SCHK(stream_wait(c));
    return SELLA_OK;
}

int sella_mat_upload(sella_ctx* c, const double* A, int rows, int cols, sella_mat* h) {
    if (!c || !A || !h) return SELLA_E_INVALID;
    SCHK(mat_new(c, rows, cols, h));
    int st = sella_mat_set(c, *h, A);
    if (st != SELLA_OK) { sella_mat_free(c, *h); *h = SELLA_NO_MAT; }
    return st;
}

int sella_mat_download(sella_ctx* c, sella_mat h, double* out) {
    Mat* m = mat_get(c, h);
    if (!m || !out) return SELLA_E_INVALID;
    if (m->rows == 0 || m->cols == 0) return SELLA_OK;
    SCHK(d2h_async_2d(c, out, m->d, (size_t)m->ld * sizeof(double), (size_t)m->cols * sizeof(double), m->rows));
    return stream_wait(c);
}

int sella_mat_shape(sella_ctx* c, sella_mat h, int* rows, int* cols) {
    Mat* m = mat_get(c, h);
    if (!m) return SELLA_E_INVALID;
    if (rows) *rows = m->rows;
    if (cols) *cols = m->cols;
    return SELLA_OK;
}

int sella_mat_free(sella_ctx* c, sella_mat h) {
    Mat* m = mat_get(c, h);
    if (!m) return SELLA_E_INVALID;
    dev_free(c, m->d, ((size_t)(m->rows > 0 ? m->rows : 1) + 2) * m->ld * sizeof(double));
    *m = Mat();
    return SELLA_OK;
}

int sella_mat_copy(sella_ctx* c, sella_mat src, sella_mat* dst) {
    Mat* s = mat_get(c, src);
    if (!s || !dst) return SELLA_E_INVALID;
    const int rows = s->rows, cols = s->cols;
    SCHK(mat_new(c, rows, cols, dst));
    s = mat_get(c, src);
    Mat* d = mat_get(c, *dst);
    return launch_axpby2d(c, rows, cols, 1.0, s->d, s->ld, 0.0, nullptr, 0, d->d, d->ld);
}

// A[i][i] += alpha (square or not: the leading min(rows, cols) diagonal entries)
__device__ __forceinline__ void add_diag_vb(const VB vb, double* __restrict__ A, int ld, int n, double alpha) {
    const int i = vb.x * 256 + threadIdx.x;
    if (i < n) A[(size_t)i * ld + i] += alpha;
}
__global__ __launch_bounds__(1024) void add_diag_kernel(double* __restrict__ A, int ld, int n, double alpha) { add_diag_vb(vb_hw(), A, ld, n, alpha); }

int sella_mat_add_diag(sella_ctx* c, sella_mat h, double alpha) {
    Mat* m = mat_get(c, h);
    if (!m) return SELLA_E_INVALID;
    const int n = std::min(m->rows, m->cols);
    if (n == 0) return SELLA_OK;
    SELLA_LAUNCHB(c, add_diag_kernel, add_diag_vb, 1024, dim3((n + 255) / 256), dim3(256), 0, m->d, m->ld, n, alpha);
    HIPCHK(hipGetLastError());
    return SELLA_OK;
}

// dst rows [0, nrows) <- src rows [0, nrows)   (same number of columns; growing a row buffer)
int sella_mat_copy_into(sella_ctx* c, sella_mat src, sella_mat dst, int nrows) {
    Mat *s = mat_get(c, src), *d = mat_get(c, dst);
    if (!s || !d || s->cols != d->cols || nrows < 0 || nrows > s->rows || nrows > d->rows) {
        set_error("mat_copy_into: shapes do not match");
        return SELLA_E_INVALID;
    }
    if (nrows == 0) return SELLA_OK;
    return launch_axpby2d(c, nrows, s->cols, 1.0, s->d, s->ld, 0.0, nullptr, 0, d->d, d->ld);
}

// new matrix holding rows [row0, row0 + nrows) of src
int sella_mat_rows(sella_ctx* c, sella_mat src, int row0, int nrows, sella_mat* dst) {
    Mat* s = mat_get(c, src);
    if (!s || !dst || row0 < 0 || nrows < 0 || row0 + nrows > s->rows) {
        set_error("mat_rows: row range outside the matrix");
        return SELLA_E_INVALID;
    }
    const int cols = s->cols;
    SCHK(mat_new(c, nrows, cols, dst));
    if (nrows == 0) return SELLA_OK;
    s = mat_get(c, src);
    Mat* d = mat_get(c, *dst);
    return launch_axpby2d(c, nrows, cols, 1.0, s->d + (size_t)row0 * s->ld, s->ld, 0.0, nullptr, 0, d->d, d->ld);
}

int sella_mat_transpose(sella_ctx* c, sella_mat src, sella_mat* dst) {
    Mat* s = mat_get(c, src);
    if (!s || !dst) return SELLA_E_INVALID;
    const int rows = s->rows, cols = s->cols;
    SCHK(mat_new(c, cols, rows, dst));
    s = mat_get(c, src);
    Mat* d = mat_get(c, *dst);
    return launch_transpose(c, s->d, rows, cols, s->ld, d->d, d->ld);
}

int sella_mat_axpby(sella_ctx* c, double alpha, sella_mat A, double beta, sella_mat B, sella_mat* C) {
    Mat* a = mat_get(c, A);
    if (!a || !C) return SELLA_E_INVALID;
    const int rows = a->rows, cols = a->cols;
    if (B != SELLA_NO_MAT) {
        Mat* b = mat_get(c, B);
        if (!b) return SELLA_E_INVALID;
        if (b->rows != rows || b->cols != cols) { set_error("axpby: shape mismatch"); return SELLA_E_INVALID; }
    }
    SCHK(mat_new(c, rows, cols, C));
    a = mat_get(c, A);
    Mat* b = (B != SELLA_NO_MAT) ? mat_get(c, B) : nullptr;
    Mat* o = mat_get(c, *C);
    return launch_axpby2d(c, rows, cols, alpha, a->d, a->ld, beta, b ? b->d : nullptr, b ? b->ld : 0,
                          o->d, o->ld);
}

// ---- products ---------------------------------------------------------------------------
int sella_symm_mm(sella_ctx* c, sella_mat A, const double* X, int k, double* Y) {
    Mat* a = mat_get(c, A);
    if (!a || !X || !Y || k <= 0) return SELLA_E_INVALID;
    const int rows = a->rows, cols = a->cols;
    const int ldx = round_up(cols, 8), ldy = round_up(rows, 8);
    double *dx, *dy;
    SCHK(scratch_get(c, SCR_X, (size_t)k * ldx * sizeof(double), &dx));
    SCHK(scratch_get(c, SCR_Y, (size_t)k * ldy * sizeof(double), &dy));
    a = mat_get(c, A);
    if (k > 8 && c->opt.panel_mfma && a->ld == ldx) {
        // block product: 16 right-hand sides per pass over the matrix (kernels.hip, panel16_mfma_kernel)
        const int kpad = round_up(k, 16);
        SCHK(scratch_get(c, SCR_X, (size_t)kpad * ldx * sizeof(double), &dx));
        HIPCHK(s_memset0(c, dx, (size_t)kpad * ldx * sizeof(double)));
        SCHK(upload_panel(c, X, cols, k, dx, ldx));
        a = mat_get(c, A);
        for (int h0 = 0; h0 < k; h0 += 16)
            SCHK(launch_panel16(c, a->d, rows, cols, a->ld, dx + (size_t)h0 * ldx, std::min(16, k - h0),
                                dy + (size_t)h0 * ldy, ldy));
        return download_panel(c, dy, ldy, rows, k, Y);
    }
    SCHK(upload_panel(c, X, cols, k, dx, ldx));
    a = mat_get(c, A);
    SCHK(launch_gemv_rows(c, a->d, rows, cols, a->ld, dx, ldx, k, dy, ldy, GemvEpi()));
    return download_panel(c, dy, ldy, rows, k, Y);
}

int sella_gemm_tn_host(sella_ctx* c, sella_mat A, const double* X, int k, double* Y) {
    Mat* a = mat_get(c, A);
    if (!a || !X || !Y || k <= 0) return SELLA_E_INVALID;
    const int rows = a->rows, cols = a->cols;
    const int ldx = round_up(rows, 8), ldy = round_up(cols, 8);
    double *dx, *dy;
    SCHK(scratch_get(c, SCR_X, (size_t)k * ldx * sizeof(double), &dx));
    SCHK(scratch_get(c, SCR_Y, (size_t)k * ldy * sizeof(double), &dy));
    SCHK(upload_panel(c, X, rows, k, dx, ldx));
    a = mat_get(c, A);
    SCHK(launch_gemv_cols(c, a->d, rows, cols, a->ld, dx, ldx, k, dy, ldy));
    return download_panel(c, dy, ldy, cols, k, Y);
}

int sella_gemm(sella_ctx* c, int transA, int transB, double alpha, sella_mat A, sella_mat B, double beta,
               sella_mat C) {
    Mat *a = mat_get(c, A), *b = mat_get(c, B), *o = mat_get(c, C);
    if (!a || !b || !o) return SELLA_E_INVALID;
    const int M = transA ? a->cols : a->rows, K = transA ? a->rows : a->cols;
    const int Kb = transB ? b->cols : b->rows, N = transB ? b->rows : b->cols;
    if (K != Kb || o->rows != M || o->cols != N) {
        set_error("gemm: shape mismatch (%d x %d)(%d x %d) -> %d x %d", M, K, Kb, N, o->rows, o->cols);
        return SELLA_E_INVALID;
    }
    return launch_gemm(c, transA, transB, M, N, K, alpha, a->d, a->ld, b->d, b->ld, beta, o->d, o->ld);
}

int sella_project_dev(sella_ctx* c, sella_mat H, sella_mat U, sella_mat* out) {
    Mat *h = mat_get(c, H), *u = mat_get(c, U);
    if (!h || !u || !out) return SELLA_E_INVALID;
    const int n = h->rows, m = u->cols;
    if (h->cols != n || u->rows != n) { set_error("project: H must be n x n and U n x m"); return SELLA_E_INVALID; }
    // T = H U (n x m), out = U^T T (m x m)
    double* T;
    const int ldt = round_up(m, 8);
    SCHK(scratch_get(c, SCR_MISC0, (size_t)n * ldt * sizeof(double), &T));
    SCHK(mat_new(c, m, m, out));
    h = mat_get(c, H);
    u = mat_get(c, U);
    Mat* o = mat_get(c, *out);
    SCHK(launch_gemm(c, 0, 0, n, m, n, 1.0, h->d, h->ld, u->d, u->ld, 0.0, T, ldt));
    SCHK(launch_gemm(c, 1, 0, m, m, n, 1.0, u->d, u->ld, T, ldt, 0.0, o->d, o->ld));
    return SELLA_OK;
}

int sella_project(sella_ctx* c, sella_mat H, const double* U, int m, double* out) {
    Mat* h = mat_get(c, H);
    if (!h || !U || !out || m <= 0) return SELLA_E_INVALID;
    sella_mat hU = SELLA_NO_MAT, hO = SELLA_NO_MAT;
    SCHK(sella_mat_upload(c, U, h->rows, m, &hU));
    int st = sella_project_dev(c, H, hU, &hO);
    if (st == SELLA_OK) st = sella_mat_download(c, hO, out);
    sella_mat_free(c, hU);
    if (hO != SELLA_NO_MAT) sella_mat_free(c, hO);
    return st;
}

// ---- profiling ----------------------------------------------------------------------------
// the context's HIP stream and the raw address of a resident matrix: what a collective library (RCCL, bound from the
// host language without PyTorch) needs to work on the library's own buffers in stream order
int sella_ctx_stream(sella_ctx* c, void** stream) {
    if (!c || !stream) return SELLA_E_INVALID;
    *stream = (void*)c->stream;
    return SELLA_OK;
}

int sella_mat_ptr(sella_ctx* c, sella_mat h, void** dptr, int* ld) {
    if (!c || !dptr) return SELLA_E_INVALID;
    Mat* m = mat_get(c, h);
    if (!m) return SELLA_E_INVALID;
    *dptr = (void*)m->d;
    if (ld) *ld = m->ld;
    return SELLA_OK;
}

// raw copies for buffers handed to callbacks (the all-gather of sella_davidson_block): kind 0 = device -> device,
// 1 = device -> host, 2 = host -> device; synchronous at the boundary
int sella_dev_copy(sella_ctx* c, void* dst, const void* src, size_t bytes, int kind) {
    if (!c || !dst || !src || kind < 0 || kind > 2) return SELLA_E_INVALID;
    const hipMemcpyKind k = kind == 0 ? hipMemcpyDeviceToDevice : kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice;
    HIPCHK(hipMemcpyAsync(dst, src, bytes, k, c->stream));
    SCHK(stream_wait(c));
    return SELLA_OK;
}

int sella_prof_enable(sella_ctx* c, int on) {
    if (!c) return SELLA_E_INVALID;
    if (!on) SCHK(prof_flush(c));
    c->prof = on != 0;
    return SELLA_OK;
}

int sella_prof_reset(sella_ctx* c) {
    if (!c) return SELLA_E_INVALID;
    SCHK(prof_flush(c));
    for (int k = 0; k < PROF_NKIND; ++k) c->slots[k] = ProfSlot();
    return SELLA_OK;
}

int sella_prof_get(sella_ctx* c, int kind, long* launches, double* total_ms, double* total_bytes,
                   double* total_flops) {
    if (!c || kind < 0 || kind >= PROF_NKIND) return SELLA_E_INVALID;
    SCHK(prof_flush(c));
    const ProfSlot& s = c->slots[kind];
    if (launches) *launches = s.launches;
    if (total_ms) *total_ms = s.ms;
    if (total_bytes) *total_bytes = s.bytes;
    if (total_flops) *total_flops = s.flops;
    return SELLA_OK;
}

}  // extern "C"
