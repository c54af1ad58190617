// Context, caching device allocator, staging copies and HIP-event profiling.
#include <cstdarg>
#include <pthread.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "mrchip_internal.h"

namespace mrchip {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int download_1d(hipStream_t s, void *dst, const void *src, size_t bytes);

// ---- guard bands (debug switch MRCHIP_CANARY=<KiB per side>, default off) -----------------------------------------
// Kernels of this library read AND write the documented slack of their images (row padding, PAD bytes in front of row 0
// and behind the last row).  With the switch on, every block the allocator hands out carries `guard` more bytes on both
// sides filled with a pattern; mrchip_canary_check (and every dev_free, for its block) reads them back: a byte that
// changed is a write outside a block -- into what would be a neighbouring plane of the cache without the guards.
static size_t canary_bytes() {
    static const size_t g = [] {
        const char *e = getenv("MRCHIP_CANARY");
        long long kib = e ? atoll(e) : 0;
        if (kib < 0) kib = 0;
        if (kib > (1 << 16)) kib = 1 << 16;
        return (size_t)kib << 10;
    }();
    return g;
}
constexpr unsigned char CANARY_BYTE = 0xC5;

static long long canary_verify_block(mrchip_ctx *ctx, DevBlock &b) {
    if (!b.guard || !b.base) return 0;
    std::vector<unsigned char> host(b.guard);
    long long bad = 0;
    for (int side = 0; side < 2; side++) {
        unsigned char *g = (unsigned char *)b.base + (side ? b.bytes - b.guard : 0);
        if (download_1d(ctx->streams[0], host.data(), g, b.guard) != 0 || hipStreamSynchronize(ctx->streams[0]) != hipSuccess) return -1;
        long long first = -1, n = 0;
        for (size_t i = 0; i < b.guard; i++)
            if (host[i] != CANARY_BYTE) { if (first < 0) first = (long long)i; n++; }
        if (n) {
            fprintf(stderr, "mrchip canary: %lld byte(s) overwritten in the %s guard of a %zu-byte block (first at guard offset %lld, "
                            "value 0x%02x)\n", n, side ? "trailing" : "leading", b.bytes - 2 * b.guard, first, host[first]);
            (void)hipMemset(g, CANARY_BYTE, b.guard);      // report a stray write once
            bad += n;
        }
    }
    ctx->canary_bad += bad;
    return bad;
}

long long canary_check_all(mrchip_ctx *ctx) {
    long long bad = 0;
    for (auto &b : ctx->blocks) {
        const long long n = canary_verify_block(ctx, b);
        if (n < 0) return -1;
        bad += n;
    }
    return bad;
}

// MRCHIP_POISON=1 (debugging switch, default off): every block is filled with 0xDD each time the allocator hands it out
// -- a byte of an output that is still 0xDD after the call was never written on the device (no kernel store reached it, or
// the download ran before the kernel had), where stale contents of the block's previous use would pass for data.
static int poison_block(void *p, size_t bytes) {
    static const bool on = getenv("MRCHIP_POISON") && atoi(getenv("MRCHIP_POISON")) != 0;
    if (!on) return 0;
    HIP_TRY(hipMemset(p, 0xDD, bytes));
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

int dev_alloc(mrchip_ctx *ctx, size_t bytes, void **out) {
    const size_t guard = canary_bytes();
    bytes = ((std::max<size_t>(bytes, 1) + 4095) & ~(size_t)4095) + 2 * guard;     // never a zero-byte (null) block
    int best = -1;
    for (size_t i = 0; i < ctx->blocks.size(); i++) {
        DevBlock &b = ctx->blocks[i];
        if (!b.busy && b.guard == guard && b.bytes >= bytes && b.bytes <= bytes * 2 + (1 << 20))
            if (best < 0 || b.bytes < ctx->blocks[best].bytes) best = (int)i;
    }
    if (best >= 0) {
        ctx->blocks[best].busy = true;
        *out = (char *)ctx->blocks[best].base + guard;
        return poison_block(*out, ctx->blocks[best].bytes - 2 * guard);
    }
    void *p = nullptr;
    auto drop_cache = [&]() {
        for (auto &b : ctx->blocks)
            if (!b.busy && b.base) { (void)hipFree(b.base); b.base = nullptr; b.bytes = 0; }
    };
    // A large block is checked against what the device has free BEFORE hipMalloc is asked: an allocation that only
    // just fits leaves the runtime nothing for its own queues and code objects, and what a driver does past that
    // point is not something a page loop should find out (MRCHIP_HBM_RESERVE_BYTES, default 2 GiB, stays free; a negative
    // or unparsable setting falls back to the default instead of turning into a huge size_t that refuses every large
    // block, and the value is capped at the size of the device).
    if (bytes >= ((size_t)64 << 20)) {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
            size_t reserve = (size_t)2 << 30;
            if (const char *e = getenv("MRCHIP_HBM_RESERVE_BYTES")) {
                char *end = nullptr;
                const long long v = strtoll(e, &end, 10);
                if (end != e && v >= 0) reserve = (size_t)v;
            }
            reserve = std::min(reserve, tot);
            if (bytes + reserve > fr) {
                drop_cache();
                if (hipMemGetInfo(&fr, &tot) == hipSuccess && bytes + reserve > fr) {
                    set_error("device memory: %zu bytes asked, %zu of %zu free (%zu kept in reserve)", bytes, fr, tot, reserve);
                    return MRCHIP_E_NOMEM;
                }
            }
        }
    }
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        drop_cache();      // and retry once
        e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
            return MRCHIP_E_NOMEM;
        }
    }
    if (guard) {
        if (hipMemset(p, CANARY_BYTE, guard) != hipSuccess ||
            hipMemset((char *)p + bytes - guard, CANARY_BYTE, guard) != hipSuccess) {
            (void)hipFree(p);
            set_error("canary: hipMemset failed");
            return MRCHIP_E_HIP;
        }
    }
    DevBlock b;
    b.base = p; b.bytes = bytes; b.busy = true; b.guard = guard;
    ctx->blocks.push_back(b);
    *out = (char *)p + guard;
    return poison_block(*out, bytes - 2 * guard);
}

// Returned blocks stay cached for reuse (hipMalloc / hipFree synchronise the device); the cache is trimmed,
// largest idle block first, once the idle bytes exceed MRCHIP_CACHE_BYTES (default 16 GiB) so that a long
// run over many page sizes does not sit on the whole HBM.
void dev_free(mrchip_ctx *ctx, void *p) {
    static const size_t cap = getenv("MRCHIP_CACHE_BYTES") ? (size_t)atoll(getenv("MRCHIP_CACHE_BYTES")) : ((size_t)16 << 30);
    size_t idle = 0;
    for (auto &b : ctx->blocks) {
        if (b.base && (char *)b.base + b.guard == p) {
            if (b.guard) (void)canary_verify_block(ctx, b);      // (callers release scratch with their stream idle)
            b.busy = false;
        }
        if (!b.busy && b.base) idle += b.bytes;
    }
    while (idle > cap) {
        int big = -1;
        for (size_t i = 0; i < ctx->blocks.size(); i++)
            if (!ctx->blocks[i].busy && ctx->blocks[i].base && (big < 0 || ctx->blocks[i].bytes > ctx->blocks[big].bytes)) big = (int)i;
        if (big < 0) break;
        (void)hipFree(ctx->blocks[big].base);
        idle -= ctx->blocks[big].bytes;
        ctx->blocks[big].base = nullptr; ctx->blocks[big].bytes = 0;
    }
    // forget released entries
    size_t k = 0;
    for (size_t i = 0; i < ctx->blocks.size(); i++)
        if (ctx->blocks[i].base) ctx->blocks[k++] = ctx->blocks[i];
    ctx->blocks.resize(k);
}

// ---- copies between the device and ORDINARY (pageable) host memory --------------------------------------------------
// Round 6 (profiles/r06_README.md, "the copies"): with 16 or more processes on one GPU, host-buffer calls on large
// images came back wrong hundreds of times per minute -- and in two ways that no kernel can cause: (a) whole 4 KiB pages of
// the caller's result array still held the pattern the test had put there BEFORE the call (the device-to-host copy never
// wrote them, although hipStreamSynchronize had returned success), and (b) results that stayed wrong, identically, when the
// same call was repeated: tables the library uploads once per process had arrived incomplete.  Eight processes: never.
// What these transfers have in common is `hipMemcpy(2D)Async` on pageable memory, which the runtime serves by pinning the
// caller's pages on the fly.  The library no longer uses that path: every transfer whose host side is not page-locked goes
// through a pair of page-locked staging buffers of its own (per thread, 8 MiB each, double-buffered: the DMA of one chunk
// overlaps the CPU copy of the next), i.e. the only DMA the runtime ever sees is to or from memory pinned with
// hipHostMalloc.  Page-locked callers' buffers (mrchip_host_alloc: the streaming pipeline) are handed to the runtime as
// before and stay asynchronous.  A download into pageable memory therefore returns only when the data is in place.
// MRCHIP_DIRECT_PAGEABLE=1 restores the direct calls (faster for one large page, the A/B switch of tools/runs/pageable_ab.sh).
static bool host_is_pinned(const void *p) {
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof(a));
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();                   // unregistered host memory: the query fails, the error state is cleared
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

static bool direct_pageable() {
    const char *e = getenv("MRCHIP_DIRECT_PAGEABLE");          // (read per call: the stress scripts run both forms)
    return e && atoi(e) != 0;
}

namespace {
// The CPU side of a staged transfer is a memcpy between the caller's pageable array and a page-locked slot; one thread
// copies ~10 GB/s, a fifth of what the link moves.  Copies of 2 MiB or more are cut into row ranges for a small pool of
// helper threads (started on first use, never joined: they idle on a condition variable and end with the process; a
// forked child starts its own).  MRCHIP_COPY_THREADS (default 4, 1 = the calling thread alone).
struct CopyJob { unsigned char *d; size_t dpitch; const unsigned char *s; size_t spitch; size_t row_bytes; size_t rows; };
static void run_copy(const CopyJob &j) {
    if (j.dpitch == j.row_bytes && j.spitch == j.row_bytes) { memcpy(j.d, j.s, j.row_bytes * j.rows); return; }
    for (size_t r = 0; r < j.rows; r++) memcpy(j.d + r * j.dpitch, j.s + r * j.spitch, j.row_bytes);
}
struct CopyPool {
    std::mutex m;
    std::condition_variable work, done;
    std::vector<CopyJob> jobs;
    size_t next = 0;
    int running = 0;
    int nthreads = 0;
    void worker() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            work.wait(lk, [&] { return next < jobs.size(); });
            const CopyJob j = jobs[next++];
            running++;
            lk.unlock();
            run_copy(j);
            lk.lock();
            if (--running == 0 && next >= jobs.size()) done.notify_all();
        }
    }
    // the calling thread takes pieces too; returns when every piece is done (one caller at a time: `call`)
    std::mutex call;
    void run(std::vector<CopyJob> &&v) {
        std::lock_guard<std::mutex> one(call);
        std::unique_lock<std::mutex> lk(m);
        jobs = std::move(v);
        next = 0;
        work.notify_all();
        while (next < jobs.size()) {
            const CopyJob j = jobs[next++];
            running++;
            lk.unlock();
            run_copy(j);
            lk.lock();
            --running;
        }
        done.wait(lk, [&] { return running == 0; });
        jobs.clear();
        next = 0;
    }
};
static std::atomic<CopyPool *> g_pool{nullptr};
static void pool_forked_child() { g_pool.store(nullptr); }          // the parent's threads do not exist here
static CopyPool *copy_pool() {
    CopyPool *p = g_pool.load();
    if (p) return p->nthreads > 0 ? p : nullptr;
    static std::mutex make;
    std::lock_guard<std::mutex> lk(make);
    if ((p = g_pool.load())) return p->nthreads > 0 ? p : nullptr;
    static bool atfork = false;
    if (!atfork) { pthread_atfork(nullptr, nullptr, pool_forked_child); atfork = true; }
    p = new CopyPool;                    // (never freed: see above)
    int n = 4;
    if (const char *e = getenv("MRCHIP_COPY_THREADS")) n = atoi(e);
    n = std::max(1, std::min(n, 16));
    for (int i = 0; i + 1 < n; i++) {
        try { std::thread(&CopyPool::worker, p).detach(); p->nthreads++; } catch (...) { break; }
    }
    g_pool.store(p);
    return p->nthreads > 0 ? p : nullptr;
}
// rows x row_bytes between two pitched buffers, on the pool when it is worth it
static void host_copy_2d(unsigned char *d, size_t dpitch, const unsigned char *s, size_t spitch, size_t row_bytes, size_t rows) {
    const size_t total = row_bytes * rows;
    CopyPool *p = total >= ((size_t)2 << 20) ? copy_pool() : nullptr;
    if (!p) { run_copy(CopyJob{d, dpitch, s, spitch, row_bytes, rows}); return; }
    std::vector<CopyJob> v;
    if (rows == 1) {                                   // one long row: cut it into byte ranges
        const size_t parts = (size_t)p->nthreads + 1, step = ((row_bytes + parts - 1) / parts + 63) & ~(size_t)63;
        for (size_t o = 0; o < row_bytes; o += step) v.push_back(CopyJob{d + o, 0, s + o, 0, std::min(step, row_bytes - o), 1});
    } else {
        const size_t parts = std::min(rows, (size_t)(p->nthreads + 1) * 2), step = (rows + parts - 1) / parts;
        for (size_t r = 0; r < rows; r += step)
            v.push_back(CopyJob{d + r * dpitch, dpitch, s + r * spitch, spitch, row_bytes, std::min(step, rows - r)});
    }
    p->run(std::move(v));
}

constexpr size_t STAGE_BYTES = (size_t)8 << 20;
struct Staging {                       // two page-locked slots per thread; `ev[i]` = the last DMA that touched slot i
    unsigned char *buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool busy[2] = {false, false};
    int next = 0;
    // (nothing is released at thread exit: for the main thread that is process exit, when the HIP runtime may already be
    // gone; 16 MiB of page-locked memory per thread that ever moved pageable data)
    int ready() {
        for (int i = 0; i < 2; i++) {
            if (!buf[i]) HIP_TRY(hipHostMalloc((void **)&buf[i], STAGE_BYTES, hipHostMallocPortable));
            if (!ev[i]) HIP_TRY(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        }
        return 0;
    }
    int acquire(int *slot) {           // the slot used least recently, its previous DMA finished
        TRY(ready());
        const int i = next;
        next ^= 1;
        if (busy[i]) { HIP_TRY(hipEventSynchronize(ev[i])); busy[i] = false; }
        *slot = i;
        return 0;
    }
};
// one pair of slots per (thread, device): an event is recorded on streams of the device it was created on
constexpr int STAGE_MAX_DEVICES = 32;
thread_local Staging g_stages[STAGE_MAX_DEVICES];
static Staging &stage_of_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= STAGE_MAX_DEVICES) dev = 0;
    return g_stages[dev];
}
}  // namespace

// rows of `row_bytes` bytes, `rows` of them; a chunk = as many whole rows as fit a slot (a single row longer than a slot
// is cut into pieces)
int upload_2d(hipStream_t s, uint8_t *dst, int dpitch, const uint8_t *src, int spitch, int row_bytes, int rows) {
    if (rows <= 0 || row_bytes <= 0) return 0;
    if (host_is_pinned(src) || direct_pageable()) {
        HIP_TRY(hipMemcpy2DAsync(dst, dpitch, src, spitch, row_bytes, rows, hipMemcpyHostToDevice, s));
        return 0;
    }
    Staging &st = stage_of_current_device();
    if ((size_t)row_bytes > STAGE_BYTES) {
        for (int y = 0; y < rows; y++)
            for (size_t o = 0; o < (size_t)row_bytes; o += STAGE_BYTES) {
                const size_t n = std::min(STAGE_BYTES, (size_t)row_bytes - o);
                int k;
                TRY(st.acquire(&k));
                host_copy_2d(st.buf[k], n, src + (size_t)y * spitch + o, n, n, 1);
                HIP_TRY(hipMemcpyAsync(dst + (size_t)y * dpitch + o, st.buf[k], n, hipMemcpyHostToDevice, s));
                HIP_TRY(hipEventRecord(st.ev[k], s));
                st.busy[k] = true;
            }
        return 0;
    }
    const int per = (int)std::max<size_t>(1, STAGE_BYTES / (size_t)row_bytes);
    for (int y = 0; y < rows; y += per) {
        const int n = std::min(per, rows - y);
        int k;
        TRY(st.acquire(&k));
        host_copy_2d(st.buf[k], (size_t)row_bytes, src + (size_t)y * spitch, (size_t)spitch, (size_t)row_bytes, (size_t)n);
        HIP_TRY(hipMemcpy2DAsync(dst + (size_t)y * dpitch, dpitch, st.buf[k], row_bytes, row_bytes, n, hipMemcpyHostToDevice, s));
        HIP_TRY(hipEventRecord(st.ev[k], s));
        st.busy[k] = true;
    }
    return 0;
}

int upload_1d(hipStream_t s, void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return 0;
    if (host_is_pinned(src) || direct_pageable()) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
        return 0;
    }
    Staging &st = stage_of_current_device();
    for (size_t o = 0; o < bytes; o += STAGE_BYTES) {
        const size_t n = std::min(STAGE_BYTES, bytes - o);
        int k;
        TRY(st.acquire(&k));
        host_copy_2d(st.buf[k], n, (const unsigned char *)src + o, n, n, 1);          // (the caller's buffer is free again when this returns)
        HIP_TRY(hipMemcpyAsync((unsigned char *)dst + o, st.buf[k], n, hipMemcpyHostToDevice, s));
        HIP_TRY(hipEventRecord(st.ev[k], s));
        st.busy[k] = true;
    }
    return 0;
}

// pageable destination: chunk i's DMA (stream-ordered behind the kernels) runs while chunk i-1 is copied out by the CPU;
// returns with everything in place
int download_2d(hipStream_t s, uint8_t *dst, int dpitch, const uint8_t *src, int spitch, int row_bytes, int rows) {
    if (rows <= 0 || row_bytes <= 0) return 0;
    if (host_is_pinned(dst) || direct_pageable()) {
        HIP_TRY(hipMemcpy2DAsync(dst, dpitch, src, spitch, row_bytes, rows, hipMemcpyDeviceToHost, s));
        return 0;
    }
    Staging &st = stage_of_current_device();
    if ((size_t)row_bytes > STAGE_BYTES) {
        for (int y = 0; y < rows; y++)
            TRY(download_1d(s, dst + (size_t)y * dpitch, src + (size_t)y * spitch, (size_t)row_bytes));
        return 0;
    }
    const int per = (int)std::max<size_t>(1, STAGE_BYTES / (size_t)row_bytes);
    int pend_k = -1, pend_y = 0, pend_n = 0;
    auto drain = [&]() -> int {
        if (pend_k < 0) return 0;
        HIP_TRY(hipEventSynchronize(st.ev[pend_k]));
        st.busy[pend_k] = false;
        host_copy_2d(dst + (size_t)pend_y * dpitch, (size_t)dpitch, st.buf[pend_k], (size_t)row_bytes, (size_t)row_bytes, (size_t)pend_n);
        pend_k = -1;
        return 0;
    };
    for (int y = 0; y < rows; y += per) {
        const int n = std::min(per, rows - y);
        int k;
        TRY(st.acquire(&k));               // (never the pending slot: two slots, strictly alternating)
        HIP_TRY(hipMemcpy2DAsync(st.buf[k], row_bytes, src + (size_t)y * spitch, spitch, row_bytes, n, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipEventRecord(st.ev[k], s));
        st.busy[k] = true;
        TRY(drain());
        pend_k = k; pend_y = y; pend_n = n;
    }
    return drain();
}

int download_1d(hipStream_t s, void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return 0;
    if (host_is_pinned(dst) || direct_pageable()) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s));
        return 0;
    }
    Staging &st = stage_of_current_device();
    int pend_k = -1;
    size_t pend_o = 0, pend_n = 0;
    auto drain = [&]() -> int {
        if (pend_k < 0) return 0;
        HIP_TRY(hipEventSynchronize(st.ev[pend_k]));
        st.busy[pend_k] = false;
        host_copy_2d((unsigned char *)dst + pend_o, pend_n, st.buf[pend_k], pend_n, pend_n, 1);
        pend_k = -1;
        return 0;
    };
    for (size_t o = 0; o < bytes; o += STAGE_BYTES) {
        const size_t n = std::min(STAGE_BYTES, bytes - o);
        int k;
        TRY(st.acquire(&k));
        HIP_TRY(hipMemcpyAsync(st.buf[k], (const unsigned char *)src + o, n, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipEventRecord(st.ev[k], s));
        st.busy[k] = true;
        TRY(drain());
        pend_k = k; pend_o = o; pend_n = n;
    }
    return drain();
}

static hipEvent_t get_event(mrchip_ctx *ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

int prof_begin(mrchip_ctx *ctx, hipStream_t s, const char *name, double alg_bytes) {
    if (!ctx->prof) return -1;
    int idx = -1;
    for (size_t i = 0; i < ctx->prof_entries.size(); i++)
        if (ctx->prof_entries[i].name == name) { idx = (int)i; break; }
    if (idx < 0) {
        ProfEntry e;
        e.name = name;
        ctx->prof_entries.push_back(e);
        idx = (int)ctx->prof_entries.size() - 1;
    }
    ctx->prof_entries[idx].launches++;
    ctx->prof_entries[idx].alg_bytes += alg_bytes;
    ProfPending p;
    p.entry = idx;
    p.a = get_event(ctx);
    p.b = get_event(ctx);
    (void)hipEventRecord(p.a, s);
    ctx->prof_pending.push_back(p);
    return (int)ctx->prof_pending.size() - 1;
}

void prof_end(mrchip_ctx *ctx, hipStream_t s, int token) {
    if (token < 0) return;
    (void)hipEventRecord(ctx->prof_pending[token].b, s);
}

int prof_resolve(mrchip_ctx *ctx) {
    for (auto &p : ctx->prof_pending) {
        HIP_TRY(hipEventSynchronize(p.b));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, p.a, p.b));
        ctx->prof_entries[p.entry].ms += ms;
        ctx->event_pool.push_back(p.a);
        ctx->event_pool.push_back(p.b);
    }
    ctx->prof_pending.clear();
    return 0;
}

}  // namespace mrchip

using namespace mrchip;

MRCHIP_EXPORT int mrchip_abi_version(void) { return MRCHIP_ABI_VERSION; }

MRCHIP_EXPORT const char *mrchip_last_error(void) { return g_err; }

MRCHIP_EXPORT int mrchip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

MRCHIP_EXPORT mrchip_ctx *mrchip_create(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (%s); libmrchip has no CPU fallback",
                  e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return nullptr;
    }
    if (device < 0 || device >= n) {
        set_error("device %d out of range (%d devices)", device, n);
        return nullptr;
    }
    if ((e = hipSetDevice(device)) != hipSuccess) {
        set_error("hipSetDevice(%d): %s", device, hipGetErrorString(e));
        return nullptr;
    }
    mrchip_ctx *ctx = new mrchip_ctx();
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        ctx->cus = prop.multiProcessorCount;
        ctx->hbm = prop.totalGlobalMem;
        snprintf(ctx->name, sizeof(ctx->name), "%s (%s)", prop.name, prop.gcnArchName);
    }
    for (int i = 0; i < NSTREAMS; i++) {
        if ((e = hipStreamCreateWithFlags(&ctx->streams[i], hipStreamNonBlocking)) != hipSuccess) {
            set_error("hipStreamCreate: %s", hipGetErrorString(e));
            delete ctx;
            return nullptr;
        }
    }
    ctx->pinned_bytes = 1 << 20;
    if ((e = hipHostMalloc(&ctx->pinned, ctx->pinned_bytes, hipHostMallocDefault)) != hipSuccess) {
        set_error("hipHostMalloc: %s", hipGetErrorString(e));
        delete ctx;
        return nullptr;
    }
    return ctx;
}

MRCHIP_EXPORT void mrchip_destroy(mrchip_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (auto &p : ctx->prof_pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    delete ctx->host_mail;                 // (before the blocks go: its DevBufs return to the allocator first)
    ctx->host_mail = nullptr;
    delete ctx->gray_pending;
    ctx->gray_pending = nullptr;
    for (auto &b : ctx->blocks)
        if (b.base) (void)hipFree(b.base);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    for (int i = 0; i < NSTREAMS; i++)
        if (ctx->streams[i]) (void)hipStreamDestroy(ctx->streams[i]);
    delete ctx;
}

MRCHIP_EXPORT int mrchip_sync(mrchip_ctx *ctx) {
    if (!ctx) return MRCHIP_E_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    for (int i = 0; i < NSTREAMS; i++) HIP_TRY(hipStreamSynchronize(ctx->streams[i]));
    return 0;
}

// Guard bands (MRCHIP_CANARY, see dev_alloc): waits for the device, then verifies the guards of every block of the
// context; *bad_bytes = overwritten guard bytes found since the context was created (0 with the switch off).
MRCHIP_EXPORT int mrchip_canary_check(mrchip_ctx *ctx, long long *bad_bytes) {
    if (!ctx) return MRCHIP_E_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipDeviceSynchronize());
    if (canary_check_all(ctx) < 0) { set_error("canary: reading a guard band failed"); return MRCHIP_E_HIP; }
    if (bad_bytes) *bad_bytes = ctx->canary_bad;
    return 0;
}

// Proof that the guard bands see a stray write: one byte in front of and one behind a scratch block of its own are
// overwritten on purpose (inside that block's guards, nowhere else); *detected = guard bytes the verification then
// reports (2 with the switch on, 0 with it off).  The context's running count is left as it was.
MRCHIP_EXPORT int mrchip_canary_selftest(mrchip_ctx *ctx, long long *detected) {
    if (!ctx || !detected) return MRCHIP_E_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipDeviceSynchronize());
    *detected = 0;
    const size_t guard = canary_bytes();
    if (!guard) return 0;
    void *p = nullptr;
    TRY(dev_alloc(ctx, 4096, &p));
    const long long before = ctx->canary_bad;
    long long n = -1;
    hipError_t e1 = hipErrorUnknown, e2 = hipErrorUnknown;
    for (auto &b : ctx->blocks)
        if (b.base && (char *)b.base + b.guard == p) {
            // (a cached block may be larger than what was asked for: its trailing guard starts at its own end)
            e1 = hipMemset((char *)p - 1, 0, 1);
            e2 = hipMemset((char *)b.base + b.bytes - b.guard, 0, 1);
            n = canary_verify_block(ctx, b);
        }
    ctx->canary_bad = before;
    dev_free(ctx, p);
    if (e1 != hipSuccess || e2 != hipSuccess || n < 0) { set_error("canary selftest: device access failed"); return MRCHIP_E_HIP; }
    *detected = n;
    return 0;
}

MRCHIP_EXPORT int mrchip_device_info(mrchip_ctx *ctx, char *name, int name_len, int *cus, size_t *hbm_bytes) {
    if (!ctx) return MRCHIP_E_ARG;
    if (name && name_len > 0) snprintf(name, name_len, "%s", ctx->name);
    if (cus) *cus = ctx->cus;
    if (hbm_bytes) *hbm_bytes = ctx->hbm;
    return 0;
}

MRCHIP_EXPORT int mrchip_device_memory(mrchip_ctx *ctx, size_t *free_bytes, size_t *total_bytes) {
    if (!ctx) return MRCHIP_E_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    size_t fr = 0, tot = 0;
    HIP_TRY(hipMemGetInfo(&fr, &tot));
    if (free_bytes) *free_bytes = fr;
    if (total_bytes) *total_bytes = tot;
    return 0;
}

// NUMA node of the host socket the context's GPU hangs off (from its PCI address in sysfs), -1 if unknown: a caller that
// streams pages over PCIe wants its page-locked buffers -- and so the thread that allocates them -- on that node.
MRCHIP_EXPORT int mrchip_device_numa_node(mrchip_ctx *ctx) {
    if (!ctx) return -1;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), ctx->device) != hipSuccess) return -1;
    for (char *p = bus; *p; p++) if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');      // sysfs spells the address in lower case
    char path[160];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}

MRCHIP_EXPORT int mrchip_prof_enable(mrchip_ctx *ctx, int enable) {
    if (!ctx) return MRCHIP_E_ARG;
    if (!enable) TRY(prof_resolve(ctx));
    ctx->prof = enable != 0;
    return 0;
}

MRCHIP_EXPORT int mrchip_prof_reset(mrchip_ctx *ctx) {
    if (!ctx) return MRCHIP_E_ARG;
    TRY(prof_resolve(ctx));
    ctx->prof_entries.clear();
    return 0;
}

MRCHIP_EXPORT int mrchip_prof_count(mrchip_ctx *ctx) {
    if (!ctx) return MRCHIP_E_ARG;
    TRY(prof_resolve(ctx));
    return (int)ctx->prof_entries.size();
}

MRCHIP_EXPORT int mrchip_prof_get(mrchip_ctx *ctx, int i, char *name, int name_len, long long *launches,
                                  double *total_ms, double *alg_bytes) {
    if (!ctx || i < 0 || i >= (int)ctx->prof_entries.size()) return MRCHIP_E_ARG;
    const ProfEntry &e = ctx->prof_entries[i];
    if (name && name_len > 0) snprintf(name, name_len, "%s", e.name.c_str());
    if (launches) *launches = e.launches;
    if (total_ms) *total_ms = e.ms;
    if (alg_bytes) *alg_bytes = e.alg_bytes;
    return 0;
}

MRCHIP_EXPORT int mrchip_hbm_copy_bandwidth(mrchip_ctx *ctx, size_t bytes, int reps, double *gbps) {
    if (!ctx) { set_error("null context"); return MRCHIP_E_ARG; }
    HIP_TRY(hipSetDevice(ctx->device));
    if (!gbps || bytes < 4096 || reps < 1) { set_error("hbm_copy_bandwidth: bad arguments"); return MRCHIP_E_ARG; }
    hipStream_t s = ctx->streams[0];
    ScratchSync scratch_guard(ctx, s);
    DevBuf a, b;
    TRY(a.alloc(ctx, bytes));
    TRY(b.alloc(ctx, bytes));
    struct Ev {          // destroyed on every path
        hipEvent_t e = nullptr;
        ~Ev() { if (e) (void)hipEventDestroy(e); }
    } ev0, ev1;
    HIP_TRY(hipEventCreate(&ev0.e));
    HIP_TRY(hipEventCreate(&ev1.e));
    hipEvent_t e0 = ev0.e, e1 = ev1.e;
    HIP_TRY(hipMemsetAsync(a.p, 1, bytes, s));
    HIP_TRY(hipMemcpyAsync(b.p, a.p, bytes, hipMemcpyDeviceToDevice, s));      // warm-up
    HIP_TRY(hipEventRecord(e0, s));
    for (int i = 0; i < reps; i++) HIP_TRY(hipMemcpyAsync(b.p, a.p, bytes, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipEventRecord(e1, s));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    *gbps = ms > 0 ? 2.0 * (double)bytes * reps / (ms * 1e-3) / 1e9 : 0.0;
    return 0;
}

// Pinned (page-locked) host memory for the arrays the caller hands to the *_async downloads / uploads.
MRCHIP_EXPORT void *mrchip_host_alloc(mrchip_ctx *ctx, size_t bytes) {
    if (!ctx || bytes == 0) { set_error("host_alloc: bad arguments"); return nullptr; }
    if (hipSetDevice(ctx->device) != hipSuccess) { set_error("host_alloc: hipSetDevice failed"); return nullptr; }
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { set_error("host_alloc: hipHostMalloc(%zu) failed", bytes); return nullptr; }
    return p;
}

MRCHIP_EXPORT void mrchip_host_free(mrchip_ctx *ctx, void *p) {
    if (!p) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    (void)hipHostFree(p);
}
