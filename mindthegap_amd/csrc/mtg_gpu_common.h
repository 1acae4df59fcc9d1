/*
 * mtg_gpu_common.h -- what the HIP translation units of libmtgfill.so share on the host side: the error macro, the device selection,
 * owning device buffers (timed: an index construction reports what hipMalloc / hipFree cost it), views on a workspace's cached buffers.
 * Included by mtg_gpu_build.hip (index construction), mtg_gpu_fill.hip (the fill kernels and device_run) and mtg_gpu_misc.hip (queries,
 * scan, alignments, the tool's formatter, the C ABI's device side).  Device code shared between them lives in the mtg_*.h headers.
 */
#ifndef MTG_GPU_COMMON_H
#define MTG_GPU_COMMON_H
#include "mtg_internal.h"
#include "mtg_marshal.h"
#include <hip/hip_runtime.h>
#include <chrono>
#include <thread>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace mtg;

namespace mtgi {

void stats_store(const mtg_batch_stats& s);

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);       \
            return (e_ == hipErrorOutOfMemory) ? MTG_ERR_NOMEM : MTG_ERR_NO_DEVICE;                     \
        }                                                                                               \
    } while (0)

static int ensure_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (%s); libmtgfill has no CPU fallback", e == hipSuccess ? "0 devices" : hipGetErrorString(e));
        return MTG_ERR_NO_DEVICE;
    }
    return MTG_OK;
}

/* the HIP device is a per-thread setting: a call on an index runs on the index's device whatever thread makes it */
static int use_device_of(const mtg_index* idx)
{
    if (int rc = ensure_device()) return rc;
    if (idx) HIP_TRY(hipSetDevice(idx->device));
    return MTG_OK;
}

namespace {
/* owning device buffer: freed on every exit path */
/* wall time this thread has spent in hipMalloc / hipFree (an index construction reports it: BuildProf) */
inline thread_local double tl_alloc_ms = 0;
struct AllocTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~AllocTimer() { tl_alloc_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
static hipError_t timed_malloc(void** p, size_t bytes)
{
    const bool dbg = tune::on(tune::T_DEBUG_TIMERS);
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e;
    { AllocTimer t; e = hipMalloc(p, bytes); }
    if (dbg && bytes > ((size_t)64 << 20)) fprintf(stderr, "  [alloc] hipMalloc %.2f GB: %.1f ms\n", bytes / 1e9, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return e;
}
static hipError_t timed_free(void* p)
{
    const bool dbg = tune::on(tune::T_DEBUG_TIMERS);
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e;
    { AllocTimer t; e = hipFree(p); }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (dbg && ms > 5.0) fprintf(stderr, "  [alloc] hipFree: %.1f ms\n", ms);
    return e;
}
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    DevBuf() {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)timed_free(p); }
    hipError_t alloc(size_t bytes) { if (p) { (void)timed_free(p); p = nullptr; cap = 0; } const hipError_t e = timed_malloc(&p, bytes ? bytes : 8); if (e == hipSuccess) cap = bytes ? bytes : 8; return e; }
    void* release() { void* q = p; p = nullptr; cap = 0; return q; }
    /* takes over the memory of another buffer */
    void adopt(DevBuf& o) { if (p) (void)timed_free(p); p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; }
    template <typename T> T* as() { return (T*)p; }
};
} // namespace

namespace {
/* a device buffer of one call: the cached slot of a workspace when the call belongs to a batch (grow-only: no hipMalloc / hipFree in the
 * steady state -- hipFree waits for the whole device, i.e. for every other batch in flight), memory of its own otherwise */
struct CallBuf {
    void* p = nullptr;
    bool own = false;
    hipError_t alloc(Workspace* ws, int slot, size_t bytes)
    {
        bytes = bytes ? bytes : 8;
        if (!ws) { own = true; return hipMalloc(&p, bytes); }
        if (ws->cap[slot] < bytes) {
            if (ws->ptr[slot]) (void)hipFree(ws->ptr[slot]);
            ws->ptr[slot] = nullptr; ws->cap[slot] = 0;
            const size_t want = bytes + bytes / 4 + 4096;
            const hipError_t e = hipMalloc(&ws->ptr[slot], want);
            if (e != hipSuccess) return e;
            ws->cap[slot] = want;
        }
        p = ws->ptr[slot];
        return hipSuccess;
    }
    ~CallBuf() { if (own && p) (void)hipFree(p); }
    template <typename T> T* as() { return (T*)p; }
};
enum { CALL_SLOT0 = 23 }; /* workspace slots 0 .. 22 belong to device_run; a batch's later calls (k_query, k_nw) use 23 .. 30, one call at a time */

} // namespace

namespace {
/* a view on a cached, grow-only workspace buffer of the index (no hipMalloc / hipFree on the steady-state path) */
struct WsBuf {
    void* p = nullptr;
    mtgi::Workspace* ws = nullptr;
    int slot = -1;
    bool fresh = false; /* the last alloc() had to get new memory (contents undefined) */
    hipError_t alloc(size_t bytes)
    {
        bytes = bytes ? bytes : 8;
        fresh = false;
        if (ws->cap[slot] < bytes) {
            fresh = true;
            if (ws->ptr[slot]) (void)hipFree(ws->ptr[slot]);
            ws->ptr[slot] = nullptr;
            ws->cap[slot] = 0;
            const size_t want = bytes + bytes / 8;
            hipError_t e = hipMalloc(&ws->ptr[slot], want);
            if (e != hipSuccess) { e = hipMalloc(&ws->ptr[slot], bytes); if (e != hipSuccess) return e; ws->cap[slot] = bytes; }
            else ws->cap[slot] = want;
        }
        p = ws->ptr[slot];
        return hipSuccess;
    }
    /* the same, but the first `keep` bytes survive a move (the stream must be idle: the copy runs on the null stream) */
    hipError_t grow_keeping(size_t bytes, size_t keep)
    {
        if (ws->cap[slot] >= bytes || !ws->ptr[slot] || keep == 0) return alloc(bytes);
        void* old = ws->ptr[slot];
        const size_t old_cap = ws->cap[slot];
        ws->ptr[slot] = nullptr;
        ws->cap[slot] = 0;
        hipError_t e = alloc(bytes);
        if (e == hipSuccess) e = hipMemcpy(ws->ptr[slot], old, std::min(keep, old_cap), hipMemcpyDeviceToDevice);
        (void)hipFree(old);
        return e;
    }
    template <typename T> T* as() { return (T*)p; }
};
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <typename B, typename T> hipError_t upload(B& b, const std::vector<T>& v)
{
    hipError_t e = b.alloc(v.size() * sizeof(T));
    if (e != hipSuccess) return e;
    return v.empty() ? hipSuccess : hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
}
} // namespace

namespace {
struct EventSet { /* events of one device_run call */
    std::vector<hipEvent_t> ev;
    ~EventSet() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); }
    /* blocking: a caller waiting for the device sleeps instead of spinning, its CPU time belongs to the worker pool (and, in a container,
     * to the CPU quota the pool lives on) */
    hipError_t make(hipEvent_t& e) { hipError_t r = hipEventCreateWithFlags(&e, hipEventBlockingSync); if (r == hipSuccess) ev.push_back(e); return r; }
};
} // namespace

/* at most MTG_COPY_SLOTS (default 3, 0 = no limit) batches of one device copy their results to the host at the same time (every device has
 * its own link: the tool's host threads, one per device, do not wait for each other) */
struct CopyTurn {
    enum { MAX_DEV = 64 };
    struct State { std::mutex m; std::condition_variable c; int busy = 0; };
    static State& state(int dev) { static State st[MAX_DEV]; return st[(unsigned)dev % MAX_DEV]; }
    static int slots() { return (int)tune::i(tune::T_COPY_SLOTS, 3); }
    State* held = nullptr;
    explicit CopyTurn(int dev)
    {
        if (slots() <= 0) return;
        State& st = state(dev);
        std::unique_lock<std::mutex> lk(st.m);
        st.c.wait(lk, [&] { return st.busy < slots(); });
        st.busy++;
        held = &st;
    }
    void release()
    {
        if (!held) return;
        { std::lock_guard<std::mutex> lk(held->m); held->busy--; }
        held->c.notify_one();
        held = nullptr;
    }
    ~CopyTurn() { release(); }
};

/* The inputs of all batches of a device go up on ONE stream (the batch's own stream waits for its event).  Measured on this box
 * (scripts/pcie_duplex.py): one host-to-device copy at a time runs next to two or three device-to-host copies at the full rate of both
 * directions (40 MB down + 16 MB up: 0.77 ms, the 40 MB alone 0.74); three uploads at a time take the link from the downloads (1.17 ms) --
 * which is what six batches in flight did to the text entry, whose upload is a third of its download.  MTG_UPLOAD_OWN_STREAM=1: as before. */

__device__ __forceinline__ uint64_t d_splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

/* a kernel one translation unit defines (mtg_gpu_fill.hip) and the others may launch */
__global__ void k_encode_targets(const uint8_t* __restrict__ traw, uint64_t* __restrict__ tle, uint64_t* __restrict__ tbad, uint64_t nt, int k);

} // namespace mtgi
#endif
