/*
 * mtg_internal.h -- interfaces between the translation units of libmtgfill.so
 *   mtg_gpu_build.hip : index construction on the device (junction table, walks, unitig store, sparse tables), container load, replicas
 *   mtg_gpu_fill.hip  : the fill kernels and the launch sequence of a batch (device_run)
 *   mtg_gpu_misc.hip  : queries, sequence scan, alignments, the tool's formatter, the device side of the C ABI
 *   mtg_host.cpp      : host orchestration of gapFillFromSource (marshalling, result objects, the multi-contig gaps the device leaves), index files
 *   mtg_cli.cpp       : the MindTheGap fill tool on the C-ABI records
 */
#ifndef MTG_INTERNAL_H
#define MTG_INTERNAL_H
#include "../../include/mtg_fill.h"
#include "mtg_hostutil.h"
#include "mtg_paths.h"
#include "mtg_general.h"
#include "mtg_emit.h"
#include "mtg_copy.h"
#include "mtg_format.h"
#include "mtg_tuning.h"
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <sched.h>
#include <string>
#include <string_view>
#include <thread>
#include <vector>

namespace mtgi {
/* grow-only device buffers reused by successive batches on one index */
struct Workspace {
    enum { NSLOTS = 40, NHOST = 16, NEVENTS = 16 };
    /* slots device_run fills that a later step of the same batch reads (format_run): the marshalled text block and the sequence arena */
    enum { SLOT_TEXT_BLOCK = 2, SLOT_SEQ = 14, SLOT_RES = 16, SLOT_FIL = 17, SLOT_FMT0 = 32 };
    void* ptr[NSLOTS] = {nullptr};
    size_t cap[NSLOTS] = {0};
    /* page-locked host staging blocks: 0..2 = the marshalled input of a batch, 3.. = what the first chunks of a batch brought back */
    void* hptr[NHOST] = {nullptr};
    size_t hcap[NHOST] = {0};
    void* events[NEVENTS] = {nullptr}; /* hipEvent_t: the events a batch records, made on first use */
    void* stream = nullptr;      /* hipStream_t of the batch's copies in and kernels */
    void* copy_stream = nullptr; /* hipStream_t for the copies back of a batch's parts */
    /* share of the gaps (in 65536ths) the walk kernel parked in this workspace's previous whole-batch launch: how the next launch serves its
     * parked gaps (rounds or not, lanes per gap in the finishing kernel); ~0 = no launch yet: the index's latest figure is taken */
    uint32_t park_share = ~0u;
    uint32_t branch_share = ~0u; /* share (16.16) of the previous launch's gaps whose walk stood on a branching node: below 1/128 the next launch starts with the light walk kernel */
    uint32_t post_general = ~0u; /* gaps of this workspace's previous launch that were not lean (the grid of k_post's general form); ~0: no launch yet */
    std::mutex mtx; /* held by a batch from marshalling until its results have been consumed */
};
}

struct mtg_index {
    mtg::Index dev{};          /* tables live in device memory */
    /* a few batches on one index run side by side (callers on several threads, like the reference's Dispatcher): each owns a workspace
     * and its streams, so the traversal of one overlaps the post-processing and the host passes of the other */
#ifndef MTG_NWS
#define MTG_NWS 6 /* batches one index serves at a time (each with its own device scratch, streams and staging blocks) */
#endif
    enum { NWS = MTG_NWS };
    mutable std::atomic<uint32_t> branch_share_any{~0u}; /* the same for the branching share (~0: no launch yet -- the full walk kernel) */
    mutable std::atomic<uint32_t> park_share_any{0}; /* the latest park share any workspace of this index has seen (a fresh workspace starts from it) */
    mutable mtgi::Workspace ws[NWS];
    int device = 0;
    mtg_index_info info{};
    /* how the index came to be (mtg_index_build_profile): the phases of its construction with their device times, and the most device
     * memory the construction held at any time (tables under construction + what it was built from + the finished parts) */
    std::vector<mtg_build_phase> build_phases;
    uint64_t build_peak_bytes = 0;
    double build_total_ms = 0;
};

namespace mtgi {
struct WorkspaceLock {
    Workspace* ws = nullptr;
    std::unique_lock<std::mutex> lk;
};
/* a free workspace of the index, or (all taken) the next one in turn once its batch is done */
inline WorkspaceLock acquire_workspace(const mtg_index* idx)
{
    for (int i = 0; i < mtg_index::NWS; i++) {
        std::unique_lock<std::mutex> lk(idx->ws[i].mtx, std::try_to_lock);
        if (lk.owns_lock()) return WorkspaceLock{&idx->ws[i], std::move(lk)};
    }
    static std::atomic<unsigned> turn{0};
    Workspace& w = idx->ws[turn.fetch_add(1, std::memory_order_relaxed) % mtg_index::NWS];
    return WorkspaceLock{&w, std::unique_lock<std::mutex>(w.mtx)};
}
}

namespace mtgi {

void set_error(const char* fmt, ...);

/* persistent host worker pool for the per-gap loops (sized by cpu_budget(), at most 64; nthreads <= 0: all of it).  A batch runs a dozen short parallel
 * regions back to back, so idle workers first spin for a few tens of microseconds on the generation counter before they go to sleep,
 * and the caller never waits for a helper that has not started by the time the work has run out. */
class Pool {
public:
    static Pool& get() { static Pool p; return p; }
    /* runs job() on up to `nworkers` threads (the caller is one of them) and waits; job must return once the shared work is exhausted */
    void run(int nworkers, const std::function<void()>& job)
    {
        std::unique_lock<std::mutex> run_lock(run_mtx_); /* one parallel region at a time */
        const int helpers = std::min<int>(nworkers - 1, (int)threads_.size());
        if (helpers > 0) {
            {
                std::lock_guard<std::mutex> lk(m_);
                job_ = &job;
                active_.store(helpers, std::memory_order_relaxed);
                /* release: a helper that is still looking at the previous generation may take one of these slots through claim()
                 * alone; it must see job_ */
                to_start_.store(helpers, std::memory_order_release);
                gen_.fetch_add(1, std::memory_order_release);
            }
            /* as many sleepers as there are places: a region of two threads used to wake the whole pool, and every thread that found no place
             * spun SPIN times before it slept again -- with 15 helpers, 0.7 CPU-ms per region burnt out of the cgroup's quota, which the tool's
             * writer threads then lacked (the tool ran FASTER with MTG_POOL_THREADS=2) */
            if (helpers >= (int)threads_.size()) cv_.notify_all();
            else for (int i = 0; i < helpers; i++) cv_.notify_one();
        }
        job();
        if (helpers > 0) {
            const int unclaimed = to_start_.exchange(0, std::memory_order_acq_rel); /* nothing left for late helpers */
            if (unclaimed > 0) active_.fetch_sub(unclaimed, std::memory_order_acq_rel);
            for (int spin = 0; spin < SPIN && active_.load(std::memory_order_acquire) != 0; spin++) cpu_relax();
            if (active_.load(std::memory_order_acquire) != 0) {
                std::unique_lock<std::mutex> lk(m_);
                done_cv_.wait(lk, [&] { return active_.load(std::memory_order_acquire) == 0; });
            }
            job_ = nullptr;
        }
    }
    int size() const { return (int)threads_.size() + 1; }

    /* CPUs this process may keep busy: the hardware threads, the affinity mask and the CFS bandwidth limit of its cgroup (a container
     * with "16 CPUs" on a 256-thread host: threads beyond the quota only get the whole group throttled, spinning ones above all) */
    static int cpu_budget()
    {
        int n = (int)std::max(1u, std::thread::hardware_concurrency());
#ifdef __linux__
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min(n, std::max(1, CPU_COUNT(&set)));
        long long quota = -1, period = 0;
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) { /* cgroup v2: "<quota|max> <period>" */
            char q[32] = "";
            if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
            fclose(f);
        } else {
            if (FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(fq, "%lld", &quota) != 1) quota = -1; fclose(fq); }
            if (FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fp, "%lld", &period) != 1) period = 0; fclose(fp); }
        }
        if (quota > 0 && period > 0) n = std::min(n, (int)std::max<long long>(1, quota / period));
#endif
        return n;
    }

    static void cpu_relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }

private:
    enum { SPIN = 4000 };
    Pool()
    {
        int n = std::min(cpu_budget(), 64);
        if (tune::is_set(tune::T_POOL_THREADS)) n = std::max(1, std::min((int)tune::i(tune::T_POOL_THREADS), 256));
        for (int i = 1; i < n; i++) threads_.emplace_back([this] { loop(); });
    }
    ~Pool()
    {
        { std::lock_guard<std::mutex> lk(m_); stop_.store(true); }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    bool claim()
    {
        int v = to_start_.load(std::memory_order_acquire);
        while (v > 0 && !to_start_.compare_exchange_weak(v, v - 1, std::memory_order_acq_rel)) {}
        return v > 0;
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            bool mine = false;
            for (int spin = 0; spin < SPIN && !mine; spin++) {
                if (stop_.load(std::memory_order_relaxed)) return;
                const uint64_t g = gen_.load(std::memory_order_acquire);
                if (g != seen) { seen = g; mine = claim(); }
                else cpu_relax();
            }
            if (!mine) {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_.load() || gen_.load(std::memory_order_acquire) != seen; });
                if (stop_.load()) return;
                seen = gen_.load(std::memory_order_acquire);
                mine = claim();
            }
            if (!mine) continue;
            (*job_)();
            if (active_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> lk(m_); /* the caller may be about to sleep on done_cv_ */
                done_cv_.notify_all();
            }
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_, run_mtx_;
    std::condition_variable cv_, done_cv_;
    const std::function<void()>* job_ = nullptr;
    std::atomic<int> to_start_{0}, active_{0};
    std::atomic<uint64_t> gen_{0};
    std::atomic<bool> stop_{false};
};

template <typename F> inline void parallel_for(size_t n, int nthreads, F f, size_t grain = 64)
{
    if (nthreads <= 0) nthreads = Pool::get().size();
    if ((size_t)nthreads * grain > n) nthreads = (int)std::max<size_t>(n / grain, 1);
    if (nthreads <= 1) { for (size_t i = 0; i < n; i++) f(i); return; }
    std::atomic<size_t> next{0};
    std::function<void()> work = [&]() {
        for (;;) {
            const size_t b = next.fetch_add(grain);
            if (b >= n) break;
            const size_t e = std::min(n, b + grain);
            for (size_t i = b; i < e; i++) f(i);
        }
    };
    Pool::get().run(nthreads, work);
}

/* out[0] = 0, out[i + 1] = out[i] + get(i), blocked over the pool; returns the total */
template <typename G> inline uint64_t parallel_prefix(size_t n, int nthreads, uint64_t* out, G get)
{
    const size_t B = 4096, nb = (n + B - 1) / B;
    std::vector<uint64_t> bs(nb + 1, 0);
    out[0] = 0;
    parallel_for(nb, nthreads, [&](size_t b) {
        uint64_t sum = 0;
        for (size_t i = b * B; i < std::min(n, (b + 1) * B); i++) { const uint64_t v = get(i); out[i + 1] = v; sum += v; }
        bs[b + 1] = sum;
    }, 1);
    for (size_t b = 0; b < nb; b++) bs[b + 1] += bs[b];
    parallel_for(nb, nthreads, [&](size_t b) {
        uint64_t run = bs[b];
        for (size_t i = b * B; i < std::min(n, (b + 1) * B); i++) { run += out[i + 1]; out[i + 1] = run; }
    }, 1);
    return bs[nb];
}

/* ---- a batch of gapFillFromSource calls, marshalled for the device -------------------------------------------------------------------
 * Three contiguous blocks that go up in one copy each (A: per-gap arrays, B: packed early-stop patterns, C: the targets as text).  With a
 * workspace they sit in its page-locked staging blocks, otherwise in `own_*`. */
struct Target { /* one entry of a gap's targetDictionary (views on the caller's strings) */
    std::string_view seq, name;
    bool is_rc = false;
};
struct TargetSpan {
    const Target* p = nullptr;
    uint32_t n = 0;
    size_t size() const { return n; }
    const Target& operator[](size_t i) const { return p[i]; }
    const Target* begin() const { return p; }
    const Target* end() const { return p + n; }
};

/* grow-only staging block `slot` of a workspace (page-locked on the device build); nullptr when it cannot be had */
void* staging_host(Workspace* ws, int slot, size_t bytes);
/* page-locked host memory for the result arrays of a batch (plain memory in the emulation); nullptr on failure */
void* pinned_alloc(size_t bytes);
void pinned_free(void* p);
int host_register(void* p, size_t bytes); /* page-lock / release a caller's range (the backends) */
int host_unregister(void* p);
/* plain copies between the host and the index's device (rare paths of the host code that touch a caller's device buffer) */
int device_download(const mtg_index* idx, void* host_dst, const void* dev_src, size_t bytes);
int device_upload(const mtg_index* idx, void* dev_dst, const void* host_src, size_t bytes);

template <typename T> struct Arr {
    T* p = nullptr;
    size_t n = 0;
    T* data() const { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T& operator[](size_t i) const { return p[i]; }
};
struct FillInput {
    int k = 31;
    bool want_all_contigs = false;
    Workspace* ws = nullptr; /* whose staging blocks to use (the caller holds its lock); nullptr: own storage */
    /* block A */
    Arr<uint64_t> src;     /* oriented source k-mer per gap */
    Arr<uint64_t> r0;      /* first k-mer of the pattern */
    Arr<uint32_t> roff;    /* first word of gap i's pattern */
    Arr<uint32_t> rlen;    /* pattern length in nt */
    Arr<uint32_t> toff, tcnt;
    Arr<uint8_t> nbmis, fast_ok, flags; /* flags: mtg_emit.h GAPF_* */
    /* block B */
    Arr<uint64_t> rwords;  /* packed swf patterns, concatenated */
    /* block C: the targets of all gaps as text (mtg_post.h: TARGET_SLOT bytes each); the device turns them into k-mers and masks */
    Arr<uint8_t> traw;
    void *block_a = nullptr, *block_b = nullptr, *block_c = nullptr;
    size_t bytes_a = 0, bytes_b = 0, bytes_c = 0;
    std::vector<uint64_t> own_a, own_b, own_c;
    /* device copies of the three blocks and of the encoded targets, when the batch was prepared ahead (mtg_batch): device_run then uploads nothing */
    void* dev_a = nullptr; void* dev_b = nullptr; void* dev_tenc = nullptr;
    /* mtg_fill_text: the strings are still text.  Block A then holds only the integer columns the host can write without looking at a
     * string (roff, rlen, toff, tcnt, nbmis, flags), block B does not exist on the host (n_rwords words on the device), and block C is the
     * text block: [source_off 8n | pattern_off 8n | dict_seq_off 8nt | source_len 4n | dict_seq_len 4nt | pad to 8 | text]; the device
     * writes src, r0, fast_ok, the patterns and the encoded targets (mtg_marshal.h). */
    bool text_mode = false;
    uint64_t n_rwords = 0, n_text_targets = 0, text_bytes = 0;
    const char* text_direct = nullptr; /* text mode: the block lies in memory the caller has page-locked (mtg_host_register): block_c holds the offset arrays only */
    static size_t text_block_off(size_t n, size_t nt, int which) /* 0 source_off, 1 pattern_off, 2 dict_seq_off, 3 source_len, 4 dict_seq_len, 5 text */
    {
        const size_t o[6] = {0, 8 * n, 16 * n, 16 * n + 8 * nt, 20 * n + 8 * nt, (20 * n + 12 * nt + 7) & ~(size_t)7};
        return o[which];
    }
    /* two-pass marshalling: size(i, ...) for every gap in order, then layout(), then set(i, ...) from any thread */
    void resize(size_t n);
    void size(size_t i, size_t swf_len, size_t n_targets) { rlen[i] = (uint32_t)swf_len; tcnt[i] = (uint32_t)n_targets; }
    void layout();
    void set_common(size_t i, std::string_view source, std::string_view swf_target, int nb_mis, uint8_t gap_flags);
    void set_target(size_t slot, std::string_view seq);
    enum { BLOCK = 512 };
    std::vector<uint64_t> blk_rw, blk_nt;
    std::vector<uint32_t> slen; /* host only: length of every gap's source (pass 1 has looked at it; nobody needs to again) */
    void alloc_b(uint64_t rw, uint64_t nt);
    /* The two passes over blocks of gaps, each one parallel region.  plan: sz(i, swf_len, n_targets) for every gap, then the offsets of
     * every block; fill: offsets of every gap of a block in turn, then st(i) (which calls set_common). */
    template <typename SizeFn> void plan(size_t n, int nthreads, SizeFn sz)
    {
        resize(n);
        slen.resize(n);
        const size_t nb = (n + BLOCK - 1) / BLOCK;
        blk_rw.assign(nb + 1, 0);
        blk_nt.assign(nb + 1, 0);
        parallel_for(nb, nthreads, [&](size_t b) {
            uint64_t rw = 0, nt = 0;
            for (size_t i = b * BLOCK; i < std::min(n, (b + 1) * (size_t)BLOCK); i++) {
                size_t sl = 0, tn = 0;
                sz(i, sl, tn);
                size(i, sl, tn);
                rw += (sl + 31) / 32 + 1;
                nt += tn;
            }
            blk_rw[b + 1] = rw;
            blk_nt[b + 1] = nt;
        }, 1);
        for (size_t b = 0; b < nb; b++) { blk_rw[b + 1] += blk_rw[b]; blk_nt[b + 1] += blk_nt[b]; }
        alloc_b(blk_rw[nb], blk_nt[nb]);
    }
    /* Both passes in one when the staging blocks of the workspace are already large enough (every batch after the first of its shape):
     * a block of gaps adds up its sizes, learns where the blocks before it end (they are handed out in order and publish their ends as
     * soon as they know their own sizes), and writes its gaps there.  sz(i, swf_len, n_targets) returns false for a gap that must not
     * be written (the caller reports the error).  Returns false, with nothing usable written, when a block did not fit or a gap was
     * refused: the caller then runs plan() and fill(), which make the staging blocks grow. */
    template <typename SizeFn, typename SetFn> bool plan_and_fill(size_t n, int nthreads, SizeFn sz, SetFn st)
    {
        resize(n);
        slen.resize(n);
        const size_t nb = (n + BLOCK - 1) / BLOCK;
        blk_rw.assign(nb + 1, 0);
        blk_nt.assign(nb + 1, 0);
        const size_t cap_b = ws ? ws->hcap[1] : 0, cap_c = ws ? ws->hcap[2] : 0;
        rwords.p = ws ? (uint64_t*)ws->hptr[1] : nullptr;
        traw.p = ws ? (uint8_t*)ws->hptr[2] : nullptr;
        std::vector<std::atomic<int64_t>> end_rw(nb + 1), end_nt(nb + 1);
        for (size_t b = 0; b <= nb; b++) { end_rw[b].store(-1, std::memory_order_relaxed); end_nt[b].store(-1, std::memory_order_relaxed); }
        end_nt[0].store(0, std::memory_order_relaxed);
        end_rw[0].store(0, std::memory_order_release);
        std::atomic<bool> deferred{false};
        parallel_for(nb, nthreads, [&](size_t b) {
            const size_t i0 = b * BLOCK, i1 = std::min(n, (b + 1) * (size_t)BLOCK);
            uint64_t rw = 0, nt = 0;
            bool ok = true;
            for (size_t i = i0; i < i1; i++) {
                size_t sl = 0, tn = 0;
                ok = sz(i, sl, tn) && ok;
                size(i, sl, tn);
                rw += (sl + 31) / 32 + 1;
                nt += tn;
            }
            int64_t rw0;
            while ((rw0 = end_rw[b].load(std::memory_order_acquire)) < 0) Pool::cpu_relax();
            const int64_t nt0 = end_nt[b].load(std::memory_order_relaxed);
            end_nt[b + 1].store(nt0 + (int64_t)nt, std::memory_order_relaxed);
            end_rw[b + 1].store(rw0 + (int64_t)rw, std::memory_order_release);
            blk_rw[b] = (uint64_t)rw0;
            blk_nt[b] = (uint64_t)nt0;
            if (!ok || deferred.load(std::memory_order_relaxed) || 8 * ((uint64_t)rw0 + rw) + 64 > cap_b || (uint64_t)mtg::TARGET_SLOT * ((uint64_t)nt0 + nt) + 64 > cap_c) {
                deferred.store(true, std::memory_order_relaxed);
                return;
            }
            uint64_t r = (uint64_t)rw0, t = (uint64_t)nt0;
            for (size_t i = i0; i < i1; i++) {
                roff[i] = (uint32_t)r;
                toff[i] = (uint32_t)t;
                r += (rlen[i] + 31) / 32 + 1; /* before st(i): set() may flag the pattern by overwriting rlen */
                t += tcnt[i];
                st(i);
            }
        }, 1);
        if (deferred.load()) return false; /* the caller falls back to plan() + fill(): sizes again (st may have flagged a pattern's length), larger blocks */
        blk_rw[nb] = (uint64_t)end_rw[nb].load(std::memory_order_acquire);
        blk_nt[nb] = (uint64_t)end_nt[nb].load(std::memory_order_relaxed);
        alloc_b(blk_rw[nb], blk_nt[nb]); /* the same blocks: they were large enough */
        return true;
    }
    template <typename SetFn> void fill(int nthreads, SetFn st)
    {
        const size_t n = src.size(), nb = (n + BLOCK - 1) / BLOCK;
        parallel_for(nb, nthreads, [&](size_t b) {
            uint64_t rw = blk_rw[b], nt = blk_nt[b];
            for (size_t i = b * BLOCK; i < std::min(n, (b + 1) * (size_t)BLOCK); i++) {
                roff[i] = (uint32_t)rw;
                toff[i] = (uint32_t)nt;
                rw += (rlen[i] + 31) / 32 + 1; /* before st(i): set() may flag the pattern by overwriting rlen */
                nt += tcnt[i];
                st(i);
            }
        }, 1);
    }
    /* byte offsets of the arrays inside block A (the device copy has the same layout) */
    static size_t off_a(size_t n, int which) { static const size_t mul[9] = {0, 8, 16, 20, 24, 28, 32, 33, 34}; return mul[which] * n8(n); }
    static size_t n8(size_t n) { return (n + 7) & ~(size_t)7; }
    enum { BYTES_A_PER_GAP = 35 };
};

/* ---- what comes back ---------------------------------------------------------------------------------------------------------------
 * The common case never reaches the host as per-gap work: the device writes the C-ABI records and the ASCII sequences (mtg_emit.h) and
 * they are copied into the arrays below.  Only multi-contig gaps (and the stage-A entry) bring their contigs back. */
struct ResultSink {
    size_t n = 0;
    mtg_gap_result* res = nullptr; /* n records; nullptr: the caller only wants the contigs (stage-A entry) */
    mtg_filled* fil = nullptr;     /* n records: slot i belongs to gap i */
    char* seq = nullptr;           /* sequence arena: NUL-terminated fills, in gap order unless `in_gap_order` comes back false */
    size_t seq_cap = 0;
    /* res, fil and seq are parts of ONE page-locked block: res == combo, fil == combo + combo_off_fil, seq == combo + combo_off_seq (the offsets a
     * function of n).  The device then lays its copies out the same way and a whole-batch launch brings its results over in one copy.  grow_seq
     * moves the whole block (and sets res, fil, seq, combo anew). */
    char* combo = nullptr;
    size_t combo_off_fil = 0, combo_off_seq = 0;
    char* seq_dev = nullptr;       /* a buffer of the caller on the index's device (seq_cap bytes): the result kernel writes the arena there instead of into the workspace */
    bool seq_on_device = false;    /* no host copy is wanted: seq == seq_dev, the records carry device addresses and nothing of the arena is copied to the host */
    bool seq_stays_in_workspace = false; /* the arena is produced in the workspace's own device buffer and NOT copied to `seq` (the records still carry the
                                            addresses it would have there): the batch's text is formatted on the device next (format_run) */
    bool device_records_whole = false;   /* out: the workspace's device copies of the records are those of the whole batch (one launch, nothing re-run, no gap for the host) */
    char* ext = nullptr;           /* extension arena; ext[0] = 0 is the empty string of every record without extension */
    size_t ext_cap = 0;
    /* an arena turned out too small: must replace it by a block of at least `need` bytes whose first `keep` bytes are those of the old
     * block, and set pointer and capacity (false: it cannot grow -- the caller's own buffer) */
    std::function<bool(size_t need, size_t keep)> grow_seq, grow_ext;
    uint64_t seq_used = 0, ext_used = 1;
    uint64_t n_filled = 0;         /* gaps filled on the common path */
    bool in_gap_order = true;
    /* the batch also in relocatable form (mtg_wire_*), produced by the result kernel in this device buffer of the caller: the send buffer of
     * a gather.  wire_ok comes back false when the batch could not leave that way (several launches, re-run gaps, multi-contig gaps whose
     * records the host writes): the caller then serialises the finished result set itself (mtg_results_to_wire) */
    void* wire_dev = nullptr;
    uint64_t wire_cap = 0, wire_tag = 0, wire_bytes = 0;
    bool wire_ok = false;
};

/* contigs of the gaps that need the host: one record per slot of a launch, the dense words and the dense contig metadata */
struct HostChunk {
    const mtg::SlotRec* recs = nullptr;
    const uint64_t* words = nullptr;
    const uint32_t* meta = nullptr;
    uint32_t m = 0;
    std::vector<uint64_t> own;
    /* contig-graph paths of the multi-contig gaps of the chunk (k_paths): PATHS_WORDS words per such gap, path_of[slot] = its rank or -1 */
    std::vector<uint32_t> paths;
    std::vector<int32_t> path_of;
    std::vector<uint32_t> gap_of; /* gap of every slot (empty: slot = gap) */
    /* what the device made of the launch's multi-contig gaps (k_general, mtg_general.h): a header per gap in the order of the launch's list (empty:
     * the host takes them all), the solutions and their ASCII */
    std::vector<mtg::GenGap> gen_gaps;
    std::vector<mtg::GenSol> gen_sols;
    std::vector<char> gen_ascii;
    bool gen_check = false; /* TEST-ONLY (set by the emulation's device_run, which always brings the contigs along): the host's path runs next to every
                               device answer and run_general compares the two */
    /* carves recs / words / meta for m slots, tw words, tc metadata entries out of `own` */
    void carve(uint32_t m_, uint64_t tw, uint64_t tc, mtg::SlotRec*& r, uint64_t*& w, uint32_t*& mt)
    {
        own.resize(bytes_for(m_, tw, tc) / 8);
        uint8_t* b = (uint8_t*)own.data();
        r = (mtg::SlotRec*)b;
        w = (uint64_t*)(b + rec_bytes(m_));
        mt = (uint32_t*)(b + rec_bytes(m_) + (tw + 1) * 8);
        recs = r; words = w; meta = mt; m = m_;
    }
    static size_t rec_bytes(uint32_t m_) { return ((size_t)m_ * sizeof(mtg::SlotRec) + 63) & ~(size_t)63; }
    static size_t bytes_for(uint32_t m_, uint64_t tw, uint64_t tc) { return (rec_bytes(m_) + (tw + 1) * 8 + tc * 20 + 64 + 7) & ~(size_t)7; }
};

/* what comes back from the device for one gap (views into a HostChunk) */
struct GapDev {
    mtg::GapOut o{};
    mtg::PostOut p{};
    uint32_t n_meta = 0; /* contigs whose data came back (all of them, or none) */
    const uint64_t* words = nullptr;
    const uint32_t *len = nullptr, *word_start = nullptr, *tpos = nullptr, *terr = nullptr, *ttgt = nullptr;
    const uint32_t* paths = nullptr; /* k_paths record of the gap (mtg_paths.h), if the device enumerated its paths */
    std::string contig(size_t i) const
    {
        std::string s;
        mtg::unpack_seq(words + word_start[i], len[i], s);
        return s;
    }
};
struct SpecialGap {
    uint32_t gap, chunk, slot;
    uint32_t rank; /* its place in the launch's list of multi-contig gaps (HostChunk::gen_gaps) */
};
struct DevBatch {
    std::vector<std::unique_ptr<HostChunk>> chunks; /* launches that had gaps for the host */
    std::vector<SpecialGap> special;                 /* multi-contig gaps (every gap with want_all_contigs), in no particular order */
    static GapDev view(const HostChunk& c, size_t slot)
    {
        const mtg::SlotRec& r = c.recs[slot];
        GapDev g;
        if (!c.path_of.empty() && c.path_of[slot] >= 0) g.paths = c.paths.data() + (size_t)c.path_of[slot] * mtg::PATHS_WORDS;
        g.o = r.o;
        g.p = r.p;
        g.n_meta = r.nc;
        g.words = c.words + r.wbase;
        if (g.n_meta) {
            const uint32_t* b0 = c.meta + 5 * r.cbase;
            g.len = b0; g.word_start = b0 + r.nc; g.tpos = b0 + 2 * (size_t)r.nc; g.terr = b0 + 3 * (size_t)r.nc; g.ttgt = b0 + 4 * (size_t)r.nc;
        }
        return g;
    }
    GapDev view(const SpecialGap& s) const { return view(*chunks[s.chunk], s.slot); }
};

/* set by fill_marshalled around the second attempt at a batch one of whose multi-contig gaps the host could not take from the device's answer (two targets
 * under one name: the reference keeps their paths in ONE group): device_run then hands every multi-contig gap to the host's path, contigs and all */
extern thread_local bool tl_host_general;
enum { MTG_INTERNAL_RETRY_HOST_GENERAL = -1000 }; /* run_general's "do the launch again with the host's path" (never leaves the library) */
/* traversal + post-processing + result emission for all gaps (chunked, tiered); returns MTG_* status.  `while_busy` runs once on the
 * calling thread after everything of the first launch has been queued: the device needs nothing more from the host. */
int device_run(const mtg_index* idx, const mtg_params* p, const FillInput& in, ResultSink& sink, DevBatch& special, mtg_batch_stats* stats,
               const std::function<void()>* while_busy = nullptr);
/* device copies of a marshalled batch (mtg_batch): block A, block B, encoded targets */
int batch_upload(const mtg_index* idx, FillInput& in);
void batch_release_device(FillInput& in);

/* membership scan over packed sequences: host arrays in (words/off/len), bit output as in mtg_index_scan_packed_device; device = 1: the pointers are device pointers */
int scan_run(const mtg_index* idx, const uint64_t* words, size_t nwords, const uint64_t* word_off, const uint32_t* len, size_t nseq, int mode, uint64_t* out_bits, int device_ptrs,
             mtg_scan_stats* st);

/* The reads of Graph::create as a stream of text blocks: whole sequences separated by '\n' (an invalid character by gatb's rule, so no k-mer
 * spans two reads), at most a few hundred MB each; rewind() starts over (the counting may need several passes over the reads). */
struct ReadStream {
    virtual ~ReadStream() {}
    virtual bool rewind() = 0;
    virtual bool next_block(const char*& p, size_t& n) = 0; /* false: end of the reads (or an error: failed() tells) */
    virtual bool failed() const = 0;
    virtual size_t size_hint() const = 0; /* rough number of characters in all */
};
/* Graph::create (src/Filler.cpp:172-213) on the device: exact k-mer counts (DSK's role), abundance histogram, solidity cut-off
 * (abundance_min < 0: automatic, floor 3), solid k-mers straight from the count table into the index tables, lookaheads, unitig store.
 * No k-mer list ever exists on the host. */
int index_from_stream(ReadStream& rs, int k, int abundance_min, int abundance_max, mtg_index** out);
int auto_cutoff(const std::vector<uint64_t>& histo, int floor_thr);

/* Needleman-Wunsch of src/Utils.cpp:87-189 (match +10, mismatch -5, gap -5; traceback preference diagonal, up, left) for a batch of
 * sequence pairs: matches[p] = number of matching positions along the traceback of pair p (a = rows, b = columns) */
struct NwPair {
    const char* a;
    uint32_t na;
    const char* b;
    uint32_t nb;
};
/* ws: the workspace of the batch that asks (its cached device buffers and its stream: no hipMalloc / hipFree -- hipFree waits for the whole
 * device, i.e. for every other batch in flight -- and no copy on the null stream); nullptr: buffers of the call's own */
int nw_run(const mtg_index* idx, const std::vector<NwPair>& pairs, std::vector<uint32_t>& matches, Workspace* ws = nullptr);

int query_run(const mtg_index* idx, const uint64_t* kmers, size_t n, uint32_t* abund, uint8_t* succ, uint8_t* pred, Workspace* ws = nullptr);
/* the solid k-mers of an index with their abundances, read back from its device tables, in pieces handed to sink(kmers, abundances, count) */
int index_export(const mtg_index* idx, const std::function<bool(const uint64_t*, const uint32_t*, size_t)>& sink);
/* An index as it is kept in its container (version 3): the unitig store -- 2-bit sequences and one abundance byte per k-mer -- and the k-mers
 * of no stored unitig.  Nothing else is needed to put the index back on a device: its tables are derived from the store (sparse form). */
struct IndexDump {
    int k = 0, abundance_min = 0, abundance_auto = -1;
    uint64_t nb_solid = 0, nb_branching = 0, nb_saturated = 0, n_words = 0, n_unitigs = 0;
    std::vector<uint64_t> words;  /* n_words in use + padding */
    std::vector<uint8_t> ab;      /* 32 bytes per word */
    std::vector<uint64_t> left_k; /* canonical k-mers of no stored unitig */
    std::vector<uint32_t> left_a; /* their abundances */
    /* a container that goes to the device while it is being read (index_load): `ab` stays empty and the abundance bytes -- four fifths of the
     * file -- are handed over piece by piece: ab_read(off, n, dst) copies bytes [off, off + n) of the abundance section to dst; it is
     * called from several threads at once, each with a page-locked dst of its own */
    std::function<bool(uint64_t off, size_t n, void* dst)> ab_read;
    /* adj_prealloc_begin's handle (or null): the memory of the largest table, being allocated on a helper thread since the container's header
     * was read -- on this pool a first large hipMalloc can take seconds, which the reading of the file then hides */
    void* prealloc = nullptr;
};
/* starts a thread that allocates the device memory the sparse ADJ table of an index of nb_solid k-mers will take; index_from_dump takes it over
 * (adj_prealloc_drop: the load did not get that far) */
void* adj_prealloc_begin(uint64_t nb_solid, int k);
void adj_prealloc_drop(void* handle);
int index_dump(const mtg_index* idx, IndexDump& d);
int index_from_dump(const IndexDump& d, mtg_index** out);

void stats_store(const mtg_batch_stats& s);

/* ---- the tool's text, formatted on the device (mtg_format.h).  Called by the batch that has just run device_run on `ws` (the workspace is
 * still its own): the marshalled text block and the sequence arena are where that run left them. */
struct FormatIn {
    Workspace* ws = nullptr;
    size_t n = 0, nt = 0;                 /* sites, dictionary entries (the layout of the text block) */
    const mtg_gap_result* res = nullptr;  /* the batch's records as the caller will see them (host) */
    const mtg_filled* fil = nullptr;      /* slot i = site i (common path) */
    bool device_records_whole = false;
    const char* host_seq = nullptr;       /* the address range the records' seq pointers of the common path lie in ... */
    uint64_t seq_used = 0;                /* ... and its length: those sequences are in the workspace's arena at the same offsets */
    const char* host_text = nullptr;      /* the caller's block (the emulation reads it; the device has its copy) */
    const uint64_t* source_off = nullptr; /* per site, into the block */
    const uint32_t* source_len = nullptr;
    const uint64_t* name_off = nullptr;   /* breakpointName of site i = text[name_off[i], + name_len[i]) */
    const uint32_t* name_len = nullptr;
};
struct FormatOut {
    char* text[mtg::FMT_STREAMS] = {nullptr, nullptr, nullptr}; /* page-locked, grow-only */
    size_t cap[mtg::FMT_STREAMS] = {0, 0, 0};
    uint64_t bytes[mtg::FMT_STREAMS] = {0, 0, 0};
    uint64_t n = 0, n_simple = 0;
    std::vector<uint32_t> complex_sites;                  /* ascending: the sites the device did not write */
    std::vector<uint64_t> complex_off[mtg::FMT_STREAMS];  /* where their text belongs: the bytes of the simple sites before them */
    double kernel_ms = 0;
};
int format_run(const mtg_index* idx, const FormatIn& fi, FormatOut& out);
/* the sequence arena the batch's device_run left in the workspace, to host memory (a batch whose arena stayed on the device and is needed after all) */
int workspace_arena_download(const mtg_index* idx, Workspace* ws, char* dst, uint64_t bytes);

/* ---- the multi-contig path on the host (gaps whose target is not on contig 0) ------------------------------------------------------- */
struct Solution { /* filled_insertion_t, src/Utils.hpp:46-104 */
    std::string seq;
    int nb_errors = 0;
    int target = -1;
    float avg = 0, median = 0;
    int qual = 0, count = 0, rank = 0;
    size_t ab_off = 0, ab_n = 0; /* slice of the batched abundance query */
};
struct GapWork {
    TargetSpan targets; /* targetDictionary in iteration order */
    std::vector<Target> target_store;
    std::string_view source;
    bool anchor_repeated = false, reverse = false;
    int nb_total_filled = 0;
    bool has_counts = false;
    std::vector<Solution> sols;
};

bool read_sequences(const std::string& path, std::vector<std::pair<std::string, std::string>>& out);
int index_from_kmers(const uint64_t*, const uint32_t*, size_t, int, mtg_index**);
/* k-mers [off, off + m) and their abundances of a counted set that is handed over piece by piece (false: failed, message set) */
using KmerFetch = std::function<bool(size_t off, size_t m, const uint64_t*& kmers, const uint32_t*& abundance)>;
int index_from_kmer_pieces(size_t n, int k, const KmerFetch& fetch, mtg_index** out);
int index_from_reads(const char*, int, int, int, mtg_index**);
int index_save(const mtg_index*, const char*);
int index_load(const char*, mtg_index**);
int fill_main(int argc, const char* const* argv);

} // namespace mtgi
#endif
