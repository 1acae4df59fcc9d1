/*
 * mtg_internal.h -- interfaces between the translation units of libmtgfill.so
 *   mtg_gpu.hip   : HIP kernels, device memory, index construction, stage A batches
 *   mtg_host.cpp  : host orchestration of gapFillFromSource (contig graph, paths, dedupe, writers), CLI
 */
#ifndef MTG_INTERNAL_H
#define MTG_INTERNAL_H
#include "../../include/mtg_fill.h"
#include "mtg_hostutil.h"
#include "mtg_post.h"
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <string_view>
#include <thread>
#include <vector>

namespace mtgi {
/* grow-only device buffers reused by successive batches on one index */
struct Workspace {
    enum { NSLOTS = 32 };
    void* ptr[NSLOTS] = {nullptr};
    size_t cap[NSLOTS] = {0};
    std::mutex mtx;
};
}

struct mtg_index {
    mtg::Index dev{};          /* tables live in device memory */
    mutable mtgi::Workspace ws;
    int device = 0;
    mtg_index_info info{};
};

namespace mtgi {

void set_error(const char* fmt, ...);

/* persistent host worker pool for the per-gap loops (nthreads <= 0: all cores, capped at 64) */
class Pool {
public:
    static Pool& get() { static Pool p; return p; }
    /* runs job(worker) on `nworkers` threads (the caller is one of them) and waits */
    void run(int nworkers, const std::function<void()>& job)
    {
        std::unique_lock<std::mutex> run_lock(run_mtx_); /* one parallel region at a time */
        const int helpers = std::min<int>(nworkers - 1, (int)threads_.size());
        if (helpers > 0) {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &job;
            to_start_ = helpers;
            active_ = helpers;
            gen_++;
        }
        if (helpers > 0) cv_.notify_all();
        job();
        if (helpers > 0) {
            std::unique_lock<std::mutex> lk(m_);
            done_cv_.wait(lk, [&] { return active_ == 0; });
            job_ = nullptr;
        }
    }
    int size() const { return (int)threads_.size() + 1; }

private:
    Pool()
    {
        int n = (int)std::min<unsigned>(std::thread::hardware_concurrency(), 64u);
        for (int i = 1; i < n; i++) threads_.emplace_back([this] { loop(); });
    }
    ~Pool()
    {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void()>* job = nullptr;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || (gen_ != seen && to_start_ > 0); });
                if (stop_) return;
                seen = gen_;
                to_start_--;
                job = job_;
            }
            (*job)();
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--active_ == 0) done_cv_.notify_all();
            }
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_, run_mtx_;
    std::condition_variable cv_, done_cv_;
    const std::function<void()>* job_ = nullptr;
    int to_start_ = 0, active_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
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

/* one gapFillFromSource call and its results (host side) */
/* strings of a batch are VIEWS on the caller's storage, which must stay alive until the results have been consumed */
struct Target {
    std::string_view seq, name;
    bool is_rc = false;
    uint64_t code = 0;    /* 2-bit code of the first k chars */
    uint64_t badmask = 0; /* positions (pair-lsb) that can never match (not ACGT/acgt) */
    bool usable = true;   /* at least k chars */
};
struct TargetSpan {
    const Target* p = nullptr;
    uint32_t n = 0;
    size_t size() const { return n; }
    const Target& operator[](size_t i) const { return p[i]; }
    const Target* begin() const { return p; }
    const Target* end() const { return p + n; }
};

/* host copies of what a chunk of gaps brought back */
struct HostChunk {
    std::vector<mtg::GapOut> out;    /* per slot */
    std::vector<mtg::PostOut> post;
    std::vector<uint32_t> nw, nc;    /* words / contig-metadata entries copied back per slot */
    std::vector<uint64_t> wbase, cbase;
    std::vector<uint64_t> words;
    std::vector<uint32_t> meta;      /* 5 arrays of tc entries: len, word_start, tpos, terr, ttgt */
    uint64_t tc = 0;
};

/* what comes back from the device for one gap (views into a HostChunk) */
struct GapDev {
    mtg::GapOut o{};
    mtg::PostOut p{};
    /* contig data: all contigs when n_meta == o.n_contigs, otherwise only the leading words of contig 0 */
    uint32_t n_meta = 0;
    const uint64_t* words = nullptr;
    const uint32_t *len = nullptr, *word_start = nullptr, *tpos = nullptr, *terr = nullptr, *ttgt = nullptr;
    std::string contig(size_t i) const
    {
        std::string s;
        mtg::unpack_seq(words + word_start[i], len[i], s);
        return s;
    }
    /* contig0[from, to) from the leading words */
    std::string contig0_slice(uint32_t from, uint32_t to) const
    {
        std::string s;
        if (to <= from) return s;
        static const char NT[4] = {'A', 'C', 'T', 'G'};
        s.resize(to - from);
        for (uint32_t i = from; i < to; i++) s[i - from] = NT[(words[i >> 5] >> (2 * (i & 31))) & 3];
        return s;
    }
};
struct DevBatch {
    std::vector<uint32_t> chunk_of, slot_of; /* where gap i's results sit */
    std::vector<std::unique_ptr<HostChunk>> chunks;
    size_t size() const { return chunk_of.size(); }
    /* view of gap i (cheap: a few pointer computations) */
    GapDev operator[](size_t i) const
    {
        const HostChunk& c = *chunks[chunk_of[i]];
        const uint32_t s = slot_of[i];
        GapDev g;
        g.o = c.out[s];
        g.p = c.post[s];
        g.n_meta = c.nc[s];
        g.words = c.words.data() + c.wbase[s];
        if (g.n_meta) {
            const uint32_t* b0 = c.meta.data() + c.cbase[s];
            g.len = b0; g.word_start = b0 + c.tc; g.tpos = b0 + 2 * c.tc; g.terr = b0 + 3 * c.tc; g.ttgt = b0 + 4 * c.tc;
        }
        return g;
    }
};

/* what to copy back for a gap: nw leading words of its arena, metadata of nc contigs (0 or all) */
inline void copy_plan(const mtg::GapOut& o, const mtg::PostOut& p, bool want_all, uint32_t& nw, uint32_t& nc)
{
    nw = nc = 0;
    if (o.status != mtg::GAP_OK) return;
    if (want_all || (p.fast == 0 && p.nb_terminal > 0)) { nw = o.n_words; nc = o.n_contigs; }
    else if (p.fast == 1) nw = (p.pos + 31) / 32;
    else if (p.fast == 2) nw = 0;
    else nw = (p.clen0 + 31) / 32; /* no terminal node: contig 0 is the extension sequence */
}

/* a batch of gapFillFromSource calls, marshalled for the device */
struct FillInput {
    int k = 31;
    bool want_all_contigs = false;
    std::vector<uint64_t> src;     /* oriented source k-mer per gap */
    std::vector<uint64_t> rwords;  /* packed swf patterns, concatenated */
    std::vector<uint32_t> roff;    /* first word of gap i's pattern */
    std::vector<uint32_t> rlen;    /* pattern length in nt */
    std::vector<uint64_t> r0;      /* first k-mer of the pattern */
    std::vector<uint64_t> tle, tbad; /* targets of all gaps: little-endian k-mer, never-match mask */
    std::vector<uint32_t> toff, tcnt;
    std::vector<uint8_t> nbmis, fast_ok;
    /* two-pass marshalling: size(i, ...) for every gap in order, then layout(), then set(i, ...) from any thread */
    void resize(size_t n);
    void size(size_t i, size_t swf_len, size_t n_targets) { rlen[i] = (uint32_t)swf_len; tcnt[i] = (uint32_t)n_targets; }
    void layout();
    void set(size_t i, std::string_view source, std::string_view swf_target, const TargetSpan* targets, int nb_mis);
};

/* stage A + post-processing kernels for all gaps (chunked, tiered); fills out[i]; returns MTG_* status */
int device_run(const mtg_index* idx, const mtg_params* p, const FillInput& in, DevBatch& out, mtg_batch_stats* stats);

/* membership scan over packed sequences: host arrays in (words/off/len), bit output as in mtg_index_scan_packed_device; device = 1: the pointers are device pointers */
int scan_run(const mtg_index* idx, const uint64_t* words, size_t nwords, const uint64_t* word_off, const uint32_t* len, size_t nseq, int mode, uint64_t* out_bits, int device_ptrs,
             mtg_scan_stats* st);

/* counts the canonical k-mers of `text` (sequences separated by '\n'): histo[c] = number of distinct k-mers seen c times (c capped at
 * histo.size()-1), kmers/counts = the distinct k-mers seen at least keep_min times (unordered) */
int count_run(const char* text, size_t n, int k, uint32_t keep_min, std::vector<uint64_t>& histo, std::vector<uint64_t>& kmers, std::vector<uint32_t>& counts);

int query_run(const mtg_index* idx, const uint64_t* kmers, size_t n, uint32_t* abund, uint8_t* succ, uint8_t* pred);

void stats_store(const mtg_batch_stats& s);

/* a filled sequence: either an owned string (general path) or a NUL-terminated view into the batch arena (common path) */
struct SeqBuf {
    std::string own;
    const char* p = nullptr;
    uint32_t n = 0;
    bool is_view() const { return p != nullptr; }
    size_t size() const { return p ? n : own.size(); }
    size_t length() const { return size(); }
    const char* data() const { return p ? p : own.data(); }
    const char* c_str() const { return p ? p : own.c_str(); }
    char operator[](size_t i) const { return data()[i]; }
    std::string str() const { return std::string(data(), size()); }
    SeqBuf& operator=(std::string s) { own = std::move(s); p = nullptr; n = 0; return *this; }
    void view(const char* q, uint32_t len) { p = q; n = len; }
    bool operator==(const SeqBuf& o) const { return size() == o.size() && memcmp(data(), o.data(), size()) == 0; }
};
/* storage of the common-path sequences of one batch; must outlive the GapWork results that point into it */
struct FillArena {
    std::vector<char> chars;
};

struct Solution { /* filled_insertion_t, src/Utils.hpp:46-104 */
    SeqBuf seq;
    int nb_errors = 0;
    int target = -1;
    float avg = 0, median = 0;
    int qual = 0, count = 0, rank = 0;
    size_t ab_off = 0, ab_n = 0; /* slice of the batched abundance query */
};
struct GapWork {
    TargetSpan targets; /* targetDictionary in iteration order (storage owned by the caller of fill_gaps) */
    std::string_view source;
    bool anchor_repeated = false, reverse = false;
    int nb_nodes = 0, total_nt = 0, nb_terminal = 0, nb_total_filled = 0;
    bool has_counts = false;
    std::vector<Solution> sols;
    std::string extension;
    void swap_into(GapWork& o) { std::swap(*this, o); } /* used to free o's storage on the calling thread */
};
bool read_sequences(const std::string& path, std::vector<std::pair<std::string, std::string>>& out);
int fill_gaps(const mtg_index* idx, const mtg_params* p, std::vector<GapWork>& gaps, const std::vector<std::string_view>& swf_targets, FillArena& arena,
              mtg_batch_stats* stats_out);
int index_from_kmers(const uint64_t*, const uint32_t*, size_t, int, mtg_index**);
int index_from_reads(const char*, int, int, int, mtg_index**);
int index_save(const mtg_index*, const char*);
int index_load(const char*, mtg_index**);
void index_forget_host_copy(const mtg_index* idx);
int fill_main(int argc, const char* const* argv);

} // namespace mtgi
#endif
