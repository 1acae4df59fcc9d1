/*
 * tests/emu/emu_backend.cpp -- TEST-ONLY stand-in for mtg_gpu_build.hip / mtg_gpu_fill.hip / mtg_gpu_misc.hip: the same device functions
 * (mtg_dev.h / mtg_traverse.h) executed lane by lane on the host, behind the internal interface of
 * mtg_internal.h.  Linked with the product's host translation units (mtg_host.cpp, mtg_cli.cpp) into
 * tests/emu/libmtgfill_emu.so so that the CPU test-suite can run the whole `MindTheGap fill` logic
 * (and sanitizers) without a GPU.  The product library never contains this file.
 */
#include "../../mindthegap_amd/csrc/mtg_internal.h"
#include "../../mindthegap_amd/csrc/mtg_marshal.h"
#include "emu_us.h"
#include "emu_walk.h"
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstring>

using namespace mtg;

namespace mtgi {
/* the scratch of slot s of a launch: zero and work regions shared by the slots (one gap at a time), a raw block of its own, its place in the
 * launch's interleaved region of small arrays */
static GapScratch carve_slot(const FillCfg& cfg, uint8_t* zero, uint8_t* raw, uint8_t* ilv, uint8_t* heads, uint32_t s)
{
    GapScratch S = carve(cfg, zero, raw, ilv, heads, 0);
    S.h = heads + (uint64_t)(s >> 6) * cfg.hd_stride;
    S.lane = s & 63u;
    return S;
}
static std::atomic<unsigned long> emu_gen_multi{0}, emu_gen_multi_host{0}; /* TEST-ONLY: gaps with several reached targets the device function finished / left to the host */
static thread_local char g_err[512] = "";
static thread_local mtg_batch_stats g_stats{};
void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
void stats_store(const mtg_batch_stats& s) { g_stats = s; }

static std::mutex g_us_mtx;
static std::map<const mtg_index*, EmuUStore*> g_us; /* storage of the unitig stores of the live indexes */

int index_from_kmers(const uint64_t* kmers, const uint32_t* ab, size_t n, int k, mtg_index** out)
{
    if (k < 11 || k > 31) { set_error("invalid argument (11 <= k <= 31)"); return MTG_ERR_ARG; }
    if (!emu_legacy_build()) {
        /* the lean build, as index_from_kmer_pieces_lean of mtg_gpu_build.hip */
        mtg_index* idx = new mtg_index();
        idx->dev.k = k;
        bloom_shape(idx->dev.bloom, n, 12.0, k);
        idx->dev.bloom.bits = (uint32_t*)calloc(idx->dev.bloom.nblocks * 16, 4);
        EmuUStore* st = new EmuUStore();
        const EmuLeanStats ls = emu_build_lean(idx->dev, *st, kmers, ab, n);
        idx->info.k = k; idx->info.nb_solid_kmers = ls.nb_solid; idx->info.nb_branching = ls.nb_branching; idx->info.abundance_auto = -1;
        idx->info.nb_saturated = ls.nb_saturated; idx->info.nb_unitigs = ls.nb_unitigs; idx->info.nb_kmers_outside_unitigs = ls.nb_left;
        idx->info.bloom_blocks = idx->dev.bloom.nblocks; idx->info.bloom_minimizer = (uint32_t)idx->dev.bloom.mm;
        idx->info.sparse = idx->dev.us.words ? 1 : 0;
        mtg_build_phase ph{};
        snprintf(ph.name, sizeof ph.name, "%s", ls.late ? "emulated_lean_late" : "emulated_lean");
        ph.units = n;
        idx->build_phases.push_back(ph);
        { std::lock_guard<std::mutex> lk(g_us_mtx); g_us[idx] = st; }
        *out = idx;
        return MTG_OK;
    }
    double load = 0.5;
    for (;;) {
        mtg_index* idx = new mtg_index();
        idx->dev.k = k;
        table_shape(idx->dev.adj, buckets_for(n + n / 8 + 1024, load, 2 * (k - 1), MTG_ADJ_SLOTS), 2 * (k - 1));
        table_shape(idx->dev.abnd, buckets_for(n, load, 2 * k, MTG_ABND_SLOTS), 2 * k);
        idx->dev.adj.slots = (uint64_t*)calloc(idx->dev.adj.nbuckets * MTG_ADJ_SLOTS * 2, 8);
        idx->dev.abnd.slots = (uint64_t*)calloc(idx->dev.abnd.nbuckets * MTG_ABND_SLOTS, 8);
        bloom_shape(idx->dev.bloom, n, 12.0, k);
        idx->dev.bloom.bits = (uint32_t*)calloc(idx->dev.bloom.nblocks * 16, 4);
        int fail = 0;
        uint64_t created = 0;
        uint64_t sat = 0;
        for (size_t i = 0; i < n; i++) { int r = index_insert(idx->dev, kmers[i], ab[i]); fail |= r & 1; created += (r >> 1) & 1; sat += ab[i] > 255u; }
        idx->info.nb_saturated = sat;
        if (!fail) {
            for (size_t i = 0; i < n; i++) { Kmer x = make_kmer(kmers[i], k); build_lookahead(idx->dev, x); Kmer y; y.f = x.r; y.r = x.f; build_lookahead(idx->dev, y); }
            uint64_t nbr = 0, mk1 = kmask(k - 1);
            uint32_t lines = 0;
            for (size_t i = 0; i < n; i++) {
                Kmer x = make_kmer(kmers[i], k);
                nbr += !(popc4(adj_right(idx->dev, x, mk1, lines).out) == 1 && popc4(adj_left(idx->dev, x, mk1, lines).in) == 1);
            }
            idx->info.k = k; idx->info.nb_solid_kmers = created; idx->info.nb_branching = nbr; idx->info.abundance_auto = -1;
            idx->info.bloom_blocks = idx->dev.bloom.nblocks; idx->info.bloom_minimizer = (uint32_t)idx->dev.bloom.mm;
            {
                EmuUStore* st = new EmuUStore();
                idx->info.nb_unitigs = emu_build_unitigs(idx->dev, *st);
                uint64_t* dense_adj = idx->dev.adj.slots;
                uint64_t* dense_abnd = idx->dev.abnd.slots;
                if (emu_sparsify(idx->dev, *st)) { free(dense_adj); free(dense_abnd); idx->info.sparse = 1; } /* the tables now live in *st */
                std::lock_guard<std::mutex> lk(g_us_mtx);
                g_us[idx] = st;
            }
            *out = idx;
            return MTG_OK;
        }
        free(idx->dev.adj.slots); free(idx->dev.abnd.slots); free(idx->dev.bloom.bits); delete idx;
        load *= 0.7;
    }
}
/* the piecewise source of the index reader: gathered here (MTG_LOAD_PIECE = k-mers per piece, as in the device build) */
int index_from_kmer_pieces(size_t n, int k, const KmerFetch& fetch, mtg_index** out)
{
    const size_t env_piece = getenv("MTG_LOAD_PIECE") ? (size_t)atol(getenv("MTG_LOAD_PIECE")) : 0;
    const size_t piece = std::min<size_t>(std::max<size_t>(n, 1), env_piece ? env_piece : (size_t)1 << 26);
    std::vector<uint64_t> km(n);
    std::vector<uint32_t> ab(n);
    for (size_t off = 0; off < n; off += piece) {
        const size_t m = std::min(piece, n - off);
        const uint64_t* hk = nullptr;
        const uint32_t* ha = nullptr;
        if (!fetch(off, m, hk, ha)) return MTG_ERR_IO;
        memcpy(km.data() + off, hk, m * 8);
        memcpy(ab.data() + off, ha, m * 4);
    }
    return index_from_kmers(km.data(), ab.data(), n, k, out);
}
void index_release(mtg_index* idx)
{
    if (!idx) return;
    bool sparse = idx->dev.adj.sp_words != nullptr; /* the tables of the sparse form live in the EmuUStore */
    { std::lock_guard<std::mutex> lk(g_us_mtx); auto it = g_us.find(idx); if (it != g_us.end()) { sparse = sparse || idx->dev.adj.slots == it->second->sp_adj.data(); delete it->second; g_us.erase(it); } }
    if (!sparse) { free(idx->dev.adj.slots); free(idx->dev.abnd.slots); }
    free(idx->dev.bloom.bits);
    for (Workspace& w : idx->ws) for (void* h : w.hptr) free(h);
    delete idx;
}

int query_run(const mtg_index* idx, const uint64_t* kmers, size_t n, uint32_t* abund, uint8_t* succ, uint8_t* pred, Workspace*)
{
    const int k = idx->dev.k;
    const uint64_t mk1 = kmask(k - 1);
    uint32_t lines = 0;
    for (size_t i = 0; i < n; i++) {
        Kmer x = make_kmer(kmers[i] & kmask(k), k);
        if (abund) abund[i] = abundance(idx->dev, x, lines);
        if (succ) succ[i] = (uint8_t)adj_right(idx->dev, x, mk1, lines).out;
        if (pred) pred[i] = (uint8_t)adj_left(idx->dev, x, mk1, lines).in;
    }
    return MTG_OK;
}

/* stand-in for index_from_stream: the same count table and insertion functions, one pass, on the host */
int index_from_stream(ReadStream& rs, int k, int abundance_min, int abundance_max, mtg_index** out)
{
    if (k < 11 || k > 31) { set_error("invalid argument (11 <= k <= 31)"); return MTG_ERR_ARG; }
    uint64_t cap = 1ull << 10;
    while (cap < std::max<size_t>(rs.size_hint(), 1 << 12) / 4) cap <<= 1;
    std::vector<uint64_t> histo(10003, 0);
    for (;; cap <<= 1) {
        std::vector<uint64_t> keys(cap, ~0ULL);
        std::vector<uint32_t> cnts(cap, 0);
        CountTable t{keys.data(), cnts.data(), cap - 1};
        bool full = false;
        if (!rs.rewind()) return MTG_ERR_IO;
        const char* p = nullptr;
        size_t n = 0;
        while (!full && rs.next_block(p, n))
            for (size_t i = 0; i + k <= n && !full; i++) {
                const uint64_t c = kmer_from_ascii(p, i, k);
                if (c != ~0ULL && !count_insert(t, c)) full = true;
            }
        if (rs.failed()) return MTG_ERR_IO;
        if (full) continue;
        std::fill(histo.begin(), histo.end(), 0);
        for (uint64_t i = 0; i < cap; i++) if (keys[i] != ~0ULL) histo[std::min<size_t>(cnts[i], histo.size() - 1)]++;
        int autoc = -1;
        if (abundance_min < 0) { autoc = auto_cutoff(histo, 3); abundance_min = autoc; }
        std::vector<uint64_t> km;
        std::vector<uint32_t> ct;
        for (uint64_t i = 0; i < cap; i++)
            if (keys[i] != ~0ULL && (int64_t)cnts[i] >= std::max(abundance_min, 1) && (abundance_max <= 0 || (int64_t)cnts[i] <= abundance_max)) { km.push_back(keys[i]); ct.push_back(cnts[i]); }
        if (int rc = index_from_kmers(km.data(), ct.data(), km.size(), k, out)) return rc;
        (*out)->info.abundance_min = abundance_min;
        (*out)->info.abundance_auto = autoc;
        return MTG_OK;
    }
}

int scan_run(const mtg_index* idx, const uint64_t* words, size_t, const uint64_t* word_off, const uint32_t* len, size_t nseq, int mode, uint64_t* out_bits, int,
             mtg_scan_stats* st)
{
    const int k = idx->dev.k;
    const uint64_t mk = kmask(k);
    mtg_scan_stats t{};
    for (size_t s = 0; s < nseq; s++) {
        if (len[s] < (uint32_t)k) continue;
        for (uint32_t p = 0; p + k <= len[s]; p++) {
            Kmer x;
            x.r = le_kmer(words + word_off[s], p, mk) ^ (0xAAAAAAAAAAAAAAAAULL & mk);
            x.f = revcomp(x.r, k);
            bool res = bloom_test(idx->dev.bloom, x, k);
            t.n_kmers++;
            t.bloom_positive += res;
            if (res && mode == 1) { uint32_t lines = 0; res = abundance(idx->dev, x, lines) != 0; t.confirmed += res; }
            if (res) out_bits[word_off[s] + (p >> 6)] |= 1ull << (p & 63);
        }
    }
    if (st) *st = t;
    return MTG_OK;
}

/* stand-in for k_nw: the two-row integer DP with the traceback's match count carried forward (scores are exact multiples of 5) */
int nw_run(const mtg_index*, const std::vector<NwPair>& pairs, std::vector<uint32_t>& matches, Workspace*)
{
    matches.assign(pairs.size(), 0);
    for (size_t p = 0; p < pairs.size(); p++) {
        const int na = (int)pairs[p].na, nb = (int)pairs[p].nb;
        const char *a = pairs[p].a, *b = pairs[p].b;
        std::vector<int32_t> sp(nb + 1), sc(nb + 1), mp(nb + 1, 0), mc(nb + 1, 0);
        for (int j = 0; j <= nb; j++) sp[j] = -5 * j;
        for (int i = 1; i <= na; i++) {
            sc[0] = -5 * i; mc[0] = 0;
            for (int j = 1; j <= nb; j++) {
                const bool eq = a[i - 1] == b[j - 1];
                const int diag = sp[j - 1] + (eq ? 10 : -5), del = sp[j] - 5, ins = sc[j - 1] - 5;
                const int best = std::max(std::max(diag, del), ins);
                sc[j] = best;
                mc[j] = best == diag ? mp[j - 1] + (eq ? 1 : 0) : (best == del ? mp[j] : mc[j - 1]);
            }
            sp.swap(sc); mp.swap(mc);
        }
        matches[p] = (uint32_t)mp[nb];
    }
    return MTG_OK;
}

/* grow-only staging blocks like the device build's, in ordinary memory (never shrunk, poisoned when they grow: a batch must write what it reads) */
void* staging_host(Workspace* ws, int slot, size_t bytes)
{
    if (!ws || slot < 0 || slot >= Workspace::NHOST) return nullptr;
    if (ws->hcap[slot] < bytes) {
        free(ws->hptr[slot]);
        const size_t want = bytes + bytes / 4 + 4096;
        ws->hptr[slot] = malloc(want);
        if (!ws->hptr[slot]) { ws->hcap[slot] = 0; return nullptr; }
        memset(ws->hptr[slot], 0xA5, want);
        ws->hcap[slot] = want;
    }
    return ws->hptr[slot];
}
void* pinned_alloc(size_t bytes)
{
    void* p = malloc(bytes ? bytes : 8);
    if (p) memset(p, 0xA5, bytes ? bytes : 8); /* poisoned: a record must be written before it is read */
    return p;
}
void pinned_free(void* p) { free(p); }
int host_register(void*, size_t) { return MTG_OK; } /* nothing to lock on the host */
int host_unregister(void*) { return MTG_OK; }
/* the emulated device's memory is host memory */
int device_download(const mtg_index*, void* host_dst, const void* dev_src, size_t bytes) { memcpy(host_dst, dev_src, bytes); return MTG_OK; }
int device_upload(const mtg_index*, void* dev_dst, const void* host_src, size_t bytes) { memcpy(dev_dst, host_src, bytes); return MTG_OK; }
int batch_upload(const mtg_index*, FillInput&) { return MTG_OK; } /* nothing to upload: the "device" reads the host blocks */
void batch_release_device(FillInput&) {}

int index_dump(const mtg_index* idx, IndexDump& d)
{
    d.k = idx->dev.k; d.abundance_min = idx->info.abundance_min; d.abundance_auto = idx->info.abundance_auto;
    d.nb_solid = idx->info.nb_solid_kmers; d.nb_branching = idx->info.nb_branching; d.nb_saturated = idx->info.nb_saturated;
    d.n_words = idx->dev.us.nwords; d.n_unitigs = idx->dev.us.nunitigs;
    d.words.clear(); d.ab.clear(); d.left_k.clear(); d.left_a.clear();
    if (d.n_words) { d.words.assign(idx->dev.us.words, idx->dev.us.words + d.n_words + 8); d.ab.assign(idx->dev.us.ab, idx->dev.us.ab + (d.n_words + 8) * 32); }
    uint32_t lines = 0;
    const uint64_t nslots = idx->dev.abnd.nbuckets * MTG_ABND_SLOTS;
    for (uint64_t s = 0; s < nslots; s++) {
        uint64_t c;
        const uint32_t ab = abnd_slot_kmer(idx->dev.abnd, s, c);
        if (ab && !(idx->dev.us.nwords && kmer_stored(idx->dev, c, lines))) { d.left_k.push_back(c); d.left_a.push_back(ab); }
    }
    return MTG_OK;
}
/* every solid k-mer of a dump: the stored ones expanded, then the others */
static void dump_kmers(const IndexDump& d, std::vector<uint64_t>& km, std::vector<uint32_t>& ab)
{
    const uint64_t mk = kmask(d.k);
    for (uint64_t h = 0; h < d.n_words;) {
        const uint64_t len = d.words[h];
        for (uint64_t i = 0; i + d.k <= len; i++) {
            const uint64_t p = (h + 1) * 32 + i, lo = d.words[p >> 5] >> (2 * (p & 31)), hi = (p & 31) ? d.words[(p >> 5) + 1] << (64 - 2 * (p & 31)) : 0;
            const uint64_t r = ((lo | hi) & mk) ^ (0xAAAAAAAAAAAAAAAAULL & mk), f = revcomp(r, d.k);
            km.push_back(f < r ? f : r);
            ab.push_back(d.ab[p]);
        }
        h += 1 + (len + 31) / 32;
    }
    km.insert(km.end(), d.left_k.begin(), d.left_k.end());
    ab.insert(ab.end(), d.left_a.begin(), d.left_a.end());
}
void* adj_prealloc_begin(uint64_t, int) { return nullptr; } /* the emulation has no device memory to allocate ahead */
void adj_prealloc_drop(void*) {}
int index_from_dump(const IndexDump& d0, mtg_index** out)
{
    /* the emulation rebuilds from the k-mers (its tables are host memory); the device build derives the tables from the store itself */
    IndexDump local;
    if (d0.ab_read) { /* a container that is being read: the abundance bytes in pieces, as the device build asks for them */
        local = d0;
        local.ab.assign(d0.words.size() * 32, 0);
        const size_t piece = 1000;
        for (size_t off = 0; off < local.ab.size(); off += piece)
            if (!d0.ab_read(off, std::min(piece, local.ab.size() - off), local.ab.data() + off)) { set_error("index container: short read"); return MTG_ERR_FORMAT; }
        local.ab_read = nullptr;
    }
    const IndexDump& d = d0.ab_read ? local : d0;
    std::vector<uint64_t> km;
    std::vector<uint32_t> ab;
    dump_kmers(d, km, ab);
    if (km.size() != d.nb_solid) { set_error("index container: %zu k-mers, %llu announced", km.size(), (unsigned long long)d.nb_solid); return MTG_ERR_FORMAT; }
    if (int rc = index_from_kmers(km.data(), ab.data(), km.size(), d.k, out)) return rc;
    (*out)->info.abundance_min = d.abundance_min;
    (*out)->info.abundance_auto = d.abundance_auto;
    return MTG_OK;
}
/* stand-in for k_fmt_size / k_fmt_scan / k_fmt_write: the same formatter (mtg_format.h), one site after the other.  The emulation's arena is host
 * memory: the records' seq pointers are followed as they are. */
int format_run(const mtg_index*, const FormatIn& fi, FormatOut& out)
{
    const size_t n = fi.n;
    out.n = n; out.n_simple = 0;
    out.complex_sites.clear();
    for (int s = 0; s < FMT_STREAMS; s++) { out.complex_off[s].clear(); out.bytes[s] = 0; }
    std::vector<FmtSite> sites(n);
    std::vector<char> simple(n, 0);
    std::vector<FmtRec> rec(n);
    uint64_t cur[FMT_STREAMS] = {0, 0, 0};
    for (size_t i = 0; i < n; i++) {
        const mtg_gap_result& r = fi.res[i];
        FmtSite& t = sites[i];
        bool ok = r.n_filled == 1 && r.filled == fi.fil + i;
        if (ok) {
            const mtg_filled& f = fi.fil[i];
            const uintptr_t q = (uintptr_t)f.seq;
            ok = q >= (uintptr_t)fi.host_seq && q < (uintptr_t)fi.host_seq + fi.seq_used;
            if (ok) {
                t.name = fi.host_text + fi.name_off[i]; t.name_len = fi.name_len[i];
                t.source = fi.host_text + fi.source_off[i]; t.source_len = fi.source_len[i];
                t.seq = f.seq; t.seq_len = fmt_strlen(f.seq);
                t.nb_nodes = r.nb_nodes; t.total_nt = r.total_nt; t.nb_terminal = r.nb_terminal; t.has_counts = r.has_solution_counts;
                t.nb_total_filled = r.nb_total_filled; t.nb_reported = r.nb_reported;
                t.qual = f.qual; t.solution_count = f.solution_count; t.avg = f.avg_coverage; t.median = f.median_coverage;
                ok = fmt_site_simple(t);
            }
        }
        simple[i] = ok;
        FmtCount c;
        c.n[0] = c.n[1] = c.n[2] = 0;
        if (ok) format_site(c, t);
        for (int s = 0; s < FMT_STREAMS; s++) { rec[i].size[s] = c.n[s]; rec[i].off[s] = cur[s]; cur[s] += c.n[s]; }
        if (!ok) { out.complex_sites.push_back((uint32_t)i); for (int s = 0; s < FMT_STREAMS; s++) out.complex_off[s].push_back(rec[i].off[s]); }
        else out.n_simple++;
    }
    for (int s = 0; s < FMT_STREAMS; s++) {
        out.bytes[s] = cur[s];
        if (out.cap[s] < cur[s] + 64) { free(out.text[s]); out.cap[s] = (size_t)cur[s] + 4096; out.text[s] = (char*)malloc(out.cap[s]); memset(out.text[s], 0xA5, out.cap[s]); }
    }
    for (size_t i = 0; i < n; i++) {
        if (!simple[i]) continue;
        FmtWrite w;
        for (int s = 0; s < FMT_STREAMS; s++) w.p[s] = out.text[s] + rec[i].off[s];
        format_site(w, sites[i]);
        for (int s = 0; s < FMT_STREAMS; s++) if (w.p[s] != out.text[s] + rec[i].off[s] + rec[i].size[s]) { set_error("formatter: the size pass and the write pass disagree at site %zu", i); return MTG_ERR_OVERFLOW; }
    }
    return MTG_OK;
}
int workspace_arena_download(const mtg_index*, Workspace*, char*, uint64_t) { return MTG_OK; } /* the emulation's arena is the host's */

int index_export(const mtg_index* idx, const std::function<bool(const uint64_t*, const uint32_t*, size_t)>& sink)
{
    std::vector<uint64_t> k;
    std::vector<uint32_t> a;
    IndexDump d;
    index_dump(idx, d);
    dump_kmers(d, k, a);
    if (k.size() != idx->info.nb_solid_kmers) { set_error("index export: %zu k-mers in the table, %llu expected", k.size(), (unsigned long long)idx->info.nb_solid_kmers); return MTG_ERR_FORMAT; }
    const size_t piece = 1000; /* several pieces, like the device build */
    for (size_t off = 0; off < k.size(); off += piece)
        if (!sink(k.data() + off, a.data() + off, std::min(piece, k.size() - off))) { set_error("index export: the writer failed"); return MTG_ERR_IO; }
    return MTG_OK;
}

/* Same shape as the device build: a launch covers the gaps still to do at a scratch tier (several launches per tier with MTG_MAX_CHUNK);
 * per launch: traversal and post-processing of every slot, layout of the results by prefix sums in slot order, emission (records, ASCII,
 * dense contig arrays of the gaps that need the host), arenas grown and the launch emitted again when they are too small. */
int device_run(const mtg_index* idx, const mtg_params* p, const FillInput& in, ResultSink& sink, DevBatch& special, mtg_batch_stats* stats, const std::function<void()>* while_busy)
{
    if (while_busy) (*while_busy)();
    /* TEST-ONLY fault injection: every batch on the named pretended device fails (the tool's multi-device driver must stop and report) */
    if (const char* e = getenv("MTG_EMU_FAIL_DEVICE"))
        if (atoi(e) == idx->device) { set_error("injected failure on device %d", idx->device); return MTG_ERR_NOMEM; }
    const int k = idx->dev.k;
    const size_t n = in.src.size();
    /* stand-in for k_encode_targets, or (mtg_fill_text) for k_marshal_text and k_marshal_targets: the same per-gap functions, from the text block */
    const size_t n_targets = in.text_mode ? (size_t)in.n_text_targets : in.traw.size() / TARGET_SLOT;
    std::vector<uint64_t> tle(n_targets), tbad(n_targets), text_rw;
    if (in.text_mode) {
        const uint8_t* c = (const uint8_t*)in.block_c;
        const uint8_t* text = in.text_direct ? (const uint8_t*)in.text_direct : c + FillInput::text_block_off(n, n_targets, 5); /* mtg_host_register: the block stays where the caller has it */
        const uint64_t* soff = (const uint64_t*)(c + FillInput::text_block_off(n, n_targets, 0));
        const uint64_t* poff = (const uint64_t*)(c + FillInput::text_block_off(n, n_targets, 1));
        const uint64_t* doff = (const uint64_t*)(c + FillInput::text_block_off(n, n_targets, 2));
        const uint32_t* slen = (const uint32_t*)(c + FillInput::text_block_off(n, n_targets, 3));
        const uint32_t* dlen = (const uint32_t*)(c + FillInput::text_block_off(n, n_targets, 4));
        text_rw.assign(in.n_rwords + 1, 0xA5A5A5A5A5A5A5A5ull);
        FillInput& w = const_cast<FillInput&>(in); /* block B exists on the device only: here it is this vector */
        w.rwords.p = text_rw.data(); w.rwords.n = in.n_rwords;
        for (size_t g = 0; g < n; g++) marshal_text_gap(text, soff[g], slen[g], poff[g], in.rlen[g], k, text_rw.data() + in.roff[g], in.src[g], in.r0[g], in.rlen[g], in.fast_ok[g]);
        for (size_t t = 0; t < n_targets; t++) marshal_text_target(text, doff[t], dlen[t], k, tle[t], tbad[t]);
    } else
        for (size_t t = 0; t < n_targets; t++) encode_target(in.traw.data() + t * TARGET_SLOT, k, tle[t], tbad[t]);
    /* stand-in for k_post_index: the piece index of the batch's dictionaries (mtg_post.h), built whenever a batch has a target (the device asks for
     * sixteen a gap on average): the emulation's cross-check in post_gap compares its result with the pass over every target */
    std::vector<uint32_t> pi_head, pi_next;
    uint32_t pi_mask = 0;
    if (n_targets && n_targets < (1ull << 30)) {
        uint64_t cap = 1024;
        while (cap < 4 * n_targets) cap <<= 1;
        pi_mask = (uint32_t)(cap - 1);
        pi_head.assign(cap, (uint32_t)POST_INDEX_NIL);
        pi_next.assign(n_targets * 4, (uint32_t)POST_INDEX_NIL);
        for (size_t g = 0; g < n; g++)
            for (uint32_t t = 0; t < in.tcnt[g]; t++)
                post_index_add(pi_head.data(), pi_next.data(), pi_mask, (uint32_t)g, in.toff[g] + t, tle[in.toff[g] + t], tbad[in.toff[g] + t], in.nbmis[g], k);
    }
    special.chunks.clear();
    special.special.clear();
    sink.seq_used = 0;
    sink.ext_used = 1;
    sink.n_filled = 0;
    sink.in_gap_order = true;
    mtg_batch_stats st = stats ? *stats : mtg_batch_stats{};
    const bool want_records = sink.res != nullptr;
    uint64_t cur_seq = 0, cur_ext = 1;
    std::vector<uint32_t> todo;
    size_t n_todo = n, launches = 0;
    const size_t env_chunk = getenv("MTG_MAX_CHUNK") ? (size_t)atol(getenv("MTG_MAX_CHUNK")) : 0;
    for (int tier = 0; tier <= MTG_MAX_TIER && n_todo; tier++) {
        FillCfg cfg = make_cfg(k, p->max_nodes, p->max_depth, p->end_rule_nonbranching, tier);
        if (getenv("MTG_NO_DEFER") || !idx->dev.us.nwords) cfg.cmd_cap = 0;
        std::vector<uint32_t> retry;
        const size_t chunk = env_chunk ? std::min(env_chunk, n_todo) : n_todo;
        for (size_t base = 0; base < n_todo; base += chunk) {
            const uint32_t m = (uint32_t)std::min(chunk, n_todo - base);
            const bool identity = todo.empty() && base == 0 && m == n;
            std::vector<uint32_t> ids(m);
            for (uint32_t s = 0; s < m; s++) ids[s] = todo.empty() ? (uint32_t)(base + s) : todo[base + s];
            if (!identity) sink.in_gap_order = false;
            launches++;
            /* every slot keeps its own scratch until the launch has been emitted, like on the device; the zero region is shared by the
             * slots here (one lane at a time) and must come back clean from every gap */
            std::vector<uint8_t> zero(cfg.zero_stride, 0), ilv(cfg.ilv_stride), fp_table(FP_SLOTS * 64);
            /* MTG_EMU_NO_POISON=1 (bench.py's same-algorithm CPU number): the scratch of a slot is not filled with a pattern first -- the fill is
             * the emulator's check that nothing is read before it is written, and costs more than the walk itself */
            static const bool poison = getenv("MTG_EMU_NO_POISON") == nullptr;
            struct RawBuf { uint8_t* p; uint8_t* data() const { return p; } };
            std::vector<RawBuf> raws(m);
            static thread_local std::vector<uint8_t> raw_arena; /* kept between launches: a fresh allocation per slot spends its time in page faults */
            const size_t raw_each = ((size_t)cfg.raw_stride + 64 + 63) & ~(size_t)63;
            if (raw_arena.size() < raw_each * m) raw_arena.resize(raw_each * m);
            /* the small per-gap arrays, interleaved over 64 slots as on the device */
            static thread_local std::vector<uint8_t> head_arena;
            const size_t head_bytes = (size_t)((m + 63) / 64) * cfg.hd_stride;
            if (head_arena.size() < head_bytes) head_arena.resize(head_bytes);
            if (poison) memset(head_arena.data(), 0xCD, head_bytes);
            uint8_t* const heads = head_arena.data();
            std::vector<SlotRec> recs(m);
            for (uint32_t s = 0; s < m; s++) {
                const size_t g = ids[s];
                raws[s].p = raw_arena.data() + raw_each * s;
                if (poison) memset(raws[s].data(), 0xCD, cfg.raw_stride + 64);
#ifdef MTG_XCHECK
                memset(fp_table.data(), 0, fp_table.size());
#endif
                GapScratch S = carve_slot(cfg, zero.data(), raws[s].data(), ilv.data(), heads, s);
                S.fp = fp_table.data();
                S.snp_fast = getenv("MTG_NO_SNP_FAST") ? 0 : 1;
                SwfPattern R;
                R.words = in.rwords.data() + in.roff[g];
                R.rlen = in.rlen[g];
                R.r0 = in.r0[g];
                SlotRec& r = recs[s];
                memset(&r, 0, sizeof r);
                st.n_parked_gaps += emu_walk(idx->dev, cfg, S, in.src[g], R, r.o); /* emu_walk.h: the device's launch sequence */
                {   /* k_copy */
                    uint64_t target = ~0ull;
                    if (!in.want_all_contigs && !getenv("MTG_NO_LEAN") && cfg.cmd_cap && r.o.status == GAP_OK && in.tcnt[g] == 1u && in.fast_ok[g] && tbad[in.toff[g]] == 0ull)
                        target = rev_fields64(tle[in.toff[g]]) >> (64 - 2 * k);
                    copy_gap(idx->dev, cfg, S, r.o, target);
                }
                st.copy_words += r.o.copy_words; st.copy_cmds += r.o.n_cmds;
#ifdef MTG_XCHECK /* the device relies on every gap handing the zero region back clean */
                for (uint8_t z : zero) if (z) { set_error("gap %zu: zero-initialised scratch not restored (status %u)", g, r.o.status); return MTG_ERR_OVERFLOW; }
#endif
                if (r.o.status == GAP_OK) {
                    PostTargets T;
                    T.le = tle.data() + in.toff[g];
                    T.bad = tbad.data() + in.toff[g];
                    T.n = in.tcnt[g];
                    T.nb_mis = in.nbmis[g];
                    T.fast_ok = in.fast_ok[g];
                    if (!pi_head.empty()) { T.pi_head = pi_head.data(); T.pi_next = pi_next.data(); T.pi_mask = pi_mask; T.gbase = in.toff[g]; T.gid = (uint32_t)g; }
                    uint32_t hist[256] = {0};
                    std::vector<uint64_t> tile(POST_TILE + 2);
                    uint64_t blk[64];
                    post_gap(idx->dev, cfg, S, r.o, T, hist, tile.data(), blk, r.p);
                    {   /* the device's k_post_lean (mtg_post.h: the lean gap by a group of lanes, here one) must leave the same record */
                        uint32_t hist2[256] = {0};
                        LeanWork lw;
                        PostOut p2;
                        memset(&p2, 0, sizeof p2);
                        const bool is_lean = post_lean_accumulate<1>(idx->dev, cfg, S, r.o, 0u, hist2, lw);
                        if (is_lean) post_lean_finish<1>(lw, 0u, hist2, p2);
                        if (is_lean != (r.p.lean != 0) || (is_lean && memcmp(&p2, &r.p, sizeof p2) != 0)) { set_error("gap %zu: the lean form of the post-processing and the general one disagree", g); return MTG_ERR_OVERFLOW; }
                    }
                } else st.n_retried_gaps++;
                emit_plan(r.o, r.p, in.want_all_contigs, k, r.nw, r.nc, r.asc, r.ext);
            }
            /* stand-in for k_scan1 / k_scan2: exclusive prefix sums in slot order on top of the batch's cursors */
            PartTot tot;
            memset(&tot, 0, sizeof tot);
            tot.begin[2] = cur_seq; tot.begin[3] = cur_ext;
            uint64_t c0 = 0, c1 = 0, c2 = cur_seq, c3 = cur_ext;
            std::vector<uint32_t> rlist, glist;
            for (uint32_t s = 0; s < m; s++) {
                SlotRec& r = recs[s];
                r.wbase = c0; r.cbase = c1; r.abase = c2; r.ebase = c3;
                r.fpos = tot.n_filled;
                c0 += r.nw; c1 += r.nc; c2 += r.asc; c3 += r.ext;
                tot.lines += r.o.lines; tot.store_runs += r.o.store_reads; tot.run_nt += r.o.run_nt; tot.contig_words += r.o.n_words;
                if (r.o.status != GAP_OK) { r.rpos = (uint32_t)rlist.size(); rlist.push_back(s); continue; }
                tot.contig_nt += r.o.total_nt; tot.post_lines += r.p.lines; tot.cov_kmers += r.p.ab_n; if (r.p.direct) tot.cov_direct += r.p.ab_n;
                tot.n_filled += r.asc != 0; tot.n_ext += r.ext != 0; tot.n_lean += r.p.lean != 0;
                if (!r.p.lean) { if (r.o.n_cmds) { tot.copy_words_exec += r.o.copy_words; tot.copy_cmds_exec += r.o.n_cmds; } tot.scan_words += r.o.n_words; }
                if (r.nc) { r.gpos = (uint32_t)glist.size(); glist.push_back(s); }
            }
            tot.end[0] = c0; tot.end[1] = c1; tot.end[2] = c2; tot.end[3] = c3;
            tot.n_retry = (uint32_t)rlist.size(); tot.n_general = (uint32_t)glist.size();
            cur_seq = c2; cur_ext = c3;
            /* arenas: grown like in the device build (the first emission is skipped here: nothing would be written anyway) */
            if (want_records && tot.end[2] > sink.seq_cap) {
                const uintptr_t old = (uintptr_t)sink.seq, old_end = old + sink.seq_cap;
                const uintptr_t old_fil = (uintptr_t)sink.fil, old_fil_end = old_fil + n * sizeof(mtg_filled);
                if (!sink.grow_seq || !sink.grow_seq((size_t)tot.end[2], (size_t)tot.begin[2])) { set_error("sequence buffer too small: %llu bytes needed", (unsigned long long)tot.end[2]); return MTG_ERR_ARG; }
                if (want_records && launches > 1 && (uintptr_t)sink.seq != old)
                    for (size_t i = 0; i < n; i++) { const uintptr_t q = (uintptr_t)sink.fil[i].seq; if (q >= old && q < old_end) sink.fil[i].seq = sink.seq + (q - old); }
                /* as device_run: the arena is one block with the records, the filled records moved with it */
                if (want_records && launches > 1 && (uintptr_t)sink.fil != old_fil)
                    for (size_t i = 0; i < n; i++) { const uintptr_t q = (uintptr_t)sink.res[i].filled; if (q >= old_fil && q < old_fil_end) sink.res[i].filled = (const mtg_filled*)((const char*)sink.fil + (q - old_fil)); }
            }
            if (want_records && tot.end[3] > sink.ext_cap) {
                const uintptr_t old = (uintptr_t)sink.ext, old_end = old + sink.ext_cap;
                if (!sink.grow_ext || !sink.grow_ext((size_t)tot.end[3], (size_t)tot.begin[3])) { set_error("extension buffer too small: %llu bytes needed", (unsigned long long)tot.end[3]); return MTG_ERR_NOMEM; }
                if (want_records && launches > 1 && (uintptr_t)sink.ext != old)
                    for (size_t i = 0; i < n; i++) { const uintptr_t q = (uintptr_t)sink.res[i].extension; if (q >= old && q < old_end) sink.res[i].extension = sink.ext + (q - old); }
            }
            /* stand-in for k_emit */
            HostChunk* hc = nullptr;
            SlotRec* h_rec = nullptr;
            uint64_t* h_w = nullptr;
            uint32_t* h_m = nullptr;
            /* the dense arrays start deliberately small, as on the device where they are sized by the last need: the first emission writes
             * only the gaps that fit, the totals say what is missing, the arrays grow and the launch is emitted again (under the sanitizers a
             * write past the small arrays would be seen) */
            std::vector<uint64_t> dw(std::min<uint64_t>(c0, 16) + 1);
            std::vector<uint32_t> dm(5 * std::min<uint64_t>(c1, 2) + 1);
            std::vector<mtg_gap_result> dres(m);
            std::vector<mtg_filled> dfil(m);
            std::vector<char> none(64);
            EmitDev D;
            EmitHost H;
            for (int pass = 0; pass < 2; pass++) {
                D.seq = sink.seq ? sink.seq : none.data(); D.ext = sink.ext ? sink.ext : none.data(); /* the emulated device writes the host arena directly; a caller's "device" buffer is mirrored below */
                D.seq_cap = sink.seq_cap; D.ext_cap = sink.ext_cap;
                D.res = dres.data(); D.fil = dfil.data();
                D.dense_words = dw.data(); D.dense_meta = dm.data();
                D.dense_cap_words = dw.size() - 1; D.dense_cap_contigs = (dm.size() - 1) / 5;
                const bool want_wire = sink.wire_dev != nullptr && identity && tier == 0;
                D.wire = want_wire ? (uint8_t*)sink.wire_dev : nullptr; D.wire_cap = sink.wire_cap; D.wire_tag = sink.wire_tag;
                D.tot = &tot; D.wire_gaps = m;
                if (want_wire) memset(sink.wire_dev, 0, sizeof(mtg_wire_header));
                H.seq = sink.seq; H.ext = sink.ext; H.fil = sink.fil;
                for (uint32_t s = 0; s < m; s++) {
                    GapScratch S = carve_slot(cfg, zero.data(), raws[s].data(), ilv.data(), heads, s);
                    emit_gap(idx->dev.us, cfg, S, recs[s], in.flags[ids[s]], s, ids[s], k, D, H);
                    if (emit_is_lean(recs[s], D) && recs[s].asc && recs[s].abase + recs[s].asc <= D.seq_cap) {
                        /* the device's k_emit_lean (mtg_emit.h: emit_lean, a group of lanes per gap, here one) must write the same bytes and records */
                        std::vector<char> tmp(recs[s].asc + 32, 'x');
                        mtg_gap_result g2;
                        mtg_filled f2;
                        memset(&g2, 0, sizeof g2); memset(&f2, 0, sizeof f2);
                        EmitDev D2 = D;
                        D2.seq = (char*)((uintptr_t)(tmp.data() + 16) - (uintptr_t)recs[s].abase); D2.seq_cap = recs[s].abase + recs[s].asc;
                        D2.res = &g2; D2.fil = &f2;
                        emit_lean<1>(idx->dev.us, cfg, S, recs[s], recs[s].abase, in.flags[ids[s]], 0, ids[s], k, D2, H, 0u);
                        /* field by field: the structs have padding (behind n_filled, at the end of mtg_filled) that neither form writes */
                        const mtg_gap_result& g1 = D.res[s];
                        const mtg_filled& f1 = D.fil[s];
                        const bool same_res = g2.nb_nodes == g1.nb_nodes && g2.total_nt == g1.total_nt && g2.nb_terminal == g1.nb_terminal && g2.has_solution_counts == g1.has_solution_counts &&
                                              g2.nb_total_filled == g1.nb_total_filled && g2.nb_reported == g1.nb_reported && g2.n_filled == g1.n_filled && g2.filled == g1.filled && g2.extension == g1.extension;
                        const bool same_fil = f2.seq == f1.seq && f2.nb_errors_in_anchor == f1.nb_errors_in_anchor && f2.target_index == f1.target_index && f2.avg_coverage == f1.avg_coverage &&
                                              f2.median_coverage == f1.median_coverage && f2.qual == f1.qual && f2.solution_count == f1.solution_count && f2.solution_rank == f1.solution_rank;
                        if (memcmp(tmp.data() + 16, D.seq + recs[s].abase, recs[s].asc) != 0 || !same_res || !same_fil) {
                            set_error("gap %u: the lean form of the result kernel and the general one disagree", ids[s]);
                            return MTG_ERR_OVERFLOW;
                        }
                    }
                }
                if (c0 <= D.dense_cap_words && c1 <= D.dense_cap_contigs) break;
                dw.assign(c0 + 1, 0);
                dm.assign(5 * c1 + 1, 0);
            }
            {   /* stand-in for k_wire_sum and for the host's view of the launch (device_run of mtg_gpu_fill.hip) */
                const WireLayout wl = wire_layout(m, tot.n_filled, tot.end[2], tot.end[3]);
                const bool wired = sink.wire_dev != nullptr && identity && tier == 0 && wl.total <= sink.wire_cap && tot.n_retry == 0 && tot.n_general == 0;
                if (sink.wire_dev && identity && tier == 0) { sink.wire_ok = wired; sink.wire_bytes = wired ? wl.total : 0; }
                if (wired) {
                    mtg_wire_header* h = (mtg_wire_header*)sink.wire_dev;
                    const uint64_t* w = (const uint64_t*)((const uint8_t*)sink.wire_dev + sizeof(mtg_wire_header));
                    uint64_t sum = 0;
                    for (uint64_t i = 0; i < (h->total_bytes - sizeof(mtg_wire_header)) / 8; i++) sum += wire_word_sum(w[i], i);
                    h->checksum = sum;
                    /* the sequences were written into the payload: the host arena gets its copy from there */
                    if (sink.seq && tot.end[2] > tot.begin[2]) memcpy(sink.seq + tot.begin[2], (const char*)sink.wire_dev + wl.o_seq + tot.begin[2], tot.end[2] - tot.begin[2]);
                }
            }
            if (want_records)
                for (uint32_t s = 0; s < m; s++) { sink.res[ids[s]] = dres[s]; if (recs[s].asc) sink.fil[ids[s]] = dfil[s]; }
            /* a caller's "device" buffer next to a host copy: the arena is the same bytes in both */
            if (sink.seq_dev && sink.seq && sink.seq_dev != sink.seq && tot.end[2] > tot.begin[2] && tot.end[2] <= sink.seq_cap)
                memcpy(sink.seq_dev + tot.begin[2], sink.seq + tot.begin[2], tot.end[2] - tot.begin[2]);
            st.index_lines += tot.lines; st.contig_nt += tot.contig_nt; st.store_runs += tot.store_runs; st.run_nt += tot.run_nt; st.post_lines += tot.post_lines;
            st.contig_words += tot.contig_words; st.coverage_kmers += tot.cov_kmers; st.coverage_direct_kmers += tot.cov_direct; st.n_lean_gaps += tot.n_lean; st.dense_words += c0;
            st.copy_words_executed += tot.copy_words_exec; st.copy_cmds_executed += tot.copy_cmds_exec; st.post_scanned_words += tot.scan_words;
            sink.seq_used = tot.end[2];
            sink.ext_used = tot.end[3];
            sink.n_filled += tot.n_filled;
            for (uint32_t s2 : rlist) retry.push_back(ids[s2]);
            if (tot.n_retry) sink.in_gap_order = false;
            if (tot.n_general) {
                special.chunks.emplace_back(new HostChunk());
                hc = special.chunks.back().get();
                hc->carve(m, c0, c1, h_rec, h_w, h_m);
                memcpy(h_rec, recs.data(), (size_t)m * sizeof(SlotRec));
                memcpy(h_w, dw.data(), c0 * 8);
                h_w[c0] = 0;
                memcpy(h_m, dm.data(), c1 * 20);
                if (!identity) hc->gap_of = ids;
                const uint32_t chunk_id = (uint32_t)special.chunks.size() - 1;
                std::vector<uint32_t> pslots;
                uint32_t grank = 0;
                for (uint32_t s2 : glist) {
                    special.special.push_back(SpecialGap{ids[s2], chunk_id, s2, grank++});
                    if (!getenv("MTG_HOST_PATHS") && !in.want_all_contigs && recs[s2].p.fast == 0 && recs[s2].p.nb_terminal > 0) pslots.push_back(s2);
                }
                if (!pslots.empty()) { /* stand-in for k_paths */
                    hc->paths.assign(pslots.size() * (size_t)PATHS_WORDS, 0);
                    hc->path_of.assign(m, -1);
                    for (size_t g2 = 0; g2 < pslots.size(); g2++) {
                        PathsWork pw;
                        GapScratch S = carve_slot(cfg, zero.data(), raws[pslots[g2]].data(), ilv.data(), heads, pslots[g2]);
                        paths_gap(cfg, S, recs[pslots[g2]].o, k, pw, hc->paths.data() + g2 * (size_t)PATHS_WORDS);
                        hc->path_of[pslots[g2]] = (int32_t)g2;
                    }
                }
                /* stand-in for k_general (mtg_general.h): the multi-contig gaps finished by the device function, one lane.  The host's path sees the
                 * same gaps (their contigs and paths are here anyway) and run_general compares the two answers: gen_check. */
                if (!getenv("MTG_HOST_PATHS") && !getenv("MTG_HOST_GENERAL") && !in.want_all_contigs && pslots.size() == glist.size()) {
                    const size_t ng = glist.size();
                    /* small arenas: a launch with many or long solutions overflows them and those gaps fall back to the host, as on the device */
                    const bool tiny = getenv("MTG_EMU_GEN_TINY") != nullptr;
                    std::vector<GenSol> sols(tiny ? 3 : 4 * ng + 64);
                    std::vector<char> ascii(tiny ? 600 : ng * 1280 + (1u << 16));
                    std::vector<uint64_t> tmp(tiny ? 40 : ng * 64 + (1u << 14));
                    GenCtl ctl{};
                    hc->gen_gaps.assign(ng, GenGap{});
                    GenDev GD{};
                    GD.gaps = hc->gen_gaps.data(); GD.sols = sols.data(); GD.ascii = ascii.data(); GD.tmp = tmp.data(); GD.bnd = nullptr;
                    GD.cap_sols = sols.size(); GD.cap_ascii = ascii.size(); GD.cap_tmp = tmp.size(); GD.cap_bnd = ~0ull; GD.ctl = &ctl;
                    for (size_t g2 = 0; g2 < ng; g2++) {
                        const uint32_t s2 = glist[g2], gi = ids[s2];
                        GenWork* gw = new GenWork();
                        GapScratch S = carve_slot(cfg, zero.data(), raws[s2].data(), ilv.data(), heads, s2);
                        gen_gap(idx->dev, cfg, S, recs[s2].o, k, hc->paths.data() + (size_t)hc->path_of[s2] * PATHS_WORDS, in.tcnt[gi], in.fast_ok[gi] != 0, in.src[gi], in.flags[gi], GD,
                                (uint32_t)g2, *gw);
                        delete gw;
                    }
                    hc->gen_sols.assign(sols.begin(), sols.begin() + std::min<size_t>(ctl.n_sols, sols.size()));
                    hc->gen_ascii.assign(ascii.begin(), ascii.begin() + std::min<size_t>(ctl.ascii_bytes, ascii.size()));
                    hc->gen_ascii.push_back(0);
                    hc->gen_check = true;
                    for (size_t g2 = 0; g2 < ng; g2++) {
                        if (hc->gen_gaps[g2].status == GEN_OK) st.n_general_device++; else st.n_general_host++;
                        if (hc->gen_gaps[g2].status == GEN_OK && hc->gen_gaps[g2].n_groups > 1) emu_gen_multi++;
                        if (hc->gen_gaps[g2].status != GEN_OK && in.tcnt[ids[glist[g2]]] > 1) emu_gen_multi_host++;
                    }
                } else st.n_general_host += glist.size();
            }
            st.n_launches++;
            st.seq_bytes += tot.end[2] - tot.begin[2];
        }
        todo.swap(retry);
        n_todo = todo.size();
    }
    if (n_todo) { set_error("%zu gap(s) exceeded the largest traversal scratch tier", n_todo); return MTG_ERR_OVERFLOW; }
    if (launches > 1) sink.in_gap_order = false;
    if (stats) *stats = st;
    return MTG_OK;
}
} // namespace mtgi

extern "C" {
/* TEST-ONLY: stored unitigs of the lean builds so far whose two walkers met in the middle / whose owner walked the whole chain */
void emu_walk_counts(unsigned long* out) { out[0] = emu_walks_met; out[1] = emu_walks_whole; }
/* TEST-ONLY: multi-contig gaps with several reached targets finished by the device function / multi-target gaps it left to the host */
void emu_gen_counts(unsigned long* out) { out[0] = mtgi::emu_gen_multi; out[1] = mtgi::emu_gen_multi_host; }
const char* mtg_last_error(void) { return mtgi::g_err; }
/* TEST-ONLY: MTG_EMU_DEVICES pretends that many devices exist, so that the tool's multi-device driver can be exercised on the CPU */
int mtg_device_count(void) { return getenv("MTG_EMU_DEVICES") ? atoi(getenv("MTG_EMU_DEVICES")) : 0; }
int mtg_set_device(int) { return MTG_OK; }
int mtg_index_replicate(const mtg_index* src, int device, mtg_index** out)
{
    /* TEST-ONLY: which device cloned to which (the tool replicates as a doubling tree), and a replication that fails */
    if (const char* lg = getenv("MTG_EMU_REPLICATE_LOG")) {
        static std::mutex log_mtx;
        std::lock_guard<std::mutex> lk(log_mtx);
        if (FILE* f = fopen(lg, "a")) { fprintf(f, "%d %d\n", src->device, device); fclose(f); }
    }
    if (const char* e = getenv("MTG_EMU_FAIL_REPLICATE"))
        if (atoi(e) == device) { mtgi::set_error("injected failure replicating to device %d", device); return MTG_ERR_NOMEM; }
    /* a deep copy: the simplest way to get one is to rebuild from the k-mers read back from the source's table */
    std::vector<uint64_t> k;
    std::vector<uint32_t> a;
    if (int rc = mtgi::index_export(src, [&](const uint64_t* kk, const uint32_t* aa, size_t m) { k.insert(k.end(), kk, kk + m); a.insert(a.end(), aa, aa + m); return true; })) return rc;
    if (int rc = mtgi::index_from_kmers(k.data(), a.data(), k.size(), src->dev.k, out)) return rc;
    (*out)->info.abundance_min = src->info.abundance_min;
    (*out)->info.abundance_auto = src->info.abundance_auto;
    (*out)->device = device;
    return MTG_OK;
}
int mtg_index_create_from_kmers(const uint64_t* k, const uint32_t* a, size_t n, int kk, mtg_index** out) { return mtgi::index_from_kmers(k, a, n, kk, out); }
void mtg_index_free(mtg_index* idx) { mtgi::index_release(idx); }
int mtg_index_get_info(const mtg_index* idx, mtg_index_info* info) { *info = idx->info; return MTG_OK; }
int mtg_index_abundance(const mtg_index* idx, const uint64_t* kmers, size_t n, uint32_t* out) { return mtgi::query_run(idx, kmers, n, out, nullptr, nullptr); }
int mtg_index_neighbors(const mtg_index* idx, const uint64_t* kmers, size_t n, uint8_t* s, uint8_t* p) { return mtgi::query_run(idx, kmers, n, nullptr, s, p); }
int mtg_last_batch_stats(mtg_batch_stats* s) { *s = mtgi::g_stats; return MTG_OK; }
int mtg_index_contains(const mtg_index* idx, const uint64_t* kmers, size_t n, uint8_t* out)
{
    std::vector<uint32_t> ab(n);
    mtgi::query_run(idx, kmers, n, ab.data(), nullptr, nullptr);
    for (size_t i = 0; i < n; i++) out[i] = ab[i] != 0;
    return MTG_OK;
}
int mtg_index_create_from_packed_device(const uint64_t*, const uint64_t*, const uint32_t*, size_t, uint64_t, int, uint32_t, uint32_t, mtg_index**)
{
    mtgi::set_error("emulation harness: no device");
    return MTG_ERR_NO_DEVICE;
}
int mtg_index_scan_packed_device(const mtg_index*, const uint64_t*, const uint64_t*, const uint32_t*, size_t, int, uint64_t*, mtg_scan_stats*)
{
    mtgi::set_error("emulation harness: no device");
    return MTG_ERR_NO_DEVICE;
}
int mtg_bench_random_lines(uint64_t, uint64_t, uint32_t, uint32_t, double*, double*) { mtgi::set_error("emulation harness: no device"); return MTG_ERR_NO_DEVICE; }
}
