/*
 * tests/emu/emu_backend.cpp -- TEST-ONLY stand-in for mtg_gpu.hip: the same device functions
 * (mtg_dev.h / mtg_traverse.h) executed lane by lane on the host, behind the internal interface of
 * mtg_internal.h.  Linked with the product's host translation units (mtg_host.cpp, mtg_cli.cpp) into
 * tests/emu/libmtgfill_emu.so so that the CPU test-suite can run the whole `MindTheGap fill` logic
 * (and sanitizers) without a GPU.  The product library never contains this file.
 */
#include "../../mindthegap_amd/csrc/mtg_internal.h"
#include "emu_us.h"
#include <map>
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstring>

using namespace mtg;

namespace mtgi {
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
        for (size_t i = 0; i < n; i++) { int r = index_insert(idx->dev, kmers[i], ab[i]); fail |= r & 1; created += (r >> 1) & 1; }
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
                emu_build_unitigs(idx->dev, *st);
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
void index_release(mtg_index* idx)
{
    if (!idx) return;
    index_forget_host_copy(idx);
    { std::lock_guard<std::mutex> lk(g_us_mtx); auto it = g_us.find(idx); if (it != g_us.end()) { delete it->second; g_us.erase(it); } }
    free(idx->dev.adj.slots); free(idx->dev.abnd.slots); free(idx->dev.bloom.bits);
    for (Workspace& w : idx->ws) for (void* h : w.hptr) free(h);
    delete idx;
}

int query_run(const mtg_index* idx, const uint64_t* kmers, size_t n, uint32_t* abund, uint8_t* succ, uint8_t* pred)
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

int count_run(const char* text, size_t n, int k, uint32_t keep_min, std::vector<uint64_t>& histo, std::vector<uint64_t>& kmers, std::vector<uint32_t>& counts)
{
    kmers.clear();
    counts.clear();
    if (n < (size_t)k) return MTG_OK;
    uint64_t cap = 1ull << 10;
    while (cap < n / 4) cap <<= 1;
    for (;; cap <<= 1) {
        std::vector<uint64_t> keys(cap, ~0ULL);
        std::vector<uint32_t> cnts(cap, 0);
        CountTable t{keys.data(), cnts.data(), cap - 1};
        bool full = false;
        for (size_t i = 0; i + k <= n && !full; i++) {
            const uint64_t c = kmer_from_ascii(text, i, k);
            if (c != ~0ULL && !count_insert(t, c)) full = true;
        }
        if (full) continue;
        for (uint64_t i = 0; i < cap; i++) {
            if (keys[i] == ~0ULL) continue;
            histo[std::min<size_t>(cnts[i], histo.size() - 1)]++;
            if (cnts[i] >= keep_min) { kmers.push_back(keys[i]); counts.push_back(cnts[i]); }
        }
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
int nw_run(const mtg_index*, const std::vector<NwPair>& pairs, std::vector<uint32_t>& matches)
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
    if (!ws || slot < 0 || slot >= Workspace::NHOST || slot >= STAGING_CHUNK0) return nullptr; /* results keep their own storage here */
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

int device_run(const mtg_index* idx, const mtg_params* p, const FillInput& in, DevBatch& batch, mtg_batch_stats* stats,
               const std::function<void()>* while_busy, const std::function<void(size_t, const uint32_t*, size_t, size_t)>* on_ready)
{
    if (while_busy) (*while_busy)();
    /* stand-in for k_encode_targets */
    const size_t n_targets = in.traw.size() / TARGET_SLOT;
    std::vector<uint64_t> tle(n_targets), tbad(n_targets);
    for (size_t t = 0; t < n_targets; t++) encode_target(in.traw.data() + t * TARGET_SLOT, idx->dev.k, tle[t], tbad[t]);
    const size_t n = in.src.size();
    batch.n = n;
    batch.chunks.clear();
    mtg_batch_stats st = stats ? *stats : mtg_batch_stats{};
    batch.part = ~(size_t)0;
    batch.chunk_of.clear();
    batch.slot_of.clear();
    /* Same shape as the device build: a launch covers the gaps still to do at a scratch tier; its results come back as parts of consecutive
     * slots, each a chunk with dense word / metadata arrays in which the gaps reserved their room in no particular order (here: last slot
     * first); the first launch maps gap i to chunk i / part, slot i % part, later ones through chunk_of / slot_of. */
    const uint32_t psize = getenv("MTG_EMU_PART") ? (uint32_t)std::max(64, atoi(getenv("MTG_EMU_PART")) / 64 * 64) : 192u;
    struct SlotData { GapOut o; PostOut po; uint32_t nw = 0, nc = 0; std::vector<uint64_t> words; std::vector<uint32_t> meta; std::vector<uint32_t> paths; bool general = false; };
    std::vector<uint32_t> todo;
    size_t n_todo = n;
    for (int tier = 0; tier <= MTG_MAX_TIER && n_todo; tier++) {
        const bool identity = todo.empty();
        FillCfg cfg = make_cfg(idx->dev.k, p->max_nodes, p->max_depth, p->end_rule_nonbranching, tier);
        std::vector<uint32_t> retry;
        const uint32_t m = (uint32_t)n_todo;
        if (identity) batch.part = psize;
        else if (batch.chunk_of.empty()) {
            batch.chunk_of.assign(n, 0);
            batch.slot_of.resize(n);
            for (size_t i = 0; i < n; i++) { batch.chunk_of[i] = batch.part == ~(size_t)0 ? 0u : (uint32_t)(i / batch.part); batch.slot_of[i] = batch.part == ~(size_t)0 ? (uint32_t)i : (uint32_t)(i % batch.part); }
        }
        /* one scratch block for the launch, like a lane's on the device: the zero region is cleared once and must come back clean from
         * every gap, the rest holds whatever the previous gap left (poisoned here) */
        std::vector<uint8_t> zero(cfg.zero_stride, 0), raw(cfg.raw_stride + 64), ilv(cfg.ilv_stride), fp_table(FP_SLOTS * 64);
        for (uint32_t s0 = 0; s0 < m; s0 += psize) {
            const uint32_t s1 = std::min(m, s0 + psize), mq = s1 - s0;
            std::vector<SlotData> sd(mq);
            for (uint32_t s = 0; s < mq; s++) {
                const size_t g = identity ? (size_t)(s0 + s) : (size_t)todo[s0 + s];
                SlotData& d = sd[s];
                memset(raw.data(), 0xCD, raw.size());
                memset(fp_table.data(), 0, fp_table.size());
                GapScratch S = carve(cfg, zero.data(), raw.data(), ilv.data(), 0);
                S.fp = fp_table.data();
                S.snp_fast = getenv("MTG_NO_SNP_FAST") ? 0 : 1;
                SwfPattern R;
                R.words = in.rwords.data() + in.roff[g];
                R.rlen = in.rlen[g];
                R.r0 = in.r0[g];
                stage_a_gap(idx->dev, cfg, S, in.src[g], R, d.o);
                /* the device never clears the zero region between launches: every exit path has to hand it back clean */
                for (uint8_t z : zero) if (z) { set_error("gap %zu: zero-initialised scratch not restored (status %u)", g, d.o.status); return MTG_ERR_OVERFLOW; }
                st.index_lines += d.o.lines;
                if (d.o.status != GAP_OK) { st.n_retried_gaps++; retry.push_back((uint32_t)g); continue; }
                PostTargets T;
                T.le = tle.data() + in.toff[g];
                T.bad = tbad.data() + in.toff[g];
                T.n = in.tcnt[g];
                T.nb_mis = in.nbmis[g];
                T.fast_ok = in.fast_ok[g];
                uint32_t hist[256] = {0};
                std::vector<uint64_t> tile(POST_TILE + 2);
                uint64_t blk[64];
                post_gap(idx->dev, cfg, S, d.o, T, hist, tile.data(), blk, d.po);
                copy_plan(d.o, d.po, in.want_all_contigs, d.nw, d.nc);
                d.words.assign(s_words(cfg, S), s_words(cfg, S) + d.nw);
                d.meta.resize(5 * (size_t)d.nc);
                for (uint32_t i = 0; i < d.nc; i++) {
                    d.meta[i] = s_clen(cfg, S)[i];
                    d.meta[d.nc + i] = s_cstart(cfg, S)[i];
                    d.meta[2 * d.nc + i] = s_tpos(cfg, S)[i];
                    d.meta[3 * d.nc + i] = s_terr(cfg, S)[i];
                    d.meta[4 * d.nc + i] = s_ttgt(cfg, S)[i];
                }
                if (!getenv("MTG_HOST_PATHS") && d.po.fast == 0 && d.po.nb_terminal > 0) { /* stand-in for k_paths */
                    PathsWork pw;
                    d.paths.assign(PATHS_WORDS, 0);
                    paths_gap(cfg, S, d.o, idx->dev.k, pw, d.paths.data());
                    d.general = true;
                }
                st.contig_nt += d.o.total_nt;
                st.store_runs += d.o.store_reads; st.run_nt += d.o.run_nt; st.post_lines += d.po.lines; st.contig_words += d.o.n_words; st.coverage_kmers += d.po.ab_n;
            }
            uint64_t tw = 0, tc = 0;
            for (const SlotData& d : sd) { tw += d.nw; tc += d.nc; }
            batch.chunks.emplace_back(new HostChunk());
            HostChunk& hc = *batch.chunks.back();
            const uint32_t chunk_id = (uint32_t)batch.chunks.size() - 1;
            SlotRec* rec = nullptr;
            uint64_t* hw = nullptr;
            uint32_t* hm = nullptr;
            hc.carve(nullptr, mq, tw, tc, rec, hw, hm);
            uint64_t wb = 0, cb = 0;
            for (uint32_t s = mq; s-- > 0;) { /* room in the dense arrays in reverse slot order */
                const SlotData& d = sd[s];
                SlotRec& r = rec[s];
                r.o = d.o; r.p = d.po; r.nw = d.nw; r.nc = d.nc; r.wbase = (decltype(r.wbase))wb; r.cbase = (decltype(r.cbase))cb;
                if (d.o.status != GAP_OK) { r.nw = r.nc = 0; continue; }
                for (uint32_t i = 0; i < d.nw; i++) hw[wb + i] = d.words[i];
                for (size_t i = 0; i < d.meta.size(); i++) hm[5 * cb + i] = d.meta[i];
                wb += d.nw;
                cb += d.nc;
            }
            hw[tw] = 0;
            uint32_t ngen = 0;
            for (const SlotData& d : sd) ngen += d.general;
            if (ngen) {
                hc.paths.resize((size_t)ngen * PATHS_WORDS);
                hc.path_of.assign(mq, -1);
                uint32_t r2 = 0;
                for (uint32_t s = 0; s < mq; s++)
                    if (sd[s].general) { std::copy(sd[s].paths.begin(), sd[s].paths.end(), hc.paths.begin() + (size_t)r2 * PATHS_WORDS); hc.path_of[s] = (int32_t)r2++; }
            }
            if (!identity)
                for (uint32_t s = 0; s < mq; s++)
                    if (sd[s].o.status == GAP_OK) { batch.chunk_of[todo[s0 + s]] = chunk_id; batch.slot_of[todo[s0 + s]] = s; }
            if (on_ready) (*on_ready)(chunk_id, identity ? nullptr : todo.data() + s0, identity ? (size_t)s0 : 0, mq);
        }
        todo.swap(retry);
        n_todo = todo.size();
    }
    if (n_todo) { set_error("%zu gap(s) exceeded the largest traversal scratch tier", n_todo); return MTG_ERR_OVERFLOW; }
    if (stats) *stats = st;
    return MTG_OK;
}
} // namespace mtgi

extern "C" {
const char* mtg_last_error(void) { return mtgi::g_err; }
int mtg_device_count(void) { return 0; }
int mtg_set_device(int) { return MTG_OK; }
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
