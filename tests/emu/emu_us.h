/*
 * tests/emu/emu_us.h -- TEST-ONLY: the unitig store construction of the device build (the dense reference construction: the product built its store this way until round 4), run serially
 * with the same device functions (mtg_dev.h: us_plan / us_emit / us_link over the solid k-mers read back from the ABND table).
 */
#ifndef MTG_EMU_US_H
#define MTG_EMU_US_H
#include "../../mindthegap_amd/csrc/mtg_hostutil.h"
#include "../../mindthegap_amd/csrc/mtg_build.h"
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

struct EmuUStore {
    std::vector<uint64_t> words;
    std::vector<uint8_t> ab;
    std::vector<mtg::UsRec> recs; /* the stored unitigs */
    std::vector<uint64_t> sp_adj, sp_abnd; /* tables of the sparse form */
};

/* returns the number of stored unitigs; MTG_NO_UNITIGS=1 leaves the index with inline lookaheads only (the pre-unitig walk) */
inline uint64_t emu_build_unitigs(mtg::Index& ix, EmuUStore& st)
{
    using namespace mtg;
    ix.us.words = nullptr;
    ix.us.ab = nullptr;
    ix.us.nwords = ix.us.nunitigs = 0;
    if (getenv("MTG_NO_UNITIGS")) return 0;
    const uint64_t nslots = ix.abnd.nbuckets * MTG_ABND_SLOTS;
    uint32_t lines = 0;
    std::vector<UsRec> recs;
    unsigned long long cw = 0, cr = 0;
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) { recs.resize(cr / 2 + 1); cw = 0; cr = 0; } /* pass 0 counted the chain starts */
        for (uint64_t s = 0; s < nslots; s++) {
            uint64_t c;
            if (!abnd_slot_kmer(ix.abnd, s, c)) continue;
            Kmer x = make_kmer(c, ix.k);
            us_plan(ix, x, &cw, &cr, pass ? recs.data() : nullptr, recs.size(), lines);
            if (x.f != x.r) { Kmer y; y.f = x.r; y.r = x.f; us_plan(ix, y, &cw, &cr, pass ? recs.data() : nullptr, recs.size(), lines); }
        }
    }
    if (cr == 0) return 0;
    if (cr > recs.size()) abort(); /* the bound on the number of records does not hold */
    recs.resize(cr);
    st.words.assign(cw + 8, 0); /* padding as in the device build */
    st.ab.assign((cw + 8) * 32, 0);
    ix.us.words = st.words.data();
    ix.us.ab = st.ab.data();
    ix.us.nwords = cw;
    ix.us.nunitigs = cr;
    for (const UsRec& r : recs) us_emit(ix, r, lines);
    for (const UsRec& r : recs)
        for (uint32_t i = 0; i < r.len_k; i++) us_link(ix, r, i, lines);
    st.recs = recs;
    return cr;
}

/* the sparse form of a dense index with its unitig store (sparsify in mtg_gpu_build.hip): new tables from the store and the
 * k-mers of no stored unitig; `ix` then points to them (the caller frees the dense tables).  MTG_DENSE_INDEX=1 keeps the dense form. */
inline bool emu_sparsify(mtg::Index& ix, EmuUStore& st)
{
    using namespace mtg;
    if (!ix.us.nwords || getenv("MTG_DENSE_INDEX")) return false;
    uint32_t lines = 0;
    std::vector<uint64_t> lk;
    std::vector<uint32_t> la;
    const uint64_t nslots = ix.abnd.nbuckets * MTG_ABND_SLOTS;
    for (uint64_t s = 0; s < nslots; s++) {
        uint64_t c;
        const uint32_t ab = abnd_slot_kmer(ix.abnd, s, c);
        if (ab && !kmer_stored(ix, c, lines)) { lk.push_back(c); la.push_back(ab); }
    }
    uint64_t nkeys = 2 * lk.size() + 64;
    for (const UsRec& r : st.recs) nkeys += r.len_k / 2 + 3;
    for (double load = 0.5;; load *= 0.7) {
        Index nx = ix;
        table_shape(nx.adj, buckets_for(nkeys, load, 2 * (ix.k - 1), MTG_ADJ_SLOTS), 2 * (ix.k - 1));
        table_shape(nx.abnd, buckets_for(lk.size() + 16, load, 2 * ix.k, MTG_ABND_SLOTS), 2 * ix.k);
        st.sp_adj.assign(nx.adj.nbuckets * MTG_ADJ_SLOTS * 2, 0);
        st.sp_abnd.assign(nx.abnd.nbuckets * MTG_ABND_SLOTS, 0);
        nx.adj.slots = st.sp_adj.data();
        nx.abnd.slots = st.sp_abnd.data();
        nx.adj.sp_words = nullptr; /* while it is being built the look-ups are the raw ones */
        int fail = 0;
        for (const UsRec& r : st.recs)
            for (uint32_t i = 0; i < r.len_k; i++) fail |= sparse_link(nx, r, i, false);
        Index nb = nx;
        nb.bloom.bits = nullptr; /* the filter already holds every k-mer */
        for (size_t i = 0; i < lk.size(); i++) fail |= index_insert(nb, lk[i], la[i]) & 1;
        if (fail) continue;
        /* lookaheads of the entries that are no pointers: the junctions around the k-mers of no unitig and at the unitigs' ends */
        for (size_t i = 0; i < lk.size(); i++) { Kmer x = make_kmer(lk[i], ix.k); build_lookahead(nx, x); Kmer y; y.f = x.r; y.r = x.f; build_lookahead(nx, y); }
        for (const UsRec& r : st.recs) {
            const Kmer first = make_kmer(r.start_f, ix.k);
            Kmer fr;
            fr.f = first.r; fr.r = first.f;
            build_lookahead(nx, fr);
            build_lookahead(nx, run_node(nx.us, (r.hdr + 1) * 32, false, r.len_k - 1, ix.k));
        }
        nx.adj.sp_words = nx.us.words;
        ix = nx;
        return true;
    }
}

/* ---- the lean build (mtg_build.h; the device build_from_jt + sparsify in mtg_gpu_build.hip), run serially with the same
 * device functions: junction table + an ABND table as the abundances' source -> scan -> plan -> emit -> abundances -> sparse tables. ---- */
inline bool emu_legacy_build() { return getenv("MTG_DENSE_INDEX") || getenv("MTG_NO_UNITIGS") || getenv("MTG_LEGACY_BUILD"); }

/* the tables of the sparse form from the store and the k-mers of no unitig (lk / la); `late`, when set, names those k-mers once the unitigs'
 * pointers are in the new ADJ table (a closed or over-long chain in the graph).  The Bloom filter of ix (if any) is filled here. */
inline void emu_sparse_from_store(mtg::Index& ix, EmuUStore& st, std::vector<uint64_t>& lk, std::vector<uint32_t>& la, uint64_t n_left_ub,
                                  const std::function<void(const mtg::Index& nx, std::vector<uint64_t>&, std::vector<uint32_t>&)>* late)
{
    using namespace mtg;
    const uint64_t n_shape = late ? std::max<uint64_t>(n_left_ub, lk.size()) : lk.size();
    uint64_t nkeys = 2 * n_shape + 64;
    for (const UsRec& r : st.recs) nkeys += r.len_k / 2 + 3;
    const uint64_t bloom_words = ix.bloom.bits ? ix.bloom.nblocks * 16 : 0;
    for (double load = 0.5;; load *= 0.7) {
        Index nx = ix;
        table_shape(nx.adj, buckets_for(nkeys, load, 2 * (ix.k - 1), MTG_ADJ_SLOTS), 2 * (ix.k - 1));
        table_shape(nx.abnd, buckets_for(n_shape + 16, load, 2 * ix.k, MTG_ABND_SLOTS), 2 * ix.k);
        st.sp_adj.assign(nx.adj.nbuckets * MTG_ADJ_SLOTS * 2, 0);
        st.sp_abnd.assign(nx.abnd.nbuckets * MTG_ABND_SLOTS, 0);
        nx.adj.slots = st.sp_adj.data();
        nx.abnd.slots = st.sp_abnd.data();
        nx.adj.sp_words = nullptr;
        if (bloom_words) memset(ix.bloom.bits, 0, bloom_words * 4);
        int fail = 0;
        for (const UsRec& r : st.recs)
            for (uint32_t i = 0; i < r.len_k; i++) fail |= sparse_link(nx, r, i, true);
        if (fail) continue;
        if (late) {
            Index nxs = nx;
            nxs.adj.sp_words = nx.us.words;
            (*late)(nxs, lk, la);
            if (lk.size() > n_shape) abort(); /* the bound on the k-mers of no unitig does not hold */
        }
        for (size_t i = 0; i < lk.size(); i++) fail |= index_insert(nx, lk[i], la[i]) & 1;
        if (fail) continue;
        for (size_t i = 0; i < lk.size(); i++) { Kmer x = make_kmer(lk[i], ix.k); build_lookahead(nx, x); Kmer y; y.f = x.r; y.r = x.f; build_lookahead(nx, y); }
        for (const UsRec& r : st.recs) {
            const Kmer first = make_kmer(r.start_f, ix.k);
            Kmer fr;
            fr.f = first.r; fr.r = first.f;
            build_lookahead(nx, fr);
            build_lookahead(nx, run_node(nx.us, (r.hdr + 1) * 32, false, r.len_k - 1, ix.k));
        }
        nx.adj.sp_words = nx.us.words;
        ix = nx;
        return;
    }
}

/* TEST-ONLY tallies: stored unitigs whose two walkers met (the owner's sequence and the partner's were joined) / whose owner walked all of it */
static unsigned long long emu_walks_met = 0, emu_walks_whole = 0;
struct EmuLeanStats {
    uint64_t nb_solid = 0, nb_branching = 0, nb_unitigs = 0, nb_saturated = 0, nb_left = 0;
    bool late = false;
};
/* ix: k set, bloom shaped and allocated (or none), no tables.  Leaves ix as the device's lean build leaves its index. */
inline EmuLeanStats emu_build_lean(mtg::Index& ix, EmuUStore& st, const uint64_t* kmers, const uint32_t* counts, size_t n)
{
    using namespace mtg;
    const int k = ix.k;
    EmuLeanStats out;
    std::vector<uint64_t> jt_slots, src_slots;
    Table jt{}, abnd{};
    for (double load = 0.5;; load *= 0.7) {
        table_shape(jt, std::max<uint64_t>(buckets_for(n + n / 8 + 16, load, 2 * (k - 1), MTG_ABND_SLOTS), jt_min_buckets(2 * (k - 1))), 2 * (k - 1));
        table_shape(abnd, buckets_for(n, load, 2 * k, MTG_ABND_SLOTS), 2 * k);
        jt.sp_words = abnd.sp_words = nullptr;
        jt_slots.assign(jt.nbuckets * MTG_ABND_SLOTS, 0);
        src_slots.assign(abnd.nbuckets * MTG_ABND_SLOTS, 0);
        jt.slots = jt_slots.data();
        abnd.slots = src_slots.data();
        int fail = 0;
        out.nb_saturated = 0;
        for (size_t i = 0; i < n; i++) {
            fail |= table_or<MTG_ABND_SLOTS>(abnd, kmers[i], ab_stored(counts[i])) & 1;
            fail |= jt_insert_kmer(jt, kmers[i], k);
            out.nb_saturated += counts[i] > 255u;
        }
        if (!fail) break;
    }
    AbFromTable src;
    src.abnd = abnd;
    const uint64_t nslots = jt.nbuckets * MTG_ABND_SLOTS;
    unsigned long long counters[JT_C_N + 2] = {0};
    uint32_t lines = 0;
    /* the bucket-wise decoding of the scan against the slot-wise one it replaced (every occupied slot) */
    for (uint64_t b = 0; b < jt.nbuckets; b++) {
        const uint64_t first = bucket_first_h(b, jt.nbuckets, jt.key_bits);
        for (int i = 0; i < MTG_ABND_SLOTS; i++) {
            const uint64_t v = jt.slots[b * MTG_ABND_SLOTS + i] & ~JT_MARK; /* (a flagged junction, jt_special: the flag is not part of the entry) */
            if (!v) continue;
            uint64_t j1, j2;
            const uint32_t m1 = jt_slot_key(jt, b * MTG_ABND_SLOTS + i, j1), m2 = slot_key_in_bucket(jt, b, first, v, j2);
            if (m1 != m2 || j1 != j2) abort();
        }
    }
    /* as build_from_jt: ONE pass with lists of guessed capacity -- here a guess that is too small for all but the smallest graphs, so that the
     * second pass with the exact sizes is what most tests run -- statistics, chain starts, k-mers of no chain */
    JtAcc acc{};
    uint64_t cap_starts = 4, cap_left = 2, n_starts = 0, n_single = 0;
    std::vector<uint64_t> starts, lk;
    std::vector<uint32_t> la;
    for (int pass = 0; pass < 2; pass++) {
        starts.assign(cap_starts + 1, 0); lk.assign(cap_left + 1, 0); la.assign(cap_left + 1, 0);
        acc = JtAcc{};
        memset(counters, 0, sizeof counters);
        for (uint64_t b = 0; b < jt.nbuckets; b++) jt_scan_bucket(jt, k, b, src, acc, counters, starts.data(), cap_starts, lk.data(), la.data(), cap_left, lines);
        n_starts = counters[JT_C_STARTS]; n_single = counters[JT_C_LEFT];
        if (n_starts <= cap_starts && n_single <= cap_left) break;
        if (pass) abort();
        cap_starts = n_starts; cap_left = n_single;
    }
    lk.resize(n_single); la.resize(n_single);
    out.nb_solid = (acc.c[JT_C_ORIENTED] + acc.c[JT_C_SELF]) / 2;
    out.nb_branching = (2 * acc.c[JT_C_IN_NOT1] - acc.c[JT_C_BOTH_NOT1] + acc.c[JT_C_SELF_BRANCH]) / 2;
    const uint64_t interior = acc.c[JT_C_INTERIOR];
    counters[JT_C_SAT] = 0; /* the source holds stored (clamped) abundances: those above 255 were counted at insertion */
    std::vector<UsRec> recs(n_starts / 2 + 1);
    std::vector<uint64_t> rec_walk(recs.size(), ~0ull), w_chunk(n_starts + 1, ~0ull);
    std::vector<uint32_t> w_cnt(n_starts + 1, 0);
    WalkShared WS{};
    ChunkPool& pool = WS.pool;
    pool.cap_chunks = chunk_pool_need(interior + n_starts, n_starts, k);
    std::vector<uint64_t> pool_words(pool.cap_chunks * MTG_CHUNK_WORDS, 0xDEADDEADDEADDEADull);
    pool.words = pool_words.data();
    pool.cursor = &counters[JT_C_N];
    uint64_t mcap = 16;
    while (mcap < 2 * ((interior + n_starts) / JT_MARK_EVERY + n_starts)) mcap <<= 1;
    std::vector<uint64_t> marks(2 * mcap, ~0ull);
    WS.jt = jt; WS.k = k;
    WS.marks.keys = marks.data(); WS.marks.vals = marks.data() + mcap; WS.marks.mask = mcap - 1;
    WS.starts = starts.data(); WS.counters = counters; WS.rec = recs.data(); WS.rec_walk = rec_walk.data(); WS.rec_cap = recs.size();
    WS.w_chunk = w_chunk.data(); WS.w_cnt = w_cnt.data();
    {   /* the walkers of the graph in TURNS, an uneven number of steps each (1, 2 or 3, by the walker's number and the round), so that the two
           walkers of a chain meet in its middle, near one end, or not at all (MTG_EMU_WALK_SERIAL=1: one after the other, as a device whose
           waves never overlap: the second walker of every chain meets the first one's mark after a few steps) */
        std::vector<JtWalker> wk(n_starts);
        const bool serial = getenv("MTG_EMU_WALK_SERIAL") != nullptr;
        if (serial) {
            for (uint64_t i = 0; i < n_starts; i++) { wk[i].begin(WS, (uint32_t)i); while (wk[i].step(WS)) {} }
        } else {
            for (uint64_t i = 0; i < n_starts; i++) wk[i].begin(WS, (uint32_t)i);
            uint64_t live = n_starts, round = 0;
            while (live) {
                live = 0;
                for (uint64_t i = 0; i < n_starts; i++) {
                    if (wk[i].done) continue;
                    const uint32_t turns = 1 + (uint32_t)((i * 2654435761ull + round * 40503ull) >> 7) % 3;
                    bool on = true;
                    for (uint32_t t = 0; t < turns && on; t++) on = wk[i].step(WS);
                    live += on;
                }
                round++;
            }
        }
    }
    const uint64_t n_rec = counters[JT_C_RECS], cw = counters[JT_C_WORDS];
    if (n_rec > recs.size() || counters[JT_C_N] > pool.cap_chunks) abort();
    recs.resize(n_rec);
    /* every chain is stored once: the two walkers of a chain agree on its length and on who owns it */
    {
        uint64_t met = 0;
        for (uint64_t u = 0; u < n_rec; u++) met += (uint32_t)(rec_walk[u] >> 32) != 0xFFFFFFFFu;
        emu_walks_met += met; emu_walks_whole += n_rec - met;
    }
    ix.us = UStore{};
    if (n_rec) {
        st.words.assign(cw + 8, 0);
        st.ab.assign((cw + 8) * 32, 0);
        ix.us.words = st.words.data();
        ix.us.ab = st.ab.data();
        ix.us.nwords = cw;
        ix.us.nunitigs = n_rec;
        for (uint64_t u = 0; u < n_rec; u++) {
            const UsRec& r = recs[u];
            if (!us_compact(ix.us, k, r, rec_walk[u], WS, 0, 1)) abort();
            /* the words the chunks of the two walkers brought against ONE walk of the whole chain (what round 4's separate emit pass wrote) */
            std::vector<uint32_t> nts;
            for (int i = k - 1; i >= 0; i--) nts.push_back((uint32_t)(r.start_f >> (2 * i)) & 3u);
            Kmer end;
            if (jt_walk(jt, k, make_kmer(r.start_f, k), end, lines, [&](uint32_t c) { nts.push_back(c); }) != r.len_k) abort();
            if (ix.us.words[r.hdr] != (uint64_t)r.len_k + (uint32_t)k - 1 || nts.size() != (size_t)r.len_k + (size_t)k - 1) abort();
            for (size_t i = 0; i < nts.size(); i += 32) {
                uint64_t w = 0;
                for (size_t j = i; j < nts.size() && j < i + 32; j++) w |= (uint64_t)nts[j] << (2 * (j - i));
                if (ix.us.words[r.hdr + 1 + i / 32] != w) abort();
            }
        }
        for (const UsRec& r : recs)
            for (uint32_t i = 0; i < r.len_k; i++) us_ab_fill(ix.us, k, r, i, src, lines);
    }
    st.recs = recs;
    out.nb_unitigs = n_rec;
    ix.adj = Table{};
    ix.abnd = Table{};
    if (interior == counters[JT_C_STORED_VIEWS]) emu_sparse_from_store(ix, st, lk, la, lk.size(), nullptr);
    else {
        out.late = true;
        const std::function<void(const Index&, std::vector<uint64_t>&, std::vector<uint32_t>&)> late = [&](const Index& nx, std::vector<uint64_t>& k2, std::vector<uint32_t>& a2) {
            counters[JT_C_LEFT] = 0;
            for (uint64_t s = 0; s < nslots; s++) {
                uint64_t J;
                const uint32_t m = jt_slot_key(jt, s, J);
                if (m) jt_unstored_entry(jt, nx, J, m, src, counters, (uint64_t*)nullptr, (uint32_t*)nullptr, 0ull, lines);
            }
            const uint64_t nl = counters[JT_C_LEFT];
            k2.assign(nl, 0);
            a2.assign(nl, 0);
            counters[JT_C_LEFT] = 0;
            for (uint64_t s = 0; s < nslots; s++) {
                uint64_t J;
                const uint32_t m = jt_slot_key(jt, s, J);
                if (m) jt_unstored_entry(jt, nx, J, m, src, counters, k2.data(), a2.data(), nl, lines);
            }
        };
        lk.clear(); la.clear();
        emu_sparse_from_store(ix, st, lk, la, n_single + (interior - counters[JT_C_STORED_VIEWS]) / 2 + n_starts + 16, &late);
    }
    out.nb_left = lk.size();
    return out;
}
#endif
