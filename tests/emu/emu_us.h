/*
 * tests/emu/emu_us.h -- TEST-ONLY: the unitig store construction of the device build (kernels k_us_* of mtg_gpu.hip), run serially
 * with the same device functions (mtg_dev.h: us_plan / us_emit / us_link over the solid k-mers read back from the ABND table).
 */
#ifndef MTG_EMU_US_H
#define MTG_EMU_US_H
#include "../../mindthegap_amd/csrc/mtg_hostutil.h"
#include <cstdlib>
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

/* the sparse form of a dense index with its unitig store (the device build: sparsify in mtg_gpu.hip): new tables from the store and the
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
#endif
