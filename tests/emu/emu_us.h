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
    return cr;
}
#endif
