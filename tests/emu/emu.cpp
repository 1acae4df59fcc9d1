/*
 * tests/emu/emu.cpp -- TEST-ONLY host emulation harness for the device code in
 * mindthegap_amd/csrc/mtg_dev.h / mtg_traverse.h (compiled by g++; every "lane" runs sequentially).
 * It lets the CPU test-suite (and CPU sanitizers) exercise the kernel logic against the oracle
 * before any GPU time is spent.  It is never built into, loaded by or reachable from the product
 * library, which requires a HIP device.
 */
#include "../../mindthegap_amd/csrc/mtg_hostutil.h"
#include "../../mindthegap_amd/csrc/mtg_copy.h"
#include "emu_us.h"
#include "emu_walk.h"
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace mtg;

struct EmuIndex {
    Index ix;
    std::vector<uint64_t> adj_slots, abnd_slots;
    EmuUStore us;
};

extern "C" {

void* emu_index_create(const uint64_t* kmers, const uint32_t* counts, size_t n, int k, double load)
{
    if (!emu_legacy_build()) {
        EmuIndex* e = new EmuIndex();
        e->ix.k = k;
        e->ix.bloom.bits = nullptr;
        e->ix.bloom.nblocks = 0;
        emu_build_lean(e->ix, e->us, kmers, counts, n);
        return e;
    }
    for (;;) {
        EmuIndex* e = new EmuIndex();
        e->ix.k = k;
        table_shape(e->ix.adj, buckets_for(n + n / 8 + 16, load, 2 * (k - 1), MTG_ADJ_SLOTS), 2 * (k - 1));
        table_shape(e->ix.abnd, buckets_for(n, load, 2 * k, MTG_ABND_SLOTS), 2 * k);
        e->adj_slots.assign(e->ix.adj.nbuckets * MTG_ADJ_SLOTS * 2, 0);
        e->abnd_slots.assign(e->ix.abnd.nbuckets * MTG_ABND_SLOTS, 0);
        e->ix.adj.slots = e->adj_slots.data();
        e->ix.abnd.slots = e->abnd_slots.data();
        e->ix.bloom.bits = nullptr;
        e->ix.bloom.nblocks = 0;
        e->ix.us = UStore{};
        e->ix.adj.sp_words = nullptr;
        int fail = 0;
        for (size_t i = 0; i < n; i++) fail |= index_insert(e->ix, kmers[i], counts[i]) & 1;
        if (!fail) {
            for (size_t i = 0; i < n; i++) { Kmer x = make_kmer(kmers[i], k); build_lookahead(e->ix, x); Kmer y; y.f = x.r; y.r = x.f; build_lookahead(e->ix, y); }
            emu_build_unitigs(e->ix, e->us);
            if (emu_sparsify(e->ix, e->us)) { e->adj_slots.clear(); e->adj_slots.shrink_to_fit(); e->abnd_slots.clear(); e->abnd_slots.shrink_to_fit(); }
            return e;
        }
        delete e;
        load *= 0.7;
    }
}
void emu_index_free(void* p) { delete (EmuIndex*)p; }
/* how many branching nodes the group form of the bubble code has answered so far (consensus found / rejected), and how many it passed on as too big */
void emu_coop_counts(unsigned long* out)
{
    const CoopTally& t = coop_tally_state();
    out[0] = t.ok; out[1] = t.fail; out[2] = 0;
    for (int i = 0; i < 10; i++) out[2] += t.big[i];
    out[3] = tip_fast_answers();
    out[4] = indel_bulk_answers();
    out[5] = merge_fast_answers();
#ifdef MTG_XCHECK
    for (int i = 0; i < 3; i++) out[6 + i] = refusal_counts()[i];
#else
    out[6] = out[7] = out[8] = 0;
#endif
}

/* stored unitigs of the lean builds so far whose two walkers met in the middle / whose owner walked the whole chain (mtg_build.h: JtWalker) */
void emu_walk_counts(unsigned long* out) { out[0] = emu_walks_met; out[1] = emu_walks_whole; }

void emu_query(void* p, const uint64_t* kmers, size_t n, uint32_t* abund, uint8_t* succ, uint8_t* pred)
{
    EmuIndex* e = (EmuIndex*)p;
    int k = e->ix.k;
    uint64_t mk1 = kmask(k - 1);
    uint32_t lines = 0;
    for (size_t i = 0; i < n; i++) {
        Kmer x = make_kmer(kmers[i], k);
        abund[i] = abundance(e->ix, x, lines);
        succ[i] = (uint8_t)adj_right(e->ix, x, mk1, lines).out;
        pred[i] = (uint8_t)adj_left(e->ix, x, mk1, lines).in;
    }
}

/* stage A for one gap; contigs joined by '\n' (malloc'd).  tier < 0: escalate tiers on overflow. */
char* emu_stage_a(void* p, int max_nodes, int max_depth, int end_rule, const char* source, const char* R, int tier, uint32_t* status,
                  uint32_t* lines, uint32_t* tier_used)
{
    EmuIndex* e = (EmuIndex*)p;
    int k = e->ix.k;
    int t0 = tier < 0 ? 0 : tier, t1 = tier < 0 ? MTG_MAX_TIER : tier;
    std::string joined;
    GapOut out{};
    for (int t = t0; t <= t1; t++) {
        FillCfg cfg = make_cfg(k, max_nodes, max_depth, end_rule, t);
        if (getenv("MTG_NO_DEFER") || !e->ix.us.nwords) cfg.cmd_cap = 0;
        std::vector<uint8_t> zero(cfg.zero_stride, 0), raw(cfg.raw_stride, 0xCD), ilv(cfg.ilv_stride), head(cfg.hd_stride, 0xCD);
        GapScratch S = carve(cfg, zero.data(), raw.data(), ilv.data(), head.data(), 0);
        std::vector<uint8_t> fp_table(FP_SLOTS * 64);
        S.fp = fp_table.data();
        S.snp_fast = getenv("MTG_NO_SNP_FAST") ? 0 : 1;
        std::vector<uint64_t> rw;
        size_t rl = strlen(R);
        pack_seq(R, rl, rw);
        SwfPattern pat;
        pat.words = rw.data();
        pat.rlen = (uint32_t)rl;
        pat.r0 = rl >= (size_t)k ? encode_kmer(R, k) : 0;
        mtg::emu_walk(e->ix, cfg, S, encode_kmer(source, k), pat, out); /* emu_walk.h: the device's launch sequence */
        copy_gap(e->ix, cfg, S, out, ~0ull); /* the device's k_copy */
        /* the device relies on every gap handing the zero region back clean: make a violation visible as a status no test expects */
        for (uint8_t z : zero) if (z) { out.status = 0xDEAD; break; }
        if (out.status == 0xDEAD) break;
        if (out.status > GAP_OVF_DFS) break; /* not an overflow (a cross-check's 0xBAD*, a guard): a larger tier must not paper over it */
        if (tier_used) *tier_used = (uint32_t)t;
        if (out.status == GAP_OK) {
            joined.clear();
            for (uint32_t i = 0; i < out.n_contigs; i++) {
                std::string s;
                unpack_seq(s_words(cfg, S) + s_cstart(cfg, S)[i], s_clen(cfg, S)[i], s);
                if (i) joined += "\n";
                joined += s;
            }
            break;
        }
    }
    if (status) *status = out.status;
    if (lines) *lines = out.lines;
    char* r = (char*)malloc(joined.size() + 1);
    memcpy(r, joined.c_str(), joined.size() + 1);
    return r;
}
void emu_free(void* p) { free(p); }
}

extern "C" void emu_debug_adj(void* p, uint64_t key)
{
    EmuIndex* e = (EmuIndex*)p;
    const Table& t = e->ix.adj;
    uint64_t H = mix(key, t.key_bits), b = bucket_of(H, t.nbuckets, t.key_bits), tag = H & ((1ULL << t.tag_bits) - 1);
    printf("key %llx H %llx nb %llu bucket %llu tag %llx tag_bits %u key_bits %u\n", (unsigned long long)key, (unsigned long long)H,
           (unsigned long long)t.nbuckets, (unsigned long long)b, (unsigned long long)tag, t.tag_bits, t.key_bits);
    for (int d = 0; d < 3; d++) {
        for (int i = 0; i < MTG_ADJ_SLOTS; i++) { uint64_t v = t.slots[((b + d) % t.nbuckets) * MTG_ADJ_SLOTS + i]; printf("  [%d,%d] tag %llx disp %llu val %llx\n", d, i, (unsigned long long)(v >> 10), (unsigned long long)((v >> 8) & 3), (unsigned long long)(v & 255)); }
    }
}

/* unit access to the device NW routine (banded, exact) for tests */
/* bucket_first_h at the sizes of a real table (the emulated builds only reach small ones): it aborts on a difference from the 128-bit quotient */
extern "C" uint64_t emu_bucket_first_sweep(uint64_t nb, uint32_t key_bits, uint64_t n, uint64_t seed)
{
    uint64_t acc = 0, s = seed;
    for (uint64_t i = 0; i < n; i++) {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        const uint64_t home = i < 4 ? (i < 2 ? i : nb - (i - 1)) : (uint64_t)(((unsigned __int128)(s >> 1) * nb) >> 63);
        acc ^= bucket_first_h(home, nb, key_bits);
    }
    return acc;
}
extern "C" int emu_nw_matches(const char* a, const char* b)
{
    FillCfg cfg = make_cfg(31, 100, 10000, 0, 0);
    std::vector<uint8_t> zero(cfg.zero_stride, 0), raw(cfg.raw_stride, 0), ilv(cfg.ilv_stride, 0), head(cfg.hd_stride, 0);
    GapScratch S = carve(cfg, zero.data(), raw.data(), ilv.data(), head.data(), 0);
    Index ix{};
    ix.k = 31;
    Worker W(ix, cfg, S);
    const int na = (int)strlen(a), nb = (int)strlen(b);
    SP<uint8_t> pa = s_cons(cfg, S), pb = s_cons(cfg, S) + CONS_LEN;
    for (int i = 0; i < na; i++) pa[i] = (uint8_t)((a[i] >> 1) & 3);
    for (int i = 0; i < nb; i++) pb[i] = (uint8_t)((b[i] >> 1) & 3);
    return nw_matches(W, pa, na, pb, nb);
}
