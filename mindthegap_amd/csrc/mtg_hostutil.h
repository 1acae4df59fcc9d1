/*
 * mtg_hostutil.h -- host-side helpers shared by the HIP library and the test-only emulation
 * harness: table sizing, default scratch capacities, ASCII <-> 2-bit conversions.
 */
#ifndef MTG_HOSTUTIL_H
#define MTG_HOSTUTIL_H
#include "mtg_traverse.h"
#include <string>
#include <vector>

namespace mtg {

inline uint32_t floor_log2_u64(uint64_t x) { uint32_t l = 0; while (x >>= 1) l++; return l; }

/* number of buckets for n keys at the given load factor */
inline uint64_t buckets_for(uint64_t nkeys, double load, uint32_t key_bits, int slots)
{
    uint64_t nb = (uint64_t)((double)nkeys / ((double)slots * load)) + 1;
    uint64_t min_nb = key_bits > 50 ? (1ULL << (key_bits - 50)) : 1;
    if (nb < min_nb) nb = min_nb;
    if (nb < 64) nb = 64;
    /* (bucket, tag) must stay lossless: nbuckets <= 2^key_bits */
    if (key_bits < 62 && nb > (1ULL << key_bits)) nb = 1ULL << key_bits;
    return nb;
}
inline void table_shape(Table& t, uint64_t nbuckets, uint32_t key_bits)
{
    t.nbuckets = nbuckets;
    t.key_bits = key_bits;
    uint32_t lg = floor_log2_u64(nbuckets);
    t.tag_bits = key_bits > lg ? key_bits - lg : 0;
}

/* Bloom shape for n k-mers */
inline void bloom_shape(Bloom& b, uint64_t nkeys, double bits_per_kmer, int k)
{
    b.nblocks = (uint64_t)((double)nkeys * bits_per_kmer / 512.0) + 16;
    b.mm = k >= 25 ? 17 : (k - 8 > 5 ? k - 8 : 5);
    b.bits = nullptr;
}

/* scratch tiers: tier 0 fits the common case, each further tier multiplies the growable capacities by 8 */
inline FillCfg make_cfg(int k, int max_nodes, int max_depth, int end_rule_nonbranching, int tier)
{
    FillCfg c;
    c.k = k;
    c.max_nodes = max_nodes;
    c.max_depth = max_depth;
    c.mono_max_depth = 500;
    c.mono_max_breadth = 20;
    c.end_rule_nonbranching = end_rule_nonbranching;
    uint32_t mul = 1;
    for (int i = 0; i < tier; i++) mul *= 8;
    c.cap_words = 2048u * mul;
    c.cap_contigs = (uint32_t)max_nodes + 1;
    c.qcap = 4 * c.cap_contigs + 2;
    c.mcap = 1024u * mul;
    c.seen_cap = 2048u * mul;
    c.inv_cap = 2048u * mul;
    c.iseen_cap = 2048u * (mul > 8 ? 8 : mul);
    c.cmd_cap = COPY_CMDS;
    finalize_cfg(c);
    return c;
}
enum { MTG_MAX_TIER = 3 };

inline uint32_t nt_code(unsigned char ch) { return (ch >> 1) & 3u; }
inline uint64_t encode_kmer(const char* s, int k)
{
    uint64_t x = 0;
    for (int i = 0; i < k; i++) x = (x << 2) | nt_code((unsigned char)s[i]);
    return x;
}
inline void pack_seq(const char* s, size_t n, std::vector<uint64_t>& out)
{
    out.assign((n + 31) / 32 + 1, 0);
    for (size_t i = 0; i < n; i++) out[i >> 5] |= (uint64_t)nt_code((unsigned char)s[i]) << (2 * (i & 31));
}
inline void unpack_seq(const uint64_t* w, uint32_t n, std::string& out)
{
    static const char NT[4] = {'A', 'C', 'T', 'G'};
    out.resize(n);
    for (uint32_t i = 0; i < n; i++) out[i] = NT[(w[i >> 5] >> (2 * (i & 31))) & 3];
}

} // namespace mtg
#endif
