/*
 * mtg_dev.h -- device-side k-mer model and the exact, neighbourhood-blocked index tables.
 *
 * Replaces, for the fill hot path, gatb-core's Graph neighbour / abundance queries
 * (call sites /root/reference/src/Filler.cpp:866-867,884,978; SURVEY.md 8a rows a5,a6).
 *
 * Layout (all in HBM, 64-byte buckets = one memory line per query):
 *   ADJ  table: key = canonical (k-1)-mer -> 8-bit edge mask (low nibble: nts b with key+b solid,
 *               high nibble: nts a with a+key solid).  One bucket answers the whole
 *               simplePathAvance neighbourhood of a node (its 4 successors AND the 4 predecessors
 *               of those successors share this (k-1)-mer), i.e. 8 gatb membership probes.
 *   ABND table: key = canonical k-mer -> 8-bit abundance (saturating at 255).
 * A bucket holds 8 slots of 64 bits: [tag : tag_bits][disp : 6][value : 8].  The key is hashed by a
 * bijection of its 2m-bit domain, the bucket is floor(H * nbuckets / 2^2m) and the tag the low
 * tag_bits = 2m - floor(log2 nbuckets) bits of H, which makes (bucket, tag) lossless: the tables are
 * exact (no false positives), unlike a Bloom filter + cFP cascade, for any query k-mer.
 *
 * The same source is compiled for gfx950 by hipcc and, TEST-ONLY, by g++ into the host emulation
 * harness under tests/emu (kernel-logic tests and CPU sanitizers; never loaded by the product).
 */
#ifndef MTG_DEV_H
#define MTG_DEV_H
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MTG_DEV __device__ __forceinline__
#ifndef MTG_NOINLINE_BUBBLES
#define MTG_DEV_NOINLINE __device__ __forceinline__
#else
#define MTG_DEV_NOINLINE __device__ __noinline__
#endif
#define MTG_UNROLL _Pragma("unroll")
#define MTG_LDS /* a typed LDS pointer (address_space(3), ds_* instructions) measured SLOWER than the generic one for the fingerprint table: 6.1 against 5.1 ms on the diploid set */
#define MTG_GLOBAL __attribute__((address_space(1))) /* same for the index tables: global_load instead of flat_load */
#else
#define MTG_EMU 1
#define MTG_DEV inline
#define MTG_DEV_NOINLINE inline
#define MTG_UNROLL
#define MTG_LDS
#define MTG_GLOBAL
#endif

namespace mtg {

/* A=0 C=1 T=2 G=3 ((ascii>>1)&3); complement = code ^ 2; first nt in the most significant bits. */
MTG_DEV uint64_t kmask(int k) { return (k >= 32) ? ~0ULL : ((1ULL << (2 * k)) - 1); }

MTG_DEV uint64_t revcomp(uint64_t x, int k)
{
    x ^= 0xAAAAAAAAAAAAAAAAULL;
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    x = ((x >> 8) & 0x00FF00FF00FF00FFULL) | ((x & 0x00FF00FF00FF00FFULL) << 8);
    x = ((x >> 16) & 0x0000FFFF0000FFFFULL) | ((x & 0x0000FFFF0000FFFFULL) << 16);
    x = (x >> 32) | (x << 32);
    return x >> (64 - 2 * k);
}

MTG_DEV int popc4(uint32_t m) { return (int)((0xE994u >> ((m & 7u) << 1)) & 3u) + (int)((m >> 3) & 1u); }
MTG_DEV int ctz4(uint32_t m) { return (m & 1u) ? 0 : (m & 2u) ? 1 : (m & 4u) ? 2 : 3; }
/* complement the nucleotides of a 4-bit nt mask: bit i <-> bit i^2 */
MTG_DEV uint32_t comp_mask(uint32_t m) { return ((m & 3u) << 2) | ((m >> 2) & 3u); }

/* oriented k-mer: forward value and its reverse complement, rolled together */
struct Kmer {
    uint64_t f, r;
};
MTG_DEV Kmer make_kmer(uint64_t f, int k) { Kmer x; x.f = f; x.r = revcomp(f, k); return x; }
MTG_DEV uint64_t canon(const Kmer& x) { return x.f < x.r ? x.f : x.r; }
MTG_DEV Kmer kmer_next(const Kmer& x, uint32_t nt, int k, uint64_t mk)
{
    Kmer y;
    y.f = ((x.f << 2) | (uint64_t)nt) & mk;
    y.r = (x.r >> 2) | ((uint64_t)(nt ^ 2u) << (2 * (k - 1)));
    return y;
}
/* the 16 two-bit fields of x in reverse order */
MTG_DEV uint32_t rev_fields32(uint32_t x)
{
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    return __builtin_bswap32(x);
}
/* x followed by cnt nucleotides at once (0 <= cnt <= 15, cnt <= k): `seq` holds them first-at-the-bottom (nt i at bits 2i), zero above */
MTG_DEV Kmer kmer_advance(const Kmer& x, uint32_t seq, uint32_t cnt, int k, uint64_t mk)
{
    Kmer y;
    const uint32_t fwd = cnt ? rev_fields32(seq) >> (32 - 2 * cnt) : 0u; /* first nucleotide most significant */
    const uint32_t cmp = seq ^ (0xAAAAAAAAu & ((1u << (2 * cnt)) - 1u));
    y.f = ((x.f << (2 * cnt)) | (uint64_t)fwd) & mk;
    y.r = (x.r >> (2 * cnt)) | ((uint64_t)cmp << (2 * ((uint32_t)k - cnt)));
    return y;
}
MTG_DEV Kmer kmer_prev(const Kmer& x, uint32_t nt, int k, uint64_t mk)
{
    Kmer y;
    y.f = (x.f >> 2) | ((uint64_t)nt << (2 * (k - 1)));
    y.r = ((x.r << 2) | (uint64_t)(nt ^ 2u)) & mk;
    return y;
}

/* ------------------------------------------------------------------------------------------- */
struct Table {
    uint64_t* slots;   /* nbuckets * SLOTS words */
    uint64_t nbuckets;
    uint32_t key_bits; /* 2m */
    uint32_t tag_bits; /* key_bits - floor(log2(nbuckets)), <= 50 */
};
/* slots per bucket: ADJ buckets are small because the walk pays per line touched, not per byte */
#ifndef MTG_ADJ_SLOTS
#define MTG_ADJ_SLOTS 2
#endif
#ifndef MTG_ABND_SLOTS
#define MTG_ABND_SLOTS 4
#endif
enum { MTG_MAX_DISP = 63, MTG_DISP_BITS = 6 };

/* bijection of the key_bits-wide domain */
MTG_DEV uint64_t mix(uint64_t x, uint32_t key_bits)
{
    const uint64_t M = (1ULL << key_bits) - 1;
    const uint32_t sh = key_bits >> 1; /* xorshift by >= 1 bit and multiplication by an odd constant are invertible mod 2^key_bits */
    x ^= x >> sh;
    x = (x * 0xBF58476D1CE4E5B9ULL) & M;
    x ^= x >> sh;
    return x;
}
MTG_DEV uint64_t bucket_of(uint64_t H, uint64_t nb, uint32_t key_bits)
{
#ifdef MTG_EMU
    return (uint64_t)(((unsigned __int128)H * nb) >> key_bits);
#else
    uint64_t hi = __umul64hi(H, nb), lo = H * nb;
    return (hi << (64 - key_bits)) | (lo >> key_bits);
#endif
}

struct alignas(16) U64x2 {
    uint64_t x, y;
};
/* 16-byte read of an index table: the tables live in device memory, and saying so turns the read into a global_load (a pointer that
 * came out of a struct is generic to the compiler, which then emits the slower flat_load) */
#ifdef MTG_EMU
MTG_DEV U64x2 ld_table(const U64x2* p) { return *p; }
#else
typedef unsigned long long mtg_ull2 __attribute__((ext_vector_type(2)));
MTG_DEV U64x2 ld_table(const U64x2* p)
{
    const mtg_ull2 v = *(const MTG_GLOBAL mtg_ull2*)p;
    U64x2 r;
    r.x = v.x;
    r.y = v.y;
    return r;
}
#endif

/* value of key, 0 if absent.  One bucket (SLOTS * 8 bytes) in the common case. */
template <int SLOTS> MTG_DEV uint32_t table_get(const Table& t, uint64_t key, uint32_t& lines)
{
    const uint64_t H = mix(key, t.key_bits);
    uint64_t b = bucket_of(H, t.nbuckets, t.key_bits);
    const uint64_t tag = H & ((1ULL << t.tag_bits) - 1);
    for (uint32_t d = 0; d <= MTG_MAX_DISP; d++) {
        const U64x2* p = reinterpret_cast<const U64x2*>(t.slots + b * SLOTS);
        U64x2 q[SLOTS / 2];
MTG_UNROLL
        for (int i = 0; i < SLOTS / 2; i++) q[i] = ld_table(p + i);
        lines++;
        const uint64_t want = (tag << MTG_DISP_BITS) | d;
        uint32_t val = 0; /* empty slots (all zero) may alias tag 0 / disp 0 but contribute no bits */
MTG_UNROLL
        for (int i = 0; i < SLOTS / 2; i++) {
            val |= ((q[i].x >> 8) == want) ? (uint32_t)(q[i].x & 255) : 0u;
            val |= ((q[i].y >> 8) == want) ? (uint32_t)(q[i].y & 255) : 0u;
        }
        if (val) return val;
        /* slots fill in order, so a free last slot means the key cannot be further away */
        if (q[SLOTS / 2 - 1].y == 0) return 0;
        b = (b + 1 == t.nbuckets) ? 0 : b + 1;
    }
    return 0;
}

MTG_DEV uint64_t atomic_cas64(uint64_t* p, uint64_t cmp, uint64_t val)
{
#ifdef MTG_EMU
    return __sync_val_compare_and_swap(p, cmp, val);
#else
    return (uint64_t)atomicCAS(reinterpret_cast<unsigned long long*>(p), (unsigned long long)cmp, (unsigned long long)val);
#endif
}
MTG_DEV void atomic_or64(uint64_t* p, uint64_t bits)
{
#ifdef MTG_EMU
    __sync_fetch_and_or(p, bits);
#else
    atomicOr(reinterpret_cast<unsigned long long*>(p), (unsigned long long)bits);
#endif
}

/* insert key with value bits, OR-ing into an existing entry.  Returns 0 (entry existed), 2 (entry created), or
 * 1 when the key would be displaced by more than MTG_MAX_DISP buckets (the host then rebuilds with more buckets). */
template <int SLOTS> MTG_DEV int table_or(const Table& t, uint64_t key, uint32_t bits)
{
    const uint64_t H = mix(key, t.key_bits);
    uint64_t b = bucket_of(H, t.nbuckets, t.key_bits);
    const uint64_t tag = H & ((1ULL << t.tag_bits) - 1);
    for (uint32_t d = 0; d <= MTG_MAX_DISP; d++) {
        const uint64_t want = (tag << MTG_DISP_BITS) | d;
        uint64_t* p = t.slots + b * SLOTS;
        for (int i = 0; i < SLOTS; i++) {
            uint64_t v = *(volatile uint64_t*)(p + i);
            if (v == 0) {
                v = atomic_cas64(p + i, 0, (want << 8) | bits);
                if (v == 0) return 2;
            }
            if ((v >> 8) == want) {
                if ((v & bits) != bits) atomic_or64(p + i, bits);
                return 0;
            }
        }
        b = (b + 1 == t.nbuckets) ? 0 : b + 1;
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------- */
/* ---- ADJ entries: 2 words.  w0 = [tag | disp:6 | edge mask:8] as above; w1 = two 32-bit LOOKAHEADS (low half: walking to the
 * right of the canonical (k-1)-mer, high half: walking to the right of its reverse complement).  A lookahead [count:4 | nt0:2 | nt1:2 ...]
 * lists up to 14 further nucleotides of the unique simple path that starts with this node's single out-edge: every node on it has
 * exactly one in- and one out-edge, so the walker may take those steps without touching memory (partial unitig compaction, filled by
 * build_lookahead once all k-mers are inserted).  A bucket holds MTG_ADJ_SLOTS entries. */
#ifndef MTG_LA_MAX_V
#define MTG_LA_MAX_V 14 /* nucleotides of lookahead per direction in an ADJ entry (4-bit count + 2 bits each in 32 bits) */
#endif
enum { MTG_LA_MAX = MTG_LA_MAX_V };

MTG_DEV uint32_t adj_get(const Table& t, uint64_t key, uint32_t& lines, uint64_t& aux)
{
    const uint64_t H = mix(key, t.key_bits);
    uint64_t b = bucket_of(H, t.nbuckets, t.key_bits);
    const uint64_t tag = H & ((1ULL << t.tag_bits) - 1);
    aux = 0;
    for (uint32_t d = 0; d <= MTG_MAX_DISP; d++) {
        const U64x2* p = reinterpret_cast<const U64x2*>(t.slots + b * (2 * MTG_ADJ_SLOTS));
        U64x2 q[MTG_ADJ_SLOTS];
MTG_UNROLL
        for (int i = 0; i < MTG_ADJ_SLOTS; i++) q[i] = ld_table(p + i);
        lines++;
        const uint64_t want = (tag << MTG_DISP_BITS) | d;
        uint32_t val = 0;
MTG_UNROLL
        for (int i = 0; i < MTG_ADJ_SLOTS; i++) {
            const bool hit = (q[i].x >> 8) == want && q[i].x != 0;
            val |= hit ? (uint32_t)(q[i].x & 255) : 0u;
            aux |= hit ? q[i].y : 0ull;
        }
        if (val) return val;
        if (q[MTG_ADJ_SLOTS - 1].x == 0) return 0;
        b = (b + 1 == t.nbuckets) ? 0 : b + 1;
    }
    return 0;
}
/* pointer to the entry of key (its w0), or nullptr */
MTG_DEV uint64_t* adj_find(const Table& t, uint64_t key)
{
    const uint64_t H = mix(key, t.key_bits);
    uint64_t b = bucket_of(H, t.nbuckets, t.key_bits);
    const uint64_t tag = H & ((1ULL << t.tag_bits) - 1);
    for (uint32_t d = 0; d <= MTG_MAX_DISP; d++) {
        const uint64_t want = (tag << MTG_DISP_BITS) | d;
        uint64_t* p = t.slots + b * (2 * MTG_ADJ_SLOTS);
        for (int i = 0; i < MTG_ADJ_SLOTS; i++) {
            const uint64_t v = p[2 * i];
            if (v == 0) return nullptr;
            if ((v >> 8) == want) return p + 2 * i;
        }
        b = (b + 1 == t.nbuckets) ? 0 : b + 1;
    }
    return nullptr;
}
MTG_DEV int adj_or(const Table& t, uint64_t key, uint32_t bits)
{
    const uint64_t H = mix(key, t.key_bits);
    uint64_t b = bucket_of(H, t.nbuckets, t.key_bits);
    const uint64_t tag = H & ((1ULL << t.tag_bits) - 1);
    for (uint32_t d = 0; d <= MTG_MAX_DISP; d++) {
        const uint64_t want = (tag << MTG_DISP_BITS) | d;
        uint64_t* p = t.slots + b * (2 * MTG_ADJ_SLOTS);
        for (int i = 0; i < MTG_ADJ_SLOTS; i++) {
            uint64_t v = *(volatile uint64_t*)(p + 2 * i);
            if (v == 0) {
                v = atomic_cas64(p + 2 * i, 0, (want << 8) | bits);
                if (v == 0) return 2;
            }
            if ((v >> 8) == want) {
                if ((v & bits) != bits) atomic_or64(p + 2 * i, bits);
                return 0;
            }
        }
        b = (b + 1 == t.nbuckets) ? 0 : b + 1;
    }
    return 1;
}

/* Blocked Bloom filter over the solid canonical k-mers, for membership scans along sequences (the `find`-style consumer,
 * /root/reference/src/FindBreakpoints.hpp:851-853,1012-1046; gatb's BLOOM_NEIGHBOR idea, src/Filler.cpp:189).  A block is 512 bits =
 * one 64-byte line; all NHASH bits of a k-mer fall in the block selected by the hash of the k-mer's MINIMIZER (smallest hashed
 * canonical mm-mer), so that the ~ (k-mm+2)/2 consecutive k-mers of a sequence that share a minimizer share one block. */
struct Bloom {
    uint32_t* bits;    /* nblocks * 16 words */
    uint64_t nblocks;
    int mm;            /* minimizer length */
};
enum { MTG_BLOOM_NHASH = 4 };

struct Index {
    Table adj;  /* canonical (k-1)-mer -> edge masks */
    Table abnd; /* canonical k-mer     -> abundance  */
    Bloom bloom;
    int k;
};

MTG_DEV uint64_t mix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}
/* block of the oriented k-mer x: orientation independent (canonical mm-mers) */
MTG_DEV uint64_t bloom_block(const Bloom& bl, const Kmer& x, int k)
{
    const int mm = bl.mm;
    const uint64_t mmask = kmask(mm);
    uint64_t best = ~0ULL;
    for (int i = 0; i + mm <= k; i++) {
        const uint64_t a = (x.f >> (2 * i)) & mmask;               /* mm-mer ending i nts before the end of x          */
        const uint64_t b = (x.r >> (2 * (k - mm - i))) & mmask;    /* its reverse complement, read from the rc strand */
        const uint64_t h = mix64(a < b ? a : b);
        best = h < best ? h : best;
    }
    best = mix64(best + 0x9E3779B97F4A7C15ULL); /* the minimum of several hashes is not uniform: scramble it again */
#ifdef MTG_EMU
    return (uint64_t)(((unsigned __int128)best * bl.nblocks) >> 64);
#else
    return __umul64hi(best, bl.nblocks);
#endif
}
/* bit positions inside the block: 4 x 9 bits of a hash of the canonical k-mer */
MTG_DEV uint64_t bloom_bits(uint64_t canon_kmer) { return mix64(canon_kmer ^ 0x9E3779B97F4A7C15ULL); }
MTG_DEV bool bloom_test_block(const uint32_t* blk, uint64_t hb)
{
    bool ok = true;
    MTG_UNROLL
    for (int j = 0; j < MTG_BLOOM_NHASH; j++) {
        const uint32_t bit = (uint32_t)(hb >> (9 * j)) & 511u;
        ok = ok && ((blk[bit >> 5] >> (bit & 31u)) & 1u);
    }
    return ok;
}
MTG_DEV void bloom_insert(const Bloom& bl, const Kmer& x, int k)
{
    uint32_t* blk = bl.bits + bloom_block(bl, x, k) * 16;
    const uint64_t hb = bloom_bits(canon(x));
    for (int j = 0; j < MTG_BLOOM_NHASH; j++) {
        const uint32_t bit = (uint32_t)(hb >> (9 * j)) & 511u;
#ifdef MTG_EMU
        __sync_fetch_and_or(&blk[bit >> 5], 1u << (bit & 31u));
#else
        atomicOr(&blk[bit >> 5], 1u << (bit & 31u));
#endif
    }
}
MTG_DEV bool bloom_test(const Bloom& bl, const Kmer& x, int k)
{
    return bloom_test_block(bl.bits + bloom_block(bl, x, k) * 16, bloom_bits(canon(x)));
}

struct Adj {
    uint32_t out; /* nts b such that x[1:]+b is solid  (successors of x)                     */
    uint32_t in;  /* nts a such that a+x[1:] is solid  (predecessors of every successor of x) */
    uint32_t la;  /* lookahead past the single successor, valid when out and in are single bits (adj_right only) */
};

/* right neighbourhood of x: successors of x and in-neighbours of those successors (one bucket read). */
MTG_DEV Adj adj_right_t(const Table& adj, const Kmer& x, uint64_t mk1, uint32_t& lines)
{
    const uint64_t s = x.f & mk1, rs = x.r >> 2;
    uint64_t aux;
    const uint32_t m = adj_get(adj, s <= rs ? s : rs, lines, aux);
    Adj a;
    if (s <= rs) { a.out = m & 15u; a.in = m >> 4; a.la = (uint32_t)aux; }
    else { a.out = comp_mask(m >> 4); a.in = comp_mask(m & 15u); a.la = (uint32_t)(aux >> 32); }
    return a;
}
MTG_DEV Adj adj_right(const Index& ix, const Kmer& x, uint64_t mk1, uint32_t& lines) { return adj_right_t(ix.adj, x, mk1, lines); }
/* right neighbourhoods of two unrelated nodes with both home buckets in flight together; a key that is not in its home bucket of a full
 * bucket is looked up again the ordinary way (rare) */
MTG_DEV void adj_right2(const Index& ix, const Kmer& xa, const Kmer& xb, uint64_t mk1, uint32_t& lines, Adj& ra, Adj& rb)
{
    const Table& t = ix.adj;
    const Kmer* xs[2] = {&xa, &xb};
    Adj* rs[2] = {&ra, &rb};
    U64x2 q[2][MTG_ADJ_SLOTS];
    uint64_t want[2];
    bool fw[2];
MTG_UNROLL
    for (int u = 0; u < 2; u++) {
        const uint64_t s = xs[u]->f & mk1, rsx = xs[u]->r >> 2;
        fw[u] = s <= rsx;
        const uint64_t H = mix(fw[u] ? s : rsx, t.key_bits);
        const uint64_t b = bucket_of(H, t.nbuckets, t.key_bits);
        want[u] = (H & ((1ULL << t.tag_bits) - 1)) << MTG_DISP_BITS;
        const U64x2* p = reinterpret_cast<const U64x2*>(t.slots + b * (2 * MTG_ADJ_SLOTS));
MTG_UNROLL
        for (int i = 0; i < MTG_ADJ_SLOTS; i++) q[u][i] = ld_table(p + i);
    }
MTG_UNROLL
    for (int u = 0; u < 2; u++) {
        uint32_t m = 0;
        uint64_t aux = 0;
MTG_UNROLL
        for (int i = 0; i < MTG_ADJ_SLOTS; i++) {
            const bool hit = (q[u][i].x >> 8) == want[u] && q[u][i].x != 0;
            m |= hit ? (uint32_t)(q[u][i].x & 255) : 0u;
            aux |= hit ? q[u][i].y : 0ull;
        }
        lines++;
        if (!m && q[u][MTG_ADJ_SLOTS - 1].x != 0) { *rs[u] = adj_right_t(t, *xs[u], mk1, lines); continue; } /* perhaps displaced */
        if (fw[u]) { rs[u]->out = m & 15u; rs[u]->in = m >> 4; rs[u]->la = (uint32_t)aux; }
        else { rs[u]->out = comp_mask(m >> 4); rs[u]->in = comp_mask(m & 15u); rs[u]->la = (uint32_t)(aux >> 32); }
    }
}
/* left neighbourhood of x: .in = predecessors of x, .out = successors of every predecessor. */
MTG_DEV Adj adj_left(const Index& ix, const Kmer& x, uint64_t mk1, uint32_t& lines)
{
    const uint64_t p = x.f >> 2, rp = x.r & mk1;
    uint64_t aux;
    const uint32_t m = adj_get(ix.adj, p <= rp ? p : rp, lines, aux);
    Adj a;
    a.la = 0;
    if (p <= rp) { a.out = m & 15u; a.in = m >> 4; }
    else { a.out = comp_mask(m >> 4); a.in = comp_mask(m & 15u); }
    return a;
}
MTG_DEV uint32_t abundance(const Index& ix, const Kmer& x, uint32_t& lines) { return table_get<MTG_ABND_SLOTS>(ix.abnd, canon(x), lines); }
/* an abundance look-up in two halves, so that the caller can do other work while the bucket travels */
struct AbPending {
    U64x2 q[MTG_ABND_SLOTS / 2];
    uint64_t want, key;
};
MTG_DEV void ab_issue(const Index& ix, uint64_t canon_kmer, AbPending& p)
{
    const Table& t = ix.abnd;
    const uint64_t H = mix(canon_kmer, t.key_bits);
    const uint64_t b = bucket_of(H, t.nbuckets, t.key_bits);
    p.want = (H & ((1ULL << t.tag_bits) - 1)) << MTG_DISP_BITS;
    p.key = canon_kmer;
    const U64x2* a = reinterpret_cast<const U64x2*>(t.slots + b * MTG_ABND_SLOTS);
MTG_UNROLL
    for (int i = 0; i < MTG_ABND_SLOTS / 2; i++) p.q[i] = ld_table(a + i);
}
MTG_DEV uint32_t ab_finish(const Index& ix, const AbPending& p, uint32_t& lines)
{
    uint32_t val = 0;
MTG_UNROLL
    for (int i = 0; i < MTG_ABND_SLOTS / 2; i++) {
        val |= ((p.q[i].x >> 8) == p.want) ? (uint32_t)(p.q[i].x & 255) : 0u;
        val |= ((p.q[i].y >> 8) == p.want) ? (uint32_t)(p.q[i].y & 255) : 0u;
    }
    lines++;
    /* not in its home bucket and the bucket is full: the key may have spilled further (rare) */
    if (!val && p.q[MTG_ABND_SLOTS / 2 - 1].y != 0) val = table_get<MTG_ABND_SLOTS>(ix.abnd, p.key, lines);
    return val;
}

/* index construction: one call per solid k-mer (canonical value c, abundance >= 1).
 * Returns bit 0 = displacement overflow, bit 1 = the k-mer was new. */
MTG_DEV int index_insert(const Index& ix, uint64_t c, uint32_t abund)
{
    const int k = ix.k;
    const uint64_t mk1 = kmask(k - 1);
    int fail = table_or<MTG_ABND_SLOTS>(ix.abnd, c, abund > 255u ? 255u : (abund ? abund : 1u));
    const int created = fail & 2;
    Kmer o[2];
    o[0].f = c; o[0].r = revcomp(c, k);
    o[1].f = o[0].r; o[1].r = c;
    if (ix.bloom.bits) bloom_insert(ix.bloom, o[0], k);
    for (int s = 0; s < 2; s++) {
        const Kmer& x = o[s];
        const uint32_t a = (uint32_t)(x.f >> (2 * (k - 1))) & 3u, b = (uint32_t)x.f & 3u;
        const uint64_t suf = x.f & mk1, rsuf = x.r >> 2; /* a + suf is solid */
        if (suf <= rsuf) fail |= adj_or(ix.adj, suf, 1u << (4 + a)) & 1;
        else fail |= adj_or(ix.adj, rsuf, 1u << (a ^ 2u)) & 1;
        const uint64_t pre = x.f >> 2, rpre = x.r & mk1; /* pre + b is solid */
        if (pre <= rpre) fail |= adj_or(ix.adj, pre, 1u << b) & 1;
        else fail |= adj_or(ix.adj, rpre, 1u << (4 + (b ^ 2u))) & 1;
    }
    return (fail & 1) | created;
}

/* Lookahead of the node "suffix (k-1)-mer of the solid k-mer x" in x's orientation; call for both orientations of every solid
 * k-mer AFTER all insertions.  Idempotent (every caller computes the same value and ORs it in). */
MTG_DEV void build_lookahead(const Index& ix, const Kmer& x)
{
    const int k = ix.k;
    const uint64_t mk1 = kmask(k - 1);
    uint64_t s = x.f & mk1, rs = x.r >> 2; /* the node and its reverse complement */
    uint32_t lines = 0;
    uint64_t aux;
    const bool fwd0 = s <= rs;
    const uint64_t key0 = fwd0 ? s : rs;
    uint32_t m = adj_get(ix.adj, key0, lines, aux);
    uint32_t out = fwd0 ? (m & 15u) : comp_mask(m >> 4), in = fwd0 ? (m >> 4) : comp_mask(m & 15u);
    if (!(popc4(out) == 1 && popc4(in) == 1)) return;
    uint32_t la = 0, n = 0;
    while (n < MTG_LA_MAX) {
        const uint32_t nt = (uint32_t)ctz4(out);
        s = ((s << 2) | nt) & mk1;
        rs = (rs >> 2) | ((uint64_t)(nt ^ 2u) << (2 * (k - 2)));
        const bool fw = s <= rs;
        m = adj_get(ix.adj, fw ? s : rs, lines, aux);
        out = fw ? (m & 15u) : comp_mask(m >> 4);
        in = fw ? (m >> 4) : comp_mask(m & 15u);
        if (!(popc4(out) == 1 && popc4(in) == 1)) break;
        la |= (uint32_t)ctz4(out) << (4 + 2 * n);
        n++;
    }
    if (n == 0) return;
    la |= n;
    uint64_t* e = adj_find(ix.adj, key0);
    if (e) atomic_or64(e + 1, fwd0 ? (uint64_t)la : ((uint64_t)la << 32));
}

/* ---- k-mer counting (Graph::create's DSK step, /root/reference/src/Filler.cpp:172-213): exact open-addressing count table ---- */
struct CountTable {
    uint64_t* keys;   /* ~0 = empty */
    uint32_t* counts;
    uint64_t mask;    /* capacity - 1 (power of two) */
};
/* gatb: bit 3 of the ASCII code flags a non-nucleotide ('N', and the '\n' separators of the concatenated reads) */
MTG_DEV bool ascii_invalid(unsigned char c) { return (c >> 3) & 1; }
/* canonical k-mer starting at text[i], or ~0 when the window holds an invalid character */
MTG_DEV uint64_t kmer_from_ascii(const char* text, uint64_t i, int k)
{
    uint64_t f = 0;
    bool bad = false;
    for (int j = 0; j < k; j++) {
        const unsigned char c = (unsigned char)text[i + j];
        bad = bad || ascii_invalid(c);
        f = (f << 2) | ((c >> 1) & 3u);
    }
    if (bad) return ~0ULL;
    const uint64_t r = revcomp(f, k);
    return f < r ? f : r;
}
/* returns false when the table is too full (probe limit reached) */
MTG_DEV bool count_insert(const CountTable& t, uint64_t c)
{
    uint64_t h = mix64(c) & t.mask;
    for (int probe = 0; probe < 8192; probe++) {
        uint64_t cur = *(volatile uint64_t*)(t.keys + h);
        if (cur == ~0ULL) {
            cur = atomic_cas64(t.keys + h, ~0ULL, c);
            if (cur == ~0ULL) cur = c;
        }
        if (cur == c) {
#ifdef MTG_EMU
            __sync_fetch_and_add(t.counts + h, 1u);
#else
            atomicAdd(t.counts + h, 1u);
#endif
            return true;
        }
        h = (h + 1) & t.mask;
    }
    return false;
}

} // namespace mtg
#endif
