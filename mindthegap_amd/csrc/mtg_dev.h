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
#define MTG_HD __host__ __device__ __forceinline__ /* small helpers the host side of the mtg_gpu_*.hip units shares with the kernels */
#ifndef MTG_NOINLINE_BUBBLES
#define MTG_DEV_NOINLINE __device__ __forceinline__
#else
#define MTG_DEV_NOINLINE __device__ __noinline__
#endif
#define MTG_DEV_COLD __device__ __noinline__ /* a rarely taken, large piece (the sparse index's second look-up, the alignment): a real call, so that its registers are not part of every kernel that might take it */
#define MTG_UNROLL _Pragma("unroll")
#define MTG_LDS /* a typed LDS pointer (address_space(3), ds_* instructions) measured SLOWER than the generic one for the fingerprint table: 6.1 against 5.1 ms on the diploid set */
#define MTG_GLOBAL __attribute__((address_space(1))) /* same for the index tables: global_load instead of flat_load */
#else
#define MTG_EMU 1
#ifndef MTG_NO_XCHECK
#define MTG_XCHECK 1 /* TEST-ONLY cross-checks of the emulation build: every shortcut of the device code next to the form it replaces (statuses 0xBAD*).
                        -DMTG_NO_XCHECK leaves them out: the device code alone on the host, for bench.py's same-algorithm CPU number */
#endif
#define MTG_DEV inline
#define MTG_HD inline
#define MTG_DEV_NOINLINE inline
#define MTG_DEV_COLD inline
#define MTG_UNROLL
#define MTG_LDS
#define MTG_GLOBAL
#endif

namespace mtg {

/* A=0 C=1 T=2 G=3 ((ascii>>1)&3); complement = code ^ 2; first nt in the most significant bits. */
MTG_HD uint64_t kmask(int k) { return (k >= 32) ? ~0ULL : ((1ULL << (2 * k)) - 1); }

MTG_HD uint64_t revcomp(uint64_t x, int k)
{
#if defined(__HIP_DEVICE_COMPILE__)
    /* the device reverses the bits of a word in one instruction: the two halves reversed and exchanged, then the two bits of every field back in order
     * (a dozen operations instead of the fifty of the five exchanges below; the index construction computes one of these per junction position) */
    uint32_t lo = __builtin_bitreverse32((uint32_t)(x >> 32) ^ 0xAAAAAAAAu), hi = __builtin_bitreverse32((uint32_t)x ^ 0xAAAAAAAAu);
    lo = ((lo >> 1) & 0x55555555u) | ((lo & 0x55555555u) << 1);
    hi = ((hi >> 1) & 0x55555555u) | ((hi & 0x55555555u) << 1);
    return (((uint64_t)hi << 32) | lo) >> (64 - 2 * k);
#endif
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
    /* ADJ only, non-null on a SPARSE index (see "sparse index" below): the words of the unitig store, through which the neighbourhood of a
     * junction that has no entry of its own is read */
    const uint64_t* sp_words;
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
MTG_DEV unsigned long long atomic_add64(unsigned long long* p, unsigned long long v)
{
#ifdef MTG_EMU
    return __sync_fetch_and_add(p, v);
#else
    return atomicAdd(p, v);
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

/* the entry of a key that is inserted ONCE (an interior junction of a stored unitig, by the one lane that has it): bits and word 1 in one pass
 * over the bucket -- the bucket read in one go, a compare-and-swap on the first free slot, the store of word 1 -- where adj_or + adj_find + a
 * store walked the bucket three times word by word.  (A blind compare-and-swap on each slot in turn, without the read, was measured: 223
 * against 122 ms for the human-scale table -- an atomic on an occupied slot costs more than the read that avoids it.)  Returns as adj_or. */
MTG_DEV int adj_set_new(const Table& t, uint64_t key, uint32_t bits, uint64_t w1)
{
    const uint64_t H = mix(key, t.key_bits);
    uint64_t b = bucket_of(H, t.nbuckets, t.key_bits);
    const uint64_t tag = H & ((1ULL << t.tag_bits) - 1);
    for (uint32_t d = 0; d <= MTG_MAX_DISP; d++) {
        const uint64_t want = (tag << MTG_DISP_BITS) | d;
        uint64_t* p = t.slots + b * (2 * MTG_ADJ_SLOTS);
        U64x2 q[MTG_ADJ_SLOTS];
MTG_UNROLL
        for (int i = 0; i < MTG_ADJ_SLOTS; i++) q[i] = ld_table(reinterpret_cast<const U64x2*>(p) + i);
        for (int i = 0; i < MTG_ADJ_SLOTS; i++) {
            uint64_t v = q[i].x; /* a stale read is harmless: a slot seen free is claimed by compare-and-swap, which says what is there now */
            if (v == 0) {
                v = atomic_cas64(p + 2 * i, 0, (want << 8) | bits);
                if (v == 0) { p[2 * i + 1] = w1; return 2; }
            }
            if ((v >> 8) == want) {
                if ((v & bits) != bits) atomic_or64(p + 2 * i, bits);
                p[2 * i + 1] = w1;
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

/* Unitig store: the maximal simple paths of the graph (at least two k-mers each), 2-bit packed one after the other.  Unitig u occupies
 * words[hdr] (header: its length in nucleotides) and the ceil(len / 32) words behind it, nucleotide i at bits 2 (i mod 32) of word
 * hdr + 1 + i / 32; ab[(hdr + 1) * 32 + i] = abundance of the k-mer starting at nucleotide i (0 for the last k-1 nucleotides and the
 * padding).  Every junction ((k-1)-mer between two consecutive k-mers) inside a unitig carries, in word 1 of its ADJ entry, a POINTER
 * to its place in the store instead of an inline lookahead: a walker that reads such an entry knows the whole rest of the simple path
 * and follows it with sequential reads (full unitig compaction; see us_* below for the construction). */
struct UStore {
    uint64_t* words;
    uint8_t* ab;
    uint64_t nwords;   /* words in use (0: no store) */
    uint64_t nunitigs;
};

struct Index {
    Table adj;  /* canonical (k-1)-mer -> edge masks */
    Table abnd; /* canonical k-mer     -> abundance  */
    Bloom bloom;
    UStore us;
    int k;
};

MTG_DEV uint64_t mix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}
/* the block of a k-mer whose smallest hashed canonical mm-mer hashes to `best` */
MTG_DEV uint64_t bloom_block_of_min(const Bloom& bl, uint64_t best)
{
    best = mix64(best + 0x9E3779B97F4A7C15ULL); /* the minimum of several hashes is not uniform: scramble it again */
#ifdef MTG_EMU
    return (uint64_t)(((unsigned __int128)best * bl.nblocks) >> 64);
#else
    return __umul64hi(best, bl.nblocks);
#endif
}
/* hash of the canonical mm-mer whose little-endian image (first nucleotide lowest, as the unitig store holds it) is le: what bloom_block
 * takes the minimum of */
MTG_DEV uint64_t bloom_mmer_hash(uint64_t le, int mm)
{
    const uint64_t r = le ^ (0xAAAAAAAAAAAAAAAAULL & kmask(mm)), f = revcomp(r, mm);
    return mix64(f < r ? f : r);
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
    return bloom_block_of_min(bl, best);
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
    uint64_t up;  /* 0, or the junction's place in the unitig store, resolved for the walking direction (up_* below; adj_right_t only) */
};

/* ---- word 1 of an ADJ entry, pointer form: [hdr : 32][off : 27][rc : 1][1111].  The low nibble 15 cannot be a lookahead count
 * (MTG_LA_MAX <= 14).  hdr = header word of the unitig, off = nucleotide offset of the junction's first nucleotide in the stored
 * unitig (1 <= off <= len - k), rc = the canonical (k-1)-mer appears reverse-complemented in the stored sequence.
 * Resolved form (Adj::up): bit 0 = valid, bit 1 = the walk runs against the stored orientation, other fields unchanged. */
enum : uint32_t { MTG_US_MAX_LEN = (1u << 27) - 1 };
MTG_DEV bool up_is(uint64_t w1) { return (w1 & 15ull) == 15ull; }
MTG_DEV uint64_t up_make(uint64_t hdr, uint32_t off, bool rc) { return 15ull | ((uint64_t)(rc ? 1 : 0) << 4) | ((uint64_t)off << 5) | (hdr << 32); }
MTG_DEV uint64_t up_resolve(uint64_t w1, bool key_is_walk_orientation)
{
    const bool rc = (w1 >> 4) & 1;
    const bool fwd = key_is_walk_orientation != rc; /* the (k-1)-mer as walked appears as it is in the store */
    return (w1 & ~31ull) | 1ull | (fwd ? 0ull : 2ull);
}
MTG_DEV uint64_t up_hdr(uint64_t up) { return up >> 32; }
MTG_DEV uint32_t up_off(uint64_t up) { return (uint32_t)(up >> 5) & MTG_US_MAX_LEN; }
MTG_DEV bool up_bwd(uint64_t up) { return (up & 2ull) != 0; }

/* ---- sparse index.  Inside a stored unitig every junction is simple: the k-mer before it has one successor, the one after it one
 * predecessor, and both are written in the store.  A SPARSE index keeps the ADJ entry of an interior junction (offset off, 1 <= off <= number
 * of k-mers - 1) only when off is odd or the last one: every k-mer of a stored unitig still has an entry next to it (its left or its right
 * junction), whose pointer says where the k-mer sits, and the store says the rest.  The ABND table of a sparse index holds only the k-mers
 * that are in no stored unitig (the abundances of the others are the bytes next to the store).  At human scale: ADJ 112 GB -> about a
 * third, ABND 41 GB -> next to nothing.
 *
 * The look-ups below are exact on both forms.  A key without an entry is either a junction that is not in the graph, or one that was left
 * out: then the OTHER junction of the query k-mer has an entry with a pointer, the k-mer is checked against the store there (a k-mer that
 * is not solid never passes: a simple junction has one k-mer on each side), and the neighbourhood is read off the store.  The junction
 * left out is interior, so the k-mer is not the unitig's last (first) one and the neighbour exists. */
MTG_DEV Adj adj_right_raw(const Table& adj, const Kmer& x, uint64_t mk1, uint32_t& lines, uint32_t& mask_found)
{
    const uint64_t s = x.f & mk1, rs = x.r >> 2;
    uint64_t aux;
    const uint32_t m = adj_get(adj, s <= rs ? s : rs, lines, aux);
    mask_found = m;
    Adj a;
    a.up = 0;
    if (s <= rs) { a.out = m & 15u; a.in = m >> 4; a.la = (uint32_t)aux; }
    else { a.out = comp_mask(m >> 4); a.in = comp_mask(m & 15u); a.la = (uint32_t)(aux >> 32); }
    if (up_is(aux)) { a.la = 0; a.up = up_resolve(aux, s <= rs); }
    return a;
}

/* ---- reading the unitig store ---------------------------------------------------------------------------------------------------
 * n <= 16 nucleotides of the store in walking order, first one in the lowest bits: forward = the nucleotides at pos, pos + 1, ...;
 * backward = the complements of those at pos, pos - 1, ... (global nucleotide index = word * 32 + i) */
MTG_DEV uint32_t us_peek(const uint64_t* words, uint64_t pos, uint32_t n, bool bwd)
{
    if (n == 0) return 0u;
    const uint64_t lo = bwd ? pos - (n - 1) : pos;
    const uint32_t sh = 2u * (uint32_t)(lo & 31u);
    uint64_t v = words[lo >> 5] >> sh;
    if ((uint32_t)(lo & 31u) + n > 32u) v |= words[(lo >> 5) + 1] << (64u - sh); /* sh > 0 here */
    const uint32_t mask = n >= 16u ? 0xFFFFFFFFu : ((1u << (2u * n)) - 1u);
    uint32_t r = (uint32_t)v & mask;
    if (bwd) r = (rev_fields32(r) >> (32u - 2u * n)) ^ (0xAAAAAAAAu & mask);
    return r;
}
/* the 32 two-bit fields of x in reverse order */
MTG_DEV uint64_t rev_fields64(uint64_t x) { return ((uint64_t)rev_fields32((uint32_t)x) << 32) | (uint64_t)rev_fields32((uint32_t)(x >> 32)); }
/* the same for up to 32 nucleotides, as a 64-bit word */
MTG_DEV uint64_t us_peek64(const uint64_t* words, uint64_t pos, uint32_t n, bool bwd)
{
    if (n == 0) return 0ull;
    const uint64_t lo = bwd ? pos - (n - 1) : pos;
    const uint32_t sh = 2u * (uint32_t)(lo & 31u);
    uint64_t v = words[lo >> 5] >> sh;
    if ((uint32_t)(lo & 31u) + n > 32u) v |= words[(lo >> 5) + 1] << (64u - sh); /* sh > 0 here */
    const uint64_t mask = n >= 32u ? ~0ull : ((1ull << (2u * n)) - 1ull);
    v &= mask;
    if (bwd) v = (rev_fields64(v) >> (64u - 2u * n)) ^ (0xAAAAAAAAAAAAAAAAULL & mask);
    return v;
}
/* the k nucleotides of the store that start at pos, as a k-mer image: first nucleotide in the lowest bits.  For an oriented k-mer x read
 * along the store this is x.r ^ (0xAAAA.. & mask); for one read against the store, x.f ^ (0xAAAA.. & mask) of its reverse complement */
MTG_DEV uint64_t us_kmer_le(const uint64_t* words, uint64_t pos, int k) { return us_peek64(words, pos, (uint32_t)k, false); }

/* the neighbourhood of an interior junction of a stored unitig, read off the store: `up` = the junction's place, resolved for the walking
 * direction (bit 1: against the stored orientation) */
MTG_DEV void adj_of_interior(const uint64_t* words, uint64_t up, int k, uint32_t& out, uint32_t& in)
{
    const uint64_t base = (up_hdr(up) + 1) * 32;
    const uint32_t off = up_off(up);
    if (!up_bwd(up)) { out = 1u << us_peek(words, base + off + (uint32_t)k - 1u, 1u, false); in = 1u << us_peek(words, base + off - 1u, 1u, false); }
    else { out = 1u << us_peek(words, base + off - 1u, 1u, true); in = 1u << us_peek(words, base + off + (uint32_t)k - 1u, 1u, true); }
}
/* Sparse index: the place of the junction j (a (k-1)-mer, jf in walking orientation, jr its reverse complement) when it is an interior junction
 * WITHOUT an entry; 0 when there is none.  Such a junction has an even offset and is not the last one, so the junction one nucleotide
 * further along the unitig has an entry: its key is j without its first nucleotide plus one of four nucleotides.  Only needed for
 * queries about k-mers that are not solid (a solid k-mer finds its place through its other junction). */
MTG_DEV uint64_t locate_junction(const Table& adj, uint64_t jf, uint64_t jr, uint32_t& lines)
{
    const int k = (int)(adj.key_bits >> 1) + 1;
    const uint64_t mk1 = kmask(k - 1), cmpl1 = 0xAAAAAAAAAAAAAAAAULL & mk1;
    for (uint32_t b = 0; b < 4; b++) {
        const uint64_t kf = ((jf << 2) | b) & mk1, kr = (jr >> 2) | ((uint64_t)(b ^ 2u) << (2 * (k - 2)));
        uint64_t aux;
        if (!adj_get(adj, kf <= kr ? kf : kr, lines, aux) || !up_is(aux)) continue;
        const uint64_t w = up_resolve(aux, kf <= kr);
        const uint64_t base = (up_hdr(w) + 1) * 32;
        const uint32_t off = up_off(w);
        lines++;
        if (!up_bwd(w)) {
            if (off < 2u) continue;
            if (us_peek64(adj.sp_words, base + off - 1u, (uint32_t)k - 1u, false) != (jr ^ cmpl1)) continue;
            return (up_hdr(w) << 32) | ((uint64_t)(off - 1u) << 5) | 1ull;
        }
        if (us_peek64(adj.sp_words, base + off + 1u, (uint32_t)k - 1u, false) != (jf ^ cmpl1)) continue;
        return (up_hdr(w) << 32) | ((uint64_t)(off + 1u) << 5) | 3ull;
    }
    return 0ull;
}

/* right neighbourhood of x: successors of x and in-neighbours of those successors (one bucket read; on a sparse index a second one and
 * two short reads of the store when the junction behind x has no entry) */
struct AdjLines { Adj a; uint32_t lines; };
MTG_DEV_COLD AdjLines adj_right_sparse(const Table adj, const Kmer x, uint64_t mk1, Adj a);
MTG_DEV Adj adj_right_t(const Table& adj, const Kmer& x, uint64_t mk1, uint32_t& lines)
{
    uint32_t m;
    const Adj a = adj_right_raw(adj, x, mk1, lines, m);
    if (m != 0 || adj.sp_words == nullptr) return a;
    const AdjLines r = adj_right_sparse(adj, x, mk1, a); /* (by value both ways: a reference into a real call would put the caller's variable into memory) */
    lines += r.lines;
    return r.a;
}
/* the junction behind x has no entry: through the other junction of x, or x is not solid */
MTG_DEV_COLD AdjLines adj_right_sparse(const Table adj, const Kmer x, uint64_t mk1, Adj a)
{
    uint32_t lines = 0;
    AdjLines out;
    /* through x's left junction: x is the k-mer behind it */
    const int k = (int)(adj.key_bits >> 1) + 1;
    const uint64_t mk = kmask(k), cmpl = 0xAAAAAAAAAAAAAAAAULL & mk;
    const uint64_t p = x.f >> 2, rp = x.r & mk1;
    uint64_t aux;
    if (adj_get(adj, p <= rp ? p : rp, lines, aux) && up_is(aux)) {
        const uint64_t w = up_resolve(aux, p <= rp);
        const uint64_t base = (up_hdr(w) + 1) * 32;
        const uint32_t off = up_off(w);
        lines += 2;
        if (!up_bwd(w)) {
            if (us_kmer_le(adj.sp_words, base + off, k) == (x.r ^ cmpl)) {
                a.out = 1u << us_peek(adj.sp_words, base + off + (uint32_t)k, 1u, false);
                a.in = 1u << ((uint32_t)(x.f >> (2 * (k - 1))) & 3u);
                a.up = (up_hdr(w) << 32) | ((uint64_t)(off + 1u) << 5) | 1ull;
                out.a = a; out.lines = lines;
                return out;
            }
        } else if (off >= 2u && us_kmer_le(adj.sp_words, base + off - 1u, k) == (x.f ^ cmpl)) {
            a.out = 1u << us_peek(adj.sp_words, base + off - 2u, 1u, true);
            a.in = 1u << ((uint32_t)(x.f >> (2 * (k - 1))) & 3u);
            a.up = (up_hdr(w) << 32) | ((uint64_t)(off - 1u) << 5) | 3ull;
            out.a = a; out.lines = lines;
            return out;
        }
    }
    /* x is not solid: the junction behind it may still be one of the graph (the walk asks this of a source k-mer that is not in the graph) */
    const uint64_t up = locate_junction(adj, x.f & mk1, x.r >> 2, lines);
    if (up) { adj_of_interior(adj.sp_words, up, k, a.out, a.in); a.up = up; }
    out.a = a; out.lines = lines;
    return out;
}

/* sum of the abundance bytes ab[start, start + count), count <= 64: aligned 8-byte reads, all in flight together, bytes outside the range
 * masked off (the array is padded past its end) */
MTG_DEV uint32_t us_ab_sum(const uint8_t* ab, uint64_t start, uint32_t count)
{
    const uint64_t a0 = start & ~7ull, end = start + count;
    uint64_t w[9];
MTG_UNROLL
    for (int j = 0; j < 9; j++) w[j] = (a0 + 8u * (uint32_t)j < end) ? *reinterpret_cast<const uint64_t*>(ab + a0 + 8u * (uint32_t)j) : 0ull;
    uint32_t sum = 0;
MTG_UNROLL
    for (int j = 0; j < 9; j++) {
        const uint64_t lo = a0 + 8u * (uint32_t)j; /* bytes lo .. lo + 7 */
        uint64_t m = ~0ull;
        if (lo < start) m &= ~0ull << (8u * (uint32_t)(start - lo));                       /* only j = 0 */
        if (lo + 8 > end) m = lo >= end ? 0ull : (m & (~0ull >> (8u * (uint32_t)(lo + 8 - end))));
        uint64_t x = w[j] & m;
        x = (x & 0x00FF00FF00FF00FFULL) + ((x >> 8) & 0x00FF00FF00FF00FFULL);
        x = (x & 0x0000FFFF0000FFFFULL) + ((x >> 16) & 0x0000FFFF0000FFFFULL);
        sum += (uint32_t)x + (uint32_t)(x >> 32);
    }
    return sum;
}
/* the same when count (<= 64) is only known after the loads have been issued (it comes with another load of the same round): nine words
 * are read whatever it is (the array is padded by more than that) and the bytes are masked afterwards */
struct AbRun { uint64_t w[9]; uint64_t start; };
MTG_DEV void us_ab_issue(const uint8_t* ab, uint64_t start, AbRun& r)
{
    const uint64_t a0 = start & ~7ull;
    r.start = start;
MTG_UNROLL
    for (int j = 0; j < 9; j++) r.w[j] = *reinterpret_cast<const uint64_t*>(ab + a0 + 8u * (uint32_t)j);
}
MTG_DEV uint32_t us_ab_finish(const AbRun& r, uint32_t count)
{
    const uint64_t start = r.start, a0 = start & ~7ull, end = start + count;
    uint32_t sum = 0;
MTG_UNROLL
    for (int j = 0; j < 9; j++) {
        const uint64_t lo = a0 + 8u * (uint32_t)j;
        uint64_t m = ~0ull;
        if (lo < start) m &= ~0ull << (8u * (uint32_t)(start - lo));
        if (lo + 8 > end) m = lo >= end ? 0ull : (m & (~0ull >> (8u * (uint32_t)(lo + 8 - end))));
        uint64_t x = r.w[j] & m;
        x = (x & 0x00FF00FF00FF00FFULL) + ((x >> 8) & 0x00FF00FF00FF00FFULL);
        x = (x & 0x0000FFFF0000FFFFULL) + ((x >> 16) & 0x0000FFFF0000FFFFULL);
        sum += (uint32_t)x + (uint32_t)(x >> 32);
    }
    return sum;
}
/* the simple path behind a pointer: position of its first nucleotide (the junction's out-edge) and how many nucleotides follow the
 * junction up to the end of the unitig in the walking direction (>= 1) */
MTG_DEV void us_run(const UStore& us, uint64_t up, int k, uint64_t& pos, uint32_t& left)
{
    const uint64_t hdr = up_hdr(up);
    const uint32_t off = up_off(up), len = (uint32_t)us.words[hdr];
    const uint64_t base = (hdr + 1) * 32;
    if (up_bwd(up)) { pos = base + off - 1; left = off; }
    else { pos = base + off + (uint32_t)k - 1; left = len - (off + (uint32_t)k - 1); }
}
/* ---- a node's place in the unitig store, for the routines that advance whole stretches of a unitig at once (frontlines, path
 * enumeration).  Known for a node whose right junction (in its walking orientation) lies inside a stored unitig: */
struct RunAt {
    uint64_t kpos;  /* store position (word * 32 + i) of the first nucleotide of the node's k-mer, in the stored orientation */
    uint32_t ahead; /* nodes ahead of it in its walking direction within the unitig (the last one is the unitig's end node) */
    uint32_t hdr;   /* header word of the unitig */
    bool bwd;       /* the node walks against the stored orientation */
};
MTG_DEV bool run_at(const UStore& us, const Adj& r, int k, RunAt& out, uint32_t& lines)
{
    if (!(r.up && popc4(r.out) == 1 && popc4(r.in) == 1)) return false;
    const uint64_t hdr = up_hdr(r.up);
    const uint32_t off = up_off(r.up), len_k = (uint32_t)us.words[hdr] - (uint32_t)k + 1u;
    lines++;
    out.hdr = (uint32_t)hdr;
    out.bwd = up_bwd(r.up);
    const uint32_t idx = out.bwd ? off : off - 1u;
    out.ahead = out.bwd ? idx : len_k - 1u - idx;
    out.kpos = (hdr + 1) * 32 + idx;
    return true;
}
/* the oriented node t nodes ahead of a node at kpos (t <= its `ahead`) */
MTG_DEV Kmer run_node(const UStore& us, uint64_t kpos, bool bwd, uint32_t t, int k)
{
    const uint64_t p = bwd ? kpos - t : kpos + t;
    const uint64_t mk = kmask(k);
    Kmer x;
    x.r = us_peek64(us.words, p, (uint32_t)k, false) ^ (0xAAAAAAAAAAAAAAAAULL & mk); /* little-endian image = reversed order: complemented = reverse complement */
    x.f = revcomp(x.r, k);
    if (bwd) { const uint64_t t2 = x.f; x.f = x.r; x.r = t2; }
    return x;
}
/* the nucleotide that leads from the node at kpos to the next one in its walking direction */
MTG_DEV uint32_t run_next_nt(const UStore& us, uint64_t kpos, bool bwd, int k)
{
    return us_peek(us.words, bwd ? kpos - 1 : kpos + (uint32_t)k, 1u, bwd);
}
/* the lookahead word an inline entry would hold (count + up to MTG_LA_MAX nucleotides past the single out-edge), read through the pointer */
MTG_DEV uint32_t la_from_up(const UStore& us, uint64_t up, int k, uint32_t& lines)
{
    uint64_t pos;
    uint32_t left;
    us_run(us, up, k, pos, left);
    lines++;
    uint32_t n = left - 1;
    if (n > (uint32_t)MTG_LA_MAX) n = MTG_LA_MAX;
    if (n == 0) return 0u;
    const bool bwd = up_bwd(up);
    return n | (us_peek(us.words, bwd ? pos - 1 : pos + 1, n, bwd) << 4);
}
/* for the bubble routines: the neighbourhood with its lookahead, whichever way the entry holds it */
MTG_DEV Adj adj_right(const Index& ix, const Kmer& x, uint64_t mk1, uint32_t& lines)
{
    Adj a = adj_right_t(ix.adj, x, mk1, lines);
    if (a.up && popc4(a.out) == 1 && popc4(a.in) == 1) a.la = la_from_up(ix.us, a.up, ix.k, lines);
    return a;
}
/* right neighbourhoods of two unrelated nodes with both home buckets in flight together; a key that is not in its home bucket of a full
 * bucket is looked up again the ordinary way (rare) */
MTG_DEV void adj_right2_raw(const Index& ix, const Kmer& xa, const Kmer& xb, uint64_t mk1, uint32_t& lines, Adj& ra, Adj& rb)
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
        if (!m && (q[u][MTG_ADJ_SLOTS - 1].x != 0 || t.sp_words != nullptr)) { *rs[u] = adj_right_t(t, *xs[u], mk1, lines); continue; } /* perhaps displaced, or (sparse index) a junction without an entry */
        rs[u]->up = 0;
        if (fw[u]) { rs[u]->out = m & 15u; rs[u]->in = m >> 4; rs[u]->la = (uint32_t)aux; }
        else { rs[u]->out = comp_mask(m >> 4); rs[u]->in = comp_mask(m & 15u); rs[u]->la = (uint32_t)(aux >> 32); }
        if (up_is(aux)) { rs[u]->la = 0; rs[u]->up = up_resolve(aux, fw[u]); }
    }
}
/* the lookahead of a pointer entry, read through the store (for callers that got the entry from adj_right2_raw) */
MTG_DEV void adj_resolve_la(const Index& ix, Adj& r, uint32_t& lines)
{
    if (r.up && popc4(r.out) == 1 && popc4(r.in) == 1) r.la = la_from_up(ix.us, r.up, ix.k, lines);
}
MTG_DEV void adj_right2(const Index& ix, const Kmer& xa, const Kmer& xb, uint64_t mk1, uint32_t& lines, Adj& ra, Adj& rb)
{
    adj_right2_raw(ix, xa, xb, mk1, lines, ra, rb);
    adj_resolve_la(ix, ra, lines);
    adj_resolve_la(ix, rb, lines);
}
/* left neighbourhood of x: .in = predecessors of x, .out = successors of every predecessor. */
MTG_DEV Adj adj_left(const Index& ix, const Kmer& x, uint64_t mk1, uint32_t& lines)
{
    const uint64_t p = x.f >> 2, rp = x.r & mk1;
    uint64_t aux;
    const uint32_t m = adj_get(ix.adj, p <= rp ? p : rp, lines, aux);
    Adj a;
    a.la = 0;
    a.up = 0;
    if (p <= rp) { a.out = m & 15u; a.in = m >> 4; }
    else { a.out = comp_mask(m >> 4); a.in = comp_mask(m & 15u); }
    if (m != 0 || ix.adj.sp_words == nullptr) return a;
    /* sparse index: through x's right junction, x is the k-mer before it */
    const int k = ix.k;
    const uint64_t mk = kmask(k), cmpl = 0xAAAAAAAAAAAAAAAAULL & mk;
    const uint64_t s = x.f & mk1, rs = x.r >> 2;
    if (adj_get(ix.adj, s <= rs ? s : rs, lines, aux) && up_is(aux)) {
        const uint64_t w = up_resolve(aux, s <= rs);
        const uint64_t base = (up_hdr(w) + 1) * 32;
        const uint32_t off = up_off(w);
        lines += 2;
        if (!up_bwd(w)) {
            if (off >= 2u && us_kmer_le(ix.adj.sp_words, base + off - 1u, k) == (x.r ^ cmpl)) {
                a.in = 1u << us_peek(ix.adj.sp_words, base + off - 2u, 1u, false);
                a.out = 1u << ((uint32_t)x.f & 3u);
                return a;
            }
        } else if (us_kmer_le(ix.adj.sp_words, base + off, k) == (x.f ^ cmpl)) {
            a.in = 1u << us_peek(ix.adj.sp_words, base + off + (uint32_t)k, 1u, true);
            a.out = 1u << ((uint32_t)x.f & 3u);
            return a;
        }
    }
    const uint64_t up = locate_junction(ix.adj, p, rp, lines); /* x is not solid */
    if (up) adj_of_interior(ix.adj.sp_words, up, k, a.out, a.in);
    return a;
}
/* the stored unitig (24 bits of its header word, as rp_unitig) whose interior holds the junction on the left of x, 0xFFFFFFFF: none */
MTG_DEV uint32_t left_junction_unitig(const Index& ix, const Kmer& x, uint64_t mk1, uint32_t& lines)
{
    const uint64_t p = x.f >> 2, rp = x.r & mk1;
    uint64_t aux;
    const uint32_t m = adj_get(ix.adj, p <= rp ? p : rp, lines, aux);
    if (m != 0) return up_is(aux) ? (uint32_t)up_hdr(aux) & 0xFFFFFFu : 0xFFFFFFFFu;
    if (ix.adj.sp_words == nullptr) return 0xFFFFFFFFu;
    const int k = ix.k;
    const uint64_t mk = kmask(k), cmpl = 0xAAAAAAAAAAAAAAAAULL & mk;
    const uint64_t s = x.f & mk1, rs = x.r >> 2;
    if (adj_get(ix.adj, s <= rs ? s : rs, lines, aux) && up_is(aux)) {
        const uint64_t w = up_resolve(aux, s <= rs);
        const uint64_t base = (up_hdr(w) + 1) * 32;
        const uint32_t off = up_off(w);
        lines++;
        const bool is_x = !up_bwd(w) ? (off >= 2u && us_kmer_le(ix.adj.sp_words, base + off - 1u, k) == (x.r ^ cmpl)) : us_kmer_le(ix.adj.sp_words, base + off, k) == (x.f ^ cmpl);
        if (is_x) return (uint32_t)up_hdr(w) & 0xFFFFFFu;
    }
    const uint64_t up = locate_junction(ix.adj, p, rp, lines);
    return up ? (uint32_t)up_hdr(up) & 0xFFFFFFu : 0xFFFFFFFFu;
}
/* abundance of x (0: not solid).  Sparse index: the byte next to x's place in the store, found through the junction behind or before it;
 * the table only for the k-mers of no stored unitig */
MTG_DEV uint32_t abundance(const Index& ix, const Kmer& x, uint32_t& lines)
{
    if (ix.adj.sp_words == nullptr) return table_get<MTG_ABND_SLOTS>(ix.abnd, canon(x), lines);
    const int k = ix.k;
    const uint64_t mk = kmask(k), mk1 = kmask(k - 1), cmpl = 0xAAAAAAAAAAAAAAAAULL & mk;
    uint64_t aux;
    const uint64_t s = x.f & mk1, rs = x.r >> 2;
    if (adj_get(ix.adj, s <= rs ? s : rs, lines, aux) && up_is(aux)) {
        const uint64_t w = up_resolve(aux, s <= rs);
        const uint64_t base = (up_hdr(w) + 1) * 32, idx = up_bwd(w) ? up_off(w) : up_off(w) - 1u;
        lines++;
        return us_kmer_le(ix.us.words, base + idx, k) == (up_bwd(w) ? (x.f ^ cmpl) : (x.r ^ cmpl)) ? (uint32_t)ix.us.ab[base + idx] : 0u;
    }
    const uint64_t p = x.f >> 2, rp = x.r & mk1;
    if (adj_get(ix.adj, p <= rp ? p : rp, lines, aux) && up_is(aux)) {
        const uint64_t w = up_resolve(aux, p <= rp);
        const uint32_t off = up_off(w);
        if (up_bwd(w) && off < 1u) return 0u;
        const uint64_t base = (up_hdr(w) + 1) * 32, idx = up_bwd(w) ? off - 1u : off;
        lines++;
        return us_kmer_le(ix.us.words, base + idx, k) == (up_bwd(w) ? (x.f ^ cmpl) : (x.r ^ cmpl)) ? (uint32_t)ix.us.ab[base + idx] : 0u;
    }
    return table_get<MTG_ABND_SLOTS>(ix.abnd, canon(x), lines);
}
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
    if (ix.adj.sp_words != nullptr) return abundance(ix, make_kmer(p.key, ix.k), lines); /* sparse index: the byte lies next to the store (the bucket requested by ab_issue only serves the k-mers of no unitig) */
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

/* ---- unitig store construction from DENSE tables (after every insertion and every lookahead): the construction of rounds 1-3.  The product
 * builds the store from the junction table since round 4 (mtg_build.h) and dropped this path in round 5; us_is_start / us_walk / us_plan /
 * us_emit / us_link below are kept as the TEST-ONLY reference construction of the emulation (tests/emu/emu_us.h), which builds every index
 * both ways and compares them (test_lean_build_equals_the_legacy_build) -------------------------------------------------------
 * A junction between the consecutive solid k-mers p = a+J and y = J+b is ELIGIBLE when it is simple (y is the only successor of p, p
 * the only predecessor of y) and neither a turning point nor a self loop: J is not its own reverse complement (then y = rc(p)), p != y,
 * and neither k-mer is its own reverse complement.  Chains of eligible junctions are linear or closed; along a linear chain all canonical
 * k-mers are distinct (a path that met rc(x) after x would be its own reverse complement and so have a palindromic junction or a
 * self-complementary node in its middle), and two chains share a canonical k-mer only if one is the reverse complement of the other.
 * The store holds every linear chain of at least two k-mers once (from the end with the smaller canonical k-mer); closed chains and
 * single k-mers keep their inline lookaheads.  The property is symmetric under reverse complement, so it is a property of the
 * canonical junction. */
MTG_DEV bool us_eligible(const Kmer& p, const Kmer& y, int k)
{
    const uint64_t mk1 = kmask(k - 1);
    if ((p.f & mk1) == (p.r >> 2)) return false; /* palindromic junction */
    if (p.f == y.f) return false;                /* self loop */
    if (p.f == p.r || y.f == y.r) return false;  /* self-complementary k-mer (even k) */
    return true;
}
/* the solid oriented k-mer x starts a chain: it has no eligible junction on its left */
MTG_DEV bool us_is_start(const Index& ix, const Kmer& x, uint32_t& lines)
{
    const int k = ix.k;
    const Adj l = adj_left(ix, x, kmask(k - 1), lines);
    if (!(popc4(l.in) == 1 && popc4(l.out) == 1)) return true;
    const Kmer p = kmer_prev(x, (uint32_t)ctz4(l.in), k, kmask(k));
    return !us_eligible(p, x, k);
}
/* Walks the chain that starts with x to its end.  Before the pointers are written the entries hold lookaheads, which save most of the
 * reads: the junctions they cover are simple, their eligibility is checked on the k-mers.  sink(nt) receives every nucleotide after the
 * first k-mer; returns the number of k-mers (capped so that the offsets fit a pointer), `end` = the last one. */
template <typename Sink> MTG_DEV uint32_t us_walk(const Index& ix, const Kmer& x, Kmer& end, uint32_t& lines, Sink sink)
{
    const int k = ix.k;
    const uint64_t mk = kmask(k), mk1 = kmask(k - 1);
    const uint32_t cap = MTG_US_MAX_LEN - (uint32_t)k; /* k-mers */
    Kmer cur = x;
    uint32_t n = 1;
    for (;;) {
        const Adj a = adj_right_t(ix.adj, cur, mk1, lines);
        if (!(popc4(a.out) == 1 && popc4(a.in) == 1)) break;
        uint32_t nt = (uint32_t)ctz4(a.out), la = a.la, known = la & 15u;
        la >>= 4;
        bool stop = false;
        for (;;) {
            const Kmer y = kmer_next(cur, nt, k, mk);
            if (!us_eligible(cur, y, k) || n >= cap) { stop = true; break; }
            cur = y;
            n++;
            sink(nt);
            if (known == 0) break;
            nt = la & 3u;
            la >>= 2;
            known--;
        }
        if (stop) break;
    }
    end = cur;
    return n;
}
struct UsNoSink { MTG_DEV void operator()(uint32_t) const {} };
/* one record per stored unitig, in the order of their header words */
struct UsRec {
    uint64_t start_f; /* first k-mer, in the stored orientation */
    uint32_t len_k;   /* k-mers */
    uint32_t pad_;
    uint64_t hdr;
};
/* pass 1, per solid oriented k-mer x: if x starts a chain of >= 2 k-mers and is the end the chain is stored from, reserves the chain's
 * words and its record.  rec == nullptr: only counts the chain starts (every stored unitig has two, one per strand, so half their number
 * bounds the number of records). */
MTG_DEV uint64_t us_words_of(uint32_t len_k, int k) { return 1ull + ((uint64_t)len_k + (uint32_t)k - 1 + 31) / 32; }
MTG_DEV void us_plan_start(const Index& ix, const Kmer& x, unsigned long long* cursor_words, unsigned long long* cursor_recs, UsRec* rec, uint64_t rec_cap, uint32_t& lines);
MTG_DEV void us_plan(const Index& ix, const Kmer& x, unsigned long long* cursor_words, unsigned long long* cursor_recs, UsRec* rec, uint64_t rec_cap, uint32_t& lines)
{
    if (!us_is_start(ix, x, lines)) return;
    if (!rec) {
#ifdef MTG_EMU
        __sync_fetch_and_add(cursor_recs, 1ull);
#else
        atomicAdd(cursor_recs, 1ull);
#endif
        return;
    }
    us_plan_start(ix, x, cursor_words, cursor_recs, rec, rec_cap, lines);
}
/* the same for an x known to start a chain (the device build collects the starts first, so that every lane of the walking kernel has one) */
MTG_DEV void us_plan_start(const Index& ix, const Kmer& x, unsigned long long* cursor_words, unsigned long long* cursor_recs, UsRec* rec, uint64_t rec_cap, uint32_t& lines)
{
    Kmer end;
    const uint32_t n = us_walk(ix, x, end, lines, UsNoSink());
    if (n < 2) return;
    if (n >= MTG_US_MAX_LEN - (uint32_t)ix.k) return; /* cut short by the offset range: not stored (its k-mers keep their lookaheads), so that the
                                                        stored unitigs are whole chains and no canonical k-mer lies in two of them */
    /* the reverse complement of the chain starts with rc(end): the one whose first k-mer has the smaller canonical value is stored; the
     * canonical k-mers of a chain are distinct, so there is no tie between different k-mers */
    if (!(canon(x) < canon(end))) return;
#ifdef MTG_EMU
    const uint64_t r = __sync_fetch_and_add(cursor_recs, 1ull);
    const uint64_t w = __sync_fetch_and_add(cursor_words, (unsigned long long)us_words_of(n, ix.k));
#else
    const uint64_t r = atomicAdd(cursor_recs, 1ull);
    const uint64_t w = atomicAdd(cursor_words, (unsigned long long)us_words_of(n, ix.k));
#endif
    if (r < rec_cap) { rec[r].start_f = x.f; rec[r].len_k = n; rec[r].pad_ = 0; rec[r].hdr = w; }
}
/* pass 2, per record: the sequence of the unitig into the store */
MTG_DEV void us_emit(const Index& ix, const UsRec& r, uint32_t& lines)
{
    const int k = ix.k;
    uint64_t* w = ix.us.words + r.hdr;
    w[0] = (uint64_t)r.len_k + (uint32_t)k - 1; /* length in nucleotides */
    uint64_t acc = 0;
    uint32_t nacc = 0, wpos = 1;
    auto push = [&](uint32_t nt) {
        acc |= (uint64_t)nt << (2 * nacc);
        if (++nacc == 32) { w[wpos++] = acc; acc = 0; nacc = 0; }
    };
    for (int i = k - 1; i >= 0; i--) push((uint32_t)(r.start_f >> (2 * i)) & 3u);
    Kmer end;
    const Kmer x = make_kmer(r.start_f, k);
    us_walk(ix, x, end, lines, push);
    if (nacc) w[wpos] = acc;
}
/* pass 3, per k-mer i of a stored unitig (any thread): its abundance into the store and, for i >= 1, the pointer into the entry of the
 * junction on its left.  Runs after every us_emit (the entries still hold lookaheads until here). */
MTG_DEV void us_link(const Index& ix, const UsRec& r, uint32_t i, uint32_t& lines)
{
    const int k = ix.k;
    const uint64_t mk = kmask(k), mk1 = kmask(k - 1);
    const uint64_t* w = ix.us.words + r.hdr + 1;
    const uint32_t s = 2u * (i & 31u);
    const uint64_t lo = w[i >> 5] >> s;
    const uint64_t hi = s ? (w[(i >> 5) + 1] << (64u - s)) : 0ull; /* the store is padded by one word */
    Kmer x;
    x.r = ((lo | hi) & mk) ^ (0xAAAAAAAAAAAAAAAAULL & mk);
    x.f = revcomp(x.r, k);
    ix.us.ab[(r.hdr + 1) * 32 + i] = (uint8_t)table_get<MTG_ABND_SLOTS>(ix.abnd, canon(x), lines);
    if (i == 0) return;
    const uint64_t J = x.f >> 2, rJ = x.r & mk1;
    uint64_t* e = adj_find(ix.adj, J <= rJ ? J : rJ);
    if (e) e[1] = up_make(r.hdr, i, !(J <= rJ));
}

/* ---- the sparse index, built from the unitig store (see "sparse index" above): per k-mer i of a stored unitig, the entries its junctions
 * need in the tables `ix` (new, empty but for what other k-mers have contributed): the end junctions of the unitig get the k-mer's edge bits
 * (what index_insert contributes on that side), an interior junction that is kept gets its two bits and its pointer.  Returns 1 when an
 * insertion overflowed its displacement range.  with_bloom: the k-mer also enters the Bloom filter (an index rebuilt from its container). */
MTG_DEV bool sparse_keeps(uint32_t off, uint32_t len_k) { return (off & 1u) != 0 || off == len_k - 1u; }
MTG_DEV int sparse_link(const Index& ix, const UsRec& r, uint32_t i, bool with_bloom)
{
    const int k = ix.k;
    const uint64_t mk = kmask(k), mk1 = kmask(k - 1);
    const uint64_t* w = ix.us.words + r.hdr + 1;
    const uint32_t sft = 2u * (i & 31u);
    const uint64_t lo = w[i >> 5] >> sft;
    const uint64_t hi = sft ? (w[(i >> 5) + 1] << (64u - sft)) : 0ull; /* the store is padded by one word */
    Kmer x;
    x.r = ((lo | hi) & mk) ^ (0xAAAAAAAAAAAAAAAAULL & mk);
    x.f = revcomp(x.r, k);
    int fail = 0;
    if (with_bloom && ix.bloom.bits) bloom_insert(ix.bloom, x, k);
    const uint32_t a = (uint32_t)(x.f >> (2 * (k - 1))) & 3u, b = (uint32_t)x.f & 3u;
    const uint64_t pre = x.f >> 2, rpre = x.r & mk1;  /* the junction before x: pre + b is solid */
    const uint64_t suf = x.f & mk1, rsuf = x.r >> 2;  /* the junction behind x: a + suf is solid */
    const uint32_t pre_bit = pre <= rpre ? 1u << b : 1u << (4 + (b ^ 2u)), suf_bit = suf <= rsuf ? 1u << (4 + a) : 1u << (a ^ 2u);
    /* an end junction may be its own reverse complement: the two strands of the k-mer then contribute different bits (index_insert adds both) */
    const uint32_t pre_bit2 = rpre <= pre ? 1u << (4 + (b ^ 2u)) : 1u << b, suf_bit2 = rsuf <= suf ? 1u << (a ^ 2u) : 1u << (4 + a);
    if (i == 0) fail |= adj_or(ix.adj, pre <= rpre ? pre : rpre, pre_bit | pre_bit2) & 1;
    else if (sparse_keeps(i, r.len_k)) {
        /* the interior junction between k-mers i - 1 and i: x's bit on this side and that of the k-mer before it */
        const uint32_t a0 = (uint32_t)((w[(i - 1) >> 5] >> (2u * ((i - 1) & 31u))) & 3ull); /* first nucleotide of k-mer i - 1 */
        const uint32_t prev_bit = pre <= rpre ? 1u << (4 + a0) : 1u << (a0 ^ 2u);          /* its "a + suf is solid" on the junction pre */
        const uint64_t key = pre <= rpre ? pre : rpre;
        fail |= adj_set_new(ix.adj, key, pre_bit | prev_bit, up_make(r.hdr, i, !(pre <= rpre))) & 1;
    }
    if (i == r.len_k - 1) fail |= adj_or(ix.adj, suf <= rsuf ? suf : rsuf, suf_bit | suf_bit2) & 1;
    return fail;
}
/* is the solid canonical k-mer c one of a stored unitig?  (on the dense index the unitigs were built from: every interior junction has its pointer) */
MTG_DEV bool kmer_stored(const Index& ix, uint64_t c, uint32_t& lines)
{
    const int k = ix.k;
    const uint64_t mk = kmask(k), mk1 = kmask(k - 1), cmpl = 0xAAAAAAAAAAAAAAAAULL & mk;
    const Kmer x = make_kmer(c, k);
    uint64_t aux;
    const uint64_t s = x.f & mk1, rs = x.r >> 2;
    if (adj_get(ix.adj, s <= rs ? s : rs, lines, aux) && up_is(aux)) {
        const uint64_t wv = up_resolve(aux, s <= rs);
        const uint64_t base = (up_hdr(wv) + 1) * 32, idx = up_bwd(wv) ? up_off(wv) : up_off(wv) - 1u;
        if (us_kmer_le(ix.us.words, base + idx, k) == (up_bwd(wv) ? (x.f ^ cmpl) : (x.r ^ cmpl))) return true;
    }
    const uint64_t p = x.f >> 2, rp = x.r & mk1;
    if (adj_get(ix.adj, p <= rp ? p : rp, lines, aux) && up_is(aux)) {
        const uint64_t wv = up_resolve(aux, p <= rp);
        const uint32_t off = up_off(wv);
        if (up_bwd(wv) && off < 1u) return false;
        const uint64_t base = (up_hdr(wv) + 1) * 32, idx = up_bwd(wv) ? off - 1u : off;
        if (us_kmer_le(ix.us.words, base + idx, k) == (up_bwd(wv) ? (x.f ^ cmpl) : (x.r ^ cmpl))) return true;
    }
    return false;
}

/* ---- the solid k-mers, read back from the ABND table: (bucket, tag) is lossless and the hash a bijection, so slot s of the table gives
 * its canonical k-mer and abundance (0: empty slot).  Every distinct solid k-mer comes out exactly once, whatever the index was built
 * from (used by the unitig construction and by the index writer). */
MTG_DEV uint64_t unmix(uint64_t h, uint32_t key_bits)
{
    const uint64_t M = (1ULL << key_bits) - 1, C = 0xBF58476D1CE4E5B9ULL;
    uint64_t inv = C; /* Newton iteration for the inverse of the odd multiplier modulo 2^64: 3 correct bits double every round */
    for (int i = 0; i < 6; i++) inv *= 2 - C * inv;
    const uint32_t sh = key_bits >> 1;
    h ^= h >> sh;
    h = (h * inv) & M;
    h ^= h >> sh;
    return h;
}
MTG_DEV uint32_t abnd_slot_kmer(const Table& t, uint64_t slot, uint64_t& kmer)
{
    const uint64_t v = t.slots[slot];
    if (v == 0) return 0;
    const uint64_t b = slot / MTG_ABND_SLOTS, tag = v >> (8 + MTG_DISP_BITS), disp = (v >> 8) & MTG_MAX_DISP;
    const uint64_t home = b >= disp ? b - disp : b + t.nbuckets - disp;
    /* H = tag + t * 2^tag_bits with bucket_of(H) == home: t is about home * 2^(key_bits - tag_bits) / nbuckets */
    const uint32_t lg = t.key_bits - t.tag_bits; /* floor(log2 nbuckets) */
#ifdef MTG_EMU
    const uint64_t t0 = (uint64_t)((((unsigned __int128)home) << lg) / t.nbuckets);
#else
    const uint64_t t0 = lg < 32 ? ((home << lg) / t.nbuckets) : (uint64_t)(((double)home / (double)t.nbuckets) * (double)(1ull << lg));
#endif
    const uint64_t span = 1ull << lg;
    for (int d = -2; d <= 3; d++) {
        const uint64_t tt = t0 + (uint64_t)(int64_t)d;
        if (tt >= span) continue; /* also catches the wrap below zero */
        const uint64_t H = (tt << t.tag_bits) | tag;
        if (bucket_of(H, t.nbuckets, t.key_bits) == home) { kmer = unmix(H, t.key_bits); return (uint32_t)(v & 255); }
    }
    kmer = ~0ULL; /* cannot happen: the slot was written from such an H */
    return 0;
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

/* look-up of a counted k-mer: its count, 0 when it is not in the table */
MTG_DEV uint32_t count_lookup(const CountTable& t, uint64_t c, uint32_t& lines)
{
    uint64_t h = mix64(c) & t.mask;
    for (int probe = 0; probe < 8192; probe++) {
        const uint64_t cur = t.keys[h];
        lines++;
        if (cur == c) return t.counts[h];
        if (cur == ~0ULL) return 0;
        h = (h + 1) & t.mask;
    }
    return 0;
}

} // namespace mtg
#endif
