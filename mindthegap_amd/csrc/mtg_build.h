/*
 * mtg_build.h -- the lean index construction (Graph::create, /root/reference/src/Filler.cpp:172-226) as device functions: the graph as a
 * junction table, chain starts / statistics from one streaming pass, the unitig store written by the walks that measure the chains.
 * Included by mtg_gpu_build.hip (the kernels) and by the test-only emulation (tests/emu/emu_us.h), which runs the same functions serially.
 * Kept out of mtg_dev.h so that a change here does not recompile the fill kernels.
 */
#ifndef MTG_BUILD_H
#define MTG_BUILD_H
#include <math.h>
#include "mtg_dev.h"

namespace mtg {

/* The same for a whole bucket, without the search: the hashes H that bucket_of() sends to bucket `home` form the interval that begins at
 * first = ceil(home * 2^key_bits / nbuckets) and is at most 2^tag_bits long (nbuckets >= 2^(key_bits - tag_bits)), so the H of a slot is the
 * one value with the slot's tag in its low bits at or after `first`.  One division per bucket instead of six 128-bit products per slot: what
 * the streaming pass over the junction table pays (k_jt_scan). */
MTG_DEV uint64_t bucket_first_h(uint64_t home, uint64_t nb, uint32_t key_bits)
{
    if (home == 0) return 0;
    /* an estimate in double (off by at most a few thousand), corrected exactly: home * 2^key_bits - e * nb is small, so its low 64 bits are it */
    const uint64_t e = (uint64_t)((double)home * ((double)(1ull << key_bits) / (double)nb));
    const int64_t diff = (int64_t)((home << key_bits) - e * nb);
    /* ceil(diff / nb) without a 64-bit division (a few hundred instructions on this machine, once per bucket of a 10^9-bucket table): |diff| is
     * below 2^46 and nb below 2^40, both exact in double, so the quotient's estimate is off by at most one */
    int64_t q = (int64_t)floor((double)diff / (double)nb);
    while (q * (int64_t)nb < diff) q++;          /* the smallest q with q * nb >= diff */
    while ((q - 1) * (int64_t)nb >= diff) q--;
    const uint64_t first = (uint64_t)((int64_t)e + q);
#ifdef MTG_EMU
    /* the emulator checks the device's arithmetic against the 128-bit quotient on every bucket it decodes */
    if (first != (uint64_t)(((((unsigned __int128)home) << key_bits) + nb - 1) / nb)) abort();
#endif
    return first;
}
/* key and value of slot word v found in bucket b, given first = bucket_first_h(b) (used when the slot sits in its home bucket) */
MTG_DEV uint32_t slot_key_in_bucket(const Table& t, uint64_t b, uint64_t first_b, uint64_t v, uint64_t& key)
{
    const uint64_t tag = v >> (8 + MTG_DISP_BITS), disp = (v >> 8) & MTG_MAX_DISP;
    uint64_t first = first_b;
    if (disp) first = bucket_first_h(b >= disp ? b - disp : b + t.nbuckets - disp, t.nbuckets, t.key_bits);
    const uint64_t tm = (1ULL << t.tag_bits) - 1;
    uint64_t H = (first & ~tm) | tag;
    if (H < first) H += tm + 1;
    key = unmix(H, t.key_bits);
    return (uint32_t)(v & 255);
}

/* ---- the lean build (round 4): the graph as a JUNCTION TABLE, the unitig store straight from it -------------------------------------
 * Graph::create without the dense tables (/root/reference/src/Filler.cpp:172-226).  The edge masks of the canonical (k-1)-mers ARE the
 * solid set: the k-mer J+b is solid iff bit b of J's entry is set, so a table of 8-byte slots [tag | disp | mask] (the layout of the ABND
 * table, key_bits = 2(k-1), MTG_ABND_SLOTS slots per 32-byte bucket) answers every neighbourhood question of the construction:
 *   - a junction seen in one of its two orientations is a VIEW {out: nts b with J+b solid, in: nts a with a+J solid};
 *   - every oriented solid k-mer is "prefix view + one out-nucleotide" exactly once (a palindromic junction has one view);
 *   - a view that is simple (one in, one out) and eligible (us_eligible) is the interior of a chain; every k-mer behind any other view
 *     starts a chain, or is a k-mer of no chain when its right view is no interior either.
 * So chain starts, the k-mers of no unitig and the statistics of the graph come out of ONE streaming pass over the table with a probe
 * only next to the (rare) views that are no chain interior, instead of a neighbourhood look-up per solid k-mer.  The abundances never
 * enter a table of their own: they are asked of the SOURCE the k-mers came from (a count table, a k-mer table, or a function of the k-mer)
 * when the store is written. */
struct JView {
    uint32_t out, in;
};
MTG_DEV JView jt_view(uint32_t m, bool key_is_oriented)
{
    JView v;
    if (key_is_oriented) { v.out = m & 15u; v.in = m >> 4; }
    else { v.out = comp_mask(m >> 4); v.in = comp_mask(m & 15u); }
    return v;
}
MTG_DEV bool jt_simple(const JView& v) { return popc4(v.out) == 1 && popc4(v.in) == 1; }
/* the view behind x (its successors, the predecessors of those) and before x (its predecessors, the successors of those) */
/* A junction's slot carries, while the chains are walked, one more bit: JT_MARK, "a walker crossed here at a multiple of JT_MARK_EVERY steps"
 * (below: the walks that meet in the middle).  Bit 63 is free because the junction table has at least 2^(key_bits - 49) buckets (jt_min_buckets);
 * every reader of the table masks it. */
static const uint64_t JT_MARK = 1ull << 63;
MTG_HD uint64_t jt_min_buckets(uint32_t key_bits) { return key_bits > 49 ? (1ULL << (key_bits - 49)) : 1; }
/* the slot of `key` (nullptr: absent) and its word without the mark; marked: the mark was set when the bucket was read */
MTG_DEV uint64_t jt_lookup(const Table& t, uint64_t key, uint32_t& lines, uint64_t*& slot, bool& marked)
{
    const uint64_t H = mix(key, t.key_bits);
    uint64_t b = bucket_of(H, t.nbuckets, t.key_bits);
    const uint64_t tag = H & ((1ULL << t.tag_bits) - 1);
    slot = nullptr;
    marked = false;
    for (uint32_t d = 0; d <= MTG_MAX_DISP; d++) {
        uint64_t* p = t.slots + b * MTG_ABND_SLOTS;
        uint64_t q[MTG_ABND_SLOTS];
MTG_UNROLL
        for (int i = 0; i < MTG_ABND_SLOTS / 2; i++) { const U64x2 v = ld_table(reinterpret_cast<const U64x2*>(p) + i); q[2 * i] = v.x; q[2 * i + 1] = v.y; }
        lines++;
        const uint64_t want = (tag << MTG_DISP_BITS) | d;
        for (int i = 0; i < MTG_ABND_SLOTS; i++) {
            const uint64_t v = q[i] & ~JT_MARK;
            if (v != 0 && (v >> 8) == want) { slot = p + i; marked = (q[i] & JT_MARK) != 0; return v; }
        }
        if (q[MTG_ABND_SLOTS - 1] == 0) return 0; /* slots fill in order: a free last slot means the key cannot be further away */
        b = (b + 1 == t.nbuckets) ? 0 : b + 1;
    }
    return 0;
}
MTG_DEV JView jt_right(const Table& jt, const Kmer& x, uint64_t mk1, uint32_t& lines)
{
    const uint64_t s = x.f & mk1, rs = x.r >> 2;
    uint64_t* slot;
    bool marked;
    return jt_view((uint32_t)(jt_lookup(jt, s <= rs ? s : rs, lines, slot, marked) & 255u), s <= rs);
}
MTG_DEV JView jt_left(const Table& jt, const Kmer& x, uint64_t mk1, uint32_t& lines)
{
    const uint64_t p = x.f >> 2, rp = x.r & mk1;
    uint64_t* slot;
    bool marked;
    return jt_view((uint32_t)(jt_lookup(jt, p <= rp ? p : rp, lines, slot, marked) & 255u), p <= rp);
}
/* table_or with the bucket read in one go before any atomic (the entry usually exists with its bits, or the first free slot takes it).
 * A stale read is harmless: the tag part of a slot is written once, a slot seen empty is claimed by compare-and-swap, bits seen missing
 * are OR-ed in again.  Returns as table_or. */
/* mark: 0, or JT_MARK for a junction the scan must look at in full (jt_special): the entry carries the bit from its first insertion on */
MTG_DEV int jt_or_h(const Table& t, uint64_t H, uint32_t bits, uint64_t mark = 0);
MTG_DEV int jt_or(const Table& t, uint64_t key, uint32_t bits, uint64_t mark = 0) { return jt_or_h(t, mix(key, t.key_bits), bits, mark); }
/* the same for a key given by its hash H = mix(key) (the partitioned construction carries hashes, not keys) */
MTG_DEV int jt_or_h(const Table& t, uint64_t H, uint32_t bits, uint64_t mark)
{
    uint64_t b = bucket_of(H, t.nbuckets, t.key_bits);
    const uint64_t tag = H & ((1ULL << t.tag_bits) - 1);
    const uint64_t add = (uint64_t)bits | mark;
    for (uint32_t d = 0; d <= MTG_MAX_DISP; d++) {
        const uint64_t want = (tag << MTG_DISP_BITS) | d;
        uint64_t* p = t.slots + b * MTG_ABND_SLOTS;
        uint64_t q[MTG_ABND_SLOTS];
MTG_UNROLL
        for (int i = 0; i < MTG_ABND_SLOTS / 2; i++) { const U64x2 v = ld_table(reinterpret_cast<const U64x2*>(p) + i); q[2 * i] = v.x; q[2 * i + 1] = v.y; }
        for (int i = 0; i < MTG_ABND_SLOTS; i++) {
            uint64_t v = q[i];
            if (v == 0) {
                v = atomic_cas64(p + i, 0, (want << 8) | add);
                if (v == 0) return 2;
            }
            if (((v & ~JT_MARK) >> 8) == want) {
                if ((v & add) != add) atomic_or64(p + i, add);
                return 0;
            }
        }
        b = (b + 1 == t.nbuckets) ? 0 : b + 1;
    }
    return 1;
}
/* The junctions whose entry the scan cannot judge by its edge bits alone (odd k; jt_scan_bucket): the palindromic ones (one view instead of two) and
 * the runs of one nucleotide (the self loop a + J == J + b that us_eligible rules out).  One junction in 10^9 of a random sequence; they are flagged
 * with JT_MARK when they are inserted, every other entry with one bit on each side is a chain interior without its key being looked at -- getting
 * the key back (the hash undone, a division's worth of arithmetic for a displaced slot) and its reverse complement made k_jt_scan the one kernel of
 * the construction that is bound by its vector instructions (290 per entry, round 6).  Even k: no flags, the scan looks at every key as before. */
MTG_DEV bool jt_special(uint64_t jf, uint64_t jr, uint32_t key_bits)
{
    if ((key_bits & 2u) != 0) return false; /* key_bits = 2 (k - 1): k - 1 odd, k even */
    const uint64_t lsb = 0x5555555555555555ULL & ((1ULL << key_bits) - 1ULL);
    const uint32_t c = (uint32_t)jf & 3u;
    const uint64_t run = ((c & 1u) ? lsb : 0ULL) | ((c & 2u) ? lsb << 1 : 0ULL);
    return jf == jr || jf == run;
}
/* one occurrence of the junction J ((k-1)-mer jf, its reverse complement jr) with the nucleotide before it (a, when has_a: the k-mer a+J is
 * solid) and behind it (b, when has_b: J+b is solid): what index_insert contributes to J's entry from the k-mers on its two sides (a
 * palindromic junction gets the bits of both strands).  Returns as table_or. */
MTG_DEV uint32_t jt_junction_bits(uint64_t jf, uint64_t jr, bool has_a, uint32_t a, bool has_b, uint32_t b)
{
    uint32_t bits = 0;
    if (jf <= jr) bits |= (has_b ? 1u << b : 0u) | (has_a ? 1u << (4 + a) : 0u);
    if (jr <= jf) bits |= (has_b ? 1u << (4 + (b ^ 2u)) : 0u) | (has_a ? 1u << (a ^ 2u) : 0u);
    return bits;
}
MTG_DEV int jt_insert_junction(const Table& jt, uint64_t jf, uint64_t jr, bool has_a, uint32_t a, bool has_b, uint32_t b)
{
    const uint32_t bits = jt_junction_bits(jf, jr, has_a, a, has_b, b);
    if (!bits) return 0;
    return jt_or(jt, jf <= jr ? jf : jr, bits, jt_special(jf, jr, jt.key_bits) ? JT_MARK : 0ULL);
}
/* the two junctions of the solid canonical k-mer c */
MTG_DEV int jt_insert_kmer(const Table& jt, uint64_t c, int k)
{
    const uint64_t mk1 = kmask(k - 1);
    const uint64_t r = revcomp(c, k);
    int fail = jt_insert_junction(jt, c >> 2, r & mk1, false, 0u, true, (uint32_t)c & 3u) & 1;
    fail |= jt_insert_junction(jt, c & mk1, r >> 2, true, (uint32_t)(c >> (2 * (k - 1))) & 3u, false, 0u) & 1;
    return fail;
}
/* what the tables store of an abundance */
MTG_DEV uint32_t ab_stored(uint32_t a) { return a > 255u ? 255u : (a ? a : 1u); }
/* abundance sources of the lean build */
struct AbFromTable {
    Table abnd;
    MTG_DEV uint32_t operator()(uint64_t c, uint32_t& lines) const { return table_get<MTG_ABND_SLOTS>(abnd, c, lines); }
};
struct AbFromCounts {
    CountTable t;
    MTG_DEV uint32_t operator()(uint64_t c, uint32_t& lines) const { return count_lookup(t, c, lines); }
};

/* counters of the scan (one slot each, added up by the kernels) */
enum {
    JT_C_ORIENTED = 0, /* oriented solid k-mers */
    JT_C_IN_NOT1,      /* ... whose in-degree is not 1 */
    JT_C_BOTH_NOT1,    /* ... whose in- and out-degree are both not 1 */
    JT_C_SELF,         /* self-complementary k-mers (even k) */
    JT_C_SELF_BRANCH,  /* ... that are branching */
    JT_C_INTERIOR,     /* views that are the interior of a chain (two per eligible junction) */
    JT_C_STARTS,       /* cursor of the starts list */
    JT_C_LEFT,         /* cursor of the list of k-mers of no chain */
    JT_C_SAT,          /* abundances above 255 */
    JT_C_WORDS,        /* cursor of the store's words (plan) */
    JT_C_RECS,         /* cursor of the records (plan) */
    JT_C_STORED_VIEWS, /* interior views of the stored unitigs (plan): equal to JT_C_INTERIOR unless a chain is closed or too long */
    JT_C_N
};
struct JtAcc {
    unsigned long long c[6];
};
/* the key of slot s of a table of MTG_ABND_SLOTS-slot buckets, and its value (0: empty) */
MTG_DEV uint32_t jt_slot_key(const Table& t, uint64_t slot, uint64_t& key)
{
    const uint64_t v = t.slots[slot] & ~JT_MARK;
    if (v == 0) return 0;
    const uint64_t b = slot / MTG_ABND_SLOTS;
    return slot_key_in_bucket(t, b, bucket_first_h(b, t.nbuckets, t.key_bits), v, key);
}
/* is the view {v of the oriented junction jf} the interior of a chain?  p = a+J, y = J+b its two k-mers */
MTG_DEV bool jt_view_interior(const JView& v, uint64_t jf, int k, Kmer& p, Kmer& y)
{
    if (!jt_simple(v)) return false;
    const uint64_t mk = kmask(k);
    p = make_kmer((((uint64_t)ctz4(v.in) << (2 * (k - 1))) | jf) & mk, k);
    y = kmer_next(p, (uint32_t)ctz4(v.out), k, mk);
    return us_eligible(p, y, k);
}
/* Scan, per entry (canonical junction J with mask m): statistics into acc, and for every view that is no chain interior the k-mers behind
 * it: chain starts (oriented k-mer, forward value) into starts[], k-mers of no chain (canonical, with the abundance the source gives) into
 * left_k / left_a.  Lists may be null (counting pass); cursors count either way. */
/* where the scan puts a chain start: the list in device memory behind its cursor (one atomic a start -- on ONE word: 1.2e6 of them are 8 ms of the
 * human-scale scan, 150 M a second is what a line takes), or (the device, k_jt_scan) a queue of the wave in LDS that is emptied with one atomic */
struct ScanStartsGlobal {
    unsigned long long* counters;
    uint64_t* starts;
    unsigned long long cap_starts;
    MTG_DEV void operator()(uint64_t start_f) const
    {
        const unsigned long long at = atomic_add64(&counters[JT_C_STARTS], 1ull);
        if (starts && at < cap_starts) starts[at] = start_f;
    }
};
template <typename Src, typename StartSink>
MTG_DEV void jt_scan_entry(const Table& jt, int k, uint64_t J, uint32_t m, const Src& src, JtAcc& acc, unsigned long long* counters,
                           const StartSink& put_start, uint64_t* left_k, uint32_t* left_a, unsigned long long cap_left, uint32_t& lines)
{
    const uint64_t mk = kmask(k), mk1 = kmask(k - 1);
    const uint64_t rJ = revcomp(J, k - 1);
    /* the entry of nearly every junction of a genome: one bit on each side.  Both views are then simple, and eligibility is a property of the
     * canonical junction (us_eligible is symmetric under reverse complement): two oriented k-mers, two interior views, nothing to probe */
    if (popc4(m & 15u) == 1 && popc4(m >> 4) == 1 && J != rJ) {
        if (k & 1) {
            /* odd k: no k-mer is its own reverse complement, the junction is not palindromic (J != rJ): us_eligible only rules out the self
             * loop a+J == J+b, i.e. J a run of one nucleotide entered and left by that nucleotide */
            const uint32_t a = (uint32_t)ctz4(m >> 4), b = (uint32_t)ctz4(m & 15u);
            const uint64_t run = (0x5555555555555555ULL & kmask(k - 1)) * a; /* k - 1 times the nucleotide a */
            if (!(a == b && J == run)) { acc.c[JT_C_ORIENTED] += 2; acc.c[JT_C_INTERIOR] += 2; return; }
        } else {
            Kmer p, y;
            if (jt_view_interior(jt_view(m, true), J, k, p, y)) { acc.c[JT_C_ORIENTED] += 2; acc.c[JT_C_INTERIOR] += 2; return; }
        }
    }
    const int nviews = (J == rJ) ? 1 : 2;
    for (int w = 0; w < nviews; w++) {
        const uint64_t jf = w ? rJ : J;
        const JView v = jt_view(m, w == 0);
        const int n_out = popc4(v.out), n_in = popc4(v.in);
        acc.c[JT_C_ORIENTED] += (unsigned)n_out;
        if (n_in != 1) acc.c[JT_C_IN_NOT1] += (unsigned)n_out;
        Kmer p, y;
        if (jt_view_interior(v, jf, k, p, y)) { acc.c[JT_C_INTERIOR]++; continue; }
        for (uint32_t rest = v.out; rest; rest &= rest - 1) {
            const uint32_t b = (uint32_t)ctz4(rest);
            const Kmer x = make_kmer(((jf << 2) | b) & mk, k);
            const bool self = x.f == x.r;
            const JView r = jt_right(jt, x, mk1, lines);
            const int out_x = popc4(r.out);
            acc.c[JT_C_SELF] += self;
            if (n_in != 1 && out_x != 1) acc.c[JT_C_BOTH_NOT1]++;
            if (self && (n_in != 1 || out_x != 1)) acc.c[JT_C_SELF_BRANCH]++;
            bool chain = false;
            if (jt_simple(r)) chain = us_eligible(x, kmer_next(x, (uint32_t)ctz4(r.out), k, mk), k);
            if (chain) put_start(x.f);
            else if (x.f <= x.r) {
                const unsigned long long at = atomic_add64(&counters[JT_C_LEFT], 1ull);
                if (left_k && at < cap_left) {
                    const uint32_t a = src(x.f, lines);
                    left_k[at] = x.f;
                    left_a[at] = ab_stored(a);
                    if (a > 255u) atomic_add64(&counters[JT_C_SAT], 1ull);
                }
            }
        }
    }
}
/* the scan, one BUCKET at a time (the device: one lane per bucket, two 16-byte reads): the keys of its slots from one division
 * (bucket_first_h), then jt_scan_entry.  With lists large enough the ONE pass leaves statistics, chain starts and the k-mers of no chain; the
 * cursors count past the capacities, so a caller whose guess was too small learns the exact sizes and scans again. */
MTG_DEV void jt_bucket_words(const Table& jt, uint64_t b, uint64_t* q)
{
    const uint64_t* p = jt.slots + b * MTG_ABND_SLOTS;
MTG_UNROLL
    for (int i = 0; i < MTG_ABND_SLOTS / 2; i++) { const U64x2 v = ld_table(reinterpret_cast<const U64x2*>(p) + i); q[2 * i] = v.x; q[2 * i + 1] = v.y; }
}
template <typename Src, typename StartSink>
MTG_DEV void jt_scan_bucket_words(const Table& jt, int k, uint64_t b, const uint64_t* q, const Src& src, JtAcc& acc, unsigned long long* counters,
                                  const StartSink& put_start, uint64_t* left_k, uint32_t* left_a, unsigned long long cap_left, uint32_t& lines);
template <typename Src>
MTG_DEV void jt_scan_bucket(const Table& jt, int k, uint64_t b, const Src& src, JtAcc& acc, unsigned long long* counters,
                            uint64_t* starts, unsigned long long cap_starts, uint64_t* left_k, uint32_t* left_a, unsigned long long cap_left, uint32_t& lines)
{
    uint64_t q[MTG_ABND_SLOTS];
    jt_bucket_words(jt, b, q);
    const ScanStartsGlobal put_start{counters, starts, cap_starts};
    jt_scan_bucket_words(jt, k, b, q, src, acc, counters, put_start, left_k, left_a, cap_left, lines);
}
/* the same with the bucket's words in hand (the device reads the buckets of several turns before it looks at the first: k_jt_scan) */
template <typename Src, typename StartSink>
MTG_DEV void jt_scan_bucket_words(const Table& jt, int k, uint64_t b, const uint64_t* q, const Src& src, JtAcc& acc, unsigned long long* counters,
                                  const StartSink& put_start, uint64_t* left_k, uint32_t* left_a, unsigned long long cap_left, uint32_t& lines)
{
    bool any = false;
MTG_UNROLL
    for (int i = 0; i < MTG_ABND_SLOTS; i++) any = any || q[i] != 0;
    if (!any) return;
    const bool by_flag = (k & 1) != 0;
    uint64_t first = 0;
    bool have_first = false;
    uint32_t n_fast = 0;
    for (int i = 0; i < MTG_ABND_SLOTS; i++) {
        if (q[i] == 0) continue;
        const uint32_t bits = (uint32_t)q[i] & 255u;
        /* odd k: an entry without the flag and with one bit on each side is a chain interior (two oriented k-mers, two interior views): nothing to probe,
         * and its key is not needed (jt_special) */
        const uint32_t lo = bits & 15u, hi = bits >> 4;
        if (by_flag && !(q[i] & JT_MARK) && lo != 0u && hi != 0u && (lo & (lo - 1u)) == 0u && (hi & (hi - 1u)) == 0u) {
#ifdef MTG_XCHECK /* TEST-ONLY: the key says the same */
            {
                uint64_t Jx;
                (void)slot_key_in_bucket(jt, b, bucket_first_h(b, jt.nbuckets, jt.key_bits), q[i], Jx);
                if (jt_special(Jx, revcomp(Jx, k - 1), jt.key_bits)) abort();
            }
#endif
            n_fast++;
            continue;
        }
        if (!have_first) { first = bucket_first_h(b, jt.nbuckets, jt.key_bits); have_first = true; }
        /* (the flag stays: a second pass of the scan needs it again, a flagged junction is no chain interior, so no walker crosses it, and a walker that
         * met the bit without an entry in the side table would only walk on) */
        uint64_t J;
        const uint32_t m = slot_key_in_bucket(jt, b, first, q[i] & ~JT_MARK, J);
        if (m) jt_scan_entry(jt, k, J, m, src, acc, counters, put_start, left_k, left_a, cap_left, lines);
    }
    acc.c[JT_C_ORIENTED] += 2u * n_fast; acc.c[JT_C_INTERIOR] += 2u * n_fast;
}
/* the chain that starts with x, walked on the junction table (us_walk without lookaheads) */
template <typename Sink> MTG_DEV uint32_t jt_walk(const Table& jt, int k, const Kmer& x, Kmer& end, uint32_t& lines, Sink sink)
{
    const uint64_t mk = kmask(k), mk1 = kmask(k - 1);
    const uint32_t cap = MTG_US_MAX_LEN - (uint32_t)k; /* k-mers */
    Kmer cur = x;
    uint32_t n = 1;
    for (;;) {
        const JView a = jt_right(jt, cur, mk1, lines);
        if (!jt_simple(a)) break;
        const uint32_t nt = (uint32_t)ctz4(a.out);
        const Kmer y = kmer_next(cur, nt, k, mk);
        if (!us_eligible(cur, y, k) || n >= cap) break;
        cur = y;
        n++;
        sink(nt);
    }
    end = cur;
    return n;
}
/* ---- plan and emit in ONE walk (round 5).  Round 4 walked every chain three times: from both ends to learn length and other end (the
 * loser's walk thrown away), and a third time to write the sequence once its place in the store was known.  Now the walk that learns the
 * length writes the sequence as it goes -- into CHUNKS of a pool (32 words: a link to the next chunk, 31 words of 2-bit sequence), a new chunk
 * from a bump cursor whenever one is full -- and the walker that turns out to own the chain (its start k-mer the canonically smaller end)
 * reserves the words of the store and leaves, next to its record, its first chunk; us_compact then copies chunk chains to their places, a
 * streaming pass.  The random reads of the construction fall from three per junction to two; the chunks cost 8-byte writes of words that are
 * complete when they are written. */
enum { MTG_CHUNK_WORDS = 32, MTG_CHUNK_PAYLOAD = 31 };
struct ChunkPool {
    uint64_t* words;           /* cap_chunks * MTG_CHUNK_WORDS; word 0 of a chunk: 1 + the next chunk of its chain, 0 at the end */
    unsigned long long* cursor; /* chunks handed out (counts past the capacity: the caller checks) */
    uint64_t cap_chunks;
};
/* chunks the walks of n_starts starts over `steps` k-mers in all can need: k - 1 + steps nucleotides per start, rounded up twice */
MTG_HD uint64_t chunk_pool_need(uint64_t steps, uint64_t n_starts, int k)
{
    const uint64_t words = (steps + n_starts * (uint64_t)(k - 1)) / 32 + 2 * n_starts;
    return words / MTG_CHUNK_PAYLOAD + 2 * n_starts + 16;
}
struct ChunkWriter {
    ChunkPool pool;
    uint64_t first, cur; /* chunk numbers; ~0: the pool ran out (the walk goes on, nothing is written) */
    uint32_t wpos;       /* next payload word of cur (1 .. MTG_CHUNK_WORDS) */
    uint64_t acc;
    uint32_t nacc;
    MTG_DEV uint64_t take()
    {
        const uint64_t c = atomic_add64(pool.cursor, 1ull);
        if (c >= pool.cap_chunks) return ~0ull;
        pool.words[c * MTG_CHUNK_WORDS] = 0;
        return c;
    }
    MTG_DEV void begin(const ChunkPool& p) { pool = p; first = cur = take(); wpos = 1; acc = 0; nacc = 0; }
    MTG_DEV void word(uint64_t w)
    {
        if (cur == ~0ull) return;
        if (wpos == MTG_CHUNK_WORDS) {
            const uint64_t nx = take();
            if (nx == ~0ull) { cur = ~0ull; return; }
            pool.words[cur * MTG_CHUNK_WORDS] = nx + 1;
            cur = nx;
            wpos = 1;
        }
        pool.words[cur * MTG_CHUNK_WORDS + wpos++] = w;
    }
    MTG_DEV void nt(uint32_t c)
    {
        acc |= (uint64_t)c << (2 * nacc);
        if (++nacc == 32) { word(acc); acc = 0; nacc = 0; }
    }
    MTG_DEV void end() { if (nacc) word(acc); }
    MTG_DEV bool ok() const { return cur != ~0ull; }
};
/* ---- the two walks of a chain MEET IN THE MIDDLE (round 5).  A chain has a start at each end, and each start's walker needs the chain's
 * length and its other end to know whether it owns the chain; walking the whole chain from both ends (rounds 1-4, and the first form of this
 * round) reads every junction twice.  Now a walker leaves a MARK at every JT_MARK_EVERY-th junction it crosses -- the junction's key with
 * (walker, crossings so far) in a side table, then bit JT_MARK of the junction's slot -- and a walker
 * that is about to cross a marked junction looks the mark up: it is the other walker's (a junction lies on one chain, and a walker never
 * comes back to its own), and crossings of the one plus crossings of the other are the chain's junctions: the length is known, the partner
 * is known (its start k-mer is the chain's other end), the walker stops.  Its sequence so far and the partner's, reverse-complemented,
 * are the unitig (us_compact: both are the same sequence where they overlap, so every nucleotide is simply OR-ed into place).  The two
 * walkers of a chain read each junction once between them, plus at most JT_MARK_EVERY on either side of where they meet; a walker whose
 * partner has not started yet (or sits in a later wave) walks the whole chain and the partner meets its first mark after a few steps.
 * Nothing waits for anything: a mark that is not seen (a stale line, a side-table entry whose value is still on its way) only lets the
 * walker go on to the next one. */
enum { JT_MARK_EVERY = 32 };
struct MarkTab {
    uint64_t* keys; /* canonical junction; ~0: free */
    uint64_t* vals; /* walker << 32 | crossings; ~0: not written yet */
    uint64_t mask;  /* capacity - 1 (a power of two) */
};
MTG_DEV uint64_t marks_load(const uint64_t* p)
{
#ifdef MTG_EMU
    return __atomic_load_n(p, __ATOMIC_ACQUIRE);
#else
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); /* past the CU's L1: another CU wrote it */
#endif
}
/* 0: (key -> val) is in the table now; 1: key was there, its value in `other`; 2: there but its value not visible yet, or the table is full */
MTG_DEV int marks_insert(const MarkTab& m, uint64_t key, uint64_t val, uint64_t& other)
{
    uint64_t h = mix64(key) & m.mask;
    for (uint32_t probe = 0; probe < 64; probe++) {
        uint64_t cur = marks_load(m.keys + h);
        if (cur == ~0ULL) cur = atomic_cas64(m.keys + h, ~0ULL, key);
        if (cur == ~0ULL) {
#ifdef MTG_EMU
            __atomic_store_n(m.vals + h, val, __ATOMIC_RELEASE);
#else
            __hip_atomic_store(m.vals + h, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
            return 0;
        }
        if (cur == key) { other = marks_load(m.vals + h); return other == ~0ULL ? 2 : 1; }
        h = (h + 1) & m.mask;
    }
    return 2;
}
MTG_DEV bool marks_find(const MarkTab& m, uint64_t key, uint64_t& val)
{
    uint64_t h = mix64(key) & m.mask;
    for (uint32_t probe = 0; probe < 64; probe++) {
        const uint64_t cur = marks_load(m.keys + h);
        if (cur == ~0ULL) return false;
        if (cur == key) { val = marks_load(m.vals + h); return val != ~0ULL; }
        h = (h + 1) & m.mask;
    }
    return false;
}
/* a set of 62-bit keys (canonical k-mers, canonical junctions) in open addressing: a slot holds key + 1, 0 = free.  Filled and read by different kernels. */
struct KeySet {
    uint64_t* keys;
    uint64_t mask; /* capacity - 1 (a power of two) */
};
/* 0: the key is in the set now and was not before; 1: it was there; 2: no free slot within reach (the set is far too full: its maker gives up on it) */
MTG_DEV int keyset_put(const KeySet& s, uint64_t key)
{
    uint64_t h = mix64(key) & s.mask;
    for (uint32_t probe = 0; probe < 512; probe++) {
        uint64_t cur = marks_load(s.keys + h);
        if (cur == 0) cur = atomic_cas64(s.keys + h, 0, key + 1);
        if (cur == 0) return 0;
        if (cur == key + 1) return 1;
        h = (h + 1) & s.mask;
    }
    return 2;
}
/* true: the key is in the set now and was not before */
MTG_DEV bool keyset_insert(const KeySet& s, uint64_t key) { return keyset_put(s, key) == 0; }
MTG_DEV bool keyset_has(const KeySet& s, uint64_t key)
{
    uint64_t h = mix64(key) & s.mask;
    for (uint32_t probe = 0; probe < 512; probe++) {
        const uint64_t cur = s.keys[h];
        if (cur == 0) return false;
        if (cur == key + 1) return true;
        h = (h + 1) & s.mask;
    }
    return true; /* (a set this full is not built; "perhaps" is the safe answer for both users) */
}
/* a record whose unitig was found by POSITION in a packed sequence (mtg_gpu_build.hip: k_pos_plan): no walker holds its sequence */
static const uint32_t REC_BY_POSITION = 0xFFFFFFFEu;
/* what the walkers of a launch share */
struct WalkShared {
    Table jt;
    int k;
    ChunkPool pool;
    MarkTab marks;
    const uint64_t* starts;       /* start k-mers (forward values), one per walker */
    unsigned long long* counters; /* JT_C_* */
    UsRec* rec;
    uint64_t* rec_walk;           /* per record: owner | partner << 32 (partner 0xFFFFFFFF: the owner walked the whole chain) */
    uint64_t rec_cap;
    uint64_t* w_chunk;            /* per walker: first chunk of its sequence (~0: the pool ran out) */
    uint32_t* w_cnt;              /* per walker: k-mers it holds */
    KeySet done;                  /* keys == nullptr: none.  Canonical end k-mers of the chains that are stored already (found by position): their walkers leave at once */
};
/* one walker.  begin(), then step() until it returns false (the device: a loop in the lane; the TEST-ONLY emulation steps all walkers of a
 * graph in turns, so that they do meet in the middle) */
struct JtWalker {
    uint32_t self, n, lines;
    Kmer cur;
    uint64_t start_f;
    ChunkWriter W;
    bool done;
    MTG_DEV void begin(const WalkShared& S, uint32_t i)
    {
        self = i; n = 1; lines = 0; done = false;
        start_f = S.starts[i];
        cur = make_kmer(start_f, S.k);
        if (S.done.keys && keyset_has(S.done, canon(cur))) { done = true; S.w_chunk[self] = ~0ULL; S.w_cnt[self] = 0; W.pool = S.pool; W.first = W.cur = ~0ULL; W.wpos = 1; W.acc = 0; W.nacc = 0; return; }
        W.begin(S.pool);
        for (int j = S.k - 1; j >= 0; j--) W.nt((uint32_t)(start_f >> (2 * j)) & 3u);
    }
    /* the walk is over: total = k-mers of the chain, partner = the walker of its other end (0xFFFFFFFF: this one reached it), end_c = canonical k-mer of the other end */
    MTG_DEV void finish(const WalkShared& S, uint32_t total, uint32_t partner, uint64_t end_c)
    {
        done = true;
        W.end();
        S.w_chunk[self] = W.ok() ? W.first : ~0ULL;
        S.w_cnt[self] = n;
        const Kmer x = make_kmer(start_f, S.k);
        if (total < 2 || total >= MTG_US_MAX_LEN - (uint32_t)S.k) return;
        if (!(canon(x) < end_c)) return; /* the chain is stored from the end with the smaller canonical k-mer */
        const uint64_t r = atomic_add64(&S.counters[JT_C_RECS], 1ull);
        const uint64_t w = atomic_add64(&S.counters[JT_C_WORDS], (unsigned long long)us_words_of(total, S.k));
        atomic_add64(&S.counters[JT_C_STORED_VIEWS], 2ull * (total - 1));
        if (r < S.rec_cap) { S.rec[r].start_f = start_f; S.rec[r].len_k = total; S.rec[r].pad_ = 0; S.rec[r].hdr = w; S.rec_walk[r] = (uint64_t)self | ((uint64_t)partner << 32); }
    }
    MTG_DEV bool step(const WalkShared& S)
    {
        if (done) return false;
        const int k = S.k;
        const uint64_t mk = kmask(k), mk1 = kmask(k - 1);
        const uint64_t s = cur.f & mk1, rs = cur.r >> 2;
        const bool fwd = s <= rs;
        const uint64_t J = fwd ? s : rs;
        uint64_t* slot;
        bool marked;
        const uint64_t v = jt_lookup(S.jt, J, lines, slot, marked);
        const JView a = jt_view((uint32_t)(v & 255u), fwd);
        const uint32_t cap = MTG_US_MAX_LEN - (uint32_t)k;
        if (!jt_simple(a)) { finish(S, n, 0xFFFFFFFFu, canon(cur)); return false; }
        const uint32_t nt = (uint32_t)ctz4(a.out);
        const Kmer y = kmer_next(cur, nt, k, mk);
        if (!us_eligible(cur, y, k) || n >= cap) { finish(S, n, 0xFFFFFFFFu, canon(cur)); return false; }
        if (marked) {
            /* the other walker crossed this junction as its u-th: n junctions on this side of it (it included), u on the other */
            uint64_t mv;
            if (marks_find(S.marks, J, mv) && (uint32_t)(mv >> 32) != self) {
                const uint32_t partner = (uint32_t)(mv >> 32), u = (uint32_t)mv;
                finish(S, n + u, partner, canon(make_kmer(S.starts[partner], k)));
                return false;
            }
        }
        cur = y;
        n++;
        W.nt(nt);
        const uint32_t c = n - 1; /* junctions crossed */
        if (c % JT_MARK_EVERY == 0 && slot) {
            uint64_t other = 0;
            const int r = marks_insert(S.marks, J, ((uint64_t)self << 32) | c, other);
            if (r == 0) atomic_or64(slot, JT_MARK); /* no fence between the entry and the mark (an agent-scope release writes the L2 back: 94 M of them doubled the kernel's time):
                                                        a reader that sees the mark before the entry's value, or before its key, walks on to the next mark */ else if (r == 1 && (uint32_t)(other >> 32) != self) {
                /* the other walker marked this very junction (as its u-th): c crossings on this side, it included */
                const uint32_t partner = (uint32_t)(other >> 32), u = (uint32_t)other;
                finish(S, c + u, partner, canon(make_kmer(S.starts[partner], k)));
                return false;
            }
        }
        return true;
    }
};
/* OR n nucleotides (2-bit fields of v, lowest first) into the sequence words w at nucleotide p0 (the words are zero where nothing was written) */
MTG_DEV void seq_or(uint64_t* w, uint64_t p0, uint64_t v, uint32_t n)
{
    if (n == 0) return;
    const uint32_t sh = 2u * (uint32_t)(p0 & 31u);
    atomic_or64(w + (p0 >> 5), v << sh);
    if (sh && (p0 & 31u) + n > 32u) atomic_or64(w + (p0 >> 5) + 1, v >> (64u - sh));
}
/* word j of a walker's chunk chain needs the chain walked: the callers below go through a chain once, chunk by chunk */
/* per record: header word and sequence of the unitig into the store, from the owner's chunks (its nucleotides at their places) and, when the
 * walks met, from the partner's (reverse-complemented, from the other end).  lane / nlanes: the lanes of a group share the words of every
 * chunk (the device: a wave per unitig; the emulation: one lane).  false: a chain of chunks is shorter than its walker's count says. */
MTG_DEV bool us_compact(const UStore& us, int k, const UsRec& r, uint64_t walk, const WalkShared& S, uint32_t lane, uint32_t nlanes)
{
    uint64_t* w = us.words + r.hdr;
    const uint64_t L = (uint64_t)r.len_k + (uint32_t)k - 1; /* nucleotides */
    const uint32_t owner = (uint32_t)walk, partner = (uint32_t)(walk >> 32);
    if (owner == REC_BY_POSITION) return true; /* its sequence comes straight from the packed input (k_us_from_seq) */
    if (lane == 0) w[0] = L;
    for (int side = 0; side < 2; side++) {
        const uint32_t who = side ? partner : owner;
        if (who == 0xFFFFFFFFu) continue;
        uint64_t have = (uint64_t)S.w_cnt[who] + (uint32_t)k - 1; /* nucleotides of this walker's sequence */
        if (have > L) have = L;
        const uint64_t nw = (have + 31) / 32;
        uint64_t c = S.w_chunk[who], done = 0;
        while (done < nw) {
            if (c == ~0ULL || c >= S.pool.cap_chunks) return false;
            const uint64_t* src = S.pool.words + c * MTG_CHUNK_WORDS;
            const uint64_t take = nw - done < MTG_CHUNK_PAYLOAD ? nw - done : MTG_CHUNK_PAYLOAD;
            for (uint64_t t = lane; t < take; t += nlanes) {
                const uint64_t wq = done + t;                                             /* word of the walker's sequence */
                const uint32_t cnt = (uint32_t)(have - 32 * wq < 32 ? have - 32 * wq : 32); /* nucleotides in it */
                uint64_t v = src[1 + t];
                if (cnt < 32) v &= (1ull << (2 * cnt)) - 1ull;
                if (!side) seq_or(w + 1, 32 * wq, v, cnt);
                else {
                    /* the partner's nucleotide q sits at L - 1 - q, complemented: the word's nucleotides in reverse order */
                    uint64_t rv = rev_fields64(v) >> (2u * (32u - cnt));
                    rv ^= cnt < 32 ? (0xAAAAAAAAAAAAAAAAULL & ((1ull << (2 * cnt)) - 1ull)) : 0xAAAAAAAAAAAAAAAAULL;
                    seq_or(w + 1, L - 32 * wq - cnt, rv, cnt);
                }
            }
            done += take;
            const uint64_t link = src[0];
            c = link ? link - 1 : ~0ULL;
        }
    }
    return true;
}
/* per k-mer i of a stored unitig: its abundance, asked of the source, into the store; returns 1 when it exceeds 255 */
template <typename Src> MTG_DEV uint32_t us_ab_fill(const UStore& us, int k, const UsRec& r, uint32_t i, const Src& src, uint32_t& lines)
{
    const uint64_t mk = kmask(k);
    const uint64_t le = us_kmer_le(us.words, (r.hdr + 1) * 32 + i, k);
    const uint64_t xr = le ^ (0xAAAAAAAAAAAAAAAAULL & mk), xf = revcomp(xr, k);
    const uint32_t a = src(xf < xr ? xf : xr, lines);
    us.ab[(r.hdr + 1) * 32 + i] = (uint8_t)ab_stored(a);
    return a > 255u;
}
/* a closed or over-long chain was met (rare): every canonical k-mer of the junction table that the finished unitig pointers of `nx` (the
 * sparse ADJ under construction, store attached) do not reach is a k-mer of no unitig.  Per entry, as the scan. */
template <typename Src>
MTG_DEV void jt_unstored_entry(const Table& jt, const Index& nx, uint64_t J, uint32_t m, const Src& src, unsigned long long* counters,
                               uint64_t* left_k, uint32_t* left_a, unsigned long long cap_left, uint32_t& lines)
{
    const int k = nx.k;
    const uint64_t mk = kmask(k);
    const uint64_t rJ = revcomp(J, k - 1);
    const int nviews = (J == rJ) ? 1 : 2;
    for (int w = 0; w < nviews; w++) {
        const uint64_t jf = w ? rJ : J;
        const JView v = jt_view(m, w == 0);
        for (uint32_t rest = v.out; rest; rest &= rest - 1) {
            const Kmer x = make_kmer(((jf << 2) | (uint32_t)ctz4(rest)) & mk, k);
            if (!(x.f <= x.r) || kmer_stored(nx, x.f, lines)) continue;
#ifdef MTG_EMU
            const unsigned long long at = __sync_fetch_and_add(&counters[JT_C_LEFT], 1ull);
#else
            const unsigned long long at = atomicAdd(&counters[JT_C_LEFT], 1ull);
#endif
            if (left_k && at < cap_left) {
                const uint32_t a = src(x.f, lines);
                left_k[at] = x.f;
                left_a[at] = ab_stored(a);
                if (a > 255u) {
#ifdef MTG_EMU
                    __sync_fetch_and_add(&counters[JT_C_SAT], 1ull);
#else
                    atomicAdd(&counters[JT_C_SAT], 1ull);
#endif
                }
            }
        }
    }
}

} // namespace mtg
#endif
