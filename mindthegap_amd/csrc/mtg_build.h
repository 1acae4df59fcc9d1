/*
 * mtg_build.h -- the lean index construction (Graph::create, /root/reference/src/Filler.cpp:172-226) as device functions: the graph as a
 * junction table, chain starts / statistics from one streaming pass, the unitig store written by the walks that measure the chains.
 * Included by mtg_gpu_build.hip (the kernels) and by the test-only emulation (tests/emu/emu_us.h), which runs the same functions serially.
 * Kept out of mtg_dev.h so that a change here does not recompile the fill kernels.
 */
#ifndef MTG_BUILD_H
#define MTG_BUILD_H
#include "mtg_dev.h"

namespace mtg {

/* The same for a whole bucket, without the search: the hashes H that bucket_of() sends to bucket `home` form the interval that begins at
 * first = ceil(home * 2^key_bits / nbuckets) and is at most 2^tag_bits long (nbuckets >= 2^(key_bits - tag_bits)), so the H of a slot is the
 * one value with the slot's tag in its low bits at or after `first`.  One division per bucket instead of six 128-bit products per slot: what
 * the streaming pass over the junction table pays (k_jt_scan). */
MTG_DEV uint64_t bucket_first_h(uint64_t home, uint64_t nb, uint32_t key_bits)
{
#ifdef MTG_EMU
    return (uint64_t)(((((unsigned __int128)home) << key_bits) + nb - 1) / nb);
#else
    if (home == 0) return 0;
    /* an estimate in double (off by at most a few thousand), corrected exactly: home * 2^key_bits - e * nb is small, so its low 64 bits are it */
    const uint64_t e = (uint64_t)((double)home * ((double)(1ull << key_bits) / (double)nb));
    const int64_t diff = (int64_t)((home << key_bits) - e * nb);
    const int64_t q = diff >= 0 ? (int64_t)(((uint64_t)diff + nb - 1) / nb) : -(int64_t)((uint64_t)(-diff) / nb);
    return (uint64_t)((int64_t)e + q);
#endif
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
MTG_DEV JView jt_right(const Table& jt, const Kmer& x, uint64_t mk1, uint32_t& lines)
{
    const uint64_t s = x.f & mk1, rs = x.r >> 2;
    return jt_view(table_get<MTG_ABND_SLOTS>(jt, s <= rs ? s : rs, lines), s <= rs);
}
MTG_DEV JView jt_left(const Table& jt, const Kmer& x, uint64_t mk1, uint32_t& lines)
{
    const uint64_t p = x.f >> 2, rp = x.r & mk1;
    return jt_view(table_get<MTG_ABND_SLOTS>(jt, p <= rp ? p : rp, lines), p <= rp);
}
/* table_or with the bucket read in one go before any atomic (the entry usually exists with its bits, or the first free slot takes it).
 * A stale read is harmless: the tag part of a slot is written once, a slot seen empty is claimed by compare-and-swap, bits seen missing
 * are OR-ed in again.  Returns as table_or. */
MTG_DEV int jt_or(const Table& t, uint64_t key, uint32_t bits)
{
    const uint64_t H = mix(key, t.key_bits);
    uint64_t b = bucket_of(H, t.nbuckets, t.key_bits);
    const uint64_t tag = H & ((1ULL << t.tag_bits) - 1);
    for (uint32_t d = 0; d <= MTG_MAX_DISP; d++) {
        const uint64_t want = (tag << MTG_DISP_BITS) | d;
        uint64_t* p = t.slots + b * MTG_ABND_SLOTS;
        uint64_t q[MTG_ABND_SLOTS];
MTG_UNROLL
        for (int i = 0; i < MTG_ABND_SLOTS / 2; i++) { const U64x2 v = ld_table(reinterpret_cast<const U64x2*>(p) + i); q[2 * i] = v.x; q[2 * i + 1] = v.y; }
        for (int i = 0; i < MTG_ABND_SLOTS; i++) {
            uint64_t v = q[i];
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
/* one occurrence of the junction J ((k-1)-mer jf, its reverse complement jr) with the nucleotide before it (a, when has_a: the k-mer a+J is
 * solid) and behind it (b, when has_b: J+b is solid): what index_insert contributes to J's entry from the k-mers on its two sides (a
 * palindromic junction gets the bits of both strands).  Returns as table_or. */
MTG_DEV int jt_insert_junction(const Table& jt, uint64_t jf, uint64_t jr, bool has_a, uint32_t a, bool has_b, uint32_t b)
{
    uint32_t bits = 0;
    if (jf <= jr) bits |= (has_b ? 1u << b : 0u) | (has_a ? 1u << (4 + a) : 0u);
    if (jr <= jf) bits |= (has_b ? 1u << (4 + (b ^ 2u)) : 0u) | (has_a ? 1u << (a ^ 2u) : 0u);
    if (!bits) return 0;
    return jt_or(jt, jf <= jr ? jf : jr, bits);
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
MTG_DEV uint32_t jt_slot_key(const Table& t, uint64_t slot, uint64_t& key) { return abnd_slot_kmer(t, slot, key); }
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
template <typename Src>
MTG_DEV void jt_scan_entry(const Table& jt, int k, uint64_t J, uint32_t m, const Src& src, JtAcc& acc, unsigned long long* counters,
                           uint64_t* starts, unsigned long long cap_starts, uint64_t* left_k, uint32_t* left_a, unsigned long long cap_left, uint32_t& lines)
{
    const uint64_t mk = kmask(k), mk1 = kmask(k - 1);
    const uint64_t rJ = revcomp(J, k - 1);
    /* the entry of nearly every junction of a genome: one bit on each side.  Both views are then simple, and eligibility is a property of the
     * canonical junction (us_eligible is symmetric under reverse complement): two oriented k-mers, two interior views, nothing to probe */
    if (popc4(m & 15u) == 1 && popc4(m >> 4) == 1 && J != rJ) {
        Kmer p, y;
        if (jt_view_interior(jt_view(m, true), J, k, p, y)) { acc.c[JT_C_ORIENTED] += 2; acc.c[JT_C_INTERIOR] += 2; return; }
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
            if (chain) {
                const unsigned long long at = atomic_add64(&counters[JT_C_STARTS], 1ull);
                if (starts && at < cap_starts) starts[at] = x.f;
            } else if (x.f <= x.r) {
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
template <typename Src>
MTG_DEV void jt_scan_bucket(const Table& jt, int k, uint64_t b, const Src& src, JtAcc& acc, unsigned long long* counters,
                            uint64_t* starts, unsigned long long cap_starts, uint64_t* left_k, uint32_t* left_a, unsigned long long cap_left, uint32_t& lines)
{
    uint64_t q[MTG_ABND_SLOTS];
    const uint64_t* p = jt.slots + b * MTG_ABND_SLOTS;
MTG_UNROLL
    for (int i = 0; i < MTG_ABND_SLOTS / 2; i++) { const U64x2 v = ld_table(reinterpret_cast<const U64x2*>(p) + i); q[2 * i] = v.x; q[2 * i + 1] = v.y; }
    bool any = false;
MTG_UNROLL
    for (int i = 0; i < MTG_ABND_SLOTS; i++) any = any || q[i] != 0;
    if (!any) return;
    const uint64_t first = bucket_first_h(b, jt.nbuckets, jt.key_bits);
    for (int i = 0; i < MTG_ABND_SLOTS; i++) {
        if (q[i] == 0) continue;
        uint64_t J;
        const uint32_t m = slot_key_in_bucket(jt, b, first, q[i], J);
        if (m) jt_scan_entry(jt, k, J, m, src, acc, counters, starts, cap_starts, left_k, left_a, cap_left, lines);
    }
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
/* per chain start: the walk to the other end, the sequence into chunks on the way; the end the chain is stored from reserves words and record
 * and notes its first chunk (rec_chunk[r]).  counters as jt_plan_start. */
MTG_DEV void jt_plan_emit_start(const Table& jt, int k, const Kmer& x, const ChunkPool& pool, unsigned long long* counters, UsRec* rec, uint64_t* rec_chunk, uint64_t rec_cap, uint32_t& lines)
{
    ChunkWriter W;
    W.begin(pool);
    for (int i = k - 1; i >= 0; i--) W.nt((uint32_t)(x.f >> (2 * i)) & 3u);
    Kmer end;
    const uint32_t n = jt_walk(jt, k, x, end, lines, [&](uint32_t c) { W.nt(c); });
    W.end();
    if (n < 2) return;
    if (n >= MTG_US_MAX_LEN - (uint32_t)k) return;
    if (!(canon(x) < canon(end))) return;
    const uint64_t r = atomic_add64(&counters[JT_C_RECS], 1ull);
    const uint64_t w = atomic_add64(&counters[JT_C_WORDS], (unsigned long long)us_words_of(n, k));
    atomic_add64(&counters[JT_C_STORED_VIEWS], 2ull * (n - 1));
    if (r < rec_cap) { rec[r].start_f = x.f; rec[r].len_k = n; rec[r].pad_ = 0; rec[r].hdr = w; rec_chunk[r] = W.ok() ? W.first : ~0ull; }
}
/* per record: header word and sequence words of the unitig from its chunk chain into the store.  lane / nlanes: the lanes of a group share
 * the words of every chunk (the device: a wave per unitig; the emulation: one lane).  Returns false when the chain is shorter than the record
 * says (the pool ran out during the walk). */
MTG_DEV bool us_compact(const UStore& us, int k, const UsRec& r, uint64_t first_chunk, const ChunkPool& pool, uint32_t lane, uint32_t nlanes)
{
    uint64_t* w = us.words + r.hdr;
    const uint64_t nw = us_words_of(r.len_k, k) - 1; /* sequence words */
    if (lane == 0) w[0] = (uint64_t)r.len_k + (uint32_t)k - 1;
    uint64_t c = first_chunk, done = 0;
    while (done < nw) {
        if (c == ~0ull || c >= pool.cap_chunks) return false;
        const uint64_t* src = pool.words + c * MTG_CHUNK_WORDS;
        const uint64_t take = nw - done < MTG_CHUNK_PAYLOAD ? nw - done : MTG_CHUNK_PAYLOAD;
        for (uint64_t t = lane; t < take; t += nlanes) w[1 + done + t] = src[1 + t];
        done += take;
        const uint64_t link = src[0];
        c = link ? link - 1 : ~0ull;
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
