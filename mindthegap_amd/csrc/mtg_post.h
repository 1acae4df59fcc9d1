/*
 * mtg_post.h -- what follows the contig construction of a gap, on the device, one wave per gap:
 *   - Filler::find_nodes_containing_multiple_R (/root/reference/src/Filler.cpp:1294-1378) for every contig
 *   - the common case "contig 0 holds the target" (GraphAnalysis::find_all_paths_rev returns the single path [0],
 *     src/GraphAnalysis.cpp:222-226; paths_to_sequences then gives contig0[k:pos], :386-423): the coverage pass of
 *     src/Filler.cpp:959-988 (abundance of every k-mer of source+fill, sum and median) is done here too.
 * Compiled for gfx950 (lanes = the 64 lanes of a wave) and, TEST-ONLY, for tests/emu (one lane).
 */
#ifndef MTG_POST_H
#define MTG_POST_H
#include "mtg_traverse.h"

namespace mtg {

struct PostOut {
    uint32_t nb_terminal; /* contigs holding a target */
    uint32_t fast;        /* 0: general path on the host; 1: solution = contig0[k:pos]; 2: target on contig 0 but empty fill */
    uint32_t pos, errors, target; /* terminal info of contig 0 when fast != 0 */
    uint32_t clen0;
    uint32_t ab_sum, ab_n;        /* coverage pass (fast == 1) */
    uint32_t med_hi, med_lo;      /* sorted[n/2], sorted[n/2-1] */
    uint32_t lines;               /* index buckets read by the coverage pass (block look-ups + k-mers the store did not confirm) */
    uint32_t direct;              /* 1: the abundances were read at places known from the traversal's copy command (no look-up) */
    uint32_t lean;                /* 1 + index of the copy command: the contig was never materialised (LeanRec); coverage and ASCII come from the unitig store */
};

/* everything the device keeps about one gap of a launch: the counters of the traversal and of the post-processing, what the gap
 * contributes to the arrays of its batch (mtg_emit.h: emit_plan) and where (exclusive prefix sums in slot order).  Metadata of a gap: 5
 * runs of nc entries at meta[5 * cbase] (length, first word, terminal position, errors, target index). */
struct alignas(16) SlotRec {
    GapOut o;
    PostOut p;
    uint32_t nw, nc;   /* dense words / contig metadata entries (multi-contig gaps and the stage-A entry only) */
    uint32_t asc, ext; /* bytes in the sequence / extension arena, NUL included */
    uint64_t wbase, cbase, abase, ebase;
    uint32_t rpos, gpos; /* rank among the gaps to re-run / among the gaps that need the host */
    uint32_t fpos, pad_; /* rank among the filled gaps (the index of its mtg_wire_filled when the batch leaves in relocatable form) */
    uint32_t pad2_[2];   /* 160 bytes: whole 16-byte pieces (k_post stores the record with ten wide stores) */
};

/* A target as the host hands it over: its first k characters in a TARGET_SLOT-byte slot, byte TARGET_SLOT - 1 = 1 when the anchor
 * has at least k characters.  identNT (src/Utils.cpp:81-84) is case-insensitive equality: the k-mer is compared as 2-bit codes with a
 * forced mismatch where the anchor character is not a nucleotide; an anchor shorter than k can never be matched. */
enum { TARGET_SLOT = 32 };
MTG_DEV void encode_target(const uint8_t* slot, int k, uint64_t& le, uint64_t& bad)
{
    le = 0;
    bad = 0;
    if (!slot[TARGET_SLOT - 1]) { bad = ~0ull; }
    else
        for (int i = 0; i < k; i++) {
            const uint32_t c = slot[i], u = c & 0xDFu;
            le |= (uint64_t)((c >> 1) & 3u) << (2 * i);
            if (!(u == 'A' || u == 'C' || u == 'G' || u == 'T')) bad |= 1ull << (2 * i);
        }
    bad &= 0x5555555555555555ULL & kmask(k);
}

struct PostTargets {
    const uint64_t* le;  /* target k-mers, little-endian packed (nt i at bits 2i), dictionary iteration order */
    const uint64_t* bad; /* bit 2i set: position i can never match (not ACGT) or the whole anchor is unusable */
    uint32_t n;
    uint32_t nb_mis;
    uint32_t fast_ok;    /* the source sequence is exactly k valid nucleotides */
    /* the piece index of the batch's dictionaries (below); pi_head == nullptr: none */
    const uint32_t* pi_head = nullptr;
    const uint32_t* pi_next = nullptr;
    uint32_t pi_mask = 0;
    uint32_t gbase = 0;  /* the number of this gap's first target among the targets of the batch */
    uint32_t gid = 0;    /* the gap's number in the batch */
};

/* ---- The piece index (round 6): the terminal search without a pass over the dictionary per contig position.  A contig position matches a target
 * when at most nb_mis of the k nucleotides differ (find_nodes_containing_multiple_R, src/Filler.cpp:1341-1351); cut the k nucleotides into
 * nb_mis + 1 pieces and one piece at least is free of differences (and of the target's unusable characters, which count as differences).  So every
 * piece of every target of the batch goes into ONE chained hash table (key: gap, piece number, the piece's nucleotides; entry: target << 2 | piece),
 * and a contig position looks its nb_mis + 1 pieces up and counts the differences only against the targets it finds there.  Every pair (position,
 * target) within nb_mis is among them, pairs that are not are harmless (the count decides, as it does in the pass over all targets), the result is
 * the same arg-max.  Contig mode hands every seed the targets of all other contigs (src/Filler.cpp:522-533): at 10 000 contigs the pass made 20 000
 * counts per contig position, 24 ms per batch of 40 seeds and 99 % of the job's device time. */
enum { POST_PIECES_MAX = 4, POST_INDEX_MIN = 16, POST_INDEX_NIL = 0xFFFFFFFFu };
MTG_DEV bool post_index_usable(int k, uint32_t nb_mis) { return nb_mis + 1u <= (uint32_t)POST_PIECES_MAX && (uint32_t)k / (nb_mis + 1u) >= 8u; }
MTG_DEV uint32_t post_piece_begin(uint32_t p, uint32_t np, int k) { return p * (uint32_t)k / np; }
MTG_DEV uint32_t post_piece_slot(uint64_t piece, uint32_t p, uint32_t gid, uint32_t mask)
{
    return (uint32_t)(mix64(((piece << 2) | p) ^ ((uint64_t)gid * 0x9E3779B97F4A7C15ULL)) >> 20) & mask;
}
/* target number gt (of the batch) of gap gid: its pieces into the table (any thread) */
MTG_DEV void post_index_add(uint32_t* head, uint32_t* next, uint32_t mask, uint32_t gid, uint32_t gt, uint64_t le, uint64_t bad, uint32_t nb_mis, int k)
{
    if (!post_index_usable(k, nb_mis)) return;
    const uint32_t np = nb_mis + 1u;
    for (uint32_t p = 0; p < np; p++) {
        const uint32_t b = post_piece_begin(p, np, k), e = post_piece_begin(p + 1u, np, k);
        const uint64_t pm = (1ull << (2u * (e - b))) - 1ull;
        if ((bad >> (2u * b)) & pm) continue; /* an unusable character in the piece: it cannot be the piece without a difference */
        const uint32_t h = post_piece_slot((le >> (2u * b)) & pm, p, gid, mask), ent = (gt << 2) | p;
#ifdef MTG_EMU
        next[ent] = __atomic_exchange_n(&head[h], ent, __ATOMIC_RELAXED);
#else
        next[ent] = atomicExch(&head[h], ent);
#endif
    }
}

#ifdef MTG_EMU
#define MTG_LANE() 0u
#define MTG_NLANES 1u
MTG_DEV uint64_t wave_max64(uint64_t x) { return x; }
MTG_DEV uint32_t wave_sum32(uint32_t x) { return x; }
MTG_DEV void hist_add(uint32_t* h, uint32_t v) { h[v]++; }
MTG_DEV void wave_sync() {}
MTG_DEV bool wave_any(bool x) { return x; }
MTG_DEV int wave_first(bool x) { return x ? 0 : -1; } /* the lowest lane with x, -1: none */
MTG_DEV uint32_t wave_max32(uint32_t x) { return x; }
#else
#define MTG_LANE() (threadIdx.x & 63u)
#define MTG_NLANES 64u
MTG_DEV uint64_t wave_max64(uint64_t x)
{
    for (int m = 32; m >= 1; m >>= 1) {
        uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)x, m, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(x >> 32), m, 64);
        uint64_t y = ((uint64_t)hi << 32) | lo;
        x = y > x ? y : x;
    }
    return x;
}
MTG_DEV uint32_t wave_sum32(uint32_t x)
{
    for (int m = 32; m >= 1; m >>= 1) x += (uint32_t)__shfl_xor((int)x, m, 64);
    return x;
}
MTG_DEV void hist_add(uint32_t* h, uint32_t v) { atomicAdd(&h[v], 1u); }
MTG_DEV bool wave_any(bool x) { return __ballot(x) != 0ull; }
MTG_DEV int wave_first(bool x) { const unsigned long long m = __ballot(x); return m ? (int)__ffsll((long long)m) - 1 : -1; }
MTG_DEV uint32_t wave_max32(uint32_t x)
{
    for (int m = 32; m >= 1; m >>= 1) { const uint32_t y = (uint32_t)__shfl_xor((int)x, m, 64); x = y > x ? y : x; }
    return x;
}
MTG_DEV void wave_sync() { __syncthreads(); }
#endif

/* k-mer starting at nt j of a packed sequence, little-endian (nt j+i at bits 2i) */
MTG_DEV uint64_t le_kmer(const uint64_t* w, uint32_t j, uint64_t mk)
{
    const uint32_t s = 2u * (j & 31u);
    const uint64_t lo = w[j >> 5] >> s;
    const uint64_t hi = s ? (w[(j >> 5) + 1] << (64u - s)) : 0ull;
    return (lo | hi) & mk;
}

/* sorted[n / 2] (hi) and sorted[n / 2 - 1] (lo) of n values 0 .. 255 given as a histogram of 256 bins (the exact median of src/Utils.cpp:241-254 is hi
 * for odd n, their mean for even n); by the lanes of a wave after a wave_sync */
MTG_DEV void hist_median(const uint32_t* hist, uint32_t nk, uint32_t& hi, uint32_t& lo)
{
    const uint32_t n2 = nk / 2;
    hi = 0; lo = 0;
#ifdef MTG_EMU
    {
        uint32_t cum = 0;
        bool got_hi = false, got_lo = (n2 == 0);
        for (uint32_t v = 0; v < 256 && !(got_hi && got_lo); v++) {
            cum += hist[v];
            if (!got_lo && cum > n2 - 1) { lo = v; got_lo = true; }
            if (!got_hi && cum > n2) { hi = v; got_hi = true; }
        }
    }
#else
    {
        /* four bins per lane, an inclusive scan over the lanes, and the two lanes that hold the ranks n2 - 1 and n2 say which bins they fall into */
        const uint32_t lane = MTG_LANE();
        uint32_t c4[4], s4 = 0;
MTG_UNROLL
        for (int i = 0; i < 4; i++) { c4[i] = hist[4u * lane + (uint32_t)i]; s4 += c4[i]; }
        uint32_t incl = s4;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, d, 64); if ((int)lane >= d) incl += y; }
        uint32_t cum = incl - s4, f_hi = 0, f_lo = 0;
MTG_UNROLL
        for (int i = 0; i < 4; i++) {
            const uint32_t nxt = cum + c4[i];
            if (cum <= n2 && nxt > n2) f_hi = 4u * lane + (uint32_t)i + 1u;
            if (n2 > 0 && cum <= n2 - 1u && nxt > n2 - 1u) f_lo = 4u * lane + (uint32_t)i + 1u;
            cum = nxt;
        }
        f_hi = wave_max32(f_hi);
        f_lo = wave_max32(f_lo);
        hi = f_hi ? f_hi - 1u : 0u;
        lo = f_lo ? f_lo - 1u : 0u;
    }
#endif
}

/* The scans read a contig through a tile of its words staged in LDS: one round of wide, coalesced loads per tile instead of one
 * dependent round trip to memory per 64 positions (which is what bound this kernel: 47 round trips for a 3 kb contig). */
#ifndef MTG_POST_TILE
#define MTG_POST_TILE 512
#endif
enum { POST_TILE = MTG_POST_TILE }; /* words; 512 = 16384 nucleotides */

/* the terminal search of one contig through the piece index: this lane's best key (count << 40 | ORD - (position * n + target)) over the positions
 * lane, lane + 64, ...; the caller takes the wave's maximum.  The contig is read where it lies (a position costs nb_mis + 1 look-ups; its word is the
 * least of it). */
MTG_DEV_COLD uint64_t post_search_indexed(const PostTargets T, const uint64_t* w, uint32_t L, int k)
{
    const uint64_t mk = kmask(k), lsb = 0x5555555555555555ULL & mk, ORD = (1ull << 40) - 1;
    const uint32_t np = T.nb_mis + 1u, npos = L - (uint32_t)k + 1;
    uint64_t best = 0;
    for (uint32_t jb = 0; jb < npos; jb += MTG_NLANES) {
        const uint32_t j = jb + MTG_LANE();
        if (j < npos) {
            const uint64_t x = le_kmer(w, j, mk);
            for (uint32_t p = 0; p < np; p++) {
                const uint32_t pb = post_piece_begin(p, np, k), pe = post_piece_begin(p + 1u, np, k);
                uint32_t e = T.pi_head[post_piece_slot((x >> (2u * pb)) & ((1ull << (2u * (pe - pb))) - 1ull), p, T.gid, T.pi_mask)];
                while (e != (uint32_t)POST_INDEX_NIL) {
                    const uint32_t t = (e >> 2) - T.gbase;
                    if (t < T.n && (e & 3u) == p) {
                        const uint64_t m = x ^ T.le[t];
                        const uint64_t mism = ((m | (m >> 1)) & lsb) | T.bad[t];
#ifdef MTG_EMU
                        const uint32_t nbm = (uint32_t)k - (uint32_t)__builtin_popcountll(mism);
#else
                        const uint32_t nbm = (uint32_t)k - (uint32_t)__popcll(mism);
#endif
                        if (nbm + T.nb_mis >= (uint32_t)k && nbm > 0) {
                            const uint64_t key = ((uint64_t)nbm << 40) | (ORD - ((uint64_t)j * T.n + t));
                            best = key > best ? key : best;
                        }
                    }
                    e = T.pi_next[e];
                }
            }
        }
        if (wave_any((uint32_t)(best >> 40) == (uint32_t)k)) break; /* an exact match: the largest count, and no later position comes before it */
    }
    return best;
}

/* hist: 256 zeroed counters shared by the lanes; tile: POST_TILE + 2 words; blk: 64 words (all LDS on the device) */
/* dbg (timing experiments only, never set by the product): bit 0 = no coverage pass, bit 1 = no terminal search */
MTG_DEV void post_gap(const Index& ix, const FillCfg& cfg, const GapScratch& S, const GapOut& o, const PostTargets& T, uint32_t* hist, uint64_t* tile, uint64_t* blk, PostOut& out,
                      uint32_t dbg = 0)
{
    const int k = ix.k;
    const uint64_t mk = kmask(k);
    const uint64_t lsb = 0x5555555555555555ULL & mk;
    const uint32_t lane = MTG_LANE();
    const uint64_t* words = s_words(cfg, S);
    const SP<uint32_t> cstart = s_cstart(cfg, S);
    const SP<uint32_t> clen = s_clen(cfg, S);
    const SP<uint32_t> tpos = s_tpos(cfg, S);
    const SP<uint32_t> terr = s_terr(cfg, S);
    const SP<uint32_t> ttgt = s_ttgt(cfg, S);
    const uint64_t ORD = (1ull << 40) - 1;
    uint32_t nterm = 0;
    uint32_t pos0 = 0, err0 = 0, tgt0 = 0;
    bool has0 = false;
    /* find_nodes_containing_multiple_R: the scan keeps the first (position-major, target-minor) occurrence of the best match count
     * >= k - nb_mis and stops at the first exact match (src/Filler.cpp:1341-1351); an exact match is the largest possible count, so the
     * result is the first occurrence of the maximum: an arg-max, evaluated here by all lanes at once. */
    const uint64_t le0 = T.n ? T.le[0] : 0ull, bad0 = T.n ? T.bad[0] : ~0ull;
    const bool single = T.n == 1;
    const bool indexed = T.pi_head != nullptr && T.n >= (uint32_t)POST_INDEX_MIN && post_index_usable(k, T.nb_mis);
    /* the lean form (mtg_copy.h): the target's place in the only contig is known, the contig itself was not materialised */
    const LeanRec lean = s_lean(cfg, S)[0];
    out.lean = 0;
    if (lean.valid) {
#ifdef MTG_XCHECK /* TEST-ONLY: the search the lean form replaces finds the same place (the emulation build has copied the contig) */
        {
            const uint32_t L0 = clen[0];
            const uint64_t* w = words + cstart[0];
            uint32_t first = 0xFFFFFFFFu;
            for (uint32_t j = 0; j + (uint32_t)k <= L0; j++) if (le_kmer(w, j, mk) == le0) { first = j; break; }
            if (first != lean.pos0 || !single || bad0 != 0ull || o.n_contigs != 1) { fprintf(stderr, "lean form: target at %u, search finds %u\n", lean.pos0, first); __builtin_trap(); }
        }
#endif
        if (lane == 0) { tpos[0] = lean.pos0; terr[0] = 0; ttgt[0] = 0; }
        has0 = true; pos0 = lean.pos0; err0 = 0; tgt0 = 0; nterm = 1;
        out.lean = 1u + lean.cmd;
    }
    for (uint32_t c = 0; c < (((dbg & 2u) || lean.valid) ? 0u : o.n_contigs); c++) {
        const uint32_t L = clen[c];
        const uint64_t* w = words + cstart[c];
        uint64_t best = 0;
        /* One target without an unusable position (breakpoint mode): an exact occurrence is the largest possible count and the first one wins,
         * whatever inexact matches precede it -- so the contig is first searched for the k-mer itself (a compare per position instead of the
         * mismatch count), and the counting scan below only runs when it is not there. */
        bool have_exact = false;
        /* (contig 0 only: that is where the target of a filled gap sits.  The further contigs of a multi-contig gap -- the alleles behind a refused
         * bubble, the rest of the donor sequence -- do not hold it, and searching them for the k-mer first and counting mismatches afterwards read
         * each of them twice: the counting scan finds an exact occurrence just as well, and stops at it.) */
        if (L >= (uint32_t)k && single && bad0 == 0ull && c == 0) {
            const uint32_t npos = L - (uint32_t)k + 1, nwc = (L + 31) / 32;
            for (uint32_t ws = 0; ws < nwc && !have_exact; ws += POST_TILE) {
                const uint32_t tw = (nwc - ws) < (uint32_t)POST_TILE + 1 ? (nwc - ws) : (uint32_t)POST_TILE + 1;
                wave_sync();
                for (uint32_t i = lane; i < tw; i += MTG_NLANES) tile[i] = w[ws + i];
                if (lane == 0) tile[tw] = 0;
                wave_sync();
                const uint32_t j_lo = 32u * ws;
                if (j_lo >= npos) break;
                const uint32_t j_hi = (npos - j_lo) < 32u * POST_TILE ? npos : j_lo + 32u * POST_TILE;
                for (uint32_t jb = j_lo; jb < j_hi; jb += MTG_NLANES) {
                    const uint32_t j = jb + lane;
                    const int f = wave_first(j < j_hi && le_kmer(tile, j - j_lo, mk) == le0);
                    if (f >= 0) { have_exact = true; best = ((uint64_t)k << 40) | (ORD - (uint64_t)(jb + (uint32_t)f)); break; }
                }
                if (j_hi >= npos) break;
            }
        }
        if (indexed && L >= (uint32_t)k) best = post_search_indexed(T, w, L, k); /* a call: the pass below keeps the registers it had before the index existed */
        else if (!have_exact && L >= (uint32_t)k && T.n) {
            const uint32_t npos = L - (uint32_t)k + 1, nwc = (L + 31) / 32;
            for (uint32_t ws = 0; ws < nwc; ws += POST_TILE) {
                const uint32_t tw = (nwc - ws) < (uint32_t)POST_TILE + 1 ? (nwc - ws) : (uint32_t)POST_TILE + 1; /* one word past the tile: a k-mer may straddle its end */
                wave_sync(); /* the previous tile has been read */
                for (uint32_t i = lane; i < tw; i += MTG_NLANES) tile[i] = w[ws + i];
                if (lane == 0) tile[tw] = 0;
                wave_sync();
                const uint32_t j_lo = 32u * ws;
                if (j_lo >= npos) break;
                const uint32_t j_hi = (npos - j_lo) < 32u * POST_TILE ? npos : j_lo + 32u * POST_TILE;
                bool exact = false; /* an exact match is the largest possible count and the earliest one wins: nothing after it matters */
                for (uint32_t jb = j_lo; jb < j_hi && !exact; jb += MTG_NLANES) {
                    const uint32_t j = jb + lane;
                    if (j < j_hi) {
                        const uint64_t x = le_kmer(tile, j - j_lo, mk);
                        for (uint32_t t = 0; t < T.n; t++) {
                            /* the first target from registers (breakpoint mode has one), the others from memory */
                            const uint64_t m = x ^ (t == 0 ? le0 : T.le[t]);
                            const uint64_t mism = ((m | (m >> 1)) & lsb) | (t == 0 ? bad0 : T.bad[t]);
#ifdef MTG_EMU
                            const uint32_t nbm = (uint32_t)k - (uint32_t)__builtin_popcountll(mism);
#else
                            const uint32_t nbm = (uint32_t)k - (uint32_t)__popcll(mism);
#endif
                            if (nbm + T.nb_mis >= (uint32_t)k && nbm > 0) {
                                const uint64_t key = ((uint64_t)nbm << 40) | (ORD - (single ? (uint64_t)j : (uint64_t)j * T.n + t));
                                best = key > best ? key : best;
                            }
                        }
                    }
                    exact = wave_any((uint32_t)(best >> 40) == (uint32_t)k);
                }
                if (exact) break;
                if (j_hi >= npos) break;
            }
        }
        if (!have_exact) best = wave_max64(best);
#ifdef MTG_XCHECK /* TEST-ONLY: the pass over every target finds the same arg-max as the piece index */
        if (indexed && L >= (uint32_t)k) {
            uint64_t want = 0;
            for (uint32_t j = 0; j + (uint32_t)k <= L && (uint32_t)(want >> 40) != (uint32_t)k; j++) {
                const uint64_t x = le_kmer(w, j, mk);
                for (uint32_t t = 0; t < T.n; t++) {
                    const uint64_t m = x ^ T.le[t];
                    const uint32_t nbm = (uint32_t)k - (uint32_t)__builtin_popcountll(((m | (m >> 1)) & lsb) | T.bad[t]);
                    if (nbm + T.nb_mis >= (uint32_t)k && nbm > 0) { const uint64_t key = ((uint64_t)nbm << 40) | (ORD - ((uint64_t)j * T.n + t)); want = key > want ? key : want; }
                }
            }
            if (want != best) { fprintf(stderr, "piece index: contig %u of gap %u: arg-max %llx, the pass over %u targets finds %llx\n", c, T.gid, (unsigned long long)best, T.n, (unsigned long long)want); __builtin_trap(); }
        }
#endif
        if (best) {
            const uint32_t nbm = (uint32_t)(best >> 40);
            const uint64_t order = ORD - (best & ORD);
            /* one target (breakpoint mode): no 64-bit division, which costs a wave more than the scan of a short contig */
            const uint32_t p = single ? (uint32_t)order : (uint32_t)(order / T.n), t = single ? 0u : (uint32_t)(order % T.n);
            if (lane == 0) { tpos[c] = p; terr[c] = (uint32_t)k - nbm; ttgt[c] = t; }
            if (c == 0) { has0 = true; pos0 = p; err0 = (uint32_t)k - nbm; tgt0 = t; }
            nterm++;
        } else if (lane == 0) {
            tpos[c] = 0xFFFFFFFFu;
        }
    }
    out.nb_terminal = nterm;
    out.clen0 = o.n_contigs ? clen[0] : 0;
    out.fast = 0;
    out.pos = pos0; out.errors = err0; out.target = tgt0;
    out.ab_sum = out.ab_n = out.med_hi = out.med_lo = out.lines = out.direct = 0;
    if (!(has0 && T.fast_ok)) return;
    /* src/GraphAnalysis.cpp:404-407 (pos <= k-1: nothing left of the first node) and :410-423 with pos == k (an empty substr): the
     * sequence is empty and dropped by `sequence.length() > 0` (:449), so the gap has a terminal node but no solution */
    if (pos0 <= (uint32_t)k) { out.fast = 2; return; }
    /* coverage of source + fill = the k-mers of contig0[0:pos0] */
    const uint64_t* w0 = words + cstart[0];
    const uint32_t nk = pos0 - (uint32_t)k + 1;
    const uint32_t nw0 = (pos0 + 31) / 32; /* words holding contig0[0:pos0] */
    uint32_t sum = 0, lines = 0;
    const uint64_t cmpl = 0xAAAAAAAAAAAAAAAAULL & mk;
    /* The traversal took contig0[0:pos0] -- usually all of it -- out of ONE stretch of the unitig store and said so in the copy command it
     * left (CopyCmd: the arena words it stands for and the nucleotides before them that continue the stretch): the k-mers of the region
     * then sit at known places of the store, their abundance bytes next to them: no look-up, nothing to verify. */
    int direct = -1;
    if (ix.us.nwords != 0 && !(dbg & 1u)) {
        const SP<CopyCmd> cmds = s_cmd(cfg, S);
        const uint64_t a_lo = 32ull * cstart[0], a_hi = a_lo + pos0;
        for (uint32_t c = 0; c < o.n_cmds; c++) {
            const uint64_t lo = 32ull * cmds[c].dst - cmds[c].lead, hi = 32ull * ((uint64_t)cmds[c].dst + cmds[c].nwords);
            if (a_lo >= lo && a_hi <= hi) { direct = (int)c; break; }
        }
    }
    if (direct < 0 && nw0 <= (uint32_t)POST_TILE) { /* the look-ups below read the region's k-mers many times: from LDS */
        wave_sync();
        for (uint32_t i = lane; i < nw0; i += MTG_NLANES) tile[i] = w0[i];
        if (lane == 0) tile[nw0] = 0;
        wave_sync();
        w0 = tile;
    }
    if (dbg & 1u) { if (lane == 0) hist[1] = nk; sum = lane == 0 ? nk : 0; } /* timing experiment: no look-ups */
    else if (direct >= 0) {
        const CopyCmd cm = s_cmd(cfg, S)[direct];
        const bool bwd = (cm.src & 1ull) != 0;
        const int64_t p = (int64_t)(cm.src >> 1), rel0 = (int64_t)(32ull * cstart[0]) - (int64_t)(32ull * cm.dst);
        for (uint32_t j = lane; j < nk; j += MTG_NLANES) {
            const int64_t rel = rel0 + (int64_t)j; /* the k-mer's first nucleotide, relative to the command's first one */
            const uint64_t kpos = (uint64_t)(bwd ? p - rel - (int64_t)(k - 1) : p + rel);
            const uint32_t a = ix.us.ab[kpos];
#ifdef MTG_XCHECK /* TEST-ONLY: the k-mer found there is the contig's, and the byte is its abundance */
            {
                Kmer x;
                x.r = le_kmer(w0, j, mk) ^ cmpl;
                x.f = revcomp(x.r, k);
                const uint64_t sr = le_kmer(ix.us.words, kpos, mk) ^ cmpl;
                uint32_t l_ = 0;
                if (sr != (bwd ? x.f : x.r) || a == 0 || a != abundance(ix, x, l_)) {
                    fprintf(stderr, "direct coverage: j %u nk %u pos0 %u cmd %d dst %u nwords %u lead %u bwd %d p %lld kpos %llu a %u ab %u sr %llx x.f %llx x.r %llx cstart0 %u ncmd %u\n", j, nk, pos0, direct, cm.dst, cm.nwords, cm.lead, (int)bwd, (long long)p, (unsigned long long)kpos, a, abundance(ix, x, l_), (unsigned long long)sr, (unsigned long long)x.f, (unsigned long long)x.r, cstart[0], o.n_cmds);
                    __builtin_trap();
                }
            }
#endif
            sum += a;
            hist_add(hist, a);
        }
        out.direct = 1;
    }
    else if (ix.us.nwords == 0) {
        for (uint32_t j = lane; j < nk; j += MTG_NLANES) {
            Kmer x;
            x.r = le_kmer(w0, j, mk) ^ cmpl; /* little-endian image = reversed order: complementing it gives revcomp */
            x.f = revcomp(x.r, k);
            const uint32_t a = abundance(ix, x, lines);
            sum += a;
            hist_add(hist, a);
        }
    } else {
        /* The k-mers of a contig follow the unitigs of the graph, whose abundances lie in the store one byte per k-mer: per block of 64
         * consecutive k-mers ONE look-up (the ADJ entry of the junction behind the block's first k-mer) tells where the block sits in
         * the store; every lane then reads its k-mer there, checks that it is the one it holds (the block may run past the end of the
         * unitig, or through a bubble), and takes the abundance byte next to it.  Whatever is not confirmed is looked up in the hash
         * table as before, so the result is exact whatever the pointers say. */
        const uint64_t mk1 = kmask(k - 1);
        for (uint32_t r0 = 0; r0 < nk; r0 += 64u * 64u) {
            const uint32_t nblk = (nk - r0 + 63u) / 64u < 64u ? (nk - r0 + 63u) / 64u : 64u;
            wave_sync(); /* the previous round's block table has been read */
            for (uint32_t b = lane; b < nblk; b += MTG_NLANES) {
                Kmer x;
                x.r = le_kmer(w0, r0 + 64u * b, mk) ^ cmpl;
                x.f = revcomp(x.r, k);
                uint64_t aux;
                const uint64_t s = x.f & mk1, rs = x.r >> 2;
                adj_get(ix.adj, s <= rs ? s : rs, lines, aux);
                blk[b] = up_is(aux) ? up_resolve(aux, s <= rs) : 0ull;
            }
            wave_sync();
            /* four blocks at a time: the store reads of a lane (two words of sequence, one abundance byte per block) are independent of
             * each other, so they are all issued before the first one is looked at */
            for (uint32_t b0 = 0; b0 < nblk; b0 += 4) {
                for (uint32_t t = lane; t < 64u; t += MTG_NLANES) {
                    uint64_t sr[4];
                    uint32_t av[4];
                    bool in_store[4], bw[4];
MTG_UNROLL
                    for (uint32_t u = 0; u < 4; u++) {
                        const uint32_t bb = b0 + u, j = r0 + 64u * bb + t;
                        const uint64_t up = bb < nblk ? blk[bb] : 0ull;
                        const uint32_t off = up_off(up);
                        bw[u] = up_bwd(up);
                        in_store[u] = up != 0ull && j < nk && (!bw[u] || t <= off);
                        const uint32_t idx = bw[u] ? off - t : off - 1u + t; /* the k-mer's place in the unitig if the walk stayed on it */
                        const uint64_t base = up_hdr(up) + 1;
                        sr[u] = in_store[u] ? le_kmer(ix.us.words + base, idx, mk) ^ cmpl : 0ull;
                        av[u] = in_store[u] ? (uint32_t)ix.us.ab[base * 32 + idx] : 0u;
                    }
MTG_UNROLL
                    for (uint32_t u = 0; u < 4; u++) {
                        const uint32_t j = r0 + 64u * (b0 + u) + t;
                        if (b0 + u >= nblk || j >= nk) continue;
                        Kmer x;
                        x.r = le_kmer(w0, j, mk) ^ cmpl;
                        x.f = revcomp(x.r, k);
                        /* the k-mer read from the store is this lane's k-mer (a read past the end of the unitig finds something else, or
                         * a place where no k-mer starts: abundance byte 0) */
                        uint32_t a = (in_store[u] && sr[u] == (bw[u] ? x.f : x.r)) ? av[u] : 0u;
                        if (a == 0) a = abundance(ix, x, lines);
                        sum += a;
                        hist_add(hist, a);
                    }
                }
            }
        }
    }
    sum = wave_sum32(sum);
    wave_sync();
    uint32_t hi = 0, lo = 0;
    hist_median(hist, nk, hi, lo);
    out.fast = 1;
    out.ab_sum = sum;
    out.ab_n = nk;
    out.med_hi = hi;
    out.med_lo = lo;
    out.lines = wave_sum32(lines);
}

/* ---- The lean gap (mtg_copy.h: the target's place in the only contig is known from the copy command the walk left; nothing was copied).
 * What post_gap does for it is small -- position and errors are known, the k-mers of source + fill are nk consecutive abundance bytes of the
 * unitig store -- and a whole wave per gap spent most of its instructions on being a wave: 100 000 waves that each zero a histogram, reduce
 * over 64 lanes and store a record made k_post the longest kernel of a haploid batch (0.21 of 0.52 ms, bound by instruction issue).  Here a
 * gap is the work of GW lanes (16 on the device: four gaps per wave; 1 in the TEST-ONLY emulation, which runs this next to post_gap for every
 * lean gap and compares the two records).  The byte range is read as aligned 8-byte words (sum and median do not care about the order, so a
 * stretch walked backwards is the same range); `hist` = 256 zeroed counters of the group.
 * post_lean_accumulate: false = not a lean gap (the general kernel's); then a barrier of the group; then post_lean_finish. */
struct LeanWork { uint32_t pos0, cmd, clen0, nk, sum; };
template <uint32_t GW> MTG_DEV bool post_lean_accumulate(const Index& ix, const FillCfg& cfg, const GapScratch& S, const GapOut& o, uint32_t gl, uint32_t* hist, LeanWork& w)
{
    const LeanRec lean = s_lean(cfg, S)[0];
    if (o.status != GAP_OK || !lean.valid) return false;
    const uint32_t k = (uint32_t)ix.k;
    const CopyCmd cm = s_cmd(cfg, S)[lean.cmd];
    const uint32_t cstart0 = s_cstart(cfg, S)[0];
    w.pos0 = lean.pos0; w.cmd = lean.cmd; w.clen0 = s_clen(cfg, S)[0];
    w.nk = lean.pos0 - k + 1u; /* lean_decide: pos0 > k */
    const bool bwd = (cm.src & 1ull) != 0;
    const int64_t p = (int64_t)(cm.src >> 1), rel0 = (int64_t)(32ull * cstart0) - (int64_t)(32ull * cm.dst);
    /* the abundance bytes of the k-mers 0 .. nk - 1 of the contig: ascending from p + rel0, or descending from p - rel0 - (k - 1) */
    const uint64_t first = bwd ? (uint64_t)(p - (rel0 + (int64_t)(w.nk - 1u)) - (int64_t)(k - 1u)) : (uint64_t)(p + rel0);
    const uint8_t* b0 = ix.us.ab + first;
    const uint8_t* b1 = b0 + w.nk;
    const uint8_t* a = (const uint8_t*)((uintptr_t)b0 & ~(uintptr_t)7) + 8u * gl;
    uint32_t sum = 0;
    for (; a < b1; a += 8u * GW) {
        uint64_t v = *(const uint64_t*)a; /* aligned; the store's arrays are padded past their ends */
MTG_UNROLL
        for (uint32_t i = 0; i < 8u; i++, v >>= 8) {
            if (a + i < b0 || a + i >= b1) continue;
            const uint32_t x = (uint32_t)(v & 255ull);
            sum += x;
            hist_add(hist, x);
        }
    }
    w.sum = sum;
    /* (the per-contig terminal arrays of the scratch are not written: they are read for the gaps whose contigs go to the host, never for a lean one) */
    return true;
}
/* the group's sum and the two middle values off the histogram (each lane 256 / GW bins, a scan over the GW lanes) */
template <uint32_t GW> MTG_DEV void post_lean_finish(const LeanWork& w, uint32_t gl, const uint32_t* hist, PostOut& out)
{
    uint32_t sum = w.sum, hi = 0, lo = 0;
    const uint32_t n2 = w.nk / 2;
#ifdef MTG_EMU
    {
        uint32_t cum = 0;
        bool got_hi = false, got_lo = (n2 == 0);
        for (uint32_t v = 0; v < 256 && !(got_hi && got_lo); v++) {
            cum += hist[v];
            if (!got_lo && cum > n2 - 1) { lo = v; got_lo = true; }
            if (!got_hi && cum > n2) { hi = v; got_hi = true; }
        }
    }
#else
    {
        constexpr uint32_t B = 256u / GW;
        for (uint32_t m = GW / 2; m >= 1; m >>= 1) sum += (uint32_t)__shfl_xor((int)sum, (int)m, (int)GW);
        uint32_t s = 0;
        for (uint32_t i = 0; i < B; i++) s += hist[B * gl + i];
        uint32_t incl = s;
        for (uint32_t d = 1; d < GW; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, d, (int)GW); if (gl >= d) incl += y; }
        uint32_t cum = incl - s, f_hi = 0, f_lo = 0;
        for (uint32_t i = 0; i < B; i++) {
            const uint32_t nxt = cum + hist[B * gl + i];
            if (cum <= n2 && nxt > n2) f_hi = B * gl + i + 1u;
            if (n2 > 0 && cum <= n2 - 1u && nxt > n2 - 1u) f_lo = B * gl + i + 1u;
            cum = nxt;
        }
        for (uint32_t m = GW / 2; m >= 1; m >>= 1) {
            const uint32_t y = (uint32_t)__shfl_xor((int)f_hi, (int)m, (int)GW), z = (uint32_t)__shfl_xor((int)f_lo, (int)m, (int)GW);
            f_hi = y > f_hi ? y : f_hi;
            f_lo = z > f_lo ? z : f_lo;
        }
        hi = f_hi ? f_hi - 1u : 0u;
        lo = f_lo ? f_lo - 1u : 0u;
    }
#endif
    out.nb_terminal = 1; out.fast = 1; out.pos = w.pos0; out.errors = 0; out.target = 0; out.clen0 = w.clen0;
    out.ab_sum = sum; out.ab_n = w.nk; out.med_hi = hi; out.med_lo = lo; out.lines = 0; out.direct = 1; out.lean = 1u + w.cmd;
}

} // namespace mtg
#endif
