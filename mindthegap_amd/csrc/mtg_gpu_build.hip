/*
 * mtg_gpu_build.hip -- index construction on the device (Graph::create / Graph::load, /root/reference/src/Filler.cpp:172-226), gfx950 only.
 *
 *   k_jt_insert_* / k_jt_scan / k_jt_plan_emit / k_us_compact / k_us_ab / k_sparse_link : the lean build (mtg_dev.h: junction table -> unitig store -> sparse tables)
 *   k_count / k_count_stats                                                      : exact k-mer counting of streamed reads (-in)
 *   k_insert_kmers / k_lookahead_kmers / k_sparse_ends / k_leftovers             : the k-mers of no unitig in the sparse tables; the container writer
 * (the construction of rounds 1-3 -- dense ADJ / ABND tables, lookaheads for every junction, the store derived from them -- left in round 5)
 */
#include "mtg_gpu_common.h"
#include "mtg_build.h"

namespace mtgi {

/* ------------------------------------------------------------------------------------------------ kernels */
/* Synthetic abundance of a k-mer of the benchmark sets: span > 0: lo + hash % span; span == 0: a Poisson(24) variate (30x reads
 * of 150 nt leave a mean k-mer coverage of 24) drawn by inversion from the 64-bit hash, at least lo.  T[i] = floor(P(X <= i) * 2^64). */
__device__ uint32_t d_synth_abundance(uint64_t c, uint32_t lo, uint32_t span)
{
    const uint64_t h = d_splitmix64(c);
    if (span) return lo + (uint32_t)(h % span);
    static const uint64_t T[64] = {
        0x0000000029820F1FULL, 0x000000040DB37A1BULL, 0x00000032C0047DE8ULL, 0x000001A8528C9C4CULL,
        0x00000A69C1BD52A7ULL, 0x00003470A440BDF3ULL, 0x0000DC8C2E4E6B24ULL, 0x00031CEA99EB0617ULL,
        0x0009DE05DCC0D6EEULL, 0x001BE0F939A5AE81ULL, 0x00471B414BCAE716ULL, 0x00A56BDE8AA7BFA0ULL,
        0x01620D19086170B3ULL, 0x02BE4A7152F35527ULL, 0x051345E41BED6F11ULL, 0x08CE71CEF7173222ULL,
        0x0E6733AF3FD5D6BBULL, 0x164DEB0A00E2FB56ULL, 0x20D6DF830249D6D0ULL, 0x2E258D951F01A8AEULL,
        0x3E1D91AADB117152ULL, 0x505D9655FB237B31ULL, 0x6446559C4CAB85F7ULL, 0x790CADE5ACE06FD0ULL,
        0x8DD3062F0D1559A9ULL, 0xA1C4A29E73AE8C13ULL, 0xB42D81CA34D97F88ULL, 0xC48AB9F119717462ULL,
        0xD2917C5B943CD88BULL, 0xDE2D2614CDB90820ULL, 0xE7767AA8FBB5FAFDULL, 0xEEA6FE347A273B24ULL,
        0xF40B60DD18FC2B41ULL, 0xF7F74B8646AE4E3FULL, 0xFABBF12ADF6848D5ULL, 0xFCA1DF1814EF208BULL,
        0xFDE5D30B8DF3B05AULL, 0xFEB7F4BE3D509803ULL, 0xFF3CABB5D47DCC02ULL, 0xFF8E5761E2C0FFB3ULL,
        0xFFBF57FC51B61EB7ULL, 0xFFDC072B030D6910ULL, 0xFFEC6B45B1886EFAULL, 0xFFF59148ADB5489AULL,
        0xFFFA8EBEAB9F33ACULL, 0xFFFD380EAA825BB5ULL, 0xFFFE9B8650E29D1FULL, 0xFFFF510A38E830E7ULL,
        0xFFFFABCC2CEAFACCULL, 0xFFFFD8401225D09FULL, 0xFFFFED966BB2B223ULL, 0xFFFFF7A0F0313A61ULL,
        0xFFFFFC4354BA6591ULL, 0xFFFFFE5C90BE8C72ULL, 0xFFFFFF4B5615BA2BULL, 0xFFFFFFB386F5F35CULL,
        0xFFFFFFE02E317995ULL, 0xFFFFFFF2FB5802F0ULL, 0xFFFFFFFAC2FE06D0ULL, 0xFFFFFFFDED278637ULL,
        0xFFFFFFFF31381F94ULL, 0xFFFFFFFFB0B85BEBULL, 0xFFFFFFFFE21349FCULL, 0xFFFFFFFFF4E0987CULL};
    uint32_t a = 0;
    for (uint32_t step = 32; step; step >>= 1) if (T[a + step - 1] <= h) a += step; /* a = number of thresholds <= h */
    if (a < 64 && T[a] <= h) a++;
    return a > lo ? a : lo;
}

/* counters[0] = overflow flag, counters[1] = new k-mers */
__global__ void k_insert_kmers(Index ix, const uint64_t* __restrict__ kmers, const uint32_t* __restrict__ ab, size_t n, unsigned long long* counters)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long created = 0, sat = 0;
    int fail = 0;
    for (; i < n; i += stride) {
        int r = index_insert(ix, kmers[i], ab[i]);
        fail |= r & 1;
        created += (r >> 1) & 1;
        sat += ab[i] > 255u;
    }
    if (fail) atomicOr(&counters[0], 1ull);
    if (created) atomicAdd(&counters[1], created);
    if (sat) atomicAdd(&counters[3], sat); /* abundances stored as 255 */
}


/* k-mer counting: one text position per lane (adjacent lanes read adjacent bytes); flags[0] = table too full */
__global__ void k_count(CountTable t, const char* __restrict__ text, uint64_t n, int k, uint32_t npass, uint32_t pass, unsigned long long* flags)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    bool full = false;
    for (; i + k <= n; i += stride) {
        const uint64_t c = kmer_from_ascii(text, i, k);
        if (c == ~0ULL) continue;
        if (npass > 1 && (uint32_t)((mix64(c ^ 0x5851F42D4C957F2DULL) >> 40) % npass) != pass) continue; /* this k-mer belongs to another pass */
        if (!count_insert(t, c)) full = true;
    }
    if (full) atomicOr(&flags[0], 1ull);
}
/* abundance histogram of the distinct k-mers (per-workgroup LDS histogram for the low, hot bins) and number of candidates */
__global__ void k_count_stats(CountTable t, uint32_t keep_min, unsigned long long* histo, uint32_t nbins, unsigned long long* n_keep)
{
    __shared__ unsigned int lh[256];
    for (uint32_t j = threadIdx.x; j < 256; j += blockDim.x) lh[j] = 0;
    __syncthreads();
    unsigned long long keep = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= t.mask; i += (uint64_t)gridDim.x * blockDim.x) {
        if (t.keys[i] == ~0ULL) continue;
        const uint32_t c = t.counts[i];
        const uint32_t b = c < nbins ? c : nbins - 1;
        if (b < 256) atomicAdd(&lh[b], 1u); else atomicAdd(&histo[b], 1ull);
        keep += c >= keep_min;
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < 256 && j < nbins; j += blockDim.x) if (lh[j]) atomicAdd(&histo[j], (unsigned long long)lh[j]);
    if (keep) atomicAdd(n_keep, keep);
}

/* second build pass: lookaheads of the ADJ entries (after every k-mer has been inserted) */
__global__ void k_lookahead_kmers(Index ix, const uint64_t* __restrict__ kmers, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        Kmer x = make_kmer(kmers[i], ix.k);
        build_lookahead(ix, x);
        Kmer y;
        y.f = x.r; y.r = x.f;
        build_lookahead(ix, y);
    }
}

/* ---- unitig store construction (mtg_dev.h: us_*), over the solid k-mers read back from the ABND table ---- */
/* ---- the sparse form (mtg_dev.h: "sparse index") ----
 * the k-mers of no stored unitig, out of the ABND table of the index the unitigs were built from (or of a sparse one): out == nullptr counts */
__global__ void k_leftovers(Index ix, uint64_t* out_k, uint32_t* out_a, unsigned long long* cursor, unsigned long long cap)
{
    const uint64_t nslots = ix.abnd.nbuckets * MTG_ABND_SLOTS;
    uint32_t lines = 0;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nslots; s += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t c;
        const uint32_t a = abnd_slot_kmer(ix.abnd, s, c);
        if (!a || (ix.us.nwords && kmer_stored(ix, c, lines))) continue;
        const unsigned long long at = atomicAdd(cursor, 1ull);
        if (out_k && at < cap) { out_k[at] = c; out_a[at] = a; }
    }
}
/* one stored unitig per wave (a wave per workgroup), its k-mers dealt to the lanes: the entries of its junctions in the new tables;
 * counters[0] = overflow flag.
 * with_bloom: the k-mers also enter the Bloom filter, and not one by one.  Consecutive k-mers of a unitig share their minimizer, hence their
 * 64-byte block, for about (k - m + 1) / 2 positions; four atomics per k-mer (round 4) made every wave instruction touch half a dozen lines
 * four times over (PMC: 112 bytes of HBM traffic per k-mer for the filter alone, 100 of the kernel's 220 ms).  Now the lanes of a run of equal
 * blocks OR their bits together in LDS (eight 64-bit words per run) and the wave flushes eight runs per instruction, the eight words of a run
 * on eight neighbouring lanes: one line per run instead of four per k-mer. */
/* items != nullptr: the work is dealt in PIECES of unitigs ((unitig, first k-mer) pairs, US_PIECE k-mers each) instead of whole unitigs: a store
 * of a few unitigs of millions of k-mers would give a few waves all the work (round 6: 37 unitigs of 125 Mbp, 11.8 s) */
enum : uint32_t { US_PIECE = 1u << 15, US_LONG = 1u << 18 };
__global__ void __launch_bounds__(64) k_sparse_link(Index ix, const UsRec* __restrict__ rec, unsigned long long n, int with_bloom, unsigned long long* counters, const uint2* __restrict__ items)
{
    __shared__ unsigned long long s_bits[64 * 8];
    __shared__ unsigned long long s_blk[64];
    __shared__ unsigned long long s_mh[64 + 32]; /* hashes of the tile's m-mers (k - m <= 31) */
    const uint32_t lane = threadIdx.x;
    const int k = ix.k;
    const bool bloom = with_bloom != 0 && ix.bloom.bits != nullptr;
    Index nb = ix;
    nb.bloom.bits = nullptr; /* sparse_link's own insertion is off: the filter is filled below */
    int fail = 0;
    for (unsigned long long w = blockIdx.x; w < n; w += gridDim.x) {
        const UsRec r = rec[items ? items[w].x : w];
        const uint32_t base_begin = items ? items[w].y : 0u, base_end = items ? (r.len_k - base_begin < (uint32_t)US_PIECE ? r.len_k : base_begin + (uint32_t)US_PIECE) : r.len_k;
        for (uint32_t base = base_begin; base < base_end; base += 64) {
            const uint32_t i = base + lane;
            const bool valid = i < base_end;
            if (valid) fail |= sparse_link(nb, r, i, false);
            if (!bloom) continue;
            /* the minimizer of a k-mer is the smallest hash among its k - m + 1 m-mers, and neighbouring k-mers share all but one of them: the tile's
             * m-mers are hashed ONCE, by the lanes, into LDS (64 + k - m of them; two 64-bit multiplications each), and a k-mer takes the minimum
             * over its window there -- bloom_block hashed fifteen m-mers per k-mer */
            const int mm = ix.bloom.mm;
            const uint32_t span = (uint32_t)(k - mm), tile_n = (base_end - base < 64u ? base_end - base : 64u) + span; /* m-mers of the tile's k-mers */
            for (uint32_t t = lane; t < tile_n; t += 64) s_mh[t] = bloom_mmer_hash(us_peek64(ix.us.words, (r.hdr + 1) * 32 + base + t, (uint32_t)mm, false), mm);
            __syncthreads();
            unsigned long long blk = ~0ull, hb = 0;
            if (valid) {
                const uint64_t mk = kmask(k);
                Kmer x;
                x.r = us_kmer_le(ix.us.words, (r.hdr + 1) * 32 + i, k) ^ (0xAAAAAAAAAAAAAAAAULL & mk);
                x.f = revcomp(x.r, k);
                unsigned long long best = ~0ull;
                for (uint32_t t = 0; t <= span; t++) { const unsigned long long h = s_mh[lane + t]; best = h < best ? h : best; }
                blk = bloom_block_of_min(ix.bloom, best);
                hb = bloom_bits(canon(x));
            }
            const unsigned long long prev = __shfl_up(blk, 1, 64);
            const bool lead = valid && (lane == 0 || prev != blk);
            const unsigned long long lm = __ballot(lead);
            const uint32_t run = (uint32_t)__popcll(lm & ((2ull << lane) - 1ull)) - 1u; /* leaders at or below this lane, minus one */
            const uint32_t nruns = (uint32_t)__popcll(lm);
            for (uint32_t t = lane; t < nruns * 8u; t += 64) s_bits[t] = 0ull;
            if (lead) s_blk[run] = blk;
            __syncthreads();
            if (valid) {
MTG_UNROLL
                for (int j = 0; j < MTG_BLOOM_NHASH; j++) {
                    const uint32_t bit = (uint32_t)(hb >> (9 * j)) & 511u;
                    atomicOr(&s_bits[run * 8u + (bit >> 6)], 1ull << (bit & 63u));
                }
            }
            __syncthreads();
            for (uint32_t t = lane; t < nruns * 8u; t += 64) {
                const unsigned long long v = s_bits[t];
                if (v) atomicOr(reinterpret_cast<unsigned long long*>(ix.bloom.bits + s_blk[t >> 3] * 16) + (t & 7u), v);
            }
            __syncthreads();
        }
    }
    if (fail) atomicOr(&counters[0], 1ull);
}
/* lookaheads of the entries at the two ends of every stored unitig (the only entries around a stored k-mer that are no pointers) */
__global__ void k_sparse_ends(Index ix, const UsRec* __restrict__ rec, unsigned long long n)
{
    for (unsigned long long u = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; u < n; u += (unsigned long long)gridDim.x * blockDim.x) {
        const UsRec r = rec[u];
        const Kmer first = make_kmer(r.start_f, ix.k);
        Kmer fr;
        fr.f = first.r; fr.r = first.f;
        build_lookahead(ix, fr);
        build_lookahead(ix, run_node(ix.us, (r.hdr + 1) * 32, false, r.len_k - 1, ix.k));
    }
}
/* the records of the stored unitigs from the store itself (an index that comes from its container): hdr[u] = header word of unitig u */
__global__ void k_recs_from_store(UStore us, const uint64_t* __restrict__ hdr, unsigned long long n, int k, UsRec* rec)
{
    const unsigned long long u = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n) return;
    UsRec r;
    r.hdr = hdr[u];
    r.len_k = (uint32_t)us.words[r.hdr] - (uint32_t)k + 1u;
    r.pad_ = 0;
    r.start_f = run_node(us, (r.hdr + 1) * 32, false, 0, k).f;
    rec[u] = r;
}

/* ---- the lean build (mtg_dev.h: "the lean build"): junction table -> unitig store -> sparse tables ---- */
/* abundance of a k-mer of the synthetic sets: a function of the k-mer, no table */
struct AbSynth {
    uint32_t lo, span;
    __device__ uint32_t operator()(uint64_t c, uint32_t&) const { return d_synth_abundance(c, lo, span); }
};
/* sum of v over the wave, valid in lane 0 */
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
    for (int d = 32; d; d >>= 1) v += __shfl_down(v, d, 64);
    return v;
}
/* the junctions of packed sequences: one workgroup per sequence, lanes stride over the (k-1)-mer positions; a junction's entry gets the
 * bits of the k-mers on its two sides in ONE table operation (index_insert makes four per k-mer).  counters[0] = overflow flag */
__global__ void __launch_bounds__(256) k_jt_insert_packed(Table jt, int k, const uint64_t* __restrict__ words, const uint64_t* __restrict__ word_off, const uint32_t* __restrict__ len,
                                                          size_t nseq, unsigned long long* counters)
{
    const uint64_t mk1 = kmask(k - 1), cmpl1 = 0xAAAAAAAAAAAAAAAAULL & mk1;
    int fail = 0;
    for (size_t s = blockIdx.x; s < nseq; s += gridDim.x) {
        const uint64_t* w = words + word_off[s];
        const uint32_t L = len[s];
        if (L < (uint32_t)k) continue;
        for (uint32_t q = threadIdx.x; q + (uint32_t)k - 1 <= L; q += blockDim.x) {
            /* nucleotides q-1 .. q+k-1 in one little-endian window [a][J: k-1][b] (at most 32 nucleotides: one 64-bit value) */
            const bool has_a = q >= 1, has_b = q + (uint32_t)k - 1 < L;
            const uint32_t q0 = has_a ? q - 1 : q, sh = 2u * (q0 & 31u), need = (uint32_t)k - 1u + (has_a ? 1u : 0u) + (has_b ? 1u : 0u);
            uint64_t win = w[q0 >> 5] >> sh; /* nt q0 + i at bits 2i */
            if ((q0 & 31u) + need > 32u) win |= w[(q0 >> 5) + 1] << (64u - sh); /* sh > 0 here */
            const uint32_t a = (uint32_t)win & 3u;
            const uint64_t body = has_a ? win >> 2 : win;
            const uint64_t jle = body & mk1;                                      /* J, little-endian image */
            const uint32_t b = (uint32_t)(body >> (2 * (k - 1))) & 3u;
            const uint64_t jr = jle ^ cmpl1, jf = revcomp(jr, k - 1);             /* complemented image = reverse complement */
            fail |= jt_insert_junction(jt, jf, jr, has_a, a, has_b, b) & 1;
        }
    }
    if (fail) atomicOr(&counters[0], 1ull);
}
/* ---- the junction table of packed sequences, PARTITIONED (round 6).  k_jt_insert_packed makes one read and one compare-and-swap per junction
 * position on a random bucket of a 34 GB table: 3.0e9 of each at human scale, 170 ms, which is the rate this memory gives scattered atomics --
 * the table's bytes crossed the memory bus a dozen times.  bucket_of() is monotone in the hash, so the junctions can be sorted by where they
 * go instead: with the bucket count a multiple of 2^(b1 + b2), the hashes with the same top b1 + b2 bits own one SEGMENT of m consecutive buckets
 * (at most 2 048: 64 KB), and
 *   1. k_bin_positions   streams the sequences and writes an 8-byte record per junction position -- [hash without its top b1 bits | edge bits] --
 *                        into the region of its level-1 bin (2^b1 bins, eight regions each: one per XCD, so that a memory line is written by
 *                        one L2), a tile of 8 192 records at a time: histogram in LDS, one global atomic per bin and tile, runs of ~8 records;
 *   2. k_bin_records     reads a level-1 bin and deals its records to the 2^b2 segments below it -- INTO THE TABLE'S OWN MEMORY: a segment's
 *                        region holds its record list (word 0: the count) until the table is built there;
 *   3. k_build_segments  one workgroup per segment: the list is read once, the segment is built in LDS (the same probing as jt_or, LDS
 *                        atomics) and written over the list in one coalesced sweep;
 *   4. what did not fit -- a record beyond a region's capacity, a key displaced past its segment's last bucket -- is on an overflow list and
 *                        goes in with the ordinary insertion (k_jt_overflow) once the segments are written: a few per million.
 * The level-1 regions live behind the table in its buffer (the sparse ADJ table that takes the buffer over is larger), the sequences go through
 * in chunks that fit there.  Traffic: 24 GB of records written and read twice, the table written once -- no clearing pass, no atomic on
 * device memory but the bins' cursors. */
struct BinShape {
    uint32_t kb, b1, b2;  /* key bits; bits of the level-1 bin, of the segment below it */
    uint64_t m;           /* buckets per segment */
    uint64_t cap1;        /* records per (level-1 bin, XCD) region */
    uint64_t seg_words;   /* m * MTG_ABND_SLOTS */
};
struct BinOverflow {
    uint64_t* h;          /* hash, edge bits: two words per entry */
    unsigned long long* cursor;
    unsigned long long cap;
};
__device__ __forceinline__ void bin_overflow_push(const BinOverflow& ov, uint64_t H, uint32_t bits)
{
    const unsigned long long at = atomicAdd(ov.cursor, 1ull);
    if (at < ov.cap) { ov.h[2 * at] = H; ov.h[2 * at + 1] = bits; }
}
enum { BIN_TILE = 8192 };
/* MTG_BIN_SORTED: a tile's records are put in the order of their bins IN LDS before they leave, so that neighbouring lanes write neighbouring words (a
 * record a lane to 1 024 bins is a write request per record -- 2.9e9 of them in k_bin_positions, PMC; in bin order a request carries a bin's run) */
#ifndef MTG_BIN_SORTED
#define MTG_BIN_SORTED 1
#endif
/* (sorted: two workgroups a compute unit with tiles of 3 584 records measured best -- 50.7 ms for the two binning kernels; one workgroup with 8 192: 54.5,
 * three with 2 048: 50.9, scripts/r6_binsort_ab.sh and the EXTRA= variants of the Makefile) */
#ifndef MTG_BIN1_TILE
#define MTG_BIN1_TILE (MTG_BIN_SORTED ? 3584 : 4096)
#endif
#ifndef MTG_BIN1_THREADS
#define MTG_BIN1_THREADS 512
#endif
#ifndef MTG_BIN1_GROUPS
#define MTG_BIN1_GROUPS (MTG_BIN_SORTED ? 512 : 768)
#endif
#ifndef BIN1_U
#define BIN1_U (MTG_BIN_SORTED ? 2 : 4)
#endif
/* exclusive prefix sums of cnt[0 .. nb) into pre[0 .. nb) by the whole workgroup (nb a power of two; s_wt: one word per wave).  Ends with a barrier. */
__device__ __forceinline__ void block_prefix(const uint32_t* cnt, uint32_t* pre, uint32_t nb, uint32_t* s_wt)
{
    const uint32_t per = nb > blockDim.x ? nb / blockDim.x : 1u; /* consecutive bins a thread */
    const uint32_t b0 = threadIdx.x * per, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t mine = 0;
    if (b0 < nb) for (uint32_t j = 0; j < per; j++) mine += cnt[b0 + j];
    uint32_t inc = mine;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64); if (lane >= (uint32_t)d) inc += o; }
    if (lane == 63u) s_wt[wave] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t v = 0; v < wave; v++) before += s_wt[v];
    uint32_t run = before + inc - mine;
    if (b0 < nb) for (uint32_t j = 0; j < per; j++) { const uint32_t c = cnt[b0 + j]; pre[b0 + j] = run; run += c; }
    __syncthreads();
}
/* a tile of staged (hash, bits) pairs of this workgroup to ITS regions of the level-1 bins (region (bin, workgroup): no other workgroup writes there, so
 * the cursors are the workgroup's own, in LDS, for as long as it runs -- with cursors in device memory the bins took an atomic per eight records, and
 * scattered atomics are what the construction is leaving behind) */
#if MTG_BIN_SORTED
__device__ __forceinline__ void bin1_flush(const BinShape& S, uint32_t n, const uint64_t* s_h, const uint8_t* s_b, uint32_t* s_hist, uint32_t* s_pre, uint32_t* s_cur, uint64_t* bins1, const BinOverflow& ov,
                                           uint64_t* s_oh, uint8_t* s_ob, uint32_t* s_wt)
{
    const uint32_t nb1 = 1u << S.b1, shb = S.kb - S.b1;
    for (uint32_t b = threadIdx.x; b < nb1; b += blockDim.x) s_hist[b] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(&s_hist[(uint32_t)(s_h[i] >> shb)], 1u);
    __syncthreads();
    block_prefix(s_hist, s_pre, nb1, s_wt);
    for (uint32_t b = threadIdx.x; b < nb1; b += blockDim.x) { s_cur[b] += s_hist[b]; s_hist[b] = 0; } /* the region's cursor BEFORE this tile: s_cur - (the bin's count = s_hist once the tile is sorted) */
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint64_t H = s_h[i];
        const uint32_t b = (uint32_t)(H >> shb);
        const uint32_t at = s_pre[b] + atomicAdd(&s_hist[b], 1u);
        s_oh[at] = H;
        s_ob[at] = s_b[i];
    }
    __syncthreads();
    const uint64_t low = (1ull << shb) - 1ull;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) { /* lane i, lane i + 1: the same bin's next word, nearly always */
        const uint64_t H = s_oh[i];
        const uint32_t b = (uint32_t)(H >> shb);
        const uint64_t pos = (uint64_t)(s_cur[b] - s_hist[b]) + (i - s_pre[b]);
        if (pos < S.cap1) bins1[(((uint64_t)blockIdx.x << S.b1) + b) * S.cap1 + pos] = ((H & low) << 8) | s_ob[i];
        else bin_overflow_push(ov, H, s_ob[i]);
    }
    __syncthreads();
}
#else
__device__ __forceinline__ void bin1_flush(const BinShape& S, uint32_t n, const uint64_t* s_h, const uint8_t* s_b, uint32_t* s_hist, uint32_t* s_base, uint32_t* s_cur, uint64_t* bins1, const BinOverflow& ov,
                                           uint64_t*, uint8_t*, uint32_t*)
{
    const uint32_t nb1 = 1u << S.b1;
    for (uint32_t b = threadIdx.x; b < nb1; b += blockDim.x) s_hist[b] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(&s_hist[(uint32_t)(s_h[i] >> (S.kb - S.b1))], 1u);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb1; b += blockDim.x) {
        s_base[b] = s_cur[b];
        s_cur[b] += s_hist[b];
        s_hist[b] = 0;
    }
    __syncthreads();
    const uint64_t low = (1ull << (S.kb - S.b1)) - 1ull;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint64_t H = s_h[i];
        const uint32_t b = (uint32_t)(H >> (S.kb - S.b1));
        const uint64_t pos = (uint64_t)s_base[b] + atomicAdd(&s_hist[b], 1u);
        if (pos < S.cap1) bins1[(((uint64_t)blockIdx.x << S.b1) + b) * S.cap1 + pos] = ((H & low) << 8) | s_b[i];
        else bin_overflow_push(ov, H, s_b[i]);
    }
    __syncthreads();
}
#endif
__global__ void __launch_bounds__(MTG_BIN1_THREADS) k_bin_positions(BinShape S, int k, const uint64_t* __restrict__ words, const uint64_t* __restrict__ word_off, const uint32_t* __restrict__ len,
                                                       size_t s0, size_t s1, uint64_t* bins1, uint32_t* cur1, BinOverflow ov)
{
    extern __shared__ uint64_t s_dyn[];
    uint64_t* s_h = s_dyn;                                        /* MTG_BIN1_TILE hashes */
#if !MTG_BIN_SORTED
    uint8_t* s_b = reinterpret_cast<uint8_t*>(s_dyn + MTG_BIN1_TILE);  /* MTG_BIN1_TILE edge masks */
#endif
#if MTG_BIN_SORTED
    uint64_t* s_oh = s_dyn + MTG_BIN1_TILE;                       /* the tile again, in bin order */
    uint8_t* s_b = reinterpret_cast<uint8_t*>(s_dyn + 2 * MTG_BIN1_TILE);
    uint8_t* s_ob = s_b + MTG_BIN1_TILE;
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(s_ob + MTG_BIN1_TILE);
#else
    uint64_t* s_oh = nullptr;
    uint8_t* s_ob = nullptr;
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(s_b + MTG_BIN1_TILE);
#endif
    uint32_t* s_base = s_hist + 1024;
    uint32_t* s_cur = s_base + 1024;
    __shared__ uint32_t s_n;
    __shared__ uint32_t s_wt[16];
    const uint32_t nb1 = 1u << S.b1;
    for (uint32_t b = threadIdx.x; b < nb1; b += blockDim.x) s_cur[b] = 0;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint64_t mk1 = kmask(k - 1), cmpl1 = 0xAAAAAAAAAAAAAAAAULL & mk1;
    for (size_t s = s0 + blockIdx.x; s < s1; s += gridDim.x) {
        const uint64_t* w = words + word_off[s];
        const uint32_t L = len[s];
        if (L < (uint32_t)k) continue;
        const uint32_t npos = L - (uint32_t)k + 2u; /* junction positions 0 .. L - k + 1 */
        for (uint32_t q0 = 0; q0 < npos; q0 += blockDim.x * BIN1_U) { /* BIN1_U positions a lane and turn: their loads are in flight together, the turn's two barriers are shared */
            const uint32_t n_now = s_n; /* every thread reads the count BEFORE any wave of this round adds to it: the barrier keeps the test uniform */
            __syncthreads();
            if (n_now + blockDim.x * BIN1_U > MTG_BIN1_TILE) {
                bin1_flush(S, n_now, s_h, s_b, s_hist, s_base, s_cur, bins1, ov, s_oh, s_ob, s_wt);
                if (threadIdx.x == 0) s_n = 0;
                __syncthreads();
            }
            uint64_t Hs[BIN1_U];
            uint32_t bs[BIN1_U];
#pragma unroll
            for (int u = 0; u < BIN1_U; u++) {
                const uint32_t q = q0 + (uint32_t)u * blockDim.x + threadIdx.x;
                Hs[u] = 0; bs[u] = 0;
                if (q < npos) { /* as k_jt_insert_packed: nucleotides q-1 .. q+k-1 in one little-endian window [a][J: k-1][b] */
                    const bool has_a = q >= 1, has_b = q + (uint32_t)k - 1 < L;
                    const uint32_t qa = has_a ? q - 1 : q, sh = 2u * (qa & 31u), need = (uint32_t)k - 1u + (has_a ? 1u : 0u) + (has_b ? 1u : 0u);
                    uint64_t win = w[qa >> 5] >> sh;
                    if ((qa & 31u) + need > 32u) win |= w[(qa >> 5) + 1] << (64u - sh);
                    const uint32_t a = (uint32_t)win & 3u;
                    const uint64_t body = has_a ? win >> 2 : win;
                    const uint64_t jle = body & mk1;
                    const uint32_t b = (uint32_t)(body >> (2 * (k - 1))) & 3u;
                    const uint64_t jr = jle ^ cmpl1, jf = revcomp(jr, k - 1);
                    bs[u] = jt_junction_bits(jf, jr, has_a, a, has_b, b);
                    Hs[u] = mix(jf <= jr ? jf : jr, S.kb);
                    /* a junction the scan must look at in full (jt_special: palindromic, or a run of one nucleotide) goes in by the ordinary insertion, which
                     * flags its entry: every occurrence of it, so none reaches a segment */
                    if (bs[u] && jt_special(jf, jr, S.kb)) { bin_overflow_push(ov, Hs[u], bs[u] | 0x100u); bs[u] = 0; }
                }
            }
#pragma unroll
            for (int u = 0; u < BIN1_U; u++) { /* the wave's records one after the other in the staging arrays */
                const unsigned long long mball = __ballot(bs[u] != 0u);
                uint32_t wbase = 0;
                if ((threadIdx.x & 63u) == 0 && mball) wbase = atomicAdd(&s_n, (uint32_t)__popcll(mball));
                wbase = (uint32_t)__shfl((int)wbase, 0, 64);
                if (bs[u]) {
                    const uint32_t at = wbase + (uint32_t)__popcll(mball & ((1ull << (threadIdx.x & 63u)) - 1ull));
                    s_h[at] = Hs[u];
                    s_b[at] = (uint8_t)bs[u];
                }
            }
            __syncthreads();
        }
    }
    const uint32_t n = s_n;
    if (n) bin1_flush(S, n, s_h, s_b, s_hist, s_base, s_cur, bins1, ov, s_oh, s_ob, s_wt);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb1; b += blockDim.x) cur1[(uint64_t)b * gridDim.x + blockIdx.x] = s_cur[b]; /* records of region (b, this workgroup): may exceed cap1 (the rest went to the overflow list) */
}
/* workgroup p: the records of level-1 bin p -- its G regions, one after the other, tile by tile -- to the lists of the 2^b2 segments below it, in the
 * table's own memory.  One workgroup writes all the lists of its bin, so their cursors are its own too: in LDS while it runs, in the lists' count words
 * (word 0 of a segment's region) between the chunks of the input. */
__global__ void __launch_bounds__(1024) k_bin_records(BinShape S, const uint64_t* __restrict__ bins1, const uint32_t* __restrict__ cur1, uint32_t G, uint64_t* table, BinOverflow ov)
{
    extern __shared__ uint64_t s_dyn2[];
    uint64_t* s_r = s_dyn2; /* BIN_TILE records */
#if MTG_BIN_SORTED
    uint64_t* s_o = s_dyn2 + BIN_TILE; /* the tile again, in bin order */
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(s_dyn2 + 2 * BIN_TILE);
    __shared__ uint32_t s_wt[16];
#else
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(s_dyn2 + BIN_TILE);
#endif
    uint32_t* s_base = s_hist + 2048;
    uint32_t* s_cur = s_base + 2048;
    const uint32_t nb2 = 1u << S.b2, p = blockIdx.x;
    const uint32_t sh2 = S.kb - S.b1 - S.b2 + 8u; /* the sub-bin's bits in a record */
    for (uint32_t b = threadIdx.x; b < nb2; b += blockDim.x) s_cur[b] = (uint32_t)table[(((uint64_t)p << S.b2) | b) * S.seg_words];
    __syncthreads();
    uint32_t g = 0;
    uint64_t off = 0; /* records of region g already taken */
    while (g < G) {
        /* a tile: what is left of region g, then whole further regions while they fit */
        uint32_t fill = 0;
        while (g < G && fill < BIN_TILE) {
            const uint64_t c = cur1[(uint64_t)p * G + g];
            const uint64_t n = c < S.cap1 ? c : S.cap1;
            const uint64_t left = n - off;
            const uint32_t take = (uint32_t)(left < BIN_TILE - fill ? left : BIN_TILE - fill);
            if (take < left && fill) break; /* a region is split only when it does not fit an empty tile */
            const uint64_t* src = bins1 + (((uint64_t)g << S.b1) + p) * S.cap1 + off; /* a workgroup's regions lie together (its writes stay within a few pages); this reader takes 14 KB from each */
            for (uint32_t i = threadIdx.x; i < take; i += blockDim.x) s_r[fill + i] = src[i];
            fill += take;
            if (take == left) { g++; off = 0; } else off += take;
        }
        for (uint32_t b = threadIdx.x; b < nb2; b += blockDim.x) s_hist[b] = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < fill; i += blockDim.x) atomicAdd(&s_hist[(uint32_t)(s_r[i] >> sh2) & (nb2 - 1u)], 1u);
        __syncthreads();
#if MTG_BIN_SORTED
        block_prefix(s_hist, s_base /* the prefix sums */, nb2, s_wt);
        for (uint32_t b = threadIdx.x; b < nb2; b += blockDim.x) { s_cur[b] += s_hist[b]; s_hist[b] = 0; }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < fill; i += blockDim.x) {
            const uint64_t rec = s_r[i];
            const uint32_t b = (uint32_t)(rec >> sh2) & (nb2 - 1u);
            s_o[s_base[b] + atomicAdd(&s_hist[b], 1u)] = rec;
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < fill; i += blockDim.x) {
            const uint64_t rec = s_o[i];
            const uint32_t b = (uint32_t)(rec >> sh2) & (nb2 - 1u);
            const uint64_t pos = (uint64_t)(s_cur[b] - s_hist[b]) + (i - s_base[b]);
            const uint64_t seg = ((uint64_t)p << S.b2) | b;
            if (pos + 1 < S.seg_words) table[seg * S.seg_words + 1 + pos] = rec;
            else bin_overflow_push(ov, ((uint64_t)p << (S.kb - S.b1)) | (rec >> 8), (uint32_t)rec & 255u);
        }
        __syncthreads();
        continue;
#endif
        for (uint32_t b = threadIdx.x; b < nb2; b += blockDim.x) {
            s_base[b] = s_cur[b];
            s_cur[b] += s_hist[b];
            s_hist[b] = 0;
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < fill; i += blockDim.x) {
            const uint64_t rec = s_r[i];
            const uint32_t b = (uint32_t)(rec >> sh2) & (nb2 - 1u);
            const uint64_t pos = (uint64_t)s_base[b] + atomicAdd(&s_hist[b], 1u);
            const uint64_t seg = ((uint64_t)p << S.b2) | b;
            if (pos + 1 < S.seg_words) table[seg * S.seg_words + 1 + pos] = rec;
            else bin_overflow_push(ov, ((uint64_t)p << (S.kb - S.b1)) | (rec >> 8), (uint32_t)rec & 255u);
        }
        __syncthreads();
    }
    for (uint32_t b = threadIdx.x; b < nb2; b += blockDim.x) table[(((uint64_t)p << S.b2) | b) * S.seg_words] = s_cur[b];
}
/* word 0 of every segment's region: the count of its list */
__global__ void k_bin_clear_counts(BinShape S, uint64_t* table, uint64_t nseg)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nseg; j += (uint64_t)gridDim.x * blockDim.x) table[j * S.seg_words] = 0;
}
/* one segment per workgroup and turn: its list -> its buckets, built in LDS.  A turn is a chain of latencies (count, records, barrier, barrier, write), so
 * several small workgroups share a compute unit rather than one large one. */
#ifndef MTG_SEG_THREADS
#define MTG_SEG_THREADS 512
#endif
/* a workgroup barrier that waits for the wave's LDS operations only: the lists' loads and the segments' stores of neighbouring turns stay in flight
 * across it (__syncthreads() drains them: every turn then pays the memory's latency three times over) */
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__global__ void __launch_bounds__(MTG_SEG_THREADS) k_build_segments(BinShape S, Table jt, uint64_t nseg, BinOverflow ov)
{
    extern __shared__ uint64_t s_tab[]; /* seg_words slots (a multiple of four: whole 32-byte buckets) */
    const uint64_t tagm = (1ull << jt.tag_bits) - 1ull;
    const uint32_t nq = (uint32_t)(S.seg_words / 2); /* 16-byte pieces */
    for (uint64_t j = blockIdx.x; j < nseg; j += gridDim.x) {
        uint64_t* reg = jt.slots + j * S.seg_words;
        const unsigned long long c = reg[0];
        const uint64_t n = c < S.seg_words - 1 ? c : S.seg_words - 1;
        {
            U64x2 z; z.x = z.y = 0;
            for (uint32_t i = threadIdx.x; i < nq; i += blockDim.x) reinterpret_cast<U64x2*>(s_tab)[i] = z;
        }
        lds_barrier();
        const uint64_t hi = (j >> S.b2) << (S.kb - S.b1);
        const uint64_t b0 = j * S.m;
        for (uint64_t base = 0; base < n; base += (uint64_t)blockDim.x * 8u) {
            uint64_t cur[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const uint64_t i = base + (uint64_t)u * blockDim.x + threadIdx.x; cur[u] = i < n ? reg[1 + i] : 0ull; } /* eight loads in flight (a record's edge bits are never 0) */
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint64_t rec = cur[u];
                if (!rec) continue;
                const uint64_t H = hi | (rec >> 8);
                const uint32_t bits = (uint32_t)rec & 255u;
                const uint64_t bl0 = bucket_of(H, jt.nbuckets, jt.key_bits) - b0;
                const uint64_t tag = H & tagm;
                bool done = false;
                for (uint32_t d = 0; d <= MTG_MAX_DISP && !done; d++) {
                    const uint64_t bl = bl0 + d;
                    if (bl >= S.m) break; /* past the segment's last bucket: the ordinary insertion, later */
                    const uint64_t want = (tag << MTG_DISP_BITS) | d;
                    unsigned long long* p = reinterpret_cast<unsigned long long*>(s_tab + bl * MTG_ABND_SLOTS);
                    /* the bucket in one go (two 16-byte reads), then one compare-and-swap on the slot it points to; a lost race reads the bucket again */
                    for (int turn = 0; turn < 6 && !done; turn++) {
                        const volatile unsigned long long* pv = p; /* (four reads issued together; volatile: a turn after a lost race must see the winner's slot) */
                        const unsigned long long q[4] = {pv[0], pv[1], pv[2], pv[3]};
                        int hit = -1, fr = -1;
#pragma unroll
                        for (int t = 3; t >= 0; t--) { if (q[t] == 0ull) fr = t; if ((q[t] >> 8) == want && q[t] != 0ull) hit = t; }
                        if (hit >= 0) { if ((q[hit] & bits) != bits) atomicOr(p + hit, (unsigned long long)bits); done = true; break; }
                        if (fr < 0) break; /* a full bucket: the next one */
                        const unsigned long long v = atomicCAS(p + fr, 0ull, (unsigned long long)((want << 8) | bits));
                        if (v == 0ull) { done = true; break; }
                        if ((v >> 8) == want) { if ((v & bits) != bits) atomicOr(p + fr, (unsigned long long)bits); done = true; break; }
                    }
                }
                if (!done) bin_overflow_push(ov, H, bits);
            }
        }
        lds_barrier();
#ifndef SEG_NO_WRITE
        for (uint32_t i = threadIdx.x; i < nq; i += blockDim.x) reinterpret_cast<U64x2*>(reg)[i] = reinterpret_cast<const U64x2*>(s_tab)[i];
#else
        if (threadIdx.x == 0) reg[0] = s_tab[0];
#endif
        lds_barrier();
    }
}
/* the overflow list through the ordinary insertion; counters[0] = displacement overflow */
__global__ void k_jt_overflow(Table jt, const uint64_t* __restrict__ h, unsigned long long n, unsigned long long* counters)
{
    unsigned long long fail = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const int f = jt_or_h(jt, h[2 * i], (uint32_t)h[2 * i + 1] & 255u, (h[2 * i + 1] & 0x100ull) ? JT_MARK : 0ull) & 1;
        if (f) { counters[3] = h[2 * i]; counters[1] = h[2 * i + 1]; }
        fail += f;
    }
    if (fail) { atomicOr(&counters[0], 1ull); atomicAdd(&counters[2], fail); }
}

/* a counted solid set handed over as a list: abundances into the (dense) ABND table that serves as their source, junctions into the
 * junction table.  counters[0] = overflow flag, counters[3] += abundances above 255 */
__global__ void k_jt_insert_kmers(Table jt, Table abnd, int k, const uint64_t* __restrict__ kmers, const uint32_t* __restrict__ ab, size_t n, unsigned long long* counters)
{
    unsigned long long sat = 0;
    int fail = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint64_t c = kmers[i];
        fail |= table_or<MTG_ABND_SLOTS>(abnd, c, ab_stored(ab[i])) & 1;
        fail |= jt_insert_kmer(jt, c, k);
        sat += ab[i] > 255u;
    }
    if (fail) atomicOr(&counters[0], 1ull);
    if (sat) atomicAdd(&counters[3], sat);
}
/* the solid k-mers of a count table (count in [lo, hi]) into the junction table; with_abnd: also into an ABND table (several counting
 * passes: the count table of a pass does not outlive it) */
__global__ void k_jt_insert_from_counts(Table jt, Table abnd, int with_abnd, int k, CountTable t, uint32_t lo, uint32_t hi, unsigned long long* counters)
{
    unsigned long long sat = 0;
    int fail = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= t.mask; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t key = t.keys[i];
        if (key == ~0ULL) continue;
        const uint32_t c = t.counts[i];
        if (c < lo || c > hi) continue;
        if (with_abnd) { fail |= table_or<MTG_ABND_SLOTS>(abnd, key, ab_stored(c)) & 1; sat += c > 255u; }
        fail |= jt_insert_kmer(jt, key, k);
    }
    if (fail) atomicOr(&counters[0], 1ull);
    if (sat) atomicAdd(&counters[3], sat);
}
/* ONE streaming pass over the junction table, a bucket per lane (jt_scan_bucket: two 16-byte reads, the keys from one division): statistics,
 * chain starts and the k-mers of no chain.  The lists have the capacities the host guessed; the cursors count past them (round 4 scanned
 * twice: count, then collect -- 2 x 48 ms at human scale, bound by the arithmetic of decoding every slot on its own). */
/* the chain starts a wave finds wait in a queue of its own in LDS and leave together: one atomic on the list's cursor per SCAN_Q_FLUSH starts instead of
 * one each (1.2e6 atomics on one word were half the pass: 16 -> 9 ms) */
enum { SCAN_Q = 192, SCAN_Q_FLUSH = 64 };
struct ScanQueue {
    uint32_t n;
    uint32_t pad_;
    uint64_t v[SCAN_Q];
};
struct ScanStartsQueued {
    ScanQueue* q;
    ScanStartsGlobal direct; /* a full queue (a wave that met more than SCAN_Q starts between two looks at it) */
    __device__ void operator()(uint64_t start_f) const
    {
        const uint32_t at = atomicAdd(&q->n, 1u);
        if (at < (uint32_t)SCAN_Q) q->v[at] = start_f;
        else direct(start_f);
    }
};
/* the wave's lanes together: the queue's entries to the list (all of them when `all`, else only once SCAN_Q_FLUSH wait) */
__device__ __forceinline__ void scan_queue_flush(ScanQueue* q, const ScanStartsGlobal& g, bool all)
{
    __builtin_amdgcn_wave_barrier();
    const uint32_t waiting = *(volatile uint32_t*)&q->n;
    const uint32_t n = waiting < (uint32_t)SCAN_Q ? waiting : (uint32_t)SCAN_Q; /* (what came past the end went to the list directly) */
    if (n == 0 || (!all && n < (uint32_t)SCAN_Q_FLUSH)) return;
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(&g.counters[JT_C_STARTS], (unsigned long long)n);
    base = (unsigned long long)__shfl((long long)base, 0, 64);
    for (uint32_t i = lane; i < n; i += 64) if (g.starts && base + i < g.cap_starts) g.starts[base + i] = q->v[i];
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) *(volatile uint32_t*)&q->n = 0;
    __builtin_amdgcn_wave_barrier();
}
template <typename Src>
__global__ void __launch_bounds__(256) k_jt_scan(Table jt, int k, Src src, unsigned long long* counters, uint64_t* starts, unsigned long long cap_starts,
                                                 uint64_t* left_k, uint32_t* left_a, unsigned long long cap_left)
{
    __shared__ ScanQueue s_q[4];
    ScanQueue* const myq = &s_q[threadIdx.x >> 6];
    if ((threadIdx.x & 63u) == 0) myq->n = 0;
    __builtin_amdgcn_wave_barrier();
    const ScanStartsGlobal direct{counters, starts, cap_starts};
    const ScanStartsQueued put_start{myq, direct};
    JtAcc acc{};
    uint32_t lines = 0;
    /* four buckets of a lane in flight (with the common entry judged by its bits the pass waits for the table, not for its arithmetic) */
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b - (threadIdx.x & 63u) < jt.nbuckets; b += 4 * stride) { /* (the wave's lanes stay together: they empty the queue together) */
        uint64_t q[4][MTG_ABND_SLOTS];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (b + (uint64_t)u * stride < jt.nbuckets) jt_bucket_words(jt, b + (uint64_t)u * stride, q[u]);
            else for (int i = 0; i < MTG_ABND_SLOTS; i++) q[u][i] = 0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) jt_scan_bucket_words(jt, k, b + (uint64_t)u * stride, q[u], src, acc, counters, put_start, left_k, left_a, cap_left, lines);
        scan_queue_flush(myq, direct, false);
    }
    scan_queue_flush(myq, direct, true);
    for (int j = 0; j < 6; j++) {
        const unsigned long long v = wave_sum_u64(acc.c[j]);
        if ((threadIdx.x & 63u) == 0 && v) atomicAdd(&counters[j], v);
    }
}
template <typename Src>
__global__ void __launch_bounds__(256) k_jt_unstored(Table jt, Index nx, Src src, unsigned long long* counters, uint64_t* left_k, uint32_t* left_a, unsigned long long cap_left)
{
    const uint64_t nslots = jt.nbuckets * MTG_ABND_SLOTS;
    uint32_t lines = 0;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nslots; s += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t J;
        const uint32_t m = jt_slot_key(jt, s, J);
        if (!m) continue;
        jt_unstored_entry(jt, nx, J, m, src, counters, left_k, left_a, cap_left, lines);
    }
}
/* one chain start per lane: its walker (mtg_build.h: JtWalker) -- the walk towards the other end with the sequence written into chunks on the
 * way, until it reaches that end or meets the walker that started there */
__global__ void __launch_bounds__(64) k_jt_walk(WalkShared S, unsigned long long n)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    JtWalker w;
    w.begin(S, (uint32_t)i);
    while (w.step(S)) {}
}
/* one stored unitig per wave: its words from the chunk chains of its walkers to their place in the store (us_compact); flag[0] |= 1 when a
 * chain is shorter than its walker's count */
__global__ void __launch_bounds__(256) k_us_compact(UStore us, int k, const UsRec* __restrict__ rec, unsigned long long n, WalkShared S, unsigned long long* flag)
{
    const unsigned long long wave = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((unsigned long long)gridDim.x * blockDim.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    bool bad = false;
    for (unsigned long long u = wave; u < n; u += nwaves) bad = !us_compact(us, k, rec[u], S.rec_walk[u], S, lane, 64u) || bad;
    if (bad && lane == 0) atomicOr(&flag[0], 1ull);
}
/* ---- chains found by POSITION (round 6).  When the graph comes from packed sequences whose every k-mer is solid, a chain of the graph that lies
 * whole inside one sequence need not be WALKED on the junction table (one dependent random read per junction: 3.0e9 of them, 97 ms at human scale,
 * the rate this memory gives random lines): it is the piece of the sequence between two consecutive junctions that are no chain interior, and those
 * -- the left junction of every chain start, both junctions of every k-mer of no chain: what the scan has just listed -- are few (two per chain).
 * They go into a set (with a bit filter that stays in L2 in front of it); one streaming pass over the sequences asks every junction position of
 * it, cuts each sequence at its hits and turns every piece between two hits into a record of the store, owned as the walkers decide it (the end with the
 * smaller canonical k-mer; none when the two are equal) and claimed once (the same chain may lie in many sequences) in a set of finished chains.
 * The walkers of those chains' starts leave at once (JtWalker::begin); chains that no sequence holds whole -- put together from overlapping
 * sequences -- are walked as before.  The sequence of such a record is copied from the input (k_us_from_seq), not from chunks. */
struct PackedSeqs {
    const uint64_t* words;    /* 2-bit nucleotides, 32 a word, every sequence from a word boundary */
    const uint64_t* word_off; /* per sequence: its first word */
    const uint32_t* len;      /* per sequence: nucleotides */
    size_t nseq;
    uint32_t longest;         /* the longest of them (0: not known) */
};
struct PosSets {
    KeySet stops;         /* canonical junctions that are no chain interior */
    uint32_t* filter;     /* one bit per hash value of a stop */
    uint32_t filter_shift; /* 32 - log2(bits) */
    KeySet done;          /* canonical end k-mers of the claimed chains */
};
/* the filter is a blocked one: a stop sets three bits of ONE 32-bit word (one load a question; with two stops a word on average one question in
 * two hundred is answered "perhaps" -- with one bit it was one in sixteen, and a wave of 64 then went to the set itself on nearly every turn).
 * Four 32-bit multiplications: a 64-bit mix is eight, and this is asked of every junction position. */
struct PosProbe { uint32_t word, mask; };
__device__ __forceinline__ PosProbe pos_filter_probe(const PosSets& P, uint64_t key)
{
    uint32_t h = ((uint32_t)key * 0x9E3779B1u) ^ ((uint32_t)(key >> 32) * 0x85EBCA77u);
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    const uint32_t g = (h * 0x297A2D39u) >> 17; /* fifteen bits: three places in the word */
    PosProbe p;
    p.word = h >> P.filter_shift;
    p.mask = (1u << (g & 31u)) | (1u << ((g >> 5) & 31u)) | (1u << (g >> 10));
    return p;
}
/* false: the set could not take the key (it never happens at half load; a set without one of its stops would merge two chains, so the host then walks every chain) */
__device__ __forceinline__ bool pos_add_stop(const PosSets& P, uint64_t jf, uint64_t jr)
{
    const uint64_t key = jf <= jr ? jf : jr;
    if (keyset_put(P.stops, key) == 2) return false;
    const PosProbe pr = pos_filter_probe(P, key);
    atomicOr(&P.filter[pr.word], pr.mask);
    return true;
}
__global__ void __launch_bounds__(256) k_pos_stops(PosSets P, int k, const uint64_t* __restrict__ starts, unsigned long long n_starts, const uint64_t* __restrict__ left_k, unsigned long long n_left,
                                                   unsigned long long* failed)
{
    const uint64_t mk1 = kmask(k - 1);
    bool ok = true;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_starts + n_left; i += (unsigned long long)gridDim.x * blockDim.x) {
        const bool lone = i >= n_starts;
        const Kmer x = make_kmer(lone ? left_k[i - n_starts] : starts[i], k);
        ok = pos_add_stop(P, x.f >> 2, x.r & mk1) && ok;           /* the junction on its left */
        if (lone) ok = pos_add_stop(P, x.f & mk1, x.r >> 2) && ok; /* a k-mer of no chain: the one on its right as well */
    }
    if (!ok) atomicOr(failed, 1ull);
}
enum { POS_R = 16, POS_TILE = 256 * POS_R, POS_Q = 512, POS_CHUNK = 1 << 17 /* positions: 32 tiles */ };
struct PosPiece {
    uint64_t start_f, src; /* first k-mer in the stored orientation; (first nucleotide in the input << 1) | against the input */
    uint32_t len_k, pad_;
};
/* the whole workgroup: the queued pieces become records -- their words in the store reserved with one atomic, their numbers with another */
__device__ __forceinline__ void pos_queue_flush(const WalkShared& S, uint64_t* pos_src, int k, const PosPiece* s_pq, uint32_t* s_pn, uint32_t* s_pw, uint32_t* s_ppre, uint32_t* s_wt)
{
    __shared__ unsigned long long s_base[2];
    __shared__ unsigned long long s_views;
    __syncthreads();
    const uint32_t n = *s_pn;
    if (n == 0) return; /* (uniform: every thread read the same count) */
    if (threadIdx.x == 0) s_views = 0;
    for (uint32_t t = threadIdx.x; t < (uint32_t)POS_Q; t += blockDim.x) s_pw[t] = t < n ? (uint32_t)us_words_of(s_pq[t].len_k, k) : 0u;
    __syncthreads();
    block_prefix(s_pw, s_ppre, (uint32_t)POS_Q, s_wt);
    unsigned long long v = 0;
    for (uint32_t t = threadIdx.x; t < n; t += blockDim.x) v += 2ull * (s_pq[t].len_k - 1u);
    if (v) atomicAdd(&s_views, v);
    __syncthreads();
    if (threadIdx.x == 0) {
        s_base[0] = atomicAdd(&S.counters[JT_C_RECS], (unsigned long long)n);
        s_base[1] = atomicAdd(&S.counters[JT_C_WORDS], (unsigned long long)s_ppre[n - 1] + s_pw[n - 1]);
        atomicAdd(&S.counters[JT_C_STORED_VIEWS], s_views);
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < n; t += blockDim.x) {
        const uint64_t r = s_base[0] + t;
        if (r < S.rec_cap) {
            S.rec[r].start_f = s_pq[t].start_f; S.rec[r].len_k = s_pq[t].len_k; S.rec[r].pad_ = 0; S.rec[r].hdr = s_base[1] + s_ppre[t];
            S.rec_walk[r] = (uint64_t)REC_BY_POSITION | (0xFFFFFFFFull << 32);
            pos_src[r] = s_pq[t].src;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) *s_pn = 0;
    __syncthreads();
}
/* a lane takes POS_R consecutive junction positions: the first junction from the words, the next ones by one nucleotide each (two shifts instead of a
 * reverse complement a position), the filter's bit from a hash of three 32-bit multiplications */
__global__ void __launch_bounds__(256) k_pos_plan(PosSets P, PackedSeqs in, int k, WalkShared S, uint64_t* __restrict__ pos_src)
{
    __shared__ uint32_t s_list[POS_TILE + 1]; /* [0]: the last stop of the tiles before (or none), then this tile's stops in order */
    __shared__ uint32_t s_wc[4];
    __shared__ uint32_t s_prev;
    __shared__ PosPiece s_pq[POS_Q];          /* claimed pieces that wait for their records */
    __shared__ uint32_t s_pw[POS_Q], s_ppre[POS_Q], s_wc2[16];
    __shared__ uint32_t s_pn;
    if (threadIdx.x == 0) s_pn = 0;
    __syncthreads();
    const uint64_t mk = kmask(k), mk1 = kmask(k - 1), cmpl = 0xAAAAAAAAAAAAAAAAULL & mk, cmpl1 = 0xAAAAAAAAAAAAAAAAULL & mk1;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t cap = MTG_US_MAX_LEN - (uint32_t)k;
    const uint32_t top = 2u * ((uint32_t)k - 2u);
    for (size_t s = blockIdx.x; s < in.nseq; s += gridDim.x) {
        const uint32_t L = in.len[s];
        if (L < (uint32_t)k) continue;
        const uint64_t* __restrict__ w = in.words + in.word_off[s];
        const uint32_t npos = L - (uint32_t)k + 2u;
        /* a sequence is taken in chunks of POS_CHUNK positions, chunk c by the workgroups with blockIdx.y == c mod gridDim.y (one chunk, gridDim.y == 1,
         * unless the host has seen a long sequence): a chunk's workgroup owns the pieces that BEGIN at a stop of its chunk, and goes on past the chunk's
         * end until it has met the stop that ends the last of them (have_prev and total are the same in every thread) */
        for (uint32_t c = blockIdx.y; (uint64_t)c * POS_CHUNK < npos; c += gridDim.y) {
        const uint32_t q_begin = c * (uint32_t)POS_CHUNK, q_own_end = npos - q_begin < (uint32_t)POS_CHUNK ? npos : q_begin + (uint32_t)POS_CHUNK;
        bool have_prev = false;
        if (threadIdx.x == 0) s_prev = 0xFFFFFFFFu;
        __syncthreads();
        for (uint32_t q0 = q_begin; q0 < npos; q0 += POS_TILE) {
            const bool past = q0 >= q_own_end;
            if (past && !have_prev) break;
            const uint32_t qb = q0 + threadIdx.x * POS_R;
            uint32_t stops = 0, cand = 0;
            if (qb < npos) {
                /* the junction at qb: nucleotides qb .. qb + k - 2; then the nucleotides that follow it, one a position */
                uint64_t jr = us_peek64(w, qb, (uint32_t)k - 1u, false) ^ cmpl1, jf = revcomp(jr, k - 1);
                const uint32_t a = qb + (uint32_t)k - 1u;
                const uint32_t nnext = a < L ? (L - a < POS_R ? L - a : POS_R) : 0u;
                const uint64_t next = nnext ? us_peek64(w, a, nnext, false) : 0ull;
                const uint32_t nq = npos - qb < POS_R ? npos - qb : POS_R;
#pragma unroll
                for (uint32_t r = 0; r < POS_R; r++) {
                    if (r < nq) {
                        const uint64_t key = jf <= jr ? jf : jr;
                        const PosProbe pr = pos_filter_probe(P, key);
                        if ((P.filter[pr.word] & pr.mask) == pr.mask) cand |= 1u << r; /* (no branch, no second look here: the sixteen loads are in flight together) */
                        const uint32_t c = (uint32_t)(next >> (2u * r)) & 3u;
                        jf = ((jf << 2) | c) & mk1;
                        jr = (jr >> 2) | ((uint64_t)(c ^ 2u) << top);
                    }
                }
            }
            /* the filter's "perhaps" to the set of stops: a few a wave and tile */
            while (__ballot(cand != 0u)) {
                if (cand) {
                    const uint32_t r = (uint32_t)__builtin_ctz(cand);
                    cand &= cand - 1u;
                    const uint64_t jr = us_peek64(w, qb + r, (uint32_t)k - 1u, false) ^ cmpl1, jf = revcomp(jr, k - 1);
                    if (keyset_has(P.stops, jf <= jr ? jf : jr)) stops |= 1u << r;
                }
            }
            /* the tile's stops in the order of their positions: lane after lane */
            const uint32_t c = (uint32_t)__popc(stops);
            uint32_t inc = c;
            if (__ballot(c != 0u)) {
                for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64); if (lane >= (uint32_t)d) inc += o; }
            }
            if (lane == 63u) s_wc[wave] = inc;
            __syncthreads();
            uint32_t before = 0, total = 0;
            for (uint32_t v = 0; v < 4; v++) { const uint32_t t = s_wc[v]; if (v < wave) before += t; total += t; }
            {
                uint32_t at = 1u + before + inc - c;
                for (uint32_t rest = stops; rest; rest &= rest - 1u) s_list[at++] = qb + (uint32_t)__builtin_ctz(rest);
            }
            if (threadIdx.x == 0) s_list[0] = s_prev;
            __syncthreads();
            if (past && total > 1u) total = 1u; /* beyond the chunk only the stop that ends its last piece counts */
            /* the pieces: k-mers a .. b - 1 between the consecutive stops a < b.  A claimed piece waits in the workgroup's queue (LDS); the queue is emptied with
             * three atomics for all it holds (pos_queue_flush) instead of three a piece: the counters share a memory line, and a line takes 150 M atomics a
             * second -- 1.8e6 of them were 6 of this pass's 24 ms */
            for (uint32_t i0 = 0; i0 < total; i0 += 256u) {
                const uint32_t waiting = s_pn; /* every thread reads the count BEFORE any thread of this round adds to it: the barrier keeps the test uniform */
                __syncthreads();
                if (waiting + 256u > (uint32_t)POS_Q) pos_queue_flush(S, pos_src, k, s_pq, &s_pn, s_pw, s_ppre, s_wc2);
                const uint32_t i = i0 + threadIdx.x;
                if (i < total) {
                    const uint32_t a = s_list[i], b = s_list[i + 1];
                    const uint32_t len_k = b - a;
                    /* (one k-mer: the scan has it; too long for a unitig: the walkers and the late pass, as ever) */
                    if (a != 0xFFFFFFFFu && len_k >= 2u && len_k < cap) {
                        Kmer X, Y;
                        X.r = us_peek64(w, a, (uint32_t)k, false) ^ cmpl; X.f = revcomp(X.r, k);
                        Y.r = us_peek64(w, b - 1u, (uint32_t)k, false) ^ cmpl; Y.f = revcomp(Y.r, k);
                        const uint64_t cX = canon(X), cY = canon(Y);
                        const bool fwd = cX < cY; /* the chain is stored from the end with the smaller canonical k-mer; equal: no walker owns such a chain either */
                        /* claimed from another sequence already (or the set is full: then the walkers do it)? */
                        if (cX != cY && keyset_insert(P.done, fwd ? cX : cY)) {
                            (void)keyset_insert(P.done, fwd ? cY : cX);
                            const uint32_t at = atomicAdd(&s_pn, 1u);
                            PosPiece pc;
                            pc.start_f = fwd ? X.f : Y.r; pc.src = ((in.word_off[s] * 32ull + a) << 1) | (fwd ? 0ull : 1ull); pc.len_k = len_k;
                            s_pq[at] = pc;
                        }
                    }
                }
                __syncthreads();
            }
            __syncthreads();
            if (threadIdx.x == 0 && total) s_prev = s_list[total];
            /* (the next tile's first barrier orders this store before s_list[0] is written from it) */
            have_prev = have_prev || total != 0u;
            if (past && total) break;
        }
        __syncthreads();
        }
    }
    pos_queue_flush(S, pos_src, k, s_pq, &s_pn, s_pw, s_ppre, s_wc2);
}
/* one record found by position per wave: header word and sequence from the packed input, as it stands or reverse-complemented */
__global__ void __launch_bounds__(256) k_us_from_seq(UStore us, int k, const UsRec* __restrict__ rec, unsigned long long n, const uint64_t* __restrict__ rec_walk,
                                                     const uint64_t* __restrict__ pos_src, const uint64_t* __restrict__ words)
{
    const unsigned long long wave = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((unsigned long long)gridDim.x * blockDim.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    for (unsigned long long u = wave; u < n; u += nwaves) {
        if ((uint32_t)rec_walk[u] != REC_BY_POSITION) continue;
        const UsRec r = rec[u];
        const uint64_t L = (uint64_t)r.len_k + (uint32_t)k - 1;
        const uint64_t nt0 = pos_src[u] >> 1;
        const bool rev = pos_src[u] & 1ull;
        uint64_t* o = us.words + r.hdr;
        if (lane == 0) o[0] = L;
        for (uint64_t j = lane; 32 * j < L; j += 64) {
            const uint32_t cnt = (uint32_t)(L - 32 * j < 32 ? L - 32 * j : 32);
            o[1 + j] = rev ? us_peek64(words, nt0 + L - 1 - 32 * j, cnt, true) : us_peek64(words, nt0 + 32 * j, cnt, false);
        }
    }
}
/* one stored unitig per wave, its k-mers dealt to the lanes: abundances from the source into the store.  counters[JT_C_SAT] += those above 255 */
template <typename Src>
__global__ void __launch_bounds__(256) k_us_ab(UStore us, int k, const UsRec* __restrict__ rec, unsigned long long n, Src src, unsigned long long* counters, const uint2* __restrict__ items)
{
    const unsigned long long wave = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((unsigned long long)gridDim.x * blockDim.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t lines = 0;
    unsigned long long sat = 0;
    for (unsigned long long w = wave; w < n; w += nwaves) { /* (items: pieces of unitigs, as in k_sparse_link) */
        const UsRec r = rec[items ? items[w].x : w];
        const uint32_t i0 = items ? items[w].y : 0u, i1 = items ? (r.len_k - i0 < (uint32_t)US_PIECE ? r.len_k : i0 + (uint32_t)US_PIECE) : r.len_k;
        for (uint32_t i = i0 + lane; i < i1; i += 64) sat += us_ab_fill(us, k, r, i, src, lines);
    }
    sat = wave_sum_u64(sat);
    if (lane == 0 && sat) atomicAdd(&counters[JT_C_SAT], sat);
}

/* the solid k-mers and their abundances out of the ABND table (index writer): out_k / out_a receive them in no particular order */
__global__ void k_abnd_export(Index ix, uint64_t* out_k, uint32_t* out_a, unsigned long long* cursor, unsigned long long cap)
{
    const uint64_t nslots = ix.abnd.nbuckets * MTG_ABND_SLOTS;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nslots; s += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t c;
        const uint32_t a = abnd_slot_kmer(ix.abnd, s, c);
        if (!a) continue;
        const unsigned long long at = atomicAdd(cursor, 1ull);
        if (at < cap) { out_k[at] = c; out_a[at] = a; }
    }
}


namespace {
/* device times and memory of an index construction (mtg_index_build_profile) */
struct BuildProf {
    std::vector<mtg_build_phase> phases;
    size_t base_used = 0, peak = 0;
    std::chrono::steady_clock::time_point t0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    BuildProf()
    {
        t0 = std::chrono::steady_clock::now();
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        size_t f = 0, t = 0;
        if (hipMemGetInfo(&f, &t) == hipSuccess) base_used = t - f;
    }
    ~BuildProf() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
    BuildProf(const BuildProf&) = delete;
    /* call after allocations: the most memory held beyond what the device held when the construction began */
    void sample()
    {
        size_t f = 0, t = 0;
        if (hipMemGetInfo(&f, &t) != hipSuccess) return;
        const size_t used = t - f;
        if (used > base_used && used - base_used > peak) peak = used - base_used;
    }
    void begin() { sample(); (void)hipEventRecord(e0, 0); }
    /* closes the phase opened by begin(): waits for the null stream */
    hipError_t end(const char* name, uint64_t bytes, uint64_t units)
    {
        (void)hipEventRecord(e1, 0);
        const hipError_t e = hipEventSynchronize(e1);
        float ms = 0;
        if (e == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
        mtg_build_phase ph{};
        snprintf(ph.name, sizeof ph.name, "%s", name);
        ph.ms = ms; ph.bytes = bytes; ph.units = units;
        phases.push_back(ph);
        if (tune::on(tune::T_DEBUG_TIMERS)) fprintf(stderr, "  [build] %-24s %9.2f ms\n", name, ms);
        return e == hipSuccess ? hipGetLastError() : e;
    }
    void host_phase(const char* name, double ms, uint64_t bytes, uint64_t units)
    {
        mtg_build_phase ph{};
        snprintf(ph.name, sizeof ph.name, "%s", name);
        ph.ms = ms; ph.bytes = bytes; ph.units = units;
        phases.push_back(ph);
    }
    double alloc0 = tl_alloc_ms;
    void store(mtg_index* idx)
    {
        sample();
        host_phase("hipMalloc+hipFree (host wall)", tl_alloc_ms - alloc0, 0, 0);
        idx->build_phases = phases;
        idx->build_peak_bytes = peak;
        idx->build_total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
};
} // namespace

static void free_tables(mtg_index* idx)
{
    if (idx->dev.adj.slots) (void)hipFree(idx->dev.adj.slots);
    if (idx->dev.abnd.slots) (void)hipFree(idx->dev.abnd.slots);
    if (idx->dev.bloom.bits) (void)hipFree(idx->dev.bloom.bits);
    if (idx->dev.us.words) (void)hipFree(idx->dev.us.words);
    if (idx->dev.us.ab) (void)hipFree(idx->dev.us.ab);
    idx->dev.adj.slots = idx->dev.abnd.slots = nullptr;
    idx->dev.bloom.bits = nullptr;
    idx->dev.us = UStore{};
}
namespace {
/* an index under construction: tables and handle go away unless the build hands it over */
struct IndexGuard {
    mtg_index* idx;
    explicit IndexGuard(mtg_index* i) : idx(i) {}
    ~IndexGuard() { if (idx) { free_tables(idx); delete idx; } }
    mtg_index* release() { mtg_index* i = idx; idx = nullptr; return i; }
};
} // namespace

namespace {
/* the k-mers of no stored unitig of a construction that can only name them once the unitigs' entries are in the new tables (lean build, a
 * closed or over-long chain in the graph): called with the sparse ADJ table holding every unitig pointer; fills the lists */
struct LateLeftovers {
    unsigned long long n_upper = 0; /* bound on their number, for the shape of the tables */
    std::function<int(const Index& nx, DevBuf& left_k, DevBuf& left_a, unsigned long long& n_left)> collect;
};
} // namespace
__global__ void k_rec_longest(const UsRec* __restrict__ rec, unsigned long long n, unsigned long long* out)
{
    unsigned long long m = 0;
    for (unsigned long long u = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; u < n; u += (unsigned long long)gridDim.x * blockDim.x) m = rec[u].len_k > m ? rec[u].len_k : m;
    if (m) atomicMax(out, m);
}
/* the (unitig, first k-mer) pieces of a store with a LONG unitig (US_LONG k-mers or more) -- empty when it has none: the kernels then take whole unitigs */
static int unitig_pieces(const std::vector<UsRec>& h_rec, DevBuf& d_items, unsigned long long& n_items)
{
    n_items = 0;
    bool any_long = false;
    for (const UsRec& r : h_rec) any_long = any_long || r.len_k >= (uint32_t)US_LONG;
    if (!any_long) return MTG_OK;
    std::vector<uint2> items;
    for (size_t u = 0; u < h_rec.size(); u++)
        for (uint32_t b = 0; b < h_rec[u].len_k; b += (uint32_t)US_PIECE) { uint2 it; it.x = (uint32_t)u; it.y = b; items.push_back(it); }
    HIP_TRY(d_items.alloc(items.size() * sizeof(uint2)));
    HIP_TRY(hipMemcpy(d_items.p, items.data(), items.size() * sizeof(uint2), hipMemcpyHostToDevice));
    n_items = items.size();
    return MTG_OK;
}
static int sparsify(mtg_index* idx, const UsRec* d_rec, unsigned long long n_rec, bool from_container, const uint64_t* d_left_k, const uint32_t* d_left_a, unsigned long long n_left,
                    BuildProf* prof = nullptr, const LateLeftovers* late = nullptr, DevBuf* adj_reuse = nullptr);
/* The sparse form, from the unitig store (idx holds only the store: the lean build, or an index out of its container) and the k-mers of no
 * unitig, which are handed over; the Bloom filter is filled here.  New tables: ADJ with the entries the sparse form keeps, ABND with the
 * k-mers of no unitig.  (from_container is true for every caller since round 5: the build that derived the store from dense tables is gone.) */
static int sparsify(mtg_index* idx, const UsRec* d_rec, unsigned long long n_rec, bool from_container, const uint64_t* d_left_k, const uint32_t* d_left_a, unsigned long long n_left,
                    BuildProf* prof, const LateLeftovers* late, DevBuf* adj_reuse)
{
    const int k = idx->dev.k;
    DevBuf d_cnt, own_k, own_a;
    HIP_TRY(d_cnt.alloc(64));
    const unsigned long long n_left_shape = late ? std::max(late->n_upper, n_left) : n_left;
    /* entries of the new ADJ: per unitig its kept interior junctions (every second one and the last) and its two ends; two per k-mer of no unitig */
    uint64_t nkeys = 2 * n_left_shape + 1024, n_unitig_kmers = 0;
    DevBuf d_items;
    unsigned long long n_items = 0;
    {
        std::vector<UsRec> h_rec(n_rec);
        if (n_rec) HIP_TRY(hipMemcpy(h_rec.data(), d_rec, n_rec * sizeof(UsRec), hipMemcpyDeviceToHost));
        for (const UsRec& r : h_rec) { nkeys += r.len_k / 2 + 3; n_unitig_kmers += r.len_k; }
        if (int rc2 = unitig_pieces(h_rec, d_items, n_items)) return rc2;
    }
    Index old = idx->dev;
    double load = 1.0;
    int rc = MTG_OK;
    for (int attempt = 0; attempt < 6; attempt++) {
        Index nx = old;
        nx.adj.sp_words = nullptr; /* raw look-ups while the tables are being built */
        const double load_adj = tune::f(tune::T_SPARSE_ADJ_LOAD, 0.49) * load; /* 0.7 overflows the displacement range at human scale (2-slot buckets) and cost a second attempt; 0.49 is what that attempt ran at */
        table_shape(nx.adj, buckets_for(nkeys, load_adj, 2 * (k - 1), MTG_ADJ_SLOTS), 2 * (k - 1));
        table_shape(nx.abnd, buckets_for(n_left_shape + 1024, 0.6 * load, 2 * k, MTG_ABND_SLOTS), 2 * k);
        const size_t ba = nx.adj.nbuckets * 16 * MTG_ADJ_SLOTS, bb = nx.abnd.nbuckets * 8 * MTG_ABND_SLOTS;
        DevBuf na, nb;
        /* the junction table has served and is large enough: its memory becomes the ADJ table (a second allocation of this size after freeing the
         * first cost 2 s of hipMalloc on the GPU box: the freed memory is scrubbed before it is handed out again) */
        if (adj_reuse && adj_reuse->p && adj_reuse->cap >= ba) na.adopt(*adj_reuse);
        else { if (adj_reuse && adj_reuse->p) (void)adj_reuse->alloc(0); HIP_TRY(na.alloc(ba)); }
        const size_t ba_held = na.cap; /* a table that took over the junction table's memory holds all of it */
        HIP_TRY(nb.alloc(bb));
        if (prof) prof->begin();
        HIP_TRY(hipMemsetAsync(na.p, 0, ba, 0));
        HIP_TRY(hipMemsetAsync(nb.p, 0, bb, 0));
        if (prof) HIP_TRY(prof->end("clear_sparse_tables", ba + bb, 0));
        nx.adj.slots = na.as<uint64_t>();
        nx.abnd.slots = nb.as<uint64_t>();
        HIP_TRY(hipMemset(d_cnt.p, 0, 64));
        if (prof) prof->begin();
        if (n_rec) {
            const unsigned long long n_work = n_items ? n_items : n_rec;
            hipLaunchKernelGGL(k_sparse_link, dim3((unsigned)std::min<unsigned long long>(n_work, 256 * 256)), dim3(64), 0, 0, nx, d_rec, n_work, from_container ? 1 : 0, d_cnt.as<unsigned long long>(),
                               n_items ? (const uint2*)d_items.as<uint2>() : (const uint2*)nullptr);
        }
        HIP_TRY(hipGetLastError());
        /* per k-mer: its window of the store (8), every second one an ADJ bucket read and written (2 x 32), with the filter a block (64 + 64) */
        if (prof) HIP_TRY(prof->end(from_container ? "sparse_link+bloom" : "sparse_link", n_unitig_kmers * (8 + 32 + (from_container && nx.bloom.bits ? 128 : 0)), n_unitig_kmers));
        DevBuf late_k, late_a;
        if (late) {
            unsigned long long cnt0[8];
            HIP_TRY(hipMemcpy(cnt0, d_cnt.p, 64, hipMemcpyDeviceToHost));
            if (cnt0[0]) { load *= 0.7; rc = MTG_ERR_OVERFLOW; set_error("index bucket displacement overflow (sparse form)"); continue; }
            if (prof) prof->begin();
            Index nxs = nx;
            nxs.adj.sp_words = nx.us.words; /* the look-ups of kmer_stored go through the pointers */
            if (int rc2 = late->collect(nxs, late_k, late_a, n_left)) return rc2;
            if (n_left > n_left_shape) { set_error("more k-mers outside the unitigs (%llu) than the junction table accounts for (%llu)", n_left, n_left_shape); return MTG_ERR_OVERFLOW; }
            d_left_k = late_k.as<uint64_t>();
            d_left_a = late_a.as<uint32_t>();
            if (prof) HIP_TRY(prof->end("late_leftovers", 0, n_left));
        }
        if (prof) prof->begin();
        if (n_left) {
            Index nxb = nx;
            if (!from_container) nxb.bloom.bits = nullptr; /* the filter holds every k-mer already */
            hipLaunchKernelGGL(k_insert_kmers, dim3((unsigned)std::min<unsigned long long>((n_left + 255) / 256 + 1, 256 * 16)), dim3(256), 0, 0, nxb, d_left_k, d_left_a, (size_t)n_left, d_cnt.as<unsigned long long>());
        }
        HIP_TRY(hipGetLastError());
        unsigned long long cnt[8];
        HIP_TRY(hipMemcpy(cnt, d_cnt.p, 64, hipMemcpyDeviceToHost));
        if (cnt[0]) { load *= 0.7; rc = MTG_ERR_OVERFLOW; set_error("index bucket displacement overflow (sparse form)"); continue; }
        /* lookaheads of the entries that are no pointers: around the k-mers of no unitig and at the unitigs' ends */
        if (n_left) hipLaunchKernelGGL(k_lookahead_kmers, dim3((unsigned)std::min<unsigned long long>((n_left + 255) / 256 + 1, 256 * 16)), dim3(256), 0, 0, nx, d_left_k, (size_t)n_left);
        if (n_rec) hipLaunchKernelGGL(k_sparse_ends, dim3((unsigned)std::min<unsigned long long>((n_rec + 255) / 256, 256 * 16)), dim3(256), 0, 0, nx, d_rec, n_rec);
        HIP_TRY(hipGetLastError());
        if (prof) HIP_TRY(prof->end("leftovers+ends", n_left * 5 * 32 + n_rec * 2 * 16 * 32, n_left + 2 * n_rec));
        HIP_TRY(hipDeviceSynchronize());
        /* the new tables take the place of the old ones */
        if (old.adj.slots) (void)timed_free(old.adj.slots);
        if (old.abnd.slots) (void)timed_free(old.abnd.slots);
        nx.adj.sp_words = nx.us.words;
        idx->dev = nx;
        (void)na.release();
        (void)nb.release();
        idx->info.device_bytes = std::max(ba, ba_held) + bb + idx->dev.bloom.nblocks * 64 + idx->info.unitig_bytes;
        idx->info.adj_buckets = nx.adj.nbuckets;
        idx->info.abnd_buckets = nx.abnd.nbuckets;
        idx->info.sparse = nx.us.words ? 1 : 0; /* no stored unitig: every k-mer has its full entries, the look-ups are the raw ones */
        idx->info.nb_kmers_outside_unitigs = n_left;
        return MTG_OK;
    }
    return rc;
}

/* ---- the lean build: Graph::create without dense tables (mtg_dev.h: "the lean build").  jt = the filled junction table (in jt_buf), src = where
 * the abundances are asked; release_source() frees what src reads once the store holds every abundance.  Leaves idx with the unitig store,
 * the sparse tables derived from it, the Bloom filter and the graph's statistics. */
template <typename Src>
static int build_from_jt(mtg_index* idx, DevBuf& jt_buf, const Table& jt, const Src& src, const std::function<void()>& release_source, uint64_t sat_at_insert, BuildProf& prof,
                         const PackedSeqs* packed = nullptr /* the sequences the table was made from, every k-mer of theirs solid: chains are looked for by position first */)
{
    const int k = idx->dev.k;
    const uint64_t nslots = jt.nbuckets * MTG_ABND_SLOTS;
    const unsigned scan_blocks = (unsigned)std::min<uint64_t>((nslots + 255) / 256, 256 * 32);
    /* What the construction needs only while the junction table is alive -- chain starts, the chunks, the marks, the walkers' arrays -- goes
     * behind the table in ITS buffer, which is sized for the sparse ADJ table that takes it over (48.8 GB for a 34.2 GB table at human scale):
     * no allocation (on this pool a hipMalloc that takes the process past what it has touched before can cost half a second), and the
     * construction's peak stays the index's resident size.  A piece that does not fit gets memory of its own. */
    struct Behind {
        char* p; size_t cap, used = 0;
        std::vector<std::unique_ptr<DevBuf>> own;
        hipError_t take(void** out, size_t bytes)
        {
            bytes = (bytes + 255) & ~(size_t)255;
            if (used + bytes <= cap) { *out = p + used; used += bytes; return hipSuccess; }
            own.emplace_back(new DevBuf());
            const hipError_t e = own.back()->alloc(bytes);
            *out = own.back()->p;
            return e;
        }
    } behind{(char*)jt_buf.p + ((nslots * 8 + 255) & ~(size_t)255), jt_buf.cap > ((nslots * 8 + 255) & ~(size_t)255) ? jt_buf.cap - ((nslots * 8 + 255) & ~(size_t)255) : 0};
    DevBuf d_cnt, d_rec, d_left_k, d_left_a;
    uint64_t* p_starts = nullptr;
    HIP_TRY(d_cnt.alloc((JT_C_N + 2) * 8)); /* + the chunk pool's cursor, the compaction's flag */
    HIP_TRY(hipMemset(d_cnt.p, 0, (JT_C_N + 2) * 8));
    unsigned long long* cnt_d = d_cnt.as<unsigned long long>();
    unsigned long long cnt[JT_C_N];
    /* chain starts, k-mers of no chain, statistics: ONE pass with lists of guessed capacity; a graph with more starts than the guess (the cursors
     * count past the capacities) is scanned a second time with lists of the exact sizes */
    unsigned long long n_starts = 0, n_single = 0, interior = 0, sat_single = 0;
    unsigned long long cap_starts = std::max<unsigned long long>(1ull << 20, nslots / 16), cap_left = std::max<unsigned long long>(1ull << 18, nslots / 128);
    {
        size_t f = 0, t = 0; /* the guess never takes more than a quarter of what is free (288 GB: it does not bind; a small device: two passes) */
        if (hipMemGetInfo(&f, &t) == hipSuccess) { cap_starts = std::min<unsigned long long>(cap_starts, f / 4 / 8 + 1024); cap_left = std::min<unsigned long long>(cap_left, f / 8 / 12 + 1024); }
    }
    const unsigned scan_grid = (unsigned)std::min<uint64_t>((jt.nbuckets + 255) / 256, 256 * 32);
    for (int pass = 0; pass < 2; pass++) {
        behind.used = 0; behind.own.clear();
        HIP_TRY(behind.take((void**)&p_starts, (cap_starts + 1) * 8));
        HIP_TRY(d_left_k.alloc((cap_left + 1) * 8));
        HIP_TRY(d_left_a.alloc((cap_left + 1) * 4));
        HIP_TRY(hipMemset(d_cnt.p, 0, JT_C_N * 8));
        prof.begin();
        hipLaunchKernelGGL(k_jt_scan<Src>, dim3(scan_grid), dim3(256), 0, 0, jt, k, src, cnt_d, p_starts, cap_starts, d_left_k.as<uint64_t>(), d_left_a.as<uint32_t>(), cap_left);
        HIP_TRY(prof.end(pass ? "jt_scan_again" : "jt_scan", nslots * 8, nslots));
        HIP_TRY(hipMemcpy(cnt, d_cnt.p, sizeof cnt, hipMemcpyDeviceToHost));
        n_starts = cnt[JT_C_STARTS]; n_single = cnt[JT_C_LEFT]; interior = cnt[JT_C_INTERIOR]; sat_single = cnt[JT_C_SAT];
        if (n_starts <= cap_starts && n_single <= cap_left) break;
        if (pass) { set_error("junction table scan: %llu chain starts and %llu single k-mers do not fit lists sized for them", n_starts, n_single); return MTG_ERR_OVERFLOW; }
        cap_starts = n_starts; cap_left = n_single;
    }
    idx->info.nb_solid_kmers = (cnt[JT_C_ORIENTED] + cnt[JT_C_SELF]) / 2;
    idx->info.nb_branching = (2 * cnt[JT_C_IN_NOT1] - cnt[JT_C_BOTH_NOT1] + cnt[JT_C_SELF_BRANCH]) / 2;
    idx->info.nb_unitigs = 0;
    idx->info.unitig_bytes = 0;
    /* abundances above 255: of the single k-mers (counted by the scan), then of the unitigs' k-mers (k_us_ab), each on its own */
    unsigned long long sat_unitigs = 0, sat_late = 0;
    HIP_TRY(hipMemset(cnt_d + JT_C_SAT, 0, 8));
    unsigned long long n_words = 0, n_rec = 0, stored_views = 0;
    uint64_t *p_pool = nullptr, *p_rec_walk = nullptr, *p_marks = nullptr, *p_wchunk = nullptr;
    uint32_t* p_wcnt = nullptr;
    uint64_t* p_pos_src = nullptr;
    bool positional = false;
    WalkShared WS{};
    ChunkPool& pool = WS.pool;
    if (n_starts) {
        if (n_starts > 0xFFFFFFF0ull) { set_error("unitig construction: %llu chain starts", n_starts); return MTG_ERR_OVERFLOW; } /* a walker is a 32-bit number */
        const unsigned long long rec_cap = n_starts / 2 + 1; /* two starts per stored unitig (one per strand) */
        HIP_TRY(d_rec.alloc(rec_cap * sizeof(UsRec)));
        HIP_TRY(behind.take((void**)&p_rec_walk, rec_cap * 8));
        HIP_TRY(behind.take((void**)&p_wchunk, n_starts * 8));
        HIP_TRY(behind.take((void**)&p_wcnt, n_starts * 4));
        /* the chunks of every walk (at worst both ends of every chain walk all of it: interior + n_starts steps in all) */
        pool.cap_chunks = chunk_pool_need(interior + n_starts, n_starts, k);
        HIP_TRY(behind.take((void**)&p_pool, pool.cap_chunks * MTG_CHUNK_WORDS * 8));
        pool.words = p_pool;
        HIP_TRY(hipMemset(cnt_d + JT_C_N, 0, 16));
        pool.cursor = cnt_d + JT_C_N;
        /* the marks: one per JT_MARK_EVERY crossings at worst, in a table at most half full */
        uint64_t mcap = 1024;
        while (mcap < 2 * ((interior + n_starts) / JT_MARK_EVERY + n_starts)) mcap <<= 1;
        HIP_TRY(behind.take((void**)&p_marks, mcap * 16));
        WS.jt = jt; WS.k = k;
        WS.marks.keys = p_marks; WS.marks.vals = p_marks + mcap; WS.marks.mask = mcap - 1;
        WS.starts = p_starts; WS.counters = cnt_d; WS.rec = d_rec.as<UsRec>(); WS.rec_walk = p_rec_walk; WS.rec_cap = rec_cap;
        WS.w_chunk = p_wchunk; WS.w_cnt = p_wcnt;
        prof.begin();
        HIP_TRY(hipMemsetAsync(p_marks, 0xFF, mcap * 16, 0));
        HIP_TRY(prof.end("clear_marks", mcap * 16, 0));
        positional = packed && packed->nseq && (tune::is_set(tune::T_BUILD_POSITIONAL) ? tune::on(tune::T_BUILD_POSITIONAL) : true);
        if (positional) {
            /* the chains that lie whole in one sequence, by position (k_pos_plan); what is left is walked */
            const unsigned long long n_stops = n_starts + 2 * n_single;
            uint64_t scap = 1024, dcap = 1024, fbits = 1ull << 16;
            while (scap < 2 * n_stops) scap <<= 1;
            while (dcap < 2 * n_starts) dcap <<= 1;
            uint32_t flog = 16;
            while (fbits < 16 * n_stops && fbits < (1ull << 31)) { fbits <<= 1; flog++; } /* 4 MB at human scale: it stays in every L2 */
            PosSets PS{};
            HIP_TRY(behind.take((void**)&PS.stops.keys, scap * 8));
            HIP_TRY(behind.take((void**)&PS.done.keys, dcap * 8));
            HIP_TRY(behind.take((void**)&PS.filter, fbits / 8));
            HIP_TRY(behind.take((void**)&p_pos_src, rec_cap * 8));
            PS.stops.mask = scap - 1; PS.done.mask = dcap - 1; PS.filter_shift = 32 - (flog - 5); /* the word of a key: the top bits of its hash */
            prof.begin();
            HIP_TRY(hipMemsetAsync(PS.stops.keys, 0, scap * 8, 0));
            HIP_TRY(hipMemsetAsync(PS.done.keys, 0, dcap * 8, 0));
            HIP_TRY(hipMemsetAsync(PS.filter, 0, fbits / 8, 0));
            HIP_TRY(hipMemsetAsync(cnt_d + JT_C_N + 1, 0, 8, 0)); /* (the compaction's flag, free until then) */
            hipLaunchKernelGGL(k_pos_stops, dim3((unsigned)std::min<unsigned long long>((n_stops + 255) / 256, 256 * 16)), dim3(256), 0, 0, PS, k, (const uint64_t*)p_starts, n_starts, (const uint64_t*)d_left_k.as<uint64_t>(), n_single,
                               cnt_d + JT_C_N + 1);
            HIP_TRY(prof.end("pos_stops", n_stops * 24 + scap * 8 + dcap * 8 + fbits / 8, n_stops));
            unsigned long long stops_failed = 0;
            HIP_TRY(hipMemcpy(&stops_failed, cnt_d + JT_C_N + 1, 8, hipMemcpyDeviceToHost));
            if (stops_failed) positional = false; /* a stop is missing from the set: every chain is walked */
            else {
                prof.begin();
                const unsigned chunks = packed->longest > (uint32_t)POS_CHUNK ? (unsigned)std::min<uint64_t>(((uint64_t)packed->longest + POS_CHUNK - 1) / POS_CHUNK, 64) : 1u;
                hipLaunchKernelGGL(k_pos_plan, dim3((unsigned)std::min<size_t>(packed->nseq, 256 * 8), chunks), dim3(256), 0, 0, PS, *packed, k, WS, p_pos_src);
                HIP_TRY(prof.end("pos_plan", (interior / 2 + n_starts) / 4 + n_starts * 40, interior / 2 + n_starts)); /* the sequences once; per chain its record and its two claims */
                WS.done = PS.done;
            }
        }
        prof.begin();
        hipLaunchKernelGGL(k_jt_walk, dim3((unsigned)((n_starts + 63) / 64)), dim3(64), 0, 0, WS, n_starts);
        HIP_TRY(prof.end("jt_walk", (interior / 2 + n_starts) * 32 + (interior / 2 + n_starts) / 4 + ((interior / 2) / JT_MARK_EVERY) * 3 * 64, interior / 2 + n_starts)); /* a chain's junctions once between its two walkers: one bucket per step, a quarter byte written; a mark every 32 */
        if (tune::on(tune::T_DEBUG_TIMERS)) { /* how much of the chains the walkers walked between them (1.0: every junction once) */
            std::vector<uint32_t> wc(n_starts);
            HIP_TRY(hipMemcpy(wc.data(), p_wcnt, n_starts * 4, hipMemcpyDeviceToHost));
            unsigned long long steps = 0;
            for (uint32_t c : wc) steps += c;
            fprintf(stderr, "  [build] walkers: %llu starts, %llu k-mers held in all, %llu interior views / 2 = %.3f of the chains\n", n_starts, steps, interior / 2, (double)steps / (double)(interior / 2 + 1));
        }
        unsigned long long tail[2];
        HIP_TRY(hipMemcpy(cnt, d_cnt.p, sizeof cnt, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(tail, cnt_d + JT_C_N, 16, hipMemcpyDeviceToHost));
        n_words = cnt[JT_C_WORDS]; n_rec = cnt[JT_C_RECS]; stored_views = cnt[JT_C_STORED_VIEWS];
        if (n_rec > rec_cap) { set_error("unitig construction: %llu records for %llu chain starts", n_rec, n_starts); return MTG_ERR_OVERFLOW; }
        if (tail[0] > pool.cap_chunks) { set_error("unitig construction: %llu chunks of sequence for a pool of %llu", tail[0], (unsigned long long)pool.cap_chunks); return MTG_ERR_OVERFLOW; }
    }
    if (n_rec) {
        const unsigned long long pad = 8; /* the coverage pass may look up to 64 + k nucleotides past the end of a unitig */
        HIP_TRY(timed_malloc((void**)&idx->dev.us.words, (n_words + pad) * 8));
        HIP_TRY(timed_malloc((void**)&idx->dev.us.ab, (n_words + pad) * 32));
        prof.begin();
        HIP_TRY(hipMemsetAsync(idx->dev.us.words, 0, (n_words + pad) * 8, 0));
        HIP_TRY(hipMemsetAsync(idx->dev.us.ab, 0, (n_words + pad) * 32, 0));
        idx->dev.us.nwords = n_words;
        idx->dev.us.nunitigs = n_rec;
        HIP_TRY(hipMemset(cnt_d + JT_C_N + 1, 0, 8));
        hipLaunchKernelGGL(k_us_compact, dim3((unsigned)std::min<unsigned long long>((n_rec + 3) / 4, 256 * 64)), dim3(256), 0, 0, idx->dev.us, k, d_rec.as<UsRec>(), n_rec, WS, cnt_d + JT_C_N + 1);
        HIP_TRY(prof.end("us_compact", n_words * 16 + n_words * 40, n_rec)); /* the words out of their chunks and into the store; the clearing of words and abundance bytes */
        if (positional) {
            prof.begin();
            hipLaunchKernelGGL(k_us_from_seq, dim3((unsigned)std::min<unsigned long long>((n_rec + 3) / 4, 256 * 64)), dim3(256), 0, 0, idx->dev.us, k, d_rec.as<UsRec>(), n_rec, (const uint64_t*)p_rec_walk, (const uint64_t*)p_pos_src, packed->words);
            HIP_TRY(prof.end("us_from_seq", n_words * 16, n_rec));
        }
        unsigned long long short_chain = 0;
        HIP_TRY(hipMemcpy(&short_chain, cnt_d + JT_C_N + 1, 8, hipMemcpyDeviceToHost));
        if (short_chain) { set_error("unitig construction: a chunk chain is shorter than its record"); return MTG_ERR_OVERFLOW; }
        behind.own.clear(); /* the pieces behind the table are dead from here on: its buffer becomes the sparse ADJ table below */
        prof.begin();
        {
            DevBuf d_items, d_max;
            unsigned long long n_items = 0, longest = 0;
            HIP_TRY(d_max.alloc(8));
            HIP_TRY(hipMemsetAsync(d_max.p, 0, 8, 0));
            hipLaunchKernelGGL(k_rec_longest, dim3((unsigned)std::min<unsigned long long>((n_rec + 255) / 256, 1024)), dim3(256), 0, 0, d_rec.as<UsRec>(), n_rec, d_max.as<unsigned long long>());
            HIP_TRY(hipMemcpy(&longest, d_max.p, 8, hipMemcpyDeviceToHost));
            if (longest >= (unsigned long long)US_LONG) { /* rare: the records come to the host for the list of pieces */
                std::vector<UsRec> h_rec(n_rec);
                HIP_TRY(hipMemcpy(h_rec.data(), d_rec.p, n_rec * sizeof(UsRec), hipMemcpyDeviceToHost));
                if (int rc2 = unitig_pieces(h_rec, d_items, n_items)) return rc2;
            }
            const unsigned long long n_work = n_items ? n_items : n_rec;
            hipLaunchKernelGGL(k_us_ab<Src>, dim3((unsigned)std::min<unsigned long long>((n_work + 3) / 4, 256 * 64)), dim3(256), 0, 0, idx->dev.us, k, d_rec.as<UsRec>(), n_work, src, cnt_d,
                               n_items ? (const uint2*)d_items.as<uint2>() : (const uint2*)nullptr);
            HIP_TRY(hipDeviceSynchronize()); /* (the pieces' list goes out of scope) */
        }
        HIP_TRY(prof.end("us_abundances", (stored_views / 2 + n_rec) * 9, stored_views / 2 + n_rec));
        HIP_TRY(hipMemcpy(&sat_unitigs, cnt_d + JT_C_SAT, 8, hipMemcpyDeviceToHost));
        idx->info.nb_unitigs = n_rec;
        idx->info.unitig_bytes = (n_words + pad) * 40;
    }
    /* the Bloom filter of the sequence scan: filled from the store and the k-mers of no unitig while the sparse tables are written */
    const double bpk = tune::f(tune::T_BLOOM_BITS, 12.0);
    idx->dev.bloom.bits = nullptr;
    idx->dev.bloom.nblocks = 0;
    if (bpk > 0) bloom_shape(idx->dev.bloom, idx->info.nb_solid_kmers, bpk, k);
    idx->info.bloom_blocks = idx->dev.bloom.nblocks;
    idx->info.bloom_minimizer = (uint32_t)idx->dev.bloom.mm;
    idx->info.adj_bucket_bytes = 16 * MTG_ADJ_SLOTS;
    idx->info.abnd_bucket_bytes = 8 * MTG_ABND_SLOTS;
    const auto alloc_bloom = [&]() -> int {
        if (!idx->dev.bloom.nblocks) return MTG_OK;
        HIP_TRY(timed_malloc((void**)&idx->dev.bloom.bits, idx->dev.bloom.nblocks * 64));
        HIP_TRY(hipMemsetAsync(idx->dev.bloom.bits, 0, idx->dev.bloom.nblocks * 64, 0));
        return MTG_OK;
    };
    int rc;
    if (interior == stored_views) {
        /* every chain is a stored unitig: the k-mers of no unitig are the single ones the scan found; table and source have served */
        prof.sample();
        release_source();
        if (int rc2 = alloc_bloom()) return rc2;
        rc = sparsify(idx, d_rec.as<UsRec>(), n_rec, true, d_left_k.as<uint64_t>(), d_left_a.as<uint32_t>(), n_single, &prof, nullptr, &jt_buf); /* the table's memory is used again */
        (void)jt_buf.alloc(0);
    } else {
        /* a closed chain, or one too long for the offsets of a pointer: its k-mers belong to no unitig, and only the finished pointers tell which */
        if (int rc2 = alloc_bloom()) return rc2;
        (void)d_left_k.alloc(0);
        (void)d_left_a.alloc(0);
        sat_single = 0; /* the late pass finds the single k-mers again */
        LateLeftovers late;
        late.n_upper = n_single + (interior - stored_views) / 2 + n_starts + 16;
        late.collect = [&](const Index& nx, DevBuf& lk, DevBuf& la, unsigned long long& n_left) -> int {
            HIP_TRY(hipMemset(cnt_d + JT_C_LEFT, 0, 8));
            hipLaunchKernelGGL(k_jt_unstored<Src>, dim3(scan_blocks), dim3(256), 0, 0, jt, nx, src, cnt_d, (uint64_t*)nullptr, (uint32_t*)nullptr, 0ull);
            HIP_TRY(hipGetLastError());
            unsigned long long n = 0;
            HIP_TRY(hipMemcpy(&n, cnt_d + JT_C_LEFT, 8, hipMemcpyDeviceToHost));
            HIP_TRY(lk.alloc((n + 1) * 8));
            HIP_TRY(la.alloc((n + 1) * 4));
            HIP_TRY(hipMemset(cnt_d + JT_C_LEFT, 0, 8));
            HIP_TRY(hipMemset(cnt_d + JT_C_SAT, 0, 8)); /* an attempt with more buckets counts again */
            hipLaunchKernelGGL(k_jt_unstored<Src>, dim3(scan_blocks), dim3(256), 0, 0, jt, nx, src, cnt_d, lk.as<uint64_t>(), la.as<uint32_t>(), n);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipDeviceSynchronize());
            n_left = n;
            return MTG_OK;
        };
        rc = sparsify(idx, d_rec.as<UsRec>(), n_rec, true, nullptr, nullptr, 0, &prof, &late);
        HIP_TRY(hipMemcpy(&sat_late, cnt_d + JT_C_SAT, 8, hipMemcpyDeviceToHost));
        prof.sample();
        (void)jt_buf.alloc(0);
        release_source();
    }
    if (rc) return rc;
    idx->info.nb_saturated = sat_at_insert + sat_single + sat_unitigs + sat_late;
    return MTG_OK;
}

static double jt_load() { return tune::f(tune::T_JT_LOAD, 0.7); }
/* a cleared table of MTG_ABND_SLOTS-slot buckets for nkeys keys of key_bits bits */
/* bytes the sparse ADJ table of a graph of n k-mers will take, give or take: half the junctions and a little (sparsify) */
static size_t adj_bytes_estimate(uint64_t n, int k)
{
    return (size_t)buckets_for(n / 2 + n / 128 + 8192, 0.49, 2 * (k - 1), MTG_ADJ_SLOTS) * 16 * MTG_ADJ_SLOTS;
}
/* min_bytes: the buffer is made at least this large (the junction table's memory is handed on to the sparse ADJ table) */
static int alloc_slot_table(Table& t, DevBuf& buf, uint64_t nkeys, double load, uint32_t key_bits, BuildProf& prof, const char* phase, size_t min_bytes = 0, uint64_t min_buckets = 0, bool clear = true)
{
    table_shape(t, std::max<uint64_t>(buckets_for(nkeys, load, key_bits, MTG_ABND_SLOTS), min_buckets), key_bits);
    t.sp_words = nullptr;
    const size_t bytes = t.nbuckets * 8 * MTG_ABND_SLOTS;
    if (!(buf.p && buf.cap >= std::max(bytes, min_bytes))) HIP_TRY(buf.alloc(std::max(bytes, min_bytes)));
    t.slots = buf.as<uint64_t>();
    if (!clear) return MTG_OK; /* (the partitioned construction writes every word of the table itself) */
    prof.begin();
    HIP_TRY(hipMemsetAsync(buf.p, 0, bytes, 0));
    HIP_TRY(prof.end(phase, bytes, 0));
    return MTG_OK;
}

static int index_from_kmer_pieces_lean(size_t n, int k, const KmerFetch& fetch, mtg_index** out)
{
    BuildProf prof;
    IndexGuard g(new mtg_index());
    mtg_index* idx = g.idx;
    idx->dev.k = k;
    HIP_TRY(hipGetDevice(&idx->device));
    DevBuf d_k, d_a, d_cnt, jt_buf, abnd_buf;
    HIP_TRY(d_cnt.alloc(4 * 8));
    const size_t env_piece = (size_t)tune::i(tune::T_LOAD_PIECE, 0); /* test hook: small pieces */
    const size_t piece = std::min<size_t>(std::max<size_t>(n, 1), env_piece ? env_piece : (size_t)1 << 26);
    HIP_TRY(d_k.alloc(piece * 8));
    HIP_TRY(d_a.alloc(piece * 4));
    Table jt{}, abnd{};
    double load = 1.0;
    int rc = MTG_OK;
    unsigned long long cnt[4] = {0, 0, 0, 0};
    for (int attempt = 0; attempt < 6; attempt++) {
        if (int rc2 = alloc_slot_table(jt, jt_buf, n + n / 8 + 1024, jt_load() * load, 2 * (k - 1), prof, "clear_jt", adj_bytes_estimate(n, k), jt_min_buckets(2 * (k - 1)))) return rc2;
        if (int rc2 = alloc_slot_table(abnd, abnd_buf, n, 0.6 * load, 2 * k, prof, "clear_abnd_source")) return rc2;
        HIP_TRY(hipMemset(d_cnt.p, 0, 32));
        double ms = 0;
        for (size_t off = 0; off < n; off += piece) {
            const size_t m = std::min(piece, n - off);
            const uint64_t* hk = nullptr;
            const uint32_t* ha = nullptr;
            if (!fetch(off, m, hk, ha)) return MTG_ERR_IO; /* the source has set the message */
            HIP_TRY(hipMemcpy(d_k.p, hk, m * 8, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(d_a.p, ha, m * 4, hipMemcpyHostToDevice));
            prof.begin();
            hipLaunchKernelGGL(k_jt_insert_kmers, dim3((unsigned)std::min<size_t>((m + 255) / 256 + 1, 256 * 16)), dim3(256), 0, 0, jt, abnd, k, d_k.as<uint64_t>(), d_a.as<uint32_t>(), m, d_cnt.as<unsigned long long>());
            HIP_TRY(prof.end("jt_insert_kmers", 0, m)); /* also: the source may reuse its buffers for the next piece */
            ms += prof.phases.back().ms;
            prof.phases.pop_back();
        }
        prof.host_phase("jt_insert_kmers", ms, (uint64_t)n * (12 + 3 * 64), n); /* a k-mer: its list entry, an ABND bucket and two junction buckets read and written */
        HIP_TRY(hipMemcpy(cnt, d_cnt.p, 32, hipMemcpyDeviceToHost));
        if (!cnt[0]) { rc = MTG_OK; break; }
        load *= 0.7;
        rc = MTG_ERR_OVERFLOW;
        set_error("index bucket displacement overflow");
    }
    if (rc) return rc;
    (void)d_k.alloc(0);
    (void)d_a.alloc(0);
    AbFromTable src;
    src.abnd = abnd;
    if (int rc2 = build_from_jt(idx, jt_buf, jt, src, [&] { (void)abnd_buf.alloc(0); }, cnt[3], prof)) return rc2;
    idx->info.k = k;
    idx->info.abundance_min = 0;
    idx->info.abundance_auto = -1;
    prof.store(idx);
    *out = g.release();
    return MTG_OK;
}

/* the shape of the partitioned construction for a table of at least nb0 buckets; false: the table is too small for it to pay (or k too small for the record) */
static bool bin_shape(uint64_t nb0, uint32_t key_bits, BinShape& S, uint64_t& nb)
{
#ifndef MTG_SEG_MAX_M
#define MTG_SEG_MAX_M 2048
#endif
    const uint64_t max_m = MTG_SEG_MAX_M; /* 2 048: 64 KB of LDS per segment */
    uint32_t b = 0;
    while (((nb0 + (1ull << b) - 1) >> b) > max_m) b++;
    if (b < 8 || b > 21 || key_bits < b + 8) return false;
    S.kb = key_bits;
    S.b1 = std::min<uint32_t>(10, (b + 1) / 2);
    S.b2 = b - S.b1;
    if (key_bits - S.b1 + 8 > 64) return false;
    S.m = (nb0 + (1ull << b) - 1) >> b;
    S.seg_words = S.m * MTG_ABND_SLOTS;
    S.cap1 = 0;
    nb = S.m << b;
    return true;
}
/* jt has the shape bin_shape asked for and is NOT cleared; spare: device memory behind the table (bytes) the level-1 regions may use.
 * Returns MTG_OK with cnt0 = 1 when a key was displaced too far (the caller retries with a lower load), MTG_ERR_OVERFLOW when the overflow list
 * did not hold what the regions could not (the caller falls back to the ordinary insertion). */
static int jt_insert_partitioned(const Table& jt, BinShape S, int k, const uint64_t* d_words, const uint64_t* d_word_off, const uint32_t* d_len, size_t nseq, uint64_t n_junctions_ub,
                                 uint8_t* spare, size_t spare_bytes, unsigned long long* d_cnt, unsigned long long& cnt0, BuildProf& prof)
{
    const uint32_t G = MTG_BIN1_GROUPS; /* workgroups of k_bin_positions = regions per level-1 bin */
    const uint64_t nseg = jt.nbuckets / S.m, nreg = ((uint64_t)1 << S.b1) * G;
    DevBuf d_cur, d_ov, d_tmp;
    /* what the regions and the segments' lists cannot hold goes through the ordinary insertion afterwards: a few in a million for a donor without
     * repeats, but every occurrence of a junction that occurs 300 000 times lands in one segment (scripts/r6_long_sequences.py ... repeats: 10^7 left
     * over).  A sixty-fourth of the positions may be left over before the whole insertion falls back (the ordinary insertion reads a hot entry from
     * cache and finds its bits there: repeats cost it nothing) */
    const unsigned long long ov_cap = std::max<unsigned long long>(1ull << 22, n_junctions_ub / 64);
    HIP_TRY(d_cur.alloc(nreg * 4 + 64));
    HIP_TRY(d_ov.alloc(ov_cap * 16 + 64));
    /* the level-1 regions: behind the table when there is room for at least an eighth of the records at a time, else a buffer of their own */
    uint64_t want = std::min<uint64_t>(n_junctions_ub, 1ull << 29) * 8 * 5 / 4 + nreg * 64 * 8;
    uint8_t* regions = spare;
    size_t reg_bytes = spare_bytes & ~(size_t)63;
    if (reg_bytes < std::min<uint64_t>(want, (n_junctions_ub / 8 + 1) * 8 * 5 / 4 + nreg * 64 * 8)) {
        HIP_TRY(d_tmp.alloc(want));
        regions = d_tmp.as<uint8_t>();
        reg_bytes = want;
    }
    S.cap1 = reg_bytes / 8 / nreg;
    if (S.cap1 < 64) { set_error("partitioned junction table: no room for the level-1 regions"); return MTG_ERR_NOMEM; }
    const uint64_t chunk_positions = S.cap1 * nreg * 5 / 6; /* a region's share of a chunk varies by a few percent (its size is in the thousands) */
    const double per_seq = (double)n_junctions_ub / (double)std::max<size_t>(nseq, 1);
    const size_t seq_per_chunk = (size_t)std::max<double>(1.0, (double)chunk_positions / std::max(per_seq, 1.0));
    BinOverflow ov{d_ov.as<uint64_t>(), reinterpret_cast<unsigned long long*>(d_ov.as<uint8_t>() + ov_cap * 16), ov_cap};
    HIP_TRY(hipMemsetAsync(ov.cursor, 0, 8, 0));
    const size_t lds1 = (size_t)MTG_BIN1_TILE * 9 * (MTG_BIN_SORTED ? 2 : 1) + 3 * 1024 * 4, lds2 = (size_t)BIN_TILE * 8 * (MTG_BIN_SORTED ? 2 : 1) + 3 * 2048 * 4, lds3 = S.seg_words * 8;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bin_positions), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bin_records), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_build_segments), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
    hipLaunchKernelGGL(k_bin_clear_counts, dim3(1024), dim3(256), 0, 0, S, jt.slots, nseg);
    for (size_t s0 = 0; s0 < nseq; s0 += seq_per_chunk) {
        const size_t s1 = std::min(nseq, s0 + seq_per_chunk);
        const uint64_t units = (uint64_t)((double)(s1 - s0) * per_seq);
        prof.begin();
        hipLaunchKernelGGL(k_bin_positions, dim3(G), dim3(MTG_BIN1_THREADS), lds1, 0, S, k, d_words, d_word_off, d_len, s0, s1, reinterpret_cast<uint64_t*>(regions), d_cur.as<uint32_t>(), ov);
        HIP_TRY(prof.end("jt_bin_positions", units * (8 + 1), units)); /* a record written; the nucleotides */
        prof.begin();
        hipLaunchKernelGGL(k_bin_records, dim3(1u << S.b1), dim3(1024), lds2, 0, S, reinterpret_cast<const uint64_t*>(regions), d_cur.as<const uint32_t>(), G, jt.slots, ov);
        HIP_TRY(prof.end("jt_bin_records", units * 16, units)); /* a record read and written */
    }
    prof.begin();
    hipLaunchKernelGGL(k_build_segments, dim3((unsigned)std::min<uint64_t>(nseg, 256 * 8)), dim3(MTG_SEG_THREADS), lds3, 0, S, jt, nseg, ov);
    HIP_TRY(prof.end("jt_build_segments", n_junctions_ub * 8 + jt.nbuckets * 8 * MTG_ABND_SLOTS, nseg)); /* the lists read, the table written */
    unsigned long long n_ov = 0;
    HIP_TRY(hipMemcpy(&n_ov, ov.cursor, 8, hipMemcpyDeviceToHost));
    if (n_ov > ov_cap) { set_error("partitioned junction table: %llu records left over", n_ov); return MTG_ERR_OVERFLOW; }
    if (n_ov) {
        prof.begin();
        hipLaunchKernelGGL(k_jt_overflow, dim3((unsigned)std::min<unsigned long long>((n_ov + 255) / 256, 4096)), dim3(256), 0, 0, jt, ov.h, n_ov, d_cnt);
        HIP_TRY(prof.end("jt_insert_leftover", n_ov * 64, n_ov));
    }
    unsigned long long cnt[4];
    HIP_TRY(hipMemcpy(cnt, d_cnt, 32, hipMemcpyDeviceToHost));
    cnt0 = cnt[0];
    if (tune::on(tune::T_DEBUG_TIMERS)) fprintf(stderr, "  [partitioned jt] segments %llu of %llu buckets, regions of %llu records, leftover %llu, displaced too far %llu (last: hash %llx bits %llx, bucket %llu of %llu)\n", (unsigned long long)nseg, (unsigned long long)S.m, (unsigned long long)S.cap1, n_ov, cnt[2],
                                                cnt[3], cnt[1], (unsigned long long)(((unsigned __int128)cnt[3] * jt.nbuckets) >> jt.key_bits), (unsigned long long)jt.nbuckets);
    if (tune::on(tune::T_DEBUG_TIMERS) && cnt[2]) {
        const uint64_t b = (uint64_t)(((unsigned __int128)cnt[3] * jt.nbuckets) >> jt.key_bits);
        std::vector<uint64_t> w(70 * MTG_ABND_SLOTS);
        const uint64_t bstart = b >= 3 ? b - 3 : 0;
        (void)hipMemcpy(w.data(), jt.slots + bstart * MTG_ABND_SLOTS, w.size() * 8, hipMemcpyDeviceToHost);
        unsigned nz = 0;
        for (uint64_t x : w) nz += x != 0;
        fprintf(stderr, "  [partitioned jt] buckets %llu..: %u of %zu slots occupied; segment %llu, bucket %llu in it; words:", (unsigned long long)bstart, nz, w.size(), (unsigned long long)(b / S.m), (unsigned long long)(b % S.m));
        for (int i = 0; i < 24; i++) fprintf(stderr, " %llx", (unsigned long long)w[i]);
        fprintf(stderr, "\n");
        unsigned long long c0 = 0;
        (void)c0;
    }
    return MTG_OK;
}

static int index_from_packed_device_lean(const uint64_t* d_words, const uint64_t* d_word_off, const uint32_t* d_len, size_t nseq, uint64_t total_kmers_ub, int k,
                                         uint32_t abund_lo, uint32_t abund_span, mtg_index** out)
{
    BuildProf prof;
    IndexGuard g(new mtg_index());
    mtg_index* idx = g.idx;
    idx->dev.k = k;
    HIP_TRY(hipGetDevice(&idx->device));
    DevBuf d_cnt, jt_buf, d_split_off, d_split_len;
    HIP_TRY(d_cnt.alloc(32));
    const uint64_t* ins_word_off = d_word_off; /* what the INSERTION streams: the sequences, the long ones in pieces */
    const uint32_t* ins_len = d_len;
    size_t ins_nseq = nseq;
    /* LONG sequences are taken in pieces.  The kernels that stream the sequences give a sequence to a workgroup (k_bin_positions, k_jt_insert_packed,
     * k_pos_plan): a donor of a few chromosome-sized sequences would keep a few workgroups busy and the rest of the device idle.  A piece of a
     * sequence that starts at a multiple of 32 nucleotides (a word boundary) and shares its last k - 1 nucleotides with the next piece is a sequence
     * like any other, the pieces have the sequence's k-mers exactly once, and the junction they share gets its two sides from the two pieces (the
     * table ORs them): the same graph.  Only the insertion sees the pieces: the chains are looked for in the sequences as they came (a chain that
     * crosses a cut would lie whole in no piece and be left to its walkers -- 142 s for 24 chains of 125 Mbp, scripts/r6_long_sequences.py).
     * The lengths come to the host for this (4 bytes a sequence). */
    uint32_t longest = 0;
    if (nseq) {
        enum : uint32_t { PIECE = 1u << 16 };
        std::vector<uint32_t> h_len(nseq);
        HIP_TRY(hipMemcpy(h_len.data(), d_len, nseq * 4, hipMemcpyDeviceToHost));
        for (uint32_t L : h_len) longest = std::max(longest, L);
        if (longest > 2u * PIECE && !tune::on(tune::T_NO_SPLIT_LONG)) {
            std::vector<uint64_t> h_off(nseq);
            HIP_TRY(hipMemcpy(h_off.data(), d_word_off, nseq * 8, hipMemcpyDeviceToHost));
            std::vector<uint64_t> p_off;
            std::vector<uint32_t> p_len;
            p_off.reserve(nseq + nseq / 8); p_len.reserve(nseq + nseq / 8);
            for (size_t s2 = 0; s2 < nseq; s2++) {
                const uint32_t L = h_len[s2];
                if (L <= 2u * PIECE) { p_off.push_back(h_off[s2]); p_len.push_back(L); continue; }
                for (uint64_t at = 0; at < L; at += PIECE) {
                    const uint64_t rest = L - at;
                    if (rest < (uint64_t)k && at) break; /* (its k-mers are the piece before's: that one ran to the sequence's end) */
                    p_off.push_back(h_off[s2] + at / 32);
                    p_len.push_back((uint32_t)std::min<uint64_t>(rest, (uint64_t)PIECE + (uint64_t)k - 1));
                }
            }
            HIP_TRY(d_split_off.alloc(p_off.size() * 8));
            HIP_TRY(d_split_len.alloc(p_len.size() * 4));
            HIP_TRY(hipMemcpy(d_split_off.p, p_off.data(), p_off.size() * 8, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(d_split_len.p, p_len.data(), p_len.size() * 4, hipMemcpyHostToDevice));
            ins_word_off = d_split_off.as<uint64_t>();
            ins_len = d_split_len.as<uint32_t>();
            ins_nseq = p_off.size();
        }
    }
    Table jt{};
    double load = 1.0;
    int rc = MTG_OK;
    const uint64_t n_junctions_ub = total_kmers_ub + ins_nseq + 1024; /* a sequence of L >= k nucleotides has L - k + 2 junction positions */
    /* large tables are built partition by partition (jt_insert_partitioned); BUILD_PARTITIONED = 0 / 1 of the tuning table forces either way */
    bool partitioned_ok = true;
    for (int attempt = 0; attempt < 6; attempt++) {
        const uint32_t kb = 2 * (uint32_t)(k - 1);
        BinShape BS{};
        uint64_t nb_part = 0;
        /* (a little emptier than the scattered insertion's table: the segments are filled in whatever order the lanes come, and at 0.7 a cluster of sixty full
         * buckets -- one in a billion buckets has one -- sent a key past the 63 buckets a look-up follows in half the runs; the buffer is sized for the larger
         * sparse ADJ table anyway) */
        const uint64_t nb0 = std::max<uint64_t>(buckets_for(n_junctions_ub, std::min(jt_load(), 0.62) * load, kb, MTG_ABND_SLOTS), jt_min_buckets(kb));
        const bool want_part = partitioned_ok && ins_nseq != 0 && (tune::is_set(tune::T_BUILD_PARTITIONED) ? tune::on(tune::T_BUILD_PARTITIONED) : n_junctions_ub >= (1ull << 27)) && bin_shape(nb0, kb, BS, nb_part);
        if (want_part) {
            if (int rc2 = alloc_slot_table(jt, jt_buf, n_junctions_ub, jt_load() * load, kb, prof, "clear_jt", adj_bytes_estimate(total_kmers_ub, k), nb_part, false)) return rc2;
            HIP_TRY(hipMemset(d_cnt.p, 0, 32));
            const size_t tb = jt.nbuckets * 8 * MTG_ABND_SLOTS;
            unsigned long long cnt0 = 0;
            const int prc = jt_insert_partitioned(jt, BS, k, d_words, ins_word_off, ins_len, ins_nseq, n_junctions_ub, jt_buf.as<uint8_t>() + tb, jt_buf.cap > tb ? jt_buf.cap - tb : 0, d_cnt.as<unsigned long long>(), cnt0, prof);
            if (prc == MTG_ERR_OVERFLOW) { partitioned_ok = false; attempt--; continue; } /* very uneven sequences or hashes: the ordinary insertion */
            if (prc) return prc;
            if (!cnt0) { rc = MTG_OK; break; }
            load *= 0.7;
            rc = MTG_ERR_OVERFLOW;
            set_error("index bucket displacement overflow");
            continue;
        }
        if (int rc2 = alloc_slot_table(jt, jt_buf, n_junctions_ub, jt_load() * load, 2 * (k - 1), prof, "clear_jt", adj_bytes_estimate(total_kmers_ub, k), jt_min_buckets(2 * (k - 1)))) return rc2;
        HIP_TRY(hipMemset(d_cnt.p, 0, 32));
        if (ins_nseq == 0) break; /* an empty graph: nothing to launch */
        prof.begin();
        hipLaunchKernelGGL(k_jt_insert_packed, dim3((unsigned)std::min<size_t>(ins_nseq, 256 * 32)), dim3(256), 0, 0, jt, k, d_words, ins_word_off, ins_len, ins_nseq, d_cnt.as<unsigned long long>());
        HIP_TRY(prof.end("jt_insert_packed", n_junctions_ub * (64 + 1), n_junctions_ub)); /* a junction position: its bucket read and written (2 x 32), its nucleotides */
        unsigned long long cnt[4];
        HIP_TRY(hipMemcpy(cnt, d_cnt.p, 32, hipMemcpyDeviceToHost));
        if (!cnt[0]) { rc = MTG_OK; break; }
        load *= 0.7;
        rc = MTG_ERR_OVERFLOW;
        set_error("index bucket displacement overflow");
    }
    if (rc) return rc;
    AbSynth src;
    src.lo = abund_lo; src.span = abund_span;
    const PackedSeqs packed{d_words, d_word_off, d_len, nseq, longest};
    if (int rc2 = build_from_jt(idx, jt_buf, jt, src, [] {}, 0, prof, &packed)) return rc2;
    idx->info.k = k;
    idx->info.abundance_min = (int)abund_lo;
    idx->info.abundance_auto = -1;
    prof.store(idx);
    *out = g.release();
    return MTG_OK;
}

/* The index of a counted solid set that arrives in pieces (host arrays, or the records of a saved index read from its file: 36 GB at human
 * size, never whole in host memory): fetch(off, m, k, a) hands over k-mers [off, off + m) and their abundances; every attempt to build the
 * tables is one pass over the pieces, the lookaheads are then derived from the tables themselves. */
int index_from_kmer_pieces(size_t n, int k, const KmerFetch& fetch, mtg_index** out)
{
    if (int rc = ensure_device()) return rc;
    if (k < 11 || k > 31 || !out) { set_error("invalid argument (11 <= k <= 31)"); return MTG_ERR_ARG; }
    return index_from_kmer_pieces_lean(n, k, fetch, out);
}

int index_from_kmers(const uint64_t* canon_kmers, const uint32_t* abundance, size_t n, int k, mtg_index** out)
{
    if (n && (!canon_kmers || !abundance)) { set_error("invalid argument (null arrays)"); return MTG_ERR_ARG; }
    return index_from_kmer_pieces(n, k, [&](size_t off, size_t, const uint64_t*& hk, const uint32_t*& ha) { hk = canon_kmers + off; ha = abundance + off; return true; }, out);
}

int index_from_packed_device(const uint64_t* d_words, const uint64_t* d_word_off, const uint32_t* d_len, size_t nseq, uint64_t total_kmers_ub, int k,
                             uint32_t abund_lo, uint32_t abund_span, mtg_index** out)
{
    if (int rc = ensure_device()) return rc;
    if (k < 11 || k > 31 || !out || (nseq && (!d_words || !d_word_off || !d_len))) { set_error("invalid argument (11 <= k <= 31)"); return MTG_ERR_ARG; }
    return index_from_packed_device_lean(d_words, d_word_off, d_len, nseq, total_kmers_ub, k, abund_lo, abund_span, out);
}

/* the whole index on another device: same shapes, tables and unitig store copied device to device */
int index_replicate(const mtg_index* src, int device, mtg_index** out)
{
    if (!src || !out) { set_error("null argument"); return MTG_ERR_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { set_error("no such device: %d", device); return MTG_ERR_ARG; }
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{prev};
    HIP_TRY(hipSetDevice(device));
    IndexGuard g(new mtg_index());
    mtg_index* idx = g.idx;
    idx->dev = src->dev;
    idx->info = src->info;
    idx->device = device;
    idx->dev.adj.slots = idx->dev.abnd.slots = nullptr; /* free_tables must only see what this copy owns */
    idx->dev.bloom.bits = nullptr;
    idx->dev.us.words = nullptr;
    idx->dev.us.ab = nullptr;
    /* peer access both ways (a copy between two devices then goes over their xGMI link without a bounce; "already enabled" is fine), every
     * buffer allocated first, then the five copies in flight together on one stream of the destination */
    if (device != src->device) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, device, src->device) == hipSuccess && can) {
            if (hipDeviceEnablePeerAccess(src->device, 0) != hipSuccess) (void)hipGetLastError();
            (void)hipSetDevice(src->device);
            if (hipDeviceEnablePeerAccess(device, 0) != hipSuccess) (void)hipGetLastError();
            HIP_TRY(hipSetDevice(device));
        }
    }
    hipStream_t cs = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } sg{cs};
    struct Part { void** dst; const void* from; size_t bytes; };
    const size_t ba = src->dev.adj.nbuckets * 16 * MTG_ADJ_SLOTS, bb = src->dev.abnd.nbuckets * 8 * MTG_ABND_SLOTS, bc = src->dev.bloom.nblocks * 64;
    const size_t nw = src->dev.us.nwords ? src->dev.us.nwords + 8 : 0;
    const Part parts[5] = {{(void**)&idx->dev.adj.slots, src->dev.adj.slots, ba}, {(void**)&idx->dev.abnd.slots, src->dev.abnd.slots, bb}, {(void**)&idx->dev.bloom.bits, src->dev.bloom.bits, bc},
                           {(void**)&idx->dev.us.words, src->dev.us.words, nw * 8}, {(void**)&idx->dev.us.ab, src->dev.us.ab, nw * 32}};
    for (const Part& p : parts) if (p.from && p.bytes) HIP_TRY(hipMalloc(p.dst, p.bytes)); /* a failure leaves what was allocated to free_tables (IndexGuard) */
    for (const Part& p : parts) if (p.from && p.bytes) HIP_TRY(hipMemcpyPeerAsync(*p.dst, device, p.from, src->device, p.bytes, cs));
    HIP_TRY(hipStreamSynchronize(cs));
    if (src->dev.adj.sp_words) idx->dev.adj.sp_words = idx->dev.us.words; /* the sparse form reads this copy's store */
    *out = g.release();
    return MTG_OK;
}

/* the solid k-mers of an index and their abundances, read back from its tables (for the index writer), in pieces of at most `piece`
 * k-mers handed to sink(kmers, abundances, count) */
int index_export(const mtg_index* idx, const std::function<bool(const uint64_t*, const uint32_t*, size_t)>& sink)
{
    if (int rc = use_device_of(idx)) return rc;
    const uint64_t n = idx->info.nb_solid_kmers;
    DevBuf d_k, d_a, d_cur;
    HIP_TRY(d_k.alloc((n + 1) * 8));
    HIP_TRY(d_a.alloc((n + 1) * 4));
    HIP_TRY(d_cur.alloc(8));
    HIP_TRY(hipMemset(d_cur.p, 0, 8));
    const uint64_t nslots = idx->dev.abnd.nbuckets * MTG_ABND_SLOTS;
    hipLaunchKernelGGL(k_abnd_export, dim3((unsigned)std::min<uint64_t>((nslots + 255) / 256, 256 * 32)), dim3(256), 0, 0, idx->dev, d_k.as<uint64_t>(), d_a.as<uint32_t>(),
                       d_cur.as<unsigned long long>(), (unsigned long long)n);
    HIP_TRY(hipGetLastError());
    unsigned long long got = 0;
    HIP_TRY(hipMemcpy(&got, d_cur.p, 8, hipMemcpyDeviceToHost));
    if (got != n) { set_error("index export: %llu k-mers in the table, %llu expected", got, (unsigned long long)n); return MTG_ERR_FORMAT; }
    const size_t piece = (size_t)1 << 24;
    std::vector<uint64_t> hk(std::min<uint64_t>(n, piece));
    std::vector<uint32_t> ha(hk.size());
    for (uint64_t off = 0; off < n; off += piece) {
        const size_t m = (size_t)std::min<uint64_t>(piece, n - off);
        HIP_TRY(hipMemcpy(hk.data(), d_k.as<uint64_t>() + off, m * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(ha.data(), d_a.as<uint32_t>() + off, m * 4, hipMemcpyDeviceToHost));
        if (!sink(hk.data(), ha.data(), m)) { set_error("index export: the writer failed"); return MTG_ERR_IO; }
    }
    return MTG_OK;
}

int index_dump(const mtg_index* idx, IndexDump& d)
{
    if (int rc = use_device_of(idx)) return rc;
    d.k = idx->dev.k; d.abundance_min = idx->info.abundance_min; d.abundance_auto = idx->info.abundance_auto;
    d.nb_solid = idx->info.nb_solid_kmers; d.nb_branching = idx->info.nb_branching; d.nb_saturated = idx->info.nb_saturated;
    d.n_words = idx->dev.us.nwords; d.n_unitigs = idx->dev.us.nunitigs;
    d.words.clear(); d.ab.clear();
    if (d.n_words) {
        const uint64_t nw = d.n_words + 8;
        d.words.resize(nw);
        d.ab.resize(nw * 32);
        HIP_TRY(hipMemcpy(d.words.data(), idx->dev.us.words, nw * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(d.ab.data(), idx->dev.us.ab, nw * 32, hipMemcpyDeviceToHost));
    }
    DevBuf d_cnt, d_k, d_a;
    HIP_TRY(d_cnt.alloc(8));
    HIP_TRY(hipMemset(d_cnt.p, 0, 8));
    const uint64_t nslots = idx->dev.abnd.nbuckets * MTG_ABND_SLOTS;
    const unsigned blocks = (unsigned)std::min<uint64_t>((nslots + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(k_leftovers, dim3(blocks), dim3(256), 0, 0, idx->dev, (uint64_t*)nullptr, (uint32_t*)nullptr, d_cnt.as<unsigned long long>(), 0ull);
    unsigned long long n_left = 0;
    HIP_TRY(hipMemcpy(&n_left, d_cnt.p, 8, hipMemcpyDeviceToHost));
    d.left_k.resize(n_left);
    d.left_a.resize(n_left);
    if (n_left) {
        HIP_TRY(d_k.alloc(n_left * 8));
        HIP_TRY(d_a.alloc(n_left * 4));
        HIP_TRY(hipMemset(d_cnt.p, 0, 8));
        hipLaunchKernelGGL(k_leftovers, dim3(blocks), dim3(256), 0, 0, idx->dev, d_k.as<uint64_t>(), d_a.as<uint32_t>(), d_cnt.as<unsigned long long>(), n_left);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(d.left_k.data(), d_k.p, n_left * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(d.left_a.data(), d_a.p, n_left * 4, hipMemcpyDeviceToHost));
    }
    return MTG_OK;
}

namespace {
struct AdjPrealloc {
    std::thread th;
    DevBuf buf;
    hipError_t err = hipSuccess;
};
} // namespace
void* adj_prealloc_begin(uint64_t nb_solid, int k)
{
    if (ensure_device() || nb_solid == 0 || k < 11 || k > 31) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    AdjPrealloc* a = new AdjPrealloc();
    const size_t bytes = adj_bytes_estimate(nb_solid, k);
    a->th = std::thread([a, dev, bytes] {
        a->err = hipSetDevice(dev);
        if (a->err == hipSuccess) a->err = a->buf.alloc(bytes);
    });
    return a;
}
void adj_prealloc_drop(void* handle)
{
    AdjPrealloc* a = (AdjPrealloc*)handle;
    if (!a) return;
    if (a->th.joinable()) a->th.join();
    delete a;
}

/* an index out of its container: the store goes up as it is, the tables are derived from it (sparsify); nothing is counted or walked */
int index_from_dump(const IndexDump& d, mtg_index** out)
{
    struct PreGuard { void* h; ~PreGuard() { adj_prealloc_drop(h); } } pre_guard{d.prealloc};
    if (int rc = ensure_device()) return rc;
    if (d.k < 11 || d.k > 31 || !out) { set_error("invalid argument (11 <= k <= 31)"); return MTG_ERR_ARG; }
    if (d.n_words == 0) {
        /* an index without stored unitigs: through the list of its k-mers */
        std::vector<uint64_t> km(d.left_k);
        std::vector<uint32_t> ab(d.left_a);
        const uint64_t mk = kmask(d.k);
        std::vector<uint8_t> ab_all; /* a container that is being read: its abundance bytes, whole */
        if (d.ab_read && d.n_words) {
            ab_all.resize((d.n_words + 8) * 32);
            if (!d.ab_read(0, ab_all.size(), ab_all.data())) { set_error("index container: short read"); return MTG_ERR_FORMAT; }
        }
        const uint8_t* dab = d.ab_read ? ab_all.data() : d.ab.data();
        for (uint64_t h = 0; h < d.n_words;) {
            const uint64_t len = d.words[h];
            for (uint64_t i = 0; i + d.k <= len; i++) {
                const uint64_t p = (h + 1) * 32 + i, lo = d.words[p >> 5] >> (2 * (p & 31)), hi = (p & 31) ? d.words[(p >> 5) + 1] << (64 - 2 * (p & 31)) : 0;
                const uint64_t r = ((lo | hi) & mk) ^ (0xAAAAAAAAAAAAAAAAULL & mk), f = revcomp(r, d.k);
                km.push_back(f < r ? f : r);
                ab.push_back(dab[p]);
            }
            h += 1 + (len + 31) / 32;
        }
        int rc = index_from_kmers(km.data(), ab.data(), km.size(), d.k, out);
        if (!rc) { (*out)->info.abundance_min = d.abundance_min; (*out)->info.abundance_auto = d.abundance_auto; }
        return rc;
    }
    BuildProf prof;
    IndexGuard g(new mtg_index());
    mtg_index* idx = g.idx;
    idx->dev.k = d.k;
    HIP_TRY(hipGetDevice(&idx->device));
    const uint64_t nw = d.n_words + 8;
    const bool streamed = (bool)d.ab_read;
    if (d.words.size() < nw || (!streamed && d.ab.size() < nw * 32) || d.left_k.size() != d.left_a.size()) { set_error("index container: inconsistent sizes"); return MTG_ERR_FORMAT; }
    HIP_TRY(timed_malloc((void**)&idx->dev.us.words, nw * 8));
    HIP_TRY(timed_malloc((void**)&idx->dev.us.ab, nw * 32));
    const auto t_up0 = std::chrono::steady_clock::now();
    HIP_TRY(hipMemcpy(idx->dev.us.words, d.words.data(), nw * 8, hipMemcpyHostToDevice));
    /* the abundance bytes: either in host memory already, or (a container that is being read) straight from the file in page-locked pieces,
     * by a few threads with a stream each -- and while the tables are derived below: those kernels read the store's words only */
    std::vector<std::thread> uploaders;
    std::atomic<uint64_t> next_piece{0};
    std::atomic<int> up_err{0};
    struct JoinAll { std::vector<std::thread>& ts; ~JoinAll() { for (auto& t : ts) if (t.joinable()) t.join(); } } join_all{uploaders};
    if (!streamed) HIP_TRY(hipMemcpy(idx->dev.us.ab, d.ab.data(), nw * 32, hipMemcpyHostToDevice));
    else {
        const uint64_t total = nw * 32, piece = (uint64_t)32 << 20, npieces = (total + piece - 1) / piece;
        const int env_threads = (int)tune::i(tune::T_LOAD_THREADS, 0);
        const int nthreads = (int)std::min<uint64_t>(npieces, (uint64_t)(env_threads > 0 ? std::min(env_threads, 32) : std::min(8, std::max(2, Pool::cpu_budget() / 2))));
        uint8_t* dst = idx->dev.us.ab;
        const int device = idx->device;
        for (int t = 0; t < nthreads; t++)
            uploaders.emplace_back([&, dst, device, total, piece, npieces] {
                void* pin = nullptr;
                hipStream_t st = nullptr;
                if (hipSetDevice(device) != hipSuccess || hipHostMalloc(&pin, piece, hipHostMallocDefault) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) up_err.store(1);
                else
                    for (;;) {
                        const uint64_t c = next_piece.fetch_add(1);
                        if (c >= npieces || up_err.load()) break;
                        const uint64_t off = c * piece, n = std::min(piece, total - off);
                        if (!d.ab_read(off, (size_t)n, pin)) { up_err.store(2); break; }
                        if (hipMemcpyAsync(dst + off, pin, n, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { up_err.store(1); break; }
                    }
                if (st) (void)hipStreamDestroy(st);
                if (pin) (void)hipHostFree(pin);
            });
    }
    idx->dev.us.nwords = d.n_words;
    /* the header words of the unitigs, one after the other */
    std::vector<uint64_t> hdr;
    hdr.reserve(d.n_unitigs);
    for (uint64_t h = 0; h < d.n_words;) {
        const uint64_t len = d.words[h];
        if (len < (uint64_t)d.k + 1 || len > MTG_US_MAX_LEN) { set_error("index container: damaged unitig store"); return MTG_ERR_FORMAT; }
        hdr.push_back(h);
        h += 1 + (len + 31) / 32;
    }
    idx->dev.us.nunitigs = hdr.size();
    DevBuf d_hdr, d_rec, d_k, d_a;
    const auto upload = [](DevBuf& b, const void* src, size_t bytes) -> hipError_t {
        hipError_t e = b.alloc(bytes);
        if (e != hipSuccess || !bytes) return e;
        return hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice);
    };
    HIP_TRY(upload(d_hdr, hdr.data(), hdr.size() * 8));
    HIP_TRY(d_rec.alloc((hdr.size() + 1) * sizeof(UsRec)));
    hipLaunchKernelGGL(k_recs_from_store, dim3((unsigned)((hdr.size() + 255) / 256)), dim3(256), 0, 0, idx->dev.us, d_hdr.as<uint64_t>(), (unsigned long long)hdr.size(), d.k, d_rec.as<UsRec>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(upload(d_k, d.left_k.data(), d.left_k.size() * 8));
    HIP_TRY(upload(d_a, d.left_a.data(), d.left_a.size() * 4));
    prof.host_phase("store_words_up+headers", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_up0).count(), nw * 8, hdr.size());
    const double bpk = tune::f(tune::T_BLOOM_BITS, 12.0);
    if (bpk > 0) {
        bloom_shape(idx->dev.bloom, d.nb_solid, bpk, d.k);
        HIP_TRY(hipMalloc((void**)&idx->dev.bloom.bits, idx->dev.bloom.nblocks * 64));
        HIP_TRY(hipMemset(idx->dev.bloom.bits, 0, idx->dev.bloom.nblocks * 64));
    }
    idx->info.k = d.k;
    idx->info.abundance_min = d.abundance_min; idx->info.abundance_auto = d.abundance_auto;
    idx->info.nb_solid_kmers = d.nb_solid; idx->info.nb_branching = d.nb_branching; idx->info.nb_saturated = d.nb_saturated;
    idx->info.nb_unitigs = hdr.size();
    idx->info.unitig_bytes = nw * 40;
    idx->info.bloom_blocks = idx->dev.bloom.nblocks;
    idx->info.bloom_minimizer = (uint32_t)idx->dev.bloom.mm;
    idx->info.adj_bucket_bytes = 16 * MTG_ADJ_SLOTS;
    idx->info.abnd_bucket_bytes = 8 * MTG_ABND_SLOTS;
    DevBuf* adj_pre = nullptr;
    if (AdjPrealloc* a = (AdjPrealloc*)d.prealloc) {
        if (a->th.joinable()) a->th.join();
        if (a->err == hipSuccess && a->buf.p) adj_pre = &a->buf; /* too small after all, or failed: sparsify allocates its own */
        else (void)hipGetLastError();
    }
    if (int rc = sparsify(idx, d_rec.as<UsRec>(), hdr.size(), true, d_k.as<uint64_t>(), d_a.as<uint32_t>(), d.left_k.size(), &prof, nullptr, adj_pre)) return rc;
    const auto t_j0 = std::chrono::steady_clock::now();
    for (auto& t : uploaders) t.join();
    if (up_err.load()) { set_error(up_err.load() == 2 ? "index container: short read" : "index container: the upload of the abundance bytes failed"); return up_err.load() == 2 ? MTG_ERR_FORMAT : MTG_ERR_NO_DEVICE; }
    if (streamed) prof.host_phase("abundances_file_to_hbm", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_up0).count(), nw * 32, nw * 32);
    (void)t_j0;
    HIP_TRY(hipDeviceSynchronize());
    prof.store(idx);
    *out = g.release();
    return MTG_OK;
}

void index_release(mtg_index* idx)
{
    if (!idx) return;
    for (Workspace& w : idx->ws) {
        for (int i = 0; i < Workspace::NSLOTS; i++) if (w.ptr[i]) (void)hipFree(w.ptr[i]);
        for (int i = 0; i < Workspace::NHOST; i++) if (w.hptr[i]) (void)hipHostFree(w.hptr[i]);
        for (int i = 0; i < Workspace::NEVENTS; i++) if (w.events[i]) (void)hipEventDestroy((hipEvent_t)w.events[i]);
        if (w.stream) (void)hipStreamDestroy((hipStream_t)w.stream);
        if (w.copy_stream) (void)hipStreamDestroy((hipStream_t)w.copy_stream);
    }
    free_tables(idx);
    delete idx;
}


/* Graph::create on the device.  The reads arrive as text blocks (whole reads separated by '\n'); k-mers are counted in an exact
 * open-addressing table, in P passes over the reads when one table for all k-mers would not fit (a k-mer belongs to the pass its hash
 * selects).  Round 1: every pass feeds the abundance histogram, from which the cut-off comes (automatic: gatb's Histogram heuristic).
 * Round 2: the solid k-mers go from the count table straight into the index tables (one pass: the table of round 1 is still there;
 * several: the passes are counted again).  Then lookaheads and the unitig store, both from the index's own tables. */
int index_from_stream(ReadStream& rs, int k, int abundance_min, int abundance_max, mtg_index** out)
{
    if (int rc = ensure_device()) return rc;
    if (k < 11 || k > 31 || !out) { set_error("invalid argument (11 <= k <= 31)"); return MTG_ERR_ARG; }
    const uint32_t nbins = 10003; /* STR_HISTOGRAM_MAX 10000, src/Filler.cpp:200 */
    BuildProf prof;
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    const size_t budget = (size_t)((double)free_b * 0.45); /* the index tables have to fit next to the count table */
    const size_t n_hint = std::max<size_t>(rs.size_hint(), 1 << 16);
    /* distinct k-mers <= instances; start from instances / 4 slots in total (30x data has ~25 instances per distinct k-mer) and grow on
     * overflow */
    uint64_t total_slots = 1ull << 16;
    while (total_slots < n_hint / 4) total_slots <<= 1;
    const uint32_t forced = (uint32_t)tune::i(tune::T_COUNT_PASSES, 0);
    DevBuf d_text, d_flags, d_histo;
    const size_t text_cap = (size_t)80 << 20;
    HIP_TRY(d_text.alloc(text_cap));
    HIP_TRY(d_flags.alloc(64));
    HIP_TRY(d_histo.alloc((size_t)nbins * 8));
    std::vector<uint64_t> histo(nbins, 0);
    for (int attempt = 0; attempt < 10; attempt++, total_slots <<= 1) {
        uint32_t npass = 1;
        while (total_slots / npass * 12 + (64u << 20) > budget && npass < 1024) npass <<= 1;
        if (forced > npass) npass = forced;
        uint64_t cap = 1ull << 10;
        while (cap < total_slots / npass) cap <<= 1;
        if (cap * 12 + (64u << 20) > free_b) { set_error("not enough device memory to count the k-mers of the reads"); return MTG_ERR_NOMEM; }
        DevBuf d_keys, d_cnts;
        HIP_TRY(d_keys.alloc(cap * 8));
        HIP_TRY(d_cnts.alloc(cap * 4));
        CountTable t;
        t.keys = d_keys.as<uint64_t>();
        t.counts = d_cnts.as<uint32_t>();
        t.mask = cap - 1;
        /* one pass over the reads into the (cleared) table; returns 1 when the table overflowed */
        auto count_pass = [&](uint32_t pass, bool& overflow) -> int {
            HIP_TRY(hipMemset(d_keys.p, 0xFF, cap * 8));
            HIP_TRY(hipMemset(d_cnts.p, 0, cap * 4));
            HIP_TRY(hipMemset(d_flags.p, 0, 64));
            if (!rs.rewind()) { set_error("cannot read the input again"); return MTG_ERR_IO; }
            const char* p = nullptr;
            size_t n = 0;
            while (rs.next_block(p, n)) {
                for (size_t off = 0; off < n;) { /* a block larger than the device buffer goes in pieces that overlap by k-1 characters */
                    const size_t len = std::min(text_cap - 64, n - off);
                    HIP_TRY(hipMemcpy(d_text.p, p + off, len, hipMemcpyHostToDevice));
                    hipLaunchKernelGGL(k_count, dim3(256 * 32), dim3(256), 0, 0, t, d_text.as<char>(), (uint64_t)len, k, npass, pass, d_flags.as<unsigned long long>());
                    HIP_TRY(hipGetLastError());
                    HIP_TRY(hipDeviceSynchronize());
                    if (off + len >= n) break;
                    off += len - (size_t)(k - 1);
                }
            }
            if (rs.failed()) return MTG_ERR_IO;
            unsigned long long flags[8];
            HIP_TRY(hipMemcpy(flags, d_flags.p, 64, hipMemcpyDeviceToHost));
            overflow = flags[0] != 0;
            return MTG_OK;
        };
        /* round 1: the histogram */
        HIP_TRY(hipMemset(d_histo.p, 0, (size_t)nbins * 8));
        bool overflow = false;
        for (uint32_t pass = 0; pass < npass && !overflow; pass++) {
            if (int rc = count_pass(pass, overflow)) return rc;
            if (overflow) break; /* table too full: double the slots and start over */
            hipLaunchKernelGGL(k_count_stats, dim3(256 * 16), dim3(256), 0, 0, t, 0u, d_histo.as<unsigned long long>(), nbins, d_flags.as<unsigned long long>() + 1);
            HIP_TRY(hipGetLastError());
        }
        if (overflow) continue;
        HIP_TRY(hipMemcpy(histo.data(), d_histo.p, (size_t)nbins * 8, hipMemcpyDeviceToHost));
        int autoc = -1;
        if (abundance_min < 0) { autoc = auto_cutoff(histo, 3); abundance_min = autoc; } /* auto never goes below 3 (src/Filler.cpp:201) */
        const uint32_t lo = (uint32_t)std::max(abundance_min, 1), hi = abundance_max > 0 ? (uint32_t)abundance_max : 0xFFFFFFFFu;
        uint64_t n_solid = 0;
        for (uint32_t c = lo; c < nbins; c++) if (c <= hi || c == nbins - 1) n_solid += histo[c]; /* the last bin holds every larger count */
        /* round 2: the index */
        IndexGuard g(new mtg_index());
        mtg_index* idx = g.idx;
        idx->dev.k = k;
        HIP_TRY(hipGetDevice(&idx->device));
        DevBuf d_cnt;
        HIP_TRY(d_cnt.alloc(32));
        double load = 1.0;
        int rc = MTG_OK;
        {
            /* the lean build: the solid k-mers' junctions into the junction table; their abundances stay in the count table (one counting
             * pass) or go into an ABND table of their own (several: the count table of a pass does not outlive it) */
            prof.host_phase("count_reads", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - prof.t0).count(), 0, n_solid);
            DevBuf jt_buf, abnd_buf;
            Table jt{}, abnd{};
            unsigned long long cnt[4] = {0, 0, 0, 0};
            for (int ia = 0; ia < 6; ia++) {
                if (int rc2 = alloc_slot_table(jt, jt_buf, n_solid + n_solid / 8 + 1024, jt_load() * load, 2 * (k - 1), prof, "clear_jt", adj_bytes_estimate(n_solid, k), jt_min_buckets(2 * (k - 1)))) return rc2;
                if (npass > 1) if (int rc2 = alloc_slot_table(abnd, abnd_buf, n_solid, 0.6 * load, 2 * k, prof, "clear_abnd_source")) return rc2;
                HIP_TRY(hipMemset(d_cnt.p, 0, 32));
                for (uint32_t pass = 0; pass < npass; pass++) {
                    if (npass > 1) {
                        bool ovf2 = false;
                        if (int rc2 = count_pass(pass, ovf2)) return rc2;
                        if (ovf2) { set_error("k-mer count table overflowed on a repeated pass"); return MTG_ERR_OVERFLOW; }
                    }
                    prof.begin();
                    hipLaunchKernelGGL(k_jt_insert_from_counts, dim3(256 * 16), dim3(256), 0, 0, jt, abnd, npass > 1 ? 1 : 0, k, t, lo, hi, d_cnt.as<unsigned long long>());
                    HIP_TRY(prof.end("jt_insert_from_counts", (t.mask + 1) * 12 + n_solid / npass * (2 * 64 + (npass > 1 ? 64 : 0)), n_solid / npass));
                }
                HIP_TRY(hipMemcpy(cnt, d_cnt.p, 32, hipMemcpyDeviceToHost));
                if (!cnt[0]) { rc = MTG_OK; break; }
                load *= 0.7;
                rc = MTG_ERR_OVERFLOW;
                set_error("index bucket displacement overflow");
            }
            if (rc) return rc;
            (void)d_text.alloc(0);
            if (npass == 1) {
                AbFromCounts src;
                src.t = t;
                rc = build_from_jt(idx, jt_buf, jt, src, [&] { (void)d_keys.alloc(0); (void)d_cnts.alloc(0); }, 0, prof);
            } else {
                (void)d_keys.alloc(0);
                (void)d_cnts.alloc(0);
                AbFromTable src;
                src.abnd = abnd;
                rc = build_from_jt(idx, jt_buf, jt, src, [&] { (void)abnd_buf.alloc(0); }, cnt[3], prof);
            }
            if (rc) return rc;
            idx->info.k = k;
            idx->info.abundance_min = abundance_min;
            idx->info.abundance_auto = autoc;
            prof.store(idx);
            *out = g.release();
            return MTG_OK;
        }
    }
    set_error("k-mer count table kept overflowing");
    return MTG_ERR_OVERFLOW;
}


} // namespace mtgi
