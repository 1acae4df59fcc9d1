/*
 * mtg_gpu.hip -- HIP kernels and device management of libmtgfill.so (gfx950 only).
 *
 *   k_insert_kmers / k_insert_packed : index construction (Graph::create, /root/reference/src/Filler.cpp:210)
 *   k_query                          : batched contains / queryAbundance / successors / predecessors
 *   k_stage_a                        : breadth-first contig construction of one gap per lane
 *                                      (IterativeExtensions::construct_linear_seqs, src/Filler.cpp:884)
 *   k_post                           : terminal-node search per contig + coverage of the single-contig solution, one wave per gap;
 *                                      then the dense copy of what the host needs of the gap
 *                                      (find_nodes_containing_multiple_R, src/Filler.cpp:1294-1378; coverage :959-988)
 *   k_paths                          : contig graph + reverse path enumeration of multi-contig gaps, one wave per gap
 *                                      (find_all_paths_rev, src/GraphAnalysis.cpp:205-326)
 *   k_nw                             : Needleman-Wunsch match counts for the de-duplication of multi-path solutions, one wave per pair
 *                                      (remove_almost_identical_solutions, src/Utils.cpp:87-189,208-238)
 *   k_chase                          : dependent random 64-byte reads (measured roofline ceiling)
 */
#include "mtg_internal.h"
#include "mtg_marshal.h"
#include <hip/hip_runtime.h>
#include <chrono>
#include <thread>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace mtg;

namespace mtgi {

static thread_local char g_err[512] = "";
static thread_local mtg_batch_stats g_stats{};
void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
void stats_store(const mtg_batch_stats& s) { g_stats = s; }

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);       \
            return (e_ == hipErrorOutOfMemory) ? MTG_ERR_NOMEM : MTG_ERR_NO_DEVICE;                     \
        }                                                                                               \
    } while (0)

static int ensure_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (%s); libmtgfill has no CPU fallback", e == hipSuccess ? "0 devices" : hipGetErrorString(e));
        return MTG_ERR_NO_DEVICE;
    }
    return MTG_OK;
}

/* the HIP device is a per-thread setting: a call on an index runs on the index's device whatever thread makes it */
static int use_device_of(const mtg_index* idx)
{
    if (int rc = ensure_device()) return rc;
    if (idx) HIP_TRY(hipSetDevice(idx->device));
    return MTG_OK;
}

/* ------------------------------------------------------------------------------------------------ kernels */
__device__ __forceinline__ uint64_t d_splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
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

/* one workgroup per sequence; lanes stride over k-mer start positions */
__global__ void k_insert_packed(Index ix, const uint64_t* __restrict__ words, const uint64_t* __restrict__ word_off, const uint32_t* __restrict__ len,
                                size_t nseq, uint32_t abund_lo, uint32_t abund_span, unsigned long long* counters)
{
    const int k = ix.k;
    const uint64_t mk = kmask(k);
    unsigned long long created = 0;
    int fail = 0;
    for (size_t s = blockIdx.x; s < nseq; s += gridDim.x) {
        const uint64_t* w = words + word_off[s];
        const uint32_t L = len[s];
        if (L < (uint32_t)k) continue;
        for (uint32_t p = threadIdx.x; p + k <= L; p += blockDim.x) {
            /* nts p .. p+k-1, nt i at bits 2*(i%32) of word i/32 */
            uint64_t f = 0;
            for (int j = 0; j < k; j++) {
                const uint32_t i = p + j;
                f = (f << 2) | ((w[i >> 5] >> (2 * (i & 31))) & 3ull);
            }
            f &= mk;
            const uint64_t r = revcomp(f, k);
            const uint64_t c = f < r ? f : r;
            const uint32_t a = d_synth_abundance(c, abund_lo, abund_span);
            int rr = index_insert(ix, c, a);
            fail |= rr & 1;
            created += (rr >> 1) & 1;
        }
    }
    if (fail) atomicOr(&counters[0], 1ull);
    if (created) atomicAdd(&counters[1], created);
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
/* the solid k-mers of a count table (count in [lo, hi]) straight into the index tables; counters as k_insert_kmers */
__global__ void k_insert_from_counts(Index ix, CountTable t, uint32_t lo, uint32_t hi, unsigned long long* counters)
{
    unsigned long long created = 0, sat = 0;
    int fail = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= t.mask; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t key = t.keys[i];
        if (key == ~0ULL) continue;
        const uint32_t c = t.counts[i];
        if (c < lo || c > hi) continue;
        const int r = index_insert(ix, key, c);
        fail |= r & 1;
        created += (r >> 1) & 1;
        sat += c > 255u;
    }
    if (fail) atomicOr(&counters[0], 1ull);
    if (created) atomicAdd(&counters[1], created);
    if (sat) atomicAdd(&counters[3], sat);
}
/* lookaheads for every solid k-mer, read back from the ABND table (an index that was not built from a k-mer list) */
__global__ void k_lookahead_table(Index ix)
{
    const uint64_t nslots = ix.abnd.nbuckets * MTG_ABND_SLOTS;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nslots; s += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t c;
        if (!abnd_slot_kmer(ix.abnd, s, c)) continue;
        Kmer x = make_kmer(c, ix.k);
        build_lookahead(ix, x);
        Kmer y;
        y.f = x.r; y.r = x.f;
        build_lookahead(ix, y);
    }
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
__global__ void k_lookahead_packed(Index ix, const uint64_t* __restrict__ words, const uint64_t* __restrict__ word_off, const uint32_t* __restrict__ len, size_t nseq)
{
    const int k = ix.k;
    const uint64_t mk = kmask(k);
    for (size_t s = blockIdx.x; s < nseq; s += gridDim.x) {
        const uint64_t* w = words + word_off[s];
        const uint32_t L = len[s];
        if (L < (uint32_t)k) continue;
        for (uint32_t p = threadIdx.x; p + k <= L; p += blockDim.x) {
            Kmer x;
            x.r = le_kmer(w, p, mk) ^ (0xAAAAAAAAAAAAAAAAULL & mk);
            x.f = revcomp(x.r, k);
            build_lookahead(ix, x);
            Kmer y;
            y.f = x.r; y.r = x.f;
            build_lookahead(ix, y);
        }
    }
}

/* ---- unitig store construction (mtg_dev.h: us_*), over the solid k-mers read back from the ABND table ---- */
/* counters[0] += chain starts; counters[1] += branching nodes (in-degree != 1 or out-degree != 1); counters[2] += solid k-mers.
 * starts != nullptr: the oriented start k-mers are also collected there (counters[3] = cursor). */
__global__ void k_us_starts(Index ix, unsigned long long* counters, uint64_t* starts, unsigned long long cap)
{
    const uint64_t nslots = ix.abnd.nbuckets * MTG_ABND_SLOTS;
    const uint64_t mk1 = kmask(ix.k - 1);
    uint32_t lines = 0;
    unsigned long long ns = 0, nbr = 0, nk = 0;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nslots; s += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t c;
        if (!abnd_slot_kmer(ix.abnd, s, c)) continue;
        nk++;
        Kmer o[2];
        o[0] = make_kmer(c, ix.k);
        o[1].f = o[0].r; o[1].r = o[0].f;
        if (!starts) nbr += !(popc4(adj_right_t(ix.adj, o[0], mk1, lines).out) == 1 && popc4(adj_left(ix, o[0], mk1, lines).in) == 1);
        for (int u = 0; u < (o[0].f == o[0].r ? 1 : 2); u++) {
            if (!us_is_start(ix, o[u], lines)) continue;
            ns++;
            if (starts) { const unsigned long long at = atomicAdd(&counters[3], 1ull); if (at < cap) starts[at] = o[u].f; }
        }
    }
    if (!starts) { if (ns) atomicAdd(&counters[0], ns); if (nbr) atomicAdd(&counters[1], nbr); if (nk) atomicAdd(&counters[2], nk); }
}
/* one chain start per lane: walks to the other end; the end with the smaller canonical k-mer reserves the unitig's words and record.
 * cursors[0] = words, cursors[1] = records */
__global__ void __launch_bounds__(64) k_us_plan(Index ix, const uint64_t* __restrict__ starts, unsigned long long n, unsigned long long* cursors, UsRec* rec, unsigned long long rec_cap)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t lines = 0;
    us_plan_start(ix, make_kmer(starts[i], ix.k), &cursors[0], &cursors[1], rec, rec_cap, lines);
}
/* one stored unitig per lane: its sequence into the store */
__global__ void __launch_bounds__(64) k_us_emit(Index ix, const UsRec* __restrict__ rec, unsigned long long n)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t lines = 0;
    us_emit(ix, rec[i], lines);
}
/* one stored unitig per wave, its k-mers dealt to the lanes: abundances into the store, pointers into the ADJ entries of its junctions */
__global__ void __launch_bounds__(256) k_us_link(Index ix, const UsRec* __restrict__ rec, unsigned long long n)
{
    const unsigned long long wave = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((unsigned long long)gridDim.x * blockDim.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t lines = 0;
    for (unsigned long long u = wave; u < n; u += nwaves) {
        const UsRec r = rec[u];
        for (uint32_t i = lane; i < r.len_k; i += 64) us_link(ix, r, i, lines);
    }
}
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
/* one stored unitig per wave, its k-mers dealt to the lanes: the entries of its junctions in the new tables; counters[0] = overflow flag */
__global__ void __launch_bounds__(256) k_sparse_link(Index ix, const UsRec* __restrict__ rec, unsigned long long n, int with_bloom, unsigned long long* counters)
{
    const unsigned long long wave = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((unsigned long long)gridDim.x * blockDim.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    int fail = 0;
    for (unsigned long long u = wave; u < n; u += nwaves) {
        const UsRec r = rec[u];
        for (uint32_t i = lane; i < r.len_k; i += 64) fail |= sparse_link(ix, r, i, with_bloom != 0);
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
/* one streaming pass over the junction table (jt_scan_entry); collect = 0: counts starts / k-mers of no chain and the statistics,
 * collect = 1: fills the lists (the statistics are left alone) */
template <typename Src>
__global__ void __launch_bounds__(256) k_jt_scan(Table jt, int k, Src src, unsigned long long* counters, int collect, uint64_t* starts, unsigned long long cap_starts,
                                                 uint64_t* left_k, uint32_t* left_a, unsigned long long cap_left)
{
    const uint64_t nslots = jt.nbuckets * MTG_ABND_SLOTS;
    JtAcc acc{};
    uint32_t lines = 0;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nslots; s += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t J;
        const uint32_t m = jt_slot_key(jt, s, J);
        if (!m) continue;
        jt_scan_entry(jt, k, J, m, src, acc, counters, starts, cap_starts, left_k, left_a, cap_left, lines);
    }
    if (collect) return;
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
/* one chain start per lane: the walk to the other end; the end the chain is stored from reserves words and record */
__global__ void __launch_bounds__(64) k_jt_plan(Table jt, int k, const uint64_t* __restrict__ starts, unsigned long long n, unsigned long long* counters, UsRec* rec, unsigned long long rec_cap)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t lines = 0;
    jt_plan_start(jt, k, make_kmer(starts[i], k), counters, rec, rec_cap, lines);
}
/* one stored unitig per lane: its sequence into the store */
__global__ void __launch_bounds__(64) k_jt_emit(Table jt, UStore us, int k, const UsRec* __restrict__ rec, unsigned long long n)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t lines = 0;
    jt_emit(jt, us, k, rec[i], lines);
}
/* one stored unitig per wave, its k-mers dealt to the lanes: abundances from the source into the store.  counters[JT_C_SAT] += those above 255 */
template <typename Src>
__global__ void __launch_bounds__(256) k_us_ab(UStore us, int k, const UsRec* __restrict__ rec, unsigned long long n, Src src, unsigned long long* counters)
{
    const unsigned long long wave = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((unsigned long long)gridDim.x * blockDim.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t lines = 0;
    unsigned long long sat = 0;
    for (unsigned long long u = wave; u < n; u += nwaves) {
        const UsRec r = rec[u];
        for (uint32_t i = lane; i < r.len_k; i += 64) sat += us_ab_fill(us, k, r, i, src, lines);
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

__global__ void k_query(Index ix, const uint64_t* __restrict__ kmers, size_t n, uint32_t* abund, uint8_t* succ, uint8_t* pred)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const uint64_t mk1 = kmask(ix.k - 1);
    uint32_t lines = 0;
    for (; i < n; i += stride) {
        Kmer x = make_kmer(kmers[i] & kmask(ix.k), ix.k);
        if (abund) abund[i] = abundance(ix, x, lines);
        if (succ) succ[i] = (uint8_t)adj_right_t(ix.adj, x, mk1, lines).out;
        if (pred) pred[i] = (uint8_t)adj_left(ix, x, mk1, lines).in;
    }
}

/* Index and configuration of the traversal kernel live in constant memory: its code takes them by reference all over (Worker, the
 * bubble routines), and a by-value kernel argument whose address is taken is copied to private memory, which turned every field
 * access into a per-lane scratch load; a reference to a __constant__ object stays a scalar load. */
enum { TRAVERSAL_SETS = mtg_index::NWS }; /* one set of constants per workspace number: that many traversals share the device */
__constant__ Index c_ix[TRAVERSAL_SETS];
__constant__ FillCfg c_cfg[TRAVERSAL_SETS];

/* ---- the traversal: two kernels.
 *
 * k_stage_a, the walk kernel: one gap per lane, one wave per workgroup (waves retire independently).  A lane follows its simple paths
 * (whole unitigs at a time) and answers the strict SNP pattern itself; at any other branching node it PARKS its gap: the walk's state goes
 * to the gap's raw block (WalkSave) and the slot to the launch's work list -- one atomic per wave, the lanes' places from a ballot and a
 * prefix popcount.
 *
 * k_finish, the finishing kernel: a group of G lanes (a wave, or an aligned part of one) takes a parked gap off the work list and runs
 * the rest of its life: frontier expansion one lane per (node, nucleotide) with ballots, visited sets / frontlines / path enumeration /
 * consensuses in LDS (mtg_bubble.h), the walk between two branching nodes by all lanes of the group with the same values.
 *
 * MTG_CLASSIC_WALK=1 (A/B measurements and tests): one kernel, every bubble resolved by its lane from HBM scratch (the round-2 shape). */
/* Registers of the classic form are capped for two waves per SIMD (left alone the compiler takes 289, one wave per SIMD); the walk kernel
 * without the general bubble code needs fewer.  -DMTG_STAGE_A_WAVES=n / -DMTG_WALK_WAVES=n: experiments with another cap. */
#ifndef MTG_STAGE_A_WAVES
#define MTG_STAGE_A_WAVES 2
#endif
#ifndef MTG_WALK_WAVES
#define MTG_WALK_WAVES 2
#endif
#define MTG_STAGE_A_ATTR __attribute__((amdgpu_waves_per_eu(MTG_STAGE_A_WAVES)))
/* device: the work lists of one launch.  count[i] = entries of list i; list i = cap slot numbers at lists + i * cap.  List 2r holds the gaps
 * parked by the r-th launch of the walk kernel (at a branching node), list 2r + 1 those of them whose bubble did not fit the LDS areas. */
enum { PARK_LISTS = 19 }; /* 0 .. 15: the rounds' lists of parked gaps; the last three: gaps with copy commands among those finished late, gaps for k_post's general form, gaps with copy commands to execute */
struct ParkCtl {
    uint32_t count[PARK_LISTS];
#ifdef MTG_BUBBLE_TIMING /* diagnostics build: how long the lanes and the waves of the bubble kernels ran (bins of log2 of 10 ns ticks) */
    uint32_t hist_lane[32], hist_wave[32], hist_walk_lane[32], hist_walk_wave[32];
#endif
};
#ifdef MTG_BUBBLE_TIMING
__device__ __forceinline__ void timing_note(uint32_t* hl, uint32_t* hw, uint64_t t0)
{
    const uint64_t dl = wall_clock64() - t0;
    atomicAdd(&hl[63 - __clzll((long long)(dl | 1ull))], 1u);
    __builtin_amdgcn_wave_barrier();
    const unsigned long long act = __ballot(1);
    if ((int)(threadIdx.x & 63u) == __ffsll((long long)act) - 1) { const uint64_t dw = wall_clock64() - t0; atomicAdd(&hw[63 - __clzll((long long)(dw | 1ull))], 1u); }
}
#endif
__device__ __forceinline__ uint32_t* park_list(ParkCtl* p, uint32_t cap, uint32_t i) { return reinterpret_cast<uint32_t*>(p + 1) + (size_t)i * cap; }
/* the lanes of a wave that park their gap append it to a list: one atomic per wave, the places from a ballot and a prefix popcount */
__device__ __forceinline__ void park_append(ParkCtl* park, uint32_t cap, uint32_t list, bool parked, uint32_t slot)
{
    const unsigned long long pm = __ballot(parked);
    if (!pm) return;
    const int leader = __ffsll((long long)pm) - 1;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(&park->count[list], (uint32_t)__popcll(pm));
    base = (uint32_t)__shfl((int)base, leader, 64);
    if (parked) park_list(park, cap, list)[base + (uint32_t)__popcll(pm & ((1ull << lane) - 1ull))] = slot;
}
/* one gap per lane.  in_list < 0: the gaps of the launch, from their source k-mers; otherwise the gaps of that work list, resumed (a bubble
 * kernel has answered the branching node they stand on).  out_list: where the gaps that park (again) go. */
template <int MODE>
__device__ __forceinline__ void stage_a_lane(uint8_t* zero, uint8_t* raw, uint8_t* ilv, const uint64_t* __restrict__ src, const uint64_t* __restrict__ rwords, const uint32_t* __restrict__ roff,
                                             const uint32_t* __restrict__ rlen, const uint64_t* __restrict__ r0, const uint32_t* __restrict__ ids, GapOut* out, uint32_t n, uint32_t cset,
                                             ParkCtl* park, uint32_t cap, int in_list, uint32_t out_list, uint32_t snp_mode = 1)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t slot = t;
    if (in_list < 0) { if (t >= n) return; }
    else {
        if (t >= park->count[in_list]) return;
        slot = park_list(park, cap, (uint32_t)in_list)[t];
    }
    const Index& ix = c_ix[cset]; /* cset is a kernel argument: still scalar loads */
    const FillCfg& cfg = c_cfg[cset];
    const uint32_t g = ids ? ids[slot] : slot; /* gap id in the input arrays; scratch is indexed by slot */
    GapScratch S = carve(cfg, zero, raw, ilv, slot);
    S.snp_fast = (int)snp_mode; /* 2: park at SNP bubbles too (mtg_traverse.h: park_all) */
    SwfPattern R;
    R.words = rwords + roff[g];
    R.rlen = rlen[g];
    R.r0 = r0[g];
    GapOut o;
#ifdef MTG_BUBBLE_TIMING
    const uint64_t t0 = wall_clock64();
#endif
    stage_a_walk<MODE, 1>(ix, cfg, S, src[g], R, o, nullptr, in_list >= 0);
#ifdef MTG_BUBBLE_TIMING
    if (MODE == WALK_PARK && in_list >= 0) timing_note(park->hist_walk_lane, park->hist_walk_wave, t0);
#endif
    out[slot] = o;
    if (MODE == WALK_PARK) park_append(park, cap, out_list, o.status == GAP_PARKED, slot);
}
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MTG_WALK_WAVES))) k_stage_a(uint8_t* zero, uint8_t* raw, uint8_t* ilv, const uint64_t* __restrict__ src,
                                                const uint64_t* __restrict__ rwords, const uint32_t* __restrict__ roff,
                                                const uint32_t* __restrict__ rlen, const uint64_t* __restrict__ r0, const uint32_t* __restrict__ ids,
                                                GapOut* out, uint32_t n, uint32_t cset, ParkCtl* park, uint32_t cap, int in_list, uint32_t out_list, uint32_t snp_mode)
{
    stage_a_lane<WALK_PARK>(zero, raw, ilv, src, rwords, roff, rlen, r0, ids, out, n, cset, park, cap, in_list, out_list, snp_mode);
}
__global__ void __launch_bounds__(64) MTG_STAGE_A_ATTR k_stage_a_classic(uint8_t* zero, uint8_t* raw, uint8_t* ilv, const uint64_t* __restrict__ src,
                                                const uint64_t* __restrict__ rwords, const uint32_t* __restrict__ roff,
                                                const uint32_t* __restrict__ rlen, const uint64_t* __restrict__ r0, const uint32_t* __restrict__ ids,
                                                GapOut* out, uint32_t n, uint32_t cset)
{
    stage_a_lane<WALK_CLASSIC>(zero, raw, ilv, src, rwords, roff, rlen, r0, ids, out, n, cset, nullptr, 0, -1, 0);
}
/* ---- the rounds between two launches of the walk kernel: the branching nodes of the parked gaps, answered on their own.
 * k_bubble: a group of G lanes per gap of list `in_list`, frontier expansion and path enumeration from LDS (mtg_bubble.h); a bubble that does
 * not fit the LDS areas sends its gap to list in_list + 1, where k_bubble_classic answers it with one lane from HBM scratch. */
#ifndef MTG_BUBBLE_WAVES
#define MTG_BUBBLE_WAVES 4
#endif
template <int G>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MTG_BUBBLE_WAVES))) k_bubble(uint8_t* zero, uint8_t* raw, uint8_t* ilv, uint32_t cset, ParkCtl* park, uint32_t cap, uint32_t in_list)
{
    __shared__ BubbleLds lds[64 / G];
    const uint32_t lane = threadIdx.x & 63u, gl = lane & (uint32_t)(G - 1);
    const uint32_t t = blockIdx.x * (64u / (uint32_t)G) + lane / (uint32_t)G;
    if (t >= park->count[in_list]) return;
    const Index& ix = c_ix[cset];
    const FillCfg& cfg = c_cfg[cset];
    const uint32_t slot = park_list(park, cap, in_list)[t];
    GapScratch S = carve(cfg, zero, raw, ilv, slot);
    S.snp_fast = 1; /* the strict SNP pattern is answered by the fast path here as in the walk */
    const bool done = bubble_coop<G>(ix, cfg, S, lds[lane / G]);
    if (!done && gl == 0) park_list(park, cap, in_list + 1)[atomicAdd(&park->count[in_list + 1], 1u)] = slot;
}
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MTG_STAGE_A_WAVES))) k_bubble_classic(uint8_t* zero, uint8_t* raw, uint8_t* ilv, uint32_t cset, ParkCtl* park, uint32_t cap, uint32_t in_list)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= park->count[in_list]) return;
    const Index& ix = c_ix[cset];
    const FillCfg& cfg = c_cfg[cset];
    const uint32_t slot = park_list(park, cap, in_list)[t];
    GapScratch S = carve(cfg, zero, raw, ilv, slot);
    S.snp_fast = 1; /* the strict SNP pattern is answered by the fast path here as in the walk */
#ifdef MTG_BUBBLE_TIMING
    const uint64_t t0 = wall_clock64();
#endif
    bubble_classic(ix, cfg, S);
#ifdef MTG_BUBBLE_TIMING
    timing_note(park->hist_lane, park->hist_wave, t0);
#endif
}
/* the gaps that are still parked after the rounds (all of them when there are no rounds), one group of G lanes each, to the end of their
 * walks: group i of the grid takes entry i of the list.  The grid is sized for the worst case (the host does not know the count when it
 * queues the kernel); a group without an entry leaves at once.  (A loop over tickets around the walk -- fewer, longer-lived groups -- made
 * this very large kernel hang on the device in every build but an instrumented one; the straight-line form has no control flow around the walk.) */
#ifndef MTG_FINISH_WAVES
#define MTG_FINISH_WAVES 2
#endif
template <int G>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MTG_FINISH_WAVES))) k_finish(uint8_t* zero, uint8_t* raw, uint8_t* ilv, const uint64_t* __restrict__ rwords, const uint32_t* __restrict__ roff,
                                               const uint32_t* __restrict__ rlen, const uint64_t* __restrict__ r0, const uint32_t* __restrict__ ids, GapOut* out, uint32_t cset,
                                               ParkCtl* park, uint32_t cap, uint32_t in_list)
{
    __shared__ BubbleLdsBig lds[64 / G];
    const uint32_t lane = threadIdx.x & 63u, gl = lane & (uint32_t)(G - 1);
    const uint32_t t = blockIdx.x * (64u / (uint32_t)G) + lane / (uint32_t)G;
    if (t >= park->count[in_list]) return;
#ifdef MTG_FINISH_ONE_LANE /* diagnostics: the group is its first lane alone (needs -DMTG_COOP_OFF) */
    if (gl != 0) return;
#endif
    const Index& ix = c_ix[cset];
    const FillCfg& cfg = c_cfg[cset];
    const uint32_t slot = park_list(park, cap, in_list)[t];
    const uint32_t g = ids ? ids[slot] : slot;
    GapScratch S = carve(cfg, zero, raw, ilv, slot);
    S.snp_fast = 1;
    SwfPattern R;
    R.words = rwords + roff[g];
    R.rlen = rlen[g];
    R.r0 = r0[g];
    GapOut o;
    stage_a_walk<WALK_FINISH, G>(ix, cfg, S, 0, R, o, &lds[lane / G]);
    if (gl == 0) out[slot] = o;
}

/* the same with one LANE per parked gap and the general code on HBM scratch (A/B hook, MTG_FINISH_G=1): the group form is faster even
 * for a handful of parked gaps */
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MTG_STAGE_A_WAVES))) k_finish_lane(uint8_t* zero, uint8_t* raw, uint8_t* ilv, const uint64_t* __restrict__ rwords,
                                               const uint32_t* __restrict__ roff, const uint32_t* __restrict__ rlen, const uint64_t* __restrict__ r0, const uint32_t* __restrict__ ids,
                                               GapOut* out, uint32_t cset, ParkCtl* park, uint32_t cap, uint32_t in_list, uint32_t first)
{
    const uint32_t t = first + blockIdx.x * blockDim.x + threadIdx.x; /* entries below `first` belong to the groups of k_finish<G> */
    if (t >= park->count[in_list]) return;
    const Index& ix = c_ix[cset];
    const FillCfg& cfg = c_cfg[cset];
    const uint32_t slot = park_list(park, cap, in_list)[t];
    const uint32_t g = ids ? ids[slot] : slot;
    GapScratch S = carve(cfg, zero, raw, ilv, slot);
    S.snp_fast = 1;
    SwfPattern R;
    R.words = rwords + roff[g];
    R.rlen = rlen[g];
    R.r0 = r0[g];
    GapOut o;
    stage_a_walk<WALK_FINISH, 1>(ix, cfg, S, 0, R, o, nullptr);
    out[slot] = o;
}

/* the long runs the traversal left as commands (mtg_copy.h).  k_lean, one gap per lane: is the target inside a run the walk took (the lean
 * form: nothing is copied, k_post and k_emit read the store)?  The gaps that do need their commands executed go on a work list (ballot +
 * prefix popcount, as for parking).  k_copy, one wave per listed gap, four per workgroup: the grid covers the launch (the host does not
 * know the count), a wave beyond the list leaves after one scalar read. */
enum { COPY_LIST = PARK_LISTS - 1, POST_LIST = PARK_LISTS - 2, COPY_LIST_LATE = PARK_LISTS - 3 };
__global__ void __launch_bounds__(64) k_lean(Index ix, FillCfg cfg, uint8_t* raw, const GapOut* __restrict__ outs, const uint32_t* __restrict__ ids, const uint64_t* __restrict__ tle,
                                             const uint64_t* __restrict__ tbad, const uint32_t* __restrict__ toff, const uint32_t* __restrict__ tcnt, const uint8_t* __restrict__ fast_ok,
                                             uint32_t lean_allowed, uint32_t n, ParkCtl* park, uint32_t cap)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    bool need = false, general = false;
    if (slot < n) {
        GapScratch S;
        S.z = nullptr;
        S.v = nullptr;
        S.lane = 0;
        S.r = raw + (uint64_t)slot * cfg.raw_stride;
        /* the lean form needs one usable target and a source of exactly k nucleotides (what the common-case result of k_post needs anyway) */
        const uint32_t g = ids ? ids[slot] : slot;
        uint64_t target = ~0ull;
        if (lean_allowed && tcnt[g] == 1u && fast_ok[g] && tbad[toff[g]] == 0ull) target = rev_fields64(tle[toff[g]]) >> (64 - 2 * ix.k);
        need = lean_decide(ix, cfg, S, outs[slot], target);
        general = !s_lean(cfg, S)->valid;
    }
    park_append(park, cap, COPY_LIST, need, slot);
    park_append(park, cap, POST_LIST, general, slot); /* every gap that is not lean (a failed one too): k_post's general form */
}
__global__ void __launch_bounds__(256) k_copy(Index ix, FillCfg cfg, uint8_t* raw, const GapOut* __restrict__ outs, ParkCtl* park, uint32_t cap, uint32_t list)
{
    const uint32_t count = park->count[list];
    for (uint32_t t = blockIdx.x * 4u + (threadIdx.x >> 6); t < count; t += gridDim.x * 4u) { /* the grid usually covers the launch; a smaller one (the late list) loops */
        const uint32_t slot = park_list(park, cap, list)[t];
        GapScratch S;
        S.z = nullptr;
        S.v = nullptr;
        S.lane = 0;
        S.r = raw + (uint64_t)slot * cfg.raw_stride;
        copy_cmds(ix, cfg, S, outs[slot]);
    }
}
/* The gaps the finishing kernel has walked while k_lean, k_copy and k_post_lean were busy with all the others (see device_run): their
 * records come over from the finishing kernel's own array, they are never lean (the general k_post / k_emit take them: k_lean has
 * listed them as such when it saw them parked), and the ones with copy commands go on the late list.  One gap per lane. */
__global__ void __launch_bounds__(64) k_late(Index ix, FillCfg cfg, uint8_t* raw, GapOut* outs, const GapOut* __restrict__ finished, ParkCtl* park, uint32_t cap, uint32_t in_list)
{
    const uint32_t count = park->count[in_list];
    for (uint32_t base = blockIdx.x * 64u; base < count; base += gridDim.x * 64u) { /* the same trips for every lane of the wave: the append below is the wave's */
        const uint32_t t = base + threadIdx.x;
        bool need = false;
        uint32_t slot = 0;
        if (t < count) {
            slot = park_list(park, cap, in_list)[t];
            const GapOut o = finished[slot];
            outs[slot] = o;
            GapScratch S;
            S.z = nullptr;
            S.v = nullptr;
            S.lane = 0;
            S.r = raw + (uint64_t)slot * cfg.raw_stride;
            need = lean_decide(ix, cfg, S, o, ~0ull); /* no target: not lean; true when there are commands to execute */
        }
        park_append(park, cap, COPY_LIST_LATE, need, slot);
    }
}

/* mtg_fill_text: a batch whose strings are still text (mtg_marshal.h).  One gap per thread: source k-mer, packed pattern, its first k-mer,
 * whether the fast forms apply; one dictionary entry per thread: little-endian k-mer and never-match mask.  The reads are a few dozen bytes
 * per thread at unrelated places of the block: 100 000 gaps take some tens of microseconds, against 1-2 ms of two host threads. */
__global__ void k_marshal_text(const uint8_t* __restrict__ text, const uint64_t* __restrict__ soff, const uint32_t* __restrict__ slen, const uint64_t* __restrict__ poff,
                               const uint32_t* __restrict__ roff, uint32_t* rlen, uint64_t* src, uint64_t* r0, uint8_t* fast_ok, uint64_t* rw, uint32_t n, int k)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    uint64_t s, r;
    uint32_t rl;
    uint8_t fo;
    marshal_text_gap(text, soff[g], slen[g], poff[g], rlen[g], k, rw + roff[g], s, r, rl, fo);
    src[g] = s; r0[g] = r; rlen[g] = rl; fast_ok[g] = fo;
}
__global__ void k_marshal_targets(const uint8_t* __restrict__ text, const uint64_t* __restrict__ doff, const uint32_t* __restrict__ dlen, uint64_t* __restrict__ tle, uint64_t* __restrict__ tbad,
                                  uint32_t nt, int k)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    uint64_t le, bad;
    marshal_text_target(text, doff[t], dlen[t], k, le, bad);
    tle[t] = le; tbad[t] = bad;
}
/* the targets of a batch from text to (little-endian k-mer, never-match mask): one target per thread */
__global__ void k_encode_targets(const uint8_t* __restrict__ traw, uint64_t* __restrict__ tle, uint64_t* __restrict__ tbad, uint64_t nt, int k)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    uint64_t le, bad;
    encode_target(traw + t * TARGET_SLOT, k, le, bad);
    tle[t] = le;
    tbad[t] = bad;
}

/* terminal-node search + coverage of the single-contig solution, one wave per gap; leaves the slot's record with what the gap will
 * contribute to the arrays of its batch (mtg_emit.h: emit_plan).  Where it goes is decided by the scan kernels below. */
#ifndef MTG_POST_WAVES
#define MTG_POST_WAVES 6
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MTG_POST_WAVES))) k_post(Index ix, FillCfg cfg, uint8_t* raw, const GapOut* __restrict__ outs, const uint32_t* __restrict__ ids,
                                             const uint64_t* __restrict__ tle, const uint64_t* __restrict__ tbad, const uint32_t* __restrict__ toff,
                                             const uint32_t* __restrict__ tcnt, const uint8_t* __restrict__ nbmis, const uint8_t* __restrict__ fast_ok,
                                             uint32_t want_all, SlotRec* recs, uint32_t n, ParkCtl* park)
{
    __shared__ uint32_t hist[256];
    __shared__ uint64_t tile[POST_TILE + 2];
    __shared__ uint64_t s_blk[64];
    /* The gaps of the list k_lean has left (everything that is not lean; k_post_lean below has the others).  One workgroup per gap measured
     * best (against persistent workgroups): the kernel lives on the number of waves in flight -- the host sizes the grid from what the
     * previous launch listed, and the loop takes what a launch lists beyond that. */
    const uint32_t n_listed = park->count[POST_LIST];
    const uint32_t* list = park_list(park, n, POST_LIST);
    for (uint32_t li = blockIdx.x; li < n_listed; li += gridDim.x) {
        const uint32_t slot = list[li];
        __syncthreads(); /* the previous gap's readers of hist are done */
        for (uint32_t i = threadIdx.x; i < 256; i += 64) hist[i] = 0;
        __syncthreads();
        const GapOut o = outs[slot];
        PostOut po;
        po.nb_terminal = po.fast = po.pos = po.errors = po.target = po.clen0 = po.ab_sum = po.ab_n = po.med_hi = po.med_lo = po.lines = po.direct = po.lean = 0;
        GapScratch S;
        S.z = nullptr;
        S.v = nullptr;
        S.lane = 0;
        S.r = raw + (uint64_t)slot * cfg.raw_stride;
        if (o.status == GAP_OK) {
            const uint32_t g = ids ? ids[slot] : slot;
            PostTargets T;
            T.le = tle + toff[g];
            T.bad = tbad + toff[g];
            T.n = tcnt[g];
            T.nb_mis = nbmis[g];
            T.fast_ok = fast_ok[g];
#ifdef MTG_POST_DBG /* timing experiments only (scripts/exp_post_parts.sh): parts of the kernel switched off, results wrong */
            post_gap(ix, cfg, S, o, T, hist, tile, s_blk, po, MTG_POST_DBG);
#else
            post_gap(ix, cfg, S, o, T, hist, tile, s_blk, po);
#endif
        }
        /* the record leaves lane 0 in ten 16-byte stores (SlotRec is 16-byte aligned): field by field it made the kernel write 1.1 KB of
         * partial lines per gap (PMC WRITE_SIZE 111.6 MB per launch for 15 MB of records, round 3).  A coalesced store of the wave through
         * LDS was measured as well: fewer bytes still, but 36 us more -- the kernel is bound by instruction issue, not by its traffic. */
        if (threadIdx.x == 0) {
            SlotRec r;
            r.o = o; r.p = po;
            emit_plan(o, po, want_all != 0, ix.k, r.nw, r.nc, r.asc, r.ext); /* po is uniform over the wave */
            r.wbase = r.cbase = r.abase = r.ebase = 0;
            r.rpos = r.gpos = 0;
            r.fpos = r.pad_ = 0;
            r.pad2_[0] = r.pad2_[1] = 0;
            recs[slot] = r;
        }
    }
}

/* the lean gaps (mtg_post.h: post_lean_*): eight lanes per gap, eight gaps per wave; a gap that is not lean is left to k_post.
 * Measured on the haploid set, k_post + scans of one batch alone: 0.222 ms with a wave per gap (round 3), 0.185 with 4 lanes per gap, 0.128 with 8, 0.136 with 16. */
#ifndef MTG_POST_LEAN_G
#define MTG_POST_LEAN_G 8
#endif
enum { POST_LEAN_G = MTG_POST_LEAN_G };
__global__ void __launch_bounds__(64) k_post_lean(Index ix, FillCfg cfg, uint8_t* raw, const GapOut* __restrict__ outs, uint32_t want_all, SlotRec* recs, uint32_t n)
{
    __shared__ uint32_t hist[64 / POST_LEAN_G][256];
    const uint32_t grp = threadIdx.x / POST_LEAN_G, gl = threadIdx.x % POST_LEAN_G;
    const uint32_t slot = blockIdx.x * (64u / POST_LEAN_G) + grp;
    for (uint32_t i = gl; i < 256u; i += POST_LEAN_G) hist[grp][i] = 0;
    __syncthreads();
    GapOut o;
    LeanWork w;
    bool lean = false;
    if (slot < n) {
        o = outs[slot];
        GapScratch S;
        S.z = nullptr;
        S.v = nullptr;
        S.lane = 0;
        S.r = raw + (uint64_t)slot * cfg.raw_stride;
        lean = post_lean_accumulate<POST_LEAN_G>(ix, cfg, S, o, gl, hist[grp], w);
    }
    __syncthreads(); /* the histograms are complete */
    if (!lean) return;
    PostOut po;
    post_lean_finish<POST_LEAN_G>(w, gl, hist[grp], po);
    if (gl == 0) {
        SlotRec r;
        r.o = o; r.p = po;
        emit_plan(o, po, want_all != 0, ix.k, r.nw, r.nc, r.asc, r.ext);
        r.wbase = r.cbase = r.abase = r.ebase = 0;
        r.rpos = r.gpos = 0;
        r.fpos = r.pad_ = 0;
        r.pad2_[0] = r.pad2_[1] = 0;
        recs[slot] = r;
    }
}

/* ---- where every slot's output goes: exclusive prefix sums, in slot order, of what the slots contribute to the dense words, the dense
 * metadata, the sequence arena, the extension arena, the list of gaps to re-run and the list of multi-contig gaps.  k_scan1: one thread per
 * slot, offsets inside its block of SCAN_SL slots + the block's totals and statistics; k_scan2 (one workgroup): offsets of the blocks on
 * top of the batch's cursors, totals of the launch; k_emit adds the two. */
enum { SCAN_SL = 256, SCAN_NV = 7, SCAN_NS = 16 };
struct ScanBlock {
    uint64_t v[SCAN_NV]; /* k_scan1: totals of the block; k_scan2: replaced by the block's base */
    uint64_t s[SCAN_NS]; /* sums: lines, store_runs, run_nt, contig_nt, contig_words, post_lines, cov_kmers, n_filled, n_ext, copy_words, copy_cmds, cov_direct, n_lean,
                            copy words / commands k_copy executed (not those of lean gaps), contig words k_post scanned (not those of lean gaps) */
};
__global__ void __launch_bounds__(SCAN_SL) k_scan1(SlotRec* recs, uint32_t m, ScanBlock* blocks)
{
    /* scans inside the waves by shuffles, the four waves of the block joined through a few words of LDS; the statistics are reduced the same
     * way (a block's contributions fit 32 bits: 256 slots of at most 2^20 words / bytes each... the sums are kept in 64 bits all the same) */
    enum { NW = SCAN_SL / 64 };
    __shared__ uint64_t wtot[NW][SCAN_NV];
    __shared__ unsigned long long wsum[NW][SCAN_NS];
    const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6, slot = blockIdx.x * SCAN_SL + t;
    uint64_t v[SCAN_NV] = {0, 0, 0, 0, 0, 0, 0};
    unsigned long long st[SCAN_NS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (slot < m) {
        const SlotRec& r = recs[slot];
        const bool ok = r.o.status == GAP_OK;
        v[0] = r.nw; v[1] = r.nc; v[2] = r.asc; v[3] = r.ext;
        v[4] = ok ? 0 : 1;
        v[5] = (ok && r.nc) ? 1 : 0; /* its contigs go back to the host: multi-contig path, or the stage-A entry */
        v[6] = r.asc ? 1 : 0;        /* filled on the common path: one solution */
        st[0] = r.o.lines; st[1] = r.o.store_reads; st[2] = r.o.run_nt; st[4] = r.o.n_words;
        if (r.o.n_cmds) { st[9] = r.o.copy_words; st[10] = r.o.n_cmds; }
        if (ok) {
            st[3] = r.o.total_nt; st[5] = r.p.lines; st[6] = r.p.ab_n;
            if (r.p.direct) st[11] = r.p.ab_n;
            st[7] = r.asc ? 1 : 0; st[8] = r.ext ? 1 : 0; st[12] = r.p.lean ? 1 : 0;
        }
        /* a lean gap's commands are never executed and its contig is never scanned: what k_copy and k_post really touched */
        if (!(ok && r.p.lean)) { if (r.o.n_cmds) { st[13] = r.o.copy_words; st[14] = r.o.n_cmds; } if (ok) st[15] = r.o.n_words; }
    }
    uint64_t incl[SCAN_NV];
    for (int j = 0; j < SCAN_NV; j++) {
        uint64_t x = v[j];
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)x, d, 64), hi = (uint32_t)__shfl_up((int)(uint32_t)(x >> 32), d, 64);
            if ((int)lane >= d) x += ((uint64_t)hi << 32) | lo;
        }
        incl[j] = x;
        if (lane == 63) wtot[wv][j] = x;
    }
    for (int j = 0; j < SCAN_NS; j++) {
        unsigned long long x = st[j];
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)x, d, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(x >> 32), d, 64);
            x += ((unsigned long long)hi << 32) | lo;
        }
        if (lane == 0) wsum[wv][j] = x;
    }
    __syncthreads();
    uint64_t before[SCAN_NV];
    for (int j = 0; j < SCAN_NV; j++) {
        uint64_t x = 0;
        for (uint32_t w2 = 0; w2 < wv; w2++) x += wtot[w2][j];
        before[j] = x;
    }
    if (slot < m) {
        SlotRec& r = recs[slot];
        r.wbase = before[0] + incl[0] - v[0]; r.cbase = before[1] + incl[1] - v[1]; r.abase = before[2] + incl[2] - v[2]; r.ebase = before[3] + incl[3] - v[3];
        r.rpos = (uint32_t)(before[4] + incl[4] - v[4]); r.gpos = (uint32_t)(before[5] + incl[5] - v[5]); r.fpos = (uint32_t)(before[6] + incl[6] - v[6]);
    }
    if (t < SCAN_NV) { uint64_t x = 0; for (int w2 = 0; w2 < NW; w2++) x += wtot[w2][t]; blocks[blockIdx.x].v[t] = x; }
    if (t < SCAN_NS) { unsigned long long x = 0; for (int w2 = 0; w2 < NW; w2++) x += wsum[w2][t]; blocks[blockIdx.x].s[t] = x; }
}
/* cursors[0..3]: words, metadata entries, sequence bytes, extension bytes of the batch so far */
__global__ void __launch_bounds__(256) k_scan2(ScanBlock* blocks, uint32_t nblocks, unsigned long long* cursors, PartTot* tot)
{
    enum { TILE = 1024 };
    __shared__ uint64_t sh[SCAN_NV][TILE];
    __shared__ uint64_t carry[SCAN_NV];
    __shared__ unsigned long long ssum[SCAN_NS];
    const uint32_t t = threadIdx.x;
    if (t < SCAN_NV) carry[t] = t < 4 ? cursors[t] : 0;
    if (t < SCAN_NS) ssum[t] = 0;
    __syncthreads();
    if (t < 4) tot->begin[t] = carry[t];
    for (uint32_t b0 = 0; b0 < nblocks; b0 += TILE) {
        const uint32_t nb = nblocks - b0 < (uint32_t)TILE ? nblocks - b0 : (uint32_t)TILE;
        for (uint32_t i = t; i < nb * SCAN_NV; i += 256) sh[i % SCAN_NV][i / SCAN_NV] = blocks[b0 + i / SCAN_NV].v[i % SCAN_NV];
        for (uint32_t i = t; i < nb * SCAN_NS; i += 256) atomicAdd(&ssum[i % SCAN_NS], (unsigned long long)blocks[b0 + i / SCAN_NS].s[i % SCAN_NS]);
        __syncthreads();
        /* the blocks' totals become their bases: a column per wave (waves 0 and 1 take a second one), 64 blocks per shuffle scan */
        for (uint32_t j = t >> 6; j < SCAN_NV; j += 4) {
            const uint32_t lane = t & 63u;
            uint64_t run = carry[j];
            for (uint32_t c0 = 0; c0 < nb; c0 += 64) {
                const uint32_t i = c0 + lane;
                const uint64_t x0 = i < nb ? sh[j][i] : 0ull;
                uint64_t x = x0;
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)x, d, 64), hi = (uint32_t)__shfl_up((int)(uint32_t)(x >> 32), d, 64);
                    if ((int)lane >= d) x += ((uint64_t)hi << 32) | lo;
                }
                if (i < nb) sh[j][i] = run + x - x0;
                const uint32_t tl = (uint32_t)__shfl((int)(uint32_t)x, 63, 64), th = (uint32_t)__shfl((int)(uint32_t)(x >> 32), 63, 64);
                run += ((uint64_t)th << 32) | tl;
            }
            if (lane == 0) carry[j] = run;
        }
        __syncthreads();
        for (uint32_t i = t; i < nb * SCAN_NV; i += 256) blocks[b0 + i / SCAN_NV].v[i % SCAN_NV] = sh[i % SCAN_NV][i / SCAN_NV];
        __syncthreads();
    }
    if (t < 4) { tot->end[t] = carry[t]; cursors[t] = carry[t]; }
    if (t == 0) {
        tot->n_retry = (uint32_t)carry[4];
        tot->n_general = (uint32_t)carry[5];
        tot->lines = ssum[0]; tot->store_runs = ssum[1]; tot->run_nt = ssum[2]; tot->contig_nt = ssum[3]; tot->contig_words = ssum[4];
        tot->post_lines = ssum[5]; tot->cov_kmers = ssum[6];
        tot->n_filled = (uint32_t)ssum[7]; tot->n_ext = (uint32_t)ssum[8];
        tot->copy_words = ssum[9]; tot->copy_cmds = ssum[10]; tot->cov_direct = ssum[11]; tot->n_lean = ssum[12];
        tot->copy_words_exec = ssum[13]; tot->copy_cmds_exec = ssum[14]; tot->scan_words = ssum[15];
    }
}
/* ---- the tool's text on the device (mtg_format.h): one wave per site.  pass 0: is the site simple, and how many bytes does it add to the
 * three files; pass 1 (after the scan): the bytes, at the site's offsets.  The records are the batch's C-ABI records (their seq pointers
 * are addresses of the HOST arena: the same offsets in the workspace's arena d_seq). */
struct FmtArgs {
    const mtg_gap_result* res;
    const mtg_filled* fil;
    const char* text;            /* the batch's text block on the device */
    const uint64_t* source_off;
    const uint32_t* source_len;
    const uint64_t* name_off;
    const uint32_t* name_len;
    const char* d_seq;           /* the sequence arena on the device */
    uint64_t host_seq;           /* address the arena has (would have) on the host */
    uint64_t seq_used;
    uint64_t host_fil;           /* address of the host's fil array: res[i].filled == host_fil + i * sizeof(mtg_filled) on the common path */
    FmtRec* rec;
    char* out[FMT_STREAMS];
    uint32_t n;
};
__device__ __forceinline__ bool fmt_load_site(const FmtArgs& A, uint32_t i, FmtSite& t)
{
    const mtg_gap_result r = A.res[i];
    /* the common path leaves a gap's one solution in slot i of the fil array; anything else (no solution, a record the host wrote) is not ours */
    if (r.n_filled != 1 || (uint64_t)(uintptr_t)r.filled != A.host_fil + (uint64_t)i * sizeof(mtg_filled)) return false;
    const mtg_filled f = A.fil[i];
    const uint64_t q = (uint64_t)(uintptr_t)f.seq;
    if (q < A.host_seq || q >= A.host_seq + A.seq_used) return false;
    t.name = A.text + A.name_off[i]; t.name_len = A.name_len[i];
    t.source = A.text + A.source_off[i]; t.source_len = A.source_len[i];
    t.seq = A.d_seq + (q - A.host_seq);
    t.seq_len = fmt_strlen(t.seq);
    t.nb_nodes = r.nb_nodes; t.total_nt = r.total_nt; t.nb_terminal = r.nb_terminal; t.has_counts = r.has_solution_counts;
    t.nb_total_filled = r.nb_total_filled; t.nb_reported = r.nb_reported;
    t.qual = f.qual; t.solution_count = f.solution_count; t.avg = f.avg_coverage; t.median = f.median_coverage;
    return fmt_site_simple(t);
}
__global__ void __launch_bounds__(64) k_fmt_size(FmtArgs A)
{
    const uint32_t i = blockIdx.x;
    if (i >= A.n) return;
    FmtSite t;
    FmtCount c;
    c.n[0] = c.n[1] = c.n[2] = 0;
    const bool simple = fmt_load_site(A, i, t);
    if (simple) format_site(c, t);
    if (threadIdx.x == 0) { FmtRec r; r.size[0] = c.n[0]; r.size[1] = c.n[1]; r.size[2] = c.n[2]; r.simple = simple ? 1u : 0u; r.off[0] = r.off[1] = r.off[2] = 0; A.rec[i] = r; }
}
/* exclusive prefix sums of the three sizes in site order, the list of the sites left to the host with the offsets where their text belongs;
 * tot[0..2] = bytes, tot[3] = simple sites, tot[4] = sites for the host.  One workgroup. */
__global__ void __launch_bounds__(1024) k_fmt_scan(FmtRec* rec, uint32_t n, uint32_t* cplx, uint64_t* cplx_off, unsigned long long* tot)
{
    __shared__ unsigned long long wsum[16][4];
    __shared__ unsigned long long carry[4];
    const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    if (t < 4) carry[t] = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < n; b0 += 1024) {
        const uint32_t i = b0 + t;
        unsigned long long v[4] = {0, 0, 0, 0};
        if (i < n) { v[0] = rec[i].size[0]; v[1] = rec[i].size[1]; v[2] = rec[i].size[2]; v[3] = rec[i].simple ? 0 : 1; }
        unsigned long long incl[4];
        for (int j = 0; j < 4; j++) {
            unsigned long long x = v[j];
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)x, d, 64), hi = (uint32_t)__shfl_up((int)(uint32_t)(x >> 32), d, 64);
                if ((int)lane >= d) x += ((unsigned long long)hi << 32) | lo;
            }
            incl[j] = x;
            if (lane == 63) wsum[wv][j] = x;
        }
        __syncthreads();
        unsigned long long before[4];
        for (int j = 0; j < 4; j++) {
            unsigned long long x = carry[j];
            for (uint32_t w2 = 0; w2 < wv; w2++) x += wsum[w2][j];
            before[j] = x;
        }
        if (i < n) {
            const unsigned long long o0 = before[0] + incl[0] - v[0], o1 = before[1] + incl[1] - v[1], o2 = before[2] + incl[2] - v[2];
            rec[i].off[0] = o0; rec[i].off[1] = o1; rec[i].off[2] = o2;
            if (v[3]) { const unsigned long long c = before[3] + incl[3] - 1; cplx[c] = i; cplx_off[3 * c] = o0; cplx_off[3 * c + 1] = o1; cplx_off[3 * c + 2] = o2; }
        }
        __syncthreads();
        if (t < 4) { unsigned long long x = carry[t]; for (int w2 = 0; w2 < 16; w2++) x += wsum[w2][t]; carry[t] = x; }
        __syncthreads();
    }
    if (t < 3) tot[t] = carry[t];
    if (t == 3) { tot[4] = carry[3]; tot[3] = (unsigned long long)n - carry[3]; }
}
__global__ void __launch_bounds__(64) k_fmt_write(FmtArgs A)
{
    const uint32_t i = blockIdx.x;
    if (i >= A.n) return;
    const FmtRec r = A.rec[i];
    if (!r.simple) return;
    FmtSite t;
    if (!fmt_load_site(A, i, t)) return;
    FmtWrite w;
    w.p[0] = A.out[0] + r.off[0]; w.p[1] = A.out[1] + r.off[1]; w.p[2] = A.out[2] + r.off[2];
    format_site(w, t);
}

/* everything a gap leaves behind (mtg_emit.h: emit_gap), one wave per slot */
__global__ void __launch_bounds__(64) k_emit(UStore us, FillCfg cfg, uint8_t* raw, SlotRec* recs, const ScanBlock* __restrict__ blocks, const uint32_t* __restrict__ ids,
                                             const uint8_t* __restrict__ gflags, int k, EmitDev D, EmitHost H, uint32_t* retry_list, uint32_t* general_list, uint32_t n, ParkCtl* park,
                                             uint32_t use_list)
{
    /* use_list: the gaps of k_lean's list (everything that is not lean: k_emit_lean has the others), the grid sized by the host from what the
     * previous launch listed; otherwise (a batch that leaves in relocatable form) every slot of the launch */
    const uint32_t count = use_list ? park->count[POST_LIST] : n;
    __shared__ SlotRec r;
    for (uint32_t li = blockIdx.x; li < count; li += gridDim.x) {
    const uint32_t slot = use_list ? park_list(park, n, POST_LIST)[li] : li;
    __syncthreads(); /* the previous gap's readers of r are done */
    if (threadIdx.x == 0) {
        r = recs[slot];
        const ScanBlock& b = blocks[slot / SCAN_SL];
        r.wbase += b.v[0]; r.cbase += b.v[1]; r.abase += b.v[2]; r.ebase += b.v[3];
        r.rpos += (uint32_t)b.v[4]; r.gpos += (uint32_t)b.v[5]; r.fpos += (uint32_t)b.v[6];
        recs[slot] = r; /* absolute from here on (the host reads the records of the gaps it has to look at) */
        if (r.o.status != GAP_OK) retry_list[r.rpos] = slot;
        else if (r.nc) general_list[r.gpos] = slot;
    }
    __syncthreads();
    GapScratch S;
    S.z = nullptr;
    S.v = nullptr;
    S.lane = 0;
    S.r = raw + (uint64_t)slot * cfg.raw_stride;
    const uint32_t g = ids ? ids[slot] : slot;
    emit_gap(us, cfg, S, r, gflags[g], slot, g, k, D, H);
    }
}
/* the lean gaps (mtg_emit.h: emit_lean): eight lanes per gap, eight gaps per wave (k_emit of one haploid batch alone: 0.063 ms with 4 lanes per gap, 0.055 with 8, 0.052 with 16) */
#ifndef MTG_EMIT_LEAN_G
#define MTG_EMIT_LEAN_G 8
#endif
enum { EMIT_LEAN_G = MTG_EMIT_LEAN_G };
__global__ void __launch_bounds__(64) k_emit_lean(UStore us, FillCfg cfg, uint8_t* raw, const SlotRec* __restrict__ recs, const ScanBlock* __restrict__ blocks, const uint32_t* __restrict__ ids,
                                                  const uint8_t* __restrict__ gflags, int k, EmitDev D, EmitHost H, uint32_t n)
{
    const uint32_t slot = blockIdx.x * (64u / EMIT_LEAN_G) + threadIdx.x / EMIT_LEAN_G, gl = threadIdx.x % EMIT_LEAN_G;
    if (slot >= n) return;
    const SlotRec r = recs[slot];
    if (!emit_is_lean(r, D)) return;
    GapScratch S;
    S.z = nullptr;
    S.v = nullptr;
    S.lane = 0;
    S.r = raw + (uint64_t)slot * cfg.raw_stride;
    const uint32_t g = ids ? ids[slot] : slot;
    emit_lean<EMIT_LEAN_G>(us, cfg, S, r, r.abase + blocks[slot / SCAN_SL].v[2], gflags[g], slot, g, k, D, H, gl);
}

/* checksum of a relocatable batch's body into its header (mtg_wire_header::checksum): a sum of scrambled 64-bit words, any order */
__global__ void __launch_bounds__(256) k_wire_sum(uint8_t* wire, uint64_t cap)
{
    mtg_wire_header* h = reinterpret_cast<mtg_wire_header*>(wire);
    if (h->magic != 0x3145524957474D54ull || h->total_bytes > cap) return; /* the batch did not leave in this form (the host knows from the totals) */
    const uint64_t* w = reinterpret_cast<const uint64_t*>(wire + sizeof(mtg_wire_header));
    const uint64_t n = (h->total_bytes - sizeof(mtg_wire_header)) / 8;
    uint64_t s = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) s += wire_word_sum(w[i], i);
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)s, d, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(s >> 32), d, 64);
        s += ((uint64_t)hi << 32) | lo;
    }
    if ((threadIdx.x & 63u) == 0 && s) atomicAdd(reinterpret_cast<unsigned long long*>(&h->checksum), (unsigned long long)s);
}

/* contig-graph walk of the multi-contig gaps of a chunk (mtg_paths.h), one wave per gap */
__global__ void __launch_bounds__(64) k_paths(FillCfg cfg, uint8_t* raw, const GapOut* __restrict__ outs, const uint32_t* __restrict__ slots, int k, uint32_t* out, uint32_t n)
{
    __shared__ PathsWork W;
    if (blockIdx.x >= n) return;
    const uint32_t slot = slots[blockIdx.x];
    GapScratch S;
    S.z = nullptr;
    S.v = nullptr;
    S.lane = 0;
    S.r = raw + (uint64_t)slot * cfg.raw_stride;
    paths_gap(cfg, S, outs[slot], k, W, out + (uint64_t)blockIdx.x * PATHS_WORDS);
}

/* Needleman-Wunsch match count of src/Utils.cpp:87-189, one wave per sequence pair (a = rows, b = columns), exact for any length.
 * The matrix is swept in strips of 64 columns; inside a strip lane l owns column j0+l+1 and works on row t-l at step t, so that the
 * cell to its left (lane l-1, previous step) and the diagonal one (lane l-1, two steps ago) arrive by a one-lane shift and the cell
 * above is its own previous value.  The column left of a strip is kept in `bnd` (score, matches per row), read 64 rows at a time and
 * overwritten by lane 63 as the strip advances (row i is read at step i and rewritten at step i+63).  Scores are the reference's
 * floats times one (all multiples of 5: exact in int); ties are broken diagonal, up, left like the traceback of :150-180. */
__global__ void __launch_bounds__(64) k_nw(const uint8_t* __restrict__ text, const uint64_t* __restrict__ off_a, const uint32_t* __restrict__ len_a,
                                           const uint64_t* __restrict__ off_b, const uint32_t* __restrict__ len_b, int2* bnd_base,
                                           const uint64_t* __restrict__ bnd_off, uint32_t* out, uint32_t npairs)
{
    const uint32_t pair = blockIdx.x, lane = threadIdx.x;
    if (pair >= npairs) return;
    const uint8_t* a = text + off_a[pair];
    const uint8_t* b = text + off_b[pair];
    const uint32_t na = len_a[pair], nb = len_b[pair];
    int2* bnd = bnd_base + bnd_off[pair];
    if (na == 0 || nb == 0) { if (lane == 0) out[pair] = 0; return; }
    for (uint32_t i = lane; i <= na; i += 64) bnd[i] = make_int2(-5 * (int)i, 0); /* column 0 */
    __syncthreads();
    int result = 0;
    for (uint32_t j0 = 0; j0 < nb; j0 += 64) {
        const uint32_t j = j0 + lane + 1; /* 1-based column of this lane */
        const bool col_ok = j <= nb;
        const uint32_t bj = col_ok ? b[j - 1] : 256u;
        int s_up = -5 * (int)j, m_up = 0;  /* cell above: row 0 to start with */
        int s_cur = 0, m_cur = 0;          /* this lane's latest cell, handed to the right-hand neighbour at the next step */
        int s_diag = 0, m_diag = 0;
        uint32_t a_cur = 0;                /* the row character travels with the wavefront */
        int2 bchunk = make_int2(0, 0);
        uint32_t achunk = 0;
        const uint32_t nsteps = na + 63;
        for (uint32_t t = 1; t <= nsteps; t++) {
            if (((t - 1) & 63u) == 0) { /* next 64 rows of the left boundary column and of a */
                const uint32_t r = t + lane;
                bchunk = r <= na ? bnd[r] : make_int2(0, 0);
                achunk = r <= na ? a[r - 1] : 257u;
            }
            int s_left = __shfl_up(s_cur, 1, 64), m_left = __shfl_up(m_cur, 1, 64);
            uint32_t a_in = (uint32_t)__shfl_up((int)a_cur, 1, 64);
            const int src = (int)((t - 1) & 63u);
            const int bs = __shfl(bchunk.x, src, 64), bm = __shfl(bchunk.y, src, 64);
            const uint32_t ba = (uint32_t)__shfl((int)achunk, src, 64);
            if (lane == 0) { s_left = bs; m_left = bm; a_in = ba; }
            a_cur = a_in;
            const int i = (int)t - (int)lane; /* row of this lane */
            if (i == 1) { s_diag = -5 * ((int)j - 1); m_diag = 0; } /* row 0 */
            if (col_ok && i >= 1 && i <= (int)na) {
                const bool eq = a_cur == bj;
                const int diag = s_diag + (eq ? 10 : -5), del = s_up - 5, ins = s_left - 5;
                const int best = max(max(diag, del), ins);
                const int m = best == diag ? m_diag + (eq ? 1 : 0) : (best == del ? m_up : m_left);
                s_cur = best; m_cur = m;
                s_up = best; m_up = m;
                if (lane == 63) bnd[i] = make_int2(best, m);
                if (i == (int)na && j == nb) result = m;
            }
            s_diag = s_left; m_diag = m_left;
        }
        __syncthreads(); /* lane 63's column is the next strip's boundary */
    }
    /* the final cell was computed by lane (nb - 1) % 64 */
    result = __shfl(result, (int)((nb - 1) & 63u), 64);
    if (lane == 0) out[pair] = (uint32_t)result;
}

/* membership scan along packed sequences: rolling k-mer per position, minimizer-blocked Bloom with the blocks of a 256-position tile
 * staged in LDS by coalesced 64-byte reads, optional exact confirmation in the ABND table.
 * counters: [0] k-mers, [1] Bloom positives, [2] confirmed, [3] blocks staged */
enum { SCAN_TILE = 256 };
__global__ void __launch_bounds__(SCAN_TILE) k_scan(Index ix, const uint64_t* __restrict__ words, const uint64_t* __restrict__ word_off,
                                                    const uint32_t* __restrict__ len, size_t nseq, int mode, uint64_t* out_bits, unsigned long long* counters)
{
    __shared__ uint32_t s_blk[SCAN_TILE][16];
    __shared__ uint64_t s_bid[SCAN_TILE];
    __shared__ uint64_t s_slot_bid[SCAN_TILE];
    __shared__ uint32_t s_wave_cnt[SCAN_TILE / 64];
    const int k = ix.k;
    const uint64_t mk = kmask(k);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    unsigned long long n_k = 0, n_pos = 0, n_conf = 0, n_staged = 0;
    for (size_t s = blockIdx.x; s < nseq; s += gridDim.x) {
        const uint32_t L = len[s];
        if (L < (uint32_t)k) continue;
        const uint64_t* w = words + word_off[s];
        uint64_t* ob = out_bits + word_off[s];
        const uint32_t npos = L - (uint32_t)k + 1;
        for (uint32_t base = 0; base < npos; base += SCAN_TILE) {
            const uint32_t p = base + tid;
            const bool valid = p < npos;
            Kmer x;
            x.f = x.r = 0;
            uint64_t b = ~0ull;
            if (valid) {
                x.r = le_kmer(w, p, mk) ^ (0xAAAAAAAAAAAAAAAAULL & mk);
                x.f = revcomp(x.r, k);
                b = bloom_block(ix.bloom, x, k);
            }
            s_bid[tid] = b;
            __syncthreads();
            const bool leader = valid && (tid == 0 || s_bid[tid - 1] != b);
            const unsigned long long bal = __ballot(leader);
            const uint32_t prefix = (uint32_t)__popcll(bal & ((lane == 63u) ? ~0ull : ((2ull << lane) - 1ull)));
            if (lane == 0) s_wave_cnt[wave] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t woff = 0, total = 0;
            for (uint32_t i = 0; i < SCAN_TILE / 64; i++) { if (i < wave) woff += s_wave_cnt[i]; total += s_wave_cnt[i]; }
            const uint32_t slot = woff + prefix - 1u; /* a follower at the start of a wave continues the last block of the previous wave */
            if (leader) s_slot_bid[slot] = b;
            __syncthreads();
            for (uint32_t i = tid; i < total * 16u; i += SCAN_TILE) s_blk[i >> 4][i & 15u] = ix.bloom.bits[s_slot_bid[i >> 4] * 16u + (i & 15u)];
            __syncthreads();
            bool res = false;
            if (valid) {
                const uint64_t c = canon(x);
                res = bloom_test_block(s_blk[slot], bloom_bits(c));
                n_k++;
                n_pos += res;
                if (res && mode == 1) {
                    uint32_t lines = 0;
                    res = abundance(ix, x, lines) != 0;
                    n_conf += res;
                }
            }
            const unsigned long long rb = __ballot(res);
            if (lane == 0 && base + wave * 64u < npos) ob[(base >> 6) + wave] = rb;
            if (tid == 0) n_staged += total;
            __syncthreads();
        }
    }
    /* per-workgroup totals */
    __shared__ unsigned long long s_tot[4];
    if (tid < 4) s_tot[tid] = 0;
    __syncthreads();
    atomicAdd(&s_tot[0], n_k); atomicAdd(&s_tot[1], n_pos); atomicAdd(&s_tot[2], n_conf); atomicAdd(&s_tot[3], n_staged);
    __syncthreads();
    if (tid < 4 && s_tot[tid]) atomicAdd(&counters[tid], s_tot[tid]);
}

/* dependent chains of random line reads: the access pattern of the simple-path walk.  LINE = bytes read per step (16..128) */
template <int LINE>
__global__ void __launch_bounds__(64) k_chase(const uint64_t* __restrict__ table, uint64_t nlines, uint64_t n_chains, uint32_t chain_len, uint64_t* sink)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_chains) return;
    uint64_t x = d_splitmix64(t + 1);
    uint64_t acc = 0;
    for (uint32_t i = 0; i < chain_len; i++) {
        const uint64_t line = x % nlines;
        const U64x2* p = reinterpret_cast<const U64x2*>(table + line * (LINE / 8));
        uint64_t v = 0;
#pragma unroll
        for (int j = 0; j < LINE / 16; j++) { const U64x2 q = p[j]; v ^= q.x ^ q.y; }
        acc += v;
        x = d_splitmix64(x ^ v);
    }
    if (acc == 0x123456789ull) sink[0] = acc;
}

__global__ void k_fill_random(uint64_t* p, uint64_t nwords, uint64_t seed)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < nwords; i += stride) p[i] = d_splitmix64(seed + i);
}

/* ------------------------------------------------------------------------------------------------ index */
namespace {
/* owning device buffer: freed on every exit path */
/* wall time this thread has spent in hipMalloc / hipFree (an index construction reports it: BuildProf) */
static thread_local double tl_alloc_ms = 0;
struct AllocTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~AllocTimer() { tl_alloc_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
static hipError_t timed_malloc(void** p, size_t bytes)
{
    const bool dbg = tune::on(tune::T_DEBUG_TIMERS);
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e;
    { AllocTimer t; e = hipMalloc(p, bytes); }
    if (dbg && bytes > ((size_t)64 << 20)) fprintf(stderr, "  [alloc] hipMalloc %.2f GB: %.1f ms\n", bytes / 1e9, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return e;
}
static hipError_t timed_free(void* p)
{
    const bool dbg = tune::on(tune::T_DEBUG_TIMERS);
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e;
    { AllocTimer t; e = hipFree(p); }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (dbg && ms > 5.0) fprintf(stderr, "  [alloc] hipFree: %.1f ms\n", ms);
    return e;
}
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    DevBuf() {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)timed_free(p); }
    hipError_t alloc(size_t bytes) { if (p) { (void)timed_free(p); p = nullptr; cap = 0; } const hipError_t e = timed_malloc(&p, bytes ? bytes : 8); if (e == hipSuccess) cap = bytes ? bytes : 8; return e; }
    void* release() { void* q = p; p = nullptr; cap = 0; return q; }
    /* takes over the memory of another buffer */
    void adopt(DevBuf& o) { if (p) (void)timed_free(p); p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; }
    template <typename T> T* as() { return (T*)p; }
};
} // namespace

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

static int alloc_tables(mtg_index* idx, uint64_t nkeys, double load_scale)
{
    const int k = idx->dev.k;
    /* ADJ entries are 16 bytes and looked up on a dependent chain: keep buckets sparse; ABND is only read by independent queries */
    const double load_adj = tune::f(tune::T_ADJ_LOAD, 0.5) * load_scale;
    const double load = tune::f(tune::T_ABND_LOAD, 0.6) * load_scale;
    table_shape(idx->dev.adj, buckets_for(nkeys + nkeys / 8 + 1024, load_adj, 2 * (k - 1), MTG_ADJ_SLOTS), 2 * (k - 1));
    table_shape(idx->dev.abnd, buckets_for(nkeys, load, 2 * k, MTG_ABND_SLOTS), 2 * k);
    const size_t ba = idx->dev.adj.nbuckets * 16 * MTG_ADJ_SLOTS, bb = idx->dev.abnd.nbuckets * 8 * MTG_ABND_SLOTS;
    HIP_TRY(hipMalloc((void**)&idx->dev.adj.slots, ba));   /* a failure further down leaves the pointers to free_tables (IndexGuard) */
    HIP_TRY(hipMalloc((void**)&idx->dev.abnd.slots, bb));
    const double bpk = tune::f(tune::T_BLOOM_BITS, 12.0);
    size_t bc = 0;
    idx->dev.bloom.bits = nullptr;
    idx->dev.bloom.nblocks = 0;
    if (bpk > 0) {
        bloom_shape(idx->dev.bloom, nkeys, bpk, k);
        bc = idx->dev.bloom.nblocks * 64;
        HIP_TRY(hipMalloc((void**)&idx->dev.bloom.bits, bc));
        HIP_TRY(hipMemsetAsync(idx->dev.bloom.bits, 0, bc, 0));
    }
    HIP_TRY(hipMemsetAsync(idx->dev.adj.slots, 0, ba, 0));
    HIP_TRY(hipMemsetAsync(idx->dev.abnd.slots, 0, bb, 0));
    idx->info.device_bytes = ba + bb + bc;
    idx->info.bloom_blocks = idx->dev.bloom.nblocks;
    idx->info.bloom_minimizer = (uint32_t)idx->dev.bloom.mm;
    idx->info.adj_buckets = idx->dev.adj.nbuckets;
    idx->info.abnd_buckets = idx->dev.abnd.nbuckets;
    idx->info.adj_bucket_bytes = 16 * MTG_ADJ_SLOTS;
    idx->info.abnd_bucket_bytes = 8 * MTG_ABND_SLOTS;
    return MTG_OK;
}

/* Unitig store of a finished index (every k-mer inserted, every lookahead written): chain starts -> one walk per start -> sequences ->
 * abundances and junction pointers (mtg_dev.h: us_*).  Also fills nb_solid_kmers / nb_branching from the table itself.
 * MTG_NO_UNITIGS=1 (test hook) leaves the index with inline lookaheads only. */
namespace {
/* the k-mers of no stored unitig of a construction that can only name them once the unitigs' entries are in the new tables (lean build, a
 * closed or over-long chain in the graph): called with the sparse ADJ table holding every unitig pointer; fills the lists */
struct LateLeftovers {
    unsigned long long n_upper = 0; /* bound on their number, for the shape of the tables */
    std::function<int(const Index& nx, DevBuf& left_k, DevBuf& left_a, unsigned long long& n_left)> collect;
};
} // namespace
static int sparsify(mtg_index* idx, const UsRec* d_rec, unsigned long long n_rec, bool from_container, const uint64_t* d_left_k, const uint32_t* d_left_a, unsigned long long n_left,
                    BuildProf* prof = nullptr, const LateLeftovers* late = nullptr, DevBuf* adj_reuse = nullptr);
static int build_unitigs(mtg_index* idx)
{
    DevBuf d_cnt, d_starts, d_rec;
    HIP_TRY(d_cnt.alloc(64));
    HIP_TRY(hipMemset(d_cnt.p, 0, 64));
    const uint64_t nslots = idx->dev.abnd.nbuckets * MTG_ABND_SLOTS;
    const unsigned blocks = (unsigned)std::min<uint64_t>((nslots + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(k_us_starts, dim3(blocks), dim3(256), 0, 0, idx->dev, d_cnt.as<unsigned long long>(), (uint64_t*)nullptr, 0ull);
    HIP_TRY(hipGetLastError());
    unsigned long long cnt[8];
    HIP_TRY(hipMemcpy(cnt, d_cnt.p, 64, hipMemcpyDeviceToHost));
    idx->info.nb_solid_kmers = cnt[2];
    idx->info.nb_branching = cnt[1];
    idx->info.nb_unitigs = 0;
    idx->info.unitig_bytes = 0;
    const unsigned long long n_starts = cnt[0];
    if (n_starts == 0 || tune::on(tune::T_NO_UNITIGS)) return MTG_OK;
    HIP_TRY(d_starts.alloc(n_starts * 8));
    hipLaunchKernelGGL(k_us_starts, dim3(blocks), dim3(256), 0, 0, idx->dev, d_cnt.as<unsigned long long>(), d_starts.as<uint64_t>(), n_starts);
    HIP_TRY(hipGetLastError());
    const unsigned long long rec_cap = n_starts / 2 + 1; /* two starts per stored unitig (one per strand) */
    HIP_TRY(d_rec.alloc(rec_cap * sizeof(UsRec)));
    HIP_TRY(hipMemset(d_cnt.p, 0, 64));
    hipLaunchKernelGGL(k_us_plan, dim3((unsigned)((n_starts + 63) / 64)), dim3(64), 0, 0, idx->dev, d_starts.as<uint64_t>(), n_starts, d_cnt.as<unsigned long long>(), d_rec.as<UsRec>(), rec_cap);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(cnt, d_cnt.p, 64, hipMemcpyDeviceToHost));
    const unsigned long long n_words = cnt[0], n_rec = cnt[1];
    if (n_rec > rec_cap) { set_error("unitig construction: %llu records for %llu chain starts", n_rec, n_starts); return MTG_ERR_OVERFLOW; }
    (void)d_starts.alloc(0);
    if (n_rec == 0) return MTG_OK;
    /* a few words of padding: the coverage pass may look up to 64 + k nucleotides past the end of a unitig */
    const unsigned long long pad = 8;
    HIP_TRY(hipMalloc((void**)&idx->dev.us.words, (n_words + pad) * 8));
    HIP_TRY(hipMalloc((void**)&idx->dev.us.ab, (n_words + pad) * 32));
    HIP_TRY(hipMemsetAsync(idx->dev.us.words, 0, (n_words + pad) * 8, 0));
    HIP_TRY(hipMemsetAsync(idx->dev.us.ab, 0, (n_words + pad) * 32, 0));
    idx->dev.us.nwords = n_words;
    idx->dev.us.nunitigs = n_rec;
    hipLaunchKernelGGL(k_us_emit, dim3((unsigned)((n_rec + 63) / 64)), dim3(64), 0, 0, idx->dev, d_rec.as<UsRec>(), n_rec);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_us_link, dim3((unsigned)std::min<unsigned long long>((n_rec + 3) / 4, 256 * 64)), dim3(256), 0, 0, idx->dev, d_rec.as<UsRec>(), n_rec);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    idx->info.nb_unitigs = n_rec;
    idx->info.unitig_bytes = (n_words + pad) * 40;
    idx->info.device_bytes += idx->info.unitig_bytes;
    /* the dense tables have served: the index proper is the store plus the few k-mers of no unitig (MTG_DENSE_INDEX=1: A/B and test hook) */
    if (tune::on(tune::T_DENSE_INDEX)) return MTG_OK;
    return sparsify(idx, d_rec.as<UsRec>(), n_rec, false, nullptr, nullptr, 0);
}

/* The sparse form.  From a dense index with its store (from_container == false: the k-mers of no unitig are read off its ABND table, the
 * dense tables are freed at the end) or from the store alone (an index out of its container: idx holds only the store; the k-mers of no
 * unitig are handed over, the Bloom filter is filled here).  New tables: ADJ with the entries the sparse form keeps, ABND with the k-mers
 * of no unitig. */
static int sparsify(mtg_index* idx, const UsRec* d_rec, unsigned long long n_rec, bool from_container, const uint64_t* d_left_k, const uint32_t* d_left_a, unsigned long long n_left,
                    BuildProf* prof, const LateLeftovers* late, DevBuf* adj_reuse)
{
    const int k = idx->dev.k;
    DevBuf d_cnt, own_k, own_a;
    HIP_TRY(d_cnt.alloc(64));
    if (!from_container) {
        if (prof) prof->begin();
        HIP_TRY(hipMemset(d_cnt.p, 0, 64));
        const uint64_t nslots = idx->dev.abnd.nbuckets * MTG_ABND_SLOTS;
        const unsigned blocks = (unsigned)std::min<uint64_t>((nslots + 255) / 256, 256 * 32);
        hipLaunchKernelGGL(k_leftovers, dim3(blocks), dim3(256), 0, 0, idx->dev, (uint64_t*)nullptr, (uint32_t*)nullptr, d_cnt.as<unsigned long long>(), 0ull);
        HIP_TRY(hipMemcpy(&n_left, d_cnt.p, 8, hipMemcpyDeviceToHost));
        HIP_TRY(own_k.alloc((n_left + 1) * 8));
        HIP_TRY(own_a.alloc((n_left + 1) * 4));
        HIP_TRY(hipMemset(d_cnt.p, 0, 64));
        hipLaunchKernelGGL(k_leftovers, dim3(blocks), dim3(256), 0, 0, idx->dev, own_k.as<uint64_t>(), own_a.as<uint32_t>(), d_cnt.as<unsigned long long>(), (unsigned long long)n_left);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        d_left_k = own_k.as<uint64_t>();
        d_left_a = own_a.as<uint32_t>();
        if (prof) HIP_TRY(prof->end("leftovers_dense", 2 * nslots * 8, nslots));
    }
    const unsigned long long n_left_shape = late ? std::max(late->n_upper, n_left) : n_left;
    /* entries of the new ADJ: per unitig its kept interior junctions (every second one and the last) and its two ends; two per k-mer of no unitig */
    uint64_t nkeys = 2 * n_left_shape + 1024, n_unitig_kmers = 0;
    {
        std::vector<UsRec> h_rec(n_rec);
        if (n_rec) HIP_TRY(hipMemcpy(h_rec.data(), d_rec, n_rec * sizeof(UsRec), hipMemcpyDeviceToHost));
        for (const UsRec& r : h_rec) { nkeys += r.len_k / 2 + 3; n_unitig_kmers += r.len_k; }
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
        if (n_rec) hipLaunchKernelGGL(k_sparse_link, dim3((unsigned)std::min<unsigned long long>((n_rec + 3) / 4, 256 * 64)), dim3(256), 0, 0, nx, d_rec, n_rec, from_container ? 1 : 0, d_cnt.as<unsigned long long>());
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
static int build_from_jt(mtg_index* idx, DevBuf& jt_buf, const Table& jt, const Src& src, const std::function<void()>& release_source, uint64_t sat_at_insert, BuildProf& prof)
{
    const int k = idx->dev.k;
    const uint64_t nslots = jt.nbuckets * MTG_ABND_SLOTS;
    const unsigned scan_blocks = (unsigned)std::min<uint64_t>((nslots + 255) / 256, 256 * 32);
    DevBuf d_cnt, d_starts, d_rec, d_left_k, d_left_a;
    HIP_TRY(d_cnt.alloc(JT_C_N * 8));
    HIP_TRY(hipMemset(d_cnt.p, 0, JT_C_N * 8));
    unsigned long long* cnt_d = d_cnt.as<unsigned long long>();
    unsigned long long cnt[JT_C_N];
    /* chain starts, k-mers of no chain, statistics: counted, then collected */
    prof.begin();
    hipLaunchKernelGGL(k_jt_scan<Src>, dim3(scan_blocks), dim3(256), 0, 0, jt, k, src, cnt_d, 0, (uint64_t*)nullptr, 0ull, (uint64_t*)nullptr, (uint32_t*)nullptr, 0ull);
    HIP_TRY(prof.end("jt_scan_count", nslots * 8, nslots));
    HIP_TRY(hipMemcpy(cnt, d_cnt.p, sizeof cnt, hipMemcpyDeviceToHost));
    const unsigned long long n_starts = cnt[JT_C_STARTS], n_single = cnt[JT_C_LEFT], interior = cnt[JT_C_INTERIOR];
    idx->info.nb_solid_kmers = (cnt[JT_C_ORIENTED] + cnt[JT_C_SELF]) / 2;
    idx->info.nb_branching = (2 * cnt[JT_C_IN_NOT1] - cnt[JT_C_BOTH_NOT1] + cnt[JT_C_SELF_BRANCH]) / 2;
    idx->info.nb_unitigs = 0;
    idx->info.unitig_bytes = 0;
    HIP_TRY(d_starts.alloc((n_starts + 1) * 8));
    HIP_TRY(d_left_k.alloc((n_single + 1) * 8));
    HIP_TRY(d_left_a.alloc((n_single + 1) * 4));
    HIP_TRY(hipMemset(cnt_d + JT_C_STARTS, 0, 8));
    HIP_TRY(hipMemset(cnt_d + JT_C_LEFT, 0, 8));
    prof.begin();
    hipLaunchKernelGGL(k_jt_scan<Src>, dim3(scan_blocks), dim3(256), 0, 0, jt, k, src, cnt_d, 1, d_starts.as<uint64_t>(), n_starts, d_left_k.as<uint64_t>(), d_left_a.as<uint32_t>(), n_single);
    HIP_TRY(prof.end("jt_scan_collect", nslots * 8 + n_starts * 8 + n_single * 12, nslots));
    /* abundances above 255: of the single k-mers (counted by the pass above), then of the unitigs' k-mers (k_us_ab), each on its own */
    unsigned long long sat_single = 0, sat_unitigs = 0, sat_late = 0;
    HIP_TRY(hipMemcpy(&sat_single, cnt_d + JT_C_SAT, 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(cnt_d + JT_C_SAT, 0, 8));
    unsigned long long n_words = 0, n_rec = 0, stored_views = 0;
    if (n_starts) {
        const unsigned long long rec_cap = n_starts / 2 + 1; /* two starts per stored unitig (one per strand) */
        HIP_TRY(d_rec.alloc(rec_cap * sizeof(UsRec)));
        prof.begin();
        hipLaunchKernelGGL(k_jt_plan, dim3((unsigned)((n_starts + 63) / 64)), dim3(64), 0, 0, jt, k, d_starts.as<uint64_t>(), n_starts, cnt_d, d_rec.as<UsRec>(), rec_cap);
        HIP_TRY(prof.end("jt_plan", (interior + n_starts) * 32, interior + n_starts)); /* both strands of every chain: one bucket per step */
        HIP_TRY(hipMemcpy(cnt, d_cnt.p, sizeof cnt, hipMemcpyDeviceToHost));
        n_words = cnt[JT_C_WORDS]; n_rec = cnt[JT_C_RECS]; stored_views = cnt[JT_C_STORED_VIEWS];
        if (n_rec > rec_cap) { set_error("unitig construction: %llu records for %llu chain starts", n_rec, n_starts); return MTG_ERR_OVERFLOW; }
    }
    (void)d_starts.alloc(0);
    if (n_rec) {
        const unsigned long long pad = 8; /* the coverage pass may look up to 64 + k nucleotides past the end of a unitig */
        HIP_TRY(timed_malloc((void**)&idx->dev.us.words, (n_words + pad) * 8));
        HIP_TRY(timed_malloc((void**)&idx->dev.us.ab, (n_words + pad) * 32));
        prof.begin();
        HIP_TRY(hipMemsetAsync(idx->dev.us.words, 0, (n_words + pad) * 8, 0));
        HIP_TRY(hipMemsetAsync(idx->dev.us.ab, 0, (n_words + pad) * 32, 0));
        idx->dev.us.nwords = n_words;
        idx->dev.us.nunitigs = n_rec;
        hipLaunchKernelGGL(k_jt_emit, dim3((unsigned)((n_rec + 63) / 64)), dim3(64), 0, 0, jt, idx->dev.us, k, d_rec.as<UsRec>(), n_rec);
        HIP_TRY(prof.end("jt_emit", (stored_views / 2) * 32 + n_words * 48, stored_views / 2 + n_rec));
        prof.begin();
        hipLaunchKernelGGL(k_us_ab<Src>, dim3((unsigned)std::min<unsigned long long>((n_rec + 3) / 4, 256 * 64)), dim3(256), 0, 0, idx->dev.us, k, d_rec.as<UsRec>(), n_rec, src, cnt_d);
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

/* MTG_DENSE_INDEX=1 / MTG_NO_UNITIGS=1 (test hooks: the dense form of the index, the index without a unitig store) and MTG_LEGACY_BUILD=1
 * (A/B) take the construction of rounds 1-3: dense ADJ + ABND tables, lookaheads, the store from them, then the sparse form */
static bool legacy_build() { return tune::on(tune::T_DENSE_INDEX) || tune::on(tune::T_NO_UNITIGS) || tune::on(tune::T_LEGACY_BUILD); }
static double jt_load() { return tune::f(tune::T_JT_LOAD, 0.7); }
/* a cleared table of MTG_ABND_SLOTS-slot buckets for nkeys keys of key_bits bits */
/* bytes the sparse ADJ table of a graph of n k-mers will take, give or take: half the junctions and a little (sparsify) */
static size_t adj_bytes_estimate(uint64_t n, int k)
{
    return (size_t)buckets_for(n / 2 + n / 128 + 8192, 0.49, 2 * (k - 1), MTG_ADJ_SLOTS) * 16 * MTG_ADJ_SLOTS;
}
/* min_bytes: the buffer is made at least this large (the junction table's memory is handed on to the sparse ADJ table) */
static int alloc_slot_table(Table& t, DevBuf& buf, uint64_t nkeys, double load, uint32_t key_bits, BuildProf& prof, const char* phase, size_t min_bytes = 0)
{
    table_shape(t, buckets_for(nkeys, load, key_bits, MTG_ABND_SLOTS), key_bits);
    t.sp_words = nullptr;
    const size_t bytes = t.nbuckets * 8 * MTG_ABND_SLOTS;
    if (!(buf.p && buf.cap >= std::max(bytes, min_bytes))) HIP_TRY(buf.alloc(std::max(bytes, min_bytes)));
    t.slots = buf.as<uint64_t>();
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
        if (int rc2 = alloc_slot_table(jt, jt_buf, n + n / 8 + 1024, jt_load() * load, 2 * (k - 1), prof, "clear_jt", adj_bytes_estimate(n, k))) return rc2;
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

static int index_from_packed_device_lean(const uint64_t* d_words, const uint64_t* d_word_off, const uint32_t* d_len, size_t nseq, uint64_t total_kmers_ub, int k,
                                         uint32_t abund_lo, uint32_t abund_span, mtg_index** out)
{
    BuildProf prof;
    IndexGuard g(new mtg_index());
    mtg_index* idx = g.idx;
    idx->dev.k = k;
    HIP_TRY(hipGetDevice(&idx->device));
    DevBuf d_cnt, jt_buf;
    HIP_TRY(d_cnt.alloc(32));
    Table jt{};
    double load = 1.0;
    int rc = MTG_OK;
    const uint64_t n_junctions_ub = total_kmers_ub + nseq + 1024; /* a sequence of L >= k nucleotides has L - k + 2 junction positions */
    for (int attempt = 0; attempt < 6; attempt++) {
        if (int rc2 = alloc_slot_table(jt, jt_buf, n_junctions_ub, jt_load() * load, 2 * (k - 1), prof, "clear_jt", adj_bytes_estimate(total_kmers_ub, k))) return rc2;
        HIP_TRY(hipMemset(d_cnt.p, 0, 32));
        if (nseq == 0) break; /* an empty graph: nothing to launch */
        prof.begin();
        hipLaunchKernelGGL(k_jt_insert_packed, dim3((unsigned)std::min<size_t>(nseq, 256 * 32)), dim3(256), 0, 0, jt, k, d_words, d_word_off, d_len, nseq, d_cnt.as<unsigned long long>());
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
    if (int rc2 = build_from_jt(idx, jt_buf, jt, src, [] {}, 0, prof)) return rc2;
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
    if (!legacy_build()) return index_from_kmer_pieces_lean(n, k, fetch, out);
    IndexGuard g(new mtg_index());
    mtg_index* idx = g.idx;
    idx->dev.k = k;
    HIP_TRY(hipGetDevice(&idx->device));
    DevBuf d_k, d_a, d_cnt;
    HIP_TRY(d_cnt.alloc(4 * 8));
    const size_t env_piece = (size_t)tune::i(tune::T_LOAD_PIECE, 0); /* test hook: small pieces */
    const size_t piece = std::min<size_t>(std::max<size_t>(n, 1), env_piece ? env_piece : (size_t)1 << 26);
    HIP_TRY(d_k.alloc(piece * 8));
    HIP_TRY(d_a.alloc(piece * 4));
    double load = 1.0; /* scale of the default load factors; lowered when an insertion overflows its displacement range */
    int rc = MTG_OK;
    for (int attempt = 0; attempt < 6; attempt++) {
        free_tables(idx);
        rc = alloc_tables(idx, n, load);
        if (rc) return rc;
        HIP_TRY(hipMemset(d_cnt.p, 0, 32));
        for (size_t off = 0; off < n; off += piece) {
            const size_t m = std::min(piece, n - off);
            const uint64_t* hk = nullptr;
            const uint32_t* ha = nullptr;
            if (!fetch(off, m, hk, ha)) return MTG_ERR_IO; /* the source has set the message */
            HIP_TRY(hipMemcpy(d_k.p, hk, m * 8, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(d_a.p, ha, m * 4, hipMemcpyHostToDevice));
            const int blocks = (int)std::min<size_t>((m + 255) / 256 + 1, 256 * 16);
            hipLaunchKernelGGL(k_insert_kmers, dim3(blocks), dim3(256), 0, 0, idx->dev, d_k.as<uint64_t>(), d_a.as<uint32_t>(), m, d_cnt.as<unsigned long long>());
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipDeviceSynchronize()); /* the source may reuse its buffers for the next piece */
        }
        unsigned long long cnt[4];
        HIP_TRY(hipMemcpy(cnt, d_cnt.p, 32, hipMemcpyDeviceToHost));
        idx->info.nb_saturated = cnt[3];
        if (!cnt[0]) {
            if (n) {
                const uint64_t nslots = idx->dev.abnd.nbuckets * MTG_ABND_SLOTS;
                hipLaunchKernelGGL(k_lookahead_table, dim3((unsigned)std::min<uint64_t>((nslots + 255) / 256, 256 * 32)), dim3(256), 0, 0, idx->dev);
                HIP_TRY(hipGetLastError());
            }
            HIP_TRY(hipDeviceSynchronize());
            rc = MTG_OK;
            break;
        }
        load *= 0.7;
        rc = MTG_ERR_OVERFLOW;
        set_error("index bucket displacement overflow");
    }
    if (rc) return rc;
    if (int rc2 = build_unitigs(idx)) return rc2;
    idx->info.k = k;
    idx->info.abundance_min = 0;
    idx->info.abundance_auto = -1;
    *out = g.release();
    return MTG_OK;
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
    if (!legacy_build()) return index_from_packed_device_lean(d_words, d_word_off, d_len, nseq, total_kmers_ub, k, abund_lo, abund_span, out);
    IndexGuard g(new mtg_index());
    mtg_index* idx = g.idx;
    idx->dev.k = k;
    HIP_TRY(hipGetDevice(&idx->device));
    DevBuf d_cnt;
    HIP_TRY(d_cnt.alloc(32));
    double load = 1.0; /* scale of the default load factors; lowered when an insertion overflows its displacement range */
    int rc = MTG_OK;
    for (int attempt = 0; attempt < 6; attempt++) {
        free_tables(idx);
        rc = alloc_tables(idx, total_kmers_ub, load);
        if (rc) return rc;
        HIP_TRY(hipMemset(d_cnt.p, 0, 32));
        if (nseq == 0) break; /* an empty graph: nothing to launch (a grid of zero blocks is an error that would surface later) */
        const int blocks = (int)std::min<size_t>(nseq, 256 * 32);
        hipLaunchKernelGGL(k_insert_packed, dim3(blocks), dim3(256), 0, 0, idx->dev, d_words, d_word_off, d_len, nseq, abund_lo, abund_span, d_cnt.as<unsigned long long>());
        HIP_TRY(hipGetLastError());
        unsigned long long cnt[4];
        HIP_TRY(hipMemcpy(cnt, d_cnt.p, 32, hipMemcpyDeviceToHost));
        if (!cnt[0]) {
            hipLaunchKernelGGL(k_lookahead_packed, dim3(blocks), dim3(256), 0, 0, idx->dev, d_words, d_word_off, d_len, nseq);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipDeviceSynchronize());
            rc = MTG_OK;
            break;
        }
        load *= 0.7;
        rc = MTG_ERR_OVERFLOW;
        set_error("index bucket displacement overflow");
    }
    if (rc) return rc;
    if (int rc2 = build_unitigs(idx)) return rc2;
    idx->info.k = k;
    idx->info.abundance_min = (int)abund_lo;
    idx->info.abundance_auto = -1;
    *out = g.release();
    return MTG_OK;
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
    auto clone = [&](void** dst, const void* from, size_t bytes) -> int {
        if (!from || !bytes) return MTG_OK;
        HIP_TRY(hipMalloc(dst, bytes));
        HIP_TRY(hipMemcpyPeer(*dst, device, from, src->device, bytes));
        return MTG_OK;
    };
    const size_t ba = src->dev.adj.nbuckets * 16 * MTG_ADJ_SLOTS, bb = src->dev.abnd.nbuckets * 8 * MTG_ABND_SLOTS, bc = src->dev.bloom.nblocks * 64;
    if (int rc = clone((void**)&idx->dev.adj.slots, src->dev.adj.slots, ba)) return rc;
    if (int rc = clone((void**)&idx->dev.abnd.slots, src->dev.abnd.slots, bb)) return rc;
    if (int rc = clone((void**)&idx->dev.bloom.bits, src->dev.bloom.bits, bc)) return rc;
    if (src->dev.us.nwords) {
        const size_t nw = src->dev.us.nwords + 8;
        if (int rc = clone((void**)&idx->dev.us.words, src->dev.us.words, nw * 8)) return rc;
        if (int rc = clone((void**)&idx->dev.us.ab, src->dev.us.ab, nw * 32)) return rc;
    }
    if (src->dev.adj.sp_words) idx->dev.adj.sp_words = idx->dev.us.words; /* the sparse form reads this copy's store */
    HIP_TRY(hipDeviceSynchronize());
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

/* an index out of its container: the store goes up as it is, the tables are derived from it (sparsify); nothing is counted or walked */
int index_from_dump(const IndexDump& d, mtg_index** out)
{
    if (int rc = ensure_device()) return rc;
    if (d.k < 11 || d.k > 31 || !out) { set_error("invalid argument (11 <= k <= 31)"); return MTG_ERR_ARG; }
    if (tune::on(tune::T_DENSE_INDEX) || d.n_words == 0) {
        /* test hook / an index without stored unitigs: through the list of its k-mers */
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
    if (int rc = sparsify(idx, d_rec.as<UsRec>(), hdr.size(), true, d_k.as<uint64_t>(), d_a.as<uint32_t>(), d.left_k.size(), &prof)) return rc;
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
        if (w.stream) (void)hipStreamDestroy((hipStream_t)w.stream);
        if (w.copy_stream) (void)hipStreamDestroy((hipStream_t)w.copy_stream);
    }
    free_tables(idx);
    delete idx;
}

/* a device buffer of one call: the cached slot of a workspace when the call belongs to a batch (grow-only: no hipMalloc / hipFree in the
 * steady state -- hipFree waits for the whole device, i.e. for every other batch in flight), memory of its own otherwise */
struct CallBuf {
    void* p = nullptr;
    bool own = false;
    hipError_t alloc(Workspace* ws, int slot, size_t bytes)
    {
        bytes = bytes ? bytes : 8;
        if (!ws) { own = true; return hipMalloc(&p, bytes); }
        if (ws->cap[slot] < bytes) {
            if (ws->ptr[slot]) (void)hipFree(ws->ptr[slot]);
            ws->ptr[slot] = nullptr; ws->cap[slot] = 0;
            const size_t want = bytes + bytes / 4 + 4096;
            const hipError_t e = hipMalloc(&ws->ptr[slot], want);
            if (e != hipSuccess) return e;
            ws->cap[slot] = want;
        }
        p = ws->ptr[slot];
        return hipSuccess;
    }
    ~CallBuf() { if (own && p) (void)hipFree(p); }
    template <typename T> T* as() { return (T*)p; }
};
enum { CALL_SLOT0 = 23 }; /* workspace slots 0 .. 22 belong to device_run; a batch's later calls (k_query, k_nw) use 23 .. 30, one call at a time */

int query_run(const mtg_index* idx, const uint64_t* kmers, size_t n, uint32_t* abund, uint8_t* succ, uint8_t* pred, Workspace* ws)
{
    if (int rc = use_device_of(idx)) return rc;
    if (!idx || (n && !kmers)) { set_error("null argument"); return MTG_ERR_ARG; }
    if (n == 0) return MTG_OK;
    hipStream_t stream = ws ? (hipStream_t)ws->stream : nullptr;
    CallBuf d_k, d_a, d_s, d_p;
    HIP_TRY(d_k.alloc(ws, CALL_SLOT0 + 0, n * 8));
    HIP_TRY(hipMemcpyAsync(d_k.p, kmers, n * 8, hipMemcpyHostToDevice, stream));
    if (abund) HIP_TRY(d_a.alloc(ws, CALL_SLOT0 + 1, n * 4));
    if (succ) HIP_TRY(d_s.alloc(ws, CALL_SLOT0 + 2, n));
    if (pred) HIP_TRY(d_p.alloc(ws, CALL_SLOT0 + 3, n));
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(k_query, dim3(blocks), dim3(256), 0, stream, idx->dev, d_k.as<uint64_t>(), n, abund ? d_a.as<uint32_t>() : nullptr, succ ? d_s.as<uint8_t>() : nullptr,
                       pred ? d_p.as<uint8_t>() : nullptr);
    HIP_TRY(hipGetLastError());
    if (abund) HIP_TRY(hipMemcpyAsync(abund, d_a.p, n * 4, hipMemcpyDeviceToHost, stream));
    if (succ) HIP_TRY(hipMemcpyAsync(succ, d_s.p, n, hipMemcpyDeviceToHost, stream));
    if (pred) HIP_TRY(hipMemcpyAsync(pred, d_p.p, n, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return MTG_OK;
}

/* ------------------------------------------------------------------------------------------------ fill batches */
namespace {
/* a view on a cached, grow-only workspace buffer of the index (no hipMalloc / hipFree on the steady-state path) */
struct WsBuf {
    void* p = nullptr;
    mtgi::Workspace* ws = nullptr;
    int slot = -1;
    bool fresh = false; /* the last alloc() had to get new memory (contents undefined) */
    hipError_t alloc(size_t bytes)
    {
        bytes = bytes ? bytes : 8;
        fresh = false;
        if (ws->cap[slot] < bytes) {
            fresh = true;
            if (ws->ptr[slot]) (void)hipFree(ws->ptr[slot]);
            ws->ptr[slot] = nullptr;
            ws->cap[slot] = 0;
            const size_t want = bytes + bytes / 8;
            hipError_t e = hipMalloc(&ws->ptr[slot], want);
            if (e != hipSuccess) { e = hipMalloc(&ws->ptr[slot], bytes); if (e != hipSuccess) return e; ws->cap[slot] = bytes; }
            else ws->cap[slot] = want;
        }
        p = ws->ptr[slot];
        return hipSuccess;
    }
    /* the same, but the first `keep` bytes survive a move (the stream must be idle: the copy runs on the null stream) */
    hipError_t grow_keeping(size_t bytes, size_t keep)
    {
        if (ws->cap[slot] >= bytes || !ws->ptr[slot] || keep == 0) return alloc(bytes);
        void* old = ws->ptr[slot];
        const size_t old_cap = ws->cap[slot];
        ws->ptr[slot] = nullptr;
        ws->cap[slot] = 0;
        hipError_t e = alloc(bytes);
        if (e == hipSuccess) e = hipMemcpy(ws->ptr[slot], old, std::min(keep, old_cap), hipMemcpyDeviceToDevice);
        (void)hipFree(old);
        return e;
    }
    template <typename T> T* as() { return (T*)p; }
};
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <typename B, typename T> hipError_t upload(B& b, const std::vector<T>& v)
{
    hipError_t e = b.alloc(v.size() * sizeof(T));
    if (e != hipSuccess) return e;
    return v.empty() ? hipSuccess : hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
}
} // namespace

void* staging_host(Workspace* wsp, int slot, size_t bytes)
{
    if (!wsp || slot < 0 || slot >= Workspace::NHOST) return nullptr;
    Workspace& ws = *wsp;
    if (ws.hcap[slot] < bytes) {
        if (ws.hptr[slot]) (void)hipHostFree(ws.hptr[slot]);
        ws.hptr[slot] = nullptr;
        ws.hcap[slot] = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        if (hipHostMalloc(&ws.hptr[slot], want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); ws.hptr[slot] = nullptr; return nullptr; }
        ws.hcap[slot] = want;
    }
    return ws.hptr[slot];
}


void* pinned_alloc(size_t bytes)
{
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 8, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void pinned_free(void* p) { if (p) (void)hipHostFree(p); }
int host_register(void* p, size_t bytes)
{
    if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); set_error("cannot page-lock %zu bytes at %p", bytes, p); return MTG_ERR_ARG; }
    return MTG_OK;
}
int host_unregister(void* p) { if (hipHostUnregister(p) != hipSuccess) { (void)hipGetLastError(); set_error("%p is not page-locked", p); return MTG_ERR_ARG; } return MTG_OK; }
int device_download(const mtg_index* idx, void* host_dst, const void* dev_src, size_t bytes)
{
    if (int rc = use_device_of(idx)) return rc;
    if (bytes) HIP_TRY(hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost));
    return MTG_OK;
}
int device_upload(const mtg_index* idx, void* dev_dst, const void* host_src, size_t bytes)
{
    if (int rc = use_device_of(idx)) return rc;
    if (bytes) HIP_TRY(hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice));
    return MTG_OK;
}

/* device copies of a marshalled batch: blocks A and B and the encoded targets (block C only serves to make those) */
int batch_upload(const mtg_index* idx, FillInput& in)
{
    if (int rc = use_device_of(idx)) return rc;
    batch_release_device(in);
    DevBuf a, b, c, t;
    const uint64_t n_targets = in.traw.size() / TARGET_SLOT;
    HIP_TRY(a.alloc(in.bytes_a));
    HIP_TRY(b.alloc(in.bytes_b));
    HIP_TRY(c.alloc(in.bytes_c));
    HIP_TRY(t.alloc(n_targets * 16 + 64));
    HIP_TRY(hipMemcpy(a.p, in.block_a, in.bytes_a, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b.p, in.block_b, in.bytes_b, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c.p, in.block_c, in.bytes_c, hipMemcpyHostToDevice));
    if (n_targets) {
        hipLaunchKernelGGL(k_encode_targets, dim3((unsigned)((n_targets + 255) / 256)), dim3(256), 0, 0, c.as<uint8_t>(), t.as<uint64_t>(), t.as<uint64_t>() + n_targets, n_targets, in.k);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipDeviceSynchronize());
    in.dev_a = a.release();
    in.dev_b = b.release();
    in.dev_tenc = t.release();
    return MTG_OK;
}
void batch_release_device(FillInput& in)
{
    if (in.dev_a) (void)hipFree(in.dev_a);
    if (in.dev_b) (void)hipFree(in.dev_b);
    if (in.dev_tenc) (void)hipFree(in.dev_tenc);
    in.dev_a = in.dev_b = in.dev_tenc = nullptr;
}

namespace {
struct EventSet { /* events of one device_run call */
    std::vector<hipEvent_t> ev;
    ~EventSet() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); }
    /* blocking: a caller waiting for the device sleeps instead of spinning, its CPU time belongs to the worker pool (and, in a container,
     * to the CPU quota the pool lives on) */
    hipError_t make(hipEvent_t& e) { hipError_t r = hipEventCreateWithFlags(&e, hipEventBlockingSync); if (r == hipSuccess) ev.push_back(e); return r; }
};
} // namespace

/* The caller holds the lock of in.ws (a workspace and its staging blocks belong to one batch at a time); everything the batch queues
 * goes to the workspace's own stream, so that two batches on the device overlap.
 *
 * One launch = as many gaps as fit the scratch: traversal (k_stage_a), terminal search + coverage (k_post), layout of the results
 * (k_scan1, k_scan2), results (k_emit); then the totals come back, and with them the sizes of the copies that bring the records and the
 * sequences to the host arrays of `sink`.  The host looks at a gap only if it has to be re-run in a larger scratch tier or takes the
 * multi-contig path (`special`). */
/* at most MTG_COPY_SLOTS (default 3, 0 = no limit) batches of one device copy their results to the host at the same time (every device has
 * its own link: the tool's host threads, one per device, do not wait for each other) */
struct CopyTurn {
    enum { MAX_DEV = 64 };
    struct State { std::mutex m; std::condition_variable c; int busy = 0; };
    static State& state(int dev) { static State st[MAX_DEV]; return st[(unsigned)dev % MAX_DEV]; }
    static int slots() { return (int)tune::i(tune::T_COPY_SLOTS, 3); }
    State* held = nullptr;
    explicit CopyTurn(int dev)
    {
        if (slots() <= 0) return;
        State& st = state(dev);
        std::unique_lock<std::mutex> lk(st.m);
        st.c.wait(lk, [&] { return st.busy < slots(); });
        st.busy++;
        held = &st;
    }
    void release()
    {
        if (!held) return;
        { std::lock_guard<std::mutex> lk(held->m); held->busy--; }
        held->c.notify_one();
        held = nullptr;
    }
    ~CopyTurn() { release(); }
};

/* The inputs of all batches of a device go up on ONE stream (the batch's own stream waits for its event).  Measured on this box
 * (scripts/pcie_duplex.py): one host-to-device copy at a time runs next to two or three device-to-host copies at the full rate of both
 * directions (40 MB down + 16 MB up: 0.77 ms, the 40 MB alone 0.74); three uploads at a time take the link from the downloads (1.17 ms) --
 * which is what six batches in flight did to the text entry, whose upload is a third of its download.  MTG_UPLOAD_OWN_STREAM=1: as before. */
static hipStream_t upload_stream_of(int dev)
{
    static std::mutex m;
    static hipStream_t s[CopyTurn::MAX_DEV] = {};
    if (tune::on(tune::T_UPLOAD_OWN_STREAM)) return nullptr;
    std::lock_guard<std::mutex> lk(m);
    hipStream_t& r = s[(unsigned)dev % CopyTurn::MAX_DEV];
    if (!r && hipStreamCreateWithFlags(&r, hipStreamNonBlocking) != hipSuccess) r = nullptr;
    return r;
}

int device_run(const mtg_index* idx, const mtg_params* p, const FillInput& in, ResultSink& sink, DevBatch& special, mtg_batch_stats* stats, const std::function<void()>* while_busy)
{
    bool busy_done = false;
    const bool dbg = tune::on(tune::T_DEBUG_TIMERS);
    double tk = now_ms();
    auto tick = [&](const char* what) { if (dbg) { double t = now_ms(); fprintf(stderr, "  [device_run] %-18s %.2f ms\n", what, t - tk); tk = t; } };
    if (int rc = use_device_of(idx)) return rc;
    if (!in.ws) { set_error("device_run: the input has no workspace"); return MTG_ERR_ARG; }
    Workspace& ws = *in.ws;
    if (!ws.stream) {
        hipStream_t s0, s1;
        HIP_TRY(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
        ws.stream = (void*)s0;
        HIP_TRY(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
        ws.copy_stream = (void*)s1;
        HIP_TRY(hipDeviceSynchronize()); /* the index was built or loaded on the null stream, which these streams do not wait for */
    }
    const hipStream_t stream = (hipStream_t)ws.stream;
    const size_t n = in.src.size();
    special.chunks.clear();
    special.special.clear();
    sink.seq_used = 0;
    sink.ext_used = 1;
    sink.n_filled = 0;
    sink.in_gap_order = true;
    mtg_batch_stats st = stats ? *stats : mtg_batch_stats{};
    if (n == 0) { if (while_busy) (*while_busy)(); if (stats) *stats = st; return MTG_OK; }
    const int k = idx->dev.k;

    int ws_next = 0;
    auto wsbuf = [&]() { WsBuf b; b.ws = &ws; b.slot = ws_next++; return b; };
    WsBuf d_ina = wsbuf(), d_inb = wsbuf(), d_inc = wsbuf(), d_tenc = wsbuf(), d_ilv = wsbuf(), d_zero = wsbuf(), d_raw = wsbuf(), d_out = wsbuf(), d_rec = wsbuf(), d_ids = wsbuf(), d_dw = wsbuf(),
          d_dm = wsbuf(), d_cnt = wsbuf(), d_blocks = wsbuf(), d_seq = wsbuf(), d_ext = wsbuf(), d_res = wsbuf(), d_fil = wsbuf(), d_tot = wsbuf(), d_rlist = wsbuf(), d_glist = wsbuf(), d_paths = wsbuf(), d_park = wsbuf(), d_out2 = wsbuf();
    /* the marshalled input: three blocks, three copies; the targets (block C, text) become k-mers and masks on the device.  A batch that
     * was prepared ahead (mtg_batch) is resident already */
    double t0 = now_ms();
    const uint64_t n_targets = in.text_mode ? in.n_text_targets : in.traw.size() / TARGET_SLOT;
    EventSet events;
    hipStream_t up = (in.text_mode || !in.dev_a) ? upload_stream_of(idx->device) : nullptr;
    if (!up) up = stream;
    auto uploaded = [&]() -> int { /* the batch's stream goes on when its blocks have arrived */
        if (up == stream) return MTG_OK;
        hipEvent_t evu;
        HIP_TRY(events.make(evu));
        HIP_TRY(hipEventRecord(evu, up));
        HIP_TRY(hipStreamWaitEvent(stream, evu, 0));
        return MTG_OK;
    };
    const uint8_t* da;
    const uint64_t* d_rw;
    uint64_t* d_tle;
    if (in.text_mode) {
        /* the strings are still text: block A (integer columns) and the text block go up, the device encodes (mtg_marshal.h) */
        HIP_TRY(d_ina.alloc(in.bytes_a));
        HIP_TRY(d_inb.alloc(in.bytes_b));
        HIP_TRY(d_inc.alloc(in.bytes_c));
        HIP_TRY(hipMemcpyAsync(d_ina.p, in.block_a, in.bytes_a, hipMemcpyHostToDevice, up));
        if (in.text_direct) { /* the block from the caller's page-locked memory, the offset arrays from the staging block */
            const size_t off5 = FillInput::text_block_off(n, (size_t)n_targets, 5);
            HIP_TRY(hipMemcpyAsync(d_inc.p, in.block_c, off5, hipMemcpyHostToDevice, up));
            HIP_TRY(hipMemcpyAsync((uint8_t*)d_inc.p + off5, in.text_direct, in.text_bytes, hipMemcpyHostToDevice, up));
        } else HIP_TRY(hipMemcpyAsync(d_inc.p, in.block_c, in.bytes_c, hipMemcpyHostToDevice, up));
        if (int rc = uploaded()) return rc;
        HIP_TRY(d_tenc.alloc(n_targets * 16 + 64));
        uint8_t* a = d_ina.as<uint8_t>();
        const uint8_t* c = d_inc.as<uint8_t>();
        const size_t nn = n, nt = (size_t)n_targets;
        hipLaunchKernelGGL(k_marshal_text, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, c + FillInput::text_block_off(nn, nt, 5), (const uint64_t*)(c + FillInput::text_block_off(nn, nt, 0)),
                           (const uint32_t*)(c + FillInput::text_block_off(nn, nt, 3)), (const uint64_t*)(c + FillInput::text_block_off(nn, nt, 1)), (const uint32_t*)(a + FillInput::off_a(n, 2)),
                           (uint32_t*)(a + FillInput::off_a(n, 3)), (uint64_t*)(a + FillInput::off_a(n, 0)), (uint64_t*)(a + FillInput::off_a(n, 1)), a + FillInput::off_a(n, 7), d_inb.as<uint64_t>(),
                           (uint32_t)n, k);
        da = a;
        d_rw = d_inb.as<uint64_t>();
        d_tle = d_tenc.as<uint64_t>();
        if (n_targets) hipLaunchKernelGGL(k_marshal_targets, dim3((unsigned)((n_targets + 255) / 256)), dim3(256), 0, stream, c + FillInput::text_block_off(nn, nt, 5),
                                          (const uint64_t*)(c + FillInput::text_block_off(nn, nt, 2)), (const uint32_t*)(c + FillInput::text_block_off(nn, nt, 4)), d_tle, d_tle + n_targets, (uint32_t)n_targets, k);
        HIP_TRY(hipGetLastError());
    } else if (in.dev_a) {
        da = (const uint8_t*)in.dev_a;
        d_rw = (const uint64_t*)in.dev_b;
        d_tle = (uint64_t*)in.dev_tenc;
    } else {
        HIP_TRY(d_ina.alloc(in.bytes_a));
        HIP_TRY(d_inb.alloc(in.bytes_b));
        HIP_TRY(d_inc.alloc(in.bytes_c));
        HIP_TRY(hipMemcpyAsync(d_ina.p, in.block_a, in.bytes_a, hipMemcpyHostToDevice, up));
        HIP_TRY(hipMemcpyAsync(d_inb.p, in.block_b, in.bytes_b, hipMemcpyHostToDevice, up));
        HIP_TRY(hipMemcpyAsync(d_inc.p, in.block_c, in.bytes_c, hipMemcpyHostToDevice, up));
        if (int rc = uploaded()) return rc;
        HIP_TRY(d_tenc.alloc(n_targets * 16 + 64));
        da = d_ina.as<uint8_t>();
        d_rw = d_inb.as<uint64_t>();
        d_tle = d_tenc.as<uint64_t>();
        if (n_targets) hipLaunchKernelGGL(k_encode_targets, dim3((unsigned)((n_targets + 255) / 256)), dim3(256), 0, stream, d_inc.as<uint8_t>(), d_tle, d_tle + n_targets, n_targets, k);
    }
    const uint64_t* d_src = (const uint64_t*)(da + FillInput::off_a(n, 0));
    const uint64_t* d_r0 = (const uint64_t*)(da + FillInput::off_a(n, 1));
    const uint32_t* d_roff = (const uint32_t*)(da + FillInput::off_a(n, 2));
    const uint32_t* d_rlen = (const uint32_t*)(da + FillInput::off_a(n, 3));
    const uint32_t* d_toff = (const uint32_t*)(da + FillInput::off_a(n, 4));
    const uint32_t* d_tcnt = (const uint32_t*)(da + FillInput::off_a(n, 5));
    const uint8_t* d_mis = da + FillInput::off_a(n, 6);
    const uint8_t* d_fok = da + FillInput::off_a(n, 7);
    const uint8_t* d_flags = da + FillInput::off_a(n, 8);
    uint64_t* d_tbad = d_tle + n_targets;
    HIP_TRY(d_cnt.alloc(64));
    HIP_TRY(d_tot.alloc(sizeof(PartTot)));
    {
        const unsigned long long init[8] = {0, 0, 0, 1 /* the extension arena starts with the empty string */, 0, 0, 0, 0};
        HIP_TRY(hipMemcpyAsync(d_cnt.p, init, 64, hipMemcpyHostToDevice, stream)); /* pageable source: copied before the call returns */
    }
    st.h2d_ms += now_ms() - t0;
    tick("upload (async)");

    hipEvent_t ev0, ev1, ev2, ev3, eve, evc, evf, evl, evl0;
    HIP_TRY(events.make(eve));
    HIP_TRY(events.make(evc));
    HIP_TRY(events.make(ev0));
    HIP_TRY(events.make(ev1));
    HIP_TRY(events.make(ev2));
    HIP_TRY(events.make(ev3));
    HIP_TRY(events.make(evf));
    HIP_TRY(events.make(evl));
    HIP_TRY(events.make(evl0));
    PartTot* h_tot = (PartTot*)staging_host(&ws, Workspace::NHOST - 1, sizeof(PartTot) + 64);
    if (!h_tot) { set_error("no page-locked memory for the totals of a launch"); return MTG_ERR_NOMEM; }

    std::vector<uint32_t> todo; /* empty at tier 0: every gap, in order */
    size_t n_todo = n;
    int rc = MTG_OK;
    const bool host_paths = tune::on(tune::T_HOST_PATHS); /* test hook: leave the path enumeration to the host */
    const bool want_records = sink.res != nullptr;
    size_t launches = 0;

    for (int tier = 0; tier <= MTG_MAX_TIER && n_todo; tier++) {
        FillCfg cfg = make_cfg(k, p->max_nodes, p->max_depth, p->end_rule_nonbranching, tier);
        const bool no_defer = tune::on(tune::T_NO_DEFER); /* test hook: the lanes of the traversal copy their long runs themselves */
        if (no_defer || !idx->dev.us.nwords) cfg.cmd_cap = 0;
        /* scratch of a gap + worst-case room in the dense arrays (its whole contig arena and the metadata of every contig) */
        const uint64_t per_gap = cfg.zero_stride + cfg.raw_stride + cfg.ilv_stride / 64 + sizeof(GapOut) + sizeof(SlotRec) + 64 + sizeof(mtg_gap_result) + sizeof(mtg_filled);
        const size_t cached = ws.cap[d_zero.slot] + ws.cap[d_raw.slot] + ws.cap[d_ilv.slot]; /* already ours */
        size_t free_b = 0, total_b = 0;
        /* steady state: every scratch buffer of the workspace already holds a batch of this size at this tier, nothing will be allocated */
        const uint64_t m0 = std::min<uint64_t>(n_todo, 1u << 20);
        const bool fits = ws.cap[d_zero.slot] >= m0 * cfg.zero_stride && ws.cap[d_raw.slot] >= m0 * cfg.raw_stride + 64 && ws.cap[d_ilv.slot] >= ((m0 + 63) / 64) * cfg.ilv_stride;
        if (!fits) HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        size_t chunk = fits ? (size_t)m0 : (size_t)(((double)free_b * 0.6 + (double)cached) / (double)per_gap);
        const size_t env_chunk = (size_t)tune::i(tune::T_MAX_CHUNK, 0); /* test hook: several launches per batch */
        if (env_chunk && chunk > env_chunk) chunk = env_chunk;
        if (chunk > n_todo) chunk = n_todo;
        if (chunk > (1u << 20)) chunk = 1u << 20;
        if (chunk == 0) { set_error("not enough device memory for one gap at scratch tier %d (%llu bytes)", tier, (unsigned long long)per_gap); rc = MTG_ERR_NOMEM; break; }
        HIP_TRY(d_zero.alloc(chunk * cfg.zero_stride));
        /* zeroed once: every gap restores what it touched (stage_a_gap), so the region stays clean from launch to launch */
        if (d_zero.fresh) HIP_TRY(hipMemsetAsync(d_zero.p, 0, ws.cap[d_zero.slot], stream));
        HIP_TRY(d_raw.alloc(chunk * cfg.raw_stride + 64));
        HIP_TRY(d_ilv.alloc(((chunk + 63) / 64) * cfg.ilv_stride));
        HIP_TRY(d_out.alloc(chunk * sizeof(GapOut)));
        HIP_TRY(d_out2.alloc(chunk * sizeof(GapOut))); /* where the finishing kernel leaves its records while the other gaps are post-processed */
        HIP_TRY(d_rec.alloc(chunk * sizeof(SlotRec)));
        HIP_TRY(d_ids.alloc(chunk * 4));
        HIP_TRY(d_blocks.alloc(((chunk + SCAN_SL - 1) / SCAN_SL + 1) * sizeof(ScanBlock)));
        HIP_TRY(d_res.alloc(chunk * sizeof(mtg_gap_result)));
        HIP_TRY(d_fil.alloc(chunk * sizeof(mtg_filled)));
        HIP_TRY(d_rlist.alloc(chunk * 4));
        HIP_TRY(d_glist.alloc(chunk * 4));
        HIP_TRY(d_park.alloc((size_t)chunk * 4 * PARK_LISTS + sizeof(ParkCtl)));
        HIP_TRY(d_seq.grow_keeping(sink.seq_dev ? 64 : std::max<size_t>(sink.seq_cap, 64), sink.seq_dev ? 0 : (size_t)sink.seq_used)); /* a caller's device buffer is written in place; what an earlier tier left stays */
        HIP_TRY(d_ext.alloc(std::max<size_t>(sink.ext_cap, 64)));
        /* dense words / metadata: only the gaps that need the host bring their contigs back; sized by the last need, grown on demand below */
        HIP_TRY(d_dw.alloc(std::max<size_t>(ws.cap[d_dw.slot], 1 << 20)));
        HIP_TRY(d_dm.alloc(std::max<size_t>(ws.cap[d_dm.slot], 1 << 16)));
        tick("workspace alloc");
        std::vector<uint32_t> retry;
        for (size_t base = 0; base < n_todo; base += chunk) {
            const uint32_t m = (uint32_t)std::min(chunk, n_todo - base);
            const bool identity = todo.empty() && base == 0 && m == n; /* the whole batch in one launch: slot = gap */
            const uint32_t* ids = nullptr;
            std::vector<uint32_t> seq_ids;
            const uint32_t* host_ids = nullptr; /* gap of every slot of this launch (nullptr: identity) */
            if (!identity) {
                t0 = now_ms();
                if (todo.empty()) { seq_ids.resize(m); for (uint32_t s = 0; s < m; s++) seq_ids[s] = (uint32_t)(base + s); host_ids = seq_ids.data(); }
                else host_ids = todo.data() + base;
                HIP_TRY(hipMemcpyAsync(d_ids.p, host_ids, (size_t)m * 4, hipMemcpyHostToDevice, stream)); /* host_ids outlives the launch */
                ids = d_ids.as<uint32_t>();
                st.h2d_ms += now_ms() - t0;
                sink.in_gap_order = false;
            }
            launches++;
            /* The traversal reads index shape and configuration from the module's constants, of which there is one set per workspace
             * number: the batches of an index never share a set, batches of different indexes may, so a set is locked until the traversal
             * that reads it has finished (below, after the host work that runs meanwhile). */
            /* the constants exist once per device: the lock is the (device, set)'s, so that the tool's host threads -- one or more per
             * device, each on its own replica -- only ever wait for a batch of another index on their own device */
            static std::mutex traversal_mtx[CopyTurn::MAX_DEV][TRAVERSAL_SETS];
            const uint32_t cset = (uint32_t)(&ws - idx->ws);
            std::unique_lock<std::mutex> traversal_lock(traversal_mtx[(unsigned)idx->device % CopyTurn::MAX_DEV][cset]);
            HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_ix), &idx->dev, sizeof(Index), cset * sizeof(Index), hipMemcpyHostToDevice, stream));
            HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_cfg), &cfg, sizeof(FillCfg), cset * sizeof(FillCfg), hipMemcpyHostToDevice, stream));
            const bool classic_walk = tune::on(tune::T_CLASSIC_WALK); /* A/B hook: every bubble by its lane, from HBM scratch */
            /* lanes per parked gap: 1, 8, 16 or 64 (anything else, a typo included, is 16) */
            const bool finish_g_set = tune::is_set(tune::T_FINISH_G);
            const int finish_g = [] { const int v = (int)tune::i(tune::T_FINISH_G, 16); return (v == 1 || v == 8 || v == 16 || v == 64) ? v : 16; }();
            const int env_rounds = (int)tune::i(tune::T_ROUNDS, -1);
            /* Rounds: when many gaps park (bubbles all over the data), their branching nodes are answered by the bubble kernels and the walks go
             * on in the walk kernel, one gap per lane again -- walking is cheap at full width, only the bubbles need a group -- for a few
             * rounds; what is still parked then (and everything, when few gaps park) is finished by groups in k_finish.  The host does not
             * know the counts when it queues the kernels: the number of rounds follows the share of gaps the previous launch of this workspace parked
             * (a workspace without a launch yet: the latest figure of any workspace of the index). */
            /* Walk mode of the launch.  0 (always, unless forced): the walking lane answers the strict SNP pattern itself.  1 (MTG_PARK_SNP=1, an A/B
             * hook): it parks there as well and the bubble kernel of the rounds answers it with the same fast path for all parked gaps at once --
             * built because the lanes of a wave meet their SNPs at different steps, and on the heterozygous set a wave spends four times as long
             * in bubble code run by a few lanes at a time as on walking.  Measured and not used: every resumed launch of the walk kernel costs
             * 45 us however short its segments (human-het: 7 launches 0.33 ms + bubble kernels 0.5 ms against 0.40 + 0.19 ms; 99 against 103 M/s),
             * the first bubble kernel waits for the launch's hundred general bubbles anyway, and a set with indels needs more rounds than it has
             * (25 against 33 M/s); choosing between the modes from the launches' own times picked the wrong one under six batches in flight. */
            const int env_park_snp = (int)tune::i(tune::T_PARK_SNP, 0);
            const bool whole = tier == 0 && identity && m >= 4096;
            const int wmode = (env_park_snp > 0 && !classic_walk) ? 1 : 0;
            const uint32_t own_share = wmode ? ws.mode_share[1] : (ws.mode_share[0] != ~0u ? ws.mode_share[0] : ws.park_share);
            const uint32_t park_share = own_share != ~0u ? own_share : (wmode ? 65536u : idx->park_share_any.load(std::memory_order_relaxed));
            const uint32_t park_hint = (uint32_t)(((uint64_t)park_share * m) >> 16); /* gaps this launch is expected to park */
            int rounds = env_rounds >= 0 ? env_rounds : (park_share > 32768u ? 6 : 0); /* measured: with an eighth of the gaps parked the finishing kernel alone is faster, with all of them six rounds are */
            if (rounds > (PARK_LISTS - 5) / 2) rounds = (PARK_LISTS - 5) / 2;
            ParkCtl* const park = d_park.as<ParkCtl>();
            bool overlap_finish = false;
            uint32_t late_list = 0, late_grid = 1;
            HIP_TRY(hipMemsetAsync(d_park.p, 0, sizeof(ParkCtl), stream)); /* the work lists of the launch: parked gaps, gaps with commands to execute */
            HIP_TRY(hipEventRecord(ev0, stream)); /* ev0 .. evf = the walk kernel's first launch, evf .. ev1 = rounds and the finishing kernel */
            if (classic_walk) {
                hipLaunchKernelGGL(k_stage_a_classic, dim3((m + 63) / 64), dim3(64), 0, stream, d_zero.as<uint8_t>(), d_raw.as<uint8_t>(), d_ilv.as<uint8_t>(), d_src, d_rw, d_roff,
                                   d_rlen, d_r0, ids, d_out.as<GapOut>(), m, cset);
                HIP_TRY(hipEventRecord(evf, stream));
            } else {
                hipLaunchKernelGGL(k_stage_a, dim3((m + 63) / 64), dim3(64), 0, stream, d_zero.as<uint8_t>(), d_raw.as<uint8_t>(), d_ilv.as<uint8_t>(), d_src, d_rw, d_roff,
                                   d_rlen, d_r0, ids, d_out.as<GapOut>(), m, cset, park, m, -1, 0u, wmode ? 2u : 1u);
                HIP_TRY(hipEventRecord(evf, stream));
                const bool skip_finish = tune::on(tune::T_DEBUG_SKIP_FINISH); /* diagnostics: the parked gaps stay parked (and fail as overflowing gaps) */
                /* FINISH_OVERLAP of the tuning table (A/B, off).  Without rounds the finishing kernel is the latency of a few parked walks (0.06 ms for two
                 * gaps of a haploid batch, 0.22 ms for the hundred of the SNP set) on an otherwise idle device: it can run on the workspace's second
                 * stream, writing its records to an array of its own, while k_lean, k_copy and k_post_lean take all the other gaps on the batch's
                 * stream; k_late then brings the finished gaps over, and the general k_post (which k_lean has listed them for when it saw them parked)
                 * follows on that stream.  Measured (scripts/r4_streams.sh): one batch alone 0.360 -> 0.337 ms (haploid), 0.836 -> 0.811 (SNP set) --
                 * the general k_post of the finished gaps still follows the finishing kernel -- and with six batches in flight NOT faster (sequences
                 * left in HBM 346 -> 341, 105 -> 107, 123 -> 122 M/s): other batches' kernels fill the device while one batch waits, and the extra
                 * kernels and events cost what the overlap saves.  The same holds for the general k_post next to the lean one (POST_SECOND_STREAM). */
                overlap_finish = rounds == 0 && !skip_finish && ws.copy_stream && tune::on(tune::T_FINISH_OVERLAP);
                const hipStream_t fstream = overlap_finish ? (hipStream_t)ws.copy_stream : stream;
                GapOut* const fin_out = overlap_finish ? d_out2.as<GapOut>() : d_out.as<GapOut>();
                if (overlap_finish) HIP_TRY(hipStreamWaitEvent(fstream, evf, 0));
                /* the bubbles of a round by one lane each: every lane of a wave is in the bubble code at the same time, and with small bubbles that
                 * keeps more of them in flight than a group of lanes per bubble does (MTG_BUBBLE_GROUPS=1: k_bubble<G>, the LDS form, first) */
                const bool one_lane_bubbles = !tune::on(tune::T_BUBBLE_GROUPS);
                for (int r = 0; r < rounds; r++) {
                    const uint32_t lin = 2u * (uint32_t)r;
                    if (one_lane_bubbles) {
                        hipLaunchKernelGGL(k_bubble_classic, dim3((m + 63) / 64), dim3(64), 0, stream, d_zero.as<uint8_t>(), d_raw.as<uint8_t>(), d_ilv.as<uint8_t>(), cset, park, m, lin);
                        hipLaunchKernelGGL(k_stage_a, dim3((m + 63) / 64), dim3(64), 0, stream, d_zero.as<uint8_t>(), d_raw.as<uint8_t>(), d_ilv.as<uint8_t>(), d_src, d_rw, d_roff,
                                           d_rlen, d_r0, ids, d_out.as<GapOut>(), m, cset, park, m, (int)lin, lin + 2, wmode ? 2u : 1u);
                        continue;
                    }
                    switch (finish_g) {
                        case 8: hipLaunchKernelGGL(k_bubble<8>, dim3((m + 7) / 8), dim3(64), 0, stream, d_zero.as<uint8_t>(), d_raw.as<uint8_t>(), d_ilv.as<uint8_t>(), cset, park, m, lin); break;
                        case 64: hipLaunchKernelGGL(k_bubble<64>, dim3(m), dim3(64), 0, stream, d_zero.as<uint8_t>(), d_raw.as<uint8_t>(), d_ilv.as<uint8_t>(), cset, park, m, lin); break;
                        default: hipLaunchKernelGGL(k_bubble<16>, dim3((m + 3) / 4), dim3(64), 0, stream, d_zero.as<uint8_t>(), d_raw.as<uint8_t>(), d_ilv.as<uint8_t>(), cset, park, m, lin); break;
                    }
                    hipLaunchKernelGGL(k_bubble_classic, dim3((m + 63) / 64), dim3(64), 0, stream, d_zero.as<uint8_t>(), d_raw.as<uint8_t>(), d_ilv.as<uint8_t>(), cset, park, m, lin + 1);
                    hipLaunchKernelGGL(k_stage_a, dim3((m + 63) / 64), dim3(64), 0, stream, d_zero.as<uint8_t>(), d_raw.as<uint8_t>(), d_ilv.as<uint8_t>(), d_src, d_rw, d_roff,
                                       d_rlen, d_r0, ids, d_out.as<GapOut>(), m, cset, park, m, (int)lin, lin + 2, wmode ? 2u : 1u);
                }
                const uint32_t lfin = 2u * (uint32_t)rounds;
                /* how the tail is finished: a group of lanes per parked gap, bubbles from LDS.  One lane per gap (MTG_FINISH_G=1, or below
                 * MTG_FINISH_LANE_BELOW parked gaps in the workspace's previous launch) was measured and is slower at every size: 0.11 against
                 * 0.10 ms for the haploid set's 1-5 gaps, 0.83 against 0.35 for 108 (heterozygous SNPs), 1.12 against 0.56 for 12 000 (tips). */
                const int finish_lane_below = (int)tune::i(tune::T_FINISH_LANE_BELOW, 0);
                /* lanes per parked gap in the finishing kernel: a whole wave while few gaps are parked (their chains are what the kernel takes:
                 * 0.17 against 0.32 ms for the 108 gaps of the heterozygous set), 16 when there are many (12 000 on the tips set: 0.46 against 0.63) */
                const int finish_wave_below = (int)tune::i(tune::T_FINISH_WAVE_BELOW, 2048);
                const int fin_g = finish_g_set ? finish_g : (rounds == 0 && park_hint < (uint32_t)finish_wave_below ? 64 : 16);
                const bool lane_finish = (finish_g_set && finish_g == 1) || (!finish_g_set && rounds == 0 && park_hint < (uint32_t)finish_lane_below);
                /* The grid.  The host does not know how many gaps are parked when it queues the kernel, and 100 000 groups that read one
                 * scalar and leave cost 63 us (round 3: 13 % of a haploid batch's kernels, for 5 parked gaps).  So the groups take the first
                 * `fin_entries` entries of the list -- four times what the previous launch of this workspace parked, plus 256 -- and the
                 * entries beyond, if a launch parks more than that after all, are walked one gap per lane (k_finish_lane, a grid of
                 * (m - fin_entries) / 64 workgroups: slower per gap, but only for the launch that outgrew the hint; the next one follows). */
                const bool finish_full_grid = tune::on(tune::T_FINISH_FULL_GRID); /* A/B hook: one group per gap of the launch, as in round 3 */
                const uint32_t fin_entries = (lane_finish || skip_finish) ? 0u : (finish_full_grid ? m : (uint32_t)std::min<uint64_t>(m, 4ull * park_hint + 256ull));
                const uint32_t per_wg = 64u / (uint32_t)fin_g;
                const uint32_t nwg = (fin_entries + per_wg - 1) / per_wg;
                if (!skip_finish && nwg) switch (fin_g) {
                    case 8: hipLaunchKernelGGL(k_finish<8>, dim3(nwg), dim3(64), 0, fstream, d_zero.as<uint8_t>(), d_raw.as<uint8_t>(), d_ilv.as<uint8_t>(), d_rw, d_roff, d_rlen, d_r0, ids, fin_out, cset, park, m, lfin); break;
                    case 64: hipLaunchKernelGGL(k_finish<64>, dim3(nwg), dim3(64), 0, fstream, d_zero.as<uint8_t>(), d_raw.as<uint8_t>(), d_ilv.as<uint8_t>(), d_rw, d_roff, d_rlen, d_r0, ids, fin_out, cset, park, m, lfin); break;
                    default: hipLaunchKernelGGL(k_finish<16>, dim3(nwg), dim3(64), 0, fstream, d_zero.as<uint8_t>(), d_raw.as<uint8_t>(), d_ilv.as<uint8_t>(), d_rw, d_roff, d_rlen, d_r0, ids, fin_out, cset, park, m, lfin); break;
                }
                if (!skip_finish && fin_entries < m)
                    hipLaunchKernelGGL(k_finish_lane, dim3((m - fin_entries + 63) / 64), dim3(64), 0, fstream, d_zero.as<uint8_t>(), d_raw.as<uint8_t>(), d_ilv.as<uint8_t>(), d_rw, d_roff, d_rlen, d_r0, ids, fin_out, cset, park, m, lfin, fin_entries);
                late_list = lfin;
                late_grid = (uint32_t)std::min<uint64_t>((m + 63) / 64, (4ull * park_hint + 256ull + 63) / 64);
                HIP_TRY(hipMemcpyAsync((uint8_t*)h_tot + sizeof(PartTot), d_park.p, 8, hipMemcpyDeviceToHost, fstream)); /* how many were parked: statistics, and the hint for the next launch */
#ifdef MTG_BUBBLE_TIMING
                {
                    static ParkCtl hc; static int shown = 0;
                    HIP_TRY(hipStreamSynchronize(stream));
                    HIP_TRY(hipMemcpy(&hc, d_park.p, sizeof hc, hipMemcpyDeviceToHost));
                    if (shown++ < 3) {
                        fprintf(stderr, "[timing] lists:"); for (int i = 0; i < PARK_LISTS; i++) fprintf(stderr, " %u", hc.count[i]); fprintf(stderr, "\n");
                        const char* nm[4] = {"bubble lane", "bubble wave", "resumed walk lane", "resumed walk wave"};
                        const uint32_t* hh[4] = {hc.hist_lane, hc.hist_wave, hc.hist_walk_lane, hc.hist_walk_wave};
                        for (int j = 0; j < 4; j++) { fprintf(stderr, "[timing] %s, log2(10 ns ticks) bins:", nm[j]); for (int i = 0; i < 24; i++) fprintf(stderr, " %u", hh[j][i]); fprintf(stderr, "\n"); }
                    }
                }
#endif
            }
            HIP_TRY(hipEventRecord(ev1, overlap_finish ? (hipStream_t)ws.copy_stream : stream)); /* the end of the walks: of the finishing kernel, wherever it ran */
            HIP_TRY(hipEventRecord(evl0, stream));
            HIP_TRY(hipGetLastError());
            /* evl0 .. evc: the long runs of the contigs, which the traversal only noted down */
            const bool no_lean = tune::on(tune::T_NO_LEAN); /* A/B and test hook: every contig is materialised */
            hipLaunchKernelGGL(k_lean, dim3((m + 63) / 64), dim3(64), 0, stream, idx->dev, cfg, d_raw.as<uint8_t>(), d_out.as<GapOut>(), ids, d_tle, d_tbad, d_toff, d_tcnt, d_fok,
                               (in.want_all_contigs || no_lean || !cfg.cmd_cap) ? 0u : 1u, m, park, m);
            HIP_TRY(hipEventRecord(evl, stream)); /* ev1 .. evl: k_lean; evl .. evc: k_copy */
            hipLaunchKernelGGL(k_copy, dim3((m + 3) / 4), dim3(256), 0, stream, idx->dev, cfg, d_raw.as<uint8_t>(), d_out.as<GapOut>(), park, m, (uint32_t)COPY_LIST);
            HIP_TRY(hipEventRecord(evc, stream));
            const uint32_t nblocks = (m + SCAN_SL - 1) / SCAN_SL;
            /* the lean gaps eight per wave; the others (k_lean's list) a wave each: a grid of four times what the previous launch of this workspace
             * listed, plus 1024 (a workspace without a launch yet: one per gap), the kernel's loop takes the rest */
            const uint32_t general_hint = ws.post_general == ~0u ? m : (uint32_t)std::min<uint64_t>(m, 4ull * ws.post_general + 1024ull);
            /* The general form is the latency of a few long gaps (30 us for the one or two of a haploid batch), the lean form the throughput of
             * all the others: they touch different slots and CAN run next to each other (POST_SECOND_STREAM of the tuning table: 9 us shorter for
             * one batch alone, 346 against 355 M/s with six in flight -- one stream per batch is the default). */
            const hipStream_t side = (ws.copy_stream && (overlap_finish || tune::on(tune::T_POST_SECOND_STREAM))) ? (hipStream_t)ws.copy_stream : stream;
            if (overlap_finish) {
                /* the second stream has the finishing kernel in it: the finished gaps' records, their copy commands, then (below) the general k_post,
                 * which also needs what k_lean and k_copy have done for the other listed gaps on the batch's stream (evc) */
                hipLaunchKernelGGL(k_late, dim3(late_grid), dim3(64), 0, side, idx->dev, cfg, d_raw.as<uint8_t>(), d_out.as<GapOut>(), d_out2.as<GapOut>(), park, m, late_list);
                hipLaunchKernelGGL(k_copy, dim3(late_grid * 16u), dim3(256), 0, side, idx->dev, cfg, d_raw.as<uint8_t>(), d_out.as<GapOut>(), park, m, (uint32_t)COPY_LIST_LATE);
                HIP_TRY(hipStreamWaitEvent(side, evc, 0));
            } else if (side != stream) {
                hipEvent_t ev_fork;
                HIP_TRY(events.make(ev_fork));
                HIP_TRY(hipEventRecord(ev_fork, stream));
                HIP_TRY(hipStreamWaitEvent(side, ev_fork, 0));
            }
            hipLaunchKernelGGL(k_post, dim3(general_hint), dim3(64), 0, side, idx->dev, cfg, d_raw.as<uint8_t>(), d_out.as<GapOut>(), ids, d_tle, d_tbad, d_toff, d_tcnt, d_mis, d_fok,
                               in.want_all_contigs ? 1u : 0u, d_rec.as<SlotRec>(), m, park);
            hipLaunchKernelGGL(k_post_lean, dim3((m + 64 / POST_LEAN_G - 1) / (64 / POST_LEAN_G)), dim3(64), 0, stream, idx->dev, cfg, d_raw.as<uint8_t>(), d_out.as<GapOut>(), in.want_all_contigs ? 1u : 0u, d_rec.as<SlotRec>(), m);
            if (side != stream) {
                hipEvent_t ev_join;
                HIP_TRY(events.make(ev_join));
                HIP_TRY(hipEventRecord(ev_join, side));
                HIP_TRY(hipStreamWaitEvent(stream, ev_join, 0));
            }
            HIP_TRY(hipMemsetAsync(d_cnt.p, 0, 16, stream)); /* the dense arrays hold one launch at a time; the two arenas the whole batch */
            hipLaunchKernelGGL(k_scan1, dim3(nblocks), dim3(SCAN_SL), 0, stream, d_rec.as<SlotRec>(), m, d_blocks.as<ScanBlock>());
            hipLaunchKernelGGL(k_scan2, dim3(1), dim3(256), 0, stream, d_blocks.as<ScanBlock>(), nblocks, d_cnt.as<unsigned long long>(), d_tot.as<PartTot>());
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(h_tot, d_tot.p, sizeof(PartTot), hipMemcpyDeviceToHost, stream));
            EmitDev D;
            EmitHost H;
            auto emit = [&]() -> int {
                D.seq = sink.seq_dev ? sink.seq_dev : d_seq.as<char>(); D.ext = d_ext.as<char>();
                D.seq_cap = sink.seq_cap; D.ext_cap = sink.ext_cap;
                D.res = d_res.as<mtg_gap_result>(); D.fil = d_fil.as<mtg_filled>();
                D.dense_words = d_dw.as<uint64_t>(); D.dense_meta = d_dm.as<uint32_t>();
                D.dense_cap_words = ws.cap[d_dw.slot] / 8; D.dense_cap_contigs = ws.cap[d_dm.slot] / 20;
                const bool want_wire = sink.wire_dev != nullptr && identity && tier == 0;
                D.wire = want_wire ? (uint8_t*)sink.wire_dev : nullptr; D.wire_cap = sink.wire_cap; D.wire_tag = sink.wire_tag;
                D.tot = d_tot.as<PartTot>(); D.wire_gaps = m;
                if (want_wire) HIP_TRY(hipMemsetAsync(sink.wire_dev, 0, sizeof(mtg_wire_header), stream)); /* no header, no payload (k_wire_sum) */
                H.seq = sink.seq; H.ext = sink.ext; H.fil = sink.fil;
                if (!want_wire)
                    hipLaunchKernelGGL(k_emit_lean, dim3((m + 64 / EMIT_LEAN_G - 1) / (64 / EMIT_LEAN_G)), dim3(64), 0, stream, idx->dev.us, cfg, d_raw.as<uint8_t>(), d_rec.as<SlotRec>(), d_blocks.as<ScanBlock>(),
                                       ids, d_flags, k, D, H, m);
                hipLaunchKernelGGL(k_emit, dim3(want_wire ? m : general_hint), dim3(64), 0, stream, idx->dev.us, cfg, d_raw.as<uint8_t>(), d_rec.as<SlotRec>(), d_blocks.as<ScanBlock>(), ids, d_flags, k, D, H,
                                   d_rlist.as<uint32_t>(), d_glist.as<uint32_t>(), m, park, want_wire ? 0u : 1u);
                if (want_wire) hipLaunchKernelGGL(k_wire_sum, dim3(256 * 4), dim3(256), 0, stream, (uint8_t*)sink.wire_dev, sink.wire_cap);
                HIP_TRY(hipGetLastError());
                return MTG_OK;
            };
            /* the results are written right away, on the assumption that the arenas are large enough (they are, from the second batch of a
             * shape on): the totals tell */
            HIP_TRY(hipEventRecord(eve, stream));
            if (int erc = emit()) return erc;
            HIP_TRY(hipEventRecord(ev2, stream));
            tick("host prep+launch");
            if (while_busy && !busy_done) { busy_done = true; (*while_busy)(); tick("host work during kernels"); }
            HIP_TRY(hipEventSynchronize(ev2));
            traversal_lock.unlock();
            tick("totals ready");
            t0 = now_ms();
            PartTot tot = *h_tot;
            /* an array that was too small: make it larger and write the launch's results again (its scratch is still in place) */
            const uint64_t need_w = tot.end[0] * 8 + 64, need_m = tot.end[1] * 20 + 64;
            bool again = false;
            if (want_records && tot.end[2] > sink.seq_cap) {
                /* the block may move: what earlier launches of the batch left in it is kept, the records that point there follow */
                const uintptr_t old = (uintptr_t)sink.seq, old_end = old + sink.seq_cap;
                if (!sink.grow_seq || !sink.grow_seq((size_t)tot.end[2], (size_t)tot.begin[2])) { set_error("sequence buffer too small: %llu bytes needed", (unsigned long long)tot.end[2]); return MTG_ERR_ARG; }
                if (tot.begin[2] > 0 && (uintptr_t)sink.seq != old)
                    for (size_t i = 0; i < n; i++) { const uintptr_t q = (uintptr_t)sink.fil[i].seq; if (q >= old && q < old_end) sink.fil[i].seq = sink.seq + (q - old); }
                again = true;
            }
            if (want_records && tot.end[3] > sink.ext_cap) {
                const uintptr_t old = (uintptr_t)sink.ext, old_end = old + sink.ext_cap;
                if (!sink.grow_ext || !sink.grow_ext((size_t)tot.end[3], (size_t)tot.begin[3])) { set_error("extension buffer too small: %llu bytes needed", (unsigned long long)tot.end[3]); return MTG_ERR_NOMEM; }
                if (launches > 1 && (uintptr_t)sink.ext != old)
                    for (size_t i = 0; i < n; i++) { const uintptr_t q = (uintptr_t)sink.res[i].extension; if (q >= old && q < old_end) sink.res[i].extension = sink.ext + (q - old); }
                again = true;
            }
            if (need_w > ws.cap[d_dw.slot] || need_m > ws.cap[d_dm.slot]) again = true;
            if (again) {
                /* the dense arrays hold this launch only (offsets relative to the batch: the launch's part is copied from begin[]) */
                HIP_TRY(hipStreamSynchronize(stream));
                /* what earlier launches of the batch left in the arena stays: when the text is formatted on the device the arena's only copy is this one
                 * (round 4: a batch of several launches lost the sequences of all but its last launch here -- the device formatter then wrote
                 * "_len_0" records; found by running the GPU tests under MAX_CHUNK) */
                if (!sink.seq_dev) HIP_TRY(d_seq.grow_keeping(std::max<size_t>(sink.seq_cap, 64), (size_t)tot.begin[2]));
                HIP_TRY(d_ext.alloc(std::max<size_t>(sink.ext_cap, 64)));
                HIP_TRY(d_dw.alloc(need_w));
                HIP_TRY(d_dm.alloc(need_m));
                /* k_emit made the records' offsets absolute: run the layout again from the launch's begin */
                HIP_TRY(hipMemcpyAsync(d_cnt.p, tot.begin, 32, hipMemcpyHostToDevice, stream));
                hipLaunchKernelGGL(k_scan1, dim3(nblocks), dim3(SCAN_SL), 0, stream, d_rec.as<SlotRec>(), m, d_blocks.as<ScanBlock>());
                hipLaunchKernelGGL(k_scan2, dim3(1), dim3(256), 0, stream, d_blocks.as<ScanBlock>(), nblocks, d_cnt.as<unsigned long long>(), d_tot.as<PartTot>());
                if (int erc = emit()) return erc;
            }
            /* bring the launch's results to the host.  Result copies of six batches at once share the link worse than two or three do
             * (scripts/pcie_d2h.py: 57 GB/s with two streams copying, 47-52 with six), so the batches of a device take turns */
            CopyTurn copy_turn(idx->device);
            std::vector<mtg_gap_result> tmp_res;
            std::vector<mtg_filled> tmp_fil;
            if (want_records) {
                if (identity) {
                    HIP_TRY(hipMemcpyAsync(sink.res, d_res.p, (size_t)m * sizeof(mtg_gap_result), hipMemcpyDeviceToHost, stream));
                    HIP_TRY(hipMemcpyAsync(sink.fil, d_fil.p, (size_t)m * sizeof(mtg_filled), hipMemcpyDeviceToHost, stream));
                } else {
                    tmp_res.resize(m);
                    tmp_fil.resize(m);
                    HIP_TRY(hipMemcpyAsync(tmp_res.data(), d_res.p, (size_t)m * sizeof(mtg_gap_result), hipMemcpyDeviceToHost, stream));
                    HIP_TRY(hipMemcpyAsync(tmp_fil.data(), d_fil.p, (size_t)m * sizeof(mtg_filled), hipMemcpyDeviceToHost, stream));
                }
                /* a batch that left in relocatable form has its sequences in the payload's sequence section */
                const WireLayout wl = wire_layout(m, tot.n_filled, tot.end[2], tot.end[3]);
                const bool wired = sink.wire_dev != nullptr && identity && tier == 0 && wl.total <= sink.wire_cap && tot.n_retry == 0 && tot.n_general == 0;
                if (sink.wire_dev && identity && tier == 0) { sink.wire_ok = wired; sink.wire_bytes = wired ? wl.total : 0; }
                const char* seq_src = wired ? (const char*)sink.wire_dev + wl.o_seq : (sink.seq_dev ? sink.seq_dev : d_seq.as<char>());
                if (!sink.seq_on_device && !sink.seq_stays_in_workspace && tot.end[2] > tot.begin[2]) HIP_TRY(hipMemcpyAsync(sink.seq + tot.begin[2], seq_src + tot.begin[2], tot.end[2] - tot.begin[2], hipMemcpyDeviceToHost, stream));
                if (tot.end[3] > tot.begin[3]) HIP_TRY(hipMemcpyAsync(sink.ext + tot.begin[3], d_ext.as<char>() + tot.begin[3], tot.end[3] - tot.begin[3], hipMemcpyDeviceToHost, stream));
            }
            std::vector<uint32_t> rlist(tot.n_retry), glist(tot.n_general);
            HostChunk* hc = nullptr;
            SlotRec* h_rec = nullptr;
            uint64_t* h_w = nullptr;
            uint32_t* h_m = nullptr;
            const uint64_t tw = tot.end[0], tc = tot.end[1];
            if (tot.n_retry) HIP_TRY(hipMemcpyAsync(rlist.data(), d_rlist.p, (size_t)tot.n_retry * 4, hipMemcpyDeviceToHost, stream));
            if (tot.n_general) {
                special.chunks.emplace_back(new HostChunk());
                hc = special.chunks.back().get();
                hc->carve(m, tw, tc, h_rec, h_w, h_m);
                HIP_TRY(hipMemcpyAsync(glist.data(), d_glist.p, (size_t)tot.n_general * 4, hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipMemcpyAsync(h_rec, d_rec.p, (size_t)m * sizeof(SlotRec), hipMemcpyDeviceToHost, stream));
                if (tw) HIP_TRY(hipMemcpyAsync(h_w, d_dw.p, tw * 8, hipMemcpyDeviceToHost, stream));
                if (tc) HIP_TRY(hipMemcpyAsync(h_m, d_dm.p, tc * 20, hipMemcpyDeviceToHost, stream));
            }
            HIP_TRY(hipEventRecord(ev3, stream));
            HIP_TRY(hipEventSynchronize(ev3));
            copy_turn.release();
            st.d2h_ms += now_ms() - t0;
            tick("results on the host");
            t0 = now_ms();
            st.index_lines += tot.lines; st.contig_nt += tot.contig_nt; st.store_runs += tot.store_runs; st.run_nt += tot.run_nt; st.post_lines += tot.post_lines;
            st.contig_words += tot.contig_words; st.coverage_kmers += tot.cov_kmers; st.dense_words += tw;
            st.copy_words += tot.copy_words; st.copy_cmds += tot.copy_cmds; st.coverage_direct_kmers += tot.cov_direct; st.n_lean_gaps += tot.n_lean;
            if (tier == 0 && identity) ws.post_general = m - (uint32_t)std::min<uint64_t>(tot.n_lean, m);
            st.copy_words_executed += tot.copy_words_exec; st.copy_cmds_executed += tot.copy_cmds_exec; st.post_scanned_words += tot.scan_words;
            sink.seq_used = tot.end[2];
            sink.ext_used = tot.end[3];
            sink.n_filled += tot.n_filled;
            if (want_records && !identity) /* records of a partial launch: to their gaps (a gap to be re-run gets its record again later) */
                for (uint32_t s2 = 0; s2 < m; s2++) { sink.res[host_ids[s2]] = tmp_res[s2]; sink.fil[host_ids[s2]] = tmp_fil[s2]; }
            for (uint32_t s2 : rlist) retry.push_back(host_ids ? host_ids[s2] : s2);
            if (tot.n_retry) sink.in_gap_order = false;
            if (hc) {
                h_w[tw] = 0;
                if (host_ids) hc->gap_of.assign(host_ids, host_ids + m);
                const uint32_t chunk_id = (uint32_t)special.chunks.size() - 1;
                std::vector<uint32_t> pslots; /* multi-contig gaps: their contig-graph paths, while the launch's scratch is still in place */
                for (uint32_t s2 : glist) {
                    special.special.push_back(SpecialGap{host_ids ? host_ids[s2] : s2, chunk_id, s2});
                    if (!host_paths && !in.want_all_contigs && h_rec[s2].p.fast == 0 && h_rec[s2].p.nb_terminal > 0) pslots.push_back(s2);
                }
                if (!pslots.empty()) {
                    HIP_TRY(d_ids.alloc(std::max<size_t>(chunk, pslots.size()) * 4)); /* the traversal is over: its slot map is free */
                    HIP_TRY(hipMemcpyAsync(d_ids.p, pslots.data(), pslots.size() * 4, hipMemcpyHostToDevice, stream));
                    HIP_TRY(d_paths.alloc(pslots.size() * (size_t)PATHS_WORDS * 4));
                    hipLaunchKernelGGL(k_paths, dim3((unsigned)pslots.size()), dim3(64), 0, stream, cfg, d_raw.as<uint8_t>(), d_out.as<GapOut>(), d_ids.as<uint32_t>(), k,
                                       d_paths.as<uint32_t>(), (uint32_t)pslots.size());
                    HIP_TRY(hipGetLastError());
                    hc->paths.resize(pslots.size() * (size_t)PATHS_WORDS);
                    HIP_TRY(hipMemcpyAsync(hc->paths.data(), d_paths.p, hc->paths.size() * 4, hipMemcpyDeviceToHost, stream));
                    HIP_TRY(hipStreamSynchronize(stream));
                    hc->path_of.assign(m, -1);
                    for (size_t g2 = 0; g2 < pslots.size(); g2++) hc->path_of[pslots[g2]] = (int32_t)g2;
                }
            }
            st.host_ms += now_ms() - t0;
            float ms = 0, ms2 = 0, ms3 = 0, msc = 0, msf = 0;
            HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
            HIP_TRY(hipEventElapsedTime(&msf, evf, ev1));
            st.finish_kernel_ms += msf;
            if (!classic_walk) { const uint32_t np = *(const uint32_t*)((const uint8_t*)h_tot + sizeof(PartTot)); st.n_parked_gaps += np; st.n_rounds += (uint64_t)rounds; if (tier == 0 && identity && m >= 64) {
                    const uint32_t share = (uint32_t)std::min<uint64_t>(((uint64_t)np << 16) / m, 65536u);
                    ws.mode_share[wmode] = share;
                    if (!wmode) { ws.park_share = share; idx->park_share_any.store(share, std::memory_order_relaxed); }
                    if (whole) { ws.mode_ns_per_gap[wmode] = ms * 1e6f / (float)m; ws.mode_launches++; } /* ev0 .. ev1: the walk, its rounds and the finishing kernel (with whatever else the device was doing: the launches of a workspace see the same company) */
                } }
            HIP_TRY(hipEventElapsedTime(&msc, evl0, evc)); /* k_lean + k_copy; when the finishing kernel runs on the second stream they start behind the walk kernel, next to it */
            { float msl = 0; HIP_TRY(hipEventElapsedTime(&msl, evl0, evl)); st.lean_kernel_ms += msl; }
            HIP_TRY(hipEventElapsedTime(&ms2, evc, eve));
            st.copy_kernel_ms += msc;
            HIP_TRY(hipEventElapsedTime(&ms3, eve, ev2));
            { float mss = 0; HIP_TRY(hipEventElapsedTime(&mss, ev0, ev2)); st.device_span_ms += mss; }
            st.kernel_ms += ms;
            st.post_kernel_ms += ms2;
            st.emit_kernel_ms += ms3;
            st.seq_bytes += tot.end[2] - tot.begin[2];
            st.n_launches++;
        }
        if (tier > 0) st.n_retried_gaps += n_todo;
        todo.swap(retry);
        n_todo = todo.size();
    }
#ifdef MTG_FINISH_DEBUG
    {
        unsigned int hd[64];
        if (hipMemcpyFromSymbol(hd, HIP_SYMBOL(mtg::g_dbg), sizeof hd) == hipSuccess) {
            for (int i = 0; i < 64; i++) if (hd[i]) fprintf(stderr, "  [finish debug] guard %d tripped %u times\n", i, hd[i]);
            unsigned int z[64] = {0};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(mtg::g_dbg), z, sizeof z);
        }
    }
#endif
#ifdef MTG_STAMPS
    {
        unsigned long long hs[16];
        if (hipMemcpyFromSymbol(hs, HIP_SYMBOL(mtg::g_stamps), sizeof hs) == hipSuccess && hs[15])
            fprintf(stderr, "  [stamps] lanes %llu  avg cycles/lane: W %.0f (long steps %.0f, bucket reads + run set-up %.0f) B %.0f | find_end %.0f dfs %.0f validate %.0f mark_inv %.0f | snp_fast %.0f (in-branching checks of find_end / alignment %.0f) consume %.0f"
                            " | per lane: general bubbles %.3f, snp_fast answers %.3f (bulk form %.3f, with alignment %.3f)\n", hs[15],
                    (double)hs[0] / hs[15], (double)hs[8] / hs[15], (double)hs[9] / hs[15], (double)hs[1] / hs[15], (double)hs[2] / hs[15], (double)hs[3] / hs[15], (double)hs[4] / hs[15], (double)hs[5] / hs[15],
                    (double)hs[6] / hs[15], (double)hs[12] / hs[15], (double)hs[7] / hs[15], (double)hs[10] / hs[15], (double)hs[14] / hs[15], (double)hs[13] / hs[15], (double)hs[11] / hs[15]);
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(mtg::g_stamps), z, sizeof z);
        unsigned long long fe[16];
        if (hipMemcpyFromSymbol(fe, HIP_SYMBOL(mtg::g_fe), sizeof fe) == hipSuccess && fe[0])
            fprintf(stderr, "  [stamps] find_end_of_branching: %llu calls; per call: levels %.2f nodes %.2f skips %.2f | ticks: skip section %.0f (left junction %.0f) children from the store %.0f ADJ read + run set-up %.0f visited set + involved list %.0f\n",
                    fe[0], (double)fe[1] / fe[0], (double)fe[7] / fe[0], (double)fe[8] / fe[0], (double)fe[2] / fe[0], (double)fe[3] / fe[0], (double)fe[4] / fe[0], (double)fe[5] / fe[0], (double)fe[6] / fe[0]);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(mtg::g_fe), z, sizeof z);
        unsigned long long lf[40];
        if (hipMemcpyFromSymbol(lf, HIP_SYMBOL(mtg::g_life), sizeof lf) == hipSuccess) {
            fprintf(stderr, "  [stamps] traversal kernel: %llu ticks from the first lane's start to the last lane's end (%.3f ms of events); lanes by log2(life in ticks):", lf[33] - lf[32], st.kernel_ms);
            for (int i = 10; i < 32; i++) if (lf[i]) fprintf(stderr, " 2^%d:%llu", i, lf[i]);
            fprintf(stderr, "\n");
        }
        unsigned long long z2[40] = {0};
        z2[32] = ~0ull;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(mtg::g_life), z2, sizeof z2);
    }
#endif
    if (while_busy && !busy_done) (*while_busy)();
    if (rc == MTG_OK && n_todo) {
        set_error("%zu gap(s) exceeded the largest traversal scratch tier", n_todo);
        rc = MTG_ERR_OVERFLOW;
    }
    if (launches > 1) sink.in_gap_order = false;
    sink.device_records_whole = rc == MTG_OK && launches == 1 && st.n_retried_gaps == 0 && special.special.empty();
    if (stats) *stats = st;
    return rc;
}

int workspace_arena_download(const mtg_index* idx, Workspace* ws, char* dst, uint64_t bytes)
{
    if (int rc = use_device_of(idx)) return rc;
    if (!ws || !ws->ptr[Workspace::SLOT_SEQ] || ws->cap[Workspace::SLOT_SEQ] < bytes) { set_error("the workspace holds no sequence arena of %llu bytes", (unsigned long long)bytes); return MTG_ERR_ARG; }
    if (bytes) HIP_TRY(hipMemcpy(dst, ws->ptr[Workspace::SLOT_SEQ], bytes, hipMemcpyDeviceToHost));
    return MTG_OK;
}
/* the text of a batch's simple sites, formatted on the device and brought to page-locked host memory (mtg_internal.h: FormatIn / FormatOut) */
int format_run(const mtg_index* idx, const FormatIn& fi, FormatOut& out)
{
    if (int rc = use_device_of(idx)) return rc;
    if (!fi.ws || !fi.ws->stream) { set_error("format_run: no batch has run on this workspace"); return MTG_ERR_ARG; }
    Workspace& ws = *fi.ws;
    const hipStream_t stream = (hipStream_t)ws.stream;
    const uint32_t n = (uint32_t)fi.n;
    out.n = n; out.n_simple = 0;
    out.bytes[0] = out.bytes[1] = out.bytes[2] = 0;
    out.complex_sites.clear();
    for (int s2 = 0; s2 < FMT_STREAMS; s2++) out.complex_off[s2].clear();
    if (n == 0) return MTG_OK;
    int slot = Workspace::SLOT_FMT0;
    auto wsbuf = [&]() { WsBuf b; b.ws = &ws; b.slot = slot++; return b; };
    WsBuf d_names = wsbuf(), d_rec = wsbuf(), d_cplx = wsbuf(), d_tot = wsbuf(), d_o0 = wsbuf(), d_o1 = wsbuf(), d_o2 = wsbuf(), d_recs_up = wsbuf();
    HIP_TRY(d_names.alloc((size_t)n * 12));
    HIP_TRY(d_rec.alloc((size_t)n * sizeof(FmtRec)));
    HIP_TRY(d_cplx.alloc((size_t)n * 28 + 64));
    HIP_TRY(d_tot.alloc(64));
    HIP_TRY(hipMemcpyAsync(d_names.p, fi.name_off, (size_t)n * 8, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync((uint8_t*)d_names.p + (size_t)n * 8, fi.name_len, (size_t)n * 4, hipMemcpyHostToDevice, stream));
    FmtArgs A;
    if (fi.device_records_whole && ws.ptr[Workspace::SLOT_RES] && ws.ptr[Workspace::SLOT_FIL]) {
        A.res = (const mtg_gap_result*)ws.ptr[Workspace::SLOT_RES];
        A.fil = (const mtg_filled*)ws.ptr[Workspace::SLOT_FIL];
    } else { /* several launches, re-run gaps, gaps the host finished: the records as the host has them go up */
        HIP_TRY(d_recs_up.alloc((size_t)n * (sizeof(mtg_gap_result) + sizeof(mtg_filled))));
        HIP_TRY(hipMemcpyAsync(d_recs_up.p, fi.res, (size_t)n * sizeof(mtg_gap_result), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync((uint8_t*)d_recs_up.p + (size_t)n * sizeof(mtg_gap_result), fi.fil, (size_t)n * sizeof(mtg_filled), hipMemcpyHostToDevice, stream));
        A.res = (const mtg_gap_result*)d_recs_up.p;
        A.fil = (const mtg_filled*)((uint8_t*)d_recs_up.p + (size_t)n * sizeof(mtg_gap_result));
    }
    const uint8_t* c = (const uint8_t*)ws.ptr[Workspace::SLOT_TEXT_BLOCK];
    A.text = (const char*)(c + FillInput::text_block_off(fi.n, fi.nt, 5));
    A.source_off = (const uint64_t*)(c + FillInput::text_block_off(fi.n, fi.nt, 0));
    A.source_len = (const uint32_t*)(c + FillInput::text_block_off(fi.n, fi.nt, 3));
    A.name_off = (const uint64_t*)d_names.p;
    A.name_len = (const uint32_t*)((uint8_t*)d_names.p + (size_t)n * 8);
    A.d_seq = (const char*)ws.ptr[Workspace::SLOT_SEQ];
    A.host_seq = (uint64_t)(uintptr_t)fi.host_seq;
    A.seq_used = fi.seq_used;
    A.host_fil = (uint64_t)(uintptr_t)fi.fil;
    A.rec = d_rec.as<FmtRec>();
    A.out[0] = A.out[1] = A.out[2] = nullptr;
    A.n = n;
    EventSet events;
    hipEvent_t e0, e1;
    HIP_TRY(events.make(e0));
    HIP_TRY(events.make(e1));
    HIP_TRY(hipEventRecord(e0, stream));
    hipLaunchKernelGGL(k_fmt_size, dim3(n), dim3(64), 0, stream, A);
    hipLaunchKernelGGL(k_fmt_scan, dim3(1), dim3(1024), 0, stream, d_rec.as<FmtRec>(), n, d_cplx.as<uint32_t>(), (uint64_t*)((uint8_t*)d_cplx.p + (((size_t)n * 4 + 7) & ~(size_t)7)), d_tot.as<unsigned long long>());
    HIP_TRY(hipGetLastError());
    unsigned long long* h_tot = (unsigned long long*)staging_host(&ws, Workspace::NHOST - 2, 64);
    if (!h_tot) { set_error("no page-locked memory"); return MTG_ERR_NOMEM; }
    HIP_TRY(hipMemcpyAsync(h_tot, d_tot.p, 40, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    WsBuf* d_o[3] = {&d_o0, &d_o1, &d_o2};
    for (int s2 = 0; s2 < FMT_STREAMS; s2++) {
        out.bytes[s2] = h_tot[s2];
        HIP_TRY(d_o[s2]->alloc((size_t)h_tot[s2] + 64));
        A.out[s2] = d_o[s2]->as<char>();
        if (out.cap[s2] < h_tot[s2] + 64) {
            pinned_free(out.text[s2]);
            out.cap[s2] = (size_t)h_tot[s2] + (size_t)h_tot[s2] / 4 + 4096;
            out.text[s2] = (char*)pinned_alloc(out.cap[s2]);
            if (!out.text[s2]) { out.cap[s2] = 0; set_error("no page-locked memory for %llu bytes of text", h_tot[s2]); return MTG_ERR_NOMEM; }
        }
    }
    out.n_simple = h_tot[3];
    const size_t nc = (size_t)h_tot[4];
    hipLaunchKernelGGL(k_fmt_write, dim3(n), dim3(64), 0, stream, A);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e1, stream));
    {
        CopyTurn copy_turn(idx->device);
        for (int s2 = 0; s2 < FMT_STREAMS; s2++)
            if (out.bytes[s2]) HIP_TRY(hipMemcpyAsync(out.text[s2], A.out[s2], out.bytes[s2], hipMemcpyDeviceToHost, stream));
        std::vector<uint64_t> co(3 * nc);
        out.complex_sites.resize(nc);
        if (nc) {
            HIP_TRY(hipMemcpyAsync(out.complex_sites.data(), d_cplx.p, nc * 4, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipMemcpyAsync(co.data(), (uint8_t*)d_cplx.p + (((size_t)n * 4 + 7) & ~(size_t)7), nc * 24, hipMemcpyDeviceToHost, stream));
        }
        HIP_TRY(hipStreamSynchronize(stream));
        for (int s2 = 0; s2 < FMT_STREAMS; s2++) { out.complex_off[s2].resize(nc); for (size_t i = 0; i < nc; i++) out.complex_off[s2][i] = co[3 * i + s2]; }
    }
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    out.kernel_ms = ms;
    return MTG_OK;
}

int nw_run(const mtg_index* idx, const std::vector<NwPair>& pairs, std::vector<uint32_t>& matches, Workspace* ws)
{
    (void)idx;
    if (int rc = use_device_of(idx)) return rc;
    const size_t np = pairs.size();
    matches.assign(np, 0);
    if (np == 0) return MTG_OK;
    std::vector<uint64_t> oa(np), ob(np), bo(np);
    std::vector<uint32_t> la(np), lb(np);
    uint64_t nt = 0, nbnd = 0;
    for (size_t i = 0; i < np; i++) { oa[i] = nt; nt += pairs[i].na; ob[i] = nt; nt += pairs[i].nb; la[i] = pairs[i].na; lb[i] = pairs[i].nb; bo[i] = nbnd; nbnd += (uint64_t)pairs[i].na + 1; }
    std::vector<uint8_t> text(nt + 1);
    for (size_t i = 0; i < np; i++) { memcpy(text.data() + oa[i], pairs[i].a, pairs[i].na); memcpy(text.data() + ob[i], pairs[i].b, pairs[i].nb); }
    hipStream_t stream = ws ? (hipStream_t)ws->stream : nullptr;
    CallBuf d_text, d_oa, d_ob, d_la, d_lb, d_bo, d_bnd, d_out;
    const auto up = [&](CallBuf& b, int slot, const void* src, size_t bytes) -> hipError_t {
        const hipError_t e = b.alloc(ws, CALL_SLOT0 + slot, bytes);
        if (e != hipSuccess || bytes == 0) return e;
        return hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, stream);
    };
    HIP_TRY(up(d_text, 0, text.data(), text.size())); HIP_TRY(up(d_oa, 1, oa.data(), np * 8)); HIP_TRY(up(d_ob, 2, ob.data(), np * 8)); HIP_TRY(up(d_la, 3, la.data(), np * 4));
    HIP_TRY(up(d_lb, 4, lb.data(), np * 4)); HIP_TRY(up(d_bo, 5, bo.data(), np * 8));
    HIP_TRY(d_bnd.alloc(ws, CALL_SLOT0 + 6, nbnd * sizeof(int2)));
    HIP_TRY(d_out.alloc(ws, CALL_SLOT0 + 7, np * 4));
    hipLaunchKernelGGL(k_nw, dim3((unsigned)np), dim3(64), 0, stream, d_text.as<uint8_t>(), d_oa.as<uint64_t>(), d_la.as<uint32_t>(), d_ob.as<uint64_t>(), d_lb.as<uint32_t>(),
                       d_bnd.as<int2>(), d_bo.as<uint64_t>(), d_out.as<uint32_t>(), (uint32_t)np);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(matches.data(), d_out.p, np * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return MTG_OK;
}

int scan_run(const mtg_index* idx, const uint64_t* words, size_t nwords, const uint64_t* word_off, const uint32_t* len, size_t nseq, int mode, uint64_t* out_bits,
             int device_ptrs, mtg_scan_stats* st)
{
    if (int rc = use_device_of(idx)) return rc;
    if (!idx || !idx->dev.bloom.bits) { set_error("the index has no Bloom filter (MTG_BLOOM_BITS=0)"); return MTG_ERR_ARG; }
    if (nseq == 0) return MTG_OK;
    DevBuf d_w, d_o, d_l, d_b, d_c;
    const uint64_t *pw = words, *po = word_off;
    const uint32_t* pl = len;
    uint64_t* pb = out_bits;
    if (!device_ptrs) {
        HIP_TRY(d_w.alloc(nwords * 8)); HIP_TRY(d_o.alloc(nseq * 8)); HIP_TRY(d_l.alloc(nseq * 4)); HIP_TRY(d_b.alloc(nwords * 8));
        HIP_TRY(hipMemcpy(d_w.p, words, nwords * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_o.p, word_off, nseq * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_l.p, len, nseq * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemset(d_b.p, 0, nwords * 8));
        pw = d_w.as<uint64_t>(); po = d_o.as<uint64_t>(); pl = d_l.as<uint32_t>(); pb = d_b.as<uint64_t>();
    }
    HIP_TRY(d_c.alloc(32));
    HIP_TRY(hipMemset(d_c.p, 0, 32));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_scan, dim3((unsigned)std::min<size_t>(nseq, 256 * 16)), dim3(SCAN_TILE), 0, 0, idx->dev, pw, po, pl, nseq, mode, pb, d_c.as<unsigned long long>());
    HIP_TRY(hipEventRecord(e1, 0));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    unsigned long long c[4];
    HIP_TRY(hipMemcpy(c, d_c.p, 32, hipMemcpyDeviceToHost));
    if (!device_ptrs) HIP_TRY(hipMemcpy(out_bits, pb, nwords * 8, hipMemcpyDeviceToHost));
    if (st) { st->n_kmers = c[0]; st->bloom_positive = c[1]; st->confirmed = c[2]; st->blocks_staged = c[3]; st->kernel_ms = ms; }
    return MTG_OK;
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
        if (!legacy_build()) {
            /* the lean build: the solid k-mers' junctions into the junction table; their abundances stay in the count table (one counting
             * pass) or go into an ABND table of their own (several: the count table of a pass does not outlive it) */
            prof.host_phase("count_reads", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - prof.t0).count(), 0, n_solid);
            DevBuf jt_buf, abnd_buf;
            Table jt{}, abnd{};
            unsigned long long cnt[4] = {0, 0, 0, 0};
            for (int ia = 0; ia < 6; ia++) {
                if (int rc2 = alloc_slot_table(jt, jt_buf, n_solid + n_solid / 8 + 1024, jt_load() * load, 2 * (k - 1), prof, "clear_jt", adj_bytes_estimate(n_solid, k))) return rc2;
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
        for (int ia = 0; ia < 6; ia++) {
            free_tables(idx);
            rc = alloc_tables(idx, n_solid, load);
            if (rc) return rc;
            HIP_TRY(hipMemset(d_cnt.p, 0, 32));
            for (uint32_t pass = 0; pass < npass; pass++) {
                if (npass > 1) { /* the table of the wanted pass has to be counted again (with one pass it still holds round 1's counts) */
                    bool ovf2 = false;
                    if (int rc2 = count_pass(pass, ovf2)) return rc2;
                    if (ovf2) { set_error("k-mer count table overflowed on a repeated pass"); return MTG_ERR_OVERFLOW; }
                }
                hipLaunchKernelGGL(k_insert_from_counts, dim3(256 * 16), dim3(256), 0, 0, idx->dev, t, lo, hi, d_cnt.as<unsigned long long>());
                HIP_TRY(hipGetLastError());
            }
            unsigned long long cnt[4];
            HIP_TRY(hipMemcpy(cnt, d_cnt.p, 32, hipMemcpyDeviceToHost));
            idx->info.nb_saturated = cnt[3];
            if (!cnt[0]) { rc = MTG_OK; break; }
            load *= 0.7;
            rc = MTG_ERR_OVERFLOW;
            set_error("index bucket displacement overflow");
        }
        if (rc) return rc;
        (void)d_keys.alloc(0); /* the count table is done with: room for the unitig construction */
        (void)d_cnts.alloc(0);
        {
            const uint64_t nslots = idx->dev.abnd.nbuckets * MTG_ABND_SLOTS;
            hipLaunchKernelGGL(k_lookahead_table, dim3((unsigned)std::min<uint64_t>((nslots + 255) / 256, 256 * 32)), dim3(256), 0, 0, idx->dev);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipDeviceSynchronize());
        }
        if (int rc2 = build_unitigs(idx)) return rc2;
        idx->info.k = k;
        idx->info.abundance_min = abundance_min;
        idx->info.abundance_auto = autoc;
        *out = g.release();
        return MTG_OK;
    }
    set_error("k-mer count table kept overflowing");
    return MTG_ERR_OVERFLOW;
}

int bench_random_lines(uint64_t table_bytes, uint64_t n_chains, uint32_t chain_len, uint32_t line_bytes, double* ms_out, double* gbps)
{
    if (int rc = ensure_device()) return rc;
    if (line_bytes != 16 && line_bytes != 32 && line_bytes != 64 && line_bytes != 128) { set_error("line_bytes must be 16, 32, 64 or 128"); return MTG_ERR_ARG; }
    const uint64_t nlines = table_bytes / line_bytes;
    if (nlines == 0 || n_chains == 0 || chain_len == 0) { set_error("invalid argument"); return MTG_ERR_ARG; }
    DevBuf tab, sink;
    HIP_TRY(tab.alloc(nlines * line_bytes));
    HIP_TRY(sink.alloc(8));
    hipLaunchKernelGGL(k_fill_random, dim3(256 * 16), dim3(256), 0, 0, tab.as<uint64_t>(), nlines * (line_bytes / 8), 12345ull);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    const uint32_t blocks = (uint32_t)((n_chains + 63) / 64);
    auto launch = [&](uint32_t len) {
        switch (line_bytes) {
            case 16: hipLaunchKernelGGL(k_chase<16>, dim3(blocks), dim3(64), 0, 0, tab.as<uint64_t>(), nlines, n_chains, len, sink.as<uint64_t>()); break;
            case 32: hipLaunchKernelGGL(k_chase<32>, dim3(blocks), dim3(64), 0, 0, tab.as<uint64_t>(), nlines, n_chains, len, sink.as<uint64_t>()); break;
            case 64: hipLaunchKernelGGL(k_chase<64>, dim3(blocks), dim3(64), 0, 0, tab.as<uint64_t>(), nlines, n_chains, len, sink.as<uint64_t>()); break;
            default: hipLaunchKernelGGL(k_chase<128>, dim3(blocks), dim3(64), 0, 0, tab.as<uint64_t>(), nlines, n_chains, len, sink.as<uint64_t>()); break;
        }
    };
    launch(std::min<uint32_t>(chain_len, 64));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipEventRecord(e0, 0));
    launch(chain_len);
    HIP_TRY(hipEventRecord(e1, 0));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (ms_out) *ms_out = ms;
    if (gbps) *gbps = (double)n_chains * chain_len * (double)line_bytes / (ms * 1e-3) / 1e9;
    return MTG_OK;
}

} // namespace mtgi

/* ------------------------------------------------------------------------------------------------ C ABI (device side) */
namespace mtgi {
int index_from_kmers(const uint64_t*, const uint32_t*, size_t, int, mtg_index**);
int index_from_packed_device(const uint64_t*, const uint64_t*, const uint32_t*, size_t, uint64_t, int, uint32_t, uint32_t, mtg_index**);
int index_replicate(const mtg_index*, int, mtg_index**);
void index_release(mtg_index*);
int bench_random_lines(uint64_t, uint64_t, uint32_t, uint32_t, double*, double*);
}

extern "C" {

const char* mtg_last_error(void) { return mtgi::g_err; }
int mtg_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int mtg_set_device(int device)
{
    if (hipSetDevice(device) != hipSuccess) { mtgi::set_error("hipSetDevice(%d) failed", device); return MTG_ERR_NO_DEVICE; }
    return MTG_OK;
}
int mtg_index_create_from_kmers(const uint64_t* canon_kmers, const uint32_t* abundance, size_t n, int k, mtg_index** out)
{
    return mtgi::index_from_kmers(canon_kmers, abundance, n, k, out);
}
int mtg_index_create_from_packed_device(const uint64_t* d_words, const uint64_t* d_word_off, const uint32_t* d_len, size_t nseq, uint64_t ub, int k,
                                        uint32_t abund_lo, uint32_t abund_span, mtg_index** out)
{
    return mtgi::index_from_packed_device(d_words, d_word_off, d_len, nseq, ub, k, abund_lo, abund_span, out);
}
void mtg_index_free(mtg_index* idx) { mtgi::index_release(idx); }
int mtg_index_replicate(const mtg_index* idx, int device, mtg_index** out) { return mtgi::index_replicate(idx, device, out); }
int mtg_index_get_info(const mtg_index* idx, mtg_index_info* info)
{
    if (!idx || !info) { mtgi::set_error("null argument"); return MTG_ERR_ARG; }
    *info = idx->info;
    return MTG_OK;
}
int mtg_index_contains(const mtg_index* idx, const uint64_t* kmers, size_t n, uint8_t* out)
{
    std::vector<uint32_t> ab(n);
    int rc = mtgi::query_run(idx, kmers, n, ab.data(), nullptr, nullptr);
    if (rc) return rc;
    for (size_t i = 0; i < n; i++) out[i] = ab[i] != 0;
    return MTG_OK;
}
int mtg_index_abundance(const mtg_index* idx, const uint64_t* kmers, size_t n, uint32_t* out) { return mtgi::query_run(idx, kmers, n, out, nullptr, nullptr); }
int mtg_index_neighbors(const mtg_index* idx, const uint64_t* kmers, size_t n, uint8_t* succ, uint8_t* pred)
{
    return mtgi::query_run(idx, kmers, n, nullptr, succ, pred);
}
int mtg_index_scan_packed_device(const mtg_index* idx, const uint64_t* d_words, const uint64_t* d_word_off, const uint32_t* d_len, size_t nseq, int mode,
                                 uint64_t* d_out_bits, mtg_scan_stats* st)
{
    if (!d_words || !d_word_off || !d_len || !d_out_bits) { mtgi::set_error("null argument"); return MTG_ERR_ARG; }
    return mtgi::scan_run(idx, d_words, 0, d_word_off, d_len, nseq, mode, d_out_bits, 1, st);
}
int mtg_last_batch_stats(mtg_batch_stats* s)
{
    if (!s) return MTG_ERR_ARG;
    *s = mtgi::g_stats;
    return MTG_OK;
}
int mtg_bench_random_lines(uint64_t table_bytes, uint64_t n_chains, uint32_t chain_len, uint32_t line_bytes, double* ms, double* gbps)
{
    return mtgi::bench_random_lines(table_bytes, n_chains, chain_len, line_bytes, ms, gbps);
}
}
