/*
 * mtg_gpu_misc.hip -- the smaller device entries of libmtgfill.so (gfx950 only) and the device side of the C ABI.
 *
 *   k_query                          : batched contains / queryAbundance / successors / predecessors
 *   k_scan                           : rolling 2-bit k-mer + minimizer-blocked Bloom probe along packed sequences, blocks staged in LDS
 *   k_nw                             : Needleman-Wunsch match counts for the de-duplication of multi-path solutions, one wave per pair
 *                                      (remove_almost_identical_solutions, /root/reference/src/Utils.cpp:87-189,208-238)
 *   k_fmt_*                          : the tool's text (FASTA / info / VCF) of the sites with one solution (mtg_format.h)
 *   k_chase                          : dependent random 64-byte reads (measured roofline ceiling)
 */
#include "mtg_gpu_common.h"

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
    __shared__ uint64_t s_mh[SCAN_TILE + 32]; /* hashes of the tile's m-mers */
    const int k = ix.k, mm = ix.bloom.mm;
    const uint32_t span = (uint32_t)(k - mm);
    const uint64_t mk = kmask(k), mmask = kmask(mm);
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
            /* the minimizer of a k-mer is the smallest hash among its k - m + 1 m-mers and neighbouring k-mers share all but one of them: the
             * tile's m-mers are hashed once (256 + k - m of them, one or two per thread) and a k-mer takes the minimum over its window in LDS
             * (rounds 1-4: bloom_block hashed fifteen m-mers per k-mer -- the kernel was bound by those multiplications, not by its reads) */
            const uint32_t tile_k = npos - base < (uint32_t)SCAN_TILE ? npos - base : (uint32_t)SCAN_TILE, tile_m = tile_k + span;
            for (uint32_t t = tid; t < tile_m; t += SCAN_TILE) {
                const uint32_t j = base + t, sh = 2u * (j & 31u);
                uint64_t v = w[j >> 5] >> sh;
                if ((j & 31u) + (uint32_t)mm > 32u) v |= w[(j >> 5) + 1] << (64u - sh);
                s_mh[t] = bloom_mmer_hash(v & mmask, mm);
            }
            __syncthreads();
            Kmer x;
            x.f = x.r = 0;
            uint64_t b = ~0ull;
            if (valid) {
                x.r = le_kmer(w, p, mk) ^ (0xAAAAAAAAAAAAAAAAULL & mk);
                x.f = revcomp(x.r, k);
                uint64_t best = ~0ull;
                for (uint32_t t = 0; t <= span; t++) { const uint64_t h = s_mh[tid + t]; best = h < best ? h : best; }
                b = bloom_block_of_min(ix.bloom, best);
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

