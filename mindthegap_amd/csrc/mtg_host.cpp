/*
 * mtg_host.cpp -- host orchestration of libmtgfill.so.
 *
 * The device builds the contigs of every gap (stage A, mtg_gpu.hip) and answers abundance queries; this file
 * holds the rest of Filler::gapFillFromSource (/root/reference/src/Filler.cpp:854-1026) for a whole batch:
 *   contig graph      IGraphOutput::construct_graph / print_edges   src/IGraphOutput.cpp:97-133,144-179
 *   terminal nodes    Filler::find_nodes_containing_multiple_R      src/Filler.cpp:1294-1378
 *   reverse DFS       GraphAnalysis::find_all_paths_rev             src/GraphAnalysis.cpp:205-326
 *   path -> sequence  GraphAnalysis::paths_to_sequences             src/GraphAnalysis.cpp:331-460
 *   dedupe            remove_almost_identical_solutions             src/Utils.cpp:208-238 (NW identity :87-189)
 *   coverage / qual   src/Filler.cpp:959-1003, src/Utils.hpp:85-103, src/Utils.cpp:241-254
 * plus the index construction from read files (Graph::create, src/Filler.cpp:172-213) and the index container.
 * The temp-file round trips of the reference (contigs FASTA + dot graph per gap) are not reproduced.
 */
#include "mtg_internal.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <thread>
#include <unordered_map>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

using namespace mtg;

namespace mtgi {

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

/* ------------------------------------------------------------------------------------------------ sequence files */
bool read_sequences(const std::string& path, std::vector<std::pair<std::string, std::string>>& out)
{
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) return false;
    gzbuffer(f, 1 << 20);
    std::string line, pending;
    std::vector<char> buf(1 << 16);
    auto getl = [&](std::string& l) -> bool {
        l.clear();
        bool got = false;
        while (gzgets(f, buf.data(), (int)buf.size())) {
            got = true;
            size_t n = strlen(buf.data());
            if (n && buf[n - 1] == '\n') {
                l.append(buf.data(), n - 1);
                if (!l.empty() && l.back() == '\r') l.pop_back();
                return true;
            }
            l.append(buf.data(), n);
        }
        return got;
    };
    bool have = getl(line);
    while (have) {
        if (!line.empty() && line[0] == '>') {
            out.emplace_back(line.substr(1), std::string());
            while ((have = getl(line)) && (line.empty() || line[0] != '>')) out.back().second += line;
        } else if (!line.empty() && line[0] == '@') {
            out.emplace_back(line.substr(1), std::string());
            if ((have = getl(line))) out.back().second = line;
            have = getl(line);
            have = getl(line);
            have = getl(line);
        } else {
            have = getl(line);
        }
    }
    gzclose(f);
    return true;
}

/* ------------------------------------------------------------------------------------------------ index from reads */
static inline bool nt_bad(unsigned char c) { return (c >> 3) & 1; } /* gatb: bit 3 of the ASCII code flags 'N' */

/* gatb's automatic solidity cut-off, restated: smoothed histogram, first minimum, coverage peak,
 * arg-min between; floor 3 (src/Filler.cpp:201).  One golden datapoint: 7 (test/full_test/gold_fill.output:11). */
static int auto_cutoff(const std::vector<uint64_t>& h, int floor_thr)
{
    const size_t len = h.size();
    if (len < 5) return floor_thr;
    std::vector<double> sm(len, 0.0);
    sm[1] = 0.6 * h[1] + 0.4 * h[2];
    for (size_t i = 2; i + 1 < len; i++) sm[i] = 0.2 * h[i - 1] + 0.6 * h[i] + 0.2 * h[i + 1];
    size_t valley = 2;
    while (valley + 2 < len && !(sm[valley] < sm[valley + 1])) valley++;
    size_t peak = valley;
    for (size_t i = valley; i + 1 < len; i++) if (sm[i] > sm[peak]) peak = i;
    size_t best = valley;
    for (size_t i = valley; i <= peak; i++) if (sm[i] < sm[best]) best = i;
    return std::max((int)best, floor_thr);
}

struct HostIndexData {
    std::vector<uint64_t> kmers;
    std::vector<uint32_t> counts;
};
static std::unordered_map<const mtg_index*, HostIndexData>& host_copies()
{
    static std::unordered_map<const mtg_index*, HostIndexData> m;
    return m;
}
static std::mutex& host_copies_mtx() { static std::mutex m; return m; }

int index_from_reads(const char* paths_csv, int k, int abundance_min, int abundance_max, mtg_index** out)
{
    if (!paths_csv || !out || k < 11 || k > 31) { set_error("invalid argument (11 <= k <= 31)"); return MTG_ERR_ARG; }
    /* the reads, concatenated with '\n' separators (an invalid character by gatb's rule, so no k-mer spans two reads) */
    std::string text;
    std::string csv(paths_csv);
    size_t pos = 0;
    while (pos <= csv.size()) {
        size_t e = csv.find(',', pos);
        if (e == std::string::npos) e = csv.size();
        std::string path = csv.substr(pos, e - pos);
        pos = e + 1;
        if (path.empty()) continue;
        std::vector<std::pair<std::string, std::string>> recs;
        if (!read_sequences(path, recs)) { set_error("cannot read %s", path.c_str()); return MTG_ERR_IO; }
        size_t add = 0;
        for (auto& r : recs) add += r.second.size() + 1;
        text.reserve(text.size() + add);
        for (auto& r : recs) { text += r.second; text += '\n'; }
    }
    /* counting on the device (k_count); sum solidity over all files (STR_SOLIDITY_KIND "sum", src/Filler.cpp:177) */
    std::vector<uint64_t> histo(10003, 0); /* STR_HISTOGRAM_MAX 10000, src/Filler.cpp:200 */
    HostIndexData cand;
    const uint32_t keep_min = abundance_min < 0 ? 3u : (uint32_t)std::max(abundance_min, 1); /* auto never goes below 3 (src/Filler.cpp:201) */
    int rc = count_run(text.data(), text.size(), k, keep_min, histo, cand.kmers, cand.counts);
    if (rc) return rc;
    std::string().swap(text);
    int autoc = -1;
    if (abundance_min < 0) { autoc = auto_cutoff(histo, 3); abundance_min = autoc; }
    /* deterministic container: sort the candidates by k-mer */
    std::vector<size_t> order(cand.kmers.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cand.kmers[a] < cand.kmers[b]; });
    HostIndexData hd;
    for (size_t i : order)
        if ((int64_t)cand.counts[i] >= abundance_min && (abundance_max <= 0 || (int64_t)cand.counts[i] <= abundance_max)) { hd.kmers.push_back(cand.kmers[i]); hd.counts.push_back(cand.counts[i]); }
    rc = index_from_kmers(hd.kmers.data(), hd.counts.data(), hd.kmers.size(), k, out);
    if (rc) return rc;
    (*out)->info.abundance_min = abundance_min;
    (*out)->info.abundance_auto = autoc;
    std::lock_guard<std::mutex> lk(host_copies_mtx());
    host_copies()[*out] = std::move(hd);
    return MTG_OK;
}

static const char IDX_MAGIC[8] = {'M', 'T', 'G', 'I', 'D', 'X', '1', 0};

int index_save(const mtg_index* idx, const char* path)
{
    if (!idx || !path) { set_error("null argument"); return MTG_ERR_ARG; }
    std::lock_guard<std::mutex> lk(host_copies_mtx());
    auto it = host_copies().find(idx);
    if (it == host_copies().end()) { set_error("this index has no host copy of its k-mers and cannot be saved"); return MTG_ERR_ARG; }
    FILE* f = fopen(path, "wb");
    if (!f) { set_error("cannot write %s", path); return MTG_ERR_IO; }
    int32_t hdr[4] = {idx->info.k, idx->info.abundance_min, idx->info.abundance_auto, 0};
    uint64_t n = it->second.kmers.size();
    bool ok = fwrite(IDX_MAGIC, 1, 8, f) == 8 && fwrite(hdr, 4, 4, f) == 4 && fwrite(&n, 8, 1, f) == 1 &&
              fwrite(it->second.kmers.data(), 8, n, f) == n && fwrite(it->second.counts.data(), 4, n, f) == n;
    fclose(f);
    if (!ok) { set_error("short write on %s", path); return MTG_ERR_IO; }
    return MTG_OK;
}

int index_load(const char* path, mtg_index** out)
{
    if (!path || !out) { set_error("null argument"); return MTG_ERR_ARG; }
    FILE* f = fopen(path, "rb");
    if (!f) { set_error("cannot read %s", path); return MTG_ERR_IO; }
    char magic[8] = {0};
    int32_t hdr[4];
    uint64_t n = 0;
    if (fread(magic, 1, 8, f) != 8) { fclose(f); set_error("%s: truncated", path); return MTG_ERR_FORMAT; }
    if (memcmp(magic, "\x89HDF\r\n\x1a\n", 8) == 0) {
        fclose(f);
        set_error("%s is an HDF5 file: GATB .h5 graphs are not readable by this library (SURVEY.md 8f-2); build the index with -in", path);
        return MTG_ERR_FORMAT;
    }
    if (memcmp(magic, IDX_MAGIC, 8) != 0 || fread(hdr, 4, 4, f) != 4 || fread(&n, 8, 1, f) != 1) { fclose(f); set_error("%s: not a mtg index", path); return MTG_ERR_FORMAT; }
    HostIndexData hd;
    hd.kmers.resize(n);
    hd.counts.resize(n);
    bool ok = fread(hd.kmers.data(), 8, n, f) == n && fread(hd.counts.data(), 4, n, f) == n;
    fclose(f);
    if (!ok) { set_error("%s: truncated", path); return MTG_ERR_FORMAT; }
    int rc = index_from_kmers(hd.kmers.data(), hd.counts.data(), n, hdr[0], out);
    if (rc) return rc;
    (*out)->info.abundance_min = hdr[1];
    (*out)->info.abundance_auto = hdr[2];
    std::lock_guard<std::mutex> lk(host_copies_mtx());
    host_copies()[*out] = std::move(hd);
    return MTG_OK;
}

void index_forget_host_copy(const mtg_index* idx)
{
    std::lock_guard<std::mutex> lk(host_copies_mtx());
    host_copies().erase(idx);
}

/* ------------------------------------------------------------------------------------------------ gap post-processing */
struct TermInfo { /* info_node_t, src/Filler.hpp:44-71 */
    int node, pos, errors, target;
};

/* marshalling of a batch of gapFillFromSource calls (targets == nullptr: contigs only, the stage A parity entry) */
void FillInput::resize(size_t n)
{
    bytes_a = 34 * n8(n) + 64;
    block_a = ws ? staging_host(ws, 0, bytes_a) : nullptr;
    if (!block_a) { own_a.resize(bytes_a / 8 + 1); block_a = own_a.data(); }
    uint8_t* b = (uint8_t*)block_a;
    src.p = (uint64_t*)(b + off_a(n, 0)); r0.p = (uint64_t*)(b + off_a(n, 1));
    roff.p = (uint32_t*)(b + off_a(n, 2)); rlen.p = (uint32_t*)(b + off_a(n, 3)); toff.p = (uint32_t*)(b + off_a(n, 4)); tcnt.p = (uint32_t*)(b + off_a(n, 5));
    nbmis.p = b + off_a(n, 6); fast_ok.p = b + off_a(n, 7);
    src.n = r0.n = roff.n = rlen.n = toff.n = tcnt.n = nbmis.n = fast_ok.n = n;
}
void FillInput::alloc_b(uint64_t rw, uint64_t nt)
{
    bytes_b = 8 * rw + 64;
    block_b = ws ? staging_host(ws, 1, bytes_b) : nullptr;
    if (!block_b) { own_b.resize(bytes_b / 8 + 1); block_b = own_b.data(); }
    rwords.p = (uint64_t*)block_b; rwords.n = rw;
    bytes_c = (size_t)TARGET_SLOT * nt + 64;
    block_c = ws ? staging_host(ws, 2, bytes_c) : nullptr;
    if (!block_c) { own_c.resize(bytes_c / 8 + 1); block_c = own_c.data(); }
    traw.p = (uint8_t*)block_c; traw.n = (size_t)TARGET_SLOT * nt;
}
void FillInput::layout()
{
    uint64_t rw = 0, nt = 0;
    for (size_t i = 0; i < src.size(); i++) {
        roff[i] = (uint32_t)rw; rw += (rlen[i] + 31) / 32 + 1;
        toff[i] = (uint32_t)nt; nt += tcnt[i];
    }
    alloc_b(rw, nt);
}
/* the marshalling of a gap is a few dozen characters turned into 2-bit codes: eight at a time where the CPU has pext */
namespace {
inline uint64_t reverse_fields(uint64_t x) /* the 32 two-bit fields of x in reverse order */
{
    x = __builtin_bswap64(x);
    x = ((x & 0xF0F0F0F0F0F0F0F0ull) >> 4) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    return ((x & 0xCCCCCCCCCCCCCCCCull) >> 2) | ((x & 0x3333333333333333ull) << 2);
}
#if defined(__x86_64__)
__attribute__((target("bmi2"))) inline uint64_t pack32_bmi2(const char* b) /* 32 characters -> 32 codes, character i at bits 2i */
{
    uint64_t c[4];
    memcpy(c, b, 32);
    const uint64_t M = 0x0606060606060606ull; /* nt_code: bits 1-2 of the ASCII code */
    return _pext_u64(c[0], M) | (_pext_u64(c[1], M) << 16) | (_pext_u64(c[2], M) << 32) | (_pext_u64(c[3], M) << 48);
}
const bool have_bmi2 = __builtin_cpu_supports("bmi2") && !getenv("MTG_NO_VEC");
#endif
inline uint64_t pack32(const char* b)
{
#if defined(__x86_64__)
    if (have_bmi2) return pack32_bmi2(b);
#endif
    uint64_t w = 0;
    for (int i = 0; i < 32; i++) w |= (uint64_t)nt_code((unsigned char)b[i]) << (2 * i);
    return w;
}
/* codes of s[0, n) into out[0, n / 32] (LSB first, a last partial word zero-padded) */
inline void pack_lsb(const char* s, size_t n, uint64_t* out)
{
    size_t i = 0, w = 0;
    for (; i + 32 <= n; i += 32) out[w++] = pack32(s + i);
    if (i < n) {
        char b[32] = {0};
        memcpy(b, s + i, n - i);
        const size_t r = n - i;
        out[w] = pack32(b) & (r < 32 ? (1ull << (2 * r)) - 1 : ~0ull);
    }
}
inline bool all_upper_acgt(const char* s, size_t n)
{
    /* 'A' 0x41, 'C' 0x43, 'G' 0x47, 'T' 0x54: a table of the 256 codes */
    static const struct Tab { bool ok[256]; Tab() { memset(ok, 0, sizeof ok); ok['A'] = ok['C'] = ok['G'] = ok['T'] = true; } } tab;
    bool ok = true;
    for (size_t i = 0; i < n; i++) ok &= tab.ok[(unsigned char)s[i]];
    return ok;
}
} // namespace

void FillInput::set_common(size_t g, std::string_view source, std::string_view swf_target, int nb_mis)
{
    const uint64_t kfields = kmask(k);
    char sb[32] = {0};
    memcpy(sb, source.data(), (size_t)k); /* k <= 31 characters; the caller has checked that the source has them */
    const uint64_t sp = pack32(sb) & kfields;
    src[g] = reverse_fields(sp) >> (64 - 2 * k); /* encode_kmer: first character in the highest field */
    const size_t rl = swf_target.size(), w0 = roff[g];
    const size_t nw = (rl + 31) / 32 + 1;
    rwords[w0 + nw - 1] = 0; /* the block is recycled, not zeroed */
    if (nw >= 2) rwords[w0 + nw - 2] = 0;
    pack_lsb(swf_target.data(), rl, rwords.p + w0);
    r0[g] = rl >= (size_t)k ? reverse_fields(rwords[w0] & kfields) >> (64 - 2 * k) : 0;
    /* the early stop is a literal strstr in upper-case contigs (IterativeExtensions [MEM]): a pattern with any other character never matches */
    if (!all_upper_acgt(swf_target.data(), rl)) { rlen[g] = 0xFFFFFFFFu; r0[g] = 0; }
    nbmis[g] = (uint8_t)nb_mis;
    bool ok = (int)source.size() == k;
    if (ok) { /* nt_bad: bit 3 of the ASCII code, in any of the k characters */
        uint64_t c[4];
        memcpy(c, sb, 32);
        ok = ((c[0] | c[1] | c[2] | c[3]) & 0x0808080808080808ull) == 0;
    }
    fast_ok[g] = ok ? 1 : 0;
}
void FillInput::set_target(size_t o, std::string_view seq)
{
    /* the first k characters as they are; the device encodes them (encode_target, mtg_post.h) */
    uint8_t* slot = traw.p + o * TARGET_SLOT;
    const bool usable = (int)seq.size() >= k;
    memset(slot, 0, TARGET_SLOT);
    if (usable) memcpy(slot, seq.data(), (size_t)k);
    slot[TARGET_SLOT - 1] = usable ? 1 : 0;
}
void FillInput::set(size_t g, std::string_view source, std::string_view swf_target, const TargetSpan* targets, int nb_mis)
{
    set_common(g, source, swf_target, nb_mis);
    if (targets) {
        size_t o = toff[g];
        for (const Target& t : *targets) set_target(o++, t.seq);
    }
}

struct ContigGraph {
    std::vector<std::vector<int>> in_edges; /* ascending, unique (std::set order of src/GraphAnalysis.cpp:110) */
    ContigGraph(const GapDev& gc, int k)
    {
        const uint32_t n = gc.o.n_contigs;
        in_edges.resize(n);
        const uint64_t mk1 = kmask(k - 1);
        auto kmer_at = [&](uint32_t c, uint32_t start) {
            const uint64_t* w = gc.words + gc.word_start[c];
            uint64_t f = 0;
            for (int j = 0; j < k - 1; j++) { uint32_t i = start + j; f = (f << 2) | ((w[i >> 5] >> (2 * (i & 31))) & 3ull); }
            return f & mk1;
        };
        std::unordered_map<uint64_t, std::vector<int>> by_prefix;
        for (uint32_t j = 0; j < n; j++) by_prefix[kmer_at(j, 0)].push_back((int)j);
        for (uint32_t i = 0; i < n; i++) {
            auto it = by_prefix.find(kmer_at(i, gc.len[i] - (k - 1)));
            if (it == by_prefix.end()) continue;
            for (int j : it->second) {
                if (j == (int)i && (int)gc.len[i] == k - 1) continue; /* src/IGraphOutput.cpp:160 */
                in_edges[j].push_back((int)i);
            }
        }
        for (auto& v : in_edges) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); }
    }
};

typedef std::vector<int> Path;
typedef std::set<std::pair<Path, int>> PathSet; /* (path, target index); see order note in process_gap */

struct RevDfs { /* GraphAnalysis::find_all_paths_rev, src/GraphAnalysis.cpp:244-326 */
    const ContigGraph& g;
    const std::vector<TermInfo>& terms;
    int terminal_node, target;
    int nb_calls = 0;
    bool success = true;
    static const size_t max_breadth = 20; /* src/GraphAnalysis.hpp:43 */
    std::set<Path> run(int start_node, const Path& current)
    {
        std::set<Path> paths;
        if (nb_calls++ > 10000000) { success = false; return paths; }
        if (start_node != terminal_node)
            for (auto& t : terms) if (t.node == start_node) return paths;
        if (start_node == 0) { paths.insert(current); return paths; }
        for (int next : g.in_edges[start_node]) {
            if (std::find(current.begin(), current.end(), next) == current.end()) {
                Path ext;
                ext.reserve(current.size() + 1);
                ext.push_back(next);
                ext.insert(ext.end(), current.begin(), current.end());
                std::set<Path> sub = run(next, ext);
                paths.insert(sub.begin(), sub.end());
                if (paths.size() >= max_breadth) success = false;
            }
            if (!success) return paths;
        }
        return paths;
    }
};

static int compute_qual(const Solution& s, bool repeated) /* src/Utils.hpp:85-103 */
{
    int q = 50;
    if (repeated) q = 25;
    if (s.count > 1) q = 15;
    if (s.nb_errors == 1) q = 10;
    if (s.nb_errors == 2) q = 5;
    return q;
}

/* ASCII of the packed nucleotides [from, from + L) of `words` (2 bits each, nucleotide i at bits 2(i mod 32) of word i / 32), four per
 * table lookup; reversed and complemented when rc is set.  Reads at most one word past the last one used (the chunk storage is padded). */
struct DecodeLut {
    uint32_t fwd[256], rc[256];
    DecodeLut()
    {
        static const char NT[4] = {'A', 'C', 'T', 'G'}, NTC[4] = {'T', 'G', 'A', 'C'};
        for (int b = 0; b < 256; b++) {
            char f[4], r[4];
            for (int j = 0; j < 4; j++) { f[j] = NT[(b >> (2 * j)) & 3]; r[3 - j] = NTC[(b >> (2 * j)) & 3]; }
            memcpy(&fwd[b], f, 4);
            memcpy(&rc[b], r, 4);
        }
    }
};
#if defined(__x86_64__)
/* 32 nucleotides per step: the 2-bit codes are spread to one per byte (pdep) and looked up 32 at a time (vpshufb).  Returns how many
 * nucleotides it wrote (a multiple of 32); the caller finishes the rest. */
__attribute__((target("avx2,bmi2"))) static uint32_t decode_slice_avx2(const uint64_t* words, uint32_t from, uint32_t L, bool rc, char* dst)
{
    const uint64_t SPREAD = 0x0303030303030303ull;
    const __m256i fwd = _mm256_setr_epi8('A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 'A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i cmp = _mm256_setr_epi8('T', 'G', 'A', 'C', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 'T', 'G', 'A', 'C', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i rev = _mm256_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
    uint32_t i = 0;
    for (; i + 32 <= L; i += 32) {
        const uint32_t j = from + i, sh = 2 * (j & 31);
        uint64_t w = words[j >> 5] >> sh;
        if (sh) w |= words[(j >> 5) + 1] << (64 - sh);
        const __m256i codes = _mm256_setr_epi64x((long long)_pdep_u64(w, SPREAD), (long long)_pdep_u64(w >> 16, SPREAD), (long long)_pdep_u64(w >> 32, SPREAD),
                                                 (long long)_pdep_u64(w >> 48, SPREAD));
        if (!rc) _mm256_storeu_si256((__m256i*)(dst + i), _mm256_shuffle_epi8(fwd, codes));
        else {
            const __m256i c = _mm256_shuffle_epi8(_mm256_shuffle_epi8(cmp, codes), rev); /* complemented, each half reversed */
            _mm256_storeu_si256((__m256i*)(dst + (L - 32 - i)), _mm256_permute2x128_si256(c, c, 1));
        }
    }
    return i;
}
#endif
static void decode_slice(const uint64_t* words, uint32_t from, uint32_t L, bool rc, char* dst)
{
    static const DecodeLut lut;
    static const char NT[4] = {'A', 'C', 'T', 'G'}, NTC[4] = {'T', 'G', 'A', 'C'};
    uint32_t i = 0;
#if defined(__x86_64__)
    static const bool vec = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2") && !getenv("MTG_NO_VEC");
    if (vec) i = decode_slice_avx2(words, from, L, rc, dst);
#endif
    for (; i + 4 <= L; i += 4) {
        const uint32_t j = from + i, sh = 2 * (j & 31);
        uint64_t w = words[j >> 5] >> sh;
        if (sh > 56) w |= words[(j >> 5) + 1] << (64 - sh);
        if (!rc) memcpy(dst + i, &lut.fwd[w & 0xFF], 4);
        else memcpy(dst + (L - 4 - i), &lut.rc[w & 0xFF], 4);
    }
    for (; i < L; i++) {
        const uint32_t j = from + i, c = (uint32_t)(words[j >> 5] >> (2 * (j & 31))) & 3;
        if (!rc) dst[i] = NT[c]; else dst[L - 1 - i] = NTC[c];
    }
}

/* a gap whose solutions come out of the contig graph: candidate sequences per target (in the order the reference visits them), waiting
 * for the alignments of remove_almost_identical_solutions, which run on the device for the whole batch */
struct GenWork {
    std::vector<std::vector<Solution>> groups;
    std::vector<std::vector<int64_t>> pair; /* per group, n x n: index into the batch's NW pairs for (row j, column i < j); -2: equal strings */
};

/* everything after the device kernels for one gapFillFromSource call, except the de-duplication and the coverage numbers of the general
 * path (returns the candidates of such a gap, nullptr otherwise) */
static GenWork* process_gap(const GapDev& gc, GapWork& W, int k, char* arena_slot)
{
    W.nb_nodes = (int)gc.o.n_contigs;
    W.total_nt = (int)gc.o.total_nt;
    W.nb_terminal = (int)gc.p.nb_terminal;
    if (gc.p.nb_terminal == 0) { /* get_first_contig, src/Filler.cpp:1381-1407 */
        W.extension.clear();
        if (gc.o.n_contigs > 0 && (int)gc.p.clen0 > k) W.extension = gc.contig0_slice((uint32_t)k, gc.p.clen0);
        return nullptr;
    }
    if (gc.p.fast == 2) { W.has_counts = W.reverse; return nullptr; } /* target at the very start of contig 0: empty fill */
    if (gc.p.fast == 1) {
        /* terminal node 0: find_all_paths_rev returns the single path [0] (src/GraphAnalysis.cpp:222-226) and
         * paths_to_sequences keeps contig0[k:pos] (:386-423); coverage was computed on the device */
        Solution& s = W.sols.emplace_first();
        /* written straight into the batch arena (reverse-complemented when the attempt is a reverse one, src/Filler.cpp:998-1001) */
        const uint32_t L = gc.p.pos - (uint32_t)k;
        decode_slice(gc.words, (uint32_t)k, L, W.reverse, arena_slot);
        arena_slot[L] = 0;
        s.seq.view(arena_slot, L);
        s.nb_errors = (int)gc.p.errors;
        s.target = (int)gc.p.target;
        s.count = 1;
        s.rank = 1;
        const uint64_t sum = gc.p.ab_sum;
        s.avg = sum / (float)gc.p.ab_n;
        s.median = (float)((gc.p.ab_n % 2 == 1) ? (double)gc.p.med_hi : 0.5 * (gc.p.med_hi + gc.p.med_lo));
        s.ab_n = 0; /* no host-side coverage query needed */
        s.qual = compute_qual(s, W.anchor_repeated);
        W.nb_total_filled = 1;
        W.has_counts = true;
        return nullptr;
    }
    GenWork* gw = nullptr;
    std::vector<TermInfo> terms;
    for (uint32_t c = 0; c < gc.o.n_contigs; c++)
        if (gc.tpos[c] != 0xFFFFFFFFu) terms.push_back(TermInfo{(int)c, (int)gc.tpos[c], (int)gc.terr[c], (int)gc.ttgt[c]});
    ContigGraph graph(gc, k);
    /* find_all_paths_rev wrapper, src/GraphAnalysis.cpp:205-237.  The reference keeps set<pair<path, bkpt_t>>; paths
     * reaching different targets end in different nodes, so ordering by (path, target index) gives the same sequence. */
    std::vector<std::pair<Path, int>> paths;
    if (gc.paths && gc.paths[0] == 0) {
        /* enumerated on the device (k_paths); the reference keeps them in a set<pair<path, bkpt_t>>: same order after sorting */
        const uint32_t* q = gc.paths + 2;
        for (uint32_t i = 0; i < gc.paths[1]; i++) {
            const int target = (int)q[0];
            const uint32_t len = q[1];
            paths.push_back({Path(q + 2, q + 2 + len), target});
            q += 2 + len;
        }
        std::sort(paths.begin(), paths.end());
        paths.erase(std::unique(paths.begin(), paths.end()), paths.end());
    } else if (terms[0].node == 0) {
        paths.push_back({Path{0}, terms[0].target});
    } else {
        PathSet all;
        for (auto& t : terms) {
            RevDfs dfs{graph, terms, t.node, t.target};
            std::set<Path> ps = dfs.run(t.node, Path{t.node});
            for (auto& p : ps) all.insert({p, t.target});
        }
        paths.assign(all.begin(), all.end());
    }
    /* group by target name: unordered_map<string, set<path>> iterated in libstdc++ order (src/Filler.cpp:924-936) */
    std::unordered_map<std::string, std::set<Path>> paths_to_compare;
    for (auto& pr : paths) {
        std::string key(W.targets[pr.second].name);
        if (W.targets[pr.second].is_rc) key += "_Rc";
        paths_to_compare[key].insert(pr.first);
    }
    std::vector<std::string> node_seq(gc.o.n_contigs);
    std::vector<char> have(gc.o.n_contigs, 0);
    auto node = [&](int i) -> const std::string& { if (!have[i]) { node_seq[i] = gc.contig(i); have[i] = 1; } return node_seq[i]; };
    const size_t K = (size_t)k;
    for (auto it = paths_to_compare.begin(); it != paths_to_compare.end(); ++it) {
        /* paths_to_sequences, src/GraphAnalysis.cpp:331-460 */
        std::vector<Solution> tmp;
        int errs = 0, tgt = -1;
        for (const Path& p : it->second) {
            std::string sequence;
            for (size_t ip = 0; ip < p.size(); ip++) {
                const std::string& ns = node(p[ip]);
                if (ip + 1 == p.size()) {
                    int pos_anchor = 0;
                    for (auto& t : terms) if (t.node == p[ip]) { pos_anchor = t.pos; errs = t.errors; tgt = t.target; break; }
                    if ((size_t)pos_anchor <= K - 1) {
                        sequence = sequence.substr(0, sequence.length() - ((K - 1) - (size_t)pos_anchor)); /* size_t wrap-around as in :406 */
                    } else {
                        const size_t from = ip != 0 ? K - 1 : K;
                        sequence.append(ns, from, (size_t)pos_anchor - from);
                    }
                    break;
                }
                sequence.append(ns, ip != 0 ? K - 1 : K, std::string::npos);
            }
            if (!sequence.empty()) { Solution s; s.seq = std::move(sequence); s.nb_errors = errs; s.target = tgt; tmp.push_back(std::move(s)); }
        }
        W.nb_total_filled += (int)tmp.size();
        if (!tmp.empty()) { if (!gw) gw = new GenWork(); gw->groups.push_back(std::move(tmp)); }
    }
    W.has_counts = (W.nb_total_filled > 0) || W.reverse; /* src/Filler.cpp:1012 */
    return gw;
}

/* remove_almost_identical_solutions(.., 90), src/Utils.cpp:208-238, for every group of a gap, with the match counts of the batch's
 * alignments; then the solutions in their final order */
static void finish_general(GapWork& W, GenWork& gw, const std::vector<uint32_t>& matches)
{
    for (size_t g = 0; g < gw.groups.size(); g++) {
        std::vector<Solution>& tmp = gw.groups[g];
        const size_t n = tmp.size();
        if (n > 1) {
            std::vector<Solution> fin;
            std::vector<size_t> fin_src; /* whose sequence a kept solution currently holds */
            fin.push_back(tmp[0]);
            fin_src.push_back(0);
            for (size_t j = 0; j < n; j++) {
                const Solution& a = tmp[j];
                bool similar = false;
                for (size_t f = 0; f < fin.size(); f++) {
                    Solution& b = fin[f];
                    const size_t i = fin_src[f];
                    bool same = (i == j);
                    if (!same) {
                        const int64_t pi = gw.pair[g][j * n + i];
                        if (pi == -2) same = true;
                        else {
                            float identity = (float)matches[(size_t)pi];
                            identity /= std::max((int)a.seq.size(), (int)b.seq.size());
                            same = identity * 100 >= 90;
                        }
                    }
                    if (same) {
                        if (a.nb_errors < b.nb_errors) { b.seq = a.seq; b.nb_errors = a.nb_errors; fin_src[f] = j; }
                        similar = true;
                        break;
                    }
                }
                if (!similar) { fin.push_back(a); fin_src.push_back(j); }
            }
            tmp.swap(fin);
        }
        int rank = 1;
        for (auto& s : tmp) { s.count = (int)tmp.size(); s.rank = rank++; s.ab_n = 1; W.sols.push_back(std::move(s)); }
    }
}

static std::string revcomp_str(const std::string& s) /* revcomp_sequence, src/Utils.cpp:44-77: other characters are dropped */
{
    std::string r;
    r.reserve(s.size());
    for (auto it = s.rbegin(); it != s.rend(); ++it) {
        switch (*it) {
            case 'a': r += 't'; break; case 't': r += 'a'; break; case 'c': r += 'g'; break; case 'g': r += 'c'; break;
            case 'A': r += 'T'; break; case 'T': r += 'A'; break; case 'C': r += 'G'; break; case 'G': r += 'C'; break;
        }
    }
    return r;
}

static double median_of(std::vector<unsigned int>& v) /* src/Utils.cpp:241-254 */
{
    size_t n = v.size() / 2;
    std::nth_element(v.begin(), v.begin() + n, v.end());
    unsigned int vn = v[n];
    if (v.size() % 2 == 1) return vn;
    std::nth_element(v.begin(), v.begin() + n - 1, v.end());
    return 0.5 * (vn + v[n - 1]);
}


/* a batch whose GapWork records already exist (the CLI drivers) */
struct VecSource : BatchSource {
    std::vector<GapWork>& g;
    const std::vector<std::string_view>& swf;
    VecSource(std::vector<GapWork>& g_, const std::vector<std::string_view>& s_) : g(g_), swf(s_) {}
    size_t count() const override { return g.size(); }
    bool sizes(size_t i, size_t& src_len, size_t& swf_len, size_t& n_targets) const override
    {
        src_len = g[i].source.size(); swf_len = swf[i].size(); n_targets = g[i].targets.size();
        return true;
    }
    void input(size_t i, FillInput& in, int nb_mis_allowed) const override
    {
        in.set_common(i, g[i].source, swf[i], g[i].anchor_repeated ? 0 : nb_mis_allowed); /* src/Filler.cpp:859-863 */
        size_t o = in.toff[i];
        for (const Target& t : g[i].targets) in.set_target(o++, t.seq);
    }
    void marshal(const FillInput&, int) override {}
    std::vector<GapWork>& gaps() override { return g; }
};
int fill_gaps(const mtg_index* idx, const mtg_params* p, std::vector<GapWork>& gaps, const std::vector<std::string_view>& swf_targets, FillArena& arena,
              mtg_batch_stats* stats_out)
{
    VecSource src(gaps, swf_targets);
    return fill_gaps(idx, p, src, arena, stats_out, nullptr);
}

/* runs a batch of gapFillFromSource calls */
int fill_gaps(const mtg_index* idx, const mtg_params* p, BatchSource& src, FillArena& arena, mtg_batch_stats* stats_out, std::vector<uint64_t>* sol_blocks)
{
    const int k = idx->dev.k;
    const size_t n = src.count();
    const int nth = p->nb_host_threads;
    const double t_begin = now_ms();
    static const bool dbg = getenv("MTG_DEBUG_TIMERS") != nullptr;
    double tk = t_begin;
    auto tick = [&](const char* what) { if (dbg) { const double t = now_ms(); fprintf(stderr, "  [fill_gaps] %-22s %.2f ms\n", what, t - tk); tk = t; } };
    /* one workspace of the index (staging blocks, device buffers, streams) holds this batch from here until its results have been
     * taken out of it */
    WorkspaceLock batch_lock = acquire_workspace(idx);
    FillInput in;
    in.k = k;
    in.ws = batch_lock.ws;
    std::atomic<long> bad_gap{-1}, short_gap{-1};
    const auto sizes_of = [&](size_t i, size_t& swf_len, size_t& n_targets) -> bool {
        size_t src_len = 0;
        bool ok = true;
        if (!src.sizes(i, src_len, swf_len, n_targets)) { bad_gap = (long)i; src_len = swf_len = n_targets = 0; ok = false; }
        else if ((int)src_len < k) { short_gap = (long)i; ok = false; }
        in.slen[i] = (uint32_t)src_len;
        return ok;
    };
    const auto input_of = [&](size_t i) { src.input(i, in, p->nb_mis_allowed); };
    const bool one_pass = in.plan_and_fill(n, nth, sizes_of, input_of);
    if (one_pass) tick("input (one pass)");
    else {
        /* the first batch of its shape on this workspace (the staging blocks have to grow), or a malformed gap */
        bad_gap = -1; short_gap = -1;
        in.plan(n, nth, sizes_of);
        if (bad_gap >= 0) { set_error("gap %ld: null field", bad_gap.load()); return MTG_ERR_ARG; }
        if (short_gap >= 0) { set_error("gap %ld: source sequence shorter than k", short_gap.load()); return MTG_ERR_ARG; }
        tick("input sizes + layout");
        in.fill(nth, input_of);
        tick("input set");
    }
    mtg_batch_stats st{};
    st.host_ms = now_ms() - t_begin;
    DevBatch batch;
    double t_marshal = 0, t_parts = 0;
    bool marshalled = false;
    const std::function<void()> while_busy = [&]() { const double t = now_ms(); src.marshal(in, nth); marshalled = true; t_marshal = now_ms() - t; };
    /* Every chunk of results is turned into solutions as soon as it is back, while the device works on the next one: arena bytes of
     * every block of its gaps, then the gaps of a block one after the other (the sequences of a chunk share one arena buffer). */
    const size_t B = RESULT_BLOCK, nb = (n + B - 1) / B;
    std::vector<std::atomic<uint64_t>> blk_sols(nb);
    for (auto& c : blk_sols) c.store(0, std::memory_order_relaxed);
    std::vector<GenWork*> genw(n, nullptr); /* gaps whose solutions come out of the host's path enumeration */
    struct GenGuard { std::vector<GenWork*>& v; ~GenGuard() { for (GenWork* g : v) delete g; } } gen_guard{genw};
    const std::function<void(size_t, const uint32_t*, size_t, size_t)> on_ready = [&](size_t chunk, const uint32_t* ids, size_t first, size_t count) {
        const double t = now_ms();
        if (!marshalled) { src.marshal(in, nth); marshalled = true; } /* the gap records have to exist by now */
        std::vector<GapWork>& gaps = src.gaps();
        const size_t nbp = (count + B - 1) / B;
        auto gap_of = [&](size_t j) -> size_t { return ids ? ids[j] : first + j; };
        /* one pass: a block of gaps adds up the bytes its sequences need, learns where the blocks before it end (they were handed out
         * in order, and each publishes its end as soon as it knows its own size) and writes its gaps one after the other; the chunk
         * never needs more than 32 bytes per dense word + one per gap */
        const HostChunk& hc = *batch.chunks[chunk];
        static const bool prefetch_words = !getenv("MTG_NO_PREFETCH");
        const size_t arena_cap = (size_t)hc.n_words * 32 + count + 64;
        bool external = false;
        if (arena.ext_ok) {
            /* the caller's buffer takes the chunk if it continues the gap order and fits */
            if (!ids && first == arena.ext_next_gap && arena.ext_used + arena_cap <= arena.ext_cap) external = true;
            else arena.ext_ok = false;
        }
        char* const arena_base = external ? arena.ext + arena.ext_used : arena.ensure(chunk, arena_cap);
        std::vector<std::atomic<int64_t>> ends(nbp + 1);
        for (auto& e : ends) e.store(-1, std::memory_order_relaxed);
        ends[0].store(0, std::memory_order_release);
        std::atomic<bool> incomplete{false};
        std::atomic<uint64_t> rec_bytes{0}, rec_filled{0};
        const bool recording = !ids;
        parallel_for(nbp, nth, [&](size_t b) {
            uint64_t need = 0, rb = 0, rf = 0;
            /* slot j of the chunk holds gap gap_of(j) */
            for (size_t j = b * B; j < std::min(count, (b + 1) * B); j++) {
                const SlotRec& r = hc.recs[j];
                need += (r.o.status == GAP_OK && r.p.fast == 1) ? (uint64_t)(r.p.pos - (uint32_t)k) + 1 : 0;
            }
            int64_t begin;
            while ((begin = ends[b].load(std::memory_order_acquire)) < 0) Pool::cpu_relax();
            ends[b + 1].store(begin + (int64_t)need, std::memory_order_release);
            uint64_t off = (uint64_t)begin, nsol = 0;
            size_t cur_blk = ~(size_t)0;
            bool odd = false;
            for (size_t j = b * B; j < std::min(count, (b + 1) * B); j++) {
                const size_t i = gap_of(j);
                if (prefetch_words && j + 8 < count) {
                    /* the gaps reserved their room in the dense words in no particular order, and the copy from the device left them in
                     * memory, not in a cache: ask for the words of a gap a few places ahead */
                    const char* pw = (const char*)(hc.words + hc.recs[j + 8].wbase);
                    __builtin_prefetch(pw); __builtin_prefetch(pw + 64); __builtin_prefetch(pw + 128);
                }
                if (hc.recs[j].o.status != GAP_OK) { odd = true; continue; } /* re-run in a larger tier: comes back with a later chunk */
                const GapDev gd = DevBatch::view(hc, j);
                if (i / B != cur_blk) { if (nsol) blk_sols[cur_blk].fetch_add(nsol, std::memory_order_relaxed); nsol = 0; cur_blk = i / B; }
                src.init_gap(i);
                if (gd.p.fast == 0 && gd.p.nb_terminal > 0) src.need_targets(i);
                genw[i] = process_gap(gd, gaps[i], k, arena_base + off);
                if (genw[i]) odd = true;
                else if (recording) src.record_gap(i, rb, rf);
                off += gd.p.fast == 1 ? (uint64_t)(gd.p.pos - (uint32_t)k) + 1 : 0;
                nsol += gaps[i].sols.size();
            }
            if (nsol) blk_sols[cur_blk].fetch_add(nsol, std::memory_order_relaxed);
            if (odd) incomplete = true;
            if (rb | rf) { rec_bytes.fetch_add(rb, std::memory_order_relaxed); rec_filled.fetch_add(rf, std::memory_order_relaxed); }
        }, 1);
        if (external) {
            arena.ext_used += (size_t)ends[nbp].load(std::memory_order_acquire);
            arena.ext_next_gap = first + count;
            if (incomplete.load()) arena.ext_ok = false; /* a gap of this chunk gets its sequences later, out of order */
        }
        if (recording) src.part_done(first, count, !incomplete.load(), rec_bytes.load(), rec_filled.load());
        t_parts += now_ms() - t;
    };
    int rc = device_run(idx, p, in, batch, &st, &while_busy, &on_ready);
    if (rc) return rc;
    if (dbg) fprintf(stderr, "  [fill_gaps] marshal (overlapped) %.2f ms, chunks processed in %.2f ms\n", t_marshal, t_parts);
    std::vector<GapWork>& gaps = src.gaps();
    st.host_ms += t_parts;
    tk = now_ms();
    double t0 = now_ms();
    std::vector<size_t> gen_idx;
    for (size_t i = 0; i < n; i++) if (genw[i]) gen_idx.push_back(i);
    if (!gen_idx.empty()) {
        /* every alignment remove_almost_identical_solutions can ask for: a later candidate (rows) against an earlier one (columns) */
        std::vector<NwPair> pairs;
        for (size_t gi : gen_idx) {
            GenWork& gw = *genw[gi];
            gw.pair.resize(gw.groups.size());
            for (size_t g = 0; g < gw.groups.size(); g++) {
                const std::vector<Solution>& tmp = gw.groups[g];
                const size_t m = tmp.size();
                if (m < 2) continue;
                gw.pair[g].assign(m * m, -1);
                for (size_t j = 1; j < m; j++)
                    for (size_t i = 0; i < j; i++) {
                        if (tmp[j].seq == tmp[i].seq) { gw.pair[g][j * m + i] = -2; continue; }
                        gw.pair[g][j * m + i] = (int64_t)pairs.size();
                        pairs.push_back(NwPair{tmp[j].seq.data(), (uint32_t)tmp[j].seq.size(), tmp[i].seq.data(), (uint32_t)tmp[i].seq.size()});
                    }
            }
        }
        std::vector<uint32_t> matches;
        if (!pairs.empty()) { rc = nw_run(idx, pairs, matches); if (rc) return rc; }
        parallel_for(gen_idx.size(), nth, [&](size_t ii) { finish_general(gaps[gen_idx[ii]], *genw[gen_idx[ii]], matches); }, 1);
        for (size_t gi : gen_idx) blk_sols[gi / B].fetch_add(gaps[gi].sols.size(), std::memory_order_relaxed); /* they had none when their block was counted */
        tick("alignments");
    }
    if (sol_blocks) { sol_blocks->resize(nb); for (size_t b = 0; b < nb; b++) (*sol_blocks)[b] = blk_sols[b].load(std::memory_order_relaxed); }
    /* coverage of the general-path solutions: abundance of every k-mer of source + seq (src/Filler.cpp:959-988), one batched device query */
    std::vector<uint64_t> q;
    const uint64_t mk = kmask(k);
    for (size_t gi : gen_idx) {
        GapWork& g = gaps[gi];
        for (auto& s : g.sols) {
            s.ab_off = q.size();
            uint64_t f = 0;
            int valid = 0;
            auto feed = [&](const char* str, size_t len) {
                for (size_t ci = 0; ci < len; ci++) {
                    const unsigned char c = (unsigned char)str[ci];
                    if (nt_bad(c)) { valid = 0; f = 0; continue; }
                    f = ((f << 2) | nt_code(c)) & mk;
                    if (++valid >= k) q.push_back(f);
                }
            };
            feed(g.source.data(), g.source.size());
            feed(s.seq.data(), s.seq.size());
            s.ab_n = q.size() - s.ab_off;
        }
    }
    st.host_ms += now_ms() - t0;
    std::vector<uint32_t> ab(q.size());
    if (!q.empty()) {
        rc = query_run(idx, q.data(), q.size(), ab.data(), nullptr, nullptr);
        if (rc) return rc;
    }
    t0 = now_ms();
    parallel_for(gen_idx.size(), p->nb_host_threads, [&](size_t ii) {
        GapWork& g = gaps[gen_idx[ii]];
        for (auto& s : g.sols) {
            std::vector<unsigned int> v(ab.begin() + s.ab_off, ab.begin() + s.ab_off + s.ab_n);
            uint64_t sum = 0;
            for (size_t j = 0; j < v.size(); j++) {
                if (v[j] == 0) {
                    uint64_t c = q[s.ab_off + j], r = revcomp(c, k);
                    c = c < r ? c : r;
                    std::string d(k, 'A');
                    for (int t = 0; t < k; t++) d[t] = "ACTG"[(c >> (2 * (k - 1 - t))) & 3];
                    fprintf(stderr, "WARNING Unknown kmer : %s\n", d.c_str());
                }
                sum += v[j];
            }
            s.avg = sum / (float)v.size();
            s.median = v.empty() ? 0.f : (float)median_of(v);
            s.qual = compute_qual(s, g.anchor_repeated);
            if (g.reverse && !s.seq.is_view()) s.seq = revcomp_str(s.seq.str()); /* views were written reverse-complemented */
        }
    }, 1);
    st.host_ms += now_ms() - t0;
    tick("general path");
    st.total_ms = now_ms() - t_begin;
    if (stats_out) *stats_out = st;
    stats_store(st);
    return MTG_OK;
}

} // namespace mtgi

/* ------------------------------------------------------------------------------------------------ C ABI (host side) */
struct mtg_results {
    mtgi::FillArena arena;
    std::vector<mtgi::Target> targets; /* flat storage of every gap's dictionary */
    std::vector<mtg_filled> filled_flat;
    std::vector<mtgi::GapWork> gaps;
    std::vector<mtg_gap_result> res;
    int nthreads = 0; /* host threads of the batch that filled it */
    /* kept up to date by the record writer: what mtg_results_summary reports */
    std::vector<uint32_t> nfilled;
    std::atomic<uint64_t> sum_bytes{0}, sum_filled{0};
    bool summary_ready = false;
};
struct mtg_contigs {
    std::vector<std::vector<std::string>> c;
};

/* Result objects are recycled: a freed one keeps its storage (a few hundred bytes per gap plus the sequence arena) for the next batch,
 * which then pays neither page faults nor allocator traffic.  At most six are kept (three batches in flight, each with
 * the previous result still in its caller's hands). */
namespace {
std::mutex g_results_mtx;
std::vector<mtg_results*> g_results_cache;
mtg_results* results_acquire()
{
    {
        std::lock_guard<std::mutex> lk(g_results_mtx);
        if (!g_results_cache.empty()) { mtg_results* r = g_results_cache.back(); g_results_cache.pop_back(); return r; }
    }
    return new mtg_results();
}
void results_release(mtg_results* r)
{
    {
        std::lock_guard<std::mutex> lk(g_results_mtx);
        if (g_results_cache.size() < 6) { g_results_cache.push_back(r); return; }
    }
    delete r;
}
} // namespace


extern "C" {

void mtg_default_params(mtg_params* p)
{
    p->max_nodes = 100;
    p->max_depth = 10000;
    p->nb_mis_allowed = 2;
    p->end_rule_nonbranching = 0;
    p->nb_host_threads = 0;
}

int mtg_index_create_from_reads(const char* paths_csv, int k, int abundance_min, int abundance_max, mtg_index** out)
{
    return mtgi::index_from_reads(paths_csv, k, abundance_min, abundance_max, out);
}
int mtg_index_save(const mtg_index* idx, const char* path) { return mtgi::index_save(idx, path); }
int mtg_index_load(const char* path, mtg_index** out) { return mtgi::index_load(path, out); }

/* a batch handed over through the C ABI: the GapWork records are written while the device is busy */
namespace {
struct AbiSource : mtgi::BatchSource {
    const mtg_gap* g;
    size_t n;
    mtg_results* R;
    AbiSource(const mtg_gap* g_, size_t n_, mtg_results* R_) : g(g_), n(n_), R(R_) {}
    size_t count() const override { return n; }
    /* the strings of a batch are wherever the caller has them: the two passes ask for those of the gaps a few places ahead early */
    void prefetch(size_t i) const
    {
        static const bool on = !getenv("MTG_NO_PREFETCH");
        if (!on) return;
        if (i + 16 < n) { const mtg_gap& b = g[i + 16]; __builtin_prefetch(b.target_seqs); __builtin_prefetch(b.target_names); }
        if (i + 8 < n) {
            const mtg_gap& b = g[i + 8];
            __builtin_prefetch(b.source);
            __builtin_prefetch(b.target);
            if (b.n_targets > 0 && b.target_seqs && b.target_names) { __builtin_prefetch(b.target_seqs[0]); __builtin_prefetch(b.target_names[0]); }
        }
    }
    bool sizes(size_t i, size_t& src_len, size_t& swf_len, size_t& n_targets) const override
    {
        prefetch(i);
        const mtg_gap& a = g[i];
        if (!a.source || !a.target || (a.n_targets > 0 && (!a.target_seqs || !a.target_names))) return false;
        for (int t = 0; t < a.n_targets; t++) if (!a.target_seqs[t] || !a.target_names[t]) return false;
        src_len = strlen(a.source); swf_len = strlen(a.target); n_targets = (size_t)std::max(a.n_targets, 0);
        return true;
    }
    void input(size_t i, mtgi::FillInput& in, int nb_mis_allowed) const override
    {
        prefetch(i);
        const mtg_gap& a = g[i];
        in.set_common(i, std::string_view(a.source, in.slen[i]), std::string_view(a.target, in.rlen[i]), a.is_anchor_repeated ? 0 : nb_mis_allowed); /* src/Filler.cpp:859-863 */
        /* of a target only the first k characters matter, and whether it has them */
        for (int t = 0; t < a.n_targets; t++) in.set_target(in.toff[i] + (size_t)t, std::string_view(a.target_seqs[t], strnlen(a.target_seqs[t], (size_t)in.k)));
    }
    const mtgi::FillInput* in_ = nullptr;
    void marshal(const mtgi::FillInput& in, int nthreads) override
    {
        in_ = &in;
        R->gaps.resize(n);
        R->res.resize(n);
        R->nfilled.resize(n);
        if (R->filled_flat.size() < n) R->filled_flat.resize(n);
        R->targets.resize(in.traw.size() / mtg::TARGET_SLOT);
        (void)nthreads; /* the per-gap part happens in init_gap, when the gap's results are there: one visit of the record, not two */
    }
    void init_gap(size_t i) override
    {
        const mtg_gap& a = g[i];
        mtgi::GapWork& w = R->gaps[i];
        w.reset(); /* a recycled object still holds the previous batch */
        w.source = std::string_view(a.source, in_->slen[i]);
        w.anchor_repeated = a.is_anchor_repeated != 0;
        w.reverse = a.reverse != 0;
    }
    /* the dictionary of a gap (names, strands) is only read on the multi-contig path: built there, for the gaps that take it */
    void need_targets(size_t i) override
    {
        const mtg_gap& a = g[i];
        mtgi::GapWork& w = R->gaps[i];
        mtgi::Target* T0 = R->targets.data() + in_->toff[i];
        for (int t = 0; t < a.n_targets; t++) {
            mtgi::Target& T = T0[t];
            T.seq = a.target_seqs[t];
            T.name = a.target_names[t];
            T.is_rc = a.target_is_rc ? a.target_is_rc[t] != 0 : false;
        }
        w.targets.p = T0;
        w.targets.n = (uint32_t)std::max(a.n_targets, 0);
    }
    std::vector<mtgi::GapWork>& gaps() override { return R->gaps; }
    /* the C-ABI records of a finished part, written while the device works on the next one: such gaps have at most one solution, which
     * takes the slot of its gap in filled_flat (sized for one per gap by marshal) */
    size_t recorded = 0; /* gaps [0, recorded) have their records */
    void record_gap(size_t i, uint64_t& bytes, uint64_t& filled) override
    {
        write_record(i, R->filled_flat.data() + i);
        tally(i, bytes, filled);
    }
    void part_done(size_t first, size_t count, bool clean, uint64_t bytes, uint64_t filled) override
    {
        if (first != recorded || !clean) return; /* from here on the records are rebuilt at the end */
        R->sum_bytes.fetch_add(bytes, std::memory_order_relaxed);
        R->sum_filled.fetch_add(filled, std::memory_order_relaxed);
        recorded = first + count;
    }
    /* what mtg_results_summary reports about gap i */
    void tally(size_t i, uint64_t& bytes, uint64_t& filled)
    {
        const mtgi::GapWork& w = R->gaps[i];
        R->nfilled[i] = (uint32_t)w.sols.size();
        filled += !w.sols.empty();
        for (auto& s : w.sols) bytes += s.seq.size() + 1;
    }
    /* returns the slot after the last one used */
    mtg_filled* write_record(size_t i, mtg_filled* F0)
    {
        mtgi::GapWork& w = R->gaps[i];
        mtg_gap_result& r = R->res[i];
        r.filled = F0;
        for (auto& s : w.sols) {
            mtg_filled& f = *F0++;
            f.seq = s.seq.c_str();
            f.nb_errors_in_anchor = s.nb_errors;
            f.target_index = s.target;
            f.avg_coverage = s.avg;
            f.median_coverage = s.median;
            f.qual = s.qual;
            f.solution_count = s.count;
            f.solution_rank = s.rank;
        }
        r.nb_nodes = w.nb_nodes; r.total_nt = w.total_nt; r.nb_terminal = w.nb_terminal;
        r.has_solution_counts = w.has_counts; r.nb_total_filled = w.nb_total_filled; r.nb_reported = (int)w.sols.size();
        r.n_filled = (int)w.sols.size();
        r.extension = w.extension.c_str();
        return F0;
    }
};
} // namespace

static int fill_batch_impl(const mtg_index* idx, const mtg_params* p, const mtg_gap* gaps, size_t n, char* seq_out, uint64_t seq_cap, uint64_t* seq_bytes, mtg_results** out)
{
    if (!idx || !p || !out || (n && !gaps)) { mtgi::set_error("null argument"); return MTG_ERR_ARG; }
    const double t_m0 = mtgi::now_ms();
    static const bool dbg = getenv("MTG_DEBUG_TIMERS") != nullptr;
    double tk = t_m0;
    auto tick = [&](const char* what) { if (dbg) { const double t = mtgi::now_ms(); fprintf(stderr, "  [fill_batch] %-21s %.2f ms\n", what, t - tk); tk = t; } };
    mtg_results* R = results_acquire();
    R->nthreads = p->nb_host_threads;
    R->sum_bytes.store(0); R->sum_filled.store(0); R->summary_ready = false;
    R->arena.set_external(seq_out, seq_out ? (size_t)seq_cap : 0);
    AbiSource src(gaps, n, R);
    mtg_batch_stats st{};
    std::vector<uint64_t> sol_blocks;
    int rc = mtgi::fill_gaps(idx, p, src, R->arena, &st, &sol_blocks);
    if (rc) { results_release(R); return rc; }
    bool relaid = false; /* the sequences were moved after the parts had been recorded */
    if (seq_out) {
        if (R->arena.ext_ok && R->arena.ext_next_gap == n) *seq_bytes = R->arena.ext_used; /* every sequence was decoded in place, in gap order */
        else {
            /* multi-contig gaps, re-run gaps or a buffer too small for the worst case: lay the sequences out again, in gap order, and
             * make the solutions point there */
            relaid = true;
            const size_t CH = 1024, nch = (n + CH - 1) / CH;
            std::vector<uint64_t> choff(nch + 1, 0);
            mtgi::parallel_for(nch, p->nb_host_threads, [&](size_t c) {
                uint64_t b = 0;
                for (size_t i = c * CH; i < std::min(n, (c + 1) * CH); i++) for (auto& s : R->gaps[i].sols) b += s.seq.size() + 1;
                choff[c + 1] = b;
            }, 1);
            for (size_t c = 0; c < nch; c++) choff[c + 1] += choff[c];
            if (choff[nch] > seq_cap) { results_release(R); mtgi::set_error("sequence buffer too small: %llu bytes needed", (unsigned long long)choff[nch]); return MTG_ERR_ARG; }
            char* tmp = R->arena.ensure(63, choff[nch] + 1); /* a buffer of its own (no chunk gets that far): the old places stay readable meanwhile */
            mtgi::parallel_for(nch, p->nb_host_threads, [&](size_t c) {
                uint64_t o = choff[c];
                for (size_t i = c * CH; i < std::min(n, (c + 1) * CH); i++)
                    for (auto& s : R->gaps[i].sols) { memcpy(tmp + o, s.seq.data(), s.seq.size()); tmp[o + s.seq.size()] = 0; o += s.seq.size() + 1; }
            }, 1);
            memcpy(seq_out, tmp, choff[nch]);
            mtgi::parallel_for(nch, p->nb_host_threads, [&](size_t c) {
                uint64_t o = choff[c];
                for (size_t i = c * CH; i < std::min(n, (c + 1) * CH); i++)
                    for (auto& s : R->gaps[i].sols) { const uint32_t len = (uint32_t)s.seq.size(); s.seq = std::string(); s.seq.view(seq_out + o, len); o += len + 1; }
            }, 1);
            *seq_bytes = choff[nch];
        }
    }
    const double t_m2 = mtgi::now_ms();
    tk = t_m2;
    if (src.recorded == n && !relaid) {
        /* every part wrote its records as it came back */
        R->summary_ready = true;
    } else {
        const size_t B = mtgi::RESULT_BLOCK, nb = (n + B - 1) / B;
        std::vector<uint64_t> blk_off(nb + 1, 0);
        for (size_t b = 0; b < nb; b++) blk_off[b + 1] = blk_off[b] + sol_blocks[b];
        R->res.resize(n);
        if (R->filled_flat.size() < blk_off[nb]) R->filled_flat.resize(blk_off[nb]);
        mtgi::parallel_for(nb, p->nb_host_threads, [&](size_t b) {
            mtg_filled* F0 = R->filled_flat.data() + blk_off[b];
            for (size_t i = b * B; i < std::min(n, (b + 1) * B); i++) F0 = src.write_record(i, F0);
        }, 1);
        R->summary_ready = false; /* computed on demand */
    }
    /* the views on the caller's strings end here */
    tick("result records");
    st.marshal_ms = 0;
    st.result_ms = mtgi::now_ms() - t_m2;
    st.total_ms = mtgi::now_ms() - t_m0;
    mtgi::stats_store(st);
    *out = R;
    return MTG_OK;
}
int mtg_fill_batch(const mtg_index* idx, const mtg_params* p, const mtg_gap* gaps, size_t n, mtg_results** out)
{
    return fill_batch_impl(idx, p, gaps, n, nullptr, 0, nullptr, out);
}
int mtg_fill_batch_serial(const mtg_index* idx, const mtg_params* p, const mtg_gap* gaps, size_t n, char* seq_out, uint64_t cap, uint64_t* seq_bytes, mtg_results** out)
{
    if (!seq_out || !seq_bytes) { mtgi::set_error("null argument"); return MTG_ERR_ARG; }
    return fill_batch_impl(idx, p, gaps, n, seq_out, cap, seq_bytes, out);
}
const mtg_gap_result* mtg_results_get(const mtg_results* r, size_t i) { return (r && i < r->res.size()) ? &r->res[i] : nullptr; }
void mtg_results_free(mtg_results* r)
{
    if (r) results_release(r); /* its storage serves the next batch */
}
int mtg_results_summary(const mtg_results* r, uint32_t* n_filled, uint64_t* seq_bytes, uint64_t* n_gaps_filled)
{
    if (!r) return MTG_ERR_ARG;
    if (r->summary_ready) { /* tallied while the batch was assembled */
        if (n_filled && !r->nfilled.empty()) memcpy(n_filled, r->nfilled.data(), r->nfilled.size() * sizeof(uint32_t));
        if (seq_bytes) *seq_bytes = r->sum_bytes.load();
        if (n_gaps_filled) *n_gaps_filled = r->sum_filled.load();
        return MTG_OK;
    }
    std::atomic<uint64_t> b{0}, nf{0};
    const size_t n = r->gaps.size();
    const size_t CH = 2048;
    mtgi::parallel_for((n + CH - 1) / CH, r->nthreads, [&](size_t c) {
        uint64_t lb = 0, lf = 0;
        for (size_t i = c * CH; i < std::min(n, (c + 1) * CH); i++) {
            if (n_filled) n_filled[i] = (uint32_t)r->gaps[i].sols.size();
            lf += !r->gaps[i].sols.empty();
            for (auto& s : r->gaps[i].sols) lb += s.seq.size() + 1;
        }
        b += lb; nf += lf;
    }, 1);
    if (seq_bytes) *seq_bytes = b.load();
    if (n_gaps_filled) *n_gaps_filled = nf.load();
    return MTG_OK;
}
int mtg_results_copy_seqs(const mtg_results* r, char* dst, uint64_t cap)
{
    if (!r || !dst) return MTG_ERR_ARG;
    const size_t n = r->gaps.size();
    std::vector<uint64_t> off(n + 1, 0);
    for (size_t i = 0; i < n; i++) {
        uint64_t b = 0;
        for (auto& s : r->gaps[i].sols) b += s.seq.size() + 1;
        off[i + 1] = off[i] + b;
    }
    if (off[n] > cap) { mtgi::set_error("destination too small"); return MTG_ERR_ARG; }
    mtgi::parallel_for(n, r->nthreads, [&](size_t i) {
        uint64_t o = off[i];
        for (auto& s : r->gaps[i].sols) {
            memcpy(dst + o, s.seq.data(), s.seq.size());
            o += s.seq.size();
            dst[o++] = '\n';
        }
    }, 512);
    return MTG_OK;
}
int mtg_index_scan_sequences(const mtg_index* idx, const char* const* seqs, size_t nseq, int mode, uint8_t* const* out, mtg_scan_stats* st)
{
    if (!idx || (nseq && (!seqs || !out))) { mtgi::set_error("null argument"); return MTG_ERR_ARG; }
    const int k = idx->dev.k;
    std::vector<uint64_t> off(nseq);
    std::vector<uint32_t> len(nseq);
    uint64_t nw = 0;
    for (size_t s = 0; s < nseq; s++) { len[s] = (uint32_t)strlen(seqs[s]); off[s] = nw; nw += (len[s] + 31) / 32 + 2; }
    std::vector<uint64_t> words(nw + 2, 0), bits(nw + 2, 0);
    for (size_t s = 0; s < nseq; s++)
        for (uint32_t i = 0; i < len[s]; i++) words[off[s] + (i >> 5)] |= (uint64_t)nt_code((unsigned char)seqs[s][i]) << (2 * (i & 31));
    int rc = mtgi::scan_run(idx, words.data(), words.size(), off.data(), len.data(), nseq, mode, bits.data(), 0, st);
    if (rc) return rc;
    for (size_t s = 0; s < nseq; s++) {
        if ((int)len[s] < k) continue;
        const uint32_t npos = len[s] - k + 1;
        int bad_until = -1; /* last position whose k-mer still contains an invalid character */
        for (uint32_t i = 0; i < (uint32_t)k - 1 && i < len[s]; i++) if (mtgi::nt_bad((unsigned char)seqs[s][i])) bad_until = (int)i;
        for (uint32_t p = 0; p < npos; p++) {
            if (mtgi::nt_bad((unsigned char)seqs[s][p + k - 1])) bad_until = (int)(p + k - 1);
            out[s][p] = (bad_until >= (int)p) ? 0 : (uint8_t)((bits[off[s] + (p >> 6)] >> (p & 63)) & 1);
        }
    }
    return MTG_OK;
}

int mtg_nw_matches(const char* const* a, const char* const* b, size_t n, uint32_t* matches)
{
    if (n && (!a || !b || !matches)) { mtgi::set_error("null argument"); return MTG_ERR_ARG; }
    std::vector<mtgi::NwPair> pairs(n);
    for (size_t i = 0; i < n; i++) {
        if (!a[i] || !b[i]) { mtgi::set_error("pair %zu: null sequence", i); return MTG_ERR_ARG; }
        pairs[i] = mtgi::NwPair{a[i], (uint32_t)strlen(a[i]), b[i], (uint32_t)strlen(b[i])};
    }
    std::vector<uint32_t> m;
    int rc = mtgi::nw_run(nullptr, pairs, m);
    if (rc) return rc;
    for (size_t i = 0; i < n; i++) matches[i] = m[i];
    return MTG_OK;
}

int mtg_stage_a_batch(const mtg_index* idx, const mtg_params* p, const char* const* sources, const char* const* targets, size_t n, mtg_contigs** out)
{
    if (!idx || !p || !out || (n && (!sources || !targets))) { mtgi::set_error("null argument"); return MTG_ERR_ARG; }
    mtgi::WorkspaceLock batch_lock = mtgi::acquire_workspace(idx);
    mtgi::FillInput in;
    in.k = idx->dev.k;
    in.ws = batch_lock.ws;
    in.want_all_contigs = true;
    in.resize(n);
    for (size_t i = 0; i < n; i++) {
        if ((int)strlen(sources[i]) < idx->dev.k) { mtgi::set_error("gap %zu: source sequence shorter than k", i); return MTG_ERR_ARG; }
        in.size(i, strlen(targets[i]), 0);
    }
    in.layout();
    for (size_t i = 0; i < n; i++) in.set(i, std::string_view(sources[i]), std::string_view(targets[i]), nullptr, 0);
    mtgi::DevBatch batch;
    mtgi::DevBatch& gc = batch;
    mtg_batch_stats st{};
    int rc = mtgi::device_run(idx, p, in, batch, &st);
    if (rc) return rc;
    mtgi::stats_store(st);
    mtg_contigs* C = new mtg_contigs();
    C->c.resize(n);
    for (size_t i = 0; i < n; i++)
        { const mtgi::GapDev gd = gc[i]; for (uint32_t j = 0; j < gd.o.n_contigs; j++) C->c[i].push_back(gd.contig(j)); }
    *out = C;
    return MTG_OK;
}
size_t mtg_contigs_count(const mtg_contigs* c, size_t gap) { return (c && gap < c->c.size()) ? c->c[gap].size() : 0; }
const char* mtg_contigs_get(const mtg_contigs* c, size_t gap, size_t i) { return (c && gap < c->c.size() && i < c->c[gap].size()) ? c->c[gap][i].c_str() : nullptr; }
void mtg_contigs_free(mtg_contigs* c) { delete c; }
}
