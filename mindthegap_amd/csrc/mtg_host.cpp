/*
 * mtg_host.cpp -- host orchestration of libmtgfill.so.
 *
 * The device builds the contigs of every gap (stage A, mtg_gpu_fill.hip) and answers abundance queries; this file
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
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>

/* ---- the tuning table (mtg_tuning.h): the C-ABI over it ---- */
extern "C" {
size_t mtg_tuning_count(void) { return (size_t)mtgi::tune::T_COUNT; }
int mtg_tuning_describe(size_t i, const char** name, const char** dflt, const char** kind, const char** what)
{
    using namespace mtgi::tune;
    if (i >= (size_t)T_COUNT) { mtgi::set_error("no tuning entry %zu", i); return MTG_ERR_ARG; }
    if (name) *name = g_entries[i].name;
    if (dflt) *dflt = g_entries[i].dflt;
    if (kind) *kind = g_entries[i].kind;
    if (what) *what = g_entries[i].what;
    return MTG_OK;
}
int mtg_tuning_get(const char* name, char* value, size_t cap)
{
    using namespace mtgi::tune;
    const int t = Values::find(name);
    if (t < 0 || !value) { mtgi::set_error("no tuning entry named %s", name ? name : "(null)"); return MTG_ERR_ARG; }
    char c[64];
    (void)values().read(t, c, true); /* as a flag reads it: an empty environment variable is "1" */
    const size_t n = strlen(c);
    if (n + 1 > cap) { mtgi::set_error("buffer too small"); return MTG_ERR_ARG; }
    memcpy(value, c, n + 1);
    return MTG_OK;
}
int mtg_tuning_set(const char* name, const char* value)
{
    using namespace mtgi::tune;
    const int t = Values::find(name);
    if (t < 0) { mtgi::set_error("no tuning entry named %s", name ? name : "(null)"); return MTG_ERR_ARG; }
    Values& V = values();
    std::lock_guard<std::mutex> lk(V.m);
    if (!V.put(t, value)) { mtgi::set_error("value too long"); return MTG_ERR_ARG; }
    return MTG_OK;
}
}
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
    /* gzgets returns NULL at the end of the file and on an error alike: a truncated or corrupt .gz must not pass for a short file */
    int zerr = Z_OK;
    (void)gzerror(f, &zerr);
    const bool clean = gzeof(f) && (zerr == Z_OK || zerr == Z_STREAM_END);
    gzclose(f);
    return clean;
}

/* ------------------------------------------------------------------------------------------------ index from reads */
static inline bool nt_bad(unsigned char c) { return (c >> 3) & 1; } /* gatb: bit 3 of the ASCII code flags 'N' */

/* gatb's automatic solidity cut-off, restated: smoothed histogram, first minimum, coverage peak,
 * arg-min between; floor 3 (src/Filler.cpp:201).  One golden datapoint: 7 (test/full_test/gold_fill.output:11). */
int auto_cutoff(const std::vector<uint64_t>& h, int floor_thr)
{
    const size_t len = h.size();
    if (len < 5) return floor_thr;
    /* integer entries, as in gatb's Histogram (the weighted sums are truncated): see the two datapoints in tests/test_micro_cases.py */
    std::vector<uint64_t> sm(len, 0);
    sm[1] = (uint64_t)(0.6 * (double)h[1] + 0.4 * (double)h[2]);
    for (size_t i = 2; i + 1 < len; i++) sm[i] = (uint64_t)(0.2 * (double)h[i - 1] + 0.6 * (double)h[i] + 0.2 * (double)h[i + 1]);
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

/* FASTA / FASTQ (.gz) records of a list of files, one after the other, as text blocks for the device */
namespace {
struct FileReadStream : ReadStream {
    std::vector<std::string> paths;
    size_t cur = 0;
    gzFile f = nullptr;
    bool bad = false;
    std::string block, line, bad_path;
    std::vector<char> buf;
    size_t hint = 0;
    bool have_line = false; /* `line` holds a line that has been read ahead */
    enum { BLOCK = 64 << 20 };
    explicit FileReadStream(const std::vector<std::string>& p) : paths(p), buf(1 << 16)
    {
        for (const std::string& q : paths) {
            FILE* t = fopen(q.c_str(), "rb");
            if (!t) continue;
            fseek(t, 0, SEEK_END);
            const long sz = ftell(t);
            fclose(t);
            const bool gz = q.size() > 3 && q.compare(q.size() - 3, 3, ".gz") == 0;
            hint += (size_t)std::max(0L, sz) * (gz ? 4 : 1);
        }
    }
    ~FileReadStream() override { if (f) gzclose(f); }
    bool rewind() override
    {
        if (f) { gzclose(f); f = nullptr; }
        cur = 0;
        bad = false;
        have_line = false;
        return true;
    }
    bool failed() const override { return bad; }
    size_t size_hint() const override { return hint; }
    bool getl()
    {
        line.clear();
        bool got = false;
        while (gzgets(f, buf.data(), (int)buf.size())) {
            got = true;
            size_t n = strlen(buf.data());
            if (n && buf[n - 1] == '\n') {
                line.append(buf.data(), n - 1);
                if (!line.empty() && line.back() == '\r') line.pop_back();
                return true;
            }
            line.append(buf.data(), n);
        }
        /* NULL from gzgets: the end of the file, or a read / inflate error (truncated or corrupt .gz) -- only the first is a clean end */
        int zerr = Z_OK;
        (void)gzerror(f, &zerr);
        if (!gzeof(f) || !(zerr == Z_OK || zerr == Z_STREAM_END)) { bad = true; bad_path = paths[cur]; }
        return got;
    }
    bool next_block(const char*& p, size_t& n) override
    {
        block.clear();
        while (block.size() < (size_t)BLOCK) {
            if (!f) {
                if (cur >= paths.size()) break;
                f = gzopen(paths[cur].c_str(), "rb");
                if (!f) { bad = true; bad_path = paths[cur]; return false; }
                gzbuffer(f, 1 << 20);
                have_line = getl();
            }
            if (bad) return false;
            if (!have_line) { gzclose(f); f = nullptr; cur++; continue; }
            if (!line.empty() && line[0] == '>') { /* FASTA record: the following lines up to the next header */
                while ((have_line = getl()) && (line.empty() || line[0] != '>')) block += line;
                block += '\n';
            } else if (!line.empty() && line[0] == '@') { /* FASTQ record: sequence, '+', qualities */
                if ((have_line = getl())) { block += line; block += '\n'; }
                have_line = getl();
                have_line = getl();
                have_line = getl();
            } else have_line = getl();
        }
        if (bad) return false;
        p = block.data();
        n = block.size();
        return n > 0;
    }
};
} // namespace

int index_from_reads(const char* paths_csv, int k, int abundance_min, int abundance_max, mtg_index** out)
{
    if (!paths_csv || !out || k < 11 || k > 31) { set_error("invalid argument (11 <= k <= 31)"); return MTG_ERR_ARG; }
    std::vector<std::string> paths;
    std::string csv(paths_csv);
    size_t pos = 0;
    while (pos <= csv.size()) {
        size_t e = csv.find(',', pos);
        if (e == std::string::npos) e = csv.size();
        std::string path = csv.substr(pos, e - pos);
        pos = e + 1;
        if (!path.empty()) paths.push_back(path);
    }
    /* sum solidity over all files (STR_SOLIDITY_KIND "sum", src/Filler.cpp:177): the files are one stream of reads */
    FileReadStream rs(paths);
    const int rc = index_from_stream(rs, k, abundance_min, abundance_max, out);
    if (rs.failed()) { set_error("cannot read %s", rs.bad_path.c_str()); return MTG_ERR_IO; }
    return rc;
}

static const char IDX_MAGIC[8] = {'M', 'T', 'G', 'I', 'D', 'X', '1', 0};

/* The container: magic, k, abundance_min, abundance_auto, number of k-mers, then (k-mer : 8 bytes, abundance : 4 bytes) records in
 * no particular order.  The k-mers are read back from the device tables (the index keeps no host copy), in pieces. */
static const char IDX_MAGIC2[8] = {'M', 'T', 'G', 'I', 'D', 'X', '2', 0};
/* Version 3: the index as it is: the unitig store (2-bit sequences + one abundance byte per k-mer) and the k-mers of no stored unitig.
 *   magic | int32 k, abundance_min, abundance_auto, 0 | uint64 nb_solid, nb_branching, nb_saturated, n_words, n_unitigs, n_left |
 *   (n_words + 8) words | (n_words + 8) * 32 abundance bytes | n_left x (k-mer : 8 bytes, abundance : 4 bytes)
 * At human scale 4 GB instead of the 36 GB of the k-mer list (version 2, still read), and nothing is rebuilt on load but the tables. */
static const char IDX_MAGIC3[8] = {'M', 'T', 'G', 'I', 'D', 'X', '3', 0};
int index_save(const mtg_index* idx, const char* path)
{
    if (!idx || !path) { set_error("null argument"); return MTG_ERR_ARG; }
    IndexDump d;
    if (int rc = index_dump(idx, d)) return rc;
    FILE* f = fopen(path, "wb");
    if (!f) { set_error("cannot write %s", path); return MTG_ERR_IO; }
    const int32_t hdr[4] = {d.k, d.abundance_min, d.abundance_auto, 0};
    const uint64_t cnt[6] = {d.nb_solid, d.nb_branching, d.nb_saturated, d.n_words, d.n_unitigs, (uint64_t)d.left_k.size()};
    bool ok = fwrite(IDX_MAGIC3, 1, 8, f) == 8 && fwrite(hdr, 4, 4, f) == 4 && fwrite(cnt, 8, 6, f) == 6;
    const uint64_t nw = d.n_words ? d.n_words + 8 : 0;
    if (ok && nw) ok = fwrite(d.words.data(), 8, nw, f) == nw && fwrite(d.ab.data(), 1, nw * 32, f) == nw * 32;
    if (ok && !d.left_k.empty()) {
        std::vector<unsigned char> rec(d.left_k.size() * 12);
        for (size_t i = 0; i < d.left_k.size(); i++) { memcpy(rec.data() + 12 * i, &d.left_k[i], 8); memcpy(rec.data() + 12 * i + 8, &d.left_a[i], 4); }
        ok = fwrite(rec.data(), 12, d.left_k.size(), f) == d.left_k.size();
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok) { set_error("short write on %s", path); return MTG_ERR_IO; }
    return MTG_OK;
}

/* bytes [off, off + n) of the file into dst, by a few threads (the page cache, or a memory-backed file system, gives one reader 3-5 GB/s) */
static bool pread_all(int fd, void* dst, size_t n, uint64_t off)
{
    size_t done = 0;
    while (done < n) {
        const ssize_t r = ::pread(fd, (char*)dst + done, n - done, (off_t)(off + done));
        if (r <= 0) return false;
        done += (size_t)r;
    }
    return true;
}
static bool pread_parallel(int fd, void* dst, size_t n, uint64_t off)
{
    const size_t nt = n < ((size_t)64 << 20) ? 1 : (size_t)std::min(8, std::max(1, Pool::cpu_budget()));
    if (nt == 1) return pread_all(fd, dst, n, off);
    std::vector<std::thread> ts;
    std::atomic<bool> ok{true};
    const size_t piece = ((n + nt - 1) / nt + 4095) & ~(size_t)4095;
    for (size_t t = 0; t < nt; t++) {
        const size_t lo = std::min(n, t * piece), hi = std::min(n, (t + 1) * piece);
        if (lo < hi) ts.emplace_back([=, &ok] { if (!pread_all(fd, (char*)dst + lo, hi - lo, off + lo)) ok.store(false); });
    }
    for (auto& t : ts) t.join();
    return ok.load();
}
static int index_load_v3(FILE* f, const char* path, mtg_index** out)
{
    int32_t hdr[4];
    uint64_t cnt[6];
    if (fread(hdr, 4, 4, f) != 4 || fread(cnt, 8, 6, f) != 6) { set_error("%s: truncated", path); return MTG_ERR_FORMAT; }
    const long data0 = ftell(f);
    fseek(f, 0, SEEK_END);
    const long fsize = ftell(f);
    fseek(f, data0, SEEK_SET);
    const uint64_t nw = cnt[3] ? cnt[3] + 8 : 0;
    if (hdr[0] < 11 || hdr[0] > 31 || fsize < data0 || cnt[3] > (1ull << 40) || cnt[5] > (1ull << 40) || nw * 40 + cnt[5] * 12 != (uint64_t)(fsize - data0)) {
        set_error("%s: truncated or damaged (the header announces %llu store words and %llu k-mers)", path, (unsigned long long)cnt[3], (unsigned long long)cnt[5]);
        return MTG_ERR_FORMAT;
    }
    IndexDump d;
    d.k = hdr[0]; d.abundance_min = hdr[1]; d.abundance_auto = hdr[2];
    d.nb_solid = cnt[0]; d.nb_branching = cnt[1]; d.nb_saturated = cnt[2]; d.n_words = cnt[3]; d.n_unitigs = cnt[4];
    if (d.n_words) d.prealloc = adj_prealloc_begin(d.nb_solid, d.k); /* the largest allocation of the load starts now, on a helper thread, and the file is read meanwhile */
    struct DropPre { IndexDump& d; bool handed = false; ~DropPre() { if (!handed) adj_prealloc_drop(d.prealloc); } } drop_pre{d};
    /* the words (a fifth of the store) come to host memory -- the walk over the unitigs' header words needs them -- read by several threads;
     * the abundance bytes go from the file to the device in page-locked pieces while the tables are being derived (index_from_dump) */
    const int fd = fileno(f);
    d.words.resize(nw);
    bool ok = !nw || pread_parallel(fd, d.words.data(), nw * 8, (uint64_t)data0);
    const uint64_t ab0 = (uint64_t)data0 + nw * 8, left0 = ab0 + nw * 32;
    d.left_k.resize(cnt[5]); d.left_a.resize(cnt[5]);
    if (ok && cnt[5]) {
        std::vector<unsigned char> rec(cnt[5] * 12);
        ok = pread_parallel(fd, rec.data(), rec.size(), left0);
        for (size_t i = 0; i < cnt[5] && ok; i++) { memcpy(&d.left_k[i], rec.data() + 12 * i, 8); memcpy(&d.left_a[i], rec.data() + 12 * i + 8, 4); }
    }
    if (!ok) { set_error("%s: truncated", path); return MTG_ERR_FORMAT; }
    if (nw) d.ab_read = [fd, ab0](uint64_t off, size_t n, void* dst) { return pread_all(fd, dst, n, ab0 + off); };
    drop_pre.handed = true; /* index_from_dump takes the pre-allocation over (and drops it on every path) */
    return index_from_dump(d, out);
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
    if (memcmp(magic, IDX_MAGIC3, 8) == 0) { const int rc3 = index_load_v3(f, path, out); fclose(f); return rc3; }
    const bool v1 = memcmp(magic, IDX_MAGIC, 8) == 0, v2 = memcmp(magic, IDX_MAGIC2, 8) == 0;
    if (!(v1 || v2) || fread(hdr, 4, 4, f) != 4 || fread(&n, 8, 1, f) != 1) { fclose(f); set_error("%s: not a mtg index", path); return MTG_ERR_FORMAT; }
    /* the records go to the device piece by piece, straight from the file (every attempt to size the tables reads them once) */
    const long data0 = ftell(f);
    {
        /* the header's record count against what the file holds, before anything is sized by it */
        fseek(f, 0, SEEK_END);
        const long fsize = ftell(f);
        fseek(f, data0, SEEK_SET);
        if (hdr[0] < 11 || hdr[0] > 31 || fsize < data0 || n > (uint64_t)(fsize - data0) / 12) { fclose(f); set_error("%s: truncated (the header announces %llu k-mers)", path, (unsigned long long)n); return MTG_ERR_FORMAT; }
    }
    std::vector<uint64_t> pk;
    std::vector<uint32_t> pa;
    std::vector<unsigned char> rec;
    auto fetch = [&](size_t off, size_t m, const uint64_t*& hk, const uint32_t*& ha) -> bool {
        pk.resize(m);
        pa.resize(m);
        bool ok;
        if (v1) { /* version 1: all k-mers, then all abundances */
            ok = fseek(f, data0 + (long)(8 * off), SEEK_SET) == 0 && fread(pk.data(), 8, m, f) == m && fseek(f, data0 + (long)(8 * n + 4 * off), SEEK_SET) == 0 &&
                 fread(pa.data(), 4, m, f) == m;
        } else {
            ok = fseek(f, data0 + (long)(12 * off), SEEK_SET) == 0;
            rec.resize((size_t)12 << 20);
            for (size_t done = 0; done < m && ok;) {
                const size_t c = std::min(m - done, rec.size() / 12);
                ok = fread(rec.data(), 12, c, f) == c;
                for (size_t i = 0; i < c && ok; i++) { memcpy(&pk[done + i], rec.data() + 12 * i, 8); memcpy(&pa[done + i], rec.data() + 12 * i + 8, 4); }
                done += c;
            }
        }
        if (!ok) set_error("%s: truncated", path);
        hk = pk.data();
        ha = pa.data();
        return ok;
    };
    int rc = index_from_kmer_pieces((size_t)n, hdr[0], fetch, out);
    fclose(f);
    if (rc == MTG_ERR_IO) rc = MTG_ERR_FORMAT;
    if (rc) return rc;
    (*out)->info.abundance_min = hdr[1];
    (*out)->info.abundance_auto = hdr[2];
    return MTG_OK;
}

/* ------------------------------------------------------------------------------------------------ gap post-processing */
struct TermInfo { /* info_node_t, src/Filler.hpp:44-71 */
    int node, pos, errors, target;
};

/* marshalling of a batch of gapFillFromSource calls (targets == nullptr: contigs only, the stage A parity entry) */
void FillInput::resize(size_t n)
{
    bytes_a = (size_t)BYTES_A_PER_GAP * n8(n) + 64;
    block_a = ws ? staging_host(ws, 0, bytes_a) : nullptr;
    if (!block_a) { own_a.resize(bytes_a / 8 + 1); block_a = own_a.data(); }
    uint8_t* b = (uint8_t*)block_a;
    src.p = (uint64_t*)(b + off_a(n, 0)); r0.p = (uint64_t*)(b + off_a(n, 1));
    roff.p = (uint32_t*)(b + off_a(n, 2)); rlen.p = (uint32_t*)(b + off_a(n, 3)); toff.p = (uint32_t*)(b + off_a(n, 4)); tcnt.p = (uint32_t*)(b + off_a(n, 5));
    nbmis.p = b + off_a(n, 6); fast_ok.p = b + off_a(n, 7); flags.p = b + off_a(n, 8);
    src.n = r0.n = roff.n = rlen.n = toff.n = tcnt.n = nbmis.n = fast_ok.n = flags.n = n;
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
const bool have_bmi2 = __builtin_cpu_supports("bmi2") && !tune::on(tune::T_NO_VEC);
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
    size_t i = 0;
#if defined(__x86_64__)
    /* sixteen characters a step (SSE2, every x86-64 has it): contig mode's early-stop pattern is every target of the dictionary one after the other --
     * 620 000 characters a seed at 10 000 contigs, and a byte a cycle through the table was 8 of a batch's 9.5 ms on the host (round 6) */
    static const bool vec = !tune::on(tune::T_NO_VEC);
    if (vec) {
        const __m128i A = _mm_set1_epi8('A'), C = _mm_set1_epi8('C'), G = _mm_set1_epi8('G'), T = _mm_set1_epi8('T');
        for (; i + 16 <= n; i += 16) {
            const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + i));
            const __m128i m = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(v, A), _mm_cmpeq_epi8(v, C)), _mm_or_si128(_mm_cmpeq_epi8(v, G), _mm_cmpeq_epi8(v, T)));
            if (_mm_movemask_epi8(m) != 0xFFFF) return false;
        }
    }
#endif
    for (; i < n; i++) ok &= tab.ok[(unsigned char)s[i]];
    return ok;
}
} // namespace

void FillInput::set_common(size_t g, std::string_view source, std::string_view swf_target, int nb_mis, uint8_t gap_flags)
{
    flags[g] = gap_flags;
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

/* a gap whose solutions come out of the contig graph: candidate sequences per target (in the order the reference visits them), waiting
 * for the alignments of remove_almost_identical_solutions, which run on the device for the whole batch */
struct GenWork {
    std::vector<std::vector<Solution>> groups;
    std::vector<std::vector<int64_t>> pair; /* per group, n x n: index into the batch's NW pairs for (row j, column i < j); -2: equal strings */
};

/* Multi-contig gap (a terminal node that is not the simple case of contig 0): everything after the device kernels except the
 * de-duplication and the coverage numbers.  Returns the candidates, nullptr when there is none. */
static GenWork* process_general(const GapDev& gc, GapWork& W, int k)
{
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

thread_local bool tl_host_general = false;

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

/* The multi-contig gaps of a batch (`special`), on the host: candidates, alignments (device), greedy de-duplication, coverage (device).
 * work[i] receives the solutions of special.special[i]; describe(gap, W) sets source, flags and dictionary of a gap. */
static int run_general(const mtg_index* idx, const mtg_params* p, const DevBatch& special, const std::function<void(size_t, GapWork&)>& describe, std::vector<GapWork>& work, Workspace* ws)
{
    const int k = idx->dev.k, nth = p->nb_host_threads;
    const size_t ng = special.special.size();
    work.clear();
    work.resize(ng);
    std::vector<GenWork*> genw(ng, nullptr);
    struct GenGuard { std::vector<GenWork*>& v; ~GenGuard() { for (GenWork* g : v) delete g; } } gen_guard{genw};
    /* a gap the device has finished (k_general, mtg_general.h): its solutions as they came back */
    auto from_device = [&](const SpecialGap& sg, GapWork& W) -> bool {
        const HostChunk& c = *special.chunks[sg.chunk];
        if (sg.rank >= c.gen_gaps.size() || c.gen_gaps[sg.rank].status != GEN_OK) return false;
        const GenGap& g = c.gen_gaps[sg.rank];
        auto take = [&](uint32_t j) {
            const GenSol& d = c.gen_sols[g.first_sol + j];
            Solution s;
            s.seq.assign(c.gen_ascii.data() + d.seq_off, d.seq_len);
            s.nb_errors = d.nb_errors; s.target = (int)d.target; s.count = d.count; s.rank = d.rank; s.qual = d.qual; s.avg = d.avg; s.median = d.median;
            s.ab_n = 0; /* coverage done */
            W.sols.push_back(std::move(s));
        };
        if (g.n_groups > 1) {
            /* several targets reached (contig mode): the device answered group by group, in the order the reference INSERTS the groups' names into
             * its unordered_map<string, set<path>> (src/Filler.cpp:924-936); the solutions come out in the order that map is ITERATED, which is
             * libstdc++'s business: the same insertions into the same kind of map give it.  Two targets under one name would be one group there:
             * such a gap (not seen with the reference's dictionaries: a name and its _Rc form per contig end) takes the host's path. */
            std::unordered_map<std::string, std::pair<uint32_t, uint32_t>> order; /* name -> first record, records */
            for (uint32_t j = 0; j < g.n_sols;) {
                const GenSol& h = c.gen_sols[g.first_sol + j];
                if (h.rank != 0 || h.target >= W.targets.size() || j + 1u + (uint32_t)h.count > g.n_sols) return false;
                std::string key(W.targets[h.target].name);
                if (W.targets[h.target].is_rc) key += "_Rc";
                if (!order.emplace(std::move(key), std::make_pair(j + 1u, (uint32_t)h.count)).second) return false;
                j += 1u + (uint32_t)h.count;
            }
            for (auto it = order.begin(); it != order.end(); ++it)
                for (uint32_t j = 0; j < it->second.second; j++) take(it->second.first + j);
        } else {
            for (uint32_t j = 0; j < g.n_sols; j++) take(j);
        }
        W.nb_total_filled += (int)g.nb_total_filled;
        W.has_counts = (W.nb_total_filled > 0) || W.reverse; /* src/Filler.cpp:1012 */
        return true;
    };
    std::vector<char> on_device(ng, 0);
    std::atomic<bool> no_contigs{false};
    /* TEST-ONLY (HostChunk::gen_check, set by the emulation's device_run): the gaps the device function finished ALSO take the host's path, in
     * `check`, and the two answers are compared at the end */
    std::vector<GapWork> check;
    std::vector<size_t> check_of;
    for (size_t i = 0; i < ng; i++) {
        const HostChunk& c = *special.chunks[special.special[i].chunk];
        if (c.gen_check && special.special[i].rank < c.gen_gaps.size() && c.gen_gaps[special.special[i].rank].status == GEN_OK) check_of.push_back(i);
    }
    parallel_for(ng, nth, [&](size_t i) {
        const SpecialGap& sg = special.special[i];
        describe(sg.gap, work[i]);
        if (from_device(sg, work[i])) { on_device[i] = 1; return; }
        /* the host's path needs the gap's contigs: a launch whose multi-contig gaps the device all finished did not bring them (device_run copies
         * them when a gap came back GEN_HOST).  A finished gap the host cannot take over (two targets under one name: one group in the reference) asks
         * for the batch to be run again with the host's path for every multi-contig gap (fill_marshalled) -- not a read through a null pointer. */
        if (!special.chunks[sg.chunk]->recs) { no_contigs.store(true, std::memory_order_relaxed); return; }
        genw[i] = process_general(special.view(sg), work[i], k);
    }, 1);
    if (no_contigs.load()) {
        if (!tl_host_general) return MTG_INTERNAL_RETRY_HOST_GENERAL;
        set_error("a multi-contig gap could not be taken from the device's answer, and its contigs are not on the host");
        return MTG_ERR_FORMAT;
    }
    if (!check_of.empty()) {
        DevBatch sub; /* the same gaps as a batch of their own with the device's answers hidden: run_general's host path, whole */
        for (size_t i : check_of) {
            const SpecialGap& sg = special.special[i];
            sub.special.push_back(SpecialGap{sg.gap, 0, sg.slot, ~0u});
        }
        /* one chunk per source chunk would need re-indexing; the emulation's batches of this kind come from one launch: check that and borrow it */
        const uint32_t ch0 = special.special[check_of[0]].chunk;
        for (size_t i : check_of) if (special.special[i].chunk != ch0) { set_error("general-path cross-check: gaps of several launches"); return MTG_ERR_OVERFLOW; }
        HostChunk* borrowed = special.chunks[ch0].get();
        sub.chunks.emplace_back(borrowed);
        struct Unborrow { DevBatch& b; ~Unborrow() { (void)b.chunks[0].release(); } } unborrow{sub};
        const bool saved = borrowed->gen_check;
        borrowed->gen_check = false;
        std::vector<GenGap> hidden;
        hidden.swap(borrowed->gen_gaps);
        const int crc = run_general(idx, p, sub, describe, check, ws);
        hidden.swap(borrowed->gen_gaps);
        borrowed->gen_check = saved;
        if (crc) return crc;
        for (size_t q = 0; q < check_of.size(); q++) {
            if (!on_device[check_of[q]]) continue; /* the host could not take this gap from the device's answer (two targets under one name): its own path is what runs, below */
            const GapWork& a = work[check_of[q]];
            const GapWork& b = check[q];
            bool same = a.sols.size() == b.sols.size() && a.nb_total_filled == b.nb_total_filled && a.has_counts == b.has_counts;
            for (size_t j = 0; same && j < a.sols.size(); j++) {
                const Solution &x = a.sols[j], &y = b.sols[j];
                same = x.seq == y.seq && x.nb_errors == y.nb_errors && x.target == y.target && x.avg == y.avg && x.median == y.median && x.qual == y.qual && x.count == y.count && x.rank == y.rank;
            }
            if (!same) { set_error("gap %u: the device's multi-contig path (mtg_general.h) and the host's disagree", special.special[check_of[q]].gap); return MTG_ERR_OVERFLOW; }
        }
    }
    /* every alignment remove_almost_identical_solutions can ask for: a later candidate (rows) against an earlier one (columns) */
    std::vector<NwPair> pairs;
    for (size_t gi = 0; gi < ng; gi++) {
        if (!genw[gi]) continue;
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
    if (!pairs.empty()) { if (int rc = nw_run(idx, pairs, matches, ws)) return rc; }
    parallel_for(ng, nth, [&](size_t i) { if (genw[i]) finish_general(work[i], *genw[i], matches); }, 1);
    /* coverage of the solutions: abundance of every k-mer of source + seq (src/Filler.cpp:959-988), one batched device query */
    std::vector<uint64_t> q;
    const uint64_t mk = kmask(k);
    for (size_t gi = 0; gi < ng; gi++) {
        GapWork& g = work[gi];
        if (on_device[gi]) continue; /* coverage, quality and orientation are the device's */
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
    std::vector<uint32_t> ab(q.size());
    if (!q.empty()) { if (int rc = query_run(idx, q.data(), q.size(), ab.data(), nullptr, nullptr, ws)) return rc; }
    parallel_for(ng, nth, [&](size_t ii) {
        GapWork& g = work[ii];
        if (on_device[ii]) return;
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
            if (g.reverse) s.seq = revcomp_str(s.seq);
        }
    }, 1);
    return MTG_OK;
}

} // namespace mtgi

/* ------------------------------------------------------------------------------------------------ C ABI (host side) */
/* A batch of gapFillFromSource calls marshalled for the device (and, when prepared ahead, resident there). */
struct mtg_batch {
    mtgi::FillInput in;
    const mtg_gap* gaps = nullptr; /* the caller's array: read again only for the gaps that take the multi-contig path */
    size_t n = 0;
    int device = 0;         /* where the device copies live: the batch can only be filled with an index of that device */
    int nb_mis_allowed = 0; /* marshalled into the batch (per-gap mismatch allowance): the params of a fill must agree */
    ~mtg_batch() { mtgi::batch_release_device(in); }
};

struct mtg_results {
    size_t n = 0;
    /* ONE page-locked block [records | filled records | sequence arena]: the device lays its copies out the same way and the results of a batch
     * cross the link in one copy (round 4: three arrays, three copies) */
    char* blk = nullptr;
    size_t blk_cap = 0;
    static size_t off_fil(size_t n) { return (n * sizeof(mtg_gap_result) + 63) & ~(size_t)63; }
    static size_t off_seq(size_t n) { return (off_fil(n) + n * sizeof(mtg_filled) + 63) & ~(size_t)63; }
    mtg_gap_result* res = nullptr; /* page-locked: the device's records are copied straight into them */
    mtg_filled* fil = nullptr;
    char* seq = nullptr;           /* sequence arena (page-locked), or the caller's buffer */
    size_t seq_cap = 0;
    bool seq_external = false;
    bool seq_on_device = false;    /* the caller's buffer is device memory: the records' seq pointers are device addresses */
    char* seq_own = nullptr;       /* the arena this object owns (kept while the caller's buffer is in use) */
    size_t seq_own_cap = 0;
    char* ext = nullptr;
    size_t ext_cap = 0;
    uint64_t seq_bytes = 0, n_gaps_filled = 0;
    bool in_gap_order = true;      /* the arena holds exactly the batch's sequences, NUL-terminated, in gap order */
    std::vector<mtgi::GapWork> gen;    /* multi-contig gaps: their solutions own their strings */
    std::vector<mtg_filled> gen_fil;   /* and these are their records */
    std::vector<char> relaid;          /* sequences laid out again in gap order (serialised form of a batch with multi-contig gaps) */
    int nthreads = 0;
    bool plain = false;                /* made by mtg_results_from_wire: malloc'd records pointing into the caller's payload, no device */
    ~mtg_results()
    {
        if (plain) { free(res); free(fil); return; }
        mtgi::pinned_free(blk);
        mtgi::pinned_free(ext);
    }
    /* the parts of the block for n_ gaps (the arena: what is left of the block, at least seq_min bytes) */
    void carve(size_t n_)
    {
        n = n_;
        res = (mtg_gap_result*)blk;
        fil = (mtg_filled*)(blk + off_fil(n_));
        seq_own = blk + off_seq(n_);
        seq_own_cap = blk_cap - off_seq(n_);
    }
    bool ensure(size_t n_, size_t seq_min)
    {
        const size_t need = off_seq(n_) + seq_min;
        if (blk_cap < need || !blk) {
            mtgi::pinned_free(blk);
            blk_cap = need + need / 8 + 4096;
            blk = (char*)mtgi::pinned_alloc(blk_cap);
            if (!blk) { blk_cap = 0; res = nullptr; fil = nullptr; seq_own = nullptr; seq_own_cap = 0; return false; }
        }
        carve(n_);
        return true;
    }
    /* a larger block whose first off_seq(n) + keep bytes are the old one's */
    bool grow(size_t seq_need, size_t keep)
    {
        const size_t cap = off_seq(n) + seq_need + seq_need / 4 + 4096;
        char* nb = (char*)mtgi::pinned_alloc(cap);
        if (!nb) return false;
        if (blk) memcpy(nb, blk, off_seq(n) + keep);
        mtgi::pinned_free(blk);
        blk = nb;
        blk_cap = cap;
        carve(n);
        return true;
    }
};
struct mtg_contigs {
    std::vector<std::vector<std::string>> c;
};
/* the text of a batch's simple sites, formatted on the device (mtg_fill_text_formatted); the arenas are page-locked and kept from batch to batch */
struct mtg_formatted {
    mtgi::FormatOut o;
    ~mtg_formatted() { for (int s = 0; s < mtg::FMT_STREAMS; s++) mtgi::pinned_free(o.text[s]); }
};

/* ---- the relocatable form of a result set (include/mtg_fill.h: mtg_wire_*) */
namespace {
const uint64_t WIRE_MAGIC = 0x3145524957474D54ull; /* "MTGWIRE1" */
inline uint64_t up8(uint64_t x) { return (x + 7) & ~7ull; }
uint64_t wire_checksum(const void* p, uint64_t bytes, int nthreads)
{
    const uint64_t* w = (const uint64_t*)p;
    const uint64_t n = bytes / 8, CH = 1 << 16, nch = (n + CH - 1) / CH;
    std::vector<uint64_t> part(nch, 0);
    mtgi::parallel_for((size_t)nch, nthreads, [&](size_t c) {
        uint64_t s = 0;
        const uint64_t e = std::min<uint64_t>(n, (c + 1) * CH);
        for (uint64_t i = c * CH; i < e; i++) s += (w[i] ^ (i * 0x9E3779B97F4A7C15ull)) * 0xBF58476D1CE4E5B9ull;
        part[c] = s;
    }, 1);
    uint64_t s = 0;
    for (uint64_t v : part) s += v;
    return s;
}
}

/* Result objects are recycled: a freed one keeps its page-locked arrays (a hundred bytes per gap plus the sequence arena) for the next
 * batch, which then pays neither allocation nor page faults.  At most six are kept (three batches in flight, each with the previous
 * result still in its caller's hands). */
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
    r->gen.clear();
    r->gen_fil.clear();
    r->relaid.clear();
    {
        std::lock_guard<std::mutex> lk(g_results_mtx);
        if (g_results_cache.size() < 12) { g_results_cache.push_back(r); return; }
    }
    delete r;
}

/* marshals gaps[0, n) into `in` (whose k and workspace are set); the strings are read here and, for the rare multi-contig gap, again later */
int marshal_gaps(const mtg_gap* g, size_t n, const mtg_params* p, mtgi::FillInput& in)
{
    using namespace mtgi;
    const int k = in.k, nth = p->nb_host_threads;
    enum { WIDE_DICT = 1024, WIDE_PIECE = 1024 };
    std::atomic<long> bad_gap{-1}, short_gap{-1};
    const bool prefetch_on = !tune::on(tune::T_NO_PREFETCH);
    /* the strings of a batch are wherever the caller has them: the passes ask for those of the gaps a few places ahead early */
    const auto prefetch = [&](size_t i) {
        if (!prefetch_on) return;
        if (i + 16 < n) { const mtg_gap& b = g[i + 16]; __builtin_prefetch(b.target_seqs); __builtin_prefetch(b.target_names); }
        if (i + 8 < n) {
            const mtg_gap& b = g[i + 8];
            __builtin_prefetch(b.source);
            __builtin_prefetch(b.target);
            if (b.n_targets > 0 && b.target_seqs && b.target_names) { __builtin_prefetch(b.target_seqs[0]); __builtin_prefetch(b.target_names[0]); }
        }
    };
    const auto sizes_of = [&](size_t i, size_t& swf_len, size_t& n_targets) -> bool {
        prefetch(i);
        const mtg_gap& a = g[i];
        bool ok = a.source && a.target && !(a.n_targets > 0 && (!a.target_seqs || !a.target_names));
        if (ok) for (int t = 0; t < a.n_targets; t++) if (!a.target_seqs[t] || !a.target_names[t]) ok = false;
        size_t src_len = 0;
        if (!ok) { bad_gap = (long)i; swf_len = n_targets = 0; }
        else {
            src_len = strlen(a.source); swf_len = strlen(a.target); n_targets = (size_t)std::max(a.n_targets, 0);
            if ((int)src_len < k) { short_gap = (long)i; ok = false; }
        }
        in.slen[i] = (uint32_t)src_len;
        return ok;
    };
    const auto input_of = [&](size_t i) {
        prefetch(i);
        const mtg_gap& a = g[i];
        const uint8_t fl = (uint8_t)((a.is_anchor_repeated ? mtg::GAPF_REPEATED : 0) | (a.reverse ? mtg::GAPF_REVERSE : 0));
        in.set_common(i, std::string_view(a.source, in.slen[i]), std::string_view(a.target, in.rlen[i]), a.is_anchor_repeated ? 0 : p->nb_mis_allowed, fl); /* src/Filler.cpp:859-863 */
        /* of a target only the first k characters matter, and whether it has them */
        if (a.n_targets >= WIDE_DICT) return; /* a dictionary of thousands (contig mode): dealt to the pool below -- a batch of such gaps has fewer gaps than the passes here have blocks */
        for (int t = 0; t < a.n_targets; t++) in.set_target(in.toff[i] + (size_t)t, std::string_view(a.target_seqs[t], strnlen(a.target_seqs[t], (size_t)in.k)));
    };
    const auto wide_targets = [&]() {
        std::vector<std::pair<uint32_t, uint32_t>> pieces; /* (gap, first target) */
        for (size_t i = 0; i < n; i++)
            if (g[i].n_targets >= WIDE_DICT) for (int t = 0; t < g[i].n_targets; t += WIDE_PIECE) pieces.push_back({(uint32_t)i, (uint32_t)t});
        if (pieces.empty()) return;
        parallel_for(pieces.size(), nth, [&](size_t j) {
            const mtg_gap& a = g[pieces[j].first];
            const size_t t0 = pieces[j].second, t1 = std::min<size_t>((size_t)a.n_targets, t0 + WIDE_PIECE), base = in.toff[pieces[j].first];
            for (size_t t = t0; t < t1; t++) in.set_target(base + t, std::string_view(a.target_seqs[t], strnlen(a.target_seqs[t], (size_t)in.k)));
        }, 1);
    };
    if (!in.plan_and_fill(n, nth, sizes_of, input_of)) {
        /* the first batch of its shape on this workspace (the staging blocks have to grow), or a malformed gap */
        bad_gap = -1; short_gap = -1;
        in.plan(n, nth, sizes_of);
        if (bad_gap >= 0) { set_error("gap %ld: null field", bad_gap.load()); return MTG_ERR_ARG; }
        if (short_gap >= 0) { set_error("gap %ld: source sequence shorter than k", short_gap.load()); return MTG_ERR_ARG; }
        in.fill(nth, input_of);
    }
    wide_targets();
    return MTG_OK;
}

/* runs a marshalled batch; gaps: the caller's array (multi-contig gaps look at their dictionary) */
struct WireReq { /* the batch also in relocatable form, in a device buffer of the caller */
    void* dev = nullptr;
    uint64_t cap = 0, tag = 0;
    uint64_t* bytes = nullptr;
};
/* mtg_fill_text_formatted: the names of the sites (into the caller's text block) and the object that receives the text */
struct FormatReq {
    const uint64_t* name_off;
    const uint32_t* name_len;
    mtg_formatted* out;
};
int fill_marshalled(const mtg_index* idx, const mtg_params* p, const mtgi::FillInput& in, const mtg_gap* gaps, size_t n, char* seq_out, uint64_t seq_cap, uint64_t* seq_bytes,
                    mtg_results** out, double t_begin, char* d_seq_out = nullptr, const WireReq* wire = nullptr, const mtg_text_gaps* tg = nullptr, const FormatReq* fmt = nullptr)
{
    using namespace mtgi;
    const bool dbg = tune::on(tune::T_DEBUG_TIMERS);
    double tk = now_ms();
    auto tick = [&](const char* what) { if (dbg) { const double t = now_ms(); fprintf(stderr, "  [fill_batch] %-21s %.2f ms\n", what, t - tk); tk = t; } };
    mtg_results* R = results_acquire();
    struct Guard { mtg_results* r; ~Guard() { if (r) results_release(r); } } guard{R};
    R->nthreads = p->nb_host_threads;
    if (!R->ensure(n, std::max<size_t>(n * 64, 1 << 16))) { set_error("no page-locked memory for the results of %zu gaps", n); return MTG_ERR_NOMEM; }
    if (!R->ext) { R->ext_cap = 1 << 16; R->ext = (char*)pinned_alloc(R->ext_cap); if (!R->ext) { R->ext_cap = 0; set_error("no page-locked memory"); return MTG_ERR_NOMEM; } }
    R->ext[0] = 0;
    R->seq_external = seq_out != nullptr;
    /* d_seq_out: the arena is produced in this device buffer of the caller; seq_out == d_seq_out: and stays there only */
    R->seq_on_device = d_seq_out != nullptr && seq_out == d_seq_out;
    if (seq_out) { R->seq = seq_out; R->seq_cap = (size_t)seq_cap; }
    else {
        R->seq = R->seq_own;
        R->seq_cap = R->seq_own_cap;
    }
    ResultSink sink;
    sink.n = n;
    sink.res = R->res; sink.fil = R->fil;
    sink.seq = R->seq; sink.seq_cap = R->seq_cap;
    sink.seq_on_device = R->seq_on_device;
    sink.seq_dev = d_seq_out;
    sink.ext = R->ext; sink.ext_cap = R->ext_cap;
    if (wire) { sink.wire_dev = wire->dev; sink.wire_cap = wire->cap; sink.wire_tag = wire->tag; }
    sink.seq_stays_in_workspace = fmt != nullptr && !seq_out; /* the text is formatted on the device: the ASCII arena does not come to the host */
    if (!seq_out && !d_seq_out && !wire && !fmt) { sink.combo = R->blk; sink.combo_off_fil = mtg_results::off_fil(n); sink.combo_off_seq = mtg_results::off_seq(n); }
    sink.grow_seq = [&](size_t need, size_t keep) -> bool {
        if (R->seq_external) return false;
        if (!R->grow(need, keep)) return false; /* the whole block moves: records, filled records, arena */
        R->seq = R->seq_own; R->seq_cap = R->seq_own_cap;
        sink.res = R->res; sink.fil = R->fil;
        sink.seq = R->seq; sink.seq_cap = R->seq_cap;
        if (sink.combo) sink.combo = R->blk;
        return true;
    };
    sink.grow_ext = [&](size_t need, size_t keep) -> bool {
        const size_t cap = need + need / 4 + 4096;
        char* nb = (char*)pinned_alloc(cap);
        if (!nb) return false;
        memcpy(nb, R->ext, std::max<size_t>(keep, 1));
        pinned_free(R->ext);
        R->ext = nb; R->ext_cap = cap;
        sink.ext = nb; sink.ext_cap = cap;
        return true;
    };
    mtg_batch_stats st{};
    st.host_ms = now_ms() - t_begin;
    DevBatch special;
    int rc = device_run(idx, p, in, sink, special, &st);
    if (rc) return rc;
    tick("device");
    R->seq_bytes = sink.seq_used;
    R->n_gaps_filled = sink.n_filled;
    R->in_gap_order = sink.in_gap_order;
    /* multi-contig gaps: solutions on the host, records rewritten */
    if (!special.special.empty()) {
        const double t0 = now_ms();
        const auto describe = [&](size_t gi, GapWork& w) {
            if (tg) { /* mtg_fill_text: the strings of the gap are pieces of the caller's block */
                w.source = std::string_view(tg->text + tg->source_off[gi], tg->source_len[gi]);
                const uint8_t fl = tg->gap_flags ? tg->gap_flags[gi] : 0;
                w.anchor_repeated = (fl & 1) != 0;
                w.reverse = (fl & 2) != 0;
                const uint32_t t0 = tg->dict_first[gi], t1 = tg->dict_first[gi + 1];
                w.target_store.resize(t1 - t0);
                for (uint32_t t = t0; t < t1; t++) {
                    Target& T = w.target_store[t - t0];
                    T.seq = std::string_view(tg->text + tg->dict_seq_off[t], tg->dict_seq_len[t]);
                    T.name = std::string_view(tg->text + tg->dict_name_off[t], tg->dict_name_len[t]);
                    T.is_rc = tg->dict_is_rc ? tg->dict_is_rc[t] != 0 : false;
                }
                w.targets.p = w.target_store.data();
                w.targets.n = (uint32_t)w.target_store.size();
                return;
            }
            const mtg_gap& a = gaps[gi];
            w.source = std::string_view(a.source, strlen(a.source));
            w.anchor_repeated = a.is_anchor_repeated != 0;
            w.reverse = a.reverse != 0;
            w.target_store.resize((size_t)std::max(a.n_targets, 0));
            for (int t = 0; t < a.n_targets; t++) {
                Target& T = w.target_store[(size_t)t];
                T.seq = a.target_seqs[t];
                T.name = a.target_names[t];
                T.is_rc = a.target_is_rc ? a.target_is_rc[t] != 0 : false;
            }
            w.targets.p = w.target_store.data();
            w.targets.n = (uint32_t)w.target_store.size();
        };
        rc = run_general(idx, p, special, describe, R->gen, in.ws);
        if (rc == MTG_INTERNAL_RETRY_HOST_GENERAL) {
            /* once more, every multi-contig gap by the host's path (its contigs come along): this attempt's result object goes back to the pool */
            struct Flag { Flag() { tl_host_general = true; } ~Flag() { tl_host_general = false; } } flag;
            return fill_marshalled(idx, p, in, gaps, n, seq_out, seq_cap, seq_bytes, out, t_begin, d_seq_out, wire, tg, fmt);
        }
        if (rc) return rc;
        size_t total = 0;
        for (const GapWork& w : R->gen) total += w.sols.size();
        R->gen_fil.resize(total);
        size_t o = 0;
        for (size_t i = 0; i < R->gen.size(); i++) {
            const GapWork& w = R->gen[i];
            mtg_gap_result& r = R->res[special.special[i].gap];
            r.has_solution_counts = w.has_counts;
            r.nb_total_filled = w.nb_total_filled;
            r.nb_reported = r.n_filled = (int)w.sols.size();
            r.filled = R->gen_fil.data() + o;
            for (const Solution& s : w.sols) {
                mtg_filled& f = R->gen_fil[o++];
                f.seq = s.seq.c_str();
                f.nb_errors_in_anchor = s.nb_errors;
                f.target_index = s.target;
                f.avg_coverage = s.avg;
                f.median_coverage = s.median;
                f.qual = s.qual;
                f.solution_count = s.count;
                f.solution_rank = s.rank;
                R->seq_bytes += s.seq.size() + 1;
            }
            if (!w.sols.empty()) { R->n_gaps_filled++; R->in_gap_order = false; }
        }
        st.host_ms += now_ms() - t0;
        tick("multi-contig gaps");
    }
    if (seq_out) {
        if (R->in_gap_order) *seq_bytes = sink.seq_used; /* every sequence was written in place, in gap order */
        else {
            /* multi-contig gaps or re-run gaps: lay the sequences out again, in gap order, and make the records point there.  A buffer in
             * device memory comes to the host for that and goes back (rare: a batch with such a gap) */
            if (R->seq_bytes > seq_cap) { set_error("sequence buffer too small: %llu bytes needed", (unsigned long long)R->seq_bytes); return MTG_ERR_ARG; }
            std::vector<char> arena;
            const uintptr_t d_lo = (uintptr_t)seq_out, d_hi = d_lo + sink.seq_used;
            if (R->seq_on_device) {
                arena.resize(sink.seq_used + 1);
                if (int drc = device_download(idx, arena.data(), seq_out, sink.seq_used)) return drc;
            }
            R->relaid.resize(R->seq_bytes + 1);
            uint64_t o = 0;
            for (size_t i = 0; i < n; i++)
                for (int j = 0; j < R->res[i].n_filled; j++) {
                    mtg_filled& f = const_cast<mtg_filled&>(R->res[i].filled[j]);
                    const uintptr_t q = (uintptr_t)f.seq;
                    const char* src = (R->seq_on_device && q >= d_lo && q < d_hi) ? arena.data() + (q - d_lo) : f.seq; /* the others are host strings of the multi-contig path */
                    const size_t len = strlen(src);
                    memcpy(R->relaid.data() + o, src, len + 1);
                    f.seq = seq_out + o;
                    o += len + 1;
                }
            if (R->seq_on_device) { if (int urc = device_upload(idx, seq_out, R->relaid.data(), o)) return urc; }
            else {
                memcpy(seq_out, R->relaid.data(), o);
                if (d_seq_out) { if (int urc = device_upload(idx, d_seq_out, R->relaid.data(), o)) return urc; } /* the device copy follows */
            }
            *seq_bytes = o;
            R->in_gap_order = true;
        }
    }
    if (wire) {
        if (sink.wire_ok && special.special.empty()) *wire->bytes = sink.wire_bytes; /* the result kernel has written it */
        else {
            /* several launches, re-run gaps or multi-contig gaps: the finished result set is serialised here and sent up (rare) */
            uint64_t need = 0;
            if (int wrc = mtg_results_wire_size(R, &need)) return wrc;
            if (need > wire->cap) { set_error("wire buffer too small: %llu bytes needed", (unsigned long long)need); return MTG_ERR_ARG; }
            std::vector<uint64_t> tmp(need / 8 + 1);
            if (int wrc = mtg_results_to_wire(R, wire->tag, tmp.data(), need, &need)) return wrc;
            if (int urc = device_upload(idx, wire->dev, tmp.data(), need)) return urc;
            *wire->bytes = need;
        }
    }
    if (fmt && tg) {
        /* the tool's text for the sites with one solution of the common path, on the device (mtg_format.h); the records and the arena are where
         * device_run left them on this batch's workspace */
        FormatIn fi;
        fi.ws = in.ws; fi.n = n; fi.nt = n ? tg->dict_first[n] : 0;
        fi.res = R->res; fi.fil = R->fil;
        fi.device_records_whole = sink.device_records_whole && special.special.empty();
        fi.host_seq = R->seq; fi.seq_used = sink.seq_used;
        fi.host_text = tg->text; fi.source_off = tg->source_off; fi.source_len = tg->source_len;
        fi.name_off = fmt->name_off; fi.name_len = fmt->name_len;
        if (int frc = format_run(idx, fi, fmt->out->o)) return frc;
        /* a site left to the caller whose sequence lies in the arena: the arena comes to the host after all (rare: a value the formatter does not cover) */
        bool need_arena = false;
        const uintptr_t a_lo = (uintptr_t)R->seq, a_hi = a_lo + sink.seq_used;
        for (uint32_t ci : fmt->out->o.complex_sites)
            for (int j = 0; j < R->res[ci].n_filled; j++) { const uintptr_t q = (uintptr_t)R->res[ci].filled[j].seq; if (q >= a_lo && q < a_hi) need_arena = true; }
        if (need_arena && sink.seq_stays_in_workspace) { if (int drc = workspace_arena_download(idx, in.ws, R->seq, sink.seq_used)) return drc; }
    }
    st.total_ms = now_ms() - t_begin;
    stats_store(st);
    guard.r = nullptr;
    *out = R;
    return MTG_OK;
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
int mtg_index_build_profile(const mtg_index* idx, mtg_build_phase* out, size_t cap, size_t* n, uint64_t* peak_device_bytes, double* total_ms)
{
    if (!idx) { mtgi::set_error("null argument"); return MTG_ERR_ARG; }
    if (n) *n = idx->build_phases.size();
    if (out) for (size_t i = 0; i < cap && i < idx->build_phases.size(); i++) out[i] = idx->build_phases[i];
    if (peak_device_bytes) *peak_device_bytes = idx->build_peak_bytes;
    if (total_ms) *total_ms = idx->build_total_ms;
    return MTG_OK;
}
int mtg_index_save(const mtg_index* idx, const char* path) { return mtgi::index_save(idx, path); }
int mtg_index_load(const char* path, mtg_index** out) { return mtgi::index_load(path, out); }

static int fill_batch_impl(const mtg_index* idx, const mtg_params* p, const mtg_gap* gaps, size_t n, char* seq_out, uint64_t seq_cap, uint64_t* seq_bytes, mtg_results** out)
{
    if (!idx || !p || !out || (n && !gaps)) { mtgi::set_error("null argument"); return MTG_ERR_ARG; }
    const double t_begin = mtgi::now_ms();
    /* one workspace of the index (staging blocks, device buffers, streams) holds this batch from here until its results are on the host */
    mtgi::WorkspaceLock batch_lock = mtgi::acquire_workspace(idx);
    mtgi::FillInput in;
    in.k = idx->dev.k;
    in.ws = batch_lock.ws;
    if (int rc = marshal_gaps(gaps, n, p, in)) return rc;
    return fill_marshalled(idx, p, in, gaps, n, seq_out, seq_cap, seq_bytes, out, t_begin);
}
/* the ranges callers have page-locked for mtg_fill_text (mtg_host_register): few, looked up once per batch */
namespace {
std::mutex g_reg_mtx;
std::vector<std::pair<uintptr_t, size_t>> g_reg;
bool registered_range(const void* p, size_t bytes)
{
    const uintptr_t a = (uintptr_t)p;
    std::lock_guard<std::mutex> lk(g_reg_mtx);
    for (const auto& r : g_reg) if (a >= r.first && a + bytes <= r.first + r.second) return true;
    return false;
}
} // namespace
/* mtg_fill_text: what the host does for a batch whose strings are still text -- the integer columns of block A from the caller's arrays
 * (no string is looked at), the offset arrays and the block itself copied into the page-locked text block; the device does the rest */
static int fill_text_impl(const mtg_index* idx, const mtg_params* p, const mtg_text_gaps* g, char* seq_out, uint64_t seq_cap, uint64_t* seq_bytes, mtg_results** out, const FormatReq* fmt = nullptr)
{
    using namespace mtgi;
    if (!idx || !p || !out || !g) { set_error("null argument"); return MTG_ERR_ARG; }
    const size_t n = (size_t)g->n;
    if (n && (!g->text || !g->source_off || !g->source_len || !g->pattern_off || !g->pattern_len || !g->dict_first)) { set_error("null argument"); return MTG_ERR_ARG; }
    const size_t nt = n ? g->dict_first[n] : 0;
    if (nt && (!g->dict_seq_off || !g->dict_seq_len || !g->dict_name_off || !g->dict_name_len)) { set_error("null argument"); return MTG_ERR_ARG; }
    if (n >> 31 || nt >> 31 || g->text_bytes >> 40) { set_error("batch too large"); return MTG_ERR_ARG; }
    const double t_begin = now_ms();
    WorkspaceLock batch_lock = acquire_workspace(idx);
    FillInput in;
    in.k = idx->dev.k;
    in.ws = batch_lock.ws;
    in.text_mode = true;
    const int k = in.k;
    in.resize(n);
    /* one pass over the arrays: bounds, offsets of the patterns and of the dictionaries */
    const uint64_t tb = g->text_bytes;
    uint64_t rw = 0;
    long bad = -1, shortg = -1, order = -1;
    for (size_t i = 0; i < n; i++) {
        const uint32_t sl = g->source_len[i], pl = g->pattern_len[i];
        if (g->source_off[i] > tb || sl > tb - g->source_off[i] || g->pattern_off[i] > tb || pl > tb - g->pattern_off[i]) bad = (long)i;
        if ((int)sl < k) shortg = (long)i;
        if (g->dict_first[i + 1] < g->dict_first[i]) order = (long)i;
        const uint8_t fl = g->gap_flags ? g->gap_flags[i] : 0;
        in.roff[i] = (uint32_t)rw;
        in.rlen[i] = pl;
        rw += (pl + 31u) / 32u + 1u;
        in.toff[i] = g->dict_first[i];
        in.tcnt[i] = g->dict_first[i + 1] - g->dict_first[i];
        in.nbmis[i] = (uint8_t)((fl & 1) ? 0 : p->nb_mis_allowed); /* src/Filler.cpp:859-863 */
        in.flags[i] = (uint8_t)(((fl & 1) ? mtg::GAPF_REPEATED : 0) | ((fl & 2) ? mtg::GAPF_REVERSE : 0));
    }
    if (order >= 0) { set_error("gap %ld: dict_first is not ascending", order); return MTG_ERR_ARG; }
    for (size_t t = 0; t < nt; t++)
        if (g->dict_seq_off[t] > tb || g->dict_seq_len[t] > tb - g->dict_seq_off[t] || g->dict_name_off[t] > tb || g->dict_name_len[t] > tb - g->dict_name_off[t]) bad = (long)n;
    if (bad >= 0) { set_error(bad == (long)n ? "a dictionary entry lies outside the text block" : "gap %ld: a string lies outside the text block", bad); return MTG_ERR_ARG; }
    if (shortg >= 0) { set_error("gap %ld: source sequence shorter than k", shortg); return MTG_ERR_ARG; }
    if (rw >> 31) { set_error("batch too large"); return MTG_ERR_ARG; }
    in.n_rwords = rw;
    in.n_text_targets = nt;
    in.text_bytes = tb;
    in.bytes_b = 8 * rw + 64;
    in.bytes_c = FillInput::text_block_off(n, nt, 5) + tb + 64;
    in.block_c = in.ws ? staging_host(in.ws, 2, in.bytes_c) : nullptr;
    if (!in.block_c) { in.own_c.resize(in.bytes_c / 8 + 1); in.block_c = in.own_c.data(); }
    uint8_t* c = (uint8_t*)in.block_c;
    if (n) {
        memcpy(c + FillInput::text_block_off(n, nt, 0), g->source_off, 8 * n);
        memcpy(c + FillInput::text_block_off(n, nt, 1), g->pattern_off, 8 * n);
        memcpy(c + FillInput::text_block_off(n, nt, 3), g->source_len, 4 * n);
    }
    if (nt) {
        memcpy(c + FillInput::text_block_off(n, nt, 2), g->dict_seq_off, 8 * nt);
        memcpy(c + FillInput::text_block_off(n, nt, 4), g->dict_seq_len, 4 * nt);
    }
    if (tb && registered_range(g->text, (size_t)tb)) in.text_direct = g->text; /* the caller has page-locked it (mtg_host_register): it goes up from where it is */
    else if (tb) { /* the block itself: 10-15 MB per 100 000 sites, a millisecond for one thread -- in pieces on the worker pool */
        uint8_t* dst = c + FillInput::text_block_off(n, nt, 5);
        const size_t piece = (size_t)1 << 20, npieces = ((size_t)tb + piece - 1) / piece;
        parallel_for(npieces, p->nb_host_threads, [&](size_t i) { memcpy(dst + i * piece, g->text + i * piece, std::min(piece, (size_t)tb - i * piece)); }, 1);
    }
    if (fmt) for (size_t i = 0; i < n; i++) if (fmt->name_off[i] > tb || fmt->name_len[i] > tb - fmt->name_off[i]) { set_error("gap %zu: the name lies outside the text block", i); return MTG_ERR_ARG; }
    return fill_marshalled(idx, p, in, nullptr, n, seq_out, seq_cap, seq_bytes, out, t_begin, nullptr, nullptr, g, fmt);
}
int mtg_fill_text_formatted(const mtg_index* idx, const mtg_params* p, const mtg_text_gaps* g, const uint64_t* name_off, const uint32_t* name_len, mtg_results** out, mtg_formatted** text)
{
    if (!text || (g && g->n && (!name_off || !name_len))) { mtgi::set_error("null argument"); return MTG_ERR_ARG; }
    if (!*text) *text = new mtg_formatted();
    FormatReq fr{name_off, name_len, *text};
    return fill_text_impl(idx, p, g, nullptr, 0, nullptr, out, &fr);
}
int mtg_formatted_get(const mtg_formatted* t, mtg_formatted_view* v)
{
    if (!t || !v) { mtgi::set_error("null argument"); return MTG_ERR_ARG; }
    for (int s = 0; s < 3; s++) { v->text[s] = t->o.text[s]; v->bytes[s] = t->o.bytes[s]; v->complex_off[s] = t->o.complex_off[s].data(); }
    v->n_sites = t->o.n; v->n_simple = t->o.n_simple;
    v->complex_sites = t->o.complex_sites.data();
    v->n_complex = t->o.complex_sites.size();
    v->kernel_ms = t->o.kernel_ms;
    return MTG_OK;
}
void mtg_formatted_free(mtg_formatted* t) { delete t; }
int mtg_fill_text(const mtg_index* idx, const mtg_params* p, const mtg_text_gaps* g, mtg_results** out) { return fill_text_impl(idx, p, g, nullptr, 0, nullptr, out); }
int mtg_host_register(void* p, size_t bytes)
{
    if (!p || !bytes) { mtgi::set_error("null argument"); return MTG_ERR_ARG; }
    const uintptr_t a = (uintptr_t)p;
    std::lock_guard<std::mutex> lk(g_reg_mtx); /* one critical section: the overlap check, the page-locking and the entry belong together */
    for (const auto& r : g_reg) if (a < r.first + r.second && r.first < a + bytes) { mtgi::set_error("the range overlaps a registered one"); return MTG_ERR_ARG; }
    if (int rc = mtgi::host_register(p, bytes)) return rc;
    g_reg.emplace_back(a, bytes);
    return MTG_OK;
}
int mtg_host_unregister(void* p)
{
    const uintptr_t a = (uintptr_t)p;
    {
        std::lock_guard<std::mutex> lk(g_reg_mtx);
        auto it = std::find_if(g_reg.begin(), g_reg.end(), [&](const std::pair<uintptr_t, size_t>& r) { return r.first == a; });
        if (it == g_reg.end()) { mtgi::set_error("%p was not registered", p); return MTG_ERR_ARG; }
        g_reg.erase(it);
    }
    return mtgi::host_unregister(p);
}
int mtg_fill_text_serial(const mtg_index* idx, const mtg_params* p, const mtg_text_gaps* g, char* seq_out, uint64_t cap, uint64_t* seq_bytes, mtg_results** out)
{
    if (!seq_out || !seq_bytes) { mtgi::set_error("null argument"); return MTG_ERR_ARG; }
    return fill_text_impl(idx, p, g, seq_out, cap, seq_bytes, out);
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

int mtg_batch_prepare(const mtg_index* idx, const mtg_params* p, const mtg_gap* gaps, size_t n, mtg_batch** out)
{
    if (!idx || !p || !out || (n && !gaps)) { mtgi::set_error("null argument"); return MTG_ERR_ARG; }
    std::unique_ptr<mtg_batch> b(new mtg_batch());
    b->in.k = idx->dev.k;
    b->in.ws = nullptr; /* its own storage: the batch outlives any workspace hold */
    b->gaps = gaps;
    b->n = n;
    b->device = idx->device;
    b->nb_mis_allowed = p->nb_mis_allowed;
    if (int rc = marshal_gaps(gaps, n, p, b->in)) return rc;
    if (n) { if (int rc = mtgi::batch_upload(idx, b->in)) return rc; }
    *out = b.release();
    return MTG_OK;
}
void mtg_batch_free(mtg_batch* b) { delete b; }
static int fill_prepared_impl(const mtg_index* idx, const mtg_params* p, const mtg_batch* b, char* seq_out, uint64_t cap, uint64_t* seq_bytes, mtg_results** out, char* d_seq_out = nullptr,
                              const WireReq* wire = nullptr)
{
    if (!idx || !p || !b || !out) { mtgi::set_error("null argument"); return MTG_ERR_ARG; }
    if (b->in.k != idx->dev.k) { mtgi::set_error("the batch was prepared for k = %d", b->in.k); return MTG_ERR_ARG; }
    if (b->device != idx->device) { mtgi::set_error("the batch was prepared on device %d, the index lives on device %d (prepare one batch per replica)", b->device, idx->device); return MTG_ERR_ARG; }
    if (b->nb_mis_allowed != p->nb_mis_allowed) { mtgi::set_error("the batch was prepared with nb_mis_allowed = %d", b->nb_mis_allowed); return MTG_ERR_ARG; }
    const double t_begin = mtgi::now_ms();
    mtgi::WorkspaceLock batch_lock = mtgi::acquire_workspace(idx);
    mtgi::FillInput in; /* views of the batch's blocks (host and device); the workspace is this call's */
    const mtgi::FillInput& s = b->in;
    in.k = s.k; in.want_all_contigs = s.want_all_contigs;
    in.src = s.src; in.r0 = s.r0; in.roff = s.roff; in.rlen = s.rlen; in.toff = s.toff; in.tcnt = s.tcnt; in.nbmis = s.nbmis; in.fast_ok = s.fast_ok; in.flags = s.flags;
    in.rwords = s.rwords; in.traw = s.traw;
    in.block_a = s.block_a; in.block_b = s.block_b; in.block_c = s.block_c; in.bytes_a = s.bytes_a; in.bytes_b = s.bytes_b; in.bytes_c = s.bytes_c;
    in.dev_a = s.dev_a; in.dev_b = s.dev_b; in.dev_tenc = s.dev_tenc;
    in.ws = batch_lock.ws;
    return fill_marshalled(idx, p, in, b->gaps, b->n, seq_out, cap, seq_bytes, out, t_begin, d_seq_out, wire);
}
int mtg_fill_prepared(const mtg_index* idx, const mtg_params* p, const mtg_batch* b, mtg_results** out) { return fill_prepared_impl(idx, p, b, nullptr, 0, nullptr, out); }
int mtg_fill_prepared_serial(const mtg_index* idx, const mtg_params* p, const mtg_batch* b, char* seq_out, uint64_t cap, uint64_t* seq_bytes, mtg_results** out)
{
    if (!seq_out || !seq_bytes) { mtgi::set_error("null argument"); return MTG_ERR_ARG; }
    return fill_prepared_impl(idx, p, b, seq_out, cap, seq_bytes, out);
}

int mtg_fill_prepared_serial_device(const mtg_index* idx, const mtg_params* p, const mtg_batch* b, char* d_seq_out, uint64_t cap, char* host_copy, uint64_t* seq_bytes, mtg_results** out)
{
    if (!d_seq_out || !seq_bytes) { mtgi::set_error("null argument"); return MTG_ERR_ARG; }
    return fill_prepared_impl(idx, p, b, host_copy ? host_copy : d_seq_out, cap, seq_bytes, out, d_seq_out);
}

int mtg_fill_prepared_wire_device(const mtg_index* idx, const mtg_params* p, const mtg_batch* b, uint64_t tag, void* d_wire, uint64_t cap, uint64_t* wire_bytes, mtg_results** out)
{
    if (!d_wire || !wire_bytes) { mtgi::set_error("null argument"); return MTG_ERR_ARG; }
    WireReq w;
    w.dev = d_wire; w.cap = cap; w.tag = tag; w.bytes = wire_bytes;
    return fill_prepared_impl(idx, p, b, nullptr, 0, nullptr, out, nullptr, &w);
}

const mtg_gap_result* mtg_results_get(const mtg_results* r, size_t i) { return (r && i < r->n) ? &r->res[i] : nullptr; }
void mtg_results_free(mtg_results* r)
{
    if (r && r->plain) { delete r; return; }
    if (r) results_release(r); /* its storage serves the next batch */
}
int mtg_results_summary(const mtg_results* r, uint32_t* n_filled, uint64_t* seq_bytes, uint64_t* n_gaps_filled)
{
    if (!r) return MTG_ERR_ARG;
    if (n_filled) {
        const size_t n = r->n, CH = 4096;
        mtgi::parallel_for((n + CH - 1) / CH, r->nthreads, [&](size_t c) {
            for (size_t i = c * CH; i < std::min(n, (c + 1) * CH); i++) n_filled[i] = (uint32_t)r->res[i].n_filled;
        }, 1);
    }
    if (seq_bytes) *seq_bytes = r->seq_bytes;
    if (n_gaps_filled) *n_gaps_filled = r->n_gaps_filled;
    return MTG_OK;
}
int mtg_results_copy_seqs(const mtg_results* r, char* dst, uint64_t cap)
{
    if (!r || !dst) return MTG_ERR_ARG;
    if (r->seq_bytes > cap) { mtgi::set_error("destination too small"); return MTG_ERR_ARG; }
    if (r->seq_on_device) { mtgi::set_error("the sequences of these results are in the caller's device buffer"); return MTG_ERR_ARG; }
    if (r->in_gap_order) {
        /* the arena is the concatenation already, with NULs where the line ends go */
        const size_t nb = (size_t)r->seq_bytes, CH = 1 << 16;
        const char* src = r->seq;
        mtgi::parallel_for((nb + CH - 1) / CH, r->nthreads, [&](size_t c) {
            const size_t e = std::min(nb, (c + 1) * CH);
            for (size_t i = c * CH; i < e; i++) { const char ch = src[i]; dst[i] = ch ? ch : '\n'; }
        }, 1);
        return MTG_OK;
    }
    uint64_t o = 0;
    for (size_t i = 0; i < r->n; i++)
        for (int j = 0; j < r->res[i].n_filled; j++) {
            const char* s = r->res[i].filled[j].seq;
            const size_t len = strlen(s);
            memcpy(dst + o, s, len);
            o += len;
            dst[o++] = '\n';
        }
    return MTG_OK;
}
static void wire_counts(const mtg_results* r, uint64_t& n_filled, uint64_t& seq_bytes, uint64_t& ext_bytes)
{
    n_filled = 0; seq_bytes = 0; ext_bytes = 1;
    for (size_t i = 0; i < r->n; i++) {
        const mtg_gap_result& g = r->res[i];
        n_filled += (uint64_t)g.n_filled;
        if (g.extension && g.extension[0]) ext_bytes += strlen(g.extension) + 1;
    }
    seq_bytes = r->seq_bytes;
}
int mtg_results_wire_size(const mtg_results* r, uint64_t* bytes)
{
    if (!r || !bytes) return MTG_ERR_ARG;
    if (r->seq_on_device) { mtgi::set_error("the sequences of these results are in the caller's device buffer"); return MTG_ERR_ARG; }
    uint64_t nf, sb, eb;
    wire_counts(r, nf, sb, eb);
    *bytes = sizeof(mtg_wire_header) + up8(r->n * sizeof(mtg_wire_gap)) + up8(nf * sizeof(mtg_wire_filled)) + up8(sb) + up8(eb);
    return MTG_OK;
}
int mtg_results_to_wire(const mtg_results* r, uint64_t tag, void* dst, uint64_t cap, uint64_t* bytes)
{
    if (!r || !dst || !bytes) return MTG_ERR_ARG;
    uint64_t total = 0;
    if (int rc = mtg_results_wire_size(r, &total)) return rc;
    if (total > cap) { mtgi::set_error("wire buffer too small: %llu bytes needed", (unsigned long long)total); return MTG_ERR_ARG; }
    uint64_t nf, sb, eb;
    wire_counts(r, nf, sb, eb);
    uint8_t* b = (uint8_t*)dst;
    mtg_wire_header* h = (mtg_wire_header*)b;
    mtg_wire_gap* wg = (mtg_wire_gap*)(b + sizeof(mtg_wire_header));
    mtg_wire_filled* wf = (mtg_wire_filled*)((uint8_t*)wg + up8(r->n * sizeof(mtg_wire_gap)));
    char* ws = (char*)wf + up8(nf * sizeof(mtg_wire_filled));
    char* we = ws + up8(sb);
    const size_t n = r->n, CH = 4096, nch = (n + CH - 1) / CH;
    /* where every gap's solutions, sequences and extension go: prefix sums over pieces of gaps */
    std::vector<uint64_t> cf(nch + 1, 0), cs(nch + 1, 0), ce(nch + 1, 0);
    mtgi::parallel_for(nch, r->nthreads, [&](size_t c) {
        uint64_t f = 0, q = 0, e = 0;
        for (size_t i = c * CH; i < std::min(n, (c + 1) * CH); i++) {
            const mtg_gap_result& g = r->res[i];
            f += (uint64_t)g.n_filled;
            for (int j = 0; j < g.n_filled; j++) q += strlen(g.filled[j].seq) + 1;
            if (g.extension && g.extension[0]) e += strlen(g.extension) + 1;
        }
        cf[c + 1] = f; cs[c + 1] = q; ce[c + 1] = e;
    }, 1);
    ce[0] = 1; /* offset 0 of the extension section is the empty string */
    for (size_t c = 0; c < nch; c++) { cf[c + 1] += cf[c]; cs[c + 1] += cs[c]; ce[c + 1] += ce[c]; }
    if (cf[nch] != nf || cs[nch] != sb || ce[nch] != eb) { mtgi::set_error("result set changed while it was serialised"); return MTG_ERR_ARG; }
    we[0] = 0;
    mtgi::parallel_for(nch, r->nthreads, [&](size_t c) {
        uint64_t f = cf[c], q = cs[c], e = ce[c];
        for (size_t i = c * CH; i < std::min(n, (c + 1) * CH); i++) {
            const mtg_gap_result& g = r->res[i];
            mtg_wire_gap& o = wg[i];
            o.nb_nodes = g.nb_nodes; o.total_nt = g.total_nt; o.nb_terminal = g.nb_terminal; o.has_solution_counts = g.has_solution_counts;
            o.nb_total_filled = g.nb_total_filled; o.nb_reported = g.nb_reported; o.n_filled = g.n_filled;
            o.first_filled = (uint32_t)f;
            o.ext_off = 0;
            if (g.extension && g.extension[0]) { const size_t l = strlen(g.extension) + 1; memcpy(we + e, g.extension, l); o.ext_off = e; e += l; }
            for (int j = 0; j < g.n_filled; j++) {
                const mtg_filled& s = g.filled[j];
                mtg_wire_filled& w = wf[f++];
                const size_t l = strlen(s.seq);
                memcpy(ws + q, s.seq, l + 1);
                w.seq_off = q; w.seq_len = (uint32_t)l;
                w.nb_errors_in_anchor = s.nb_errors_in_anchor; w.target_index = s.target_index; w.qual = s.qual;
                w.solution_count = s.solution_count; w.solution_rank = s.solution_rank;
                w.avg_coverage = s.avg_coverage; w.median_coverage = s.median_coverage;
                q += l + 1;
            }
        }
    }, 1);
    /* the padding of the sections is part of the checksum: make it defined */
    for (uint64_t x = r->n * sizeof(mtg_wire_gap); x < up8(r->n * sizeof(mtg_wire_gap)); x++) ((uint8_t*)wg)[x] = 0;
    for (uint64_t x = nf * sizeof(mtg_wire_filled); x < up8(nf * sizeof(mtg_wire_filled)); x++) ((uint8_t*)wf)[x] = 0;
    for (uint64_t x = sb; x < up8(sb); x++) ws[x] = 0;
    for (uint64_t x = eb; x < up8(eb); x++) we[x] = 0;
    h->magic = WIRE_MAGIC; h->tag = tag; h->n_gaps = n; h->n_filled = nf; h->seq_bytes = sb; h->ext_bytes = eb; h->total_bytes = total;
    h->checksum = wire_checksum(b + sizeof(mtg_wire_header), total - sizeof(mtg_wire_header), r->nthreads);
    *bytes = total;
    return MTG_OK;
}
int mtg_results_from_wire(const void* wire, uint64_t bytes, mtg_results** out, uint64_t* tag)
{
    if (!wire || !out) return MTG_ERR_ARG;
    const uint8_t* b = (const uint8_t*)wire;
    const mtg_wire_header* h = (const mtg_wire_header*)b;
    if (bytes < sizeof(mtg_wire_header) || h->magic != WIRE_MAGIC || h->total_bytes > bytes || (h->total_bytes & 7) || h->ext_bytes < 1) { mtgi::set_error("not a result payload"); return MTG_ERR_FORMAT; }
    /* every section is bounded by the payload before anything is added up: sizes near 2^64 must not wrap the sum back onto total_bytes */
    const uint64_t tb = h->total_bytes;
    if (tb < sizeof(mtg_wire_header) || h->seq_bytes > tb || h->ext_bytes > tb || h->n_gaps > tb / sizeof(mtg_wire_gap) || h->n_filled > tb / sizeof(mtg_wire_filled)) { mtgi::set_error("result payload: inconsistent sizes"); return MTG_ERR_FORMAT; }
    const uint64_t want = sizeof(mtg_wire_header) + up8(h->n_gaps * sizeof(mtg_wire_gap)) + up8(h->n_filled * sizeof(mtg_wire_filled)) + up8(h->seq_bytes) + up8(h->ext_bytes);
    if (want != tb) { mtgi::set_error("result payload: inconsistent sizes"); return MTG_ERR_FORMAT; }
    if (wire_checksum(b + sizeof(mtg_wire_header), h->total_bytes - sizeof(mtg_wire_header), 0) != h->checksum) { mtgi::set_error("result payload: checksum mismatch"); return MTG_ERR_FORMAT; }
    const mtg_wire_gap* wg = (const mtg_wire_gap*)(b + sizeof(mtg_wire_header));
    const mtg_wire_filled* wf = (const mtg_wire_filled*)((const uint8_t*)wg + up8(h->n_gaps * sizeof(mtg_wire_gap)));
    const char* ws = (const char*)wf + up8(h->n_filled * sizeof(mtg_wire_filled));
    const char* we = ws + up8(h->seq_bytes);
    std::unique_ptr<mtg_results> R(new mtg_results());
    R->n = (size_t)h->n_gaps;
    R->plain = true;
    R->res = (mtg_gap_result*)malloc(std::max<size_t>(R->n, 1) * sizeof(mtg_gap_result));
    R->fil = (mtg_filled*)malloc(std::max<size_t>((size_t)h->n_filled, 1) * sizeof(mtg_filled));
    if (!R->res || !R->fil) { mtgi::set_error("out of memory"); return MTG_ERR_NOMEM; }
    R->seq = const_cast<char*>(ws); R->seq_external = true; R->seq_cap = (size_t)h->seq_bytes;
    R->seq_bytes = h->seq_bytes;
    R->in_gap_order = true;
    bool ok = true;
    for (uint64_t f = 0; f < h->n_filled; f++) {
        const mtg_wire_filled& w = wf[f];
        if (w.seq_off >= h->seq_bytes || w.seq_len >= h->seq_bytes - w.seq_off || ws[w.seq_off + w.seq_len] != 0) { ok = false; break; } /* no sum that could wrap */
        mtg_filled& s = R->fil[f];
        s.seq = ws + w.seq_off;
        s.nb_errors_in_anchor = w.nb_errors_in_anchor; s.target_index = w.target_index; s.avg_coverage = w.avg_coverage; s.median_coverage = w.median_coverage;
        s.qual = w.qual; s.solution_count = w.solution_count; s.solution_rank = w.solution_rank;
    }
    uint64_t filled_gaps = 0;
    for (uint64_t i = 0; i < h->n_gaps && ok; i++) {
        const mtg_wire_gap& g = wg[i];
        if (g.n_filled < 0 || (uint64_t)g.first_filled + (uint64_t)g.n_filled > h->n_filled || g.ext_off >= h->ext_bytes) { ok = false; break; }
        mtg_gap_result& o = R->res[i];
        o.nb_nodes = g.nb_nodes; o.total_nt = g.total_nt; o.nb_terminal = g.nb_terminal; o.has_solution_counts = g.has_solution_counts;
        o.nb_total_filled = g.nb_total_filled; o.nb_reported = g.nb_reported; o.n_filled = g.n_filled;
        o.filled = R->fil + g.first_filled;
        o.extension = we + g.ext_off;
        filled_gaps += g.n_filled > 0;
    }
    if (ok && we[h->ext_bytes - 1] != 0) ok = false;
    if (!ok) { mtgi::set_error("result payload: offsets out of range"); return MTG_ERR_FORMAT; }
    R->n_gaps_filled = filled_gaps;
    if (tag) *tag = h->tag;
    *out = R.release();
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
        if (!sources[i] || !targets[i]) { mtgi::set_error("gap %zu: null field", i); return MTG_ERR_ARG; }
        if ((int)strlen(sources[i]) < idx->dev.k) { mtgi::set_error("gap %zu: source sequence shorter than k", i); return MTG_ERR_ARG; }
        in.size(i, strlen(targets[i]), 0);
    }
    in.layout();
    for (size_t i = 0; i < n; i++) in.set_common(i, std::string_view(sources[i]), std::string_view(targets[i]), 0, 0);
    mtgi::ResultSink sink; /* no records: the contigs themselves come back */
    sink.n = n;
    mtgi::DevBatch special;
    mtg_batch_stats st{};
    int rc = mtgi::device_run(idx, p, in, sink, special, &st);
    if (rc) return rc;
    mtgi::stats_store(st);
    mtg_contigs* C = new mtg_contigs();
    C->c.resize(n);
    for (const mtgi::SpecialGap& sg : special.special) {
        const mtgi::GapDev gd = special.view(sg);
        for (uint32_t j = 0; j < gd.o.n_contigs; j++) C->c[sg.gap].push_back(gd.contig(j));
    }
    *out = C;
    return MTG_OK;
}
size_t mtg_contigs_count(const mtg_contigs* c, size_t gap) { return (c && gap < c->c.size()) ? c->c[gap].size() : 0; }
const char* mtg_contigs_get(const mtg_contigs* c, size_t gap, size_t i) { return (c && gap < c->c.size() && i < c->c[gap].size()) ? c->c[gap][i].c_str() : nullptr; }
void mtg_contigs_free(mtg_contigs* c) { delete c; }
}
