/*
 * mtg_internal.h -- interfaces between the translation units of libmtgfill.so
 *   mtg_gpu.hip   : HIP kernels, device memory, index construction, stage A batches
 *   mtg_host.cpp  : host orchestration of gapFillFromSource (contig graph, paths, dedupe, writers), CLI
 */
#ifndef MTG_INTERNAL_H
#define MTG_INTERNAL_H
#include "../../include/mtg_fill.h"
#include "mtg_hostutil.h"
#include "mtg_post.h"
#include <string>
#include <vector>

struct mtg_index {
    mtg::Index dev{};          /* tables live in device memory */
    int device = 0;
    mtg_index_info info{};
};

namespace mtgi {

void set_error(const char* fmt, ...);

/* one gapFillFromSource call and its results (host side) */
struct Target {
    std::string seq, name;
    bool is_rc = false;
    uint64_t code = 0;    /* 2-bit code of the first k chars */
    uint64_t badmask = 0; /* positions (pair-lsb) that can never match (not ACGT/acgt) */
    bool usable = true;   /* at least k chars */
};
/* what comes back from the device for one gap */
struct GapDev {
    mtg::GapOut o{};
    mtg::PostOut p{};
    /* contig data: all contigs when n_meta == o.n_contigs, otherwise only the leading words of contig 0 */
    uint32_t n_meta = 0;
    std::vector<uint64_t> words;
    std::vector<uint32_t> len, word_start, tpos, terr, ttgt;
    std::string contig(size_t i) const
    {
        std::string s;
        mtg::unpack_seq(words.data() + word_start[i], len[i], s);
        return s;
    }
    /* contig0[from, to) from the leading words */
    std::string contig0_slice(uint32_t from, uint32_t to) const
    {
        std::string s;
        if (to <= from) return s;
        static const char NT[4] = {'A', 'C', 'T', 'G'};
        s.resize(to - from);
        for (uint32_t i = from; i < to; i++) s[i - from] = NT[(words[i >> 5] >> (2 * (i & 31))) & 3];
        return s;
    }
};

/* what to copy back for a gap: nw leading words of its arena, metadata of nc contigs (0 or all) */
inline void copy_plan(const mtg::GapOut& o, const mtg::PostOut& p, bool want_all, uint32_t& nw, uint32_t& nc)
{
    nw = nc = 0;
    if (o.status != mtg::GAP_OK) return;
    if (want_all || (p.fast == 0 && p.nb_terminal > 0)) { nw = o.n_words; nc = o.n_contigs; }
    else if (p.fast == 1) nw = (p.pos + 31) / 32;
    else if (p.fast == 2) nw = 0;
    else nw = (p.clen0 + 31) / 32; /* no terminal node: contig 0 is the extension sequence */
}

/* a batch of gapFillFromSource calls, marshalled for the device */
struct FillInput {
    int k = 31;
    bool want_all_contigs = false;
    std::vector<uint64_t> src;     /* oriented source k-mer per gap */
    std::vector<uint64_t> rwords;  /* packed swf patterns, concatenated */
    std::vector<uint32_t> roff;    /* first word of gap i's pattern */
    std::vector<uint32_t> rlen;    /* pattern length in nt */
    std::vector<uint64_t> r0;      /* first k-mer of the pattern */
    std::vector<uint64_t> tle, tbad; /* targets of all gaps: little-endian k-mer, never-match mask */
    std::vector<uint32_t> toff, tcnt;
    std::vector<uint8_t> nbmis, fast_ok;
    void add(const std::string& source, const std::string& swf_target, const std::vector<Target>* targets, int nb_mis);
};

/* stage A + post-processing kernels for all gaps (chunked, tiered); fills out[i]; returns MTG_* status */
int device_run(const mtg_index* idx, const mtg_params* p, const FillInput& in, std::vector<GapDev>& out, mtg_batch_stats* stats);

int query_run(const mtg_index* idx, const uint64_t* kmers, size_t n, uint32_t* abund, uint8_t* succ, uint8_t* pred);

void stats_store(const mtg_batch_stats& s);

struct Solution { /* filled_insertion_t, src/Utils.hpp:46-104 */
    std::string seq;
    int nb_errors = 0;
    int target = -1;
    float avg = 0, median = 0;
    int qual = 0, count = 0, rank = 0;
    size_t ab_off = 0, ab_n = 0; /* slice of the batched abundance query */
};
struct GapWork {
    std::vector<Target> targets; /* targetDictionary in iteration order */
    std::string source;
    bool anchor_repeated = false, reverse = false;
    int nb_nodes = 0, total_nt = 0, nb_terminal = 0, nb_total_filled = 0;
    bool has_counts = false;
    std::vector<Solution> sols;
    std::string extension;
};
bool read_sequences(const std::string& path, std::vector<std::pair<std::string, std::string>>& out);
int fill_gaps(const mtg_index* idx, const mtg_params* p, std::vector<GapWork>& gaps, const std::vector<std::string>& swf_targets, mtg_batch_stats* stats_out);
int index_from_kmers(const uint64_t*, const uint32_t*, size_t, int, mtg_index**);
int index_from_reads(const char*, int, int, int, mtg_index**);
int index_save(const mtg_index*, const char*);
int index_load(const char*, mtg_index**);
void index_forget_host_copy(const mtg_index* idx);
int fill_main(int argc, const char* const* argv);

} // namespace mtgi
#endif
