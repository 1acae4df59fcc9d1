/*
 * mtg_cli.cpp -- `MindTheGap fill` front end of libmtgfill.so: option parsing, the two drivers and the writers.
 *
 *   options / checks     Filler::Filler, Filler::execute          /root/reference/src/Filler.cpp:76-113,136-165,231-317
 *   -bkpt driver         breakpointFunctor::operator()            src/Filler.cpp:623-699
 *   -contig driver       fillAny (contig part) + contigFunctor    src/Filler.cpp:755-829, 492-572
 *   writers              writeFilledBreakpoint / writeVcf / writeToGFA / writeExtensions / writeVcfHeader
 *                        src/Filler.cpp:1029-1093, 1095-1214, 1216-1273, 1275-1291, 349-383
 * Records are written in input order, which is the reference's order with -nb-cores 1.
 */
#include "mtg_internal.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <fstream>
#include <sstream>
#include <unordered_map>
#include "mtg_dict_order.h"
#include <condition_variable>
#include <thread>
#include <zlib.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cerrno>
#include <deque>
#include <atomic>
#include <chrono>

namespace mtgi {

static const char* MTG_VERSION = "2.3.0";

typedef std::pair<std::string, bool> bkpt_t;
typedef std::unordered_map<std::string, bkpt_t> bkpt_dict_t; /* src/Utils.hpp:43-44; iteration order is part of the output */

static std::string short_name(const std::string& c) { size_t p = c.find(' '); return p == std::string::npos ? c : c.substr(0, p); }
static std::string revcomp_str(const std::string& s)
{
    std::string r;
    for (auto it = s.rbegin(); it != s.rend(); ++it) {
        switch (*it) {
            case 'a': r += 't'; break; case 't': r += 'a'; break; case 'c': r += 'g'; break; case 'g': r += 'c'; break;
            case 'A': r += 'T'; break; case 'T': r += 'A'; break; case 'C': r += 'G'; break; case 'G': r += 'C'; break;
        }
    }
    return r;
}

/* one gapFillFromSource call as the C ABI wants it; the strings and arrays it points to live in the GapArgs */
struct GapArgs {
    std::string source, target; /* sourceSequence; targetSequence (the early-stop pattern) */
    std::vector<std::string> tseq, tname; /* targetDictionary in iteration order */
    std::vector<uint8_t> trc;
    std::vector<const char*> pseq, pname;
    std::vector<uint32_t> tidx; /* contig mode: the dictionary's entries as numbers into the job's table of targets (pseq / pname / trc are filled from it: no string is copied per seed) */
    bool shared_dict = false;
    bool repeated = false, reverse = false;
    void set_dict(const bkpt_dict_t& dict)
    {
        for (auto it = dict.begin(); it != dict.end(); ++it) { tseq.push_back(it->first); tname.push_back(it->second.first); trc.push_back(it->second.second ? 1 : 0); }
    }
    mtg_gap abi()
    {
        if (!shared_dict) {
            pseq.clear(); pname.clear();
            for (size_t i = 0; i < tseq.size(); i++) { pseq.push_back(tseq[i].c_str()); pname.push_back(tname[i].c_str()); }
        }
        mtg_gap g;
        g.source = source.c_str();
        g.target = target.c_str();
        g.n_targets = (int)pseq.size();
        g.target_seqs = pseq.data();
        g.target_names = pname.data();
        g.target_is_rc = trc.data();
        g.is_anchor_repeated = repeated;
        g.reverse = reverse;
        return g;
    }
};
/* runs a batch through the C ABI; the results stay valid until the handle is freed */
struct BatchRun {
    mtg_results* h = nullptr;
    ~BatchRun() { if (h) mtg_results_free(h); }
    int run(const mtg_index* idx, const mtg_params& P, std::vector<GapArgs>& args)
    {
        std::vector<mtg_gap> gaps(args.size());
        for (size_t i = 0; i < args.size(); i++) gaps[i] = args[i].abi();
        return mtg_fill_batch(idx, &P, gaps.data(), gaps.size(), &h);
    }
    const mtg_gap_result& operator[](size_t i) const { return *mtg_results_get(h, i); }
};

static void put_int(std::string& o, long long v);
static std::string info_string(const mtg_gap_result& g) /* "\t%i\t%i\t%d" [ "\t%d\t%d" ] */
{
    std::string s;
    s += '\t'; put_int(s, g.nb_nodes); s += '\t'; put_int(s, g.total_nt); s += '\t'; put_int(s, g.nb_terminal);
    if (g.nb_terminal > 0 && g.has_solution_counts) { s += '\t'; put_int(s, g.nb_total_filled); s += '\t'; put_int(s, g.nb_reported); }
    return s;
}

struct Options {
    std::string in, graph, bkpt, contig, out;
    int k = 31, abundance_min = -1, abundance_max = 0, max_nodes = 100, max_depth = 10000, overlap = 0, nb_cores = 0, nb_gpus = 0;
    bool fwd_only = false, filter = false, extend = false, has_out = false;
};

struct Files {
    FILE *insert = nullptr, *info = nullptr, *vcf = nullptr, *gfa = nullptr, *ext = nullptr;
    ~Files() { for (FILE* f : {insert, info, vcf, gfa, ext}) if (f) fclose(f); }
    FILE* stream(int i) const { return i == 0 ? insert : i == 1 ? info : i == 2 ? vcf : i == 3 ? gfa : ext; }
};
/* the text a run of sites adds to the output files: formatted by whoever has the records (several threads, one piece each), written by the
 * one thread that owns the files, in input order */
struct OutText {
    std::string insert, info, vcf, gfa, ext;
    void clear() { insert.clear(); info.clear(); vcf.clear(); gfa.clear(); ext.clear(); } /* the capacities stay */
    void write(Files& F) const
    {
        if (F.insert && !insert.empty()) fwrite(insert.data(), 1, insert.size(), F.insert);
        if (F.info && !info.empty()) fwrite(info.data(), 1, info.size(), F.info);
        if (F.vcf && !vcf.empty()) fwrite(vcf.data(), 1, vcf.size(), F.vcf);
        if (F.gfa && !gfa.empty()) fwrite(gfa.data(), 1, gfa.size(), F.gfa);
        if (F.ext && !ext.empty()) fwrite(ext.data(), 1, ext.size(), F.ext);
    }
};
/* Buffers of a long run are used again: a batch's text (17 MB) and its formatted pieces (88 MB) in fresh memory cost a page fault per 4 KB,
 * which at 20 batches a second was most of the reader's and a third of the formatters' time.  Process-wide, bounded. */
template <class T> struct Recycler {
    std::mutex m;
    std::vector<T> free;
    size_t keep;
    explicit Recycler(size_t k) : keep(k) {}
    T get()
    {
        std::lock_guard<std::mutex> lk(m);
        if (free.empty()) return T();
        T t = std::move(free.back());
        free.pop_back();
        return t;
    }
    void put(T&& t)
    {
        t.clear();
        std::lock_guard<std::mutex> lk(m);
        if (free.size() < keep) free.push_back(std::move(t));
    }
};
static Recycler<OutText>& piece_pool() { static Recycler<OutText> p(512); return p; }
/* the library's formatted-text objects (three page-locked arenas each) go round as the batches do */
struct FormattedPool {
    std::mutex m;
    std::vector<mtg_formatted*> free;
    mtg_formatted* get() { std::lock_guard<std::mutex> lk(m); if (free.empty()) return nullptr; mtg_formatted* t = free.back(); free.pop_back(); return t; }
    void put(mtg_formatted* t) { if (!t) return; std::lock_guard<std::mutex> lk(m); if (free.size() < 12) free.push_back(t); else mtg_formatted_free(t); }
    ~FormattedPool() { for (mtg_formatted* t : free) mtg_formatted_free(t); }
};
static FormattedPool& ftext_pool() { static FormattedPool p; return p; }
/* The output files of a long run, written by their own threads: the pieces of text get their places in input order (one thread hands them
 * out, as it would have written them), the bytes go there with pwrite -- ONE writer thread per file.  write(2) on a file is serialised by
 * the file's lock: the memory-backed file system of the GPU box takes 8-9 GB/s for a file from one thread, the same from two, and on some
 * boxes HALF of it from four (scripts/tmpfs_write_ceiling.py: 8.1 / 7.7 / 3.7 / 3.5 GB/s with 1 / 2 / 4 / 8 threads; profiles/
 * r04_tmpfs_write_ceiling.txt) -- six writers taking whole pieces, as before, put three or four of them on the two large files at any time.
 * Different files do not share a lock: the FASTA and the VCF file (45 % and 52 % of the bytes) are written side by side at the rate of
 * one each.  Whatever was written through the FILE* before begin() stays in front; after finish() the FILE* continue behind. */
struct PositionedWriter {
    struct Span { const char* p = nullptr; size_t n = 0; };
    struct Item { std::shared_ptr<void> owner; Span s; off_t at; };
    Files& F;
    int fd[5];
    off_t pos[5];
    std::vector<std::thread> threads;
    std::mutex mtx;
    std::condition_variable cv_job[5], cv_room;
    std::deque<Item> q[5];
    size_t pending_bytes = 0;
    bool done = false;
    std::atomic<bool> failed{false};
    explicit PositionedWriter(Files& f) : F(f)
    {
        for (int i = 0; i < 5; i++) {
            FILE* s = F.stream(i);
            fd[i] = -1; pos[i] = 0;
            if (s) { fflush(s); fd[i] = fileno(s); pos[i] = ftello(s); }
        }
        const int per_file = std::max(1, (int)tune::i(tune::T_CLI_WRITERS, 1));
        for (int i = 0; i < 5; i++)
            if (fd[i] >= 0)
                for (int t = 0; t < per_file; t++) threads.emplace_back([this, i] { run(i); });
    }
    static const std::string& text_of(const OutText& T, int i) { return i == 0 ? T.insert : i == 1 ? T.info : i == 2 ? T.vcf : i == 3 ? T.gfa : T.ext; }
    /* called in input order: the piece's places are the current ends of the files */
    void add(std::shared_ptr<void> owner, const OutText& T)
    {
        Span sp[5];
        for (int i = 0; i < 5; i++) { sp[i].p = text_of(T, i).data(); sp[i].n = text_of(T, i).size(); }
        add_spans(std::move(owner), sp);
    }
    /* the same for bytes that live somewhere else (a batch's text formatted on the device: page-locked arenas of the library) */
    void add_spans(std::shared_ptr<void> owner, const Span sp[5])
    {
        size_t bytes = 0;
        for (int i = 0; i < 5; i++) if (fd[i] >= 0) bytes += sp[i].n;
        std::unique_lock<std::mutex> lk(mtx);
        cv_room.wait(lk, [&] { return pending_bytes < ((size_t)1 << 30); }); /* formatted text waiting for its writer: bounded */
        pending_bytes += bytes;
        for (int i = 0; i < 5; i++) {
            if (fd[i] < 0 || sp[i].n == 0) continue;
            Item it;
            it.owner = owner; /* the text lives until the last of its files has it */
            it.s = sp[i];
            it.at = pos[i];
            pos[i] += (off_t)sp[i].n;
            q[i].push_back(std::move(it));
            cv_job[i].notify_one();
        }
    }
    void run(int i)
    {
        for (;;) {
            Item it;
            {
                std::unique_lock<std::mutex> lk(mtx);
                cv_job[i].wait(lk, [&] { return done || !q[i].empty(); });
                if (q[i].empty()) return;
                it = std::move(q[i].front());
                q[i].pop_front();
            }
            size_t w = 0;
            while (w < it.s.n) {
                const ssize_t got = ::pwrite(fd[i], it.s.p + w, it.s.n - w, it.at + (off_t)w);
                if (got < 0) { if (errno == EINTR) continue; failed = true; break; }
                w += (size_t)got;
            }
            it.owner.reset();
            std::lock_guard<std::mutex> lk(mtx);
            pending_bytes -= it.s.n;
            cv_room.notify_all();
        }
    }
    /* true when every byte is in its file; the FILE* then continue behind what was written */
    bool finish()
    {
        { std::lock_guard<std::mutex> lk(mtx); done = true; }
        for (int i = 0; i < 5; i++) cv_job[i].notify_all();
        for (std::thread& t : threads) t.join();
        threads.clear();
        for (int i = 0; i < 5; i++) if (fd[i] >= 0) fseeko(F.stream(i), pos[i], SEEK_SET);
        return !failed;
    }
    ~PositionedWriter() { if (!threads.empty()) finish(); }
};
static void appendf(std::string& o, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
static void appendf(std::string& o, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    const int n = vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (n < 0) return;
    if ((size_t)n < sizeof buf) { o.append(buf, (size_t)n); return; }
    std::vector<char> big((size_t)n + 1);
    va_start(ap, fmt);
    vsnprintf(big.data(), big.size(), fmt, ap);
    va_end(ap);
    o.append(big.data(), (size_t)n);
}

/* the writers' numbers without printf: a decimal integer, and a double with two decimals exactly as "%.2f" prints it (round to nearest on
 * the exact binary value; the shortcut decides by x * 100 only when that is safely away from a tie, anything else goes to snprintf) */
static void put_int(std::string& o, long long v)
{
    char b[24];
    int n = 0;
    unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
    do { b[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) b[n++] = '-';
    while (n) o += b[--n];
}
static void put_fixed2(std::string& o, double x)
{
    if (x >= 0 && x < 1e9) {
        const double t = x * 100.0, fl = std::floor(t), fr = t - fl;
        if (std::fabs(fr - 0.5) > 1e-6) {
            const unsigned long long r = (unsigned long long)fl + (fr > 0.5 ? 1u : 0u);
            put_int(o, (long long)(r / 100));
            o += '.';
            o += (char)('0' + (r / 10) % 10);
            o += (char)('0' + r % 10);
            return;
        }
    }
    char b[400];
    const int n = snprintf(b, sizeof b, "%.2f", x);
    if (n > 0) o.append(b, std::min((size_t)n, sizeof b - 1));
}

struct Sols { /* a run of solutions */
    const mtg_filled* p = nullptr;
    size_t n = 0;
    const mtg_filled* begin() const { return p; }
    const mtg_filled* end() const { return p + n; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
};
static Sols sols_of(const mtg_gap_result& g) { Sols s; s.p = g.filled; s.n = (size_t)g.n_filled; return s; }

static std::string solu_str(const mtg_filled& s)
{
    if (s.solution_count <= 1) return "";
    char b[64];
    snprintf(b, sizeof b, "solution %d/%d", s.solution_rank, s.solution_count);
    return b;
}
/* the targetDictionary of a gap as the writers see it */
struct DictView {
    const std::string* tname = nullptr;
    const uint8_t* trc = nullptr;
    const uint32_t* idx = nullptr; /* non-null: entry t's name is tname[idx[t]] (contig mode: the names of the job's targets, once) */
    const std::string& name_of(int t) const { return tname[idx ? idx[t] : (uint32_t)t]; }
};

/* writeFilledBreakpoint, src/Filler.cpp:1029-1093.  The bkpt-mode header passes its arguments in a different order
 * than its format (:1052-1054); the visible x86-64 result is NAME_len_L_qual_Q_avg_cov_A_median_cov_M   SOLU. */
static void write_filled(OutText& F, bool bkpt_mode, const DictView& g, Sols sols, std::string_view seedName, const std::string& info)
{
    for (auto& s : sols) {
        const int llen = (int)strlen(s.seq);
        const std::string solu = solu_str(s);
        if (bkpt_mode) { /* ">%.*s_len_%d_qual_%i_avg_cov_%.2f_median_cov_%.2f   %s\n" */
            std::string& o = F.insert;
            o += '>'; o.append(seedName.data(), seedName.size());
            o += "_len_"; put_int(o, llen);
            o += "_qual_"; put_int(o, s.qual);
            o += "_avg_cov_"; put_fixed2(o, (double)s.avg_coverage);
            o += "_median_cov_"; put_fixed2(o, (double)s.median_coverage);
            o += "   "; o += solu; o += '\n';
        } else {
            std::string targetName(g.name_of(s.target_index));
            if (g.trc[s.target_index]) targetName.append("_Rc");
            int cov = s.median_coverage + 0.5;
            appendf(F.insert, ">%.*s;%s;len_%d_qual_%d_median_cov_%d\t%s\n", (int)seedName.size(), seedName.data(), targetName.c_str(), llen, s.qual, cov, solu.c_str());
        }
        F.insert.append(s.seq, (size_t)llen);
        F.insert += '\n';
    }
    F.info.append(seedName.data(), seedName.size());
    F.info += '\t';
    F.info += info;
    F.info += '\n';
}

/* writeVcf, src/Filler.cpp:1095-1214 */
static void write_vcf(OutText& F, bool filter, Sols sols, std::string_view breakpointName, std::string_view sourceSequence)
{
    for (auto& s : sols) {
        const size_t slen = strlen(s.seq), srcn = sourceSequence.size();
        int repeatSize = 0;
        int i = (int)srcn - 1, j = (int)slen - 1;
        while (i > 0 && j >= 0) { /* longest common suffix with a circular insert index, :1107-1126 */
            if (sourceSequence[(size_t)i] != s.seq[j]) break;
            repeatSize++; i--; j--;
            if (j == -1) j = (int)slen - 1;
        }
        /* insertion = the last repeatSize + 1 nucleotides of the source + the sequence, less its last repeatSize characters (:1147-1149):
         * the first slen + 1 characters of that concatenation */
        const size_t tail0 = srcn - (size_t)(repeatSize + 1), tail_n = (size_t)repeatSize + 1, ins_n = slen + 1;
        const char ref = sourceSequence[tail0];
        /* the name cut at its underscores (:1165-1182) */
        std::string_view tokens[9];
        size_t ntok = 0;
        {
            size_t b = 0;
            for (;;) {
                const size_t e = breakpointName.find('_', b);
                if (ntok < 9) tokens[ntok] = breakpointName.substr(b, e == std::string_view::npos ? std::string_view::npos : e - b);
                ntok++;
                if (e == std::string_view::npos) break;
                b = e + 1;
            }
            if (!breakpointName.empty() && breakpointName.back() == '_') ntok--; /* getline yields no empty last token */
        }
        auto to_int = [](std::string_view t) { char b[32]; const size_t n = std::min(t.size(), sizeof b - 1); memcpy(b, t.data(), n); b[n] = 0; return t.size() < sizeof b ? atoi(b) : atoi(std::string(t).c_str()); };
        const bool named = ntok == 7 || ntok == 8;
        const std::string_view genotype = ntok == 7 ? tokens[6] : ntok == 8 ? tokens[7] : std::string_view();
        const int nsol = s.solution_count, npos = repeatSize + 1;
        const char* filt = "PASS";
        if ((genotype == "HET" && nsol > 1) || (genotype == "HOM" && nsol > 1)) {
            if (filter) break;
            filt = "LOW_QUAL";
        }
        std::string& o = F.vcf; /* "%s\t%s\t%s\t%c\t" chromosome, position, bkpt, ref */
        if (named) o.append(tokens[1].data(), tokens[1].size()); else o += '.';
        o += '\t';
        if (named) put_int(o, (long long)to_int(tokens[ntok == 7 ? 3 : 4]) - repeatSize); else o += '.';
        o += '\t';
        if (!named) o.append(breakpointName.data(), breakpointName.size());
        else { o.append(tokens[0].data(), tokens[0].size()); if (ntok == 8) o.append(tokens[2].data(), tokens[2].size()); }
        o += '\t'; o += ref; o += '\t';
        {
            const size_t from_tail = std::min(tail_n, ins_n);
            o.append(sourceSequence.data() + tail0, from_tail);
            if (ins_n > from_tail) o.append(s.seq, ins_n - from_tail);
        }
        /* "\t.\t%s\tTYPE=INS;LEN=%i;QUAL=%i;NSOL=%i;NPOS=%i;AVK=%.2f;MDK=%.2f\tGT\t%s\n" */
        o += "\t.\t"; o += filt;
        o += "\tTYPE=INS;LEN="; put_int(o, (long long)ins_n - 1);
        o += ";QUAL="; put_int(o, s.qual);
        o += ";NSOL="; put_int(o, nsol);
        o += ";NPOS="; put_int(o, npos);
        o += ";AVK="; put_fixed2(o, (double)s.avg_coverage);
        o += ";MDK="; put_fixed2(o, (double)s.median_coverage);
        o += "\tGT\t"; o += !named ? "./." : genotype == "HOM" ? "1/1" : "0/1";
        o += '\n';
    }
}

/* writeToGFA, src/Filler.cpp:1216-1273 */
static void write_gfa(OutText& F, int trim, const DictView& g, Sols sols, std::string seedName, bool isRc)
{
    const std::string seedNameNode = seedName;
    std::string seedDirection = "+";
    if (isRc) { seedName = seedName.substr(0, seedName.size() - 3); seedDirection = "-"; }
    for (auto& s : sols) {
        const std::string tname(g.name_of(s.target_index));
        const bool trc = g.trc[s.target_index] != 0;
        const std::string targetNameNode = trc ? tname + "_Rc" : tname;
        int cov = s.median_coverage + 0.5;
        const std::string nodeName = seedNameNode + ";" + targetNameNode + ";len_" + std::to_string((int)strlen(s.seq)) + "_qual_" + std::to_string(s.qual) +
                                     "_median_cov_" + std::to_string(cov) + " " + solu_str(s);
        F.gfa += "S\t"; F.gfa += nodeName; F.gfa += '\t'; F.gfa += s.seq; F.gfa += '\n';
        appendf(F.gfa, "L\t%s\t%s\t%s\t+\t%iM\n", seedName.c_str(), seedDirection.c_str(), nodeName.c_str(), trim);
        appendf(F.gfa, "L\t%s\t+\t%s\t%s\t%iM\n", nodeName.c_str(), tname.c_str(), trc ? "-" : "+", trim);
    }
}

static void write_extension(OutText& F, const char* contigSeq, std::string_view seedName, const char* suffix, std::string_view sourceSequence) /* :1275-1291 */
{
    const size_t llen = contigSeq ? strlen(contigSeq) : 0;
    if (llen > 0) {
        appendf(F.ext, ">%.*s%s_len_%d source=%.*s\n", (int)seedName.size(), seedName.data(), suffix, (int)llen, (int)sourceSequence.size(), sourceSequence.data());
        F.ext.append(contigSeq, llen);
        F.ext += '\n';
    }
}

static std::string vcf_header(const std::string& sample, const std::string& prefix) /* src/Filler.cpp:349-383 */
{
    time_t now = time(NULL);
    std::string o;
    appendf(o,
            "##fileformat=VCFv4.1\n##filedate=%s##source=MindTheGap fill version %s\n##SAMPLE=file:%s\n##REF=file:%s\n"
            "##INFO=<ID=TYPE,Number=1,Type=String,Description=\"INS\">\n##INFO=<ID=LEN,Number=1,Type=Integer,Description=\"variant size\">\n"
            "##INFO=<=QUAL,Number=.,Type=Integer,Description=\"Quality of the insertion\">\n"
            "##INFO=<=AVK,Number=.,Type=Float,Description=\"Average k-mer coverage along the insertion\">\n"
            "##INFO=<=MDK,Number=.,Type=Float,Description=\"Median k-mer coverage along the insertion\">\n"
            "##INFO=<=NSOL,Number=1,Type=String,Description=\"number of alternative insertion sequences for the breakpoint\">\n"
            "##INFO=<ID=NPOS,Number=1,Type=Integer,Description=\"number of alternative positions for the insertion site (= size of repeat (fuzzy) +1)\">\n"
            "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tG1\n",
            ctime(&now), MTG_VERSION, sample.c_str(), prefix.c_str());
    return o;
}
static void write_vcf_header(FILE* f, const std::string& sample, const std::string& prefix)
{
    const std::string h = vcf_header(sample, prefix);
    fwrite(h.data(), 1, h.size(), f);
}

static void usage()
{
    fprintf(stdout,
            "\nUsage:  MindTheGap fill (-in <reads.fq> | -graph <graph.mtgidx>) -bkpt <breakpoints.fa or -contig <contig.fa> [options]\n"
            "   -in -graph -contig -bkpt -out -overlap -filter -extend | -kmer-size (31) -abundance-min (auto) -abundance-max |\n"
            "   -max-nodes (100) -max-length (10000) -fwd-only | -nb-cores (0) -max-disk -max-memory -verbose\n");
}

struct Summary {
    int nb_breakpoints = 0, nb_filled = 0, nb_multiple = 0, nb_contigs = 0, nb_used_contigs = 0;
};

/* The reference hands groups of records to its Dispatcher threads (src/Filler.cpp:824,844); here the sites go in batches to the devices.
 * Every device (each with its replica of the index) is served by MTG_CLI_IN_FLIGHT host threads (default 3: the library runs up to six
 * batches of an index side by side), each of which takes the next batch, fills it and formats its text; the calling thread writes the
 * batches' text in input order as it becomes complete -- the reference's order with -nb-cores 1. */
struct Replicas {
    std::vector<const mtg_index*> idx; /* idx[0] = the index itself */
    std::vector<mtg_index*> owned;
    ~Replicas() { for (mtg_index* i : owned) mtg_index_free(i); }
    /* The replicas of a node, as a DOUBLING TREE (1 -> 2 -> 4 -> 8): in every round each device that holds the index clones it to one that does
     * not, all pairs of a round at the same time (a host thread per destination; the five buffers of a replica in flight together, peer access
     * enabled both ways: index_replicate).  Round 4 cloned from device 0 to the others one after the other: seven blocking copies of 58 GB over
     * one link each for a node of eight -- several seconds for a job whose fills take milliseconds; three rounds of concurrent copies over
     * distinct xGMI links now.  MTG_TOOL_TIMERS=1 prints the time. */
    int make(const mtg_index* primary, int want)
    {
        idx.push_back(primary);
        const int ndev = mtg_device_count();
        if (tune::is_set(tune::T_NB_GPUS)) { if (want <= 0) want = (int)tune::i(tune::T_NB_GPUS); }
        const int use = want > 0 ? std::min(want, ndev) : ndev;
        std::vector<int> targets;
        for (int d = 0; d < ndev && (int)targets.size() + 1 < use; d++) if (d != primary->device) targets.push_back(d);
        const auto t0 = std::chrono::steady_clock::now();
        size_t next = 0;
        int rounds = 0;
        while (next < targets.size()) {
            const size_t pairs = std::min(idx.size(), targets.size() - next);
            std::vector<mtg_index*> made(pairs, nullptr);
            std::vector<int> rcs(pairs, MTG_OK);
            std::vector<std::string> msgs(pairs);
            std::vector<std::thread> th;
            for (size_t p = 0; p < pairs; p++)
                th.emplace_back([&, p] {
                    rcs[p] = mtg_index_replicate(idx[p], targets[next + p], &made[p]);
                    if (rcs[p]) msgs[p] = mtg_last_error(); /* the message is the worker thread's: brought over below */
                });
            for (std::thread& t : th) t.join();
            int rc = MTG_OK;
            for (size_t p = 0; p < pairs; p++) {
                if (made[p]) owned.push_back(made[p]); /* freed with the others whatever happens next */
                if (rcs[p] && !rc) { rc = rcs[p]; mtgi::set_error("%s", msgs[p].c_str()); }
            }
            if (rc) return rc;
            for (size_t p = 0; p < pairs; p++) idx.push_back(made[p]);
            next += pairs;
            rounds++;
        }
        if (tune::on(tune::T_TOOL_TIMERS) && !targets.empty())
            fprintf(stderr, "[tool] index replicated to %zu more device(s) in %d round(s): %.3f s\n", targets.size(), rounds,
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        return MTG_OK;
    }
};
static size_t cli_batch_size()
{
    const long v = tune::i(tune::T_CLI_BATCH, 0);
    return v > 0 ? (size_t)v : (size_t)100000;
}
static int cli_in_flight()
{
    const int v = (int)tune::i(tune::T_CLI_IN_FLIGHT, 0);
    return v > 0 ? std::min(v, (int)mtg_index::NWS) : 3;
}
/* next(b) hands out batch b (false: the input is exhausted; called by one thread at a time, in order); process(b, idx) runs it on a
 * device; consume(b) on the calling thread, in order, once batch b is complete.  At most `window` batches are between next() and the end of
 * consume() at any time (bounded memory).  The first error stops the hand-out; what was complete before it is still consumed in order. */
static int run_batches(const Replicas& R, const std::function<bool(size_t)>& next, const std::function<int(size_t, const mtg_index*)>& process,
                       const std::function<void(size_t)>& consume)
{
    std::mutex m, read_m;
    std::condition_variable cv;
    size_t handed = 0, consumed = 0; /* batches handed out / consumed */
    bool exhausted = false;
    std::vector<char> done; /* done[b]: batch b has been processed */
    int err = MTG_OK;
    std::string err_text;
    const int per_dev = cli_in_flight();
    const size_t window = (size_t)per_dev * R.idx.size() + 2;
    auto worker = [&](size_t d) {
        for (;;) {
            size_t b;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return exhausted || err || handed - consumed < window; });
                if (exhausted || err) return;
            }
            {   /* the next batch is read by one worker at a time, in order, but not under the scheduler's lock: completions and the writer go on */
                std::lock_guard<std::mutex> rl(read_m);
                {
                    /* the window is checked again by the one worker that holds the reader: several may have passed the wait above for the
                     * same free place, and each would otherwise hand out a batch (the window would be exceeded by the number of workers) */
                    std::unique_lock<std::mutex> lk(m);
                    cv.wait(lk, [&] { return exhausted || err || handed - consumed < window; });
                    if (exhausted || err) return;
                    b = handed;
                }
                const bool more = next(b);
                std::lock_guard<std::mutex> lk(m);
                if (!more) { exhausted = true; cv.notify_all(); return; }
                handed++;
                done.push_back(0);
            }
            const int rc = process(b, R.idx[d]);
            std::lock_guard<std::mutex> lk(m);
            if (rc && !err) { err = rc; err_text = mtg_last_error(); } /* the message is thread-local: keep the first one */
            done[b] = rc ? 2 : 1;
            cv.notify_all();
        }
    };
    std::vector<std::thread> workers;
    for (size_t d = 0; d < R.idx.size(); d++)
        for (int t = 0; t < per_dev; t++) workers.emplace_back(worker, d);
    for (;;) {
        size_t b;
        {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return (consumed < handed && done[consumed]) || ((exhausted || err) && consumed == handed); });
            if (consumed == handed) break;
            b = consumed;
            if (done[b] == 2 || err) { consumed++; cv.notify_all(); continue; } /* a failed batch, or one behind a failure: nothing is written after the error */
        }
        consume(b);
        std::lock_guard<std::mutex> lk(m);
        consumed++;
        cv.notify_all();
    }
    for (auto& t : workers) t.join();
    if (err) set_error("%s", err_text.c_str());
    return err;
}

/* ---- the breakpoint file as a stream of batches (src/Filler.cpp:285,844: records 2i / 2i+1 = left / right k-mer of site i).  A batch owns
 * the text of its records; headers and sequences are NUL-terminated in place, so that the gaps handed to the library point straight into it. */
struct BkptRec { const char* hdr; uint32_t hdr_len; const char* seq; uint32_t seq_len; };
/* the text of a batch: a vector whose resize() leaves the new bytes as they are (they are read into at once; zeroing 17 MB per batch was a
 * third of the reader's time) */
template <class T> struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    template <class U, class... A> void construct(U* p, A&&... a)
    {
        if constexpr (sizeof...(A) == 0) ::new ((void*)p) U;
        else ::new ((void*)p) U(std::forward<A>(a)...);
    }
};
typedef std::vector<char, NoInitAlloc<char>> TextBuf;
static Recycler<TextBuf>& text_pool() { static Recycler<TextBuf> p(16); return p; }
struct BkptReader {
    gzFile f = nullptr;
    int fd = -1; /* a file that is not gzip-compressed is read with read(2): zlib's pass-through copies and checksums for nothing */
    TextBuf carry; /* the beginning of the next batch: a partial record, or records beyond the batch size */
    bool eof = false, bad = false;
    size_t per_batch; /* records */
    /* a plain file of some size is mapped: the one thread that hands out batches only finds where a batch ends (memchr over the mapping), the
     * worker that processes the batch copies its bytes (the records are edited in place) -- the copy was two thirds of the serial part */
    const char* map = nullptr;
    size_t map_size = 0, map_pos = 0;
    explicit BkptReader(size_t sites_per_batch) : per_batch(2 * sites_per_batch) {}
    ~BkptReader() { if (map) ::munmap((void*)map, map_size); if (f) gzclose(f); if (fd >= 0) ::close(fd); }
    /* mapped file: the byte range of the next per_batch records; false: no record is left */
    bool next_span(size_t& begin, size_t& end)
    {
        size_t p = map_pos, nrec = 0;
        begin = p;
        while (p < map_size) {
            const char* hit = (const char*)memchr(map + p, '>', map_size - p);
            if (!hit) { p = map_size; break; }
            p = (size_t)(hit - map);
            if (p == 0 || map[p - 1] == '\n') {
                if (nrec == per_batch) break;
                nrec++;
            }
            p++;
        }
        end = p;
        map_pos = p;
        return nrec > 0;
    }
    void copy_span(size_t begin, size_t end, TextBuf& text) const
    {
        text = text_pool().get();
        text.clear();
        text.resize(end - begin + 2);
        memcpy(text.data(), map + begin, end - begin);
        text[end - begin] = '\n'; /* the last record may lack its line end */
        text[end - begin + 1] = '\0';
    }
    bool open(const std::string& path)
    {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        unsigned char magic[2] = {0, 0};
        const ssize_t got = ::pread(fd, magic, 2, 0);
        if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
            ::close(fd); fd = -1;
            f = gzopen(path.c_str(), "rb");
            if (f) gzbuffer(f, 1 << 20);
            return f != nullptr;
        }
        struct stat sb;
        if (fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0 && !tune::on(tune::T_CLI_NO_MMAP)) {
            void* m = ::mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) { map = (const char*)m; map_size = (size_t)sb.st_size; (void)::madvise(m, map_size, MADV_SEQUENTIAL); }
        }
        return true;
    }
    /* the text of the next per_batch records (fewer at the end of the file) into `text`; false: no record is left */
    size_t last_bytes = 0;
    bool next(TextBuf& text)
    {
        text = text_pool().get();
        text.clear();
        text.insert(text.end(), carry.begin(), carry.end());
        carry.clear();
        text.reserve(last_bytes + last_bytes / 16 + ((size_t)9 << 20)); /* one allocation: the batches of a file are of a size */
        size_t scanned = 0, nrec = 0; /* record starts seen in text[0, scanned) */
        size_t cut = std::string::npos;
        for (;;) {
            /* record starts: a '>' (FASTA) at the beginning of a line */
            while (scanned < text.size()) {
                const char* hit = (const char*)memchr(text.data() + scanned, '>', text.size() - scanned);
                if (!hit) { scanned = text.size(); break; }
                scanned = (size_t)(hit - text.data());
                if (scanned == 0 || text[scanned - 1] == '\n') {
                    if (nrec == per_batch) { cut = scanned; break; }
                    nrec++;
                }
                scanned++;
            }
            if (cut != std::string::npos || eof) break;
            const size_t old = text.size(), want = (size_t)8 << 20;
            text.resize(old + want);
            if (fd >= 0) {
                size_t have = 0;
                while (have < want) {
                    const ssize_t got = ::read(fd, text.data() + old + have, want - have);
                    if (got < 0) { if (errno == EINTR) continue; bad = true; eof = true; break; }
                    if (got == 0) { eof = true; break; }
                    have += (size_t)got;
                }
                text.resize(old + have);
                if (bad) break;
                continue;
            }
            const int got = gzread(f, text.data() + old, (unsigned)want);
            if (got < 0) { bad = true; eof = true; text.resize(old); break; }
            text.resize(old + (size_t)got);
            if ((size_t)got < want) {
                int zerr = Z_OK;
                (void)gzerror(f, &zerr);
                if (!gzeof(f) || !(zerr == Z_OK || zerr == Z_STREAM_END)) bad = true; /* a truncated or corrupt .gz does not pass for a short file */
                eof = true;
            }
        }
        if (cut != std::string::npos) { carry.assign(text.begin() + (ptrdiff_t)cut, text.end()); text.resize(cut); }
        last_bytes = text.size();
        text.push_back('\n'); /* the last record may lack its line end */
        text.push_back('\0');
        return nrec > 0 && !bad;
    }
};
/* the records of a batch's text: header lines cut at their end, sequence lines joined in place (a sequence may span several lines) */
static void parse_records(TextBuf& text, std::vector<BkptRec>& recs)
{
    recs.clear();
    char* p = text.data();
    char* const end = p + text.size() - 1; /* the final NUL */
    while (p < end) {
        if (*p != '>') { char* nl = (char*)memchr(p, '\n', (size_t)(end - p)); p = nl ? nl + 1 : end; continue; } /* anything before the first record */
        char* nl = (char*)memchr(p, '\n', (size_t)(end - p));
        if (!nl) nl = end;
        BkptRec r;
        r.hdr = p + 1;
        char* he = nl;
        if (he > p + 1 && he[-1] == '\r') he--;
        *he = 0;
        r.hdr_len = (uint32_t)(he - (p + 1));
        p = nl < end ? nl + 1 : end;
        char* out = p; /* the sequence is compacted to here */
        r.seq = out;
        while (p < end && *p != '>') {
            char* l2 = (char*)memchr(p, '\n', (size_t)(end - p));
            if (!l2) l2 = end;
            char* le = l2;
            if (le > p && le[-1] == '\r') le--;
            if (out != p) memmove(out, p, (size_t)(le - p));
            out += le - p;
            p = l2 < end ? l2 + 1 : end;
        }
        r.seq_len = (uint32_t)(out - r.seq);
        *out = 0; /* out < p: at least the line end of the header or of a sequence line lies in between (or the final NUL) */
        recs.push_back(r);
    }
}
static void revcomp_into(const char* s, size_t n, std::string& o) /* revcomp_sequence, src/Utils.cpp:44-77: other characters are dropped */
{
    for (size_t i = n; i-- > 0;) {
        switch (s[i]) {
            case 'a': o += 't'; break; case 't': o += 'a'; break; case 'c': o += 'g'; break; case 'g': o += 'c'; break;
            case 'A': o += 'T'; break; case 'T': o += 'A'; break; case 'C': o += 'G'; break; case 'G': o += 'C'; break;
        }
    }
}

/* what one breakpoint site adds to the output files (breakpointFunctor, src/Filler.cpp:665-683, and the writers behind it): fwd = the
 * forward attempt's record, rev = the reverse attempt's (nullptr: there was none) */
struct SiteRef { std::string_view name, name_r, source, target; };
static size_t format_bkpt_site(OutText& T, const SiteRef& s, const mtg_gap_result* fwd, const mtg_gap_result* rev, bool filter, bool extend, std::string& info)
{
    const mtg_gap_result* res = fwd;
    info = info_string(*fwd);
    std::string_view name = s.name;
    if (rev) {
        res = rev;
        info += info_string(*rev);
        name = s.name_r; /* src/Filler.cpp:674 */
    }
    const Sols sols = sols_of(*res);
    write_filled(T, true, DictView{}, sols, name, info);
    write_vcf(T, filter, sols, name, s.source);
    if (sols.empty() && extend) {
        write_extension(T, fwd->extension, name, "", s.source);
        std::string rsrc;
        revcomp_into(s.target.data(), s.target.size(), rsrc);
        write_extension(T, rev ? rev->extension : "", name, "_reverse", rsrc);
    }
    return sols.size();
}

static int run_bkpt(const Replicas& R, const mtg_params& P, const Options& O, Files& F, Summary& S)
{
    BkptReader reader(cli_batch_size());
    if (!reader.open(O.bkpt)) { set_error("cannot read %s", O.bkpt.c_str()); return MTG_ERR_IO; }
    struct Batch {
        TextBuf text;
        size_t span_begin = 0, span_end = 0; /* mapped input: the batch's bytes are copied by the worker that processes it */
        std::vector<BkptRec> recs;
        size_t n = 0; /* sites */
        /* the gaps as the library takes them from text (mtg_fill_text): offsets into bt.text (forward attempts) or rev_text (reverse attempts) */
        struct TextArrays {
            std::vector<uint64_t> so, po, no;
            std::vector<uint32_t> sl, pl, nl, first;
            std::vector<uint8_t> flags;
            void resize(size_t n) { so.resize(n); po.resize(n); no.resize(n); sl.resize(n); pl.resize(n); nl.resize(n); flags.resize(n); first.resize(n + 1); for (size_t i = 0; i <= n; i++) first[i] = (uint32_t)i; }
            mtg_text_gaps view(const char* text, size_t bytes) const
            {
                mtg_text_gaps g{};
                g.text = text; g.text_bytes = bytes; g.n = so.size();
                g.source_off = so.data(); g.source_len = sl.data(); g.pattern_off = po.data(); g.pattern_len = pl.data();
                g.dict_first = first.data(); g.dict_seq_off = po.data(); g.dict_seq_len = pl.data(); /* the one-entry dictionary: the pattern itself */
                g.dict_name_off = no.data(); g.dict_name_len = nl.data(); g.dict_is_rc = nullptr; g.gap_flags = flags.data();
                return g;
            }
        } fwd, rev;
        std::vector<uint32_t> name_len, name_r_len;          /* the names cut at their first space (src/Filler.cpp:631-636) */
        std::vector<uint64_t> name_off;                      /* where the left record's name starts in the batch's text */
        mtg_formatted* ftext = nullptr;                      /* the text of the simple sites, formatted on the device (page-locked arenas of the library) */
        std::vector<uint32_t> host_sites;                    /* device formatting: the sites the host formats, ascending; out[q] = the text of host_sites[q] */
        std::vector<long> rev_idx;
        std::string rev_text;
        mtg_results *rf = nullptr, *rr = nullptr;
        std::vector<OutText> out; /* one piece per FORMAT_CHUNK sites */
        size_t filled = 0, multiple = 0;
        ~Batch()
        {
            if (rf) mtg_results_free(rf);
            if (rr) mtg_results_free(rr);
            if (ftext) ftext_pool().put(std::move(ftext));
            for (OutText& T : out) piece_pool().put(std::move(T));
            if (text.capacity()) text_pool().put(std::move(text));
        }
    };
    enum { FORMAT_CHUNK = 2048 };
    /* MTG_TOOL_TIMERS=1: where the time of a run goes (summed over the worker threads; read and write are one thread each) */
    const bool timers = tune::on(tune::T_TOOL_TIMERS);
    std::atomic<long long> t_read{0}, t_parse{0}, t_fill{0}, t_rev{0}, t_format{0}, t_write{0};
    const auto usec = [] { return (long long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const long long w_start = usec();
    std::atomic<long long> w_first_read{0}, w_last_done{0};
    PositionedWriter writer(F);
    std::mutex bm;
    std::vector<std::unique_ptr<Batch>> batches;
    const auto next = [&](size_t b) -> bool {
        std::unique_ptr<Batch> bt(new Batch());
        const long long tr = usec();
        const bool more = reader.map ? reader.next_span(bt->span_begin, bt->span_end) : reader.next(bt->text);
        t_read += usec() - tr;
        if (!w_first_read) w_first_read = usec();
        if (!more) return false;
        std::lock_guard<std::mutex> lk(bm);
        if (batches.size() <= b) batches.resize(b + 1);
        batches[b] = std::move(bt);
        return true;
    };
    const auto process = [&](size_t b, const mtg_index* idx) -> int {
        Batch* btp;
        { std::lock_guard<std::mutex> lk(bm); btp = batches[b].get(); }
        Batch& bt = *btp;
        long long tp = usec();
        if (reader.map) reader.copy_span(bt.span_begin, bt.span_end, bt.text);
        parse_records(bt.text, bt.recs);
        const size_t n = bt.recs.size() / 2; /* records 2i / 2i+1 = left / right k-mer, src/Filler.cpp:625-629 */
        bt.n = n;
        bt.fwd.resize(n); bt.name_len.resize(n); bt.name_r_len.resize(n); bt.name_off.resize(n);
        const char* const t0 = bt.text.data();
        for (size_t j = 0; j < n; j++) {
            const BkptRec &l = bt.recs[2 * j], &r = bt.recs[2 * j + 1];
            const void* sp = memchr(l.hdr, ' ', l.hdr_len);
            bt.name_len[j] = sp ? (uint32_t)((const char*)sp - l.hdr) : l.hdr_len;
            bt.name_off[j] = (uint64_t)(l.hdr - t0);
            sp = memchr(r.hdr, ' ', r.hdr_len);
            bt.name_r_len[j] = sp ? (uint32_t)((const char*)sp - r.hdr) : r.hdr_len;
            bt.fwd.so[j] = (uint64_t)(l.seq - t0); bt.fwd.sl[j] = l.seq_len;
            bt.fwd.po[j] = (uint64_t)(r.seq - t0); bt.fwd.pl[j] = r.seq_len; /* target sequence = key of the one-entry dictionary */
            bt.fwd.no[j] = (uint64_t)(r.hdr - t0); bt.fwd.nl[j] = r.hdr_len;  /* its value: only multi-contig gaps look at it, and only to group by it */
            const bool rep = std::string_view(l.hdr, l.hdr_len).find("REPEATED") != std::string_view::npos || std::string_view(r.hdr, r.hdr_len).find("REPEATED") != std::string_view::npos;
            bt.fwd.flags[j] = rep ? 1 : 0;
        }
        /* the strings stay text: the library sends the block up and encodes on the device (mtg_fill_text) */
        const mtg_text_gaps gf = bt.fwd.view(t0, bt.text.size());
        t_parse += usec() - tp; tp = usec();
        /* the text of the sites with one solution is written on the device (mtg_fill_text_formatted); MTG_HOST_FORMAT=1: A/B hook, every site by the host's writers */
        const bool dev_format = !tune::on(tune::T_HOST_FORMAT);
        int rc;
        /* with the text formatted on the device a call needs the worker pool for one thing only, the copy of its 13 MB block of text: two threads do
         * that (measured: 10.9-12.4 M sites/s with two pool threads, 6.7-10.8 with sixteen that spin next to the writers on a 16-CPU quota) */
        mtg_params Pf = P;
        if (dev_format && (Pf.nb_host_threads <= 0 || Pf.nb_host_threads > 2)) Pf.nb_host_threads = 2;
        if (dev_format) { bt.ftext = ftext_pool().get(); rc = mtg_fill_text_formatted(idx, &Pf, &gf, bt.name_off.data(), bt.name_len.data(), &bt.rf, &bt.ftext); }
        else rc = mtg_fill_text(idx, &P, &gf, &bt.rf);
        if (rc) return rc;
        t_fill += usec() - tp; tp = usec();
        /* reverse attempt for the sites without solution, src/Filler.cpp:669-680 */
        bt.rev_idx.assign(n, -1);
        if (!O.fwd_only) {
            size_t nr = 0;
            for (size_t j = 0; j < n; j++) if (mtg_results_get(bt.rf, j)->n_filled == 0) bt.rev_idx[j] = (long)nr++;
            bt.rev.resize(nr);
            size_t q = 0;
            for (size_t j = 0; j < n && nr; j++)
                if (bt.rev_idx[j] >= 0) {
                    const BkptRec &l = bt.recs[2 * j], &r = bt.recs[2 * j + 1];
                    bt.rev.so[q] = bt.rev_text.size();
                    revcomp_into(r.seq, r.seq_len, bt.rev_text); /* the reverse attempt's source */
                    bt.rev.sl[q] = (uint32_t)(bt.rev_text.size() - bt.rev.so[q]);
                    bt.rev.po[q] = bt.rev_text.size();
                    revcomp_into(l.seq, l.seq_len, bt.rev_text); /* and its target */
                    bt.rev.pl[q] = (uint32_t)(bt.rev_text.size() - bt.rev.po[q]);
                    bt.rev.no[q] = bt.rev_text.size();
                    bt.rev_text.append(l.hdr, l.hdr_len);
                    bt.rev.nl[q] = l.hdr_len;
                    bt.rev.flags[q] = (uint8_t)(bt.fwd.flags[j] | 2);
                    q++;
                }
            if (nr) {
                const mtg_text_gaps gr = bt.rev.view(bt.rev_text.data(), bt.rev_text.size());
                rc = mtg_fill_text(idx, dev_format ? &Pf : &P, &gr, &bt.rr);
                if (rc) return rc;
            }
        }
        t_rev += usec() - tp; tp = usec();
        if (bt.ftext) {
            /* the device has written the simple sites; the host's writers take the others (reverse attempts, several solutions, records the host
             * wrote), one small piece per site, placed between the device's bytes when the batch is consumed */
            mtg_formatted_view fv;
            if ((rc = mtg_formatted_get(bt.ftext, &fv))) return rc;
            bt.host_sites.assign(fv.complex_sites, fv.complex_sites + fv.n_complex);
            bt.out.resize(bt.host_sites.size());
            std::vector<size_t> pf(bt.host_sites.size(), 0), pm(bt.host_sites.size(), 0);
            parallel_for((bt.host_sites.size() + 63) / 64, P.nb_host_threads, [&](size_t pc) {
                std::string info;
                for (size_t q = pc * 64; q < std::min(bt.host_sites.size(), (pc + 1) * 64); q++) {
                    const size_t j = bt.host_sites[q];
                    OutText& T = bt.out[q];
                    T = piece_pool().get();
                    const BkptRec &l = bt.recs[2 * j], &r = bt.recs[2 * j + 1];
                    const SiteRef site{std::string_view(l.hdr, bt.name_len[j]), std::string_view(r.hdr, bt.name_r_len[j]), std::string_view(l.seq, l.seq_len), std::string_view(r.seq, r.seq_len)};
                    const size_t nsol = format_bkpt_site(T, site, mtg_results_get(bt.rf, j), bt.rev_idx[j] >= 0 ? mtg_results_get(bt.rr, (size_t)bt.rev_idx[j]) : nullptr, O.filter, O.extend, info);
                    pf[q] = nsol > 0; pm[q] = nsol > 1;
                }
            }, 1);
            bt.filled = (size_t)fv.n_simple;
            for (size_t q = 0; q < bt.host_sites.size(); q++) { bt.filled += pf[q]; bt.multiple += pm[q]; }
        } else {
        /* the batch's text, formatted in pieces by the library's worker pool (the writers of src/Filler.cpp:1029-1214) */
        const size_t npieces = (n + FORMAT_CHUNK - 1) / FORMAT_CHUNK;
        bt.out.resize(npieces);
        std::vector<size_t> pf(npieces, 0), pm(npieces, 0);
        parallel_for(npieces, P.nb_host_threads, [&](size_t pc) {
            OutText& T = bt.out[pc];
            T = piece_pool().get();
            const size_t j1 = std::min(n, (pc + 1) * (size_t)FORMAT_CHUNK);
            T.insert.reserve((j1 - pc * FORMAT_CHUNK) * 512);
            T.vcf.reserve((j1 - pc * FORMAT_CHUNK) * 600);
            std::string info;
            for (size_t j = pc * FORMAT_CHUNK; j < j1; j++) {
                const BkptRec &l = bt.recs[2 * j], &r = bt.recs[2 * j + 1];
                const SiteRef site{std::string_view(l.hdr, bt.name_len[j]), std::string_view(r.hdr, bt.name_r_len[j]), std::string_view(l.seq, l.seq_len), std::string_view(r.seq, r.seq_len)};
                const size_t nsol = format_bkpt_site(T, site, mtg_results_get(bt.rf, j), bt.rev_idx[j] >= 0 ? mtg_results_get(bt.rr, (size_t)bt.rev_idx[j]) : nullptr, O.filter, O.extend, info);
                pf[pc] += nsol > 0;
                pm[pc] += nsol > 1;
            }
        }, 1);
        for (size_t pc = 0; pc < npieces; pc++) { bt.filled += pf[pc]; bt.multiple += pm[pc]; }
        }
        /* records and sequences are in the text now */
        mtg_results_free(bt.rf); bt.rf = nullptr;
        if (bt.rr) { mtg_results_free(bt.rr); bt.rr = nullptr; }
        text_pool().put(std::move(bt.text));
        bt.text = TextBuf();
        t_format += usec() - tp;
        w_last_done = usec();
        return MTG_OK;
    };
    const auto consume = [&](size_t b) {
        std::unique_ptr<Batch> bt;
        { std::lock_guard<std::mutex> lk(bm); bt = std::move(batches[b]); }
        const long long tw = usec();
        std::shared_ptr<Batch> keep(bt.release()); /* the text lives until its last piece is in the files */
        if (keep->ftext) {
            /* the device's bytes in pieces of a few MB (so that several writer threads share a batch), the host's sites between them where they belong */
            mtg_formatted_view fv;
            (void)mtg_formatted_get(keep->ftext, &fv);
            uint64_t cur[3] = {0, 0, 0};
            const auto device_bytes_up_to = [&](const uint64_t upto[3]) {
                const size_t piece = (size_t)4 << 20;
                for (int s3 = 0; s3 < 3; s3++)
                    while (cur[s3] < upto[s3]) {
                        PositionedWriter::Span sp[5];
                        const size_t nb = (size_t)std::min<uint64_t>(piece, upto[s3] - cur[s3]);
                        sp[s3].p = fv.text[s3] + cur[s3]; sp[s3].n = nb; /* FASTA, info, VCF are streams 0, 1, 2 of the writer as well */
                        writer.add_spans(keep, sp);
                        cur[s3] += nb;
                    }
            };
            for (size_t q = 0; q < keep->host_sites.size(); q++) {
                const uint64_t upto[3] = {fv.complex_off[0][q], fv.complex_off[1][q], fv.complex_off[2][q]};
                device_bytes_up_to(upto);
                writer.add(keep, keep->out[q]);
            }
            device_bytes_up_to(fv.bytes);
        } else
            for (const OutText& T : keep->out) writer.add(keep, T);
        Batch* const btq = keep.get();
        t_write += usec() - tw;
        S.nb_breakpoints += (int)btq->n;
        S.nb_filled += (int)btq->filled;
        S.nb_multiple += (int)btq->multiple;
    };
    int rc = run_batches(R, next, process, consume);
    {
        const long long tw = usec();
        if (!writer.finish() && !rc) { set_error("cannot write the output files"); rc = MTG_ERR_IO; }
        t_write += usec() - tw;
    }
    if (timers) fprintf(stderr, "[tool] wall ms: first batch read %.1f, last batch processed %.1f, files complete %.1f\n", (w_first_read - w_start) / 1e3, (w_last_done - w_start) / 1e3, (usec() - w_start) / 1e3);
    if (timers)
        fprintf(stderr, "[tool] ms summed over threads: read %.1f | parse + gaps %.1f, forward fill %.1f, reverse attempts %.1f, format %.1f | write %.1f\n", t_read / 1e3, t_parse / 1e3, t_fill / 1e3,
                t_rev / 1e3, t_format / 1e3, t_write / 1e3);
    if (!rc && reader.bad) { set_error("cannot read %s (truncated or corrupt)", O.bkpt.c_str()); return MTG_ERR_IO; }
    return rc;
}

static int run_contig(const Replicas& R, const mtg_params& P, const Options& O, Files& F, Summary& S, int trim)
{
    const int k = R.idx[0]->info.k;
    /* DEBUG_TIMERS: where a contig job's time goes (thread time of the workers for the three per-batch parts) */
    const bool dbg = tune::on(tune::T_DEBUG_TIMERS);
    const auto clock_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    std::atomic<long> us_prep{0}, us_run{0}, us_text{0};
    const double t_begin = clock_ms();
    std::vector<std::pair<std::string, std::string>> recs;
    if (!read_sequences(O.contig, recs)) { set_error("cannot read %s", O.contig.c_str()); return MTG_ERR_IO; }
    bkpt_dict_t all_targets;
    std::vector<std::pair<std::string, std::string>> seeds; /* name, k-mer */
    {
        std::ofstream seedFile(O.out + "_seed_dictionary.fasta");
        for (auto& r : recs) {
            const std::string& contig = r.second;
            const std::string name = short_name(r.first);
            S.nb_contigs++;
            fprintf(F.gfa, "S\t%s\t%s\n", name.c_str(), contig.c_str());
            if (contig.size() > (size_t)(2 * trim + k)) { /* src/Filler.cpp:776 */
                const std::string rc = revcomp_str(contig);
                const std::string seed_f = contig.substr(contig.size() - (trim + k), k), target_f = contig.substr(trim, k);
                const std::string seed_rc = rc.substr(rc.size() - (trim + k), k), target_rc = rc.substr(trim, k);
                all_targets.insert({{target_f, std::make_pair(name, false)}, {target_rc, std::make_pair(name, true)}});
                seedFile << ">" + name + "\n" << seed_f << std::endl;
                seedFile << ">" + name + "_Rc\n" << seed_rc << std::endl;
                seeds.push_back({name, seed_f});
                seeds.push_back({name + "_Rc", seed_rc});
                S.nb_used_contigs++;
            } else {
                fprintf(stderr, "Warning contig not used (too short: <= 2 x overlap + kmerSize = %d nt): %s of size %zu nt\n", 2 * trim + k, name.c_str(), contig.size());
            }
        }
    }
    /* contigFunctor, src/Filler.cpp:492-572; every seed sees the targets of all the other contigs (:522-533).  The targets once, in the iteration order
     * of their dictionary: a seed's own dictionary is this one without its contig's entry, in the order mtg_dict_order.h derives from the hash codes */
    struct Targets {
        std::vector<const char*> seq, pname;
        std::vector<std::string> name;
        std::vector<uint8_t> rc;
        std::vector<size_t> code;
        std::string text; /* every key one after the other: the early-stop pattern of a seed is this without its own entry */
        std::unordered_map<std::string, std::vector<uint32_t>> by_seed_name; /* the entries a seed of that name leaves out */
    } T;
    {
        const size_t n = all_targets.size();
        T.seq.reserve(n); T.name.reserve(n); T.rc.reserve(n); T.code.reserve(n); T.text.reserve(n * (size_t)k);
        uint32_t i = 0;
        for (auto its = all_targets.begin(); its != all_targets.end(); ++its, ++i) {
            T.seq.push_back(its->first.c_str());
            T.name.push_back(its->second.first);
            T.rc.push_back(its->second.second ? 1 : 0);
            T.code.push_back(std::hash<std::string>()(its->first));
            T.text.append(its->first);
            T.by_seed_name[its->second.second ? its->second.first + "_Rc" : its->second.first].push_back(i);
        }
        for (const std::string& nm : T.name) T.pname.push_back(nm.c_str());
    }
    struct Batch {
        size_t s0 = 0, s1 = 0;
        std::vector<GapArgs> gaps;
        BatchRun run;
        OutText out;
        size_t n = 0, filled = 0, multiple = 0;
    };
    const size_t B = std::max<size_t>(1, cli_batch_size() / std::max<size_t>(1, all_targets.size() / 8)), nb = (seeds.size() + B - 1) / B; /* a gap carries the whole dictionary: fewer per batch */
    std::mutex bm;
    std::vector<std::unique_ptr<Batch>> batches(nb);
    const auto next = [&](size_t b) -> bool { return b < nb; };
    const auto process = [&](size_t b, const mtg_index* idx) -> int {
        std::unique_ptr<Batch> bt(new Batch());
        bt->s0 = b * B; bt->s1 = std::min(seeds.size(), (b + 1) * B);
        bt->gaps.resize(bt->s1 - bt->s0);
        const double tp0 = clock_ms();
        /* (the seeds of a batch over the host's worker pool: a seed's dictionary is 2 (N - 1) entries to put in order and to point at) */
        parallel_for(bt->s1 - bt->s0, P.nb_host_threads, [&](size_t sj) {
            static thread_local mtgcli::DictOrder dict_order;
            static thread_local std::vector<uint8_t> skip;
            if (skip.size() != T.seq.size()) skip.assign(T.seq.size(), 0);
            const size_t si = bt->s0 + sj;
            auto& sd = seeds[si];
            GapArgs& g = bt->gaps[si - bt->s0];
            const auto own = T.by_seed_name.find(sd.first);
            static const std::vector<uint32_t> none;
            const std::vector<uint32_t>& out = own == T.by_seed_name.end() ? none : own->second; /* ascending */
            /* targetSequence: the keys (k characters each) of every other entry, in the order of the dictionary of all (:526-531) */
            {
                size_t from = 0;
                g.target.reserve(T.text.size());
                for (uint32_t o : out) { g.target.append(T.text, from, (size_t)o * (size_t)k - from); from = ((size_t)o + 1) * (size_t)k; }
                g.target.append(T.text, from, std::string::npos);
            }
            for (uint32_t o : out) skip[o] = 1;
            dict_order.order(T.code.data(), (uint32_t)T.seq.size(), out.empty() ? nullptr : skip.data(), g.tidx);
            for (uint32_t o : out) skip[o] = 0;
            g.shared_dict = true;
            g.pseq.resize(g.tidx.size()); g.pname.resize(g.tidx.size()); g.trc.resize(g.tidx.size());
            for (size_t t = 0; t < g.tidx.size(); t++) { const uint32_t e = g.tidx[t]; g.pseq[t] = T.seq[e]; g.pname[t] = T.pname[e]; g.trc[t] = T.rc[e]; }
            g.source = sd.second;
        }, 1);
        const double tp1 = clock_ms();
        if (int rc = bt->run.run(idx, P, bt->gaps)) return rc;
        const double tp2 = clock_ms();
        us_prep += (long)((tp1 - tp0) * 1e3); us_run += (long)((tp2 - tp1) * 1e3);
        for (size_t i = bt->s0; i < bt->s1; i++) {
            const size_t j = i - bt->s0;
            const std::string& seedName = seeds[i].first;
            const bool isRc = seedName.length() >= 3 && seedName.compare(seedName.length() - 3, 3, "_Rc") == 0;
            std::vector<mtg_filled> kept;
            for (auto& s : sols_of(bt->run[j])) { /* drop loops: target == seed reversed, :540-557 */
                const std::string& tn = T.name[bt->gaps[j].tidx[s.target_index]];
                const std::string revTargetName = bt->gaps[j].trc[s.target_index] ? tn : tn + "_Rc";
                if (revTargetName != seedName) kept.push_back(s);
            }
            Sols ks;
            ks.p = kept.data(); ks.n = kept.size();
            const DictView dv{T.name.data(), bt->gaps[j].trc.data(), bt->gaps[j].tidx.data()};
            write_filled(bt->out, false, dv, ks, seedName, info_string(bt->run[j]));
            write_gfa(bt->out, trim, dv, ks, seedName, isRc);
            if (kept.empty() && O.extend) write_extension(bt->out, bt->run[j].extension, seedName, "", seeds[i].second);
            bt->n++;
            bt->filled += kept.size() > 0;
            bt->multiple += kept.size() > 1;
        }
        us_text += (long)((clock_ms() - tp2) * 1e3);
        std::lock_guard<std::mutex> lk(bm);
        batches[b] = std::move(bt);
        return MTG_OK;
    };
    const double t_setup = clock_ms();
    const auto consume = [&](size_t b) {
        std::unique_ptr<Batch> bt;
        { std::lock_guard<std::mutex> lk(bm); bt = std::move(batches[b]); }
        bt->out.write(F);
        S.nb_breakpoints += (int)bt->n;
        S.nb_filled += (int)bt->filled;
        S.nb_multiple += (int)bt->multiple;
    };
    const int rc_all = run_batches(R, next, process, consume);
    if (dbg) fprintf(stderr, "  [contig] %zu seeds x %zu targets in %zu batches: contigs read + dictionary %.1f ms, batches %.1f ms wall (workers' time: dictionaries of the seeds %.1f, mtg_fill_batch %.1f, text %.1f ms)\n",
                     seeds.size(), all_targets.size(), nb, t_setup - t_begin, clock_ms() - t_setup, us_prep / 1e3, us_run / 1e3, us_text / 1e3);
    return rc_all;
}

static int run_tool(Options& O, mtg_index* idx, bool resident);
/* resident != nullptr: the graph is this index, already on a device (mtg_fill_main_on_index): -in / -graph are not expected */
static int fill_main_impl(int argc, const char* const* argv, mtg_index* resident)
{
    Options O;
    for (int i = 0; i < argc; i++) {
        const std::string a = argv[i];
        auto val = [&](std::string& dst) -> bool { if (i + 1 >= argc) return false; dst = argv[++i]; return true; };
        std::string v;
        bool ok = true;
        if (a == "-in") ok = val(O.in);
        else if (a == "-graph") ok = val(O.graph);
        else if (a == "-bkpt") ok = val(O.bkpt);
        else if (a == "-contig") ok = val(O.contig);
        else if (a == "-out") { ok = val(O.out); O.has_out = true; }
        else if (a == "-kmer-size") { ok = val(v); O.k = atoi(v.c_str()); }
        else if (a == "-abundance-min") { ok = val(v); O.abundance_min = v == "auto" ? -1 : atoi(v.c_str()); }
        else if (a == "-abundance-max") { ok = val(v); O.abundance_max = atoi(v.c_str()); }
        else if (a == "-max-nodes") { ok = val(v); O.max_nodes = atoi(v.c_str()); }
        else if (a == "-max-length") { ok = val(v); O.max_depth = atoi(v.c_str()); }
        else if (a == "-overlap") { ok = val(v); O.overlap = atoi(v.c_str()); }
        else if (a == "-nb-cores") { ok = val(v); O.nb_cores = atoi(v.c_str()); }
        else if (a == "-nb-gpus") { ok = val(v); O.nb_gpus = atoi(v.c_str()); } /* not an option of the reference: devices to use (0 = every visible one) */
        else if (a == "-max-memory" || a == "-max-disk" || a == "-verbose") ok = val(v);
        else if (a == "-fwd-only") O.fwd_only = true;
        else if (a == "-filter") O.filter = true;
        else if (a == "-extend") O.extend = true;
        else if (a == "-help" || a == "-h") { usage(); return 1; }
        else { fprintf(stderr, "EXCEPTION: unknown option '%s'\n", a.c_str()); usage(); return 1; }
        if (!ok) { fprintf(stderr, "EXCEPTION: missing value for option '%s'\n", a.c_str()); return 1; }
    }
    /* src/Filler.cpp:140-150 */
    if (!resident && O.graph.empty() == O.in.empty()) { fprintf(stderr, "EXCEPTION: options -graph and -in are incompatible, but at least one of these is mandatory\n"); return 1; }
    if (resident && !(O.graph.empty() && O.in.empty())) { fprintf(stderr, "EXCEPTION: the graph is already resident: options -graph and -in are not expected\n"); return 1; }
    if (O.bkpt.empty() == O.contig.empty()) { fprintf(stderr, "EXCEPTION: option -bkpt and -contig are incompatible, but at least one of these is mandatory\n"); return 1; }
    if (!O.has_out) { /* src/Filler.cpp:154-165 */
        time_t now = time(0);
        struct tm tstruct = *localtime(&now);
        char buf[80];
        strftime(buf, sizeof(buf), "%Y-%m-%d.%I:%M", &tstruct);
        O.out = std::string("MindTheGap_Expe-") + buf;
    }
    mtg_index* idx = resident;
    int rc = MTG_OK;
    if (resident) {
        O.graph = "(resident index)";
    } else if (!O.in.empty()) {
        rc = index_from_reads(O.in.c_str(), O.k, O.abundance_min, O.abundance_max, &idx);
        if (!rc) (void)index_save(idx, (O.out + ".mtgidx").c_str()); /* the reference leaves <out>.h5 behind */
    } else {
        fprintf(stderr, "Loading the graph...");
        rc = index_load(O.graph.c_str(), &idx);
        if (!rc) fprintf(stderr, "done\n");
    }
    if (rc) { fprintf(stderr, "EXCEPTION: %s\n", mtg_last_error()); return 1; }
    rc = run_tool(O, idx, resident != nullptr);
    if (!resident) mtg_index_free(idx);
    return rc;
}
int fill_main(int argc, const char* const* argv) { return fill_main_impl(argc, argv, nullptr); }

static int run_tool(Options& O, mtg_index* idx, bool resident)
{
    int rc = MTG_OK;
    const int k = idx->info.k;
    if (idx->info.nb_saturated)
        fprintf(stderr, "Warning : %llu solid k-mers are more abundant than 255 and are stored as 255 (coverage statistics of such regions may differ from gatb's discretised values)\n",
                (unsigned long long)idx->info.nb_saturated);
    const bool bkpt_mode = !O.bkpt.empty();
    Files F;
    const std::string insert_name = O.out + ".insertions.fasta", info_name = O.out + ".info.txt", vcf_name = O.out + ".insertions.vcf", gfa_name = O.out + ".gfa",
                      ext_name = O.out + ".extensions.fasta";
    auto open_w = [&](FILE*& f, const std::string& name) -> bool {
        f = fopen(name.c_str(), "w");
        if (!f) fprintf(stderr, "EXCEPTION: Cannot open file %s for writing\n", name.c_str());
        return f != nullptr;
    };
    if (!open_w(F.insert, insert_name) || !open_w(F.info, info_name)) return 1;
    if (bkpt_mode) { if (!open_w(F.vcf, vcf_name)) return 1; write_vcf_header(F.vcf, O.in.empty() ? O.graph : O.in, O.out); }
    else if (!open_w(F.gfa, gfa_name)) return 1;
    if (O.extend && !open_w(F.ext, ext_name)) return 1;
    int trim = O.overlap; /* src/Filler.cpp:299-307 */
    if (trim == 0) trim = k;
    if (trim < k) { trim = k; fprintf(stderr, "Warning :  the contig overlap parameter should be greater or equal to kmer size, setting it to %d\n", k); }
    mtg_params P;
    mtg_default_params(&P);
    P.max_nodes = O.max_nodes;
    P.max_depth = O.max_depth;
    P.nb_host_threads = O.nb_cores;
    Summary S;
    const time_t t_start = time(0);
    {
        Replicas R;
        rc = R.make(idx, O.nb_gpus);
        if (!rc) rc = bkpt_mode ? run_bkpt(R, P, O, F, S) : run_contig(R, P, O, F, S, trim);
    }
    const double seconds = difftime(time(0), t_start);
    if (rc) { fprintf(stderr, "EXCEPTION: %s\n", mtg_last_error()); return 1; }
    if (tune::on(tune::T_TOOL_QUIET)) return 0; /* measurements: no summary (a caller that owns stdout, bench.py) */
    /* resumeParameters / resumeResults, src/Filler.cpp:385-481 */
    printf("MindTheGap fill\n    version                                  : %s\n    backend                                  : mindthegap_amd (HIP, gfx950)\n", MTG_VERSION);
    printf("Parameters\n    Input data\n");
    if (!O.in.empty()) printf("        Reads                                    : %s\n", O.in.c_str());
    else printf("        Graph                                    : %s\n", O.graph.c_str());
    printf("        %-40s : %s\n", bkpt_mode ? "Breakpoints" : "Contigs", bkpt_mode ? O.bkpt.c_str() : O.contig.c_str());
    printf("    Graph\n        kmer-size                                : %i\n", k);
    if (idx->info.abundance_auto >= 0) printf("        abundance_min (auto inferred)            : %d\n", idx->info.abundance_auto);
    printf("        abundance_min (used)                     : %d\n        nb_solid_kmers                           : %llu\n        nb_branching_nodes                       : %llu\n",
           idx->info.abundance_min, (unsigned long long)idx->info.nb_solid_kmers, (unsigned long long)idx->info.nb_branching);
    printf("    Assembly options\n        max_depth                                : %i\n        max_nodes                                : %i\n", O.max_depth, O.max_nodes);
    if (!bkpt_mode) printf("        contig trim size before gap-filling      : %i\n", trim);
    printf("Results\n");
    if (bkpt_mode) printf("    Breakpoints\n        nb_input_breakpoints                     : %i\n        nb_filled_breakpoints                    : %i\n", S.nb_breakpoints, S.nb_filled);
    else printf("    Contigs\n        nb_input_contigs                         : %i\n        nb_used_contigs                          : %i\n        nb_input_seeds                           : %i\n        nb_filled_seeds                          : %i\n",
                S.nb_contigs, S.nb_used_contigs, S.nb_breakpoints, S.nb_filled);
    printf("            as_unique_sequence                       : %i\n            as_multiple_sequence                     : %i\n", S.nb_filled - S.nb_multiple, S.nb_multiple);
    printf("    Time                                     : %.1f s\n    Output files\n        assembled sequence file                  : %s\n", seconds, insert_name.c_str());
    if (bkpt_mode) printf("        insertion variant vcf file               : %s\n", vcf_name.c_str());
    else printf("        assembly graph file                      : %s\n", gfa_name.c_str());
    printf("        assembly statistics file                 : %s\n", info_name.c_str());
    if (O.extend) printf("        extension sequence file                  : %s\n", ext_name.c_str());
    return 0;
}

} // namespace mtgi

extern "C" int mtg_fill_main(int argc, const char* const* argv) { return mtgi::fill_main(argc, argv); }
extern "C" int mtg_fill_main_on_index(mtg_index* idx, int argc, const char* const* argv)
{
    if (!idx) { mtgi::set_error("null argument"); return 1; }
    return mtgi::fill_main_impl(argc, argv, idx);
}

extern "C" int mtg_format_bkpt(const mtg_site* sites, size_t n, const mtg_results* fwd, const mtg_results* rev, const int64_t* rev_index, int filter, int extend, mtg_text* out)
{
    using namespace mtgi;
    if (!out || (n && (!sites || !fwd))) { set_error("null argument"); return MTG_ERR_ARG; }
    memset(out, 0, sizeof *out);
    enum { CHUNK = 2048 };
    const size_t npieces = (n + CHUNK - 1) / CHUNK;
    std::vector<OutText> pieces(npieces);
    std::atomic<int> bad{0};
    parallel_for(npieces, 0, [&](size_t pc) {
        std::string info;
        for (size_t j = pc * CHUNK; j < std::min(n, (pc + 1) * (size_t)CHUNK); j++) {
            const mtg_site& s = sites[j];
            const mtg_gap_result* f = mtg_results_get(fwd, j);
            const mtg_gap_result* r = (rev_index && rev_index[j] >= 0) ? (rev ? mtg_results_get(rev, (size_t)rev_index[j]) : nullptr) : nullptr;
            if (!s.name || !s.name_r || !s.source || !s.target || !f || (rev_index && rev_index[j] >= 0 && !r)) { bad = 1; continue; }
            format_bkpt_site(pieces[pc], SiteRef{s.name, s.name_r, s.source, s.target}, f, r, filter != 0, extend != 0, info);
        }
    }, 1);
    if (bad) { set_error("a site lacks a field or its record"); return MTG_ERR_ARG; }
    auto join = [&](std::string OutText::*m, char*& dst, uint64_t& bytes) -> bool {
        uint64_t t = 0;
        for (const OutText& p : pieces) t += (p.*m).size();
        dst = (char*)malloc(t ? t : 1);
        if (!dst) return false;
        uint64_t o = 0;
        for (const OutText& p : pieces) { memcpy(dst + o, (p.*m).data(), (p.*m).size()); o += (p.*m).size(); }
        bytes = t;
        return true;
    };
    if (!join(&OutText::insert, out->fasta, out->fasta_bytes) || !join(&OutText::info, out->info, out->info_bytes) || !join(&OutText::vcf, out->vcf, out->vcf_bytes) ||
        !join(&OutText::ext, out->ext, out->ext_bytes)) { mtg_text_free(out); set_error("out of memory"); return MTG_ERR_NOMEM; }
    return MTG_OK;
}
extern "C" int mtg_format_vcf_header(const char* sample, const char* prefix, mtg_text* out)
{
    if (!sample || !prefix || !out) { mtgi::set_error("null argument"); return MTG_ERR_ARG; }
    memset(out, 0, sizeof *out);
    const std::string h = mtgi::vcf_header(sample, prefix);
    out->vcf = (char*)malloc(h.size() + 1);
    if (!out->vcf) { mtgi::set_error("out of memory"); return MTG_ERR_NOMEM; }
    memcpy(out->vcf, h.data(), h.size());
    out->vcf_bytes = h.size();
    return MTG_OK;
}
extern "C" void mtg_text_free(mtg_text* t)
{
    if (!t) return;
    free(t->fasta); free(t->info); free(t->vcf); free(t->ext);
    memset(t, 0, sizeof *t);
}
