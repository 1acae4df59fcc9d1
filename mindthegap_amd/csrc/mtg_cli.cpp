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

#include <cstdio>
#include <cstring>
#include <ctime>
#include <fstream>
#include <sstream>
#include <unordered_map>
#include <condition_variable>
#include <thread>

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
    bool repeated = false, reverse = false;
    void set_dict(const bkpt_dict_t& dict)
    {
        for (auto it = dict.begin(); it != dict.end(); ++it) { tseq.push_back(it->first); tname.push_back(it->second.first); trc.push_back(it->second.second ? 1 : 0); }
    }
    mtg_gap abi()
    {
        pseq.clear(); pname.clear();
        for (size_t i = 0; i < tseq.size(); i++) { pseq.push_back(tseq[i].c_str()); pname.push_back(tname[i].c_str()); }
        mtg_gap g;
        g.source = source.c_str();
        g.target = target.c_str();
        g.n_targets = (int)tseq.size();
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

static std::string info_string(const mtg_gap_result& g)
{
    char buf[128];
    std::string s;
    snprintf(buf, sizeof buf, "\t%i\t%i\t%d", g.nb_nodes, g.total_nt, g.nb_terminal);
    s += buf;
    if (g.nb_terminal > 0 && g.has_solution_counts) { snprintf(buf, sizeof buf, "\t%d\t%d", g.nb_total_filled, g.nb_reported); s += buf; }
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
};

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
    std::ostringstream o;
    o << "solution " << s.solution_rank << "/" << s.solution_count;
    return o.str();
}

/* writeFilledBreakpoint, src/Filler.cpp:1029-1093.  The bkpt-mode header passes its arguments in a different order
 * than its format (:1052-1054); the visible x86-64 result is NAME_len_L_qual_Q_avg_cov_A_median_cov_M   SOLU. */
static void write_filled(Files& F, bool bkpt_mode, const GapArgs& g, Sols sols, const std::string& seedName, const std::string& info)
{
    for (auto& s : sols) {
        const int llen = (int)strlen(s.seq);
        const std::string solu = solu_str(s);
        if (bkpt_mode) {
            fprintf(F.insert, ">%s_len_%d_qual_%i_avg_cov_%.2f_median_cov_%.2f   %s\n", seedName.c_str(), llen, s.qual, (double)s.avg_coverage, (double)s.median_coverage, solu.c_str());
        } else {
            std::string targetName(g.tname[s.target_index]);
            if (g.trc[s.target_index]) targetName.append("_Rc");
            int cov = s.median_coverage + 0.5;
            fprintf(F.insert, ">%s;%s;len_%d_qual_%d_median_cov_%d\t%s\n", seedName.c_str(), targetName.c_str(), llen, s.qual, cov, solu.c_str());
        }
        fprintf(F.insert, "%.*s\n", llen, s.seq);
    }
    fprintf(F.info, "%s\t%s\n", seedName.c_str(), info.c_str());
}

/* writeVcf, src/Filler.cpp:1095-1214 */
static void write_vcf(Files& F, bool filter, Sols sols, const std::string& breakpointName, const std::string& sourceSequence)
{
    for (auto& s : sols) {
        const std::string seq(s.seq);
        std::string insertion = seq;
        int repeatSize = 0;
        int i = (int)sourceSequence.size() - 1, j = (int)seq.size() - 1;
        while (i > 0 && j >= 0) { /* longest common suffix with a circular insert index, :1107-1126 */
            if (sourceSequence[i] != seq[j]) break;
            repeatSize++; i--; j--;
            if (j == -1) j = (int)seq.size() - 1;
        }
        insertion = sourceSequence.substr(sourceSequence.size() - (repeatSize + 1), repeatSize + 1) + insertion;
        insertion = insertion.substr(0, insertion.size() - repeatSize);
        const std::string ref = sourceSequence.substr(sourceSequence.size() - (repeatSize + 1), 1);
        std::vector<std::string> tokens;
        { std::istringstream iss(breakpointName); std::string tok; while (getline(iss, tok, '_')) tokens.push_back(tok); }
        std::string bkpt = breakpointName, position = ".", chromosome = ".", GT = "./.", genotype = "";
        if (tokens.size() == 7) {
            bkpt = tokens[0]; position = std::to_string(atoi(tokens[3].c_str()) - repeatSize); chromosome = tokens[1]; genotype = tokens[6];
            GT = genotype == "HOM" ? "1/1" : "0/1";
        }
        if (tokens.size() == 8) {
            bkpt = tokens[0] + tokens[2]; position = std::to_string(atoi(tokens[4].c_str()) - repeatSize); chromosome = tokens[1]; genotype = tokens[7];
            GT = genotype == "HOM" ? "1/1" : "0/1";
        }
        const int size = (int)(insertion.size() - ref.size()), nsol = s.solution_count, npos = repeatSize + 1;
        std::string filt = "PASS";
        if ((genotype == "HET" && nsol > 1) || (genotype == "HOM" && nsol > 1)) {
            if (filter) break;
            filt = "LOW_QUAL";
        }
        fprintf(F.vcf, "%s\t%s\t%s\t%s\t%s\t.\t%s\tTYPE=INS;LEN=%i;QUAL=%i;NSOL=%i;NPOS=%i;AVK=%.2f;MDK=%.2f\tGT\t%s\n", chromosome.c_str(), position.c_str(), bkpt.c_str(),
                ref.c_str(), insertion.c_str(), filt.c_str(), size, s.qual, nsol, npos, (double)s.avg_coverage, (double)s.median_coverage, GT.c_str());
    }
}

/* writeToGFA, src/Filler.cpp:1216-1273 */
static void write_gfa(Files& F, int trim, const GapArgs& g, Sols sols, std::string seedName, bool isRc)
{
    const std::string seedNameNode = seedName;
    std::string seedDirection = "+";
    if (isRc) { seedName = seedName.substr(0, seedName.size() - 3); seedDirection = "-"; }
    for (auto& s : sols) {
        const std::string tname(g.tname[s.target_index]);
        const bool trc = g.trc[s.target_index] != 0;
        const std::string targetNameNode = trc ? tname + "_Rc" : tname;
        int cov = s.median_coverage + 0.5;
        const std::string nodeName = seedNameNode + ";" + targetNameNode + ";len_" + std::to_string((int)strlen(s.seq)) + "_qual_" + std::to_string(s.qual) +
                                     "_median_cov_" + std::to_string(cov) + " " + solu_str(s);
        fprintf(F.gfa, "S\t%s\t%s\n", nodeName.c_str(), s.seq);
        fprintf(F.gfa, "L\t%s\t%s\t%s\t+\t%iM\n", seedName.c_str(), seedDirection.c_str(), nodeName.c_str(), trim);
        fprintf(F.gfa, "L\t%s\t+\t%s\t%s\t%iM\n", nodeName.c_str(), tname.c_str(), trc ? "-" : "+", trim);
    }
}

static void write_extension(Files& F, const std::string& contigSeq, const std::string& seedName, const std::string& sourceSequence) /* :1275-1291 */
{
    const int llen = (int)contigSeq.length();
    if (llen > 0) {
        fprintf(F.ext, ">%s_len_%d source=%s\n", seedName.c_str(), llen, sourceSequence.c_str());
        fprintf(F.ext, "%.*s\n", llen, contigSeq.c_str());
    }
}

static void write_vcf_header(FILE* f, const std::string& sample, const std::string& prefix) /* src/Filler.cpp:349-383 */
{
    time_t now = time(NULL);
    fprintf(f,
            "##fileformat=VCFv4.1\n##filedate=%s##source=MindTheGap fill version %s\n##SAMPLE=file:%s\n##REF=file:%s\n"
            "##INFO=<ID=TYPE,Number=1,Type=String,Description=\"INS\">\n##INFO=<ID=LEN,Number=1,Type=Integer,Description=\"variant size\">\n"
            "##INFO=<=QUAL,Number=.,Type=Integer,Description=\"Quality of the insertion\">\n"
            "##INFO=<=AVK,Number=.,Type=Float,Description=\"Average k-mer coverage along the insertion\">\n"
            "##INFO=<=MDK,Number=.,Type=Float,Description=\"Median k-mer coverage along the insertion\">\n"
            "##INFO=<=NSOL,Number=1,Type=String,Description=\"number of alternative insertion sequences for the breakpoint\">\n"
            "##INFO=<ID=NPOS,Number=1,Type=Integer,Description=\"number of alternative positions for the insertion site (= size of repeat (fuzzy) +1)\">\n"
            "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tG1\n",
            ctime(&now), MTG_VERSION, sample.c_str(), prefix.c_str());
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
    void count(size_t nsol) { nb_breakpoints++; if (nsol > 0) { nb_filled++; if (nsol > 1) nb_multiple++; } }
};

/* The reference hands groups of records to its Dispatcher threads (src/Filler.cpp:824,844); here the sites go in batches to the devices:
 * one host thread per device (each with its replica of the index) takes the next batch, and the calling thread writes the batches' records
 * in input order as they become complete -- the reference's order with -nb-cores 1. */
struct Replicas {
    std::vector<const mtg_index*> idx; /* idx[0] = the index itself */
    std::vector<mtg_index*> owned;
    ~Replicas() { for (mtg_index* i : owned) mtg_index_free(i); }
    int make(const mtg_index* primary, int want)
    {
        idx.push_back(primary);
        const int ndev = mtg_device_count();
        if (const char* e = getenv("MTG_NB_GPUS")) { if (want <= 0) want = atoi(e); }
        int use = want > 0 ? std::min(want, ndev) : ndev;
        for (int d = 0, made = 1; d < ndev && made < use; d++) {
            if (d == primary->device) continue;
            mtg_index* r = nullptr;
            if (int rc = mtg_index_replicate(primary, d, &r)) return rc;
            owned.push_back(r);
            idx.push_back(r);
            made++;
        }
        return MTG_OK;
    }
};
static size_t cli_batch_size()
{
    const char* e = getenv("MTG_CLI_BATCH");
    const long v = e ? atol(e) : 0;
    return v > 0 ? (size_t)v : (size_t)200000;
}
/* process(b, idx) for every batch b on the devices; consume(b) on the calling thread, in order, once batch b is complete */
static int run_batches(const Replicas& R, size_t nb, const std::function<int(size_t, const mtg_index*)>& process, const std::function<void(size_t)>& consume)
{
    std::atomic<size_t> next{0};
    std::vector<char> done(nb, 0);
    std::mutex m;
    std::condition_variable cv;
    std::atomic<int> err{MTG_OK};
    std::string err_text;
    std::vector<std::thread> workers;
    for (size_t d = 0; d < R.idx.size() && d < std::max<size_t>(nb, 1); d++)
        workers.emplace_back([&, d]() {
            for (;;) {
                const size_t b = next.fetch_add(1);
                if (b >= nb) return;
                int rc = err.load() ? err.load() : process(b, R.idx[d]);
                std::lock_guard<std::mutex> lk(m);
                if (rc && !err.load()) { err = rc; err_text = mtg_last_error(); } /* the message is thread-local: keep the first one */
                done[b] = 1;
                cv.notify_all();
            }
        });
    for (size_t b = 0; b < nb; b++) {
        { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return done[b] != 0; }); }
        if (!err.load()) consume(b);
    }
    for (auto& t : workers) t.join();
    if (err.load()) set_error("%s", err_text.c_str());
    return err.load();
}

static int run_bkpt(const Replicas& R, const mtg_params& P, const Options& O, Files& F, Summary& S)
{
    std::vector<std::pair<std::string, std::string>> recs;
    if (!read_sequences(O.bkpt, recs)) { set_error("cannot read %s", O.bkpt.c_str()); return MTG_ERR_IO; }
    const size_t nsites = recs.size() / 2; /* records 2i / 2i+1 = left / right k-mer, src/Filler.cpp:625-629 */
    struct Site { std::string name, name_r; };
    struct Batch {
        size_t s0 = 0, s1 = 0;
        std::vector<Site> sites;
        std::vector<GapArgs> fwd, rev;
        std::vector<long> rev_idx;
        BatchRun rf, rr;
    };
    const size_t B = cli_batch_size(), nb = (nsites + B - 1) / B;
    std::vector<std::unique_ptr<Batch>> batches(nb);
    const auto process = [&](size_t b, const mtg_index* idx) -> int {
        std::unique_ptr<Batch> bt(new Batch());
        bt->s0 = b * B; bt->s1 = std::min(nsites, (b + 1) * B);
        const size_t n = bt->s1 - bt->s0;
        bt->sites.resize(n);
        bt->fwd.resize(n);
        for (size_t j = 0; j < n; j++) {
            const size_t i = bt->s0 + j;
            Site& s = bt->sites[j];
            GapArgs& g = bt->fwd[j];
            g.source = recs[2 * i].second;
            g.target = recs[2 * i + 1].second;
            s.name = short_name(recs[2 * i].first);
            s.name_r = short_name(recs[2 * i + 1].first);
            g.repeated = recs[2 * i].first.find("REPEATED") != std::string::npos || recs[2 * i + 1].first.find("REPEATED") != std::string::npos;
            bkpt_dict_t dict;
            dict.insert({g.target, std::make_pair(s.name_r, false)});
            g.set_dict(dict);
        }
        int rc = bt->rf.run(idx, P, bt->fwd);
        if (rc) return rc;
        /* reverse attempt for the sites without solution, src/Filler.cpp:669-680 */
        bt->rev_idx.assign(n, -1);
        if (!O.fwd_only)
            for (size_t j = 0; j < n; j++)
                if (bt->rf[j].n_filled == 0) {
                    GapArgs g;
                    g.target = revcomp_str(bt->fwd[j].source);
                    g.source = revcomp_str(bt->fwd[j].target);
                    g.repeated = bt->fwd[j].repeated;
                    g.reverse = true;
                    bkpt_dict_t dict;
                    dict.insert({g.target, std::make_pair(bt->sites[j].name, false)});
                    g.set_dict(dict);
                    bt->rev_idx[j] = (long)bt->rev.size();
                    bt->rev.push_back(std::move(g));
                }
        if (!bt->rev.empty()) { rc = bt->rr.run(idx, P, bt->rev); if (rc) return rc; }
        batches[b] = std::move(bt);
        return MTG_OK;
    };
    const auto consume = [&](size_t b) {
        Batch& bt = *batches[b];
        for (size_t j = 0; j < bt.s1 - bt.s0; j++) {
            const Site& s = bt.sites[j];
            std::string info = info_string(bt.rf[j]);
            std::string name = s.name;
            const mtg_gap_result* res = &bt.rf[j];
            const GapArgs* gw = &bt.fwd[j];
            if (bt.rev_idx[j] >= 0) {
                res = &bt.rr[(size_t)bt.rev_idx[j]];
                info += info_string(*res);
                name = s.name_r; /* src/Filler.cpp:674 */
                gw = &bt.rev[(size_t)bt.rev_idx[j]];
            }
            const Sols sols = sols_of(*res);
            write_filled(F, true, *gw, sols, name, info);
            write_vcf(F, O.filter, sols, name, bt.fwd[j].source);
            if (sols.empty() && O.extend) {
                write_extension(F, bt.rf[j].extension, name, bt.fwd[j].source);
                write_extension(F, bt.rev_idx[j] >= 0 ? std::string(bt.rr[(size_t)bt.rev_idx[j]].extension) : std::string(), name + "_reverse", revcomp_str(bt.fwd[j].target));
            }
            S.count(sols.size());
        }
        batches[b].reset(); /* its records and sequences are written */
    };
    return run_batches(R, nb, process, consume);
}

static int run_contig(const Replicas& R, const mtg_params& P, const Options& O, Files& F, Summary& S, int trim)
{
    const int k = R.idx[0]->info.k;
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
    /* contigFunctor, src/Filler.cpp:492-572; every seed sees the targets of all the other contigs (:522-533) */
    struct Batch {
        size_t s0 = 0, s1 = 0;
        std::vector<GapArgs> gaps;
        BatchRun run;
    };
    const size_t B = std::max<size_t>(1, cli_batch_size() / std::max<size_t>(1, all_targets.size() / 8)), nb = (seeds.size() + B - 1) / B; /* a gap carries the whole dictionary: fewer per batch */
    std::vector<std::unique_ptr<Batch>> batches(nb);
    const auto process = [&](size_t b, const mtg_index* idx) -> int {
        std::unique_ptr<Batch> bt(new Batch());
        bt->s0 = b * B; bt->s1 = std::min(seeds.size(), (b + 1) * B);
        bt->gaps.resize(bt->s1 - bt->s0);
        for (size_t si = bt->s0; si < bt->s1; si++) {
            auto& sd = seeds[si];
            GapArgs& g = bt->gaps[si - bt->s0];
            bkpt_dict_t dict;
            for (auto its = all_targets.begin(); its != all_targets.end(); ++its) {
                std::string tempName = its->second.first;
                if (its->second.second) tempName += "_Rc";
                if (tempName.compare(sd.first) != 0) { g.target.append(its->first); dict.insert({its->first, its->second}); }
            }
            g.source = sd.second;
            g.set_dict(dict);
        }
        if (int rc = bt->run.run(idx, P, bt->gaps)) return rc;
        batches[b] = std::move(bt);
        return MTG_OK;
    };
    const auto consume = [&](size_t b) {
        Batch& bt = *batches[b];
        for (size_t i = bt.s0; i < bt.s1; i++) {
            const size_t j = i - bt.s0;
            const std::string& seedName = seeds[i].first;
            const bool isRc = seedName.length() >= 3 && seedName.compare(seedName.length() - 3, 3, "_Rc") == 0;
            std::vector<mtg_filled> kept;
            for (auto& s : sols_of(bt.run[j])) { /* drop loops: target == seed reversed, :540-557 */
                const std::string& tn = bt.gaps[j].tname[s.target_index];
                const std::string revTargetName = bt.gaps[j].trc[s.target_index] ? tn : tn + "_Rc";
                if (revTargetName != seedName) kept.push_back(s);
            }
            Sols ks;
            ks.p = kept.data(); ks.n = kept.size();
            write_filled(F, false, bt.gaps[j], ks, seedName, info_string(bt.run[j]));
            write_gfa(F, trim, bt.gaps[j], ks, seedName, isRc);
            if (kept.empty() && O.extend) write_extension(F, bt.run[j].extension, seedName, seeds[i].second);
            S.count(kept.size());
        }
        batches[b].reset();
    };
    return run_batches(R, nb, process, consume);
}

int fill_main(int argc, const char* const* argv)
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
    if (O.graph.empty() == O.in.empty()) { fprintf(stderr, "EXCEPTION: options -graph and -in are incompatible, but at least one of these is mandatory\n"); return 1; }
    if (O.bkpt.empty() == O.contig.empty()) { fprintf(stderr, "EXCEPTION: option -bkpt and -contig are incompatible, but at least one of these is mandatory\n"); return 1; }
    if (!O.has_out) { /* src/Filler.cpp:154-165 */
        time_t now = time(0);
        struct tm tstruct = *localtime(&now);
        char buf[80];
        strftime(buf, sizeof(buf), "%Y-%m-%d.%I:%M", &tstruct);
        O.out = std::string("MindTheGap_Expe-") + buf;
    }
    mtg_index* idx = nullptr;
    int rc;
    if (!O.in.empty()) {
        rc = index_from_reads(O.in.c_str(), O.k, O.abundance_min, O.abundance_max, &idx);
        if (!rc) (void)index_save(idx, (O.out + ".mtgidx").c_str()); /* the reference leaves <out>.h5 behind */
    } else {
        fprintf(stderr, "Loading the graph...");
        rc = index_load(O.graph.c_str(), &idx);
        if (!rc) fprintf(stderr, "done\n");
    }
    if (rc) { fprintf(stderr, "EXCEPTION: %s\n", mtg_last_error()); return 1; }
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
    if (!open_w(F.insert, insert_name) || !open_w(F.info, info_name)) { mtg_index_free(idx); return 1; }
    if (bkpt_mode) { if (!open_w(F.vcf, vcf_name)) { mtg_index_free(idx); return 1; } write_vcf_header(F.vcf, O.in.empty() ? O.graph : O.in, O.out); }
    else if (!open_w(F.gfa, gfa_name)) { mtg_index_free(idx); return 1; }
    if (O.extend && !open_w(F.ext, ext_name)) { mtg_index_free(idx); return 1; }
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
    if (rc) { fprintf(stderr, "EXCEPTION: %s\n", mtg_last_error()); mtg_index_free(idx); return 1; }
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
    mtg_index_free(idx);
    return 0;
}

} // namespace mtgi

extern "C" int mtg_fill_main(int argc, const char* const* argv) { return mtgi::fill_main(argc, argv); }
