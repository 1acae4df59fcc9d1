/*
 * mtg_oracle.cpp -- CPU ORACLE for the MindTheGap `fill` hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see mtg_oracle.h).  Never linked into, imported by or called from the
 * product.  It exists to (1) be pinned against the reference's golden files, (2) check the HIP path
 * bit for bit, (3) be timed as the "port" CPU baseline by bench.py.
 *
 * Citation convention: `src/...:line` is relative to /root/reference.  [MEM] marks behaviour of
 * GATB/gatb-core (absent dependency, versions 1.4.1/1.4.2 pinned by the goldens) restated from its
 * published algorithm as distilled in SURVEY.md Appendix A; everything else follows in-tree code.
 *
 * Parity status: PINNED by tests/golden (full_test FASTA+VCF incl. headers, contig_test GFA/FASTA/
 * seed dictionary, nb_solid_kmers / nb_branching_nodes KATs).  UNPINNED (no golden exercises them):
 * multi-path bubbles, max_nodes/max_depth cut-offs, abundance > 70, auto cut-off beyond one datapoint.
 */
#include "mtg_oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <fstream>
#include <iostream>
#include <map>
#include <mutex>
#include <queue>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include <chrono>
#include <zlib.h>

using namespace std;

namespace {

typedef uint64_t kmer_t;

/* ---------------------------------------------------------------------------------------------
 * k-mer model  [MEM] gatb/kmer/impl/Model.hpp : A=0 C=1 T=2 G=3 = (ascii>>1)&3, invalid = bit 3 of
 * the ASCII code ('N'), canonical = min(forward, revcomp) as integers, first nt most significant.
 * SURVEY Appendix A.1.
 * ------------------------------------------------------------------------------------------- */
static inline int nt2int(unsigned char c) { return (c >> 1) & 3; }
static inline bool nt_invalid(unsigned char c) { return (c >> 3) & 1; }
static const char INT2NT[4] = {'A', 'C', 'T', 'G'};

static inline kmer_t kmask(int k) { return (k >= 32) ? ~0ULL : ((1ULL << (2 * k)) - 1); }

static inline kmer_t revcomp(kmer_t x, int k)
{
    x ^= 0xAAAAAAAAAAAAAAAAULL; /* complement: A<->T (0<->2), C<->G (1<->3) */
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    x = __builtin_bswap64(x);
    return x >> (64 - 2 * k);
}
static inline kmer_t canon(kmer_t x, int k) { kmer_t r = revcomp(x, k); return r < x ? r : x; }

static kmer_t encode(const char* s, int k)
{
    kmer_t x = 0;
    for (int i = 0; i < k; i++) x = (x << 2) | nt2int((unsigned char)s[i]);
    return x;
}
static string decode(kmer_t x, int k)
{
    string s(k, 'A');
    for (int i = k - 1; i >= 0; i--) { s[i] = INT2NT[x & 3]; x >>= 2; }
    return s;
}
static inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

/* ---------------------------------------------------------------------------------------------
 * Sequence files: FASTA / FASTQ, optionally gzipped (gatb BankFasta [MEM]; name = header up to the
 * first space = getCommentShort, comment = whole header line).
 * ------------------------------------------------------------------------------------------- */
struct SeqRecord { string comment; string seq; };

static bool gz_getline(gzFile f, string& line)
{
    line.clear();
    char buf[65536];
    bool got = false;
    while (gzgets(f, buf, sizeof buf)) {
        got = true;
        size_t n = strlen(buf);
        if (n && buf[n - 1] == '\n') { line.append(buf, n - 1); if (!line.empty() && line.back() == '\r') line.pop_back(); return true; }
        line.append(buf, n);
    }
    return got;
}

static bool read_seq_file(const string& path, vector<SeqRecord>& out)
{
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) return false;
    gzbuffer(f, 1 << 20);
    string line;
    bool have = gz_getline(f, line);
    while (have) {
        if (line.empty()) { have = gz_getline(f, line); continue; }
        if (line[0] == '>') {
            SeqRecord r; r.comment = line.substr(1);
            while ((have = gz_getline(f, line)) && (line.empty() || line[0] != '>')) r.seq += line;
            out.push_back(std::move(r));
        } else if (line[0] == '@') {
            SeqRecord r; r.comment = line.substr(1);
            have = gz_getline(f, line); if (have) r.seq = line;
            have = gz_getline(f, line); /* + */
            have = gz_getline(f, line); /* qual */
            have = gz_getline(f, line);
            out.push_back(std::move(r));
        } else {
            have = gz_getline(f, line);
        }
    }
    gzclose(f);
    return true;
}
static string comment_short(const string& c) { size_t p = c.find(' '); return p == string::npos ? c : c.substr(0, p); }

} // namespace

/* ---------------------------------------------------------------------------------------------
 * Index: exact set of solid canonical k-mers with abundance.  SURVEY 4.3-1: the graph's node set is
 * precisely "canonical k-mers (no N) with summed abundance >= threshold"; gatb's Bloom + cFP is an
 * exact-set implementation detail for neighbour queries from solid nodes.
 * ------------------------------------------------------------------------------------------- */
struct mtgo_index {
    int k = 31;
    vector<kmer_t> keys;     /* open addressing, EMPTY = ~0 */
    vector<uint32_t> vals;
    size_t cap_mask = 0;
    size_t n = 0;
    int abundance_min = 0;
    int auto_cutoff = -1;

    void init(size_t nkeys)
    {
        size_t cap = 16;
        while (cap < nkeys * 2) cap <<= 1;
        keys.assign(cap, ~0ULL); vals.assign(cap, 0); cap_mask = cap - 1; n = 0;
    }
    void put(kmer_t c, uint32_t v)
    {
        size_t h = splitmix64(c) & cap_mask;
        while (keys[h] != ~0ULL) { if (keys[h] == c) { vals[h] = v; return; } h = (h + 1) & cap_mask; }
        keys[h] = c; vals[h] = v; n++;
    }
    inline uint32_t get(kmer_t c) const
    {
        size_t h = splitmix64(c) & cap_mask;
        while (keys[h] != ~0ULL) { if (keys[h] == c) return vals[h]; h = (h + 1) & cap_mask; }
        return 0;
    }
};

namespace {

/* [MEM] gatb Histogram::compute_threshold (auto cut-off): smooth with a 3-window, take the first
 * local minimum as the cut-off, floor at STR_KMER_ABUNDANCE_MIN_THRESHOLD=3 (src/Filler.cpp:201).
 * Pinned by ONE datapoint only: 7 on data/reads_r{1,2}.fastq (test/full_test/gold_fill.output:11). */
static int auto_cutoff_from_histogram(const vector<uint64_t>& histo /* index = abundance */, int floor_thr)
{
    size_t len = histo.size();
    if (len < 5) return floor_thr;
    /* gatb keeps the smoothed histogram in integer entries (Histogram::Entry::abundance is a u_int64_t [MEM]): the weighted sums are
     * truncated.  Pinned by two datapoints: the golden cut-off 7 (test/full_test/gold_fill.output:11) and test/simple_test.sh's
     * "clean-insert" case, whose fill only succeeds when the cut-off inferred from reads/master.fasta (a sparse histogram of counts
     * 0..3) is at most 33 -- without the truncation the first minimum of that histogram is at 68. */
    vector<uint64_t> sm(len, 0);
    sm[1] = (uint64_t)(0.6 * histo[1] + 0.4 * histo[2]);
    for (size_t i = 2; i + 1 < len; i++) sm[i] = (uint64_t)(0.2 * histo[i - 1] + 0.6 * histo[i] + 0.2 * histo[i + 1]);
    size_t valley = 2;
    while (valley + 2 < len && !(sm[valley] < sm[valley + 1])) valley++; /* first minimum */
    size_t peak = valley;
    for (size_t i = valley; i + 1 < len; i++) if (sm[i] > sm[peak]) peak = i; /* coverage peak */
    size_t best = valley;
    for (size_t i = valley; i <= peak; i++) if (sm[i] < sm[best]) best = i; /* arg-min between */
    int cutoff = (int)best;
    if (cutoff < floor_thr) cutoff = floor_thr;
    return cutoff;
}

static void count_kmers_of_seq(const string& s, int k, vector<kmer_t>& out)
{
    kmer_t fwd = 0, mask = kmask(k);
    int valid = 0;
    for (size_t i = 0; i < s.size(); i++) {
        unsigned char c = (unsigned char)s[i];
        if (nt_invalid(c)) { valid = 0; fwd = 0; continue; }
        fwd = ((fwd << 2) | nt2int(c)) & mask;
        if (++valid >= k) out.push_back(canon(fwd, k));
    }
}

/* ---------------------------------------------------------------------------------------------
 * Graph view: neighbour queries + algorithmic probe counters.
 * [MEM] gatb Graph::successors/predecessors/degree/simplePathAvance/queryAbundance, SURVEY A.2.
 * Neighbour enumeration order is A,C,T,G on the traversal strand.
 * A "neighbour evaluation" = 4 membership probes.  A 2-entry memo removes the re-evaluation gatb
 * performs inside BranchingTerminator::mark so that the counter is the algorithmic minimum of
 * 8 probes per simple-path nucleotide.
 * ------------------------------------------------------------------------------------------- */
struct GraphView {
    const mtgo_index* idx;
    int k;
    kmer_t mask;
    uint64_t probes = 0, abund_lookups = 0;
    struct Memo { kmer_t x; int dir; uint8_t m; bool valid; } memo[2];
    int memo_next = 0;

    explicit GraphView(const mtgo_index* i) : idx(i), k(i->k), mask(kmask(i->k)) { memo[0].valid = memo[1].valid = false; }

    inline kmer_t neighbor(kmer_t x, int dir, int nt) const
    {
        return dir == 0 ? (((x << 2) | (kmer_t)nt) & mask) : ((x >> 2) | ((kmer_t)nt << (2 * (k - 1))));
    }
    inline bool contains(kmer_t x) const { return idx->get(canon(x, k)) != 0; }
    uint8_t nbr_mask(kmer_t x, int dir)
    {
        for (int i = 0; i < 2; i++) if (memo[i].valid && memo[i].x == x && memo[i].dir == dir) return memo[i].m;
        uint8_t m = 0;
        for (int nt = 0; nt < 4; nt++) if (contains(neighbor(x, dir, nt))) m |= (uint8_t)(1 << nt);
        probes += 4;
        memo[memo_next] = Memo{x, dir, m, true}; memo_next ^= 1;
        return m;
    }
    inline int degree(kmer_t x, int dir) { return __builtin_popcount(nbr_mask(x, dir)); }
    inline bool is_branching(kmer_t x) { return !(degree(x, 0) == 1 && degree(x, 1) == 1); }
    /* 8-bit abundance saturating at 255 (gatb stores 8 bits and discretises above ~70: unpinned) */
    inline uint32_t abundance(kmer_t x) { abund_lookups++; uint32_t a = idx->get(canon(x, k)); return a > 255 ? 255 : a; }
    inline kmer_t can(kmer_t x) const { return canon(x, k); }
};

/* [MEM] gatb BranchingTerminator.  Only the node bit of BRANCHING k-mers is ever read
 * back on this path (Frontline: is_branching && is_marked_branching); the marked-extension bits that
 * gatb also records are write-only here, so the state is the set of marked branching k-mers. */
struct Terminator {
    GraphView& g;
    unordered_set<kmer_t> marked;
    explicit Terminator(GraphView& gv) : g(gv) {}
    void reset() { marked.clear(); }
    void mark(kmer_t x) { if (g.is_branching(x)) marked.insert(g.can(x)); }
    bool is_marked_branching(kmer_t x) const { return marked.count(g.can(x)) != 0; }
};

/* needleman_wunsch: src/Utils.cpp:87-189 (gatb's Traversal carries the same routine [MEM]). */
static float needleman_wunsch(const string& a, const string& b)
{
    float gap_score = -5, mismatch_score = -5, match_score = 10;
    int n_a = (int)a.length(), n_b = (int)b.length();
    vector<vector<float>> score(n_a + 1, vector<float>(n_b + 1));
    for (int i = 0; i <= n_a; i++) score[i][0] = gap_score * i;
    for (int j = 0; j <= n_b; j++) score[0][j] = gap_score * j;
    for (int i = 1; i <= n_a; i++)
        for (int j = 1; j <= n_b; j++) {
            float match = score[i - 1][j - 1] + ((a[i - 1] == b[j - 1]) ? match_score : mismatch_score);
            float del = score[i - 1][j] + gap_score;
            float insert = score[i][j - 1] + gap_score;
            score[i][j] = max(max(match, del), insert);
        }
    int i = n_a, j = n_b;
    float identity = 0;
    while (i > 0 && j > 0) {
        float score_current = score[i][j], score_diagonal = score[i - 1][j - 1], score_up = score[i][j - 1], score_left = score[i - 1][j];
        if (score_current == score_diagonal + ((a[i - 1] == b[j - 1]) ? match_score : mismatch_score)) {
            if (a[i - 1] == b[j - 1]) identity++;
            i -= 1; j -= 1;
        } else {
            if (score_current == score_left + gap_score) i -= 1;
            else if (score_current == score_up + gap_score) j -= 1;
        }
    }
    identity /= max(n_a, n_b);
    return identity;
}

/* ---------------------------------------------------------------------------------------------
 * [MEM] gatb Frontline / FrontlineBranching.
 * ------------------------------------------------------------------------------------------- */
struct Frontline {
    GraphView& g; Terminator& term; int dir;
    deque<pair<kmer_t, int>> front;  /* (oriented node, first nt on the path from the start) */
    unordered_set<kmer_t> seen;      /* canonical, "already_frontlined" */
    unordered_set<kmer_t>* involved; /* canonical */
    int depth = 0;
    bool branching_mode;

    Frontline(GraphView& g_, Terminator& t, int dir_, kmer_t start, kmer_t prev_canon, unordered_set<kmer_t>* inv, bool bm)
        : g(g_), term(t), dir(dir_), involved(inv), branching_mode(bm)
    {
        seen.insert(g.can(start));
        seen.insert(prev_canon);
        front.push_back({start, -1});
    }
    size_t size() const { return front.size(); }

    bool check(kmer_t m)
    {
        if (!branching_mode) return true;
        uint8_t pm = g.nbr_mask(m, 1 - dir);
        for (int nt = 0; nt < 4; nt++) {
            if (!(pm & (1 << nt))) continue;
            kmer_t b = g.neighbor(m, 1 - dir, nt);
            if (seen.count(g.can(b))) continue;
            Frontline inner(g, term, 1 - dir, b, g.can(m), involved, false);
            do {
                bool cont = inner.go_next_depth();
                if (!cont) break;
                if (inner.depth > 3 * g.k) break;
                if (inner.size() > 10) break;
                if (inner.size() == 0) break;
            } while (1);
            if (inner.size() > 0) return false; /* large in-branching */
        }
        return true;
    }

    bool go_next_depth()
    {
        deque<pair<kmer_t, int>> next;
        while (!front.empty()) {
            pair<kmer_t, int> cur = front.front();
            front.pop_front();
            if (depth > 0 && !check(cur.first)) return false;
            uint8_t nm = g.nbr_mask(cur.first, dir);
            for (int nt = 0; nt < 4; nt++) {
                if (!(nm & (1 << nt))) continue;
                kmer_t nb = g.neighbor(cur.first, dir, nt);
                kmer_t cn = g.can(nb);
                if (seen.count(cn)) continue;
                if (term.is_marked_branching(nb)) return false; /* bubble touches an assembled region */
                int from_nt = cur.second < 0 ? nt : cur.second;
                next.push_back({nb, from_nt});
                seen.insert(cn);
                if (involved) involved->insert(cn);
            }
        }
        front = next;
        ++depth;
        return true;
    }
};

/* ---------------------------------------------------------------------------------------------
 * [MEM] gatb MonumentTraversal (TRAVERSAL_CONTIG at src/Filler.cpp:867), SURVEY A.5.
 * ------------------------------------------------------------------------------------------- */
struct Monument {
    GraphView& g; Terminator& term;
    int max_depth = 500, max_breadth = 20;
    int end_rule_nonbranching = 0;
    static const long long MAXLEN = 10LL * 1000 * 1000;

    Monument(GraphView& gv, Terminator& t) : g(gv), term(t) {}

    /* returns 1 and nt when (cur) has exactly one successor whose in-degree is 1 */
    int simple_path_avance(kmer_t cur, int& nt_out)
    {
        uint8_t sm = g.nbr_mask(cur, 0);
        int outdeg = __builtin_popcount(sm);
        if (outdeg == 1) {
            int nt = __builtin_ctz(sm);
            kmer_t nx = g.neighbor(cur, 0, nt);
            if (g.degree(nx, 1) > 1) return -2;
            nt_out = nt;
            return 1;
        }
        if (outdeg > 1) return -1;
        return 0;
    }

    int find_end_of_branching(kmer_t start, kmer_t prev_canon, kmer_t& end_node, unordered_set<kmer_t>& involved)
    {
        Frontline fl(g, term, 0, start, prev_canon, &involved, true);
        do {
            bool cont = fl.go_next_depth();
            if (!cont) return 0;
            if (fl.depth > max_depth) return 0;
            if ((int)fl.size() > max_breadth) return 0;
            if (fl.size() == 0) return 0;
            if (fl.size() == 1 && (!end_rule_nonbranching || !g.is_branching(fl.front.front().first))) break;
        } while (1);
        end_node = fl.front.front().first;
        return fl.depth;
    }

    void all_consensuses_between(kmer_t start, kmer_t end_canon, int traversal_depth, vector<kmer_t>& used,
                                 vector<int>& current, set<vector<int>>& out, bool& success)
    {
        if (traversal_depth < -1) { success = false; return; }
        if (g.can(start) == end_canon) { out.insert(current); return; }
        uint8_t sm = g.nbr_mask(start, 0);
        for (int nt = 0; nt < 4; nt++) {
            if (!(sm & (1 << nt))) continue;
            kmer_t nx = g.neighbor(start, 0, nt);
            kmer_t cn = g.can(nx);
            if (find(used.begin(), used.end(), cn) != used.end()) { success = false; return; }
            current.push_back(nt); used.push_back(cn);
            all_consensuses_between(nx, end_canon, traversal_depth - 1, used, current, out, success);
            current.pop_back(); used.pop_back();
            if ((int)out.size() > max_breadth) success = false;
            if (!success) return;
        }
    }

    bool validate_consensuses(kmer_t start, const set<vector<int>>& cons, vector<int>& result)
    {
        if (cons.empty()) return false;
        int mean = 0;
        for (auto& c : cons) mean += (int)c.size();
        mean /= (int)cons.size();
        double stdev = 0;
        for (auto& c : cons) { int l = (int)c.size(); stdev += pow(fabs((double)(l - mean)), 2); }
        stdev = sqrt(stdev / cons.size());
        if (mean > max_depth) return false;
        if (cons.size() == 1 && mean > g.k + 1) return false; /* long dead-end alternative */
        if (stdev > mean / 5) return false;
        /* all pairs >= 90 % identity */
        vector<string> strs;
        for (auto& c : cons) { string s; for (int nt : c) s.push_back(INT2NT[nt]); strs.push_back(s); }
        for (size_t a = 0; a < strs.size(); a++)
            for (size_t b = a + 1; b < strs.size(); b++)
                if (needleman_wunsch(strs[a], strs[b]) * 100 < 90) return false;
        /* most abundant consensus (MPHF available): integer mean abundance over its first len k-mers */
        unsigned long best = 0;
        vector<int> chosen;
        for (auto& c : cons) {
            if (c.empty()) continue;
            unsigned long mean_ab = 0;
            kmer_t x = start;
            for (size_t i = 0; i < c.size(); i++) {
                uint32_t ab = g.abundance(x);
                mean_ab += (unsigned char)(ab > 255 ? 255 : ab);
                x = g.neighbor(x, 0, c[i]);
            }
            mean_ab /= c.size();
            if (mean_ab > best) { best = mean_ab; chosen = c; }
        }
        if ((int)chosen.size() > max_depth) return false;
        result = chosen;
        return true;
    }

    bool explore_branching(kmer_t cur, kmer_t prev_canon, vector<int>& consensus)
    {
        unordered_set<kmer_t> involved;
        kmer_t end_node = 0;
        int d = find_end_of_branching(cur, prev_canon, end_node, involved);
        if (getenv("MTGO_TRACE")) fprintf(stderr, "EB cur=%llx d=%d end=%llx ninv=%zu\n", (unsigned long long)cur, d, (unsigned long long)end_node, involved.size());
        if (!d) return false;
        set<vector<int>> cons;
        vector<kmer_t> used; used.push_back(g.can(cur));
        vector<int> current;
        bool success = true;
        all_consensuses_between(cur, g.can(end_node), d + 1, used, current, cons, success);
        if (!success) return false;
        consensus.clear();
        bool okv = validate_consensuses(cur, cons, consensus);
        if (getenv("MTGO_TRACE")) fprintf(stderr, "   ncons=%zu ok=%d len=%zu\n", cons.size(), (int)okv, consensus.size());
        if (!okv) return false;
        /* mark every involved extension; node bit only matters for branching ones */
        for (kmer_t c : involved) if (g.is_branching(c)) term.marked.insert(c);
        return true;
    }

    int avance(kmer_t cur, kmer_t prev_canon, vector<int>& path)
    {
        int nt;
        if (simple_path_avance(cur, nt) > 0) { path.assign(1, nt); return 1; }
        if (!explore_branching(cur, prev_canon, path)) return 0;
        return (int)path.size();
    }

    /* [MEM] Traversal::traverse.  The start node itself is not marked. */
    void traverse(kmer_t start, kmer_t& end_node, string& consensus)
    {
        kmer_t cur = start;
        kmer_t prev_canon = 0; /* gatb: default-constructed previousNode has k-mer value 0 */
        kmer_t start_canon = g.can(start);
        bool looping = false;
        vector<int> path;
        consensus.clear();
        int nnt;
        while ((nnt = avance(cur, prev_canon, path)) > 0) {
            for (int i = 0; i < nnt; i++) {
                consensus.push_back(INT2NT[path[i]]);
                prev_canon = g.can(cur);
                cur = g.neighbor(cur, 0, path[i]);
                term.mark(cur);
                if (g.can(cur) == start_canon) looping = true;
            }
            if (looping) break;
            if ((long long)consensus.size() > MAXLEN) break;
        }
        end_node = cur;
    }
};

/* [MEM] gatb IterativeExtensions::construct_linear_seqs (call site src/Filler.cpp:884; ctor args
 * src/Filler.cpp:867: TRAVERSAL_CONTIG, until_max_depth, Breadth, dont_output_first_nucleotide=false,
 * max_depth, max_nodes), SURVEY A.6. */
static void construct_linear_seqs(GraphView& g, const mtgo_params& P, const string& L, const string& R, bool swf,
                                  vector<string>& contigs)
{
    int k = g.k;
    Terminator term(g);
    term.reset();
    Monument trav(g, term);
    trav.end_rule_nonbranching = P.end_rule_nonbranching;
    contigs.clear();
    if ((int)L.size() < k) return;
    struct NodeDepth { kmer_t node; int depth; };
    deque<NodeDepth> q;
    q.push_back({encode(L.c_str(), k), 0});
    unordered_set<kmer_t> already_extended_from;
    long long nbNodes = 0;
    while (!q.empty()) {
        NodeDepth ksd = q.front();
        q.pop_front();
        kmer_t end_node;
        string ext;
        trav.traverse(ksd.node, end_node, ext);
        int len_right = (int)ext.size();
        string seq = decode(ksd.node, k) + ext;
        contigs.push_back(seq);
        int node_len = len_right + k;
        nbNodes++;
        if (swf) {
            if (seq.find(R) != string::npos && ksd.depth > k) break;
        }
        if (nbNodes > P.max_nodes) break;
        if (ksd.depth + node_len > P.max_depth) continue;
        uint8_t sm = g.nbr_mask(end_node, 0);
        for (int nt = 0; nt < 4; nt++) {
            if (!(sm & (1 << nt))) continue;
            kmer_t s = g.neighbor(end_node, 0, nt);
            kmer_t cs = g.can(s);
            if (already_extended_from.find(cs) == already_extended_from.end()) {
                q.push_back({s, ksd.depth + len_right + 1});
                already_extended_from.insert(cs);
            }
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * In-tree reference types: src/Utils.hpp:43-104, src/Filler.hpp:44-71, src/GraphAnalysis.hpp:37.
 * ------------------------------------------------------------------------------------------- */
typedef pair<string, bool> bkpt_t;
typedef unordered_map<string, bkpt_t> bkpt_dict_t;
typedef vector<int> unlabeled_path;

struct filled_insertion_t {
    string seq; int nb_errors_in_anchor; float avg_coverage = 0; float median_coverage = 0; bkpt_t targetId_anchor;
    int qual = 0; int solution_count = 0; int solution_rank = 0;
    filled_insertion_t(string insert, int nb_errors, bkpt_t targetId) : seq(insert), nb_errors_in_anchor(nb_errors), targetId_anchor(targetId) {}
    void compute_qual(bool is_anchor_repeated) /* src/Utils.hpp:85-103 */
    {
        int quality = 50;
        if (is_anchor_repeated) quality = 25;
        if (solution_count > 1) quality = 15;
        if (nb_errors_in_anchor == 1) quality = 10;
        if (nb_errors_in_anchor == 2) quality = 5;
        qual = quality;
    }
};

struct info_node_t { /* src/Filler.hpp:44-71 */
    int node_id; int pos; int nb_errors; bkpt_t targetId;
    bool operator<(const info_node_t& o) const { if (node_id != o.node_id) return node_id < o.node_id; return pos < o.pos; }
};

static string revcomp_sequence(const string& dna) /* src/Utils.cpp:44-77 */
{
    string rc;
    for (auto it = dna.rbegin(); it != dna.rend(); ++it) {
        switch (*it) {
            case 'a': rc += "t"; break; case 't': rc += "a"; break; case 'c': rc += "g"; break; case 'g': rc += "c"; break;
            case 'A': rc += "T"; break; case 'T': rc += "A"; break; case 'C': rc += "G"; break; case 'G': rc += "C"; break;
        }
    }
    return rc;
}
static inline int identNT(char a, char b) { return ((a == b || a - b == 32 || a - b == -32) && a != 'N'); } /* src/Utils.cpp:81-84 */

static double median(vector<unsigned int>& v) /* src/Utils.cpp:241-254 */
{
    size_t n = v.size() / 2;
    nth_element(v.begin(), v.begin() + n, v.end());
    unsigned int vn = v[n];
    if (v.size() % 2 == 1) return vn;
    nth_element(v.begin(), v.begin() + n - 1, v.end());
    return 0.5 * (vn + v[n - 1]);
}

static void remove_almost_identical_solutions(vector<filled_insertion_t>& consensuses, int identity_threshold) /* src/Utils.cpp:208-238 */
{
    vector<filled_insertion_t> final_set;
    final_set.push_back(*consensuses.begin());
    for (auto it_a = consensuses.begin(); it_a != consensuses.end(); ++it_a) {
        bool found = false;
        for (auto it_b = final_set.begin(); it_b != final_set.end(); ++it_b) {
            if (it_a->seq.compare(it_b->seq) == 0 || needleman_wunsch(it_a->seq, it_b->seq) * 100 >= identity_threshold) {
                if (it_a->nb_errors_in_anchor < it_b->nb_errors_in_anchor) { it_b->seq = it_a->seq; it_b->nb_errors_in_anchor = it_a->nb_errors_in_anchor; }
                found = true;
                break;
            }
        }
        if (!found) final_set.push_back(*it_a);
    }
    consensuses = final_set;
}

/* Contig graph: src/IGraphOutput.cpp:54-86 (extremities), :97-133,144-179 (edges), src/GraphOutputDot.cpp
 * :144-164 (dot text) and src/GraphAnalysis.cpp:56-118 (parser).  Only labels "FF" survive the parser
 * (:98-105); an FF edge a->b exists iff suffix_{k-1}(a) == prefix_{k-1}(b) on the same strand.  The temp
 * file round trip is not reproduced. */
struct ContigGraph {
    int k; int nb_nodes = 0;
    vector<string> node_sequences;
    map<int, set<int>> in_edges, out_edges;
    static const size_t max_breadth = 20; /* src/GraphAnalysis.hpp:43 */

    ContigGraph(const vector<string>& contigs, int k_) : k(k_)
    {
        nb_nodes = (int)contigs.size();
        node_sequences = contigs;
        unordered_map<string, vector<int>> by_prefix;
        for (int j = 0; j < nb_nodes; j++) by_prefix[contigs[j].substr(0, k - 1)].push_back(j);
        for (int i = 0; i < nb_nodes; i++) {
            const string& s = contigs[i];
            auto it = by_prefix.find(s.substr(s.size() - (k - 1)));
            if (it == by_prefix.end()) continue;
            for (int j : it->second) {
                if (j == i && (int)s.size() == k - 1) continue; /* src/IGraphOutput.cpp:160 */
                out_edges[i].insert(j);
                in_edges[j].insert(i);
            }
        }
    }

    /* src/GraphAnalysis.cpp:244-326 */
    set<pair<unlabeled_path, bkpt_t>> find_all_paths_rev(int start_node, const set<info_node_t>& terms, unlabeled_path current_path,
                                                         int& nb_calls, bool& success, int& terminal_node, bkpt_t& target_id)
    {
        set<pair<unlabeled_path, bkpt_t>> paths;
        if (nb_calls++ > 10000000) { success = false; return paths; }
        if (start_node != terminal_node)
            for (auto it = terms.begin(); it != terms.end(); ++it)
                if (it->node_id == start_node) return paths;
        if (start_node == 0) { paths.insert(make_pair(current_path, target_id)); return paths; }
        for (auto it_edge = in_edges[start_node].begin(); it_edge != in_edges[start_node].end(); ++it_edge) {
            int next_node = *it_edge;
            if (find(current_path.begin(), current_path.end(), next_node) == current_path.end()) {
                unlabeled_path extended_path(current_path);
                extended_path.insert(extended_path.begin(), next_node);
                auto new_paths = find_all_paths_rev(next_node, terms, extended_path, nb_calls, success, terminal_node, target_id);
                paths.insert(new_paths.begin(), new_paths.end());
                if (paths.size() >= max_breadth) success = false;
            }
            if (success == false) return paths;
        }
        return paths;
    }
    /* src/GraphAnalysis.cpp:205-237 */
    set<pair<unlabeled_path, bkpt_t>> find_all_paths_rev(const set<info_node_t>& terms)
    {
        set<pair<unlabeled_path, bkpt_t>> all_paths;
        for (auto it = terms.begin(); it != terms.end(); ++it) {
            int terminal_node = it->node_id;
            bkpt_t target_id = it->targetId;
            bool success = true;
            unlabeled_path start_path; start_path.push_back(terminal_node);
            int nb_calls = 0;
            if (terminal_node == 0) { set<pair<unlabeled_path, bkpt_t>> the_path; the_path.insert(make_pair(start_path, target_id)); return the_path; }
            auto paths = find_all_paths_rev(terminal_node, terms, start_path, nb_calls, success, terminal_node, target_id);
            all_paths.insert(paths.begin(), paths.end());
        }
        return all_paths;
    }
    /* src/GraphAnalysis.cpp:331-460 */
    vector<filled_insertion_t> paths_to_sequences(const set<unlabeled_path>& paths, const set<info_node_t>& terms)
    {
        vector<filled_insertion_t> sequences;
        int errs_in_anchor = 0;
        bkpt_t targetId_anchor;
        size_t _sizeKmer = (size_t)k;
        for (auto it = paths.begin(); it != paths.end(); ++it) {
            unlabeled_path p = *it;
            string sequence;
            for (auto it_path = p.begin(); it_path != p.end(); ++it_path) {
                int node = *it_path;
                int pos_anchor = 0;
                string node_sequence = node_sequences[node];
                if (it_path == (p.end() - 1)) {
                    for (auto t = terms.begin(); t != terms.end(); ++t)
                        if (t->node_id == node) { pos_anchor = t->pos; errs_in_anchor = t->nb_errors; targetId_anchor = t->targetId; break; }
                    node_sequence = node_sequence.substr(0, pos_anchor);
                    if ((size_t)pos_anchor <= (_sizeKmer - 1)) {
                        sequence = sequence.substr(0, sequence.length() - ((_sizeKmer - 1) - pos_anchor));
                    } else {
                        if (it_path != p.begin()) node_sequence = node_sequence.substr(_sizeKmer - 1, node_sequence.npos);
                        else node_sequence = node_sequence.substr(_sizeKmer, node_sequence.npos);
                        sequence += node_sequence;
                    }
                    break;
                }
                if (it_path != p.begin()) node_sequence = node_sequence.substr(_sizeKmer - 1, node_sequence.npos);
                else node_sequence = node_sequence.substr(_sizeKmer, node_sequence.npos);
                sequence += node_sequence;
            }
            if (sequence.length() > 0) sequences.push_back(filled_insertion_t(sequence, errs_in_anchor, targetId_anchor));
        }
        return sequences;
    }
};

/* src/Filler.cpp:1294-1378 */
static set<info_node_t> find_nodes_containing_multiple_R(const bkpt_dict_t& targetDictionary, const vector<string>& contigs, int k, int nb_mis_allowed)
{
    set<info_node_t> terminal_nodes;
    long nodeNb = 0;
    for (const string& node : contigs) {
        int anchor_size = k;
        size_t nodelen = node.size();
        if (nodelen < (size_t)k) { nodeNb++; continue; }
        const char* nodeseq = node.c_str();
        int best_match = 0; bkpt_t best_id; int position = 0; bool arret = false;
        for (unsigned int j = 0; j < nodelen - k + 1 && !arret; j++) {
            for (auto it = targetDictionary.begin(); it != targetDictionary.end() && !arret; ++it) {
                const char* anchor = (it->first).c_str();
                int nbmatch = 0;
                for (int i = 0; i < anchor_size; i++) nbmatch += identNT(nodeseq[j + i], anchor[i]);
                if (nbmatch > best_match && nbmatch >= (anchor_size - nb_mis_allowed)) {
                    best_id = it->second; position = j; best_match = nbmatch;
                    if (nbmatch == anchor_size) { arret = true; break; }
                }
            }
        }
        if (best_match != 0) terminal_nodes.insert(info_node_t{(int)nodeNb, position, anchor_size - best_match, best_id});
        nodeNb++;
    }
    return terminal_nodes;
}

/* ---------------------------------------------------------------------------------------------
 * Filler (driver): src/Filler.cpp
 * ------------------------------------------------------------------------------------------- */
struct Filler {
    const mtgo_index* idx; mtgo_params P; int k; bool breakpointMode = true; int contig_trim_size = 0;
    FILE *insert_file = nullptr, *info_file = nullptr, *vcf_file = nullptr, *gfa_file = nullptr, *extension_file = nullptr;
    mutex mtx_insert, mtx_info, mtx_vcf, mtx_gfa, mtx_ext;
    atomic<int> nb_breakpoints{0}, nb_filled{0}, nb_multiple{0};
    atomic<uint64_t> probes{0}, abund_lookups{0};
    int nb_contigs = 0, nb_used_contigs = 0;

    /* src/Filler.cpp:854-1026 */
    void gapFillFromSource(GraphView& g, string& infostring, string sourceSequence, string targetSequence, vector<filled_insertion_t>& filledSequences,
                           bkpt_dict_t targetDictionary, bool is_anchor_repeated, bool reverse, string& extensionSequence)
    {
        int nb_mis_allowed = P.nb_mis_allowed;
        if (is_anchor_repeated) nb_mis_allowed = 0;
        vector<string> contigs;
        construct_linear_seqs(g, P, sourceSequence, targetSequence, true, contigs);
        /* load_nodes_extremities: src/IGraphOutput.cpp:82-83 */
        long totalnt = 0; for (auto& c : contigs) totalnt += (long)c.size();
        char buf[64];
        snprintf(buf, sizeof buf, "\t%i", (int)contigs.size()); infostring += buf;
        snprintf(buf, sizeof buf, "\t%i", (int)totalnt); infostring += buf;
        set<info_node_t> terminal_nodes_with_endpos = find_nodes_containing_multiple_R(targetDictionary, contigs, k, nb_mis_allowed);
        set<int> terminal_nodes;
        for (auto& t : terminal_nodes_with_endpos) terminal_nodes.insert(t.node_id);
        snprintf(buf, sizeof buf, "\t%d", (int)terminal_nodes.size()); infostring += buf;
        if (terminal_nodes.size() > 0) {
            ContigGraph graph(contigs, k);
            set<pair<unlabeled_path, bkpt_t>> paths = graph.find_all_paths_rev(terminal_nodes_with_endpos);
            unordered_map<string, set<unlabeled_path>> paths_to_compare;
            for (auto it = paths.begin(); it != paths.end(); it++) {
                string key = it->second.first;
                if (it->second.second) key += "_Rc";
                paths_to_compare[key].insert(it->first);
            }
            int nbTotal_filled_insertions = 0;
            for (auto it = paths_to_compare.begin(); it != paths_to_compare.end(); ++it) {
                set<unlabeled_path> current_paths = it->second;
                vector<filled_insertion_t> tmpSequences = graph.paths_to_sequences(current_paths, terminal_nodes_with_endpos);
                nbTotal_filled_insertions += (int)tmpSequences.size();
                if (tmpSequences.size() > 1) remove_almost_identical_solutions(tmpSequences, 90);
                int nb_reported_insertions = (int)tmpSequences.size();
                int solution_rank = 1;
                for (auto its = tmpSequences.begin(); its != tmpSequences.end(); ++its) {
                    /* coverage: src/Filler.cpp:959-988 */
                    string cseq = sourceSequence + its->seq;
                    vector<unsigned int> vec_abundances;
                    uint64_t sum = 0; int nbkmers = 0;
                    kmer_t fwd = 0, mask = kmask(k); int valid = 0;
                    for (size_t i = 0; i < cseq.size(); i++) {
                        unsigned char c = (unsigned char)cseq[i];
                        if (nt_invalid(c)) { valid = 0; fwd = 0; continue; }
                        fwd = ((fwd << 2) | nt2int(c)) & mask;
                        if (++valid >= k) {
                            unsigned int cov = g.abundance(fwd);
                            if (cov == 0) cerr << "WARNING Unknown kmer : " << decode(canon(fwd, k), k) << endl;
                            sum += cov; nbkmers++; vec_abundances.push_back(cov);
                        }
                    }
                    its->median_coverage = vec_abundances.empty() ? 0 : (float)median(vec_abundances);
                    its->avg_coverage = sum / (float)nbkmers;
                    its->solution_count = nb_reported_insertions;
                    its->solution_rank = solution_rank;
                    its->compute_qual(is_anchor_repeated);
                    if (reverse) its->seq = revcomp_sequence(its->seq);
                    solution_rank += 1;
                }
                filledSequences.insert(filledSequences.end(), tmpSequences.begin(), tmpSequences.end());
            }
            if ((nbTotal_filled_insertions > 0) | reverse) {
                snprintf(buf, sizeof buf, "\t%d", nbTotal_filled_insertions); infostring += buf;
                snprintf(buf, sizeof buf, "\t%d", (int)filledSequences.size()); infostring += buf;
            }
        } else {
            /* get_first_contig: src/Filler.cpp:1381-1407 */
            extensionSequence = "";
            if (!contigs.empty() && (int)contigs[0].size() > k) extensionSequence = contigs[0].substr(k);
        }
    }

    /* src/Filler.cpp:1029-1093 */
    void writeFilledBreakpoint(vector<filled_insertion_t>& filledSequences, string seedName, string info)
    {
        {
            lock_guard<mutex> lk(mtx_insert);
            for (auto it = filledSequences.begin(); it != filledSequences.end(); ++it) {
                string insertion = it->seq;
                int llen = (int)insertion.length();
                bkpt_t targetId = it->targetId_anchor;
                ostringstream osolu_i; osolu_i << "solution " << it->solution_rank << "/" << it->solution_count;
                string solu_i = it->solution_count > 1 ? osolu_i.str() : "";
                if (breakpointMode) {
                    /* the reference passes (name,len,qual,solu,avg,median) to "%s..%d..%i..%.2f..%.2f   %s" (src/Filler.cpp:1052-1054);
                       on x86-64 SysV the doubles come from XMM registers and the last %s from the next GP register, so the
                       visible result is NAME_len_L_qual_Q_avg_cov_A_median_cov_M   SOLU (test/full_test/gold.insertions.fasta:1) */
                    fprintf(insert_file, ">%s_len_%d_qual_%i_avg_cov_%.2f_median_cov_%.2f   %s\n", seedName.c_str(), llen, it->qual,
                            (double)it->avg_coverage, (double)it->median_coverage, solu_i.c_str());
                } else {
                    string targetName = targetId.first;
                    if (targetId.second) targetName.append("_Rc");
                    int cov = it->median_coverage + 0.5;
                    string insertionName = ">" + seedName + ";" + targetName + ";len_" + to_string(llen) + "_qual_" + to_string(it->qual) + "_median_cov_" + to_string(cov) + "\t" + solu_i + "\n";
                    fprintf(insert_file, "%s", insertionName.c_str());
                }
                fprintf(insert_file, "%.*s\n", (int)llen, insertion.c_str());
            }
        }
        if (filledSequences.size() > 0) { nb_filled++; if (filledSequences.size() > 1) nb_multiple++; }
        lock_guard<mutex> lk(mtx_info);
        fprintf(info_file, "%s\t%s\n", seedName.c_str(), info.c_str());
    }

    /* src/Filler.cpp:1095-1214 */
    void writeVcf(vector<filled_insertion_t>& filledSequences, string breakpointName, string sourceSequence)
    {
        lock_guard<mutex> lk(mtx_vcf);
        for (auto it = filledSequences.begin(); it != filledSequences.end(); ++it) {
            string insertion = it->seq;
            vector<char> left(sourceSequence.begin(), sourceSequence.end());
            vector<char> filled(it->seq.begin(), it->seq.end());
            int repeatSize = 0;
            int i = (int)left.size() - 1;
            int j = (int)filled.size() - 1;
            while (i > 0 && j >= 0) {
                if (left[i] == filled[j]) { repeatSize++; i -= 1; j -= 1; if (j == -1) j = (int)filled.size() - 1; }
                else break;
            }
            insertion = sourceSequence.substr(sourceSequence.size() - (repeatSize + 1), repeatSize + 1) + insertion;
            insertion = insertion.substr(0, insertion.size() - repeatSize);
            string ref = sourceSequence.substr(sourceSequence.size() - (repeatSize + 1), 1);
            string token; istringstream iss(breakpointName); vector<string> tokens;
            while (getline(iss, token, '_')) tokens.push_back(token);
            string bkpt = breakpointName, position = ".", chromosome = ".", GT = "./.", genotype = "";
            if (tokens.size() == 7) {
                bkpt = tokens[0]; int pos = atoi(tokens[3].c_str()) - repeatSize; position = to_string(pos);
                chromosome = tokens[1]; genotype = tokens[6]; GT = genotype.compare("HOM") == 0 ? "1/1" : "0/1";
            }
            if (tokens.size() == 8) {
                bkpt = tokens[0]; bkpt += tokens[2]; int pos = atoi(tokens[4].c_str()) - repeatSize; position = to_string(pos);
                chromosome = tokens[1]; genotype = tokens[7]; GT = genotype.compare("HOM") == 0 ? "1/1" : "0/1";
            }
            int qual = it->qual; int size = (int)(insertion.size() - ref.size()); int nsol = it->solution_count; int npos = repeatSize + 1;
            string filter = "PASS";
            if ((genotype == "HET" && nsol > 1) || (genotype == "HOM" && nsol > 1)) {
                if (P.filter) break;
                else filter = "LOW_QUAL";
            }
            fprintf(vcf_file, "%s\t%s\t%s\t%s\t%s\t.\t%s\tTYPE=INS;LEN=%i;QUAL=%i;NSOL=%i;NPOS=%i;AVK=%.2f;MDK=%.2f\tGT\t%s\n", chromosome.c_str(), position.c_str(),
                    bkpt.c_str(), ref.c_str(), insertion.c_str(), filter.c_str(), size, qual, nsol, npos, (double)it->avg_coverage, (double)it->median_coverage, GT.c_str());
        }
    }

    /* src/Filler.cpp:1216-1273 */
    void writeToGFA(vector<filled_insertion_t>& filledSequences, string sourceSequence, string seedName, bool isRc)
    {
        string seedDirection = "+", targetDirection, seedNameNode = seedName, targetNameNode;
        if (isRc) { seedName = seedName.substr(0, seedName.size() - 3); seedDirection = "-"; }
        lock_guard<mutex> lk(mtx_gfa);
        for (auto it = filledSequences.begin(); it != filledSequences.end(); ++it) {
            int qual = it->qual; string insertion = it->seq; int llen = (int)insertion.length();
            ostringstream osolu_i; osolu_i << "solution " << it->solution_rank << "/" << it->solution_count;
            string solu_i = it->solution_count > 1 ? osolu_i.str() : "";
            bkpt_t targetId = it->targetId_anchor; string targetName = targetId.first;
            if (targetId.second) { targetDirection = "-"; targetNameNode = targetName + "_Rc"; }
            else { targetDirection = "+"; targetNameNode = targetName; }
            int cov = it->median_coverage + 0.5;
            string nodeName = seedNameNode + ";" + targetNameNode + ";len_" + to_string(llen) + "_qual_" + to_string(qual) + "_median_cov_" + to_string(cov) + " " + solu_i;
            fprintf(gfa_file, "S\t%s\t%s\n", nodeName.c_str(), insertion.c_str());
            fprintf(gfa_file, "L\t%s\t%s\t%s\t+\t%iM\n", seedName.c_str(), seedDirection.c_str(), nodeName.c_str(), contig_trim_size);
            fprintf(gfa_file, "L\t%s\t+\t%s\t%s\t%iM\n", nodeName.c_str(), targetName.c_str(), targetDirection.c_str(), contig_trim_size);
        }
    }

    /* src/Filler.cpp:1275-1291 */
    void writeExtensions(string contigSeq, string seedName, string sourceSequence)
    {
        int llen = (int)contigSeq.length();
        if (llen > 0) {
            lock_guard<mutex> lk(mtx_ext);
            fprintf(extension_file, ">%s_len_%d source=%s\n", seedName.c_str(), llen, sourceSequence.c_str());
            fprintf(extension_file, "%.*s\n", (int)llen, contigSeq.c_str());
        }
    }

    /* breakpointFunctor::operator(): src/Filler.cpp:623-699 (one call per PAIR of records here) */
    void do_breakpoint(const SeqRecord& left, const SeqRecord& right)
    {
        GraphView g(idx);
        string sourceSequence = left.seq;
        string breakpointName = comment_short(left.comment);
        string infostring;
        bool begin_kmer_repeated = left.comment.find("REPEATED") != string::npos;
        string targetSequence = right.seq;
        string breakpointName_R = comment_short(right.comment);
        bool end_kmer_repeated = right.comment.find("REPEATED") != string::npos;
        bool is_anchor_repeated = begin_kmer_repeated || end_kmer_repeated;
        vector<filled_insertion_t> filledSequences;
        bkpt_dict_t targetDictionary;
        targetDictionary.insert({targetSequence, make_pair(breakpointName_R, false)});
        string extensionSequence, extensionSequenceRev;
        gapFillFromSource(g, infostring, sourceSequence, targetSequence, filledSequences, targetDictionary, is_anchor_repeated, false, extensionSequence);
        if (!P.fwd_only && filledSequences.size() == 0) {
            string targetSequence2 = revcomp_sequence(sourceSequence);
            targetDictionary.clear();
            targetDictionary.insert({targetSequence2, make_pair(breakpointName, false)});
            string sourceSequence2 = revcomp_sequence(targetSequence);
            breakpointName = breakpointName_R;
            gapFillFromSource(g, infostring, sourceSequence2, targetSequence2, filledSequences, targetDictionary, is_anchor_repeated, true, extensionSequenceRev);
        }
        writeFilledBreakpoint(filledSequences, breakpointName, infostring);
        writeVcf(filledSequences, breakpointName, sourceSequence);
        if (filledSequences.size() == 0 && P.extend) {
            writeExtensions(extensionSequence, breakpointName, sourceSequence);
            string sourceSequence2 = revcomp_sequence(targetSequence);
            writeExtensions(extensionSequenceRev, breakpointName + "_reverse", sourceSequence2);
        }
        nb_breakpoints++;
        probes += g.probes; abund_lookups += g.abund_lookups;
    }

    /* contigFunctor::operator(): src/Filler.cpp:492-572 */
    void do_seed(const SeqRecord& seed, const bkpt_dict_t& all_targetDictionary)
    {
        GraphView g(idx);
        string sourceSequence = seed.seq;
        string seedName = seed.comment;
        string infostring;
        bool isRc;
        if (seedName.length() < 3) isRc = false;
        else isRc = !seedName.compare(seedName.length() - 3, 3, "_Rc");
        string conc_targetSequence;
        bkpt_dict_t targetDictionary;
        for (auto its = all_targetDictionary.begin(); its != all_targetDictionary.end(); ++its) {
            string tempName = its->second.first;
            if (its->second.second) tempName += "_Rc";
            if (tempName.compare(seedName) != 0) { conc_targetSequence.append(its->first); targetDictionary.insert({its->first, its->second}); }
        }
        vector<filled_insertion_t> filledSequences;
        string extensionSequence;
        gapFillFromSource(g, infostring, sourceSequence, conc_targetSequence, filledSequences, targetDictionary, false, false, extensionSequence);
        for (auto it = filledSequences.begin(); it != filledSequences.end();) {
            string revTargetName;
            bkpt_t target = it->targetId_anchor;
            if (target.second) revTargetName = target.first; else revTargetName = target.first + "_Rc";
            if (revTargetName == seedName) it = filledSequences.erase(it); else ++it;
        }
        writeFilledBreakpoint(filledSequences, seedName, infostring);
        writeToGFA(filledSequences, sourceSequence, seedName, isRc);
        if (filledSequences.size() == 0 && P.extend) writeExtensions(extensionSequence, seedName, sourceSequence);
        nb_breakpoints++;
        probes += g.probes; abund_lookups += g.abund_lookups;
    }
};

/* gatb Dispatcher::iterate(it, functor, 30) [MEM]: worker threads pull groups of 30 records. */
template <typename F> static void dispatch(size_t n_items, int nb_cores, size_t group, F f)
{
    if (nb_cores <= 0) nb_cores = (int)thread::hardware_concurrency();
    if (nb_cores <= 1) { for (size_t i = 0; i < n_items; i++) f(i); return; }
    atomic<size_t> next{0};
    vector<thread> th;
    for (int t = 0; t < nb_cores; t++)
        th.emplace_back([&]() {
            for (;;) {
                size_t b = next.fetch_add(group);
                if (b >= n_items) break;
                size_t e = min(n_items, b + group);
                for (size_t i = b; i < e; i++) f(i);
            }
        });
    for (auto& t : th) t.join();
}

static void write_vcf_header(FILE* f, const string& sample, const string& out_prefix) /* src/Filler.cpp:349-383 */
{
    time_t current_time = time(NULL);
    char* c_time_string = ctime(&current_time);
    fprintf(f,
            "##fileformat=VCFv4.1\n##filedate=%s##source=MindTheGap fill version %s\n##SAMPLE=file:%s\n##REF=file:%s\n"
            "##INFO=<ID=TYPE,Number=1,Type=String,Description=\"INS\">\n##INFO=<ID=LEN,Number=1,Type=Integer,Description=\"variant size\">\n"
            "##INFO=<=QUAL,Number=.,Type=Integer,Description=\"Quality of the insertion\">\n"
            "##INFO=<=AVK,Number=.,Type=Float,Description=\"Average k-mer coverage along the insertion\">\n"
            "##INFO=<=MDK,Number=.,Type=Float,Description=\"Median k-mer coverage along the insertion\">\n"
            "##INFO=<=NSOL,Number=1,Type=String,Description=\"number of alternative insertion sequences for the breakpoint\">\n"
            "##INFO=<ID=NPOS,Number=1,Type=Integer,Description=\"number of alternative positions for the insertion site (= size of repeat (fuzzy) +1)\">\n"
            "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tG1\n",
            c_time_string, "2.3.0", sample.c_str(), out_prefix.c_str());
}

} // namespace

/* =============================================================================================
 * C ABI
 * =========================================================================================== */
extern "C" {

void mtgo_default_params(mtgo_params* p)
{
    p->max_nodes = 100; p->max_depth = 10000; p->nb_mis_allowed = 2; p->overlap = 0; p->fwd_only = 0; p->filter = 0; p->extend = 0;
    p->nb_cores = 1; p->end_rule_nonbranching = 0; p->seed_stride = 0;
}

mtgo_index* mtgo_index_from_files(const char* paths_csv, int k, int abundance_min, int abundance_max)
{
    if (k < 5 || k > 31) return nullptr;
    vector<kmer_t> all;
    string csv(paths_csv), tok;
    istringstream iss(csv);
    while (getline(iss, tok, ',')) {
        if (tok.empty()) continue;
        vector<SeqRecord> recs;
        if (!read_seq_file(tok, recs)) return nullptr;
        for (auto& r : recs) count_kmers_of_seq(r.seq, k, all);
    }
    sort(all.begin(), all.end());
    /* histogram (STR_HISTOGRAM_MAX 10000, src/Filler.cpp:200) */
    vector<uint64_t> histo(10001 + 2, 0);
    vector<pair<kmer_t, uint32_t>> uniq;
    for (size_t i = 0; i < all.size();) {
        size_t j = i;
        while (j < all.size() && all[j] == all[i]) j++;
        uint32_t c = (uint32_t)(j - i);
        histo[min<uint32_t>(c, 10001)]++;
        uniq.push_back({all[i], c});
        i = j;
    }
    mtgo_index* idx = new mtgo_index();
    idx->k = k;
    if (abundance_min < 0) { idx->auto_cutoff = auto_cutoff_from_histogram(histo, 3); abundance_min = idx->auto_cutoff; }
    idx->abundance_min = abundance_min;
    size_t nsolid = 0;
    for (auto& u : uniq) if ((int64_t)u.second >= abundance_min && (abundance_max <= 0 || (int64_t)u.second <= abundance_max)) nsolid++;
    idx->init(nsolid);
    for (auto& u : uniq) if ((int64_t)u.second >= abundance_min && (abundance_max <= 0 || (int64_t)u.second <= abundance_max)) idx->put(u.first, u.second);
    return idx;
}

mtgo_index* mtgo_index_from_kmers(const uint64_t* canon_kmers, const uint32_t* counts, size_t n, int k)
{
    mtgo_index* idx = new mtgo_index();
    idx->k = k; idx->init(n);
    for (size_t i = 0; i < n; i++) idx->put(canon_kmers[i], counts[i]);
    return idx;
}

/* synthetic abundance of the benchmark sets (include/mtg_fill.h: mtg_index_create_from_packed_device): span > 0: lo + hash % span;
 * span == 0: Poisson(24) by inversion from the 64-bit hash (T[i] = floor(P(X <= i) * 2^64)), at least lo */
static uint32_t synth_abundance(uint64_t c, uint32_t lo, uint32_t span)
{
    const uint64_t h = splitmix64(c);
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
    while (a < 64 && T[a] <= h) a++;
    return a > lo ? a : lo;
}
mtgo_index* mtgo_index_from_sequences(const char* const* seqs, size_t nseq, int k, uint32_t abund_lo, uint32_t abund_span)
{
    vector<kmer_t> all;
    for (size_t i = 0; i < nseq; i++) count_kmers_of_seq(string(seqs[i]), k, all);
    sort(all.begin(), all.end());
    all.erase(unique(all.begin(), all.end()), all.end());
    mtgo_index* idx = new mtgo_index();
    idx->k = k; idx->init(all.size());
    for (kmer_t c : all) idx->put(c, synth_abundance(c, abund_lo, abund_span));
    idx->abundance_min = (int)abund_lo;
    return idx;
}

void mtgo_index_free(mtgo_index* i) { delete i; }
int mtgo_index_k(const mtgo_index* i) { return i->k; }
size_t mtgo_index_size(const mtgo_index* i) { return i->n; }
int mtgo_index_abundance_min(const mtgo_index* i) { return i->abundance_min; }
int mtgo_index_auto_cutoff(const mtgo_index* i) { return i->auto_cutoff; }

size_t mtgo_index_export(const mtgo_index* idx, uint64_t* kmers, uint32_t* counts, size_t cap)
{
    vector<pair<kmer_t, uint32_t>> v;
    for (size_t i = 0; i < idx->keys.size(); i++) if (idx->keys[i] != ~0ULL) v.push_back({idx->keys[i], idx->vals[i]});
    sort(v.begin(), v.end());
    size_t n = min(cap, v.size());
    for (size_t i = 0; i < n; i++) { kmers[i] = v[i].first; counts[i] = v[i].second; }
    return v.size();
}

void mtgo_index_stats(const mtgo_index* idx, uint64_t* nb_solid, uint64_t* nb_branching)
{
    GraphView g(idx);
    uint64_t nb = 0;
    for (size_t i = 0; i < idx->keys.size(); i++) if (idx->keys[i] != ~0ULL && g.is_branching(idx->keys[i])) nb++;
    *nb_solid = idx->n; *nb_branching = nb;
}

void mtgo_contains_batch(const mtgo_index* idx, const uint64_t* kmers, size_t n, uint8_t* out)
{
    for (size_t i = 0; i < n; i++) out[i] = idx->get(canon(kmers[i], idx->k)) != 0;
}
void mtgo_abundance_batch(const mtgo_index* idx, const uint64_t* kmers, size_t n, uint32_t* out)
{
    for (size_t i = 0; i < n; i++) { uint32_t a = idx->get(canon(kmers[i], idx->k)); out[i] = a > 255 ? 255 : a; }
}

char* mtgo_stage_a(const mtgo_index* idx, const mtgo_params* P, const char* source, const char* target_R, uint64_t* probes_out)
{
    GraphView g(idx);
    vector<string> contigs;
    construct_linear_seqs(g, *P, source, target_R, true, contigs);
    string joined;
    for (size_t i = 0; i < contigs.size(); i++) { if (i) joined += "\n"; joined += contigs[i]; }
    if (probes_out) *probes_out = g.probes;
    char* r = (char*)malloc(joined.size() + 1);
    memcpy(r, joined.c_str(), joined.size() + 1);
    return r;
}

int mtgo_fill_files(const mtgo_index* idx, const mtgo_params* P, int mode, const char* input_path, const char* out_prefix,
                    const char* sample_name, uint64_t* stats, double* seconds)
{
    Filler F;
    F.idx = idx; F.P = *P; F.k = idx->k; F.breakpointMode = (mode == 0);
    int k = idx->k;
    string prefix(out_prefix);
    F.insert_file = fopen((prefix + ".insertions.fasta").c_str(), "w");
    F.info_file = fopen((prefix + ".info.txt").c_str(), "w");
    if (!F.insert_file || !F.info_file) return 1;
    if (mode == 0) { F.vcf_file = fopen((prefix + ".insertions.vcf").c_str(), "w"); if (!F.vcf_file) return 1; write_vcf_header(F.vcf_file, sample_name ? sample_name : "", prefix); }
    else { F.gfa_file = fopen((prefix + ".gfa").c_str(), "w"); if (!F.gfa_file) return 1; }
    if (P->extend) { F.extension_file = fopen((prefix + ".extensions.fasta").c_str(), "w"); if (!F.extension_file) return 1; }
    F.contig_trim_size = P->overlap; /* src/Filler.cpp:299-307 */
    if (F.contig_trim_size == 0) F.contig_trim_size = k;
    if (F.contig_trim_size < k) { F.contig_trim_size = k; cerr << "Warning :  the contig overlap parameter should be greater or equal to kmer size, setting it to " << k << endl; }

    vector<SeqRecord> recs;
    if (!read_seq_file(input_path, recs)) return 2;
    double t_fill = 0;
    if (mode == 0) {
        size_t npairs = recs.size() / 2;
        auto t0 = chrono::steady_clock::now();
        /* 30 records = 15 sites per task, src/Filler.cpp:844-845 */
        dispatch(npairs, P->nb_cores, 15, [&](size_t i) { F.do_breakpoint(recs[2 * i], recs[2 * i + 1]); });
        t_fill = chrono::duration<double>(chrono::steady_clock::now() - t0).count();
    } else {
        /* fillAny, contig mode: src/Filler.cpp:755-829 */
        int overlap = F.contig_trim_size;
        bkpt_dict_t seedDictionary, all_targetDictionary;
        vector<SeqRecord> seeds;
        ofstream seedFile(prefix + "_seed_dictionary.fasta");
        for (auto& r : recs) {
            string contigSequence = r.seq;
            F.nb_contigs++;
            string name = comment_short(r.comment);
            fprintf(F.gfa_file, "S\t%s\t%s\n", name.c_str(), contigSequence.c_str());
            if (contigSequence.size() > (size_t)(2 * overlap + k)) {
                string seedSequence_f = contigSequence.substr(contigSequence.size() - (overlap + k), k);
                string targetSequence_f = contigSequence.substr(overlap, k);
                string contigSequence_Rc = revcomp_sequence(contigSequence);
                string seedSequence_Rc = contigSequence_Rc.substr(contigSequence_Rc.size() - (overlap + k), k);
                string targetSequence_Rc = contigSequence_Rc.substr(overlap, k);
                seedDictionary.insert({{seedSequence_f, make_pair(name, false)}, {seedSequence_Rc, make_pair(name, true)}});
                all_targetDictionary.insert({{targetSequence_f, make_pair(name, false)}, {targetSequence_Rc, make_pair(name, true)}});
                seedFile << ">" + name + "\n"; seedFile << seedSequence_f << endl;
                seedFile << ">" + name + "_Rc\n"; seedFile << seedSequence_Rc << endl;
                seeds.push_back(SeqRecord{name, seedSequence_f});
                seeds.push_back(SeqRecord{name + "_Rc", seedSequence_Rc});
                F.nb_used_contigs++;
            } else {
                int limit = 2 * overlap + k;
                cerr << "Warning contig not used (too short: <= 2 x overlap + kmerSize = " << limit << " nt): " << name << " of size " << contigSequence.size() << " nt" << endl;
            }
        }
        seedFile.close();
        auto t0 = chrono::steady_clock::now();
        const size_t stride = P->seed_stride > 1 ? (size_t)P->seed_stride : 1;
        dispatch(seeds.size(), P->nb_cores, 30, [&](size_t i) { if (i % stride == 0) F.do_seed(seeds[i], all_targetDictionary); });
        t_fill = chrono::duration<double>(chrono::steady_clock::now() - t0).count();
    }
    fclose(F.insert_file); fclose(F.info_file);
    if (F.vcf_file) fclose(F.vcf_file);
    if (F.gfa_file) fclose(F.gfa_file);
    if (F.extension_file) fclose(F.extension_file);
    if (stats) {
        stats[0] = (uint64_t)F.nb_breakpoints.load(); stats[1] = (uint64_t)F.nb_filled.load(); stats[2] = (uint64_t)F.nb_multiple.load();
        stats[3] = F.probes.load(); stats[4] = F.abund_lookups.load(); stats[5] = (uint64_t)F.nb_contigs; stats[6] = (uint64_t)F.nb_used_contigs;
    }
    if (seconds) *seconds = t_fill;
    return 0;
}

void mtgo_free(void* p) { free(p); }
float mtgo_needleman_wunsch(const char* a, const char* b) { return needleman_wunsch(a, b); }

} /* extern "C" */
