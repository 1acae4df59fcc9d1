/* TEST-ONLY: several threads call mtg_fill_batch on ONE index at the same time, on the emulation build of the device code (tests/emu),
 * under ThreadSanitizer: workspace hand-out, the shared worker pool, the result cache and the per-batch state of the host code.
 * Every batch must equal the batch a single thread gets.  Prints OK. */
#include "../../include/mtg_fill.h"
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

static uint64_t canon_of(const char* s, int k)
{
    uint64_t f = 0, r = 0;
    for (int i = 0; i < k; i++) {
        const uint64_t c = ((unsigned char)s[i] >> 1) & 3;
        f = (f << 2) | c;
        r |= (c ^ 2) << (2 * i);
    }
    return f < r ? f : r;
}

int main()
{
    const int k = 31, nloci = 60, flank = 120;
    std::mt19937_64 rng(7);
    auto rnd = [&](size_t n) { std::string s(n, 'A'); for (auto& c : s) c = "ACGT"[rng() & 3]; return s; };
    std::vector<std::string> donor;
    std::vector<std::string> left, right, names;
    for (int i = 0; i < nloci; i++) {
        const std::string a = rnd(flank), ins = rnd(40 + (size_t)(rng() % 200)), b = rnd(flank);
        donor.push_back(a + ins + b);
        left.push_back(a.substr(a.size() - k));
        right.push_back(b.substr(0, k));
        names.push_back("bkpt" + std::to_string(i) + "_s" + std::to_string(i) + "_pos_1_fuzzy_0_HOM");
    }
    /* a locus with two alleles (a substitution in the insert): a bubble on the way */
    donor.push_back(donor[0]);
    donor.back()[flank + 20] = donor.back()[flank + 20] == 'A' ? 'C' : 'A';
    std::vector<uint64_t> km;
    std::vector<uint32_t> ab;
    for (const std::string& d : donor)
        for (size_t p = 0; p + k <= d.size(); p++) { km.push_back(canon_of(d.data() + p, k)); ab.push_back(5 + (uint32_t)(p % 7)); }
    mtg_index* idx = nullptr;
    if (mtg_index_create_from_kmers(km.data(), ab.data(), km.size(), k, &idx)) { fprintf(stderr, "index: %s\n", mtg_last_error()); return 1; }
    std::vector<mtg_gap> gaps;
    std::vector<const char*> tseq(nloci), tname(nloci);
    std::vector<uint8_t> trc(nloci, 0);
    for (int rep = 0; rep < 20; rep++)
        for (int i = 0; i < nloci; i++) {
            tseq[i] = right[i].c_str();
            tname[i] = names[i].c_str();
            mtg_gap g;
            memset(&g, 0, sizeof g);
            g.source = left[i].c_str();
            g.target = right[i].c_str();
            g.n_targets = 1;
            g.target_seqs = &tseq[i];
            g.target_names = &tname[i];
            g.target_is_rc = &trc[i];
            gaps.push_back(g);
        }
    mtg_params p;
    mtg_default_params(&p);
    auto run = [&](std::vector<std::string>& out) -> bool {
        mtg_results* r = nullptr;
        if (mtg_fill_batch(idx, &p, gaps.data(), gaps.size(), &r)) { fprintf(stderr, "fill: %s\n", mtg_last_error()); return false; }
        out.clear();
        for (size_t i = 0; i < gaps.size(); i++) {
            const mtg_gap_result* g = mtg_results_get(r, i);
            std::string s = std::to_string(g->n_filled);
            for (int f = 0; f < g->n_filled; f++) { s += ':'; s += g->filled[f].seq; }
            out.push_back(s);
        }
        mtg_results_free(r);
        return true;
    };
    std::vector<std::string> want;
    if (!run(want)) return 1;
    size_t filled = 0;
    for (auto& s : want) filled += s[0] != '0';
    if (filled < gaps.size() / 2) { fprintf(stderr, "only %zu of %zu gaps filled\n", filled, gaps.size()); return 1; }
    int bad = 0;
    std::vector<std::thread> ts;
    for (int t = 0; t < 3; t++)
        ts.emplace_back([&] {
            std::vector<std::string> got;
            for (int it = 0; it < 6; it++)
                if (!run(got) || got != want) __atomic_add_fetch(&bad, 1, __ATOMIC_RELAXED);
        });
    for (auto& t : ts) t.join();
    mtg_index_free(idx);
    if (bad) { fprintf(stderr, "%d batches differ\n", bad); return 1; }
    printf("OK\n");
    return 0;
}
