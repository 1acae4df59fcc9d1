/* TEST-ONLY: mtg_dict_order.h (the order of a contig-mode seed's target dictionary from the hash codes of its keys) against the literal
 * construction of the reference (src/Filler.cpp:522-533: a fresh std::unordered_map<std::string, std::pair<std::string, bool>> filled in the
 * iteration order of the dictionary of all targets, the seed's own entries left out), for dictionaries of 0 .. 6000 random k-mer keys and
 * every kind of omission (none, one, a few, all).  Prints OK. */
#include "../../mindthegap_amd/csrc/mtg_dict_order.h"
#include <cstdio>
#include <random>
#include <string>

typedef std::unordered_map<std::string, std::pair<std::string, bool>> dict_t; /* src/Utils.hpp:43-44 */

int main()
{
    std::mt19937_64 rng(5);
    mtgcli::DictOrder sim;
    std::vector<uint32_t> got;
    long checked = 0;
    for (int round = 0; round < 400; round++) {
        const uint32_t n_contigs = round < 40 ? (uint32_t)round : (uint32_t)(rng() % (round % 7 == 0 ? 3000 : 300));
        const int k = 11 + (int)(rng() % 21);
        dict_t all;
        for (uint32_t c = 0; c < n_contigs; c++)
            for (int rc = 0; rc < 2; rc++) {
                std::string key(k, 'A');
                for (auto& ch : key) ch = "ACGT"[rng() & 3];
                all.insert({key, std::make_pair("contig" + std::to_string(c), rc != 0)}); /* (a duplicate key is dropped, as in the tool) */
            }
        std::vector<const std::string*> keys;
        std::vector<size_t> code;
        for (auto it = all.begin(); it != all.end(); ++it) { keys.push_back(&it->first); code.push_back(std::hash<std::string>()(it->first)); }
        const uint32_t n = (uint32_t)keys.size();
        for (int variant = 0; variant < 6; variant++) {
            std::vector<uint8_t> skip(n, 0);
            if (n) {
                if (variant == 1) skip[rng() % n] = 1;
                if (variant == 2) { skip[0] = 1; skip[n - 1] = 1; }
                if (variant == 3) for (int j = 0; j < 5; j++) skip[rng() % n] = 1;
                if (variant == 4) for (auto& s : skip) s = (rng() & 3) == 0;
                if (variant == 5) for (auto& s : skip) s = 1;
            }
            dict_t dict;
            uint32_t i = 0;
            for (auto it = all.begin(); it != all.end(); ++it, ++i)
                if (!skip[i]) dict.insert({it->first, it->second});
            sim.order(code.data(), n, variant ? skip.data() : nullptr, got);
            if (got.size() != dict.size()) { fprintf(stderr, "round %d variant %d: %zu entries, the dictionary has %zu\n", round, variant, got.size(), dict.size()); return 1; }
            size_t j = 0;
            for (auto it = dict.begin(); it != dict.end(); ++it, ++j)
                if (*keys[got[j]] != it->first) { fprintf(stderr, "round %d variant %d (n = %u): entry %zu differs\n", round, variant, n, j); return 1; }
            checked += (long)dict.size();
        }
    }
    printf("OK %ld entries\n", checked);
    return 0;
}
