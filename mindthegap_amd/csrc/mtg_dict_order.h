/* mtg_dict_order.h -- host side of contig mode: the ORDER of a seed's target dictionary without building the dictionary.
 *
 * The reference (src/Filler.cpp:522-533, contigFunctor) gives every seed a fresh bkpt_dict_t (std::unordered_map<std::string, ...>,
 * src/Utils.hpp:43-44) and inserts, in the iteration order of the dictionary of ALL targets, every target but the seed's own contig; the iteration
 * order of that fresh map decides the order of a seed's solutions in every output file.  Built literally -- a node, a key string and a name string
 * per (seed, target) pair -- this is four allocations and a string hash per pair: at 10 000 contigs 4e8 pairs, 45 of the job's 47 seconds (round 6).
 *
 * The order depends on the keys only through their hash codes and on the container only through the sequence of insertions: a map of the
 * targets' NUMBERS whose hash function returns the strings' hash codes goes through the same _Hashtable code (same bucket counts from the
 * same rehash policy, same insertion at the beginning of a bucket, same relinking on a rehash) and iterates in the same order.  Its nodes come
 * from a block that is reused from seed to seed (std::pmr::monotonic_buffer_resource), the hash codes are computed once per job.
 * tests/emu/dict_order.cpp checks the order against the literal construction for thousands of random dictionaries. */
#ifndef MTG_DICT_ORDER_H
#define MTG_DICT_ORDER_H
#include <cstddef>
#include <cstdint>
#include <memory_resource>
#include <unordered_map>
#include <vector>

namespace mtgcli {

class DictOrder {
    struct CodeOf {
        static const size_t*& table() { static thread_local const size_t* t = nullptr; return t; }
        size_t operator()(uint32_t i) const noexcept { return table()[i]; }
    };
    std::vector<char> block_;

public:
    /* out = the numbers i in [0, n) with skip[i] == 0, in the order in which a default-constructed std::unordered_map<std::string, V> iterates after
     * keys with the hash codes code[i] were inserted for i = 0, 1, ... (those with skip[i] != 0 left out).  skip may be null. */
    void order(const size_t* code, uint32_t n, const uint8_t* skip, std::vector<uint32_t>& out)
    {
        const size_t want = (size_t)n * 64 + (1u << 16); /* nodes of 16 bytes, bucket arrays of every size the table goes through (about 4 n pointers in all) */
        if (block_.size() < want) block_.resize(want);
        out.clear();
        out.reserve(n);
        CodeOf::table() = code;
        std::pmr::monotonic_buffer_resource pool(block_.data(), block_.size());
        {
            std::pmr::unordered_map<uint32_t, char, CodeOf> m(&pool); /* the allocator-only constructor: the state of `bkpt_dict_t dict;` */
            for (uint32_t i = 0; i < n; i++)
                if (!skip || !skip[i]) m.emplace(i, 0);
            for (auto it = m.begin(); it != m.end(); ++it) out.push_back(it->first);
        }
        CodeOf::table() = nullptr;
    }
};

} // namespace mtgcli
#endif
