/* TEST-ONLY: stress of the host worker pool and of the block patterns built on it (mtg_internal.h), meant to be run under
 * ThreadSanitizer: many short parallel regions in a row, work-exhaustion exit, prefix sums, the chained block ends of the arena layout */
#define MTG_EMU 1
#include "../../mindthegap_amd/csrc/mtg_internal.h"
#include <cstdio>
#include <numeric>

namespace mtgi { void set_error(const char*, ...) {} }

int main()
{
    using namespace mtgi;
    uint64_t bad = 0;
    for (int round = 0; round < 3000; round++) {
        const size_t n = (size_t)(round * 37 % 5000) + 1;
        const int nth = round % 9; /* 0 = all */
        std::vector<uint32_t> v(n, 0);
        parallel_for(n, nth, [&](size_t i) { v[i] += (uint32_t)i + 1; }, (size_t)(round % 7) * 16 + 1);
        uint64_t s = 0;
        for (size_t i = 0; i < n; i++) s += v[i];
        if (s != (uint64_t)n * (n + 1) / 2) bad++;
        if (round % 5 == 0) {
            std::vector<uint64_t> out(n + 1);
            const uint64_t tot = parallel_prefix(n, nth, out.data(), [&](size_t i) { return (uint64_t)(i % 7); });
            uint64_t run = 0;
            for (size_t i = 0; i < n; i++) { if (out[i] != run) bad++; run += i % 7; }
            if (tot != run || out[n] != run) bad++;
        }
        if (round % 3 == 0) {
            /* blocks handed out in order publish where they end; each waits for its predecessor */
            const size_t B = 64, nb = (n + B - 1) / B;
            std::vector<std::atomic<int64_t>> ends(nb + 1);
            for (auto& e : ends) e.store(-1, std::memory_order_relaxed);
            ends[0].store(0, std::memory_order_release);
            std::vector<int64_t> begin_of(nb, -1);
            parallel_for(nb, nth, [&](size_t b) {
                int64_t need = 0;
                for (size_t i = b * B; i < std::min(n, (b + 1) * B); i++) need += (int64_t)(i % 5);
                int64_t begin;
                while ((begin = ends[b].load(std::memory_order_acquire)) < 0) Pool::cpu_relax();
                ends[b + 1].store(begin + need, std::memory_order_release);
                begin_of[b] = begin;
            }, 1);
            int64_t run = 0;
            for (size_t b = 0; b < nb; b++) {
                if (begin_of[b] != run) bad++;
                for (size_t i = b * B; i < std::min(n, (b + 1) * B); i++) run += (int64_t)(i % 5);
            }
        }
    }
    printf(bad ? "FAILED %llu\n" : "OK\n", (unsigned long long)bad);
    return bad ? 1 : 0;
}
