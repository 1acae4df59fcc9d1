/*
 * mtg_paths.h -- the contig-graph walk of a multi-contig gap, on the device, one wave per gap:
 *   - edges of the contig graph: a -> b iff suffix_{k-1}(a) == prefix_{k-1}(b) on the same strand
 *     (IGraphOutput::construct_graph / print_edges, /root/reference/src/IGraphOutput.cpp:97-133,144-179; only the FF edges survive
 *     GraphAnalysis's parser, src/GraphAnalysis.cpp:98-105)
 *   - GraphAnalysis::find_all_paths_rev (src/GraphAnalysis.cpp:205-237,244-326): for every terminal node in ascending order a DFS over
 *     the in-edges (ascending), prepending nodes, never reusing a node, dropping a path that touches another terminal node, accepting it
 *     at node 0; the search of a terminal is abandoned level by level once a level holds >= 20 paths, or after 10^7 calls.
 * The lanes build the graph together; the DFS itself is a serial recursion, unrolled here into an explicit stack run by lane 0.
 * Output (uint32 words, PATHS_WORDS per gap): [0] status (0 = complete, 1 = did not fit / too many contigs: the host enumerates),
 * [1] number of paths, then per path: target index, length, the nodes from node 0 to the terminal node.
 * Compiled for gfx950 and, TEST-ONLY, for tests/emu (one lane).
 */
#ifndef MTG_PATHS_H
#define MTG_PATHS_H
#include "mtg_post.h"

namespace mtg {

enum { PATHS_WORDS = 4096, PATHS_MAXN = 128, PATHS_MAX_BREADTH = 20 /* src/GraphAnalysis.hpp:43 */ };

/* working set of one gap (LDS on the device) */
struct PathsWork {
    uint64_t pre[PATHS_MAXN], suf[PATHS_MAXN];
    uint64_t in_lo[PATHS_MAXN], in_hi[PATHS_MAXN]; /* in-edges of node j as a bit set over the nodes */
    uint8_t st_node[PATHS_MAXN + 2];
    uint64_t st_lo[PATHS_MAXN + 2], st_hi[PATHS_MAXN + 2]; /* in-edges of the frame still to visit */
    uint32_t st_cnt[PATHS_MAXN + 2];                      /* paths found below the frame so far */
};

MTG_DEV int paths_lowest(uint64_t lo, uint64_t hi)
{
#ifdef MTG_EMU
    return lo ? __builtin_ctzll(lo) : 64 + __builtin_ctzll(hi);
#else
    return lo ? (int)__ffsll((unsigned long long)lo) - 1 : 64 + (int)__ffsll((unsigned long long)hi) - 1;
#endif
}

MTG_DEV void paths_gap(const FillCfg& cfg, const GapScratch& S, const GapOut& o, int k, PathsWork& W, uint32_t* out)
{
    const uint32_t lane = MTG_LANE();
    const uint32_t n = o.n_contigs;
    const uint64_t* words = s_words(cfg, S);
    const SP<uint32_t> cstart = s_cstart(cfg, S);
    const SP<uint32_t> clen = s_clen(cfg, S);
    const SP<uint32_t> tpos = s_tpos(cfg, S);
    const SP<uint32_t> ttgt = s_ttgt(cfg, S);
    if (n == 0 || n > PATHS_MAXN) {
        if (lane == 0) { out[0] = 1; out[1] = 0; }
        return;
    }
    const uint64_t mk1 = kmask(k - 1);
    for (uint32_t c = lane; c < n; c += MTG_NLANES) {
        const uint64_t* w = words + cstart[c];
        W.pre[c] = le_kmer(w, 0, mk1);
        W.suf[c] = le_kmer(w, clen[c] - (uint32_t)(k - 1), mk1);
    }
    wave_sync();
    for (uint32_t j = lane; j < n; j += MTG_NLANES) {
        uint64_t lo = 0, hi = 0;
        const uint64_t pj = W.pre[j];
        for (uint32_t i = 0; i < n; i++) {
            if (W.suf[i] != pj) continue;
            if (i == j && clen[i] == (uint32_t)(k - 1)) continue; /* src/IGraphOutput.cpp:160 */
            if (i < 64) lo |= 1ull << i; else hi |= 1ull << (i - 64);
        }
        W.in_lo[j] = lo;
        W.in_hi[j] = hi;
    }
    wave_sync();
    if (lane != 0) return;

    uint64_t term_lo = 0, term_hi = 0;
    int first_term = -1;
    for (uint32_t c = 0; c < n; c++)
        if (tpos[c] != 0xFFFFFFFFu) {
            if (first_term < 0) first_term = (int)c;
            if (c < 64) term_lo |= 1ull << c; else term_hi |= 1ull << (c - 64);
        }
    uint32_t nout = 2, npaths = 0;
    bool overflow = false;
    if (first_term == 0) { /* src/GraphAnalysis.cpp:222-226: the terminal node is the start node */
        out[nout++] = ttgt[0];
        out[nout++] = 1;
        out[nout++] = 0;
        npaths = 1;
    } else {
        for (uint32_t t = 0; t < n && !overflow; t++) {
            if (!((t < 64 ? term_lo >> t : term_hi >> (t - 64)) & 1ull)) continue;
            uint32_t nb_calls = 0;
            bool success = true;
            uint64_t on_lo = 0, on_hi = 0; /* nodes of the current path */
            int d = 0;
            W.st_node[0] = (uint8_t)t;
            if (t < 64) on_lo |= 1ull << t; else on_hi |= 1ull << (t - 64);
            enum { ENTER, NEXT_EDGE, RETURN } state = ENTER;
            uint32_t ret = 0;
            for (;;) {
                if (state == ENTER) {
                    const uint32_t v = W.st_node[d];
                    if (nb_calls++ > 10000000u) { success = false; ret = 0; state = RETURN; }
                    else if (v != t && ((v < 64 ? term_lo >> v : term_hi >> (v - 64)) & 1ull)) { ret = 0; state = RETURN; } /* touches another terminal node */
                    else if (v == 0) {
                        /* the path: nodes of the stack from the top (node 0) down to the terminal node */
                        if (nout + 2 + (uint32_t)d + 1 > PATHS_WORDS) { overflow = true; break; }
                        out[nout++] = ttgt[t];
                        out[nout++] = (uint32_t)d + 1;
                        for (int q = d; q >= 0; q--) out[nout++] = W.st_node[q];
                        npaths++;
                        ret = 1;
                        state = RETURN;
                    } else {
                        W.st_lo[d] = W.in_lo[v];
                        W.st_hi[d] = W.in_hi[v];
                        W.st_cnt[d] = 0;
                        state = NEXT_EDGE;
                    }
                } else if (state == NEXT_EDGE) {
                    if (!(W.st_lo[d] | W.st_hi[d])) { ret = W.st_cnt[d]; state = RETURN; continue; }
                    const int nx = paths_lowest(W.st_lo[d], W.st_hi[d]);
                    if (nx < 64) W.st_lo[d] &= ~(1ull << nx); else W.st_hi[d] &= ~(1ull << (nx - 64));
                    const bool on_path = ((nx < 64 ? on_lo >> nx : on_hi >> (nx - 64)) & 1ull) != 0;
                    if (!on_path) {
                        d++;
                        W.st_node[d] = (uint8_t)nx;
                        if (nx < 64) on_lo |= 1ull << nx; else on_hi |= 1ull << (nx - 64);
                        state = ENTER;
                        continue;
                    }
                    if (!success) { ret = W.st_cnt[d]; state = RETURN; }
                } else { /* RETURN: hand `ret` paths to the caller's frame */
                    if (d == 0) break;
                    const uint32_t v = W.st_node[d];
                    if (v < 64) on_lo &= ~(1ull << v); else on_hi &= ~(1ull << (v - 64));
                    d--;
                    W.st_cnt[d] += ret;
                    if (W.st_cnt[d] >= PATHS_MAX_BREADTH) success = false;
                    if (!success) { ret = W.st_cnt[d]; state = RETURN; } else state = NEXT_EDGE;
                }
            }
        }
    }
    out[0] = overflow ? 1u : 0u;
    out[1] = overflow ? 0u : npaths;
}

} // namespace mtg
#endif
