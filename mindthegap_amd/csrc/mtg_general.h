/*
 * mtg_general.h -- the multi-contig gap on the device, one wave per gap (round 5; the host did this for every such gap until round 4):
 *   - the paths k_paths enumerated, as the reference keeps them: a std::set of node vectors (src/Filler.cpp:924-932): sorted, unique;
 *   - GraphAnalysis::paths_to_sequences (/root/reference/src/GraphAnalysis.cpp:331-460): the contigs of a path joined with k nucleotides of the
 *     first node and k - 1 of every other dropped, the last node cut at its anchor position, the back-trim when the anchor lies inside the
 *     overlap (:404-407, with its size_t wrap-around), empty results skipped (:449);
 *   - remove_almost_identical_solutions(.., 90) (src/Utils.cpp:208-238) with needleman_wunsch's match count (:87-189) by the wave
 *     (nw_wave: the sweep of k_nw, here on 2-bit sequences);
 *   - coverage of source + sequence (src/Filler.cpp:959-988): abundance of every k-mer, float mean, exact median; compute_qual
 *     (src/Utils.hpp:85-103); the reverse complement of a reverse attempt (:78-83); the solutions' ASCII.
 * A dictionary with several targets (contig mode): the reference groups the paths by target NAME and works through the groups in libstdc++'s hash
 * order of those names.  The device groups by target (in the order of first appearance in the sorted paths = the order the reference
 * inserts the names), answers every group, and says which target each group belongs to (a header record before the group's solutions); the host,
 * which has the names, puts the groups in the map's order (mtg_host.cpp: run_general).
 * What stays on the host: gaps k_paths could not enumerate, a source that is not k clean nucleotides, a k-mer of abundance 0 (the host
 * prints the reference's warning), and whatever does not fit the work areas: such a gap says so (GenGap::status) and the host's path, which
 * is also the specification the TEST-ONLY emulation checks every device answer against, takes it.
 * Compiled for gfx950 and, TEST-ONLY, for tests/emu (one lane).
 */
#ifndef MTG_GENERAL_H
#define MTG_GENERAL_H
#include "mtg_paths.h"
#include "mtg_emit.h"

namespace mtg {

/* a cell of an alignment's boundary column: score, matches */
struct NwCell { int s, m; };
enum { GEN_MAX_CAND = 32, GEN_SEQ_WORDS = 512 /* 16 384 nucleotides per candidate */, GEN_OK = 0, GEN_HOST = 1 };

/* one solution of a gap (what the host turns into its Solution / mtg_filled) */
struct GenSol {
    uint64_t seq_off; /* into the ASCII arena; NUL-terminated */
    uint32_t seq_len;
    uint32_t target;
    int32_t nb_errors, qual, count, rank;
    float avg, median;
};
struct GenGap {
    uint32_t status;   /* GEN_OK: sols[first_sol .. first_sol + n_sols) are the gap's records; GEN_HOST: the host's path takes the gap */
    uint32_t n_sols, first_sol;
    uint32_t nb_total_filled; /* candidates before the de-duplication (info.txt) */
    uint32_t n_groups; /* 1: the records are the solutions.  > 1 (several targets reached): every group is a header record (rank 0, target = the group's
                          target, count = its solutions) followed by its solutions, groups in the order the reference inserts their names */
    uint32_t pad_;
};
/* cursors of a launch (the host reads them back) */
struct GenCtl {
    unsigned long long n_sols, ascii_bytes, tmp_words, bnd_cells;
};
struct GenDev {
    GenGap* gaps;   /* one per general gap of the launch, in the order of its list */
    GenSol* sols;
    char* ascii;
    uint64_t* tmp;  /* packed candidates */
    NwCell* bnd;    /* boundary columns of the alignments */
    uint64_t cap_sols, cap_ascii, cap_tmp, cap_bnd;
    GenCtl* ctl;
};
/* work areas of one gap (LDS on the device) */
struct GenWork {
    uint64_t seq[GEN_SEQ_WORDS + 2];      /* the candidate under construction; source + sequence for the coverage pass */
    uint32_t hist[256];
    uint32_t p_off[GEN_MAX_CAND];         /* word offset of path i in the paths block (its target, length, nodes) */
    uint64_t c_off[GEN_MAX_CAND];         /* candidates: word offset in GenDev::tmp */
    uint32_t c_len[GEN_MAX_CAND];
    int32_t c_err[GEN_MAX_CAND];
    uint32_t c_tgt[GEN_MAX_CAND];
    uint32_t f_src[GEN_MAX_CAND];         /* kept solutions: whose sequence, with whose errors, for which target */
    int32_t f_err[GEN_MAX_CAND];
    uint32_t f_tgt[GEN_MAX_CAND];
    uint32_t p_grp[GEN_MAX_CAND];         /* group of sorted path i */
    uint32_t g_tgt[GEN_MAX_CAND];         /* groups: target, first candidate (one more entry: the end), kept solutions */
    uint32_t g_c0[GEN_MAX_CAND + 1];
    uint32_t g_f0[GEN_MAX_CAND + 1];
    uint32_t scal[8];                     /* lane 0's verdicts, read by all */
};

MTG_DEV unsigned long long gen_reserve(unsigned long long* cur, unsigned long long n)
{
#ifdef MTG_EMU
    return __sync_fetch_and_add(cur, n);
#else
    return atomicAdd(cur, n);
#endif
}
MTG_DEV uint32_t gen_nt(const uint64_t* w, uint32_t i) { return (uint32_t)(w[i >> 5] >> (2u * (i & 31u))) & 3u; }

/* std::vector<int> operator< on two paths of the block */
MTG_DEV int gen_path_cmp(const uint32_t* paths, uint32_t a, uint32_t b)
{
    const uint32_t la = paths[a + 1], lb = paths[b + 1];
    const uint32_t n = la < lb ? la : lb;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t x = paths[a + 2 + i], y = paths[b + 2 + i];
        if (x != y) return x < y ? -1 : 1;
    }
    return la == lb ? 0 : (la < lb ? -1 : 1);
}

/* seq[dst_nt ..] |= n nucleotides of src from src_nt on; seq is zero beyond what has been appended.  By the lanes of the wave. */
MTG_DEV void gen_append(uint64_t* seq, uint32_t dst_nt, const uint64_t* src, uint32_t src_nt, uint32_t n)
{
    if (n == 0) return;
    const uint32_t w0 = dst_nt >> 5, w1 = (dst_nt + n - 1u) >> 5;
    for (uint32_t w = w0 + MTG_LANE(); w <= w1; w += MTG_NLANES) {
        /* destination nucleotides [lo, hi) of word w */
        const uint32_t lo = w == w0 ? dst_nt : 32u * w, hi = (w == w1) ? dst_nt + n : 32u * (w + 1u);
        const uint32_t cnt = hi - lo, s = src_nt + (lo - dst_nt);
        const uint32_t sh = 2u * (s & 31u);
        uint64_t v = src[s >> 5] >> sh;
        if (sh && (s & 31u) + cnt > 32u) v |= src[(s >> 5) + 1] << (64u - sh);
        if (cnt < 32u) v &= (1ull << (2u * cnt)) - 1ull;
        seq[w] |= v << (2u * (lo & 31u)); /* a word is the work of one lane per call; calls are separated by wave_sync */
    }
}

/* needleman_wunsch (src/Utils.cpp:87-189) match count of a (rows, na nucleotides) against b (columns), both 2-bit packed: the sweep of k_nw.
 * bnd: na + 1 cells.  Returns the count to every lane. */
MTG_DEV uint32_t nw_wave(const uint64_t* a, uint32_t na, const uint64_t* b, uint32_t nb, NwCell* bnd)
{
    if (na == 0 || nb == 0) return 0;
#ifdef MTG_EMU
    /* TEST-ONLY: the plain recurrence, ties broken diagonal, up, left (the traceback of :150-180) */
    int32_t* sp = new int32_t[4 * ((size_t)nb + 1)];
    int32_t *sc = sp + nb + 1, *mp = sc + nb + 1, *mc = mp + nb + 1;
    for (uint32_t j = 0; j <= nb; j++) { sp[j] = -5 * (int)j; mp[j] = 0; }
    for (uint32_t i = 1; i <= na; i++) {
        sc[0] = -5 * (int)i; mc[0] = 0;
        for (uint32_t j = 1; j <= nb; j++) {
            const bool eq = gen_nt(a, i - 1) == gen_nt(b, j - 1);
            const int diag = sp[j - 1] + (eq ? 10 : -5), del = sp[j] - 5, ins = sc[j - 1] - 5;
            const int best = diag > del ? (diag > ins ? diag : ins) : (del > ins ? del : ins);
            sc[j] = best;
            mc[j] = best == diag ? mp[j - 1] + (eq ? 1 : 0) : (best == del ? mp[j] : mc[j - 1]);
        }
        int32_t* t = sp; sp = sc; sc = t;
        t = mp; mp = mc; mc = t;
    }
    const uint32_t r = (uint32_t)mp[nb];
    int32_t* base = sp < sc ? sp : sc;
    base = base < mp ? base : mp;
    base = base < mc ? base : mc;
    delete[] base;
    (void)bnd;
    return r;
#else
    const uint32_t lane = MTG_LANE();
    for (uint32_t i = lane; i <= na; i += 64) bnd[i] = NwCell{-5 * (int)i, 0}; /* column 0 */
    __syncthreads();
    int result = 0;
    for (uint32_t j0 = 0; j0 < nb; j0 += 64) {
        const uint32_t j = j0 + lane + 1; /* 1-based column of this lane */
        const bool col_ok = j <= nb;
        const uint32_t bj = col_ok ? gen_nt(b, j - 1) : 256u;
        int s_up = -5 * (int)j, m_up = 0;
        int s_cur = 0, m_cur = 0;
        int s_diag = 0, m_diag = 0;
        uint32_t a_cur = 0;
        NwCell bchunk{0, 0};
        uint32_t achunk = 0;
        const uint32_t nsteps = na + 63;
        for (uint32_t t = 1; t <= nsteps; t++) {
            if (((t - 1) & 63u) == 0) {
                const uint32_t r = t + lane;
                bchunk = r <= na ? bnd[r] : NwCell{0, 0};
                achunk = r <= na ? gen_nt(a, r - 1) : 257u;
            }
            int s_left = __shfl_up(s_cur, 1, 64), m_left = __shfl_up(m_cur, 1, 64);
            uint32_t a_in = (uint32_t)__shfl_up((int)a_cur, 1, 64);
            const int src = (int)((t - 1) & 63u);
            const int bs = __shfl(bchunk.s, src, 64), bm = __shfl(bchunk.m, src, 64);
            const uint32_t ba = (uint32_t)__shfl((int)achunk, src, 64);
            if (lane == 0) { s_left = bs; m_left = bm; a_in = ba; }
            a_cur = a_in;
            const int i = (int)t - (int)lane;
            if (i == 1) { s_diag = -5 * ((int)j - 1); m_diag = 0; }
            if (col_ok && i >= 1 && i <= (int)na) {
                const bool eq = a_cur == bj;
                const int diag = s_diag + (eq ? 10 : -5), del = s_up - 5, ins = s_left - 5;
                const int best = max(max(diag, del), ins);
                const int m = best == diag ? m_diag + (eq ? 1 : 0) : (best == del ? m_up : m_left);
                s_cur = best; m_cur = m;
                s_up = best; m_up = m;
                if (lane == 63) bnd[i] = NwCell{best, m};
                if (i == (int)na && j == nb) result = m;
            }
            s_diag = s_left; m_diag = m_left;
        }
        __syncthreads();
    }
    return (uint32_t)__shfl(result, (int)((nb - 1) & 63u), 64);
#endif
}

/* are the packed sequences equal (same length) */
MTG_DEV bool gen_equal(const uint64_t* a, const uint64_t* b, uint32_t len)
{
    bool diff = false;
    const uint32_t nw = (len + 31u) / 32u;
    for (uint32_t w = MTG_LANE(); w < nw; w += MTG_NLANES) {
        uint64_t x = a[w] ^ b[w];
        if (w == nw - 1u && (len & 31u)) x &= (1ull << (2u * (len & 31u))) - 1ull;
        diff = diff || x != 0ull;
    }
    return !wave_any(diff);
}

/* One multi-contig gap.  paths: its PATHS_WORDS block from k_paths; n_targets: entries of its dictionary; src_f: its source k-mer (first
 * nucleotide in the highest field); flags: GAPF_*.  rank: its place in the launch's list (D.gaps[rank]).  Every lane returns the status. */
MTG_DEV uint32_t gen_gap(const Index& ix, const FillCfg& cfg, const GapScratch& S, const GapOut& o, int k, const uint32_t* paths, uint32_t n_targets, bool source_ok,
                         uint64_t src_f, uint32_t flags, const GenDev& D, uint32_t rank, GenWork& W)
{
    const uint32_t lane = MTG_LANE();
    const uint64_t* words = s_words(cfg, S);
    const SP<uint32_t> cstart = s_cstart(cfg, S);
    const SP<uint32_t> clen = s_clen(cfg, S);
    const SP<uint32_t> tpos = s_tpos(cfg, S);
    const SP<uint32_t> terr = s_terr(cfg, S);
    const SP<uint32_t> ttgt = s_ttgt(cfg, S);
    const uint32_t K = (uint32_t)k;
    GenGap out;
    out.status = GEN_HOST; out.n_sols = 0; out.first_sol = 0; out.nb_total_filled = 0; out.n_groups = 1; out.pad_ = 0;
    auto leave = [&](uint32_t status) -> uint32_t {
        out.status = status;
        if (lane == 0) D.gaps[rank] = out;
        return status;
    };
    if (n_targets == 0 || !source_ok || paths[0] != 0 || paths[1] > GEN_MAX_CAND || o.n_contigs == 0) return leave(GEN_HOST);
    const uint32_t np0 = paths[1];
    /* the set of paths: sorted by their node vectors, equal ones once (lane 0; a handful of short vectors) */
    if (lane == 0) {
        uint32_t q = 2, n = 0;
        for (uint32_t i = 0; i < np0; i++) {
            const uint32_t len = paths[q + 1];
            uint32_t at = n;
            bool dup = false;
            while (at > 0) {
                const int c = gen_path_cmp(paths, W.p_off[at - 1], q);
                if (c == 0) { dup = true; break; }
                if (c < 0) break;
                at--;
            }
            if (!dup) {
                for (uint32_t m = n; m > at; m--) W.p_off[m] = W.p_off[m - 1];
                W.p_off[at] = q;
                n++;
            }
            q += 2 + len;
        }
        W.scal[0] = n;
        /* the groups: paths of one target (paths_to_compare[name], src/Filler.cpp:924-936), numbered in the order of first appearance */
        uint32_t ng = 0;
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t t = paths[W.p_off[i]];
            uint32_t g = 0;
            while (g < ng && W.g_tgt[g] != t) g++;
            if (g == ng) W.g_tgt[ng++] = t;
            W.p_grp[i] = g;
        }
        W.scal[5] = ng;
    }
    wave_sync();
    const uint32_t np = W.scal[0], n_groups = W.scal[5];
    /* paths_to_sequences: every path's sequence, built in LDS, kept (2-bit) in the launch's arena of candidates */
    uint32_t nc = 0;
    for (uint32_t grp = 0; grp < n_groups; grp++) {
    if (lane == 0) W.g_c0[grp] = nc;
    for (uint32_t pi = 0; pi < np; pi++) {
        if (W.p_grp[pi] != grp) continue;
        const uint32_t q = W.p_off[pi], plen = paths[q + 1];
        for (uint32_t w = lane; w < GEN_SEQ_WORDS + 2; w += MTG_NLANES) W.seq[w] = 0;
        wave_sync();
        uint32_t L = 0;
        int32_t errs = 0;
        uint32_t tgt = 0;
        bool too_long = false;
        for (uint32_t ip = 0; ip < plen && !too_long; ip++) {
            const uint32_t node = paths[q + 2 + ip];
            const uint64_t* nw = words + cstart[node];
            const uint32_t nl = clen[node];
            const uint32_t from = ip != 0 ? K - 1u : K;
            if (ip + 1 == plen) {
                const uint32_t pos_anchor = tpos[node] == 0xFFFFFFFFu ? 0u : tpos[node]; /* a path ends in a terminal node; 0 is what the reference's loop leaves otherwise */
                if (tpos[node] != 0xFFFFFFFFu) { errs = (int32_t)terr[node]; tgt = ttgt[node]; }
                if (pos_anchor <= K - 1u) {
                    const uint32_t cut = (K - 1u) - pos_anchor;
                    if (L >= cut) L -= cut; /* sequence.substr(0, length - cut): a shorter sequence wraps around to "all of it" (:406) */
                } else {
                    const uint32_t n = pos_anchor - from;
                    if (L + n > 32u * GEN_SEQ_WORDS) { too_long = true; break; }
                    gen_append(W.seq, L, nw, from, n);
                    L += n;
                }
                break;
            }
            const uint32_t n = nl > from ? nl - from : 0u;
            if (L + n > 32u * GEN_SEQ_WORDS) { too_long = true; break; }
            gen_append(W.seq, L, nw, from, n);
            L += n;
            wave_sync();
        }
        wave_sync();
        if (too_long) return leave(GEN_HOST);
        if (L == 0) continue; /* :449 */
        const uint32_t nwrd = (L + 31u) / 32u;
        if (lane == 0) { const unsigned long long at = gen_reserve(&D.ctl->tmp_words, nwrd + 1u); W.scal[1] = (uint32_t)at; W.scal[2] = (uint32_t)(at >> 32); }
        wave_sync();
        const uint64_t at = (uint64_t)W.scal[1] | ((uint64_t)W.scal[2] << 32);
        if (at + nwrd + 1u > D.cap_tmp) return leave(GEN_HOST);
        for (uint32_t w = lane; w <= nwrd; w += MTG_NLANES) {
            uint64_t v = w < nwrd ? W.seq[w] : 0ull;
            if (w == nwrd - 1u && (L & 31u)) v &= (1ull << (2u * (L & 31u))) - 1ull; /* the back-trim leaves nucleotides behind the end */
            D.tmp[at + w] = v;
        }
        if (lane == 0) { W.c_off[nc] = at; W.c_len[nc] = L; W.c_err[nc] = errs; W.c_tgt[nc] = tgt; }
        nc++;
        wave_sync();
    }
    }
    if (lane == 0) W.g_c0[n_groups] = nc;
    wave_sync();
    out.nb_total_filled = nc;
    out.n_groups = n_groups;
    if (nc == 0) { out.n_sols = 0; out.n_groups = 1; return leave(GEN_OK); } /* nothing to order */
#ifndef MTG_EMU
    __threadfence_block(); /* the candidates are read back from the arena below */
#endif
    /* remove_almost_identical_solutions(.., 90), group by group: a group's kept solutions are f_src[g_f0[g] .. g_f0[g + 1]) */
    uint32_t nf = 0;
    for (uint32_t grp = 0; grp < n_groups; grp++) {
        const uint32_t c0 = W.g_c0[grp], c1 = W.g_c0[grp + 1], f0 = nf;
        if (lane == 0) W.g_f0[grp] = f0;
        if (c1 > c0) {
            if (lane == 0) { W.f_src[nf] = c0; W.f_err[nf] = W.c_err[c0]; W.f_tgt[nf] = W.c_tgt[c0]; }
            nf++;
            wave_sync();
        }
        for (uint32_t j = c0; j < c1 && c1 - c0 > 1u; j++) {
            bool similar = false;
            for (uint32_t f = f0; f < nf; f++) {
                const uint32_t i = W.f_src[f];
                bool same = i == j;
                if (!same) {
                    const uint32_t la = W.c_len[j], lb = W.c_len[i];
                    if (la == lb && gen_equal(D.tmp + W.c_off[j], D.tmp + W.c_off[i], la)) same = true;
                    else {
                        if (lane == 0) { const unsigned long long at = gen_reserve(&D.ctl->bnd_cells, (unsigned long long)la + 1u); W.scal[1] = (uint32_t)at; W.scal[2] = (uint32_t)(at >> 32); }
                        wave_sync();
                        const uint64_t at = (uint64_t)W.scal[1] | ((uint64_t)W.scal[2] << 32);
                        if (at + la + 1u > D.cap_bnd) return leave(GEN_HOST);
                        const uint32_t m = nw_wave(D.tmp + W.c_off[j], la, D.tmp + W.c_off[i], lb, D.bnd + at);
                        float identity = (float)m;
#ifdef MTG_EMU
                        identity /= (float)(la > lb ? la : lb);
#else
                        identity = __fdiv_rn(identity, (float)(la > lb ? la : lb));
#endif
                        same = identity * 100 >= 90;
                        wave_sync();
                    }
                }
                if (same) {
                    if (lane == 0 && W.c_err[j] < W.f_err[f]) { W.f_src[f] = j; W.f_err[f] = W.c_err[j]; } /* the kept one takes the better sequence, keeps its target */
                    similar = true;
                    wave_sync();
                    break;
                }
            }
            if (!similar) {
                if (lane == 0) { W.f_src[nf] = j; W.f_err[nf] = W.c_err[j]; W.f_tgt[nf] = W.c_tgt[j]; }
                nf++;
                wave_sync();
            }
        }
    }
    if (lane == 0) W.g_f0[n_groups] = nf;
    wave_sync();
    /* the kept solutions: coverage of source + sequence, quality, ASCII (reverse-complemented for a reverse attempt) */
    unsigned long long ascii_need = 0;
    for (uint32_t f = 0; f < nf; f++) ascii_need += (unsigned long long)W.c_len[W.f_src[f]] + 1u;
    const uint32_t n_rec = nf + (n_groups > 1u ? n_groups : 0u); /* a header record before every group when there are several */
    if (lane == 0) {
        const unsigned long long s0 = gen_reserve(&D.ctl->n_sols, n_rec), a0 = gen_reserve(&D.ctl->ascii_bytes, ascii_need);
        W.scal[1] = (uint32_t)s0; W.scal[2] = (uint32_t)(s0 >> 32); W.scal[3] = (uint32_t)a0; W.scal[4] = (uint32_t)(a0 >> 32);
    }
    wave_sync();
    const uint64_t s0 = (uint64_t)W.scal[1] | ((uint64_t)W.scal[2] << 32);
    uint64_t a0 = (uint64_t)W.scal[3] | ((uint64_t)W.scal[4] << 32);
    if (s0 + n_rec > D.cap_sols || a0 + ascii_need > D.cap_ascii) return leave(GEN_HOST);
    const uint64_t mk = kmask(k), cmpl = 0xAAAAAAAAAAAAAAAAULL & mk;
    const bool reverse = (flags & GAPF_REVERSE) != 0, repeated = (flags & GAPF_REPEATED) != 0;
    bool unknown = false;
    uint32_t rec = 0;
    for (uint32_t grp = 0; grp < n_groups; grp++) {
    const uint32_t gf0 = W.g_f0[grp], gn = W.g_f0[grp + 1] - gf0; /* the group's solutions */
    if (n_groups > 1u) {
        GenSol hd;
        hd.seq_off = 0; hd.seq_len = 0; hd.target = W.g_tgt[grp]; hd.nb_errors = 0; hd.qual = 0; hd.count = (int32_t)gn; hd.rank = 0; hd.avg = 0.f; hd.median = 0.f;
        if (lane == 0) D.sols[s0 + rec] = hd;
        rec++;
    }
    for (uint32_t f = gf0; f < gf0 + gn; f++) {
        const uint32_t c = W.f_src[f], L = W.c_len[c];
        const uint64_t* cw = D.tmp + W.c_off[c];
        /* source + sequence, little-endian, in LDS: L + 1 k-mers */
        for (uint32_t w = lane; w < GEN_SEQ_WORDS + 2; w += MTG_NLANES) W.seq[w] = 0;
        for (uint32_t w = lane; w < 256; w += MTG_NLANES) W.hist[w] = 0;
        wave_sync();
        if (lane == 0) W.seq[0] = rev_fields64(src_f) >> (64 - 2 * k); /* the source's first nucleotide in the lowest field */
        wave_sync();
        gen_append(W.seq, K, cw, 0, L); /* K + L <= 32 * GEN_SEQ_WORDS + 31: the two spare words */
        wave_sync();
        const uint32_t nk = L + 1u;
        uint32_t sum = 0, lines = 0;
        for (uint32_t p = lane; p < nk; p += MTG_NLANES) {
            Kmer x;
            x.r = le_kmer(W.seq, p, mk) ^ cmpl;
            x.f = revcomp(x.r, k);
            const uint32_t a = abundance(ix, x, lines);
            if (a == 0) unknown = true;
            sum += a;
            hist_add(W.hist, a > 255u ? 255u : a);
        }
        sum = wave_sum32(sum);
        wave_sync();
        uint32_t hi = 0, lo = 0;
        hist_median(W.hist, nk, hi, lo);
        GenSol s;
        s.seq_off = a0;
        s.seq_len = L;
        s.target = W.f_tgt[f];
        s.nb_errors = W.f_err[f];
        s.count = (int32_t)gn;
        s.rank = (int32_t)(f - gf0) + 1;
#ifdef MTG_EMU
        s.avg = (float)sum / (float)nk;
#else
        s.avg = __fdiv_rn((float)sum, (float)nk);
#endif
        s.median = (nk & 1u) ? (float)hi : 0.5f * (float)(hi + lo);
        int q = 50; /* compute_qual, src/Utils.hpp:85-103 */
        if (repeated) q = 25;
        if (gn > 1) q = 15;
        if (s.nb_errors == 1) q = 10;
        if (s.nb_errors == 2) q = 5;
        s.qual = q;
        if (lane == 0) D.sols[s0 + rec] = s;
        rec++;
        emit_ascii(cw, 0, L, reverse, D.ascii + a0);
        a0 += (uint64_t)L + 1u;
        wave_sync();
    }
    }
    if (wave_any(unknown)) return leave(GEN_HOST); /* "WARNING Unknown kmer" is the host's to print (src/Filler.cpp:980-982) */
    out.n_sols = n_rec;
    out.first_sol = (uint32_t)s0;
    return leave(GEN_OK);
}

} // namespace mtg
#endif
