/*
 * mtg_copy.h -- the long runs of a gap's contigs, copied out of the unitig store by a whole wave.
 *
 * The traversal (mtg_traverse.h: run_long_step) follows a stored unitig without copying it: it fills the word it was writing, leaves one
 * CopyCmd for the whole words in between and continues behind them.  copy_gap executes a gap's commands: lane t produces the t-th of the
 * gap's deferred words from two neighbouring words of the store (funnel shift; reversed and complemented when the walk ran against the
 * stored orientation), so that source and destination are both read and written in consecutive 8-byte pieces by consecutive lanes.
 * What the reference does at this point is append one nucleotide per graph step to a std::string (gatb Traversal::traverse [MEM]);
 * the result -- the contigs of src/Filler.cpp:884 -- is the same, 2-bit packed.
 */
#ifndef MTG_COPY_H
#define MTG_COPY_H
#include "mtg_post.h"

namespace mtg {

/* contig nucleotide x of a command's stretch (x in arena coordinates, 32 * dst - lead <= x < 32 * (dst + nwords)) <-> nucleotide of the store */
MTG_DEV uint64_t cmd_store_nt(const CopyCmd& cm, int64_t x)
{
    const int64_t rel = x - (int64_t)(32ull * cm.dst);
    return (cm.src & 1ull) ? (uint64_t)((int64_t)(cm.src >> 1) - rel) : (uint64_t)((int64_t)(cm.src >> 1) + rel);
}

/* ---- the lean form: where is the target?  One look-up (the junction behind the target's k-mer) instead of copying the contig -- on the
 * bench set 2 500 nucleotides past the target -- and searching it.  An exact occurrence at a known place is what
 * find_nodes_containing_multiple_R reports (src/Filler.cpp:1341-1351: the first exact match ends the scan, and a contig never holds a
 * node twice), so k_post takes the position as it is; the emulation build runs the search as well and compares.
 * target: the gap's single target k-mer (forward value) when the lean form may be used (one usable target, a source of exactly k
 * nucleotides, records wanted), ~0 otherwise.  Decided by ONE lane (the one that finished the walk: mtg_gpu_fill.hip, lean_and_list); returns whether the gap's commands still
 * have to be executed (copy_cmds). */
MTG_DEV bool lean_decide(const Index& ix, const FillCfg& cfg, const GapScratch& S, const GapOut& o, uint64_t target)
{
    const UStore& us = ix.us;
    const SP<LeanRec> lr = s_lean(cfg, S);
    if (o.status != GAP_OK) { LeanRec r; r.valid = 0; r.pos0 = 0; r.cmd = 0; r.pad_ = 0; lr[0] = r; return false; } /* the record is read for every gap (k_lean's list for k_post) */
    const SP<CopyCmd> cmds = s_cmd(cfg, S);
    const int k = ix.k;
    bool lean = false;
    uint32_t pos0 = 0, ci = 0;
    if (target != ~0ull && o.n_contigs == 1 && o.n_cmds != 0 && us.nwords != 0) {
        uint32_t l_ = 0;
        const Adj e = adj_right_t(ix.adj, make_kmer(target, k), kmask(k - 1), l_);
        if (e.up && popc4(e.out) == 1 && popc4(e.in) == 1) {
            const bool fwd = !up_bwd(e.up); /* the target as given appears as it is stored */
            const uint64_t q = (up_hdr(e.up) + 1) * 32 + (fwd ? up_off(e.up) - 1u : up_off(e.up)); /* first nucleotide of its k-mer in the store */
            /* the junction behind a k-mer that is NOT solid can be one of the graph all the same (adj_right_t answers for the junction): the
             * k-mer the store holds there must be the target itself, or the counting search decides (it may still match with mismatches).
             * Round 4: found by a site whose anchor differs from the graph in its first nucleotide -- the record said 0 errors for 1. */
            const Kmer tk = make_kmer(target, k);
            const uint64_t cmpl = 0xAAAAAAAAAAAAAAAAULL & kmask(k);
            const bool is_target = us_kmer_le(us.words, q, k) == (fwd ? (tk.r ^ cmpl) : (tk.f ^ cmpl));
            const uint32_t clen0 = s_clen(cfg, S)[0];
            const int64_t a0 = (int64_t)(32ull * s_cstart(cfg, S)[0]);
            for (uint32_t c = 0; c < o.n_cmds && !lean && is_target; c++) {
                const CopyCmd cm = cmds[c];
                const bool bwd = (cm.src & 1ull) != 0;
                if (bwd == fwd) continue; /* the walk must run the way the target reads */
                const int64_t p = (int64_t)(cm.src >> 1), lo = (int64_t)(32ull * cm.dst) - (int64_t)cm.lead, hi = (int64_t)(32ull * ((uint64_t)cm.dst + cm.nwords));
                /* arena nucleotide of the k-mer's first nucleotide in walking order */
                const int64_t a = bwd ? (int64_t)(32ull * cm.dst) + (p - (int64_t)(q + (uint64_t)k - 1)) : (int64_t)(32ull * cm.dst) + ((int64_t)q - p);
                if (a < lo || a + k > hi || a0 < lo) continue;
                const int64_t ps = a - a0;
                if (ps <= (int64_t)k || ps + k > (int64_t)clen0) continue; /* an empty fill, or not on this contig: the general code decides */
                lean = true; pos0 = (uint32_t)ps; ci = c;
            }
        }
    }
    { LeanRec r; r.valid = lean ? 1u : 0u; r.pos0 = pos0; r.cmd = ci; r.pad_ = 0; lr[0] = r; }
#ifdef MTG_XCHECK
    return o.n_cmds != 0; /* the emulation build copies all the same: its cross-checks read the contig */
#else
    return !lean && o.n_cmds != 0; /* a lean gap: nothing of the contig is read from the arena */
#endif
}
/* the gap's commands, by a whole wave */
MTG_DEV void copy_cmds(const Index& ix, const FillCfg& cfg, const GapScratch& S, const GapOut& o)
{
    const UStore& us = ix.us;
    if (o.status != GAP_OK || o.n_cmds == 0) return;
    const SP<CopyCmd> cmds = s_cmd(cfg, S);
    uint64_t* words = s_words(cfg, S);
    /* the lane's words t, t + NLANES, ... in the concatenation of the commands: (c, base) follows t */
    uint32_t c = 0, base = 0;
    CopyCmd cm = cmds[0];
    for (uint32_t t = MTG_LANE(); t < o.copy_words; t += MTG_NLANES) {
        while (t >= base + cm.nwords) { base += cm.nwords; cm = cmds[++c]; }
        const uint32_t w = t - base;
        const bool bwd = (cm.src & 1ull) != 0;
        const uint64_t p = cm.src >> 1;
        words[cm.dst + w] = us_peek64(us.words, bwd ? p - 32ull * w : p + 32ull * w, 32u, bwd);
    }
}
/* both, one gap at a time (the emulation) */
MTG_DEV void copy_gap(const Index& ix, const FillCfg& cfg, const GapScratch& S, const GapOut& o, uint64_t target)
{
    if (lean_decide(ix, cfg, S, o, target)) copy_cmds(ix, cfg, S, o);
}

} // namespace mtg
#endif
