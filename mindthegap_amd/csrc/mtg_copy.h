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

MTG_DEV void copy_gap(const UStore& us, const FillCfg& cfg, const GapScratch& S, const GapOut& o)
{
    if (o.status != GAP_OK || o.n_cmds == 0) return;
    const CopyCmd* cmds = s_cmd(cfg, S);
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

} // namespace mtg
#endif
