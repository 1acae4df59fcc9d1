/* The device's launch sequence for the traversal of one gap, on one lane: the walk kernel parks the gap at the first branching node that is
 * not the strict SNP pattern; for MTG_ROUNDS rounds (default 2 here, so that both the rounds and the finishing kernel are exercised) the
 * bubble kernels answer the node (the LDS form first, the one-lane form when it does not fit) and the walk kernel resumes; k_finish takes
 * what is still parked.  MTG_EMU_CLASSIC=1: the one-lane reference form of the whole walk.  MTG_EMU_PARK_SNP=1: the walks of the rounds park at
 * SNP bubbles too and the bubble kernels answer them with the fast path (what a launch with many bubbles does).  MTG_LIGHT_WALK=1: the first
 * walk is the light kernel's (WALK_SIMPLE). */
#pragma once
#include <cstdlib>
namespace mtg {
inline uint32_t emu_walk(const Index& ix, const FillCfg& cfg, GapScratch& S, uint64_t src_f, const SwfPattern& R, GapOut& out)
{
    static const bool classic = getenv("MTG_EMU_CLASSIC") != nullptr;
    static const int rounds = getenv("MTG_ROUNDS") ? atoi(getenv("MTG_ROUNDS")) : 2;
    static const bool coop_off = getenv("MTG_EMU_BUBBLE_CLASSIC") != nullptr; /* every bubble of the rounds through k_bubble_classic */
    if (classic) { stage_a_gap(ix, cfg, S, src_f, R, out); return 0; }
    static const bool park_snp = getenv("MTG_EMU_PARK_SNP") != nullptr; /* the walk kernel parks at SNP bubbles too (a launch that serves bubbles in rounds) */
    const int snp0 = S.snp_fast;
    if (park_snp && snp0) S.snp_fast = 2;
    static thread_local BubbleLds lds;
    static thread_local BubbleLdsBig lds_big;
    static thread_local uint32_t walk_park[WALK_PARK_WORDS * 64]; /* the walk kernel's LDS words (the walk's own state while the fork forms run) */
    /* MTG_LIGHT_WALK=1 (read at every gap: tests switch it): the launch's first walk by the light kernel's form -- simple paths only, the gap parks at
     * its first branching node whatever its shape -- and the rounds / the finishing form take it from there, as on the device */
    const char* lw = getenv("MTG_LIGHT_WALK");
    if (lw && lw[0] == '1') stage_a_walk<WALK_SIMPLE, 1>(ix, cfg, S, src_f, R, out, nullptr);
    else stage_a_walk<WALK_PARK, 1>(ix, cfg, S, src_f, R, out, nullptr, false, nullptr, walk_park);
    if (out.status != GAP_PARKED) { S.snp_fast = snp0; return 0; }
    for (int r = 0; r < rounds && out.status == GAP_PARKED; r++) {
        if (coop_off || !bubble_coop<1>(ix, cfg, S, lds)) bubble_classic(ix, cfg, S);
        stage_a_walk<WALK_PARK, 1>(ix, cfg, S, 0, R, out, nullptr, true, nullptr, walk_park);
    }
    S.snp_fast = snp0;
    if (out.status == GAP_PARKED) stage_a_walk<WALK_FINISH, 1>(ix, cfg, S, 0, R, out, &lds_big);
    return 1;
}
}
