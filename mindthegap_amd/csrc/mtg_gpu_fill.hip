/*
 * mtg_gpu_fill.hip -- the fill kernels and the launch sequence of a batch (gfx950 only).
 *
 *   k_stage_a                        : breadth-first contig construction of one gap per lane
 *                                      (IterativeExtensions::construct_linear_seqs, /root/reference/src/Filler.cpp:884)
 *   k_bubble* / k_finish*            : the branching nodes of parked gaps (MonumentTraversal::explore_branching): groups of lanes, LDS work areas
 *   (lean_and_list) / k_copy         : where the target lies in the run a walk took (by the walking lane); the long runs of the contigs out of the unitig store
 *   k_post / k_post_lean             : terminal-node search per contig + coverage of the single-contig solution
 *                                      (find_nodes_containing_multiple_R, src/Filler.cpp:1294-1378; coverage :959-988)
 *   k_scan1 / k_scan2 / k_emit*      : layout of a batch's results and the results themselves (records + ASCII)
 *   k_paths                          : contig graph + reverse path enumeration of multi-contig gaps, one wave per gap
 *                                      (find_all_paths_rev, src/GraphAnalysis.cpp:205-326)
 */
#include "mtg_gpu_common.h"

namespace mtgi {

/* Index and configuration of the traversal kernel live in constant memory: its code takes them by reference all over (Worker, the
 * bubble routines), and a by-value kernel argument whose address is taken is copied to private memory, which turned every field
 * access into a per-lane scratch load; a reference to a __constant__ object stays a scalar load. */
enum { TRAVERSAL_SETS = mtg_index::NWS }; /* one set of constants per workspace number: that many traversals share the device */
__constant__ Index c_ix[TRAVERSAL_SETS];
__constant__ FillCfg c_cfg[TRAVERSAL_SETS];

/* ---- the traversal: two kernels.
 *
 * k_stage_a, the walk kernel: one gap per lane, one wave per workgroup (waves retire independently).  A lane follows its simple paths
 * (whole unitigs at a time) and answers the strict SNP pattern itself; at any other branching node it PARKS its gap: the walk's state goes
 * to the gap's raw block (WalkSave) and the slot to the launch's work list -- one atomic per wave, the lanes' places from a ballot and a
 * prefix popcount.
 *
 * k_finish, the finishing kernel: a group of G lanes (a wave, or an aligned part of one) takes a parked gap off the work list and runs
 * the rest of its life: frontier expansion one lane per (node, nucleotide) with ballots, visited sets / frontlines / path enumeration /
 * consensuses in LDS (mtg_bubble.h), the walk between two branching nodes by all lanes of the group with the same values.
 */
/* Registers of the kernels that hold the general bubble code (k_bubble_classic, k_finish_lane) are capped for two waves per SIMD (left alone the
 * compiler takes 289, one wave per SIMD); the walk kernel without it needs fewer.  -DMTG_STAGE_A_WAVES=n / -DMTG_WALK_WAVES=n: experiments with another cap. */
#ifndef MTG_STAGE_A_WAVES
#define MTG_STAGE_A_WAVES 2
#endif
#ifndef MTG_WALK_WAVES
#define MTG_WALK_WAVES 2
#endif
/* device: the work lists of one launch.  count[i] = entries of list i; list i = cap slot numbers at lists + i * cap.  List 2r holds the gaps
 * parked by the r-th launch of the walk kernel (at a branching node), list 2r + 1 those of them whose bubble did not fit the LDS areas. */
enum { PARK_LISTS = 19 }; /* 0 .. 15: the rounds' lists of parked gaps; the last two: gaps for k_post's general form, gaps with copy commands to execute */
struct ParkCtl {
    uint32_t count[PARK_LISTS];
    uint32_t n_branching; /* gaps of the launch whose walk (first launch of the walk kernel) stood on a branching node with successors */
    uint32_t pad_[3];
#ifdef MTG_BUBBLE_TIMING /* diagnostics build: how long the lanes and the waves of the bubble kernels ran (bins of log2 of 10 ns ticks) */
    uint32_t hist_lane[32], hist_wave[32], hist_walk_lane[32], hist_walk_wave[32];
#endif
};
#ifdef MTG_BUBBLE_TIMING
__device__ __forceinline__ void timing_note(uint32_t* hl, uint32_t* hw, uint64_t t0)
{
    const uint64_t dl = wall_clock64() - t0;
    atomicAdd(&hl[63 - __clzll((long long)(dl | 1ull))], 1u);
    __builtin_amdgcn_wave_barrier();
    const unsigned long long act = __ballot(1);
    if ((int)(threadIdx.x & 63u) == __ffsll((long long)act) - 1) { const uint64_t dw = wall_clock64() - t0; atomicAdd(&hw[63 - __clzll((long long)(dw | 1ull))], 1u); }
}
#endif
__device__ __forceinline__ uint32_t* park_list(ParkCtl* p, uint32_t cap, uint32_t i) { return reinterpret_cast<uint32_t*>(p + 1) + (size_t)i * cap; }
/* the lanes of a wave that park their gap append it to a list: one atomic per wave, the places from a ballot and a prefix popcount */
__device__ __forceinline__ void park_append(ParkCtl* park, uint32_t cap, uint32_t list, bool parked, uint32_t slot)
{
    const unsigned long long pm = __ballot(parked);
    if (!pm) return;
    const int leader = __ffsll((long long)pm) - 1;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(&park->count[list], (uint32_t)__popcll(pm));
    base = (uint32_t)__shfl((int)base, leader, 64);
    if (parked) park_list(park, cap, list)[base + (uint32_t)__popcll(pm & ((1ull << lane) - 1ull))] = slot;
}
/* ---- what becomes of a walk's long runs (mtg_copy.h), decided by the lane that finished the walk (a kernel of its own, k_lean, until late round 5:
 * one more launch and another read of the record, the commands and the contig's start / length the lane has just written): is the target inside a run
 * the walk took (the lean form: nothing is copied, k_post_lean and k_emit_lean read the store)?  The gaps that do need their commands executed go on
 * COPY_LIST, every gap that is not lean on POST_LIST (ballot + prefix popcount, as for parking).  decides: the lane holds a finished gap. */
enum { COPY_LIST = PARK_LISTS - 1, POST_LIST = PARK_LISTS - 2 };
struct LeanIn {
    const uint64_t* tle;      /* encoded targets, their never-match masks */
    const uint64_t* tbad;
    const uint32_t* toff;     /* per gap: first target, number of targets */
    const uint32_t* tcnt;
    const uint8_t* fast_ok;   /* per gap: the source is exactly k clean nucleotides */
    uint32_t lean_allowed;
    uint32_t decide_parked;   /* diagnostics (DEBUG_SKIP_FINISH): nobody will finish the parked gaps, the first walk lists them as they are */
};
/* the gap's single target k-mer (forward value) when the lean form may be used -- one usable target and a source of exactly k nucleotides (what the
 * common-case result of k_post needs anyway) --, ~0 otherwise.  Read BEFORE the walk: two dependent rounds of input reads that would otherwise sit
 * behind it */
__device__ __forceinline__ uint64_t lean_target(const LeanIn& li, uint32_t g, int k)
{
    uint64_t target = ~0ull;
    if (li.lean_allowed && li.tcnt[g] == 1u && li.fast_ok[g] && li.tbad[li.toff[g]] == 0ull) target = rev_fields64(li.tle[li.toff[g]]) >> (64 - 2 * k);
    return target;
}
__device__ __forceinline__ void lean_and_list(const Index& ix, const FillCfg& cfg, const GapScratch& S, const GapOut& o, uint64_t target, uint32_t slot, bool decides, ParkCtl* park, uint32_t cap)
{
    bool need = false, general = false;
    if (decides) {
        need = lean_decide(ix, cfg, S, o, target);
        general = !s_lean(cfg, S)[0].valid;
    }
    park_append(park, cap, COPY_LIST, need, slot);
    park_append(park, cap, POST_LIST, general, slot); /* every gap that is not lean (a failed one too): k_post's general form */
}
/* one gap per lane.  in_list < 0: the gaps of the launch, from their source k-mers; otherwise the gaps of that work list, resumed (a bubble
 * kernel has answered the branching node they stand on).  out_list: where the gaps that park (again) go. */
template <int MODE>
__device__ __forceinline__ void stage_a_lane(uint8_t* zero, uint8_t* raw, uint8_t* ilv, uint8_t* head, const uint64_t* __restrict__ src, const uint64_t* __restrict__ rwords, const uint32_t* __restrict__ roff,
                                             const uint32_t* __restrict__ rlen, const uint64_t* __restrict__ r0, const uint32_t* __restrict__ ids, GapOut* out, uint32_t n, uint32_t cset,
                                             ParkCtl* park, uint32_t cap, int in_list, uint32_t out_list, const LeanIn& li, uint32_t* walk_park = nullptr)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t slot = t;
    if (in_list < 0) { if (t >= n) return; }
    else {
        if (t >= park->count[in_list]) return;
        slot = park_list(park, cap, (uint32_t)in_list)[t];
    }
    const Index& ix = c_ix[cset]; /* cset is a kernel argument: still scalar loads */
    const FillCfg& cfg = c_cfg[cset];
    const uint32_t g = ids ? ids[slot] : slot; /* gap id in the input arrays; scratch is indexed by slot */
    GapScratch S = carve(cfg, zero, raw, ilv, head, slot);
    S.snp_fast = 1; /* the walking lane answers the strict SNP pattern itself */
    SwfPattern R;
    R.words = rwords + roff[g];
    R.rlen = rlen[g];
    R.r0 = r0[g];
    GapOut o;
#ifdef MTG_BUBBLE_TIMING
    const uint64_t t0 = wall_clock64();
#endif
    uint32_t met = 0;
    const uint64_t target = lean_target(li, g, ix.k);
    stage_a_walk<MODE, 1>(ix, cfg, S, src[g], R, o, nullptr, in_list >= 0, &met, walk_park);
#ifdef MTG_BUBBLE_TIMING
    if (MODE == WALK_PARK && in_list >= 0) timing_note(park->hist_walk_lane, park->hist_walk_wave, t0);
#endif
    out[slot] = o;
    if (in_list < 0) { /* how much of the data branches: what chooses the next launch's walk kernel (one atomic per wave that met any) */
        const unsigned long long bm = __ballot(met != 0u);
        if (bm && (int)(threadIdx.x & 63u) == __ffsll((long long)bm) - 1) atomicAdd(&park->n_branching, (uint32_t)__popcll(bm));
    }
    if (MODE == WALK_PARK || MODE == WALK_SIMPLE) park_append(park, cap, out_list, o.status == GAP_PARKED, slot);
    lean_and_list(ix, cfg, S, o, target, slot, o.status != GAP_PARKED || li.decide_parked != 0u, park, cap);
}
/* the light walk kernel (WALK_SIMPLE): simple paths only, every branching node parks the gap.  No bubble code in its call graph */
__global__ void __launch_bounds__(64) k_walk(uint8_t* zero, uint8_t* raw, uint8_t* ilv, uint8_t* head, const uint64_t* __restrict__ src, const uint64_t* __restrict__ rwords,
                                             const uint32_t* __restrict__ roff, const uint32_t* __restrict__ rlen, const uint64_t* __restrict__ r0, const uint32_t* __restrict__ ids,
                                             GapOut* out, uint32_t n, uint32_t cset, ParkCtl* park, uint32_t cap, LeanIn li)
{
    stage_a_lane<WALK_SIMPLE>(zero, raw, ilv, head, src, rwords, roff, rlen, r0, ids, out, n, cset, park, cap, -1, 0u, li);
}
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MTG_WALK_WAVES))) k_stage_a(uint8_t* zero, uint8_t* raw, uint8_t* ilv, uint8_t* head, const uint64_t* __restrict__ src,
                                                const uint64_t* __restrict__ rwords, const uint32_t* __restrict__ roff,
                                                const uint32_t* __restrict__ rlen, const uint64_t* __restrict__ r0, const uint32_t* __restrict__ ids,
                                                GapOut* out, uint32_t n, uint32_t cset, ParkCtl* park, uint32_t cap, int in_list, uint32_t out_list, LeanIn li)
{
    __shared__ uint32_t s_walk_park[WALK_PARK_WORDS * 64]; /* one wave per workgroup: the walking lanes' own state while the fork forms run (mtg_traverse.h: WALK_PARK_WORDS) */
    stage_a_lane<WALK_PARK>(zero, raw, ilv, head, src, rwords, roff, rlen, r0, ids, out, n, cset, park, cap, in_list, out_list, li, s_walk_park);
}
/* ---- the rounds between two launches of the walk kernel: the branching nodes of the parked gaps, answered on their own, one lane per
 * bubble from HBM scratch -- all 64 lanes of a wave are in the bubble code at the same time.  (k_bubble<G>, a group of G lanes per bubble with
 * LDS work areas, was the alternative of round 3; it lost on the narrow bubbles of heterozygous data -- DESIGN.md section 4 -- and is gone.  The
 * group form lives on where few gaps are in the bubble code: k_finish<G>.) */
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MTG_STAGE_A_WAVES))) k_bubble_classic(uint8_t* zero, uint8_t* raw, uint8_t* ilv, uint8_t* head, uint32_t cset, ParkCtl* park, uint32_t cap, uint32_t in_list)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= park->count[in_list]) return;
    const Index& ix = c_ix[cset];
    const FillCfg& cfg = c_cfg[cset];
    const uint32_t slot = park_list(park, cap, in_list)[t];
    GapScratch S = carve(cfg, zero, raw, ilv, head, slot);
    S.snp_fast = 1; /* the strict SNP pattern is answered by the fast path here as in the walk */
#ifdef MTG_BUBBLE_TIMING
    const uint64_t t0 = wall_clock64();
#endif
    bubble_classic(ix, cfg, S);
#ifdef MTG_BUBBLE_TIMING
    timing_note(park->hist_lane, park->hist_wave, t0);
#endif
}
/* the gaps that are still parked after the rounds (all of them when there are no rounds), one group of G lanes each, to the end of their
 * walks: group i of the grid takes entry i of the list.  The grid is sized for the worst case (the host does not know the count when it
 * queues the kernel); a group without an entry leaves at once.  (A loop over tickets around the walk -- fewer, longer-lived groups -- hung on the
 * device in round 3.  Round 6 reproduced it in sixty lines, scripts/r6_ticket_loop.hip, profiles/r06_ticket_loop.txt: it is not the size of the
 * kernel.  With hipcc 7.2 (clang 22.0.0git roc-7.2.0) the shape "the group's first lane takes the ticket under `if (lane == 0)`, the others get it
 * by shuffle, the loop body holds ballots / shuffles" is compiled so that every lane but the first is marked done after the first pass through
 * the body and parks in a loop without exit at the kernel's end -- also with one group per wave, also with the body inlined, also at -O1; without
 * collectives in the body, or with EVERY lane taking a ticket (no branch around the atomic), the same loop runs and agrees with the straight-line
 * form entry by entry.  The product keeps the straight-line form.) */
#ifndef MTG_FINISH_WAVES
#define MTG_FINISH_WAVES 2
#endif
template <int G>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MTG_FINISH_WAVES))) k_finish(uint8_t* zero, uint8_t* raw, uint8_t* ilv, uint8_t* head, const uint64_t* __restrict__ rwords, const uint32_t* __restrict__ roff,
                                               const uint32_t* __restrict__ rlen, const uint64_t* __restrict__ r0, const uint32_t* __restrict__ ids, GapOut* out, uint32_t cset,
                                               ParkCtl* park, uint32_t cap, uint32_t in_list, LeanIn li)
{
    __shared__ BubbleLdsBig lds[64 / G];
    const uint32_t lane = threadIdx.x & 63u, gl = lane & (uint32_t)(G - 1);
    const uint32_t t = blockIdx.x * (64u / (uint32_t)G) + lane / (uint32_t)G;
    if (t >= park->count[in_list]) return;
#ifdef MTG_FINISH_ONE_LANE /* diagnostics: the group is its first lane alone (needs -DMTG_COOP_OFF) */
    if (gl != 0) return;
#endif
    const Index& ix = c_ix[cset];
    const FillCfg& cfg = c_cfg[cset];
    const uint32_t slot = park_list(park, cap, in_list)[t];
    const uint32_t g = ids ? ids[slot] : slot;
    GapScratch S = carve(cfg, zero, raw, ilv, head, slot);
    S.snp_fast = 1;
    SwfPattern R;
    R.words = rwords + roff[g];
    R.rlen = rlen[g];
    R.r0 = r0[g];
    GapOut o;
    const uint64_t target = lean_target(li, g, ix.k);
    stage_a_walk<WALK_FINISH, G>(ix, cfg, S, 0, R, o, &lds[lane / G]);
    if (gl == 0) out[slot] = o;
    lean_and_list(ix, cfg, S, o, target, slot, gl == 0, park, cap);
}

/* the same with one LANE per parked gap and the general code on HBM scratch (A/B hook, MTG_FINISH_G=1): the group form is faster even
 * for a handful of parked gaps */
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MTG_STAGE_A_WAVES))) k_finish_lane(uint8_t* zero, uint8_t* raw, uint8_t* ilv, uint8_t* head, const uint64_t* __restrict__ rwords,
                                               const uint32_t* __restrict__ roff, const uint32_t* __restrict__ rlen, const uint64_t* __restrict__ r0, const uint32_t* __restrict__ ids,
                                               GapOut* out, uint32_t cset, ParkCtl* park, uint32_t cap, uint32_t in_list, uint32_t first, LeanIn li)
{
    const uint32_t t = first + blockIdx.x * blockDim.x + threadIdx.x; /* entries below `first` belong to the groups of k_finish<G> */
    if (t >= park->count[in_list]) return;
    const Index& ix = c_ix[cset];
    const FillCfg& cfg = c_cfg[cset];
    const uint32_t slot = park_list(park, cap, in_list)[t];
    const uint32_t g = ids ? ids[slot] : slot;
    GapScratch S = carve(cfg, zero, raw, ilv, head, slot);
    S.snp_fast = 1;
    SwfPattern R;
    R.words = rwords + roff[g];
    R.rlen = rlen[g];
    R.r0 = r0[g];
    GapOut o;
    const uint64_t target = lean_target(li, g, ix.k);
    stage_a_walk<WALK_FINISH, 1>(ix, cfg, S, 0, R, o, nullptr);
    out[slot] = o;
    lean_and_list(ix, cfg, S, o, target, slot, true, park, cap);
}

/* the long runs the traversal left as commands (mtg_copy.h), for the gaps on COPY_LIST (lean_and_list above).  k_copy, one wave per listed gap, four per
 * workgroup: the grid covers the launch (the host does not know the count), a wave beyond the list leaves after one scalar read. */
__global__ void __launch_bounds__(256) k_copy(Index ix, FillCfg cfg, uint8_t* raw, uint8_t* head, const GapOut* __restrict__ outs, ParkCtl* park, uint32_t cap, uint32_t list)
{
    const uint32_t count = park->count[list];
    for (uint32_t t = blockIdx.x * 4u + (threadIdx.x >> 6); t < count; t += gridDim.x * 4u) { /* the grid covers the launch */
        const uint32_t slot = park_list(park, cap, list)[t];
        GapScratch S;
        S.z = nullptr;
        S.v = nullptr;
        S.lane = slot & 63u;
        S.r = raw + (uint64_t)slot * cfg.raw_stride;
        S.h = head + (uint64_t)(slot >> 6) * cfg.hd_stride;
        copy_cmds(ix, cfg, S, outs[slot]);
    }
}
/* mtg_fill_text: a batch whose strings are still text (mtg_marshal.h).  One gap per thread: source k-mer, packed pattern, its first k-mer,
 * whether the fast forms apply; one dictionary entry per thread: little-endian k-mer and never-match mask.  The reads are a few dozen bytes
 * per thread at unrelated places of the block: 100 000 gaps take some tens of microseconds, against 1-2 ms of two host threads. */
__global__ void k_marshal_text(const uint8_t* __restrict__ text, const uint64_t* __restrict__ soff, const uint32_t* __restrict__ slen, const uint64_t* __restrict__ poff,
                               const uint32_t* __restrict__ roff, uint32_t* rlen, uint64_t* src, uint64_t* r0, uint8_t* fast_ok, uint64_t* rw, uint32_t n, int k)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    uint64_t s, r;
    uint32_t rl;
    uint8_t fo;
    marshal_text_gap(text, soff[g], slen[g], poff[g], rlen[g], k, rw + roff[g], s, r, rl, fo);
    src[g] = s; r0[g] = r; rlen[g] = rl; fast_ok[g] = fo;
}
__global__ void k_marshal_targets(const uint8_t* __restrict__ text, const uint64_t* __restrict__ doff, const uint32_t* __restrict__ dlen, uint64_t* __restrict__ tle, uint64_t* __restrict__ tbad,
                                  uint32_t nt, int k)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    uint64_t le, bad;
    marshal_text_target(text, doff[t], dlen[t], k, le, bad);
    tle[t] = le; tbad[t] = bad;
}
/* the targets of a batch from text to (little-endian k-mer, never-match mask): one target per thread */
__global__ void k_encode_targets(const uint8_t* __restrict__ traw, uint64_t* __restrict__ tle, uint64_t* __restrict__ tbad, uint64_t nt, int k)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    uint64_t le, bad;
    encode_target(traw + t * TARGET_SLOT, k, le, bad);
    tle[t] = le;
    tbad[t] = bad;
}

/* terminal-node search + coverage of the single-contig solution, one wave per gap; leaves the slot's record with what the gap will
 * contribute to the arrays of its batch (mtg_emit.h: emit_plan).  Where it goes is decided by the scan kernels below. */
/* the piece index of a batch's dictionaries (mtg_post.h): a workgroup per gap and turn, its targets dealt to the threads */
__global__ void __launch_bounds__(256) k_post_index(uint32_t* head, uint32_t* next, uint32_t mask, const uint64_t* __restrict__ tle, const uint64_t* __restrict__ tbad, const uint32_t* __restrict__ toff,
                                                    const uint32_t* __restrict__ tcnt, const uint8_t* __restrict__ nbmis, uint32_t n, int k)
{
    for (uint32_t g = blockIdx.x; g < n; g += gridDim.x) {
        const uint32_t c = tcnt[g], o = toff[g], mis = nbmis[g];
        if (c < (uint32_t)POST_INDEX_MIN) continue; /* (such a gap's search goes over its targets) */
        for (uint32_t t = threadIdx.x; t < c; t += blockDim.x) post_index_add(head, next, mask, g, o + t, tle[o + t], tbad[o + t], mis, k);
    }
}
#ifndef MTG_POST_WAVES
#define MTG_POST_WAVES 6
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MTG_POST_WAVES))) k_post(Index ix, FillCfg cfg, uint8_t* raw, uint8_t* head, const GapOut* __restrict__ outs, const uint32_t* __restrict__ ids,
                                             const uint64_t* __restrict__ tle, const uint64_t* __restrict__ tbad, const uint32_t* __restrict__ toff,
                                             const uint32_t* __restrict__ tcnt, const uint8_t* __restrict__ nbmis, const uint8_t* __restrict__ fast_ok,
                                             uint32_t want_all, SlotRec* recs, uint32_t n, ParkCtl* park, const uint32_t* __restrict__ pi_head, const uint32_t* __restrict__ pi_next, uint32_t pi_mask)
{
    __shared__ uint32_t hist[256];
    __shared__ uint64_t tile[POST_TILE + 2];
    __shared__ uint64_t s_blk[64];
    /* The gaps of POST_LIST (lean_and_list) (everything that is not lean; k_post_lean below has the others).  One workgroup per gap measured
     * best (against persistent workgroups): the kernel lives on the number of waves in flight -- the host sizes the grid from what the
     * previous launch listed, and the loop takes what a launch lists beyond that. */
    const uint32_t n_listed = park->count[POST_LIST];
    const uint32_t* list = park_list(park, n, POST_LIST);
    for (uint32_t li = blockIdx.x; li < n_listed; li += gridDim.x) {
        const uint32_t slot = list[li];
        __syncthreads(); /* the previous gap's readers of hist are done */
        for (uint32_t i = threadIdx.x; i < 256; i += 64) hist[i] = 0;
        __syncthreads();
        const GapOut o = outs[slot];
        PostOut po;
        po.nb_terminal = po.fast = po.pos = po.errors = po.target = po.clen0 = po.ab_sum = po.ab_n = po.med_hi = po.med_lo = po.lines = po.direct = po.lean = 0;
        GapScratch S;
        S.z = nullptr;
        S.v = nullptr;
        S.lane = slot & 63u;
        S.r = raw + (uint64_t)slot * cfg.raw_stride;
        S.h = head + (uint64_t)(slot >> 6) * cfg.hd_stride;
        if (o.status == GAP_OK) {
            const uint32_t g = ids ? ids[slot] : slot;
            PostTargets T;
            T.le = tle + toff[g];
            T.bad = tbad + toff[g];
            T.n = tcnt[g];
            T.nb_mis = nbmis[g];
            T.fast_ok = fast_ok[g];
            T.pi_head = pi_head; T.pi_next = pi_next; T.pi_mask = pi_mask; T.gbase = toff[g]; T.gid = g;
#ifdef MTG_POST_DBG /* timing experiments only (scripts/exp_post_parts.sh): parts of the kernel switched off, results wrong */
            post_gap(ix, cfg, S, o, T, hist, tile, s_blk, po, MTG_POST_DBG);
#else
            post_gap(ix, cfg, S, o, T, hist, tile, s_blk, po);
#endif
        }
        /* the record leaves lane 0 in ten 16-byte stores (SlotRec is 16-byte aligned): field by field it made the kernel write 1.1 KB of
         * partial lines per gap (PMC WRITE_SIZE 111.6 MB per launch for 15 MB of records, round 3).  A coalesced store of the wave through
         * LDS was measured as well: fewer bytes still, but 36 us more -- the kernel is bound by instruction issue, not by its traffic. */
        if (threadIdx.x == 0) {
            SlotRec r;
            r.o = o; r.p = po;
            emit_plan(o, po, want_all != 0, ix.k, r.nw, r.nc, r.asc, r.ext); /* po is uniform over the wave */
            r.wbase = r.cbase = r.abase = r.ebase = 0;
            r.rpos = r.gpos = 0;
            r.fpos = r.pad_ = 0;
            r.pad2_[0] = r.pad2_[1] = 0;
            recs[slot] = r;
        }
    }
}

/* the lean gaps (mtg_post.h: post_lean_*): eight lanes per gap, eight gaps per wave; a gap that is not lean is left to k_post.
 * Measured on the haploid set, k_post + scans of one batch alone: 0.222 ms with a wave per gap (round 3), 0.185 with 4 lanes per gap, 0.128 with 8, 0.136 with 16. */
#ifndef MTG_POST_LEAN_G
#define MTG_POST_LEAN_G 8
#endif
enum { POST_LEAN_G = MTG_POST_LEAN_G };
__global__ void __launch_bounds__(64) k_post_lean(Index ix, FillCfg cfg, uint8_t* raw, uint8_t* head, const GapOut* __restrict__ outs, uint32_t want_all, SlotRec* recs, uint32_t n)
{
    __shared__ uint32_t hist[64 / POST_LEAN_G][256];
    const uint32_t grp = threadIdx.x / POST_LEAN_G, gl = threadIdx.x % POST_LEAN_G;
    const uint32_t slot = blockIdx.x * (64u / POST_LEAN_G) + grp;
    for (uint32_t i = gl; i < 256u; i += POST_LEAN_G) hist[grp][i] = 0;
    __syncthreads();
    GapOut o;
    LeanWork w;
    bool lean = false;
    if (slot < n) {
        o = outs[slot];
        GapScratch S;
        S.z = nullptr;
        S.v = nullptr;
        S.lane = slot & 63u;
        S.r = raw + (uint64_t)slot * cfg.raw_stride;
        S.h = head + (uint64_t)(slot >> 6) * cfg.hd_stride;
        lean = post_lean_accumulate<POST_LEAN_G>(ix, cfg, S, o, gl, hist[grp], w);
    }
    __syncthreads(); /* the histograms are complete */
    if (!lean) return;
    PostOut po;
    post_lean_finish<POST_LEAN_G>(w, gl, hist[grp], po);
    if (gl == 0) {
        SlotRec r;
        r.o = o; r.p = po;
        emit_plan(o, po, want_all != 0, ix.k, r.nw, r.nc, r.asc, r.ext);
        r.wbase = r.cbase = r.abase = r.ebase = 0;
        r.rpos = r.gpos = 0;
        r.fpos = r.pad_ = 0;
        r.pad2_[0] = r.pad2_[1] = 0;
        recs[slot] = r;
    }
}

/* ---- where every slot's output goes: exclusive prefix sums, in slot order, of what the slots contribute to the dense words, the dense
 * metadata, the sequence arena, the extension arena, the list of gaps to re-run and the list of multi-contig gaps.  k_scan1: one thread per
 * slot, offsets inside its block of SCAN_SL slots + the block's totals and statistics; k_scan2 (one workgroup): offsets of the blocks on
 * top of the batch's cursors, totals of the launch; k_emit adds the two. */
enum { SCAN_SL = 256, SCAN_NV = 7, SCAN_NS = 16 };
struct ScanBlock {
    uint64_t v[SCAN_NV]; /* k_scan1: totals of the block; k_scan2: replaced by the block's base */
    uint64_t s[SCAN_NS]; /* sums: lines, store_runs, run_nt, contig_nt, contig_words, post_lines, cov_kmers, n_filled, n_ext, copy_words, copy_cmds, cov_direct, n_lean,
                            copy words / commands k_copy executed (not those of lean gaps), contig words k_post scanned (not those of lean gaps) */
};
__global__ void __launch_bounds__(SCAN_SL) k_scan1(SlotRec* recs, uint32_t m, ScanBlock* blocks)
{
    /* scans inside the waves by shuffles, the four waves of the block joined through a few words of LDS; the statistics are reduced the same
     * way (a block's contributions fit 32 bits: 256 slots of at most 2^20 words / bytes each... the sums are kept in 64 bits all the same) */
    enum { NW = SCAN_SL / 64 };
    __shared__ uint64_t wtot[NW][SCAN_NV];
    __shared__ unsigned long long wsum[NW][SCAN_NS];
    const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6, slot = blockIdx.x * SCAN_SL + t;
    uint64_t v[SCAN_NV] = {0, 0, 0, 0, 0, 0, 0};
    unsigned long long st[SCAN_NS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (slot < m) {
        const SlotRec& r = recs[slot];
        const bool ok = r.o.status == GAP_OK;
        v[0] = r.nw; v[1] = r.nc; v[2] = r.asc; v[3] = r.ext;
        v[4] = ok ? 0 : 1;
        v[5] = (ok && r.nc) ? 1 : 0; /* its contigs go back to the host: multi-contig path, or the stage-A entry */
        v[6] = r.asc ? 1 : 0;        /* filled on the common path: one solution */
        st[0] = r.o.lines; st[1] = r.o.store_reads; st[2] = r.o.run_nt; st[4] = r.o.n_words;
        if (r.o.n_cmds) { st[9] = r.o.copy_words; st[10] = r.o.n_cmds; }
        if (ok) {
            st[3] = r.o.total_nt; st[5] = r.p.lines; st[6] = r.p.ab_n;
            if (r.p.direct) st[11] = r.p.ab_n;
            st[7] = r.asc ? 1 : 0; st[8] = r.ext ? 1 : 0; st[12] = r.p.lean ? 1 : 0;
        }
        /* a lean gap's commands are never executed and its contig is never scanned: what k_copy and k_post really touched */
        if (!(ok && r.p.lean)) { if (r.o.n_cmds) { st[13] = r.o.copy_words; st[14] = r.o.n_cmds; } if (ok) st[15] = r.o.n_words; }
    }
    uint64_t incl[SCAN_NV];
    for (int j = 0; j < SCAN_NV; j++) {
        uint64_t x = v[j];
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)x, d, 64), hi = (uint32_t)__shfl_up((int)(uint32_t)(x >> 32), d, 64);
            if ((int)lane >= d) x += ((uint64_t)hi << 32) | lo;
        }
        incl[j] = x;
        if (lane == 63) wtot[wv][j] = x;
    }
    for (int j = 0; j < SCAN_NS; j++) {
        unsigned long long x = st[j];
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)x, d, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(x >> 32), d, 64);
            x += ((unsigned long long)hi << 32) | lo;
        }
        if (lane == 0) wsum[wv][j] = x;
    }
    __syncthreads();
    uint64_t before[SCAN_NV];
    for (int j = 0; j < SCAN_NV; j++) {
        uint64_t x = 0;
        for (uint32_t w2 = 0; w2 < wv; w2++) x += wtot[w2][j];
        before[j] = x;
    }
    if (slot < m) {
        SlotRec& r = recs[slot];
        r.wbase = before[0] + incl[0] - v[0]; r.cbase = before[1] + incl[1] - v[1]; r.abase = before[2] + incl[2] - v[2]; r.ebase = before[3] + incl[3] - v[3];
        r.rpos = (uint32_t)(before[4] + incl[4] - v[4]); r.gpos = (uint32_t)(before[5] + incl[5] - v[5]); r.fpos = (uint32_t)(before[6] + incl[6] - v[6]);
    }
    if (t < SCAN_NV) { uint64_t x = 0; for (int w2 = 0; w2 < NW; w2++) x += wtot[w2][t]; blocks[blockIdx.x].v[t] = x; }
    if (t < SCAN_NS) { unsigned long long x = 0; for (int w2 = 0; w2 < NW; w2++) x += wsum[w2][t]; blocks[blockIdx.x].s[t] = x; }
}
/* begin.v[0..3]: words, metadata entries, sequence bytes, extension bytes of the batch so far (the host knows them: it has seen the previous
 * launch's totals).  The totals go to the device copy k_emit reads AND straight into the host's page-locked copy (host_tot: a store over the
 * link instead of a copy the host would have to queue), with the number of gaps the walk kernel parked. */
struct ScanBegin { uint64_t v[4]; };
__global__ void __launch_bounds__(256) k_scan2(ScanBlock* blocks, uint32_t nblocks, ScanBegin begin, PartTot* tot, PartTot* host_tot, const ParkCtl* park)
{
    enum { TILE = 1024 };
    __shared__ uint64_t sh[SCAN_NV][TILE];
    __shared__ uint64_t carry[SCAN_NV];
    __shared__ unsigned long long ssum[SCAN_NS];
    const uint32_t t = threadIdx.x;
    if (t < SCAN_NV) carry[t] = t < 4 ? begin.v[t] : 0;
    if (t < SCAN_NS) ssum[t] = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < nblocks; b0 += TILE) {
        const uint32_t nb = nblocks - b0 < (uint32_t)TILE ? nblocks - b0 : (uint32_t)TILE;
        for (uint32_t i = t; i < nb * SCAN_NV; i += 256) sh[i % SCAN_NV][i / SCAN_NV] = blocks[b0 + i / SCAN_NV].v[i % SCAN_NV];
        for (uint32_t i = t; i < nb * SCAN_NS; i += 256) atomicAdd(&ssum[i % SCAN_NS], (unsigned long long)blocks[b0 + i / SCAN_NS].s[i % SCAN_NS]);
        __syncthreads();
        /* the blocks' totals become their bases: a column per wave (waves 0 and 1 take a second one), 64 blocks per shuffle scan */
        for (uint32_t j = t >> 6; j < SCAN_NV; j += 4) {
            const uint32_t lane = t & 63u;
            uint64_t run = carry[j];
            for (uint32_t c0 = 0; c0 < nb; c0 += 64) {
                const uint32_t i = c0 + lane;
                const uint64_t x0 = i < nb ? sh[j][i] : 0ull;
                uint64_t x = x0;
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)x, d, 64), hi = (uint32_t)__shfl_up((int)(uint32_t)(x >> 32), d, 64);
                    if ((int)lane >= d) x += ((uint64_t)hi << 32) | lo;
                }
                if (i < nb) sh[j][i] = run + x - x0;
                const uint32_t tl = (uint32_t)__shfl((int)(uint32_t)x, 63, 64), th = (uint32_t)__shfl((int)(uint32_t)(x >> 32), 63, 64);
                run += ((uint64_t)th << 32) | tl;
            }
            if (lane == 0) carry[j] = run;
        }
        __syncthreads();
        for (uint32_t i = t; i < nb * SCAN_NV; i += 256) blocks[b0 + i / SCAN_NV].v[i % SCAN_NV] = sh[i % SCAN_NV][i / SCAN_NV];
        __syncthreads();
    }
    __syncthreads();
    if (t == 0) { /* one thread, from the block's shared sums: the same record to the device's copy and to the host's */
        PartTot pt;
        for (int j = 0; j < 4; j++) { pt.begin[j] = begin.v[j]; pt.end[j] = carry[j]; }
        pt.n_retry = (uint32_t)carry[4];
        pt.n_general = (uint32_t)carry[5];
        pt.lines = ssum[0]; pt.store_runs = ssum[1]; pt.run_nt = ssum[2]; pt.contig_nt = ssum[3]; pt.contig_words = ssum[4];
        pt.post_lines = ssum[5]; pt.cov_kmers = ssum[6];
        pt.n_filled = (uint32_t)ssum[7]; pt.n_ext = (uint32_t)ssum[8];
        pt.copy_words = ssum[9]; pt.copy_cmds = ssum[10]; pt.cov_direct = ssum[11]; pt.n_lean = ssum[12];
        pt.copy_words_exec = ssum[13]; pt.copy_cmds_exec = ssum[14]; pt.scan_words = ssum[15];
        pt.n_parked = park ? park->count[0] : 0u;
        pt.n_branching = park ? park->n_branching : 0u;

        *tot = pt;
        if (host_tot) *host_tot = pt;
    }
}
/* everything a gap leaves behind (mtg_emit.h: emit_gap), one wave per slot */
__global__ void __launch_bounds__(64) k_emit(UStore us, FillCfg cfg, uint8_t* raw, uint8_t* head, SlotRec* recs, const ScanBlock* __restrict__ blocks, const uint32_t* __restrict__ ids,
                                             const uint8_t* __restrict__ gflags, int k, EmitDev D, EmitHost H, uint32_t* retry_list, uint32_t* general_list, uint32_t n, ParkCtl* park,
                                             uint32_t use_list)
{
    /* use_list: the gaps of POST_LIST (everything that is not lean: k_emit_lean has the others), the grid sized by the host from what the
     * previous launch listed; otherwise (a batch that leaves in relocatable form) every slot of the launch */
    const uint32_t count = use_list ? park->count[POST_LIST] : n;
    __shared__ SlotRec r;
    for (uint32_t li = blockIdx.x; li < count; li += gridDim.x) {
    const uint32_t slot = use_list ? park_list(park, n, POST_LIST)[li] : li;
    __syncthreads(); /* the previous gap's readers of r are done */
    if (threadIdx.x == 0) {
        r = recs[slot];
        const ScanBlock& b = blocks[slot / SCAN_SL];
        r.wbase += b.v[0]; r.cbase += b.v[1]; r.abase += b.v[2]; r.ebase += b.v[3];
        r.rpos += (uint32_t)b.v[4]; r.gpos += (uint32_t)b.v[5]; r.fpos += (uint32_t)b.v[6];
        recs[slot] = r; /* absolute from here on (the host reads the records of the gaps it has to look at) */
        if (r.o.status != GAP_OK) retry_list[r.rpos] = slot;
        else if (r.nc) general_list[r.gpos] = slot;
    }
    __syncthreads();
    GapScratch S;
    S.z = nullptr;
    S.v = nullptr;
    S.lane = slot & 63u;
    S.r = raw + (uint64_t)slot * cfg.raw_stride;
    S.h = head + (uint64_t)(slot >> 6) * cfg.hd_stride;
    const uint32_t g = ids ? ids[slot] : slot;
    emit_gap(us, cfg, S, r, gflags[g], slot, g, k, D, H);
    }
}
/* the lean gaps (mtg_emit.h: emit_lean): eight lanes per gap, eight gaps per wave (k_emit of one haploid batch alone: 0.063 ms with 4 lanes per gap, 0.055 with 8, 0.052 with 16) */
#ifndef MTG_EMIT_LEAN_G
#define MTG_EMIT_LEAN_G 8
#endif
enum { EMIT_LEAN_G = MTG_EMIT_LEAN_G };
__global__ void __launch_bounds__(64) k_emit_lean(UStore us, FillCfg cfg, uint8_t* raw, uint8_t* head, const SlotRec* __restrict__ recs, const ScanBlock* __restrict__ blocks, const uint32_t* __restrict__ ids,
                                                  const uint8_t* __restrict__ gflags, int k, EmitDev D, EmitHost H, uint32_t n)
{
    const uint32_t slot = blockIdx.x * (64u / EMIT_LEAN_G) + threadIdx.x / EMIT_LEAN_G, gl = threadIdx.x % EMIT_LEAN_G;
    if (slot >= n) return;
    const SlotRec r = recs[slot];
    if (!emit_is_lean(r, D)) return;
    GapScratch S;
    S.z = nullptr;
    S.v = nullptr;
    S.lane = slot & 63u;
    S.r = raw + (uint64_t)slot * cfg.raw_stride;
    S.h = head + (uint64_t)(slot >> 6) * cfg.hd_stride;
    const uint32_t g = ids ? ids[slot] : slot;
    emit_lean<EMIT_LEAN_G>(us, cfg, S, r, r.abase + blocks[slot / SCAN_SL].v[2], gflags[g], slot, g, k, D, H, gl);
}

/* checksum of a relocatable batch's body into its header (mtg_wire_header::checksum): a sum of scrambled 64-bit words, any order */
__global__ void __launch_bounds__(256) k_wire_sum(uint8_t* wire, uint64_t cap)
{
    mtg_wire_header* h = reinterpret_cast<mtg_wire_header*>(wire);
    if (h->magic != 0x3145524957474D54ull || h->total_bytes > cap) return; /* the batch did not leave in this form (the host knows from the totals) */
    const uint64_t* w = reinterpret_cast<const uint64_t*>(wire + sizeof(mtg_wire_header));
    const uint64_t n = (h->total_bytes - sizeof(mtg_wire_header)) / 8;
    uint64_t s = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) s += wire_word_sum(w[i], i);
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)s, d, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(s >> 32), d, 64);
        s += ((uint64_t)hi << 32) | lo;
    }
    if ((threadIdx.x & 63u) == 0 && s) atomicAdd(reinterpret_cast<unsigned long long*>(&h->checksum), (unsigned long long)s);
}

/* contig-graph walk of the multi-contig gaps of a chunk (mtg_paths.h), one wave per gap */
__global__ void __launch_bounds__(64) k_paths(FillCfg cfg, uint8_t* raw, uint8_t* head, const GapOut* __restrict__ outs, const uint32_t* __restrict__ slots, int k, uint32_t* out, uint32_t n)
{
    __shared__ PathsWork W;
    if (blockIdx.x >= n) return;
    const uint32_t slot = slots[blockIdx.x];
    GapScratch S;
    S.z = nullptr;
    S.v = nullptr;
    S.lane = slot & 63u;
    S.r = raw + (uint64_t)slot * cfg.raw_stride;
    S.h = head + (uint64_t)(slot >> 6) * cfg.hd_stride;
    paths_gap(cfg, S, outs[slot], k, W, out + (uint64_t)blockIdx.x * PATHS_WORDS);
}

/* the multi-contig gaps of a launch on the device (mtg_general.h), one wave per gap of the launch's list: paths -> candidate sequences ->
 * de-duplication -> coverage, quality, ASCII.  slots = the list k_emit made (general_list); paths = k_paths' blocks in the same order. */
__global__ void __launch_bounds__(64) k_general(Index ix, FillCfg cfg, uint8_t* raw, uint8_t* head, const GapOut* __restrict__ outs, const uint32_t* __restrict__ slots, const uint32_t* __restrict__ ids,
                                                const uint32_t* __restrict__ paths, const uint32_t* __restrict__ tcnt, const uint8_t* __restrict__ fast_ok, const uint64_t* __restrict__ src,
                                                const uint8_t* __restrict__ gflags, int k, GenDev D, uint32_t n)
{
    __shared__ GenWork W;
    if (blockIdx.x >= n) return;
    const uint32_t slot = slots[blockIdx.x];
    const uint32_t g = ids ? ids[slot] : slot;
    GapScratch S;
    S.z = nullptr;
    S.v = nullptr;
    S.lane = slot & 63u;
    S.r = raw + (uint64_t)slot * cfg.raw_stride;
    S.h = head + (uint64_t)(slot >> 6) * cfg.hd_stride;
    gen_gap(ix, cfg, S, outs[slot], k, paths + (uint64_t)blockIdx.x * PATHS_WORDS, tcnt[g], fast_ok[g] != 0, src[g], gflags[g], D, blockIdx.x, W);
}


/* device copies of a marshalled batch: blocks A and B and the encoded targets (block C only serves to make those) */
int batch_upload(const mtg_index* idx, FillInput& in)
{
    if (int rc = use_device_of(idx)) return rc;
    batch_release_device(in);
    DevBuf a, b, c, t;
    const uint64_t n_targets = in.traw.size() / TARGET_SLOT;
    HIP_TRY(a.alloc(in.bytes_a));
    HIP_TRY(b.alloc(in.bytes_b));
    HIP_TRY(c.alloc(in.bytes_c));
    HIP_TRY(t.alloc(n_targets * 16 + 64));
    HIP_TRY(hipMemcpy(a.p, in.block_a, in.bytes_a, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b.p, in.block_b, in.bytes_b, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c.p, in.block_c, in.bytes_c, hipMemcpyHostToDevice));
    if (n_targets) {
        hipLaunchKernelGGL(k_encode_targets, dim3((unsigned)((n_targets + 255) / 256)), dim3(256), 0, 0, c.as<uint8_t>(), t.as<uint64_t>(), t.as<uint64_t>() + n_targets, n_targets, in.k);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipDeviceSynchronize());
    in.dev_a = a.release();
    in.dev_b = b.release();
    in.dev_tenc = t.release();
    return MTG_OK;
}
void batch_release_device(FillInput& in)
{
    if (in.dev_a) (void)hipFree(in.dev_a);
    if (in.dev_b) (void)hipFree(in.dev_b);
    if (in.dev_tenc) (void)hipFree(in.dev_tenc);
    in.dev_a = in.dev_b = in.dev_tenc = nullptr;
}


/* The caller holds the lock of in.ws (a workspace and its staging blocks belong to one batch at a time); everything the batch queues
 * goes to the workspace's own stream, so that two batches on the device overlap.
 *
 * One launch = as many gaps as fit the scratch: traversal (k_stage_a), terminal search + coverage (k_post), layout of the results
 * (k_scan1, k_scan2), results (k_emit); then the totals come back, and with them the sizes of the copies that bring the records and the
 * sequences to the host arrays of `sink`.  The host looks at a gap only if it has to be re-run in a larger scratch tier or takes the
 * multi-contig path (`special`). */
static hipStream_t upload_stream_of(int dev)
{
    static std::mutex m;
    static hipStream_t s[CopyTurn::MAX_DEV] = {};
    if (tune::on(tune::T_UPLOAD_OWN_STREAM)) return nullptr;
    std::lock_guard<std::mutex> lk(m);
    hipStream_t& r = s[(unsigned)dev % CopyTurn::MAX_DEV];
    if (!r && hipStreamCreateWithFlags(&r, hipStreamNonBlocking) != hipSuccess) r = nullptr;
    return r;
}

int device_run(const mtg_index* idx, const mtg_params* p, const FillInput& in, ResultSink& sink, DevBatch& special, mtg_batch_stats* stats, const std::function<void()>* while_busy)
{
    bool busy_done = false;
    const bool dbg = tune::on(tune::T_DEBUG_TIMERS);
    double tk = now_ms();
    auto tick = [&](const char* what) { if (dbg) { double t = now_ms(); fprintf(stderr, "  [device_run] %-18s %.2f ms\n", what, t - tk); tk = t; } };
    if (int rc = use_device_of(idx)) return rc;
    if (!in.ws) { set_error("device_run: the input has no workspace"); return MTG_ERR_ARG; }
    Workspace& ws = *in.ws;
    if (!ws.stream) {
        hipStream_t s0, s1;
        HIP_TRY(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
        ws.stream = (void*)s0;
        HIP_TRY(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
        ws.copy_stream = (void*)s1;
        HIP_TRY(hipDeviceSynchronize()); /* the index was built or loaded on the null stream, which these streams do not wait for */
    }
    const hipStream_t stream = (hipStream_t)ws.stream;
    const size_t n = in.src.size();
    enum : size_t { SEQ_SLACK = 64 }; /* the device's formatter measures a sequence 64 characters at a time (fmt_strlen): the last one may be read that far past its NUL */
    special.chunks.clear();
    special.special.clear();
    sink.seq_used = 0;
    sink.ext_used = 1;
    sink.n_filled = 0;
    sink.in_gap_order = true;
    mtg_batch_stats st = stats ? *stats : mtg_batch_stats{};
    if (n == 0) { if (while_busy) (*while_busy)(); if (stats) *stats = st; return MTG_OK; }
    const int k = idx->dev.k;

    int ws_next = 0;
    auto wsbuf = [&]() { WsBuf b; b.ws = &ws; b.slot = ws_next++; return b; };
    WsBuf d_ina = wsbuf(), d_inb = wsbuf(), d_inc = wsbuf(), d_tenc = wsbuf(), d_ilv = wsbuf(), d_zero = wsbuf(), d_raw = wsbuf(), d_out = wsbuf(), d_rec = wsbuf(), d_ids = wsbuf(), d_dw = wsbuf(),
          d_dm = wsbuf(), d_combo = wsbuf(), d_blocks = wsbuf(), d_seq = wsbuf(), d_ext = wsbuf(), d_res = wsbuf(), d_fil = wsbuf(), d_tot = wsbuf(), d_rlist = wsbuf(), d_glist = wsbuf(), d_paths = wsbuf(), d_park = wsbuf(), d_ggaps = wsbuf(), d_gsols = wsbuf(), d_gascii = wsbuf(), d_gtmp = wsbuf(), d_gbnd = wsbuf(), d_head = wsbuf(), d_pidx = wsbuf();
    /* the marshalled input: three blocks, three copies; the targets (block C, text) become k-mers and masks on the device.  A batch that
     * was prepared ahead (mtg_batch) is resident already */
    double t0 = now_ms();
    const uint64_t n_targets = in.text_mode ? in.n_text_targets : in.traw.size() / TARGET_SLOT;
    /* the events of a batch belong to its workspace: made once (round 4 created and destroyed nine per batch: eighteen runtime calls of the ~50 a
     * batch made, and the runtime serialises them across the caller threads) */
    struct WsEvents {
        Workspace& w;
        int next = 0;
        hipError_t make(hipEvent_t& e)
        {
            if (next >= Workspace::NEVENTS) return hipErrorOutOfMemory;
            if (!w.events[next]) { hipEvent_t ne; const hipError_t r = hipEventCreateWithFlags(&ne, hipEventBlockingSync); if (r != hipSuccess) return r; w.events[next] = (void*)ne; }
            e = (hipEvent_t)w.events[next++];
            return hipSuccess;
        }
    } events{ws};
    hipStream_t up = (in.text_mode || !in.dev_a) ? upload_stream_of(idx->device) : nullptr;
    if (!up) up = stream;
    auto uploaded = [&]() -> int { /* the batch's stream goes on when its blocks have arrived */
        if (up == stream) return MTG_OK;
        hipEvent_t evu;
        HIP_TRY(events.make(evu));
        HIP_TRY(hipEventRecord(evu, up));
        HIP_TRY(hipStreamWaitEvent(stream, evu, 0));
        return MTG_OK;
    };
    const uint8_t* da;
    const uint64_t* d_rw;
    uint64_t* d_tle;
    if (in.text_mode) {
        /* the strings are still text: block A (integer columns) and the text block go up, the device encodes (mtg_marshal.h) */
        HIP_TRY(d_ina.alloc(in.bytes_a));
        HIP_TRY(d_inb.alloc(in.bytes_b));
        HIP_TRY(d_inc.alloc(in.bytes_c));
        HIP_TRY(hipMemcpyAsync(d_ina.p, in.block_a, in.bytes_a, hipMemcpyHostToDevice, up));
        if (in.text_direct) { /* the block from the caller's page-locked memory, the offset arrays from the staging block */
            const size_t off5 = FillInput::text_block_off(n, (size_t)n_targets, 5);
            HIP_TRY(hipMemcpyAsync(d_inc.p, in.block_c, off5, hipMemcpyHostToDevice, up));
            HIP_TRY(hipMemcpyAsync((uint8_t*)d_inc.p + off5, in.text_direct, in.text_bytes, hipMemcpyHostToDevice, up));
        } else HIP_TRY(hipMemcpyAsync(d_inc.p, in.block_c, in.bytes_c, hipMemcpyHostToDevice, up));
        if (int rc = uploaded()) return rc;
        HIP_TRY(d_tenc.alloc(n_targets * 16 + 64));
        uint8_t* a = d_ina.as<uint8_t>();
        const uint8_t* c = d_inc.as<uint8_t>();
        const size_t nn = n, nt = (size_t)n_targets;
        hipLaunchKernelGGL(k_marshal_text, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, c + FillInput::text_block_off(nn, nt, 5), (const uint64_t*)(c + FillInput::text_block_off(nn, nt, 0)),
                           (const uint32_t*)(c + FillInput::text_block_off(nn, nt, 3)), (const uint64_t*)(c + FillInput::text_block_off(nn, nt, 1)), (const uint32_t*)(a + FillInput::off_a(n, 2)),
                           (uint32_t*)(a + FillInput::off_a(n, 3)), (uint64_t*)(a + FillInput::off_a(n, 0)), (uint64_t*)(a + FillInput::off_a(n, 1)), a + FillInput::off_a(n, 7), d_inb.as<uint64_t>(),
                           (uint32_t)n, k);
        da = a;
        d_rw = d_inb.as<uint64_t>();
        d_tle = d_tenc.as<uint64_t>();
        if (n_targets) hipLaunchKernelGGL(k_marshal_targets, dim3((unsigned)((n_targets + 255) / 256)), dim3(256), 0, stream, c + FillInput::text_block_off(nn, nt, 5),
                                          (const uint64_t*)(c + FillInput::text_block_off(nn, nt, 2)), (const uint32_t*)(c + FillInput::text_block_off(nn, nt, 4)), d_tle, d_tle + n_targets, (uint32_t)n_targets, k);
        HIP_TRY(hipGetLastError());
    } else if (in.dev_a) {
        da = (const uint8_t*)in.dev_a;
        d_rw = (const uint64_t*)in.dev_b;
        d_tle = (uint64_t*)in.dev_tenc;
    } else {
        HIP_TRY(d_ina.alloc(in.bytes_a));
        HIP_TRY(d_inb.alloc(in.bytes_b));
        HIP_TRY(d_inc.alloc(in.bytes_c));
        HIP_TRY(hipMemcpyAsync(d_ina.p, in.block_a, in.bytes_a, hipMemcpyHostToDevice, up));
        HIP_TRY(hipMemcpyAsync(d_inb.p, in.block_b, in.bytes_b, hipMemcpyHostToDevice, up));
        HIP_TRY(hipMemcpyAsync(d_inc.p, in.block_c, in.bytes_c, hipMemcpyHostToDevice, up));
        if (int rc = uploaded()) return rc;
        HIP_TRY(d_tenc.alloc(n_targets * 16 + 64));
        da = d_ina.as<uint8_t>();
        d_rw = d_inb.as<uint64_t>();
        d_tle = d_tenc.as<uint64_t>();
        if (n_targets) hipLaunchKernelGGL(k_encode_targets, dim3((unsigned)((n_targets + 255) / 256)), dim3(256), 0, stream, d_inc.as<uint8_t>(), d_tle, d_tle + n_targets, n_targets, k);
    }
    const uint64_t* d_src = (const uint64_t*)(da + FillInput::off_a(n, 0));
    const uint64_t* d_r0 = (const uint64_t*)(da + FillInput::off_a(n, 1));
    const uint32_t* d_roff = (const uint32_t*)(da + FillInput::off_a(n, 2));
    const uint32_t* d_rlen = (const uint32_t*)(da + FillInput::off_a(n, 3));
    const uint32_t* d_toff = (const uint32_t*)(da + FillInput::off_a(n, 4));
    const uint32_t* d_tcnt = (const uint32_t*)(da + FillInput::off_a(n, 5));
    const uint8_t* d_mis = da + FillInput::off_a(n, 6);
    const uint8_t* d_fok = da + FillInput::off_a(n, 7);
    const uint8_t* d_flags = da + FillInput::off_a(n, 8);
    uint64_t* d_tbad = d_tle + n_targets;
    /* dictionaries of many targets (contig mode: every seed has the targets of all other contigs): their pieces in one hash table, so that the
     * terminal search looks a contig position up instead of counting against every target (mtg_post.h: the piece index) */
    uint32_t* d_pi_head = nullptr;
    uint32_t* d_pi_next = nullptr;
    uint32_t pi_mask = 0;
    if (n_targets >= (uint64_t)POST_INDEX_MIN * n && n_targets < (1ull << 30) && !tune::on(tune::T_NO_POST_INDEX)) {
        uint64_t cap = 1024;
        while (cap < 4 * n_targets) cap <<= 1;
        HIP_TRY(d_pidx.alloc((cap + 4 * n_targets) * 4));
        d_pi_head = d_pidx.as<uint32_t>();
        d_pi_next = d_pi_head + cap;
        pi_mask = (uint32_t)(cap - 1);
        HIP_TRY(hipMemsetAsync(d_pi_head, 0xFF, cap * 4, stream));
        hipLaunchKernelGGL(k_post_index, dim3((unsigned)std::min<size_t>(n, 4096)), dim3(256), 0, stream, d_pi_head, d_pi_next, pi_mask, (const uint64_t*)d_tle, (const uint64_t*)d_tbad, d_toff, d_tcnt, d_mis, (uint32_t)n, k);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(d_tot.alloc(sizeof(PartTot)));
    /* where the batch's two arenas stand: sequence bytes and extension bytes so far (the extension arena starts with the empty string); the layout
     * kernel gets them by value */
    uint64_t arena_used[2] = {0, 1};
    st.h2d_ms += now_ms() - t0;
    tick("upload (async)");

    /* ev0 .. ev2: the kernels of a launch, ev3: its results on the host.  KERNEL_TIMERS of the tuning table: the six events between the kernels as
     * well (mtg_last_batch_stats then carries every kernel's own time: what bench.py's pass with one batch alone asks for); off, a batch records three
     * events instead of nine and asks for one elapsed time instead of seven */
    const bool ktimers = tune::on(tune::T_KERNEL_TIMERS);
    hipEvent_t ev0, ev1 = nullptr, ev2, ev3, eve = nullptr, evc = nullptr, evf = nullptr, evl = nullptr, evl0 = nullptr;
    HIP_TRY(events.make(ev0));
    HIP_TRY(events.make(ev2));
    HIP_TRY(events.make(ev3));
    if (ktimers) {
        HIP_TRY(events.make(eve));
        HIP_TRY(events.make(evc));
        HIP_TRY(events.make(ev1));
        HIP_TRY(events.make(evf));
        HIP_TRY(events.make(evl));
        HIP_TRY(events.make(evl0));
    }
    auto mark = [&](hipEvent_t e) -> hipError_t { return e ? hipEventRecord(e, stream) : hipSuccess; };
    PartTot* h_tot = (PartTot*)staging_host(&ws, Workspace::NHOST - 1, sizeof(PartTot) + 64);
    if (!h_tot) { set_error("no page-locked memory for the totals of a launch"); return MTG_ERR_NOMEM; }

    std::vector<uint32_t> todo; /* empty at tier 0: every gap, in order */
    size_t n_todo = n;
    int rc = MTG_OK;
    const bool host_paths = tune::on(tune::T_HOST_PATHS); /* test hook: leave the path enumeration to the host */
    const bool want_records = sink.res != nullptr;
    size_t launches = 0;

    for (int tier = 0; tier <= MTG_MAX_TIER && n_todo; tier++) {
        FillCfg cfg = make_cfg(k, p->max_nodes, p->max_depth, p->end_rule_nonbranching, tier);
        const bool no_defer = tune::on(tune::T_NO_DEFER); /* test hook: the lanes of the traversal copy their long runs themselves */
        if (no_defer || !idx->dev.us.nwords) cfg.cmd_cap = 0;
        /* scratch of a gap + worst-case room in the dense arrays (its whole contig arena and the metadata of every contig) */
        const uint64_t per_gap = cfg.zero_stride + cfg.raw_stride + cfg.ilv_stride / 64 + cfg.hd_stride / 64 + sizeof(GapOut) + sizeof(SlotRec) + 64 + sizeof(mtg_gap_result) + sizeof(mtg_filled);
        const size_t cached = ws.cap[d_zero.slot] + ws.cap[d_raw.slot] + ws.cap[d_ilv.slot] + ws.cap[d_head.slot]; /* already ours */
        size_t free_b = 0, total_b = 0;
        /* steady state: every scratch buffer of the workspace already holds a batch of this size at this tier, nothing will be allocated */
        const uint64_t m0 = std::min<uint64_t>(n_todo, 1u << 20);
        const bool fits = ws.cap[d_zero.slot] >= m0 * cfg.zero_stride && ws.cap[d_raw.slot] >= m0 * cfg.raw_stride + 64 && ws.cap[d_ilv.slot] >= ((m0 + 63) / 64) * cfg.ilv_stride &&
                          ws.cap[d_head.slot] >= ((m0 + 63) / 64) * cfg.hd_stride;
        if (!fits) HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        size_t chunk = fits ? (size_t)m0 : (size_t)(((double)free_b * 0.6 + (double)cached) / (double)per_gap);
        const size_t env_chunk = (size_t)tune::i(tune::T_MAX_CHUNK, 0); /* test hook: several launches per batch */
        if (env_chunk && chunk > env_chunk) chunk = env_chunk;
        if (chunk > n_todo) chunk = n_todo;
        if (chunk > (1u << 20)) chunk = 1u << 20;
        if (chunk == 0) { set_error("not enough device memory for one gap at scratch tier %d (%llu bytes)", tier, (unsigned long long)per_gap); rc = MTG_ERR_NOMEM; break; }
        HIP_TRY(d_zero.alloc(chunk * cfg.zero_stride));
        /* zeroed once: every gap restores what it touched (stage_a_gap), so the region stays clean from launch to launch */
        if (d_zero.fresh) HIP_TRY(hipMemsetAsync(d_zero.p, 0, ws.cap[d_zero.slot], stream));
        HIP_TRY(d_raw.alloc(chunk * cfg.raw_stride + 64));
        HIP_TRY(d_head.alloc(((chunk + 63) / 64) * cfg.hd_stride));
        HIP_TRY(d_ilv.alloc(((chunk + 63) / 64) * cfg.ilv_stride));
        HIP_TRY(d_out.alloc(chunk * sizeof(GapOut)));
        HIP_TRY(d_rec.alloc(chunk * sizeof(SlotRec)));
        HIP_TRY(d_ids.alloc(chunk * 4));
        HIP_TRY(d_blocks.alloc(((chunk + SCAN_SL - 1) / SCAN_SL + 1) * sizeof(ScanBlock)));
        /* ONE block [records | filled | sequence arena] when the host's result object is laid out the same way (ResultSink::combo): the results
         * of a whole-batch launch then cross the link as one copy instead of three (round 4: 5.3 MB of a 12 500-gap batch in three copies came
         * over at 42 GB/s, the 40 MB of a 100 000-gap batch at 54) */
        const bool combo = want_records && sink.combo != nullptr && !sink.seq_dev && !sink.seq_on_device && !sink.seq_stays_in_workspace && !sink.wire_dev;
        if (!combo) {
            HIP_TRY(d_res.alloc(chunk * sizeof(mtg_gap_result)));
            HIP_TRY(d_fil.alloc(chunk * sizeof(mtg_filled)));
        }
        HIP_TRY(d_rlist.alloc(chunk * 4));
        HIP_TRY(d_glist.alloc(chunk * 4));
        HIP_TRY(d_park.alloc((size_t)chunk * 4 * PARK_LISTS + sizeof(ParkCtl)));
        if (combo) HIP_TRY(d_combo.grow_keeping(sink.combo_off_seq + std::max<size_t>(sink.seq_cap, 64) + SEQ_SLACK, sink.combo_off_seq + (size_t)sink.seq_used));
        else HIP_TRY(d_seq.grow_keeping(sink.seq_dev ? 64 : std::max<size_t>(sink.seq_cap, 64) + SEQ_SLACK, sink.seq_dev ? 0 : (size_t)sink.seq_used)); /* a caller's device buffer is written in place; what an earlier tier left stays */
        auto p_res = [&]() { return combo ? (mtg_gap_result*)d_combo.p : d_res.as<mtg_gap_result>(); };
        auto p_fil = [&]() { return combo ? (mtg_filled*)((char*)d_combo.p + sink.combo_off_fil) : d_fil.as<mtg_filled>(); };
        auto p_seq = [&]() { return combo ? (char*)d_combo.p + sink.combo_off_seq : d_seq.as<char>(); };
        HIP_TRY(d_ext.alloc(std::max<size_t>(sink.ext_cap, 64)));
        /* dense words / metadata: only the gaps that need the host bring their contigs back; sized by the last need, grown on demand below */
        HIP_TRY(d_dw.alloc(std::max<size_t>(ws.cap[d_dw.slot], 1 << 20)));
        HIP_TRY(d_dm.alloc(std::max<size_t>(ws.cap[d_dm.slot], 1 << 16)));
        tick("workspace alloc");
        std::vector<uint32_t> retry;
        for (size_t base = 0; base < n_todo; base += chunk) {
            const uint32_t m = (uint32_t)std::min(chunk, n_todo - base);
            const bool identity = todo.empty() && base == 0 && m == n; /* the whole batch in one launch: slot = gap */
            const uint32_t* ids = nullptr;
            std::vector<uint32_t> seq_ids;
            const uint32_t* host_ids = nullptr; /* gap of every slot of this launch (nullptr: identity) */
            if (!identity) {
                t0 = now_ms();
                if (todo.empty()) { seq_ids.resize(m); for (uint32_t s = 0; s < m; s++) seq_ids[s] = (uint32_t)(base + s); host_ids = seq_ids.data(); }
                else host_ids = todo.data() + base;
                HIP_TRY(hipMemcpyAsync(d_ids.p, host_ids, (size_t)m * 4, hipMemcpyHostToDevice, stream)); /* host_ids outlives the launch */
                ids = d_ids.as<uint32_t>();
                st.h2d_ms += now_ms() - t0;
                sink.in_gap_order = false;
            }
            launches++;
            /* The traversal reads index shape and configuration from the module's constants, of which there is one set per workspace
             * number: the batches of an index never share a set, batches of different indexes may, so a set is locked until the traversal
             * that reads it has finished (below, after the host work that runs meanwhile). */
            /* the constants exist once per device: the lock is the (device, set)'s, so that the tool's host threads -- one or more per
             * device, each on its own replica -- only ever wait for a batch of another index on their own device */
            static std::mutex traversal_mtx[CopyTurn::MAX_DEV][TRAVERSAL_SETS];
            const uint32_t cset = (uint32_t)(&ws - idx->ws);
            std::unique_lock<std::mutex> traversal_lock(traversal_mtx[(unsigned)idx->device % CopyTurn::MAX_DEV][cset]);
            {   /* the set keeps what it was last given: batch after batch of one index at one tier upload nothing (two runtime calls per batch in round 4) */
                struct Held { Index ix; FillCfg cfg; bool valid = false; };
                static Held held[CopyTurn::MAX_DEV][TRAVERSAL_SETS];
                Held& h = held[(unsigned)idx->device % CopyTurn::MAX_DEV][cset];
                if (!h.valid || memcmp(&h.ix, &idx->dev, sizeof(Index)) != 0 || memcmp(&h.cfg, &cfg, sizeof(FillCfg)) != 0) {
                    h.valid = false;
                    HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_ix), &idx->dev, sizeof(Index), cset * sizeof(Index), hipMemcpyHostToDevice, stream));
                    HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_cfg), &cfg, sizeof(FillCfg), cset * sizeof(FillCfg), hipMemcpyHostToDevice, stream));
                    memcpy(&h.ix, &idx->dev, sizeof(Index));
                    memcpy(&h.cfg, &cfg, sizeof(FillCfg));
                    h.valid = true;
                }
            }
            /* lanes per parked gap: 1, 8, 16 or 64 (anything else, a typo included, is 16) */
            const bool finish_g_set = tune::is_set(tune::T_FINISH_G);
            const int finish_g = [] { const int v = (int)tune::i(tune::T_FINISH_G, 16); return (v == 1 || v == 8 || v == 16 || v == 64) ? v : 16; }();
            const int env_rounds = (int)tune::i(tune::T_ROUNDS, -1);
            /* Rounds: when many gaps park (bubbles all over the data), their branching nodes are answered by the bubble kernels and the walks go
             * on in the walk kernel, one gap per lane again -- walking is cheap at full width, only the bubbles need a group -- for a few
             * rounds; what is still parked then (and everything, when few gaps park) is finished by groups in k_finish.  The host does not
             * know the counts when it queues the kernels: the number of rounds follows the share of gaps the previous launch of this workspace parked
             * (a workspace without a launch yet: the latest figure of any workspace of the index). */
            const uint32_t park_share = ws.park_share != ~0u ? ws.park_share : idx->park_share_any.load(std::memory_order_relaxed);
            const uint32_t park_hint = (uint32_t)(((uint64_t)park_share * m) >> 16); /* gaps this launch is expected to park */
            int rounds = env_rounds >= 0 ? env_rounds : (park_share > 32768u ? 6 : 0); /* measured: with an eighth of the gaps parked the finishing kernel alone is faster, with all of them six rounds are */
            /* The light walk kernel (k_walk: simple paths only, every branching node parks) where hardly any walk of the previous launch met a
             * branching node -- a haploid donor: 2 of 100 000 -- and the full one (k_stage_a: SNP / tip / indel / merge forms in the walking lane)
             * everywhere else; LIGHT_WALK=1 / 0 forces either (tests run the bubble cases under both). */
            const uint32_t branch_share = ws.branch_share != ~0u ? ws.branch_share : idx->branch_share_any.load(std::memory_order_relaxed);
            const int env_light = (int)tune::i(tune::T_LIGHT_WALK, -1);
            const bool light = env_light >= 0 ? env_light != 0 : (branch_share != ~0u && branch_share < 512u && env_rounds < 0);
            const uint32_t first_hint = light && branch_share != ~0u ? (uint32_t)(((uint64_t)branch_share * m) >> 16) : park_hint; /* gaps the first walk launch is expected to park */
            if (rounds > (PARK_LISTS - 5) / 2) rounds = (PARK_LISTS - 5) / 2;
            ParkCtl* const park = d_park.as<ParkCtl>();
            HIP_TRY(hipMemsetAsync(d_park.p, 0, sizeof(ParkCtl), stream)); /* the work lists of the launch: parked gaps, gaps with commands to execute */
            HIP_TRY(hipEventRecord(ev0, stream)); /* ev0 .. evf = the walk kernel's first launch, evf .. ev1 = rounds and the finishing kernel */
            const bool no_lean = tune::on(tune::T_NO_LEAN); /* A/B and test hook: every contig is materialised */
            const bool skip_finish = tune::on(tune::T_DEBUG_SKIP_FINISH); /* diagnostics: the parked gaps stay parked (and fail as overflowing gaps) */
            if (skip_finish) rounds = 0; /* (with rounds a resumed walk would list a gap its first walk has already listed: the lists hold one entry per gap -- advisor, round 5) */
            LeanIn li;
            li.tle = d_tle; li.tbad = d_tbad; li.toff = d_toff; li.tcnt = d_tcnt; li.fast_ok = d_fok;
            li.lean_allowed = (in.want_all_contigs || no_lean || !cfg.cmd_cap) ? 0u : 1u;
            li.decide_parked = skip_finish ? 1u : 0u;
            {
                if (light)
                    hipLaunchKernelGGL(k_walk, dim3((m + 63) / 64), dim3(64), 0, stream, d_zero.as<uint8_t>(), d_raw.as<uint8_t>(), d_ilv.as<uint8_t>(), d_head.as<uint8_t>(), d_src, d_rw, d_roff,
                                       d_rlen, d_r0, ids, d_out.as<GapOut>(), m, cset, park, m, li);
                else
                    hipLaunchKernelGGL(k_stage_a, dim3((m + 63) / 64), dim3(64), 0, stream, d_zero.as<uint8_t>(), d_raw.as<uint8_t>(), d_ilv.as<uint8_t>(), d_head.as<uint8_t>(), d_src, d_rw, d_roff,
                                       d_rlen, d_r0, ids, d_out.as<GapOut>(), m, cset, park, m, -1, 0u, li);
                HIP_TRY(mark(evf));
                /* the bubbles of a round by one lane each: every lane of a wave is in the bubble code at the same time, and with the narrow bubbles of
                 * heterozygous data that keeps more of them in flight than a group of lanes per bubble does (round 3 measured the LDS group form
                 * k_bubble<G> here: 20-22 against 25 M/s on the indel set, 51 against 63 on tips; removed in round 5) */
                for (int r = 0; r < rounds; r++) {
                    const uint32_t lin = 2u * (uint32_t)r;
                    hipLaunchKernelGGL(k_bubble_classic, dim3((m + 63) / 64), dim3(64), 0, stream, d_zero.as<uint8_t>(), d_raw.as<uint8_t>(), d_ilv.as<uint8_t>(), d_head.as<uint8_t>(), cset, park, m, lin);
                    hipLaunchKernelGGL(k_stage_a, dim3((m + 63) / 64), dim3(64), 0, stream, d_zero.as<uint8_t>(), d_raw.as<uint8_t>(), d_ilv.as<uint8_t>(), d_head.as<uint8_t>(), d_src, d_rw, d_roff,
                                       d_rlen, d_r0, ids, d_out.as<GapOut>(), m, cset, park, m, (int)lin, lin + 2, li);
                }
                const uint32_t lfin = 2u * (uint32_t)rounds;
                /* how the tail is finished: a group of lanes per parked gap, bubbles from LDS.  One lane per gap (MTG_FINISH_G=1) was measured and is
                 * slower at every size: 0.11 against 0.10 ms for the haploid set's 1-5 gaps, 0.83 against 0.35 for 108 (heterozygous SNPs), 1.12
                 * against 0.56 for 12 000 (tips). */
                /* lanes per parked gap in the finishing kernel: a whole wave while few gaps are parked (their chains are what the kernel takes:
                 * 0.17 against 0.32 ms for the 108 gaps of the heterozygous set), 16 when there are many (12 000 on the tips set: 0.46 against 0.63) */
                const int finish_wave_below = (int)tune::i(tune::T_FINISH_WAVE_BELOW, 2048);
                const int fin_g = finish_g_set ? finish_g : (rounds == 0 && first_hint < (uint32_t)finish_wave_below ? 64 : 16);
                const bool lane_finish = finish_g_set && finish_g == 1;
                /* The grid.  The host does not know how many gaps are parked when it queues the kernel, and 100 000 groups that read one
                 * scalar and leave cost 63 us (round 3: 13 % of a haploid batch's kernels, for 5 parked gaps).  So the groups take the first
                 * `fin_entries` entries of the list -- four times what the previous launch of this workspace parked, plus 256 -- and the
                 * entries beyond, if a launch parks more than that after all, are walked one gap per lane (k_finish_lane, a grid of
                 * (m - fin_entries) / 64 workgroups: slower per gap, but only for the launch that outgrew the hint; the next one follows). */
                const uint32_t per_wg = 64u / (uint32_t)fin_g;
                /* whole workgroups: k_finish<G> is bounded by the list's count only, so the entries it takes and those k_finish_lane starts from must
                 * meet at a multiple of per_wg (the advisor's round-4 finding: G = 8 with an odd hint walked four entries twice) */
                const uint32_t fin_entries = (lane_finish || skip_finish) ? 0u : (uint32_t)std::min<uint64_t>(m, ((4ull * first_hint + 256ull + per_wg - 1) / per_wg) * per_wg);
                const uint32_t nwg = (fin_entries + per_wg - 1) / per_wg;
                if (!skip_finish && nwg) switch (fin_g) {
                    case 8: hipLaunchKernelGGL(k_finish<8>, dim3(nwg), dim3(64), 0, stream, d_zero.as<uint8_t>(), d_raw.as<uint8_t>(), d_ilv.as<uint8_t>(), d_head.as<uint8_t>(), d_rw, d_roff, d_rlen, d_r0, ids, d_out.as<GapOut>(), cset, park, m, lfin, li); break;
                    case 64: hipLaunchKernelGGL(k_finish<64>, dim3(nwg), dim3(64), 0, stream, d_zero.as<uint8_t>(), d_raw.as<uint8_t>(), d_ilv.as<uint8_t>(), d_head.as<uint8_t>(), d_rw, d_roff, d_rlen, d_r0, ids, d_out.as<GapOut>(), cset, park, m, lfin, li); break;
                    default: hipLaunchKernelGGL(k_finish<16>, dim3(nwg), dim3(64), 0, stream, d_zero.as<uint8_t>(), d_raw.as<uint8_t>(), d_ilv.as<uint8_t>(), d_head.as<uint8_t>(), d_rw, d_roff, d_rlen, d_r0, ids, d_out.as<GapOut>(), cset, park, m, lfin, li); break;
                }
                if (!skip_finish && fin_entries < m)
                    hipLaunchKernelGGL(k_finish_lane, dim3((m - fin_entries + 63) / 64), dim3(64), 0, stream, d_zero.as<uint8_t>(), d_raw.as<uint8_t>(), d_ilv.as<uint8_t>(), d_head.as<uint8_t>(), d_rw, d_roff, d_rlen, d_r0, ids, d_out.as<GapOut>(), cset, park, m, lfin, fin_entries, li);
#ifdef MTG_BUBBLE_TIMING
                {
                    static ParkCtl hc; static int shown = 0;
                    HIP_TRY(hipStreamSynchronize(stream));
                    HIP_TRY(hipMemcpy(&hc, d_park.p, sizeof hc, hipMemcpyDeviceToHost));
                    if (shown++ < 3) {
                        fprintf(stderr, "[timing] lists:"); for (int i = 0; i < PARK_LISTS; i++) fprintf(stderr, " %u", hc.count[i]); fprintf(stderr, "\n");
                        const char* nm[4] = {"bubble lane", "bubble wave", "resumed walk lane", "resumed walk wave"};
                        const uint32_t* hh[4] = {hc.hist_lane, hc.hist_wave, hc.hist_walk_lane, hc.hist_walk_wave};
                        for (int j = 0; j < 4; j++) { fprintf(stderr, "[timing] %s, log2(10 ns ticks) bins:", nm[j]); for (int i = 0; i < 24; i++) fprintf(stderr, " %u", hh[j][i]); fprintf(stderr, "\n"); }
                    }
                }
#endif
            }
            HIP_TRY(mark(ev1)); /* the end of the walks */
            HIP_TRY(mark(evl0));
            HIP_TRY(hipGetLastError());
            /* evl0 .. evc: the long runs of the contigs, which the traversal only noted down */
            HIP_TRY(mark(evl)); /* (evl0 .. evl was k_lean: the walking lanes decide now) evl .. evc: k_copy */
            hipLaunchKernelGGL(k_copy, dim3((m + 3) / 4), dim3(256), 0, stream, idx->dev, cfg, d_raw.as<uint8_t>(), d_head.as<uint8_t>(), d_out.as<GapOut>(), park, m, (uint32_t)COPY_LIST);
            HIP_TRY(mark(evc));
            const uint32_t nblocks = (m + SCAN_SL - 1) / SCAN_SL;
            /* the lean gaps eight per wave; the others (POST_LIST) a wave each: a grid of four times what the previous launch of this workspace
             * listed, plus 1024 (a workspace without a launch yet: one per gap), the kernel's loop takes the rest */
            const uint32_t general_hint = ws.post_general == ~0u ? m : (uint32_t)std::min<uint64_t>(m, 4ull * ws.post_general + 1024ull);
            /* The general form is the latency of a few long gaps (30 us for the one or two of a haploid batch), the lean form the throughput of all
             * the others.  Both on the batch's one stream: the general form next to the lean one on a second stream, and the finishing kernel there as
             * well, were built and measured in round 4 (9 and 25 us shorter for one batch alone, no faster with six in flight) and removed in round 5. */
            hipLaunchKernelGGL(k_post, dim3(general_hint), dim3(64), 0, stream, idx->dev, cfg, d_raw.as<uint8_t>(), d_head.as<uint8_t>(), d_out.as<GapOut>(), ids, d_tle, d_tbad, d_toff, d_tcnt, d_mis, d_fok,
                               in.want_all_contigs ? 1u : 0u, d_rec.as<SlotRec>(), m, park, (const uint32_t*)d_pi_head, (const uint32_t*)d_pi_next, pi_mask);
            hipLaunchKernelGGL(k_post_lean, dim3((m + 64 / POST_LEAN_G - 1) / (64 / POST_LEAN_G)), dim3(64), 0, stream, idx->dev, cfg, d_raw.as<uint8_t>(), d_head.as<uint8_t>(), d_out.as<GapOut>(), in.want_all_contigs ? 1u : 0u, d_rec.as<SlotRec>(), m);
            /* the dense arrays hold one launch at a time; the two arenas the whole batch */
            const ScanBegin sbegin{{0, 0, arena_used[0], arena_used[1]}};
            hipLaunchKernelGGL(k_scan1, dim3(nblocks), dim3(SCAN_SL), 0, stream, d_rec.as<SlotRec>(), m, d_blocks.as<ScanBlock>());
            hipLaunchKernelGGL(k_scan2, dim3(1), dim3(256), 0, stream, d_blocks.as<ScanBlock>(), nblocks, sbegin, d_tot.as<PartTot>(), h_tot, park);
            HIP_TRY(hipGetLastError());
            EmitDev D;
            EmitHost H;
            auto emit = [&]() -> int {
                D.seq = sink.seq_dev ? sink.seq_dev : p_seq(); D.ext = d_ext.as<char>();
                D.seq_cap = sink.seq_cap; D.ext_cap = sink.ext_cap;
                D.res = p_res(); D.fil = p_fil();
                D.dense_words = d_dw.as<uint64_t>(); D.dense_meta = d_dm.as<uint32_t>();
                D.dense_cap_words = ws.cap[d_dw.slot] / 8; D.dense_cap_contigs = ws.cap[d_dm.slot] / 20;
                const bool want_wire = sink.wire_dev != nullptr && identity && tier == 0;
                D.wire = want_wire ? (uint8_t*)sink.wire_dev : nullptr; D.wire_cap = sink.wire_cap; D.wire_tag = sink.wire_tag;
                D.tot = d_tot.as<PartTot>(); D.wire_gaps = m;
                if (want_wire) HIP_TRY(hipMemsetAsync(sink.wire_dev, 0, sizeof(mtg_wire_header), stream)); /* no header, no payload (k_wire_sum) */
                H.seq = sink.seq; H.ext = sink.ext; H.fil = sink.fil;
                if (!want_wire)
                    hipLaunchKernelGGL(k_emit_lean, dim3((m + 64 / EMIT_LEAN_G - 1) / (64 / EMIT_LEAN_G)), dim3(64), 0, stream, idx->dev.us, cfg, d_raw.as<uint8_t>(), d_head.as<uint8_t>(), d_rec.as<SlotRec>(), d_blocks.as<ScanBlock>(),
                                       ids, d_flags, k, D, H, m);
                hipLaunchKernelGGL(k_emit, dim3(want_wire ? m : general_hint), dim3(64), 0, stream, idx->dev.us, cfg, d_raw.as<uint8_t>(), d_head.as<uint8_t>(), d_rec.as<SlotRec>(), d_blocks.as<ScanBlock>(), ids, d_flags, k, D, H,
                                   d_rlist.as<uint32_t>(), d_glist.as<uint32_t>(), m, park, want_wire ? 0u : 1u);
                if (want_wire) hipLaunchKernelGGL(k_wire_sum, dim3(256 * 4), dim3(256), 0, stream, (uint8_t*)sink.wire_dev, sink.wire_cap);
                HIP_TRY(hipGetLastError());
                return MTG_OK;
            };
            /* the results are written right away, on the assumption that the arenas are large enough (they are, from the second batch of a
             * shape on): the totals tell */
            HIP_TRY(mark(eve));
            if (int erc = emit()) return erc;
            HIP_TRY(hipEventRecord(ev2, stream));
            tick("host prep+launch");
            if (while_busy && !busy_done) { busy_done = true; (*while_busy)(); tick("host work during kernels"); }
            HIP_TRY(hipEventSynchronize(ev2));
            traversal_lock.unlock();
            tick("totals ready");
            t0 = now_ms();
            PartTot tot = *h_tot;
            /* an array that was too small: make it larger and write the launch's results again (its scratch is still in place) */
            const uint64_t need_w = tot.end[0] * 8 + 64, need_m = tot.end[1] * 20 + 64;
            bool again = false;
            if (want_records && tot.end[2] > sink.seq_cap) {
                /* the block may move: what earlier launches of the batch left in it is kept, the records that point there follow */
                const uintptr_t old = (uintptr_t)sink.seq, old_end = old + sink.seq_cap;
                const uintptr_t old_fil = (uintptr_t)sink.fil, old_fil_end = old_fil + n * sizeof(mtg_filled);
                if (!sink.grow_seq || !sink.grow_seq((size_t)tot.end[2], (size_t)tot.begin[2])) { set_error("sequence buffer too small: %llu bytes needed", (unsigned long long)tot.end[2]); return MTG_ERR_ARG; }
                if (launches > 1 && (uintptr_t)sink.seq != old)
                    for (size_t i = 0; i < n; i++) { const uintptr_t q = (uintptr_t)sink.fil[i].seq; if (q >= old && q < old_end) sink.fil[i].seq = sink.seq + (q - old); }
                /* the arena is part of one block with the records (ResultSink::combo): the filled records moved with it */
                if (launches > 1 && (uintptr_t)sink.fil != old_fil)
                    for (size_t i = 0; i < n; i++) { const uintptr_t q = (uintptr_t)sink.res[i].filled; if (q >= old_fil && q < old_fil_end) sink.res[i].filled = (const mtg_filled*)((const char*)sink.fil + (q - old_fil)); }
                again = true;
            }
            if (want_records && tot.end[3] > sink.ext_cap) {
                const uintptr_t old = (uintptr_t)sink.ext, old_end = old + sink.ext_cap;
                if (!sink.grow_ext || !sink.grow_ext((size_t)tot.end[3], (size_t)tot.begin[3])) { set_error("extension buffer too small: %llu bytes needed", (unsigned long long)tot.end[3]); return MTG_ERR_NOMEM; }
                if (launches > 1 && (uintptr_t)sink.ext != old)
                    for (size_t i = 0; i < n; i++) { const uintptr_t q = (uintptr_t)sink.res[i].extension; if (q >= old && q < old_end) sink.res[i].extension = sink.ext + (q - old); }
                again = true;
            }
            if (need_w > ws.cap[d_dw.slot] || need_m > ws.cap[d_dm.slot]) again = true;
            if (again) {
                /* the dense arrays hold this launch only (offsets relative to the batch: the launch's part is copied from begin[]) */
                HIP_TRY(hipStreamSynchronize(stream));
                /* what earlier launches of the batch left in the arena stays: when the text is formatted on the device the arena's only copy is this one
                 * (round 4: a batch of several launches lost the sequences of all but its last launch here -- the device formatter then wrote
                 * "_len_0" records; found by running the GPU tests under MAX_CHUNK) */
                if (combo) HIP_TRY(d_combo.grow_keeping(sink.combo_off_seq + std::max<size_t>(sink.seq_cap, 64) + SEQ_SLACK, sink.combo_off_seq + (size_t)tot.begin[2]));
                else if (!sink.seq_dev) HIP_TRY(d_seq.grow_keeping(std::max<size_t>(sink.seq_cap, 64) + SEQ_SLACK, (size_t)tot.begin[2]));
                HIP_TRY(d_ext.alloc(std::max<size_t>(sink.ext_cap, 64)));
                HIP_TRY(d_dw.alloc(need_w));
                HIP_TRY(d_dm.alloc(need_m));
                /* k_emit made the records' offsets absolute: run the layout again from the launch's begin */
                hipLaunchKernelGGL(k_scan1, dim3(nblocks), dim3(SCAN_SL), 0, stream, d_rec.as<SlotRec>(), m, d_blocks.as<ScanBlock>());
                hipLaunchKernelGGL(k_scan2, dim3(1), dim3(256), 0, stream, d_blocks.as<ScanBlock>(), nblocks, sbegin, d_tot.as<PartTot>(), (PartTot*)nullptr, park);
                if (int erc = emit()) return erc;
            }
            /* bring the launch's results to the host.  Result copies of six batches at once share the link worse than two or three do
             * (scripts/pcie_d2h.py: 57 GB/s with two streams copying, 47-52 with six), so the batches of a device take turns */
            CopyTurn copy_turn(idx->device);
            std::vector<mtg_gap_result> tmp_res;
            std::vector<mtg_filled> tmp_fil;
            const bool one_copy = combo && identity && tot.begin[2] == 0; /* records, filled records and sequences of the launch: one block on both sides */
            if (want_records) {
                if (one_copy) HIP_TRY(hipMemcpyAsync(sink.combo, d_combo.p, sink.combo_off_seq + (size_t)tot.end[2], hipMemcpyDeviceToHost, stream));
                else if (identity) {
                    HIP_TRY(hipMemcpyAsync(sink.res, p_res(), (size_t)m * sizeof(mtg_gap_result), hipMemcpyDeviceToHost, stream));
                    HIP_TRY(hipMemcpyAsync(sink.fil, p_fil(), (size_t)m * sizeof(mtg_filled), hipMemcpyDeviceToHost, stream));
                } else {
                    tmp_res.resize(m);
                    tmp_fil.resize(m);
                    HIP_TRY(hipMemcpyAsync(tmp_res.data(), p_res(), (size_t)m * sizeof(mtg_gap_result), hipMemcpyDeviceToHost, stream));
                    HIP_TRY(hipMemcpyAsync(tmp_fil.data(), p_fil(), (size_t)m * sizeof(mtg_filled), hipMemcpyDeviceToHost, stream));
                }
                /* a batch that left in relocatable form has its sequences in the payload's sequence section */
                const WireLayout wl = wire_layout(m, tot.n_filled, tot.end[2], tot.end[3]);
                const bool wired = sink.wire_dev != nullptr && identity && tier == 0 && wl.total <= sink.wire_cap && tot.n_retry == 0 && tot.n_general == 0;
                if (sink.wire_dev && identity && tier == 0) { sink.wire_ok = wired; sink.wire_bytes = wired ? wl.total : 0; }
                const char* seq_src = wired ? (const char*)sink.wire_dev + wl.o_seq : (sink.seq_dev ? sink.seq_dev : p_seq());
                if (!one_copy && !sink.seq_on_device && !sink.seq_stays_in_workspace && tot.end[2] > tot.begin[2]) HIP_TRY(hipMemcpyAsync(sink.seq + tot.begin[2], seq_src + tot.begin[2], tot.end[2] - tot.begin[2], hipMemcpyDeviceToHost, stream));
                if (tot.end[3] > tot.begin[3]) HIP_TRY(hipMemcpyAsync(sink.ext + tot.begin[3], d_ext.as<char>() + tot.begin[3], tot.end[3] - tot.begin[3], hipMemcpyDeviceToHost, stream));
            }
            std::vector<uint32_t> rlist(tot.n_retry), glist(tot.n_general);
            /* The multi-contig gaps on the DEVICE (round 5; HOST_GENERAL of the tuning table: the host's path for all of them, as until round 4): k_paths
             * and k_general over the list k_emit made, queued before the result copies; what comes back is a header per gap, the solutions and their
             * ASCII -- not every record of the launch, the gaps' contigs and 16 KB of paths per gap (540 MB of a 100 000-gap indel batch). */
            const bool dev_general = tot.n_general > 0 && !host_paths && !in.want_all_contigs && !tune::on(tune::T_HOST_GENERAL) && !tl_host_general;
            GenCtl* h_gctl = nullptr;
            GenDev GD{};
            if (dev_general) {
                const uint64_t ng = tot.n_general;
                HIP_TRY(d_paths.alloc(ng * (size_t)PATHS_WORDS * 4));
                HIP_TRY(d_ggaps.alloc(ng * sizeof(GenGap) + sizeof(GenCtl)));
                GD.cap_sols = 4 * ng + 1024; GD.cap_ascii = ng * 1280 + (1u << 20); GD.cap_tmp = ng * 48 + (1u << 17); GD.cap_bnd = 1u << 22;
                HIP_TRY(d_gsols.alloc(GD.cap_sols * sizeof(GenSol)));
                HIP_TRY(d_gascii.alloc(GD.cap_ascii + 64));
                HIP_TRY(d_gtmp.alloc(GD.cap_tmp * 8));
                HIP_TRY(d_gbnd.alloc(GD.cap_bnd * sizeof(NwCell)));
                GD.ctl = (GenCtl*)d_ggaps.p;
                GD.gaps = (GenGap*)((char*)d_ggaps.p + sizeof(GenCtl));
                GD.sols = d_gsols.as<GenSol>(); GD.ascii = d_gascii.as<char>(); GD.tmp = d_gtmp.as<uint64_t>(); GD.bnd = d_gbnd.as<NwCell>();
                HIP_TRY(hipMemsetAsync(GD.ctl, 0, sizeof(GenCtl), stream));
                hipLaunchKernelGGL(k_paths, dim3((unsigned)ng), dim3(64), 0, stream, cfg, d_raw.as<uint8_t>(), d_head.as<uint8_t>(), d_out.as<GapOut>(), d_glist.as<uint32_t>(), k, d_paths.as<uint32_t>(), (uint32_t)ng);
                hipLaunchKernelGGL(k_general, dim3((unsigned)ng), dim3(64), 0, stream, idx->dev, cfg, d_raw.as<uint8_t>(), d_head.as<uint8_t>(), d_out.as<GapOut>(), d_glist.as<uint32_t>(), ids, d_paths.as<uint32_t>(),
                                   d_tcnt, d_fok, d_src, d_flags, k, GD, (uint32_t)ng);
                HIP_TRY(hipGetLastError());
                h_gctl = (GenCtl*)staging_host(&ws, Workspace::NHOST - 2, sizeof(GenCtl) + ng * sizeof(GenGap) + 64);
                if (!h_gctl) { set_error("no page-locked memory for the multi-contig gaps of a launch"); return MTG_ERR_NOMEM; }
                HIP_TRY(hipMemcpyAsync(h_gctl, d_ggaps.p, sizeof(GenCtl) + ng * sizeof(GenGap), hipMemcpyDeviceToHost, stream));
            }
            HostChunk* hc = nullptr;
            SlotRec* h_rec = nullptr;
            uint64_t* h_w = nullptr;
            uint32_t* h_m = nullptr;
            const uint64_t tw = tot.end[0], tc = tot.end[1];
            if (tot.n_retry) HIP_TRY(hipMemcpyAsync(rlist.data(), d_rlist.p, (size_t)tot.n_retry * 4, hipMemcpyDeviceToHost, stream));
            /* the contigs of the gaps the host works on (every record of the launch, the dense words and metadata) */
            auto contigs_to_host = [&]() -> int {
                hc->carve(m, tw, tc, h_rec, h_w, h_m);
                HIP_TRY(hipMemcpyAsync(h_rec, d_rec.p, (size_t)m * sizeof(SlotRec), hipMemcpyDeviceToHost, stream));
                if (tw) HIP_TRY(hipMemcpyAsync(h_w, d_dw.p, tw * 8, hipMemcpyDeviceToHost, stream));
                if (tc) HIP_TRY(hipMemcpyAsync(h_m, d_dm.p, tc * 20, hipMemcpyDeviceToHost, stream));
                return MTG_OK;
            };
            if (tot.n_general) {
                special.chunks.emplace_back(new HostChunk());
                hc = special.chunks.back().get();
                HIP_TRY(hipMemcpyAsync(glist.data(), d_glist.p, (size_t)tot.n_general * 4, hipMemcpyDeviceToHost, stream));
                if (!dev_general) { if (int crc = contigs_to_host()) return crc; }
            }
            HIP_TRY(hipEventRecord(ev3, stream));
            HIP_TRY(hipEventSynchronize(ev3));
            copy_turn.release();
            st.d2h_ms += now_ms() - t0;
            tick("results on the host");
            t0 = now_ms();
            st.index_lines += tot.lines; st.contig_nt += tot.contig_nt; st.store_runs += tot.store_runs; st.run_nt += tot.run_nt; st.post_lines += tot.post_lines;
            st.contig_words += tot.contig_words; st.coverage_kmers += tot.cov_kmers; st.dense_words += tw;
            st.copy_words += tot.copy_words; st.copy_cmds += tot.copy_cmds; st.coverage_direct_kmers += tot.cov_direct; st.n_lean_gaps += tot.n_lean;
            if (tier == 0 && identity) ws.post_general = m - (uint32_t)std::min<uint64_t>(tot.n_lean, m);
            st.copy_words_executed += tot.copy_words_exec; st.copy_cmds_executed += tot.copy_cmds_exec; st.post_scanned_words += tot.scan_words;
            sink.seq_used = tot.end[2];
            sink.ext_used = tot.end[3];
            arena_used[0] = tot.end[2];
            arena_used[1] = tot.end[3];
            sink.n_filled += tot.n_filled;
            if (want_records && !identity) /* records of a partial launch: to their gaps (a gap to be re-run gets its record again later) */
                for (uint32_t s2 = 0; s2 < m; s2++) { sink.res[host_ids[s2]] = tmp_res[s2]; sink.fil[host_ids[s2]] = tmp_fil[s2]; }
            for (uint32_t s2 : rlist) retry.push_back(host_ids ? host_ids[s2] : s2);
            if (tot.n_retry) sink.in_gap_order = false;
            if (hc && dev_general) {
                /* what the device made of the launch's multi-contig gaps: solutions and their ASCII for the gaps it finished; the contigs and the paths
                 * of the gaps it left to the host (rare: several targets, a candidate that does not fit, an unknown k-mer) */
                const uint64_t ng = tot.n_general;
                const GenCtl ctl = *h_gctl;
                const GenGap* hg = (const GenGap*)(h_gctl + 1);
                hc->gen_gaps.assign(hg, hg + ng);
                uint64_t n_host = 0;
                for (uint64_t i = 0; i < ng; i++) n_host += hg[i].status != GEN_OK;
                const uint64_t ns = std::min<uint64_t>(ctl.n_sols, GD.cap_sols), na = std::min<uint64_t>(ctl.ascii_bytes, GD.cap_ascii);
                hc->gen_sols.resize(ns);
                hc->gen_ascii.resize(na + 1);
                if (ns) HIP_TRY(hipMemcpyAsync(hc->gen_sols.data(), d_gsols.p, ns * sizeof(GenSol), hipMemcpyDeviceToHost, stream));
                if (na) HIP_TRY(hipMemcpyAsync(hc->gen_ascii.data(), d_gascii.p, na, hipMemcpyDeviceToHost, stream));
                if (host_ids) hc->gap_of.assign(host_ids, host_ids + m);
                const uint32_t chunk_id = (uint32_t)special.chunks.size() - 1;
                for (uint32_t i = 0; i < (uint32_t)ng; i++) special.special.push_back(SpecialGap{host_ids ? host_ids[glist[i]] : glist[i], chunk_id, glist[i], i});
                if (n_host) {
                    if (int crc = contigs_to_host()) return crc;
                    hc->path_of.assign(m, -1);
                    if (2 * n_host > ng) { /* most of them (contig mode: every gap has several targets): the blocks in one copy */
                        hc->paths.resize(ng * (size_t)PATHS_WORDS);
                        HIP_TRY(hipMemcpyAsync(hc->paths.data(), d_paths.p, ng * (size_t)PATHS_WORDS * 4, hipMemcpyDeviceToHost, stream));
                        for (uint32_t i = 0; i < (uint32_t)ng; i++) if (hg[i].status != GEN_OK) hc->path_of[glist[i]] = (int32_t)i;
                    } else {
                        hc->paths.resize(n_host * (size_t)PATHS_WORDS);
                        uint64_t q = 0;
                        for (uint32_t i = 0; i < (uint32_t)ng; i++)
                            if (hg[i].status != GEN_OK) {
                                HIP_TRY(hipMemcpyAsync(hc->paths.data() + q * PATHS_WORDS, d_paths.as<uint32_t>() + (size_t)i * PATHS_WORDS, (size_t)PATHS_WORDS * 4, hipMemcpyDeviceToHost, stream));
                                hc->path_of[glist[i]] = (int32_t)q++;
                            }
                    }
                }
                HIP_TRY(hipStreamSynchronize(stream));
                if (n_host) h_w[tw] = 0;
                st.n_general_host += n_host;
                st.n_general_device += ng - n_host;
            } else if (hc) {
                h_w[tw] = 0;
                if (host_ids) hc->gap_of.assign(host_ids, host_ids + m);
                const uint32_t chunk_id = (uint32_t)special.chunks.size() - 1;
                std::vector<uint32_t> pslots; /* multi-contig gaps: their contig-graph paths, while the launch's scratch is still in place */
                st.n_general_host += glist.size();
                uint32_t grank = 0;
                for (uint32_t s2 : glist) {
                    special.special.push_back(SpecialGap{host_ids ? host_ids[s2] : s2, chunk_id, s2, grank++});
                    if (!host_paths && !in.want_all_contigs && h_rec[s2].p.fast == 0 && h_rec[s2].p.nb_terminal > 0) pslots.push_back(s2);
                }
                if (!pslots.empty()) {
                    HIP_TRY(d_ids.alloc(std::max<size_t>(chunk, pslots.size()) * 4)); /* the traversal is over: its slot map is free */
                    HIP_TRY(hipMemcpyAsync(d_ids.p, pslots.data(), pslots.size() * 4, hipMemcpyHostToDevice, stream));
                    HIP_TRY(d_paths.alloc(pslots.size() * (size_t)PATHS_WORDS * 4));
                    hipLaunchKernelGGL(k_paths, dim3((unsigned)pslots.size()), dim3(64), 0, stream, cfg, d_raw.as<uint8_t>(), d_head.as<uint8_t>(), d_out.as<GapOut>(), d_ids.as<uint32_t>(), k,
                                       d_paths.as<uint32_t>(), (uint32_t)pslots.size());
                    HIP_TRY(hipGetLastError());
                    hc->paths.resize(pslots.size() * (size_t)PATHS_WORDS);
                    HIP_TRY(hipMemcpyAsync(hc->paths.data(), d_paths.p, hc->paths.size() * 4, hipMemcpyDeviceToHost, stream));
                    HIP_TRY(hipStreamSynchronize(stream));
                    hc->path_of.assign(m, -1);
                    for (size_t g2 = 0; g2 < pslots.size(); g2++) hc->path_of[pslots[g2]] = (int32_t)g2;
                }
            }
            st.host_ms += now_ms() - t0;
            {
                const uint32_t np = tot.n_parked;
                st.n_parked_gaps += np;
                st.n_rounds += (uint64_t)rounds;
                st.n_light_walks += light ? 1u : 0u;
                st.n_branching_gaps += tot.n_branching;
                if (tier == 0 && identity && m >= 64) { /* the share of gaps this launch parked: how the workspace's next launch serves its parked gaps */
                    const uint32_t share = (uint32_t)std::min<uint64_t>(((uint64_t)np << 16) / m, 65536u);
                    ws.park_share = share;
                    idx->park_share_any.store(share, std::memory_order_relaxed);
                    const uint32_t bshare = (uint32_t)std::min<uint64_t>(((uint64_t)tot.n_branching << 16) / m, 65536u);
                    ws.branch_share = bshare;
                    idx->branch_share_any.store(bshare, std::memory_order_relaxed);
                }
            }
            { float mss = 0; HIP_TRY(hipEventElapsedTime(&mss, ev0, ev2)); st.device_span_ms += mss; }
            if (ktimers) {
                float ms = 0, ms2 = 0, ms3 = 0, msc = 0, msf = 0, msl = 0;
                HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
                HIP_TRY(hipEventElapsedTime(&msf, evf, ev1));
                HIP_TRY(hipEventElapsedTime(&msc, evl0, evc)); /* k_copy (k_lean's share is the walk kernels' now) */
                HIP_TRY(hipEventElapsedTime(&msl, evl0, evl));
                HIP_TRY(hipEventElapsedTime(&ms2, evc, eve));
                HIP_TRY(hipEventElapsedTime(&ms3, eve, ev2));
                st.kernel_ms += ms; st.finish_kernel_ms += msf; st.lean_kernel_ms += msl; st.copy_kernel_ms += msc; st.post_kernel_ms += ms2; st.emit_kernel_ms += ms3;
            }
            st.seq_bytes += tot.end[2] - tot.begin[2];
            st.n_launches++;
        }
        if (tier > 0) st.n_retried_gaps += n_todo;
        todo.swap(retry);
        n_todo = todo.size();
    }
#ifdef MTG_FINISH_DEBUG
    {
        unsigned int hd[64];
        if (hipMemcpyFromSymbol(hd, HIP_SYMBOL(mtg::g_dbg), sizeof hd) == hipSuccess) {
            for (int i = 0; i < 64; i++) if (hd[i]) fprintf(stderr, "  [finish debug] guard %d tripped %u times\n", i, hd[i]);
            unsigned int z[64] = {0};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(mtg::g_dbg), z, sizeof z);
        }
    }
#endif
#ifdef MTG_STAMPS
    {
        unsigned long long hs[16];
        if (hipMemcpyFromSymbol(hs, HIP_SYMBOL(mtg::g_stamps), sizeof hs) == hipSuccess && hs[15])
            fprintf(stderr, "  [stamps] lanes %llu  avg cycles/lane: W %.0f (long steps %.0f, bucket reads + run set-up %.0f) B %.0f | find_end %.0f dfs %.0f validate %.0f mark_inv %.0f | snp_fast %.0f (in-branching checks of find_end / alignment %.0f) consume %.0f"
                            " | per lane: general bubbles %.3f, snp_fast answers %.3f (bulk form %.3f, with alignment %.3f)\n", hs[15],
                    (double)hs[0] / hs[15], (double)hs[8] / hs[15], (double)hs[9] / hs[15], (double)hs[1] / hs[15], (double)hs[2] / hs[15], (double)hs[3] / hs[15], (double)hs[4] / hs[15], (double)hs[5] / hs[15],
                    (double)hs[6] / hs[15], (double)hs[12] / hs[15], (double)hs[7] / hs[15], (double)hs[10] / hs[15], (double)hs[14] / hs[15], (double)hs[13] / hs[15], (double)hs[11] / hs[15]);
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(mtg::g_stamps), z, sizeof z);
        unsigned long long ff[16];
        if (hipMemcpyFromSymbol(ff, HIP_SYMBOL(mtg::g_forms), sizeof ff) == hipSuccess && ff[15]) {
            const char* nm[7] = {"merge_fast", "marked-successor test", "first reads of both branches", "tip_fast", "indel_bulk", "snp_bulk", "step-by-step loop"};
            fprintf(stderr, "  [stamps] fast forms per lane (%llu lanes):", ff[15]);
            for (int i = 0; i < 7; i++) fprintf(stderr, " %s %.2f calls x %.0f ticks;", nm[i], (double)ff[2 * i + 1] / ff[15], ff[2 * i + 1] ? (double)ff[2 * i] / ff[2 * i + 1] : 0.0);
            fprintf(stderr, " loop steps per lane %.2f\n", (double)ff[14] / ff[15]);
        }
        (void)hipMemcpyToSymbol(HIP_SYMBOL(mtg::g_forms), z, sizeof z);
        unsigned long long fe[16];
        if (hipMemcpyFromSymbol(fe, HIP_SYMBOL(mtg::g_fe), sizeof fe) == hipSuccess && fe[0])
            fprintf(stderr, "  [stamps] find_end_of_branching: %llu calls; per call: levels %.2f nodes %.2f skips %.2f | ticks: skip section %.0f (left junction %.0f) children from the store %.0f ADJ read + run set-up %.0f visited set + involved list %.0f\n",
                    fe[0], (double)fe[1] / fe[0], (double)fe[7] / fe[0], (double)fe[8] / fe[0], (double)fe[2] / fe[0], (double)fe[3] / fe[0], (double)fe[4] / fe[0], (double)fe[5] / fe[0], (double)fe[6] / fe[0]);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(mtg::g_fe), z, sizeof z);
        unsigned long long lf[40];
        if (hipMemcpyFromSymbol(lf, HIP_SYMBOL(mtg::g_life), sizeof lf) == hipSuccess) {
            fprintf(stderr, "  [stamps] traversal kernel: %llu ticks from the first lane's start to the last lane's end (%.3f ms of events); lanes by log2(life in ticks):", lf[33] - ~lf[32], st.kernel_ms);
            for (int i = 10; i < 32; i++) if (lf[i]) fprintf(stderr, " 2^%d:%llu", i, lf[i]);
            fprintf(stderr, "\n");
        }
        unsigned long long z2[40] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(mtg::g_life), z2, sizeof z2);
        unsigned long long ph[8];
        if (hipMemcpyFromSymbol(ph, HIP_SYMBOL(mtg::g_phase), sizeof ph) == hipSuccess && hs[15])
            fprintf(stderr, "  [stamps] outside W and B, per lane: contig starts %.2f x %.0f ticks; phase E %.2f x %.0f ticks; flat-loop iterations %.2f; from the lane's last iteration to the wave's end %.0f ticks\n",
                    (double)ph[1] / hs[15], ph[1] ? (double)ph[0] / ph[1] : 0.0, (double)ph[3] / hs[15], ph[3] ? (double)ph[2] / ph[3] : 0.0, (double)ph[4] / hs[15], (double)ph[5] / hs[15]);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(mtg::g_phase), z2, sizeof ph);
    }
#endif
    if (while_busy && !busy_done) (*while_busy)();
    if (rc == MTG_OK && n_todo) {
        set_error("%zu gap(s) exceeded the largest traversal scratch tier", n_todo);
        rc = MTG_ERR_OVERFLOW;
    }
    if (launches > 1) sink.in_gap_order = false;
    sink.device_records_whole = rc == MTG_OK && launches == 1 && st.n_retried_gaps == 0 && special.special.empty();
    if (stats) *stats = st;
    return rc;
}

int workspace_arena_download(const mtg_index* idx, Workspace* ws, char* dst, uint64_t bytes)
{
    if (int rc = use_device_of(idx)) return rc;
    if (!ws || !ws->ptr[Workspace::SLOT_SEQ] || ws->cap[Workspace::SLOT_SEQ] < bytes) { set_error("the workspace holds no sequence arena of %llu bytes", (unsigned long long)bytes); return MTG_ERR_ARG; }
    if (bytes) HIP_TRY(hipMemcpy(dst, ws->ptr[Workspace::SLOT_SEQ], bytes, hipMemcpyDeviceToHost));
    return MTG_OK;
}

} // namespace mtgi
