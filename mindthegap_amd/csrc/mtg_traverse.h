/*
 * mtg_traverse.h -- stage A of Filler::gapFillFromSource on the device: the breadth-first contig
 * construction that the reference delegates to gatb-core
 *   IterativeExtensions<span>::construct_linear_seqs   (call site /root/reference/src/Filler.cpp:884,
 *                                                        ctor args src/Filler.cpp:867)
 *   MonumentTraversal (TRAVERSAL_CONTIG) + Frontline + BranchingTerminator (src/Filler.cpp:866)
 * restated from SURVEY.md Appendix A.3-A.6.  One gap is processed by one lane; every lane owns a
 * private scratch block in HBM (layout: struct GapScratch).  The hot loop (simple-path walking) costs
 * one 64-byte ADJ line per nucleotide; the bubble / tip logic lives in noinline functions.
 *
 * Compiled for gfx950 by hipcc, and TEST-ONLY by g++ for tests/emu (see mtg_dev.h).
 */
#ifndef MTG_TRAVERSE_H
#define MTG_TRAVERSE_H
#include "mtg_dev.h"
#ifdef MTG_EMU
#include <cstdio>
#include <cstdlib>
#endif

namespace mtg {

enum GapStatus {
    GAP_OK = 0,
    GAP_OVF_CONTIG = 1,  /* contig arena full            */
    GAP_OVF_MARKED = 2,  /* marked-node set full          */
    GAP_OVF_SEEN = 3,    /* frontline visited set full    */
    GAP_OVF_INVOLVED = 4,
    GAP_OVF_QUEUE = 5,
    GAP_OVF_DFS = 6,
    GAP_PARKED = 100     /* the walk kernel met a branching node that is not the strict SNP pattern: its state is saved (WalkSave) and the
                            finishing kernel takes the gap from there (never seen outside the two kernels) */
};

/* uniform launch configuration */
struct FillCfg {
    int k;
    int max_nodes;       /* -max-nodes,  src/Filler.cpp:101 */
    int max_depth;       /* -max-length, src/Filler.cpp:100 */
    int mono_max_depth;  /* gatb MonumentTraversal max_depth   = 500 [MEM] */
    int mono_max_breadth;/* gatb MonumentTraversal max_breadth = 20  [MEM] */
    int end_rule_nonbranching;
    /* scratch capacities (per gap) */
    uint32_t cap_words;   /* contig arena, 32 nt per word */
    uint32_t cap_contigs; /* max_nodes + 1 */
    uint32_t qcap;        /* 4 * cap_contigs + 2 */
    uint32_t mcap;        /* marked set slots, power of two */
    uint32_t seen_cap;    /* frontline visited set slots, power of two */
    uint32_t inv_cap;     /* involved list */
    uint32_t iseen_cap;   /* nested frontline visited set, power of two */
    uint32_t cmd_cap;     /* deferred copies of a gap (CopyCmd), 0 = the lanes copy their long runs themselves */
    uint64_t zero_stride; /* bytes per gap in the zero-initialised region */
    uint64_t raw_stride;  /* bytes per gap in the raw region */
    uint64_t ilv_stride;  /* bytes per WAVE (64 gaps) in the lane-interleaved region */
    uint64_t hd_stride;   /* bytes per 64 gaps in the region of the small per-gap arrays (interleaved over the 64 gaps) */
    /* byte offsets of the per-gap arrays (filled by finalize_cfg) */
    uint32_t z_seen, z_iseen;
    uint32_t o_cstart, o_clen, o_qf, o_qc, o_qd, o_marklog, o_seenlog, o_iseenlog, o_inv, o_fl0, o_fl1, o_ifl0, o_ifl1, o_flnt0, o_flnt1, o_flaux0, o_flaux1, o_dfsf, o_dfsc,
        o_dfsmask, o_dfsnt, o_dfskid, o_cons, o_conslen, o_nw, o_tpos, o_terr, o_ttgt, o_flrp0, o_flrp1, o_flra0, o_flra1, o_iflrp0, o_iflrp1, o_iflra0, o_iflra1, o_dfsrp, o_dfsra, o_dfsdep, o_dfsxsn, o_cmd, o_save, o_lean;
};

enum { FL_CAP = 96, DFS_CAP = 512, CONS_CAP = 22, CONS_LEN = 512 };

/* per-gap scratch: two base pointers; the arrays sit at uniform offsets (FillCfg::o_*) */
struct GapScratch {
    uint8_t* z; /* all zero before the launch and restored to zero by the gap itself: marked | seen | iseen (canonical k-mer + 1, open addressing) */
    uint8_t* r; /* raw, contiguous per gap: contigs, queue, terminal info (read by k_post / k_compact / the host) */
    uint8_t* v; /* work areas of the bubble code, interleaved over the 64 lanes of a wave: element i of lane l sits at (i * 64 + l), so
                   that lanes touching the same index (frontline slot, DFS depth, ...) make one coalesced request instead of 64 */
    uint8_t* h; /* the small per-gap arrays every kernel of a launch reads or writes -- contig starts / lengths / terminal info, the queue, the copy
                   commands, the lean record --, interleaved over 64 consecutive gaps like v: with a gap per lane (the walk and its lean decision) or a few lanes per gap
                   (k_post_lean, k_emit_lean) neighbouring lanes touch neighbouring bytes, and a write of element 0 by the 64 lanes of a wave fills whole
                   memory lines (contiguous per gap, every such write dirtied a line of its own: 620 bytes written per gap for some 100 of content) */
    uint32_t lane;
    uint32_t snp_fast; /* 1: the SNP fast path may be used */
    MTG_LDS uint8_t* fp; /* TEST-ONLY emulation: a fingerprint table (FP_SLOTS slots for each of 64 lanes, see fp_at) for the cross-check of the fast path's distinctness test */
};
enum { FP_SLOTS = 256 };
/* strided view of one lane's array in the interleaved region */
template <typename T> struct SP {
    T* p;
    MTG_DEV T& operator[](size_t i) const { return p[i * 64]; }
    MTG_DEV SP operator+(size_t off) const { SP r; r.p = p + off * 64; return r; }
};
#define MTG_ILV(T, name, off) \
    MTG_DEV SP<T> name(const FillCfg& c, const GapScratch& S) { SP<T> r; r.p = reinterpret_cast<T*>(S.v + (uint64_t)(off) * 64) + S.lane; return r; }
#define MTG_HIL(T, name, off) \
    MTG_DEV SP<T> name(const FillCfg& c, const GapScratch& S) { SP<T> r; r.p = reinterpret_cast<T*>(S.h + (uint64_t)(off) * 64) + S.lane; return r; }
#define MTG_ARR(T, name, base, off) \
    MTG_DEV T* name(const FillCfg& c, const GapScratch& S) { return reinterpret_cast<T*>(S.base + (off)); }
MTG_ARR(uint64_t, s_marked, z, 0)
MTG_ARR(uint64_t, s_seen, z, c.z_seen)
MTG_ARR(uint64_t, s_iseen, z, c.z_iseen)
MTG_ARR(uint64_t, s_words, r, 0)             /* contig arena */
MTG_HIL(uint32_t, s_cstart, c.o_cstart)   /* first word of contig i */
MTG_HIL(uint32_t, s_clen, c.o_clen)       /* length in nt */
MTG_HIL(uint64_t, s_qf, c.o_qf)           /* BFS queue: oriented k-mer */
MTG_HIL(uint64_t, s_qc, c.o_qc)           /*   canonical k-mer (doubles as already_extended_from) */
MTG_HIL(int32_t, s_qd, c.o_qd)
/* A deferred copy: nwords whole words of the gap's contig arena, from word dst on, are the 32 * nwords nucleotides of the unitig store that
 * start at position src >> 1 and run forward (bit 0 clear) or backward, complemented (bit 0 set).  Written by the traversal instead of the
 * nucleotides themselves, executed by copy_gap (mtg_copy.h) with all lanes of a wave before anything reads the contigs. */
struct CopyCmd {
    uint64_t src;
    uint32_t dst;
    uint32_t nwords;
    uint32_t lead; /* nucleotides of the arena right before word dst that continue the same stretch of the store (the node the run was entered by, what the walk took
                      from the run itself before the long step): k_post finds the abundances of the k-mers in [32 dst - lead, 32 (dst + nwords)) next to the stretch */
    uint32_t pad_;
};
enum { COPY_CMDS = 32 };
MTG_HIL(CopyCmd, s_cmd, c.o_cmd)
/* a walk interrupted at a branching node (GAP_PARKED): everything stage_a_gap needs to go on from there, whoever resumes it.  The contigs
 * built so far, the queue and the copy commands are in the gap's raw block, the marked set in its zero block. */
struct WalkSave {
    uint64_t cur_f, prev_c, start_f, acc, start_base, r_base, msig[4];
    uint32_t len, c_first, nacc, wpos, head, tail, nb, total_nt, start_idx, r_idx, ncmd, copy_words, store_reads, run_nt, lines, n_marked, flags;
    uint32_t answered;  /* 1: a bubble kernel has answered the branching node the walk stands on: bn, bchosen, the consensus in the gap's s_cons area;
                           2: it was the strict SNP pattern: bn nucleotides in snp_lo / snp_hi (the SNP fast path's answer, consumed as the walk consumes its own) */
    int32_t node_depth, bn, bchosen, pad_;
    uint64_t snp_lo, snp_hi;
};
MTG_ARR(WalkSave, s_save, r, c.o_save)
/* A gap whose only contig holds the target at a place known without looking at the contig (the target's k-mer sits in a stored unitig, and the
 * stretch of the contig from its start to the target came out of that unitig in one run): nothing of the contig needs to be materialised --
 * coverage and ASCII are read off the unitig store (mtg_copy.h: copy_gap decides, k_post and k_emit follow). */
struct LeanRec {
    uint32_t valid; /* 0 / 1 */
    uint32_t pos0;  /* position of the target in contig 0 */
    uint32_t cmd;   /* the copy command that describes the run */
    uint32_t pad_;
};
MTG_HIL(LeanRec, s_lean, c.o_lean)
MTG_ILV(uint32_t, s_marklog, c.o_marklog) /* slots used in marked[] */
MTG_ILV(uint32_t, s_seenlog, c.o_seenlog) /* slots touched in seen[] */
MTG_ILV(uint32_t, s_iseenlog, c.o_iseenlog)
MTG_ILV(uint64_t, s_inv, c.o_inv)
MTG_ILV(uint64_t, s_fl0, c.o_fl0)         /* frontline double buffer */
MTG_ILV(uint64_t, s_fl1, c.o_fl1)
MTG_ILV(uint64_t, s_ifl0, c.o_ifl0)       /* nested frontline */
MTG_ILV(uint64_t, s_ifl1, c.o_ifl1)
MTG_ILV(uint8_t, s_flnt0, c.o_flnt0)
MTG_ILV(uint8_t, s_flnt1, c.o_flnt1)
MTG_ILV(uint32_t, s_flaux0, c.o_flaux0)   /* what is already known about a frontline node (node_aux) */
MTG_ILV(uint32_t, s_flaux1, c.o_flaux1)
MTG_ILV(uint64_t, s_flrp0, c.o_flrp0)     /* a frontline node's place in the unitig store (run_pack), 0 = unknown */
MTG_ILV(uint64_t, s_flrp1, c.o_flrp1)
MTG_ILV(uint32_t, s_flra0, c.o_flra0)     /* nodes ahead of it inside its unitig */
MTG_ILV(uint32_t, s_flra1, c.o_flra1)
MTG_ILV(uint64_t, s_iflrp0, c.o_iflrp0)   /* the same for the nested frontline */
MTG_ILV(uint64_t, s_iflrp1, c.o_iflrp1)
MTG_ILV(uint32_t, s_iflra0, c.o_iflra0)
MTG_ILV(uint32_t, s_iflra1, c.o_iflra1)
MTG_ILV(uint64_t, s_dfsrp, c.o_dfsrp)     /* the same for the frames of the consensus enumeration */
MTG_ILV(uint32_t, s_dfsra, c.o_dfsra)
MTG_ILV(uint32_t, s_dfsdep, c.o_dfsdep)   /* depth of a frame's node */
MTG_ILV(uint32_t, s_dfsxsn, c.o_dfsxsn)   /* abundance sum of the path up to and including the frame's stretch */
MTG_ILV(uint64_t, s_dfsf, c.o_dfsf)       /* consensus enumeration stack */
MTG_ILV(uint64_t, s_dfsc, c.o_dfsc)
MTG_ILV(uint8_t, s_dfsmask, c.o_dfsmask)
MTG_ILV(uint8_t, s_dfsnt, c.o_dfsnt)
MTG_ILV(uint32_t, s_dfskid, c.o_dfskid)   /* node_aux of the children of a frame */
MTG_ILV(uint8_t, s_cons, c.o_cons)        /* CONS_CAP x CONS_LEN nts */
MTG_ILV(uint16_t, s_conslen, c.o_conslen)
MTG_ILV(int32_t, s_nw, c.o_nw)            /* 4 rows x (CONS_LEN+1) */
MTG_HIL(uint32_t, s_tpos, c.o_tpos)       /* per contig: position of the best target match (0xFFFFFFFF: none) */
MTG_HIL(uint32_t, s_terr, c.o_terr)       /*             mismatches in the anchor */
MTG_HIL(uint32_t, s_ttgt, c.o_ttgt)       /*             index of the target */

inline uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }

/* host side: derive strides and offsets from the capacities */
inline void finalize_cfg(FillCfg& c)
{
    c.z_seen = 8u * c.mcap;
    c.z_iseen = c.z_seen + 8u * c.seen_cap;
    c.zero_stride = align_up((uint64_t)c.z_iseen + 8ull * c.iseen_cap, 64);
    /* contiguous per gap: the contig arena, the state of a parked walk */
    uint64_t b = 8ull * c.cap_words;
    b = align_up(b, 16);
    c.o_save = (uint32_t)b; b += (uint64_t)sizeof(WalkSave);
    c.raw_stride = align_up(b + 8, 64);
    /* interleaved over 64 gaps: byte offsets within one gap's share (every array starts 8-byte aligned) */
    b = 0;
    c.o_cstart = (uint32_t)b; b += 4ull * c.cap_contigs;
    b = align_up(b, 8);
    c.o_clen = (uint32_t)b; b += 4ull * c.cap_contigs;
    c.o_tpos = (uint32_t)b; b += 4ull * c.cap_contigs;
    c.o_terr = (uint32_t)b; b += 4ull * c.cap_contigs;
    c.o_ttgt = (uint32_t)b; b += 4ull * c.cap_contigs;
    b = align_up(b, 8);
    c.o_qf = (uint32_t)b; b += 8ull * c.qcap;
    c.o_qc = (uint32_t)b; b += 8ull * c.qcap;
    c.o_qd = (uint32_t)b; b += 4ull * c.qcap;
    b = align_up(b, 8);
    c.o_cmd = (uint32_t)b; b += (uint64_t)sizeof(CopyCmd) * COPY_CMDS; /* the room is there whether or not cmd_cap lets it be used */
    c.o_lean = (uint32_t)b; b += 16;
    c.hd_stride = align_up(b, 8) * 64;
    /* interleaved per wave: byte offsets within one lane's share (every array starts 8-byte aligned) */
    b = 0;
    c.o_marklog = (uint32_t)b; b += align_up(4ull * c.mcap, 8);
    c.o_seenlog = (uint32_t)b; b += align_up(4ull * c.seen_cap, 8);
    c.o_iseenlog = (uint32_t)b; b += align_up(4ull * c.iseen_cap, 8);
    c.o_inv = (uint32_t)b; b += 8ull * c.inv_cap;
    c.o_fl0 = (uint32_t)b; b += 8ull * FL_CAP;
    c.o_fl1 = (uint32_t)b; b += 8ull * FL_CAP;
    c.o_ifl0 = (uint32_t)b; b += 8ull * FL_CAP;
    c.o_ifl1 = (uint32_t)b; b += 8ull * FL_CAP;
    c.o_flnt0 = (uint32_t)b; b += align_up(FL_CAP, 8);
    c.o_flnt1 = (uint32_t)b; b += align_up(FL_CAP, 8);
    c.o_flaux0 = (uint32_t)b; b += 4ull * FL_CAP;
    c.o_flaux1 = (uint32_t)b; b += 4ull * FL_CAP;
    c.o_dfsf = (uint32_t)b; b += 8ull * DFS_CAP;
    c.o_dfsc = (uint32_t)b; b += 8ull * DFS_CAP;
    c.o_dfsmask = (uint32_t)b; b += DFS_CAP;
    c.o_dfsnt = (uint32_t)b; b += DFS_CAP;
    c.o_dfskid = (uint32_t)b; b += 4ull * DFS_CAP;
    c.o_cons = (uint32_t)b; b += align_up((uint64_t)CONS_CAP * CONS_LEN, 8);
    c.o_conslen = (uint32_t)b; b += align_up(2ull * CONS_CAP, 8);
    c.o_nw = (uint32_t)b; b += 4ull * 4 * (CONS_LEN + 1);
    b = align_up(b, 8);
    c.o_flrp0 = (uint32_t)b; b += 8ull * FL_CAP;
    c.o_flrp1 = (uint32_t)b; b += 8ull * FL_CAP;
    c.o_iflrp0 = (uint32_t)b; b += 8ull * FL_CAP;
    c.o_iflrp1 = (uint32_t)b; b += 8ull * FL_CAP;
    c.o_dfsrp = (uint32_t)b; b += 8ull * DFS_CAP;
    c.o_flra0 = (uint32_t)b; b += 4ull * FL_CAP;
    c.o_flra1 = (uint32_t)b; b += 4ull * FL_CAP;
    c.o_iflra0 = (uint32_t)b; b += 4ull * FL_CAP;
    c.o_iflra1 = (uint32_t)b; b += 4ull * FL_CAP;
    c.o_dfsra = (uint32_t)b; b += 4ull * DFS_CAP;
    c.o_dfsdep = (uint32_t)b; b += 4ull * DFS_CAP;
    c.o_dfsxsn = (uint32_t)b; b += 4ull * DFS_CAP;
    c.ilv_stride = align_up(b, 8) * 64;
}

MTG_DEV GapScratch carve(const FillCfg& c, uint8_t* zero_base, uint8_t* raw_base, uint8_t* ilv_base, uint8_t* head_base, uint64_t gap)
{
    GapScratch S;
    S.z = zero_base + gap * c.zero_stride;
    S.r = raw_base + gap * c.raw_stride;
    S.v = ilv_base + (gap >> 6) * c.ilv_stride;
    S.h = head_base + (gap >> 6) * c.hd_stride;
    S.lane = (uint32_t)(gap & 63);
    S.fp = nullptr;
    S.snp_fast = 0;
    return S;
}

/* ---- small open-addressing sets of canonical k-mers (stored +1, 0 = empty) ---------------- */
MTG_DEV uint32_t set_hash(uint64_t c, uint32_t cap) { return (uint32_t)((c * 0x9E3779B97F4A7C15ULL) >> 32) & (cap - 1); }
MTG_DEV bool set_has(const uint64_t* tab, uint32_t cap, uint64_t c)
{
    uint32_t h = set_hash(c, cap);
    for (;;) {
        uint64_t v = tab[h];
        if (v == 0) return false;
        if (v == c + 1) return true;
        h = (h + 1) & (cap - 1);
    }
}
/* returns the slot used (>=0), -1 if already present, -2 if full (never fills past cap-1) */
MTG_DEV int set_add(uint64_t* tab, uint32_t cap, uint32_t& count, uint64_t c)
{
    uint32_t h = set_hash(c, cap);
    for (;;) {
        uint64_t v = tab[h];
        if (v == c + 1) return -1;
        if (v == 0) {
            if (count + 1 >= cap) return -2;
            tab[h] = c + 1;
            count++;
            return (int)h;
        }
        h = (h + 1) & (cap - 1);
    }
}

#if defined(MTG_FINISH_DEBUG) && !defined(MTG_EMU)
__device__ unsigned int g_dbg[64];
#endif
/* diagnostic build (-DMTG_FINISH_DEBUG): every loop of the finishing kernel counts its rounds; one that exceeds its limit notes where in
 * g_dbg and leaves, so that a kernel that would hang ends and says where */
#if defined(MTG_FINISH_DEBUG) && !defined(MTG_EMU)
#define MTG_GUARD_DECL(v) unsigned v = 0
#ifndef MTG_FINISH_DEBUG_MASK
#define MTG_FINISH_DEBUG_MASK 0xFFFFFFFFu
#endif
#define MTG_GUARD(v, limit, code, action) do { if (((MTG_FINISH_DEBUG_MASK >> ((code) & 31)) & 1u) && ++(v) > (limit)) { atomicAdd(&g_dbg[(code) & 63], 1u); action; } } while (0)
#else
#define MTG_GUARD_DECL(v)
#define MTG_GUARD(v, limit, code, action)
#endif
/* diagnostic build (-DMTG_STAMPS): shader-clock time per phase, summed over lanes into a global array (never in the product build) */
#if defined(MTG_STAMPS) && !defined(MTG_EMU)
__device__ unsigned long long g_stamps[16];
__device__ unsigned long long g_forms[16]; /* the fast forms in detail, ticks and calls: [0,1] merge_fast [2,3] marked-successor test [4,5] first reads of the two branches [6,7] tip_fast [8,9] indel_bulk [10,11] snp_bulk [12,13] the step-by-step loop [14] its steps [15] lanes */
__device__ unsigned long long g_fe[16]; /* find_end_of_branching in detail: [0] calls [1] levels [2] skip section [3] left junction [4] children known from the store [5] ADJ read + run set-up [6] visited set + involved list [7] nodes [8] skips */
__device__ unsigned long long g_life[40]; /* [0..31]: lanes by log2 of their life in clock ticks; [32] ~(earliest start), [33] latest end (0 = unset) */
__device__ unsigned long long g_phase[8]; /* the flat loop outside W and B: [0] ticks starting a contig (pop, first neighbourhood, the target's place) [1] contigs started [2] ticks in phase E (close, push successors) [3] contigs closed [4] iterations of the flat loop [5] ticks from the lane's own last iteration to the wave's end */
#define MTG_T0(v) unsigned long long v = __builtin_amdgcn_s_memtime()
#define MTG_T1(v, slot) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp_acc[slot] += t_ - v; } while (0)
#define MTG_COUNT(W_, slot) ((W_).stamp_acc[slot] += 1ull)
#define MTG_F0(v) unsigned long long v = __builtin_amdgcn_s_memtime()
#define MTG_F1(W_, v, slot) do { (W_).form_acc[slot] += __builtin_amdgcn_s_memtime() - v; (W_).form_acc[(slot) + 1] += 1ull; } while (0)
#else
#define MTG_F0(v)
#define MTG_F1(W_, v, slot)
#define MTG_T0(v)
#define MTG_T1(v, slot)
#define MTG_COUNT(W_, slot)
#endif

/* ---- per-gap walker state ------------------------------------------------------------------ */
struct Worker {
    const Index& ix;
    const FillCfg& cfg;
    GapScratch S;
    int k;
    uint64_t mk, mk1;
    uint32_t lines;      /* 64-byte index lines read */
    uint32_t status;
    uint32_t n_marked, n_seen, n_iseen, n_inv;
    uint64_t msig0 = 0, msig1 = 0, msig2 = 0, msig3 = 0;
    bool no_dp = false; /* the SNP fast path gives up where it would need the alignment itself (the walk kernel: the bubble is then parked) */
#if defined(MTG_STAMPS) && !defined(MTG_EMU)
    unsigned long long stamp_acc[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long form_acc[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long phase_acc[6] = {0, 0, 0, 0, 0, 0};
#endif

    MTG_DEV Worker(const Index& i, const FillCfg& c, const GapScratch& s)
        : ix(i), cfg(c), S(s), k(i.k), mk(kmask(i.k)), mk1(kmask(i.k - 1)), lines(0), status(GAP_OK), n_marked(0), n_seen(0), n_iseen(0),
          n_inv(0)
    {
    }

    /* 256-bit register signature of the marked set: most membership tests are answered without touching memory */
    MTG_DEV static uint32_t sig_bit(uint64_t c) { return (uint32_t)((c * 0x9E3779B97F4A7C15ULL) >> 56); }
    MTG_DEV void mark_canon(uint64_t c)
    {
        const int slot = set_add(s_marked(cfg, S), cfg.mcap, n_marked, c);
        if (slot == -2) status = GAP_OVF_MARKED;
        if (slot >= 0) s_marklog(cfg, S)[n_marked - 1] = (uint32_t)slot;
        const uint32_t b = sig_bit(c), q = b >> 6;
        const uint64_t m = 1ull << (b & 63);
        /* four unconditional ORs of selected VALUES: an if / else chain over the four words became one read-modify-write through a selected ADDRESS
         * (round 6, read off the IR), and a Worker whose address is computed at run time lives in private memory with every one of its fields --
         * lines, status, the counters -- a scratch access per use */
        msig0 |= q == 0u ? m : 0ull;
        msig1 |= q == 1u ? m : 0ull;
        msig2 |= q == 2u ? m : 0ull;
        msig3 |= q == 3u ? m : 0ull;
    }
    /* the three sets live in memory that is zero before a launch and must be zero again after it (nobody clears it in between) */
    MTG_DEV void marked_clear() { for (uint32_t i = 0; i < n_marked; i++) s_marked(cfg, S)[s_marklog(cfg, S)[i]] = 0; n_marked = 0; }
    /* register-only half of is_marked: false = certainly not marked */
    MTG_DEV bool maybe_marked(uint64_t c) const
    {
        const uint32_t b = sig_bit(c), q = b >> 6;
        const uint64_t w = (q == 0u ? msig0 : 0ull) | (q == 1u ? msig1 : 0ull) | (q == 2u ? msig2 : 0ull) | (q == 3u ? msig3 : 0ull); /* values, not a selected address (see mark_canon) */
        return ((w >> (b & 63)) & 1) != 0;
    }
    MTG_DEV bool is_marked(uint64_t c) const
    {
        if (!maybe_marked(c)) return false;
        return set_has(s_marked(cfg, S), cfg.mcap, c);
    }
    /* adds c to the frontline visited set; true when it was not there yet (one probe sequence for test + insert) */
    MTG_DEV bool seen_test_add(uint64_t c)
    {
        int s = set_add(s_seen(cfg, S), cfg.seen_cap, n_seen, c);
        if (s == -2) { status = GAP_OVF_SEEN; return false; }
        if (s >= 0) s_seenlog(cfg, S)[n_seen - 1] = (uint32_t)s;
        return s >= 0;
    }
    MTG_DEV bool is_branching(const Kmer& x)
    {
        Adj l = adj_left(ix, x, mk1, lines);
        if (popc4(l.in) != 1) return true;
        Adj r = adj_right(ix, x, mk1, lines);
        return popc4(r.out) != 1;
    }
    MTG_DEV void mark(const Kmer& x) { if (is_branching(x)) mark_canon(canon(x)); }

    /* frontline visited sets with undo logs */
    MTG_DEV bool seen_add(uint64_t c)
    {
        int s = set_add(s_seen(cfg, S), cfg.seen_cap, n_seen, c);
        if (s == -2) { status = GAP_OVF_SEEN; return false; }
        if (s >= 0) s_seenlog(cfg, S)[n_seen - 1] = (uint32_t)s;
        return s >= 0;
    }
    MTG_DEV void seen_clear() { for (uint32_t i = 0; i < n_seen; i++) s_seen(cfg, S)[s_seenlog(cfg, S)[i]] = 0; n_seen = 0; }
    MTG_DEV bool iseen_add(uint64_t c)
    {
        int s = set_add(s_iseen(cfg, S), cfg.iseen_cap, n_iseen, c);
        if (s == -2) { status = GAP_OVF_SEEN; return false; }
        if (s >= 0) s_iseenlog(cfg, S)[n_iseen - 1] = (uint32_t)s;
        return s >= 0;
    }
    MTG_DEV void iseen_clear() { for (uint32_t i = 0; i < n_iseen; i++) s_iseen(cfg, S)[s_iseenlog(cfg, S)[i]] = 0; n_iseen = 0; }
    MTG_DEV void involve(uint64_t c)
    {
        if (n_inv >= cfg.inv_cap) { status = GAP_OVF_INVOLVED; return; }
        s_inv(cfg, S)[n_inv++] = c;
    }
};

/* What one bucket read tells about the nodes ahead, carried along by the bubble routines so that they ask the index only for what
 * they do not know yet.  node_aux of a node y: bit 31 = y has in-degree 1; bits 0..3 = c, bits 4.. = nucleotides: the next c nodes
 * along y's single out-edge are known (y has out-degree 1 and leads, by the first nucleotide, to a node of in-degree 1 whose node_aux
 * is this one shifted by one).  From a lookahead entry (MTG_LA_MAX = 14 nucleotides) at most 13 are kept so that it fits 32 bits. */
enum : uint32_t { AUX_IN1 = 0x80000000u };
enum : uint64_t { INV_SIMPLE = 1ull << 63, INV_KMER = (1ull << 62) - 1 }; /* flag on an entry of the involved list: known to be a simple node */
/* node_aux of the successors of x, given x's right neighbourhood a (successors of x, their common in-edges, lookahead) */
MTG_DEV uint32_t aux_of_children(const Adj& a)
{
    if (popc4(a.in) != 1) return 0u;
    uint32_t aux = AUX_IN1;
    if (popc4(a.out) == 1) {
        uint32_t known = a.la & 15u;
        if (known > 13u) known = 13u;
        aux |= known | (((a.la >> 4) & ((1u << (2 * known)) - 1u)) << 4);
    }
    return aux;
}
/* node_aux of the single successor of a node whose own node_aux has a positive count */
MTG_DEV uint32_t aux_step(uint32_t aux) { return AUX_IN1 | ((aux & 15u) - 1u) | (((aux & 0x7FFFFFFFu) >> 6) << 4); }
MTG_DEV uint64_t inv_flags(uint32_t aux) { return (aux & 15u) ? INV_SIMPLE : 0ull; }

/* a node's place in the unitig store packed into one word: [valid:1][bwd:1][hdr:24 low bits of the header word, for the distinctness
 * test only][kpos:38].  The header word itself is recovered where needed from kpos (not needed: only compared). */
enum : uint64_t { RP_VALID = 1ull << 63, RP_BWD = 1ull << 62, RP_KPOS = (1ull << 38) - 1 };
MTG_DEV uint64_t rp_pack(const RunAt& r) { return RP_VALID | (r.bwd ? RP_BWD : 0ull) | ((uint64_t)(r.hdr & 0xFFFFFFu) << 38) | (r.kpos & RP_KPOS); }
MTG_DEV uint64_t rp_step(uint64_t rp, uint32_t t) { return (rp & ~RP_KPOS) | (((rp & RP_BWD) ? (rp & RP_KPOS) - t : (rp & RP_KPOS) + t) & RP_KPOS); }
/* lowest set bit of a 4-bit edge mask.  The loops over the edges of a node go from set bit to set bit (ascending nucleotide, as the
 * reference enumerates) instead of over the four nucleotides: the lanes of a wave then run the loop body -- whose memory reads are what a
 * trip costs -- together for their first edge, whatever its nucleotide, and not once per nucleotide any lane has. */
MTG_DEV uint32_t low_nt(uint32_t m) { return (m & 1u) ? 0u : (m & 2u) ? 1u : (m & 4u) ? 2u : 3u; }
MTG_DEV uint32_t rp_unitig(uint64_t rp) { return (uint32_t)(rp >> 38) & 0xFFFFFFu; } /* 24 bits of the header word: equal unitigs give equal values (a false "equal" only costs speed) */

/* [MEM] gatb FrontlineBranching::check: look for large in-branching at m.
 * dir: 0 = frontline moves along successors.  Only used with dir 0 on this path. */
MTG_DEV_NOINLINE bool fl_check(Worker& W, uint64_t mf)
{
    const int k = W.k;
    const UStore& us = W.ix.us;
    Kmer m = make_kmer(mf, k);
    Adj l = adj_left(W.ix, m, W.mk1, W.lines);
    /* gatb's "just a speedup": with in-degree 1 the only predecessor is the frontline node m was reached from, which is in the visited set */
    if (popc4(l.in) == 1) return true;
    for (uint32_t em = l.in & 15u; em; em &= em - 1u) {
        const uint32_t nt = low_nt(em);
        Kmer b = kmer_prev(m, nt, k, W.mk);
        if (set_has(s_seen(W.cfg, W.S), W.cfg.seen_cap, canon(b))) continue;
        /* plain frontline from b along predecessors, previous node = m.  Unitig-aware like find_end_of_branching: walking backwards from
         * a node = walking forwards from its reverse complement, whose right junction tells where it sits in the store; when every node
         * of this frontline has two or more nodes of its unitig behind it, and they sit in pairwise different unitigs, the frontline moves
         * by the smallest such distance (less one) at once -- it keeps its size, so only the depth limit can end it meanwhile. */
        W.iseen_add(canon(b));
        W.iseen_add(canon(m));
        int cur = 0, ncur = 1, depth = 0, remaining = 0;
        s_ifl0(W.cfg, W.S)[0] = b.f;
        s_iflrp0(W.cfg, W.S)[0] = 0;
        s_iflra0(W.cfg, W.S)[0] = 0;
        for (;;) {
            /* go_next_depth */
            bool cont = true;
            int nnext = 0;
            const SP<uint64_t> cf = cur ? s_ifl1(W.cfg, W.S) : s_ifl0(W.cfg, W.S);
            const SP<uint64_t> nf = cur ? s_ifl0(W.cfg, W.S) : s_ifl1(W.cfg, W.S);
            const SP<uint64_t> crp = cur ? s_iflrp1(W.cfg, W.S) : s_iflrp0(W.cfg, W.S);
            const SP<uint64_t> nrp = cur ? s_iflrp0(W.cfg, W.S) : s_iflrp1(W.cfg, W.S);
            const SP<uint32_t> cra = cur ? s_iflra1(W.cfg, W.S) : s_iflra0(W.cfg, W.S);
            const SP<uint32_t> nra = cur ? s_iflra0(W.cfg, W.S) : s_iflra1(W.cfg, W.S);
            if (us.nwords && ncur >= 1 && ncur <= FL_CAP && depth > 0) {
                uint32_t D = 0xFFFFFFFFu;
                bool ok = true;
                for (int i = 0; i < ncur && ok; i++) {
                    const uint32_t ra = cra[i];
                    if (!(crp[i] & RP_VALID) || ra < 2u) ok = false;
                    else if (ra - 1u < D) D = ra - 1u;
                    for (int j = 0; j < i && ok; j++) if (rp_unitig(crp[j]) == rp_unitig(crp[i])) ok = false;
                }
                if (ok) {
                    if ((uint32_t)depth + D > 3u * (uint32_t)k) { remaining = ncur; break; } /* it would still be there when the depth limit ends the search */
                    for (int i = 0; i < ncur; i++) {
                        const uint64_t rp = crp[i];
                        cf[i] = run_node(us, rp & RP_KPOS, (rp & RP_BWD) != 0, D, k).r; /* the store walk is that of the reverse complement */
                        crp[i] = rp_step(rp, D);
                        cra[i] = cra[i] - D;
                    }
                    W.lines += (uint32_t)ncur;
                    depth += (int)D;
                    remaining = ncur;
                    continue;
                }
            }
            for (int i = 0; i < ncur && cont; i++) {
                Kmer x = make_kmer(cf[i], k);
                uint32_t in_mask;
                uint64_t krp = 0;
                uint32_t kra = 0;
                const uint64_t rp = crp[i];
                if ((rp & RP_VALID) && cra[i] >= 1u) {
                    in_mask = 1u << (run_next_nt(us, rp & RP_KPOS, (rp & RP_BWD) != 0, k) ^ 2u); /* the reverse complement's next nucleotide, complemented */
                    krp = rp_step(rp, 1);
                    kra = cra[i] - 1u;
                } else {
                    Kmer xr;
                    xr.f = x.r; xr.r = x.f;
                    const Adj ar = adj_right_t(W.ix.adj, xr, W.mk1, W.lines); /* the same entry adj_left(x) reads */
                    in_mask = comp_mask(ar.out);
                    RunAt r;
                    if (us.nwords && run_at(us, ar, k, r, W.lines)) { krp = rp_step(rp_pack(r), 1); kra = r.ahead - 1u; }
                }
                for (uint32_t em = in_mask & 15u; em; em &= em - 1u) {
                    const uint32_t n2 = low_nt(em);
                    Kmer y = kmer_prev(x, n2, k, W.mk);
                    uint64_t cy = canon(y);
                    if (set_has(s_iseen(W.cfg, W.S), W.cfg.iseen_cap, cy)) continue;
                    if (W.is_marked(cy)) { cont = false; remaining = ncur - i - 1; break; }
                    if (nnext < FL_CAP) { nf[nnext] = y.f; nrp[nnext] = krp; nra[nnext] = kra; }
                    nnext++;
                    W.iseen_add(cy);
                    W.involve(cy | (((krp & RP_VALID) && kra >= 1u) ? INV_SIMPLE : 0ull));
                }
            }
            if (!cont) break;
            cur ^= 1; ncur = nnext; remaining = ncur; depth++;
            if (depth > 3 * k) break;
            if (ncur > 10) break;
            if (ncur == 0) break;
            if (W.status) break;
        }
        W.iseen_clear();
        if (remaining > 0) return false;
        if (W.status) return false;
    }
    return true;
}

/* [MEM] MonumentTraversal::find_end_of_branching.  Returns depth (0 = failure).
 *
 * Unitig-aware: a frontline node whose right junction lies inside a stored unitig knows the nodes ahead of it (run_at); its child
 * inherits the knowledge.  When EVERY node of the frontline (at least two) has two or more nodes of its unitig ahead, the frontline
 * advances by D = the smallest such distance (less one: a unitig's last node is always reached level by level) in one go.  Nothing the
 * reference does on those D levels can change the outcome: the nodes passed have one in- and one out-edge (no FrontlineBranching check, no
 * node is ever marked: only branching nodes are), the frontline keeps its size, and the visited set would only matter if a node passed
 * were met again -- a node inside a unitig is reached through the unitig alone, so that takes another node of the frontline travelling the
 * same unitig (the other way), or the unitig of the node before the start (which is in the visited set from the beginning): the frontline
 * only skips while its nodes sit in pairwise different unitigs, none of them the previous node's.  The nodes passed are not entered into
 * the visited set or the involved list (of the involved nodes only the branching ones are marked afterwards).  prev_c == 0 (no previous
 * node: gatb's default-constructed node, whose k-mer value 0 is a real k-mer) keeps the level-by-level search. */
MTG_DEV_NOINLINE int find_end_of_branching(Worker& W, const Kmer& start, uint64_t prev_c, uint64_t& end_f, uint64_t& end_rp)
{
    const int k = W.k;
    const UStore& us = W.ix.us;
    W.seen_add(canon(start));
    W.seen_add(prev_c);
    int cur = 0, ncur = 1, depth = 0;
    s_fl0(W.cfg, W.S)[0] = start.f;
    s_flnt0(W.cfg, W.S)[0] = 255;
    s_flaux0(W.cfg, W.S)[0] = 0;
    s_flrp0(W.cfg, W.S)[0] = 0;
    s_flra0(W.cfg, W.S)[0] = 0;
    const bool may_skip = us.nwords != 0 && prev_c != 0;
    uint32_t prev_unitig = 0xFFFFFFFFu; /* unitig of the junction between the previous node and the start, looked up when the first skip is considered */
    bool prev_known = false;
#if defined(MTG_STAMPS) && !defined(MTG_EMU)
    struct FeAcc { unsigned long long v[9] = {1, 0, 0, 0, 0, 0, 0, 0, 0}; MTG_DEV ~FeAcc() { for (int i = 0; i < 9; i++) if (v[i]) atomicAdd(&g_fe[i], v[i]); } } fe;
#define FE_T0(x) unsigned long long x = __builtin_amdgcn_s_memtime()
#define FE_T1(x, slot) fe.v[slot] += __builtin_amdgcn_s_memtime() - x
#define FE_N(slot, n) fe.v[slot] += (n)
#else
#define FE_T0(x)
#define FE_T1(x, slot)
#define FE_N(slot, n)
#endif
    for (;;) {
        const SP<uint64_t> cf = cur ? s_fl1(W.cfg, W.S) : s_fl0(W.cfg, W.S);
        const SP<uint8_t> cn = cur ? s_flnt1(W.cfg, W.S) : s_flnt0(W.cfg, W.S);
        const SP<uint32_t> ca = cur ? s_flaux1(W.cfg, W.S) : s_flaux0(W.cfg, W.S);
        const SP<uint64_t> crp = cur ? s_flrp1(W.cfg, W.S) : s_flrp0(W.cfg, W.S);
        const SP<uint32_t> cra = cur ? s_flra1(W.cfg, W.S) : s_flra0(W.cfg, W.S);
        const SP<uint64_t> nf = cur ? s_fl0(W.cfg, W.S) : s_fl1(W.cfg, W.S);
        const SP<uint8_t> nn = cur ? s_flnt0(W.cfg, W.S) : s_flnt1(W.cfg, W.S);
        const SP<uint32_t> na = cur ? s_flaux0(W.cfg, W.S) : s_flaux1(W.cfg, W.S);
        const SP<uint64_t> nrp = cur ? s_flrp0(W.cfg, W.S) : s_flrp1(W.cfg, W.S);
        const SP<uint32_t> nra = cur ? s_flra0(W.cfg, W.S) : s_flra1(W.cfg, W.S);
        /* ---- the skip */
        FE_T0(t_sk);
        if (may_skip && ncur >= 2 && ncur <= FL_CAP && depth > 0) {
            uint32_t D = 0xFFFFFFFFu;
            bool ok = true;
            for (int i = 0; i < ncur && ok; i++) {
                const uint32_t ra = cra[i];
                if (!(crp[i] & RP_VALID) || ra < 2u) ok = false;
                else if (ra - 1u < D) D = ra - 1u;
            }
            if (ok) {
                if (!prev_known) { /* the junction between the previous node and the start = the start's left junction */
                    prev_known = true;
                    FE_T0(t_lj);
                    prev_unitig = left_junction_unitig(W.ix, start, W.mk1, W.lines);
                    FE_T1(t_lj, 3);
                }
                for (int i = 0; i < ncur && ok; i++) {
                    const uint32_t u = rp_unitig(crp[i]);
                    if (u == prev_unitig) ok = false;
                    for (int j = 0; j < i && ok; j++) if (rp_unitig(crp[j]) == u) ok = false;
                }
            }
            if (ok) {
                /* D levels at once; the frontline keeps its size (>= 2), so the only way out is the depth limit */
                if ((uint32_t)depth + D > (uint32_t)W.cfg.mono_max_depth) return 0;
                for (int i = 0; i < ncur; i++) {
                    const uint64_t rp = crp[i];
                    cf[i] = run_node(us, rp & RP_KPOS, (rp & RP_BWD) != 0, D, k).f;
                    crp[i] = rp_step(rp, D);
                    cra[i] = cra[i] - D;
                    ca[i] = AUX_IN1;
                }
                W.lines += (uint32_t)ncur;
                depth += (int)D;
                FE_T1(t_sk, 2); FE_N(8, 1);
                continue;
            }
        }
        FE_T1(t_sk, 2);
        /* ---- one level */
        FE_N(1, 1); FE_N(7, ncur);
        int nnext = 0;
        for (int i = 0; i < ncur; i++) {
            const uint32_t aux = ca[i];
            /* a node of in-degree 1 passes the check at once (its only predecessor is the frontline node it was reached from) */
#if defined(MTG_STAMPS) && !defined(MTG_EMU)
            unsigned long long* stamp_acc = W.stamp_acc;
#endif
            MTG_T0(t_flc);
            const bool flc_ok = !(depth > 0 && !(aux & AUX_IN1)) || fl_check(W, cf[i]);
            MTG_T1(t_flc, 12);
            if (!flc_ok) return 0;
            Kmer x = make_kmer(cf[i], k);
            uint32_t out, kid;
            uint64_t krp = 0; /* the children's place in the store, when the node's right junction lies inside a unitig */
            uint32_t kra = 0;
            const uint64_t rp = crp[i];
            FE_T0(t_ch);
            if ((rp & RP_VALID) && cra[i] >= 1u) { /* inside a unitig: the way ahead is known */
                out = 1u << run_next_nt(us, rp & RP_KPOS, (rp & RP_BWD) != 0, k);
                kid = AUX_IN1;
                krp = rp_step(rp, 1);
                kra = cra[i] - 1u;
                FE_T1(t_ch, 4);
            } else if (aux & 15u) { out = 1u << ((aux >> 4) & 3u); kid = aux_step(aux); } /* inline lookahead: nothing to read */
            else {
                Adj a = adj_right_t(W.ix.adj, x, W.mk1, W.lines);
                RunAt r;
                if (run_at(us, a, k, r, W.lines)) { krp = rp_step(rp_pack(r), 1); kra = r.ahead - 1u; }
                else adj_resolve_la(W.ix, a, W.lines);
                out = a.out;
                kid = aux_of_children(a);
                FE_T1(t_ch, 5);
            }
            FE_T0(t_set);
            for (uint32_t em = out & 15u; em; em &= em - 1u) {
                const uint32_t nt = low_nt(em);
                Kmer y = kmer_next(x, nt, k, W.mk);
                uint64_t cy = canon(y);
                if (!W.seen_test_add(cy)) continue;  /* already explored (on failure below the whole set is discarded anyway) */
                if (W.is_marked(cy)) return 0; /* bubble touches an assembled region */
                if (nnext < FL_CAP) { nf[nnext] = y.f; nn[nnext] = (cn[i] == 255) ? (uint8_t)nt : cn[i]; na[nnext] = kid; nrp[nnext] = krp; nra[nnext] = kra; }
                nnext++;
                W.involve(cy | ((kid & 15u) || ((krp & RP_VALID) && kra >= 1u) ? INV_SIMPLE : 0ull));
            }
            FE_T1(t_set, 6);
            if (W.status) return 0;
        }
        cur ^= 1; ncur = nnext; depth++;
        if (depth > W.cfg.mono_max_depth) return 0;
        if (ncur > W.cfg.mono_max_breadth) return 0;
        if (ncur == 0) return 0;
        if (ncur == 1) {
            if (!W.cfg.end_rule_nonbranching) break;
            Kmer e = make_kmer((cur ? s_fl1(W.cfg, W.S) : s_fl0(W.cfg, W.S))[0], k);
            if (!W.is_branching(e)) break;
        }
    }
    end_f = (cur ? s_fl1(W.cfg, W.S) : s_fl0(W.cfg, W.S))[0];
    end_rp = (cur ? s_flrp1(W.cfg, W.S) : s_flrp0(W.cfg, W.S))[0];
    return depth;
}

#ifdef MTG_XCHECK
/* TEST-ONLY: the enumeration node by node, one frame per node, exactly as the reference recurses; the emulation build runs it after
 * all_consensuses_between and compares (status 0xBAD4) */
MTG_DEV_NOINLINE bool all_consensuses_between_nodes(Worker& W, const Kmer& start, uint64_t end_c, int traversal_depth, int& ncons)
{
    const int k = W.k;
    const SP<uint64_t> dfs_f = s_dfsf(W.cfg, W.S);
    const SP<uint64_t> dfs_c = s_dfsc(W.cfg, W.S);
    const SP<uint8_t> dfs_mask = s_dfsmask(W.cfg, W.S);
    const SP<uint8_t> dfs_nt = s_dfsnt(W.cfg, W.S);
    const SP<uint32_t> dfs_kid = s_dfskid(W.cfg, W.S);
    const SP<uint8_t> cons = s_cons(W.cfg, W.S);
    const SP<uint16_t> cons_len = s_conslen(W.cfg, W.S);
    ncons = 0;
    int d = 0; /* current frame */
    dfs_f[0] = start.f;
    dfs_c[0] = canon(start);
    /* k-mers of the current path: kept in the nested-frontline hash set (idle here) so that the loop test is one probe instead of a scan
     * of the whole path; a popped k-mer leaves a tombstone (canonical k-mers are < 2^62, so ~0 - 1 is free) */
    uint64_t* pset = s_iseen(W.cfg, W.S);
    const uint32_t pcap = W.cfg.iseen_cap;
    const uint64_t TOMB = ~0ULL - 1;
    uint32_t plog_n = 0;
    const SP<uint32_t> plog = s_iseenlog(W.cfg, W.S);
    /* 0: added, 1: already on the path, 2: table full.  A tombstone met on the way is reused only after the probe sequence has
     * proven the key absent. */
    auto path_add = [&](uint64_t c) -> int {
        uint32_t h = set_hash(c, pcap);
        int64_t tomb = -1;
        for (;;) {
            const uint64_t v = pset[h];
            if (v == c + 1) return 1;
            if (v == TOMB && tomb < 0) tomb = (int64_t)h;
            if (v == 0) {
                if (tomb >= 0) { pset[(uint32_t)tomb] = c + 1; return 0; }
                if (plog_n + 2 >= pcap) return 2;
                plog[plog_n++] = h;
                pset[h] = c + 1;
                return 0;
            }
            h = (h + 1) & (pcap - 1);
        }
    };
    auto path_del = [&](uint64_t c) {
        uint32_t h = set_hash(c, pcap);
        for (;;) {
            const uint64_t v = pset[h];
            if (v == 0) return;
            if (v == c + 1) { pset[h] = TOMB; return; }
            h = (h + 1) & (pcap - 1);
        }
    };
    auto path_clear = [&]() { for (uint32_t i = 0; i < plog_n; i++) pset[plog[i]] = 0; plog_n = 0; };
    path_add(dfs_c[0]);
    /* enter frame 0 */
    bool entering = true;
    auto run = [&]() -> bool {
    for (;;) {
        if (entering) {
            entering = false;
            int depth_left = traversal_depth - d;
            if (depth_left < -1) return false;
            if (dfs_c[d] == end_c) {
                if (ncons >= CONS_CAP || d > CONS_LEN) { W.status = GAP_OVF_DFS; return false; }
                for (int i = 0; i < d; i++) cons[(size_t)ncons * CONS_LEN + i] = dfs_nt[i];
                cons_len[ncons] = (uint16_t)d;
                ncons++;
                dfs_mask[d] = 0; /* return */
            } else {
                const uint32_t aux = d ? dfs_kid[d - 1] : 0u; /* what the parent's read told about this node */
                if (aux & 15u) { dfs_mask[d] = (uint8_t)(1u << ((aux >> 4) & 3u)); dfs_kid[d] = aux_step(aux); }
                else {
                    Kmer x = make_kmer(dfs_f[d], k);
                    const Adj a = adj_right(W.ix, x, W.mk1, W.lines);
                    dfs_mask[d] = (uint8_t)a.out;
                    dfs_kid[d] = aux_of_children(a);
                }
            }
        }
        uint32_t mask = dfs_mask[d];
        if (mask == 0) {
            /* return to the parent; the parent re-checks the breadth limit after each child */
            if (d == 0) return true;
            path_del(dfs_c[d]);
            d--;
            if (ncons > W.cfg.mono_max_breadth) return false;
            continue;
        }
        uint32_t nt = (uint32_t)ctz4(mask);
        dfs_mask[d] = (uint8_t)(mask & (mask - 1));
        Kmer x = make_kmer(dfs_f[d], k);
        Kmer y = kmer_next(x, nt, k, W.mk);
        uint64_t cy = canon(y);
        if (d + 1 >= DFS_CAP) { W.status = GAP_OVF_DFS; return false; }
        const int pa = path_add(cy);
        if (pa == 1) return false; /* loop inside the bubble */
        if (pa == 2) { W.status = GAP_OVF_DFS; return false; }
        dfs_nt[d] = (uint8_t)nt;
        d++;
        dfs_f[d] = y.f;
        dfs_c[d] = cy;
        entering = true;
    }
    };
    const bool ok = run();
    path_clear();
    return ok;
}
#endif

/* [MEM] MonumentTraversal::all_consensuses_between, explicit stack.
 * Consensuses come out in lexicographic (A,C,T,G) order, which is std::set<Path> order.
 *
 * Unitig-aware.  A frame is a node of the current path; a node whose right junction lies inside a stored unitig, with at least two nodes of
 * the unitig ahead, has ONE child frame: the unitig's end node, `ahead` levels deeper.  What the reference does on the levels in between
 * cannot change the outcome:
 *   - the nodes passed have one out-edge (one recursive call each) and the depth test is monotone: it fails on one of them iff it fails on
 *     the deepest, which is tested;
 *   - none of them is the end node: a node that is not the first of its unitig (in the direction the frontline reached it) carries its place
 *     in the store (end_rp), and a unitig that may be the end node's is walked node by node;
 *   - none of them is on the path already: a node inside a unitig is reached through the unitig alone, so a path that holds it also holds
 *     the node the stretch was entered by or the unitig's end node (either strand: same canonical k-mers), and those two are frames, tested
 *     and kept in the path set as always; for the same reason a later node of the path that equals a passed node is caught at one of the two;
 *   - the breadth test after each return is repeated unchanged (the number of consensuses does not change on the way up).
 * The path's nucleotides are written in bulk from the store.  Along the way the abundances are summed (one byte per node next to the unitig,
 * wide loads for a stretch; an ABND look-up for a node outside the store), so that every consensus comes with the sum the reference's
 * most_abundant_consensus computes over [start, nodes before the end] (cons_sum), and validate_consensuses need not walk them again. */
MTG_DEV SP<int32_t> s_cons_sum(const FillCfg& c, const GapScratch& S) { return s_nw(c, S) + 2 * (CONS_LEN + 1); } /* rows 2.. of the alignment area (nw_matches uses rows 0, 1) */
MTG_DEV_NOINLINE bool all_consensuses_between(Worker& W, const Kmer& start, uint64_t end_c, uint64_t end_rp, int traversal_depth, int& ncons)
{
    const int k = W.k;
    const UStore& us = W.ix.us;
    const SP<uint64_t> dfs_f = s_dfsf(W.cfg, W.S);
    const SP<uint64_t> dfs_c = s_dfsc(W.cfg, W.S);
    const SP<uint8_t> dfs_mask = s_dfsmask(W.cfg, W.S);
    const SP<uint8_t> dfs_nt = s_dfsnt(W.cfg, W.S);      /* by depth */
    const SP<uint32_t> dfs_kid = s_dfskid(W.cfg, W.S);
    const SP<uint64_t> dfs_rp = s_dfsrp(W.cfg, W.S);
    const SP<uint32_t> dfs_ra = s_dfsra(W.cfg, W.S);
    const SP<uint32_t> dfs_dep = s_dfsdep(W.cfg, W.S);   /* depth of the frame's node = nucleotides of the path before it */
    const SP<uint32_t> dfs_xsn = s_dfsxsn(W.cfg, W.S);   /* abundances of the path's nodes before the frame's child */
    const SP<uint8_t> cons = s_cons(W.cfg, W.S);
    const SP<uint16_t> cons_len = s_conslen(W.cfg, W.S);
    const SP<int32_t> cons_sum = s_cons_sum(W.cfg, W.S);
    ncons = 0;
    int fr = 0; /* current frame */
    dfs_f[0] = start.f;
    dfs_c[0] = canon(start);
    dfs_dep[0] = 0;
    dfs_rp[0] = 0;
    dfs_ra[0] = 0;
    const uint32_t end_u = (end_rp & RP_VALID) ? rp_unitig(end_rp) : 0xFFFFFFFFu;
    /* k-mers of the current path: kept in the nested-frontline hash set (idle here) so that the loop test is one probe instead of a scan
     * of the whole path; a popped k-mer leaves a tombstone (canonical k-mers are < 2^62, so ~0 - 1 is free) */
    uint64_t* pset = s_iseen(W.cfg, W.S);
    const uint32_t pcap = W.cfg.iseen_cap;
    const uint64_t TOMB = ~0ULL - 1;
    uint32_t plog_n = 0;
    const SP<uint32_t> plog = s_iseenlog(W.cfg, W.S);
    /* 0: added, 1: already on the path, 2: table full.  A tombstone met on the way is reused only after the probe sequence has
     * proven the key absent. */
    auto path_add = [&](uint64_t c) -> int {
        uint32_t h = set_hash(c, pcap);
        int64_t tomb = -1;
        for (;;) {
            const uint64_t v = pset[h];
            if (v == c + 1) return 1;
            if (v == TOMB && tomb < 0) tomb = (int64_t)h;
            if (v == 0) {
                if (tomb >= 0) { pset[(uint32_t)tomb] = c + 1; return 0; }
                if (plog_n + 2 >= pcap) return 2;
                plog[plog_n++] = h;
                pset[h] = c + 1;
                return 0;
            }
            h = (h + 1) & (pcap - 1);
        }
    };
    auto path_del = [&](uint64_t c) {
        uint32_t h = set_hash(c, pcap);
        for (;;) {
            const uint64_t v = pset[h];
            if (v == 0) return;
            if (v == c + 1) { pset[h] = TOMB; return; }
            h = (h + 1) & (pcap - 1);
        }
    };
    auto path_clear = [&]() { for (uint32_t i = 0; i < plog_n; i++) pset[plog[i]] = 0; plog_n = 0; };
    path_add(dfs_c[0]);
    bool entering = true;
    auto run = [&]() -> bool {
    for (;;) {
        if (entering) {
            entering = false;
            const int d = (int)dfs_dep[fr];
            if (traversal_depth - d < -1) return false;
            if (dfs_c[fr] == end_c) {
                if (ncons >= CONS_CAP || d > CONS_LEN) { W.status = GAP_OVF_DFS; return false; }
                for (int i = 0; i < d; i += 8) { /* eight at a time: the loads of a group do not wait for the stores of the one before */
                    uint8_t t8[8];
MTG_UNROLL
                    for (int j = 0; j < 8; j++) t8[j] = (i + j < d) ? dfs_nt[i + j] : (uint8_t)0;
MTG_UNROLL
                    for (int j = 0; j < 8; j++) if (i + j < d) cons[(size_t)ncons * CONS_LEN + i + j] = t8[j];
                }
                cons_len[ncons] = (uint16_t)d;
                cons_sum[ncons] = fr ? (int32_t)dfs_xsn[fr - 1] : 0;
                ncons++;
                dfs_mask[fr] = 0; /* return */
            } else {
                const Kmer x = make_kmer(dfs_f[fr], k);
                const uint32_t xs = fr ? dfs_xsn[fr - 1] : 0u;
                const uint32_t aux = fr ? dfs_kid[fr - 1] : 0u; /* what the parent's read told about this node */
                uint64_t rp = dfs_rp[fr];
                uint32_t ra = dfs_ra[fr], mask, kid, abx;
                if ((rp & RP_VALID) && ra >= 1u) { /* inside a unitig: the way ahead is known */
                    mask = 1u << run_next_nt(us, rp & RP_KPOS, (rp & RP_BWD) != 0, k);
                    kid = AUX_IN1;
                    abx = us.ab[rp & RP_KPOS];
                    W.lines++;
                } else if (aux & 15u) { /* inline lookahead (a junction that is in no stored unitig) */
                    mask = 1u << ((aux >> 4) & 3u);
                    kid = aux_step(aux);
                    abx = abundance(W.ix, x, W.lines);
                    rp = 0; ra = 0;
                } else {
                    Adj a = adj_right_t(W.ix.adj, x, W.mk1, W.lines);
                    RunAt r;
                    if (us.nwords && run_at(us, a, k, r, W.lines)) { rp = rp_pack(r); ra = r.ahead; abx = us.ab[r.kpos]; }
                    else { adj_resolve_la(W.ix, a, W.lines); rp = 0; ra = 0; abx = abundance(W.ix, x, W.lines); }
                    mask = a.out;
                    kid = aux_of_children(a);
                }
                dfs_mask[fr] = (uint8_t)mask;
                dfs_kid[fr] = kid;
                dfs_rp[fr] = rp;
                dfs_ra[fr] = ra;
                dfs_xsn[fr] = xs + abx;
            }
        }
        const uint32_t mask = dfs_mask[fr];
        if (mask == 0) {
            /* return to the parent; the parent re-checks the breadth limit after each child */
            if (fr == 0) return true;
            path_del(dfs_c[fr]);
            fr--;
            if (ncons > W.cfg.mono_max_breadth) return false;
            continue;
        }
        const uint32_t nt = (uint32_t)ctz4(mask);
        dfs_mask[fr] = (uint8_t)(mask & (mask - 1));
        const int d = (int)dfs_dep[fr];
        const uint64_t rp = dfs_rp[fr];
        const uint32_t ra = dfs_ra[fr];
        Kmer y;
        uint32_t t = 1u, kra = 0;
        uint64_t krp = 0;
        if ((rp & RP_VALID) && ra >= 2u && rp_unitig(rp) != end_u) {
            /* the whole stretch: the child is the unitig's end node */
            t = ra;
            if (d + (int)t > traversal_depth + 1) return false; /* the reference fails on the way */
            const bool bwd = (rp & RP_BWD) != 0;
            const uint64_t kpos = rp & RP_KPOS;
            for (uint32_t i = 0; i < t; i += 16u) {
                const uint32_t n = t - i < 16u ? t - i : 16u;
                uint32_t seq = us_peek(us.words, bwd ? kpos - 1u - i : kpos + (uint32_t)k + i, n, bwd);
                for (uint32_t j = 0; j < n; j++) { dfs_nt[(size_t)d + i + j] = (uint8_t)(seq & 3u); seq >>= 2; }
            }
            /* abundances of the t - 1 nodes passed (the node itself is in dfs_xsn[fr] already) */
            uint32_t sum = 0;
            for (uint32_t i = 1; i < t; i += 64u) {
                const uint32_t n = t - i < 64u ? t - i : 64u; /* nodes i .. i + n - 1 ahead */
                sum += us_ab_sum(us.ab, bwd ? kpos - (i + n - 1u) : kpos + i, n);
            }
            W.lines += (t >> 5) + 2u;
            dfs_xsn[fr] += sum;
            dfs_kid[fr] = AUX_IN1; /* the end node has in-degree 1; nothing is known beyond it */
            y = run_node(us, kpos, bwd, t, k);
            krp = rp_step(rp, t);
            kra = 0;
        } else {
            const Kmer x = make_kmer(dfs_f[fr], k);
            y = kmer_next(x, nt, k, W.mk);
            dfs_nt[d] = (uint8_t)nt;
            if ((rp & RP_VALID) && ra >= 1u) { krp = rp_step(rp, 1u); kra = ra - 1u; }
        }
        const uint64_t cy = canon(y);
        if (fr + 1 >= DFS_CAP || d + (int)t >= DFS_CAP) { W.status = GAP_OVF_DFS; return false; }
        const int pa = path_add(cy);
        if (pa == 1) return false; /* loop inside the bubble */
        if (pa == 2) { W.status = GAP_OVF_DFS; return false; }
        fr++;
        dfs_f[fr] = y.f;
        dfs_c[fr] = cy;
        dfs_dep[fr] = (uint32_t)d + t;
        dfs_rp[fr] = krp;
        dfs_ra[fr] = kra;
        entering = true;
    }
    };
    const bool ok = run();
    path_clear();
    return ok;
}

/* identity of src/Utils.cpp:87-189 (same routine in gatb's Traversal [MEM]) without the full matrix:
 * the traceback's choice at (i,j) only depends on scores already known when (i,j) is filled, so the
 * number of matches on the traceback path is carried forward.  Scores are multiples of 5 (exact). */
MTG_DEV_NOINLINE int nw_matches(Worker& W, SP<uint8_t> a, int na, SP<uint8_t> b, int nb)
{
    /* one packed word per cell: (score + 16384) << 10 | matches (|score| <= 10 * 512, matches <= 512); the left and diagonal
     * neighbours travel in registers, so a cell costs one load (upper neighbour), one load of b and one store.
     *
     * Exact band: the alignment "diagonal, then the length difference as end gaps" scores s0, and any alignment with g gaps scores at
     * most 5 (na + nb) - 10 g (every aligned pair a match), so an optimal alignment has g <= G = (5 (na + nb) - s0) / 10.  A path through
     * cell (i, j) needs at least |i - j| + |(i - j) - (na - nb)| gaps: cells beyond that bound lie on no optimal path and are skipped
     * (treated as minus infinity).  Cells on optimal paths keep their exact values and the traceback's equality tests can only succeed
     * towards such cells, so the match count is the one of the full matrix (src/Utils.cpp:87-189). */
    SP<int32_t> prev = s_nw(W.cfg, W.S);
    SP<int32_t> curr = prev + (CONS_LEN + 1);
    const int OFF = 16384;
    const int NEG = (OFF - 12000) << 10; /* below every reachable score */
    const int mn = na < nb ? na : nb, delta = na - nb;
    int s0 = -5 * (delta < 0 ? -delta : delta);
    for (int i = 0; i < mn; i++) s0 += (a[i] == b[i]) ? 10 : -5;
    const int G = (5 * (na + nb) - s0) / 10;
    /* feasible offsets o = i - j: |o| + |o - delta| <= G  <=>  omin <= o <= omax (G >= |delta| always holds) */
    const int omax = (G + delta) / 2, omin = -((G - delta) / 2);
    for (int j = 0; j <= nb; j++) prev[j] = (-j >= omin) ? ((-5 * j + OFF) << 10) : NEG; /* row 0: o = -j */
    for (int i = 1; i <= na; i++) {
        const uint32_t ai = a[i - 1];
        int jlo = i - omax, jhi = i - omin;
        if (jlo < 1) jlo = 1;
        if (jhi > nb) jhi = nb;
        int diag = prev[jlo - 1];
        int left;
        if (jlo == 1) { left = (i <= omax) ? ((-5 * i + OFF) << 10) : NEG; curr[0] = left; } /* column 0: o = i */
        else { left = NEG; curr[jlo - 1] = NEG; }
        for (int j = jlo; j <= jhi; j++) {
            const int up = prev[j];
            const bool eq = ai == b[j - 1];
            const int sd = (diag >> 10) + (eq ? 10 : -5), su = (up >> 10) - 5, sl = (left >> 10) - 5;
            int best = sd > su ? sd : su;
            best = best > sl ? best : sl;
            const int m = (best == sd) ? (diag & 1023) + (eq ? 1 : 0) : (best == su) ? (up & 1023) : (left & 1023);
            const int cell = (best << 10) | m;
            curr[j] = cell;
            diag = up;
            left = cell;
        }
        if (jhi < nb) curr[jhi + 1] = NEG; /* the next row reads one cell past this row's band */
        SP<int32_t> t = prev; prev = curr; curr = t;
    }
    return prev[nb] & 1023;
}

MTG_DEV bool identity_below_90(int matches, int na, int nb)
{
    int mx = na > nb ? na : nb;
#ifdef MTG_EMU
    float identity = (float)matches;
    identity /= (float)mx;
    return identity * 100 < 90;
#else
    float identity = __fdiv_rn((float)matches, (float)mx);
    return __fmul_rn(identity, 100.0f) < 90.0f;
#endif
}

/* [MEM] MonumentTraversal::validate_consensuses + most_abundant_consensus.
 * Returns the index of the chosen consensus or -1. */
MTG_DEV_NOINLINE int validate_consensuses(Worker& W, const Kmer& start, int ncons)
{
    if (ncons <= 0) return -1;
    const int k = W.k;
    const SP<uint8_t> cons = s_cons(W.cfg, W.S);
    const SP<uint16_t> cons_len = s_conslen(W.cfg, W.S);
    int mean = 0;
    for (int c = 0; c < ncons; c++) mean += cons_len[c];
    mean /= ncons;
    long long ss = 0;
    for (int c = 0; c < ncons; c++) { long long dl = (long long)cons_len[c] - mean; ss += dl * dl; }
    if (mean > W.cfg.mono_max_depth) return -1;
    if (ncons == 1 && mean > k + 1) return -1;
    /* stdev > mean/5  <=>  ss > (mean/5)^2 * n  (exact in integers) */
    long long t = mean / 5;
    if (ss > t * t * ncons) return -1;
    for (int a = 0; a < ncons; a++)
        for (int b = a + 1; b < ncons; b++) {
            int na = cons_len[a], nb = cons_len[b];
            /* The alignment is only asked whether its match count m reaches 90 % of the longer length.  An alignment with p aligned pairs, m
             * of them matches, and g gap columns scores 15 m - 5 (p + g), and p + g >= max(na, nb); the optimal one scores at least s0, the
             * score of "diagonal, then the length difference as end gaps".  So m >= (s0 + 5 max(na, nb)) / 15 on every optimal alignment,
             * whichever the traceback picks; the float test is monotone in m: when the bound passes, m passes, and no DP is needed.  (Equal
             * lengths, h substitutions: the bound is n - h; for h <= 1 it is the exact count.) */
            const SP<uint8_t> pa = cons + (size_t)a * CONS_LEN;
            const SP<uint8_t> pb = cons + (size_t)b * CONS_LEN;
            const int mn = na < nb ? na : nb, mx = na < nb ? nb : na;
            int s0 = -5 * (mx - mn);
            for (int i = 0; i < mn; i += 8) {
                uint8_t ta[8], tb[8];
MTG_UNROLL
                for (int j = 0; j < 8; j++) { ta[j] = (i + j < mn) ? pa[i + j] : (uint8_t)0; tb[j] = (i + j < mn) ? pb[i + j] : (uint8_t)0; }
MTG_UNROLL
                for (int j = 0; j < 8; j++) if (i + j < mn) s0 += (ta[j] == tb[j]) ? 10 : -5;
            }
            if (mx != mn && identity_below_90((s0 + 5 * mx <= 0) ? 0 : (s0 + 5 * mx + 14) / 15, na, nb)) {
                /* Different lengths (an insertion or deletion between two alleles) and the first bound does not pass: behind the indel the
                 * diagonal compares shifted sequences.  The alignments "diagonal up to position p, the length difference as ONE gap there,
                 * shifted diagonal behind it" are alignments too, so the best of them is a lower bound of the optimal score as well; their scores
                 * follow from one another (moving the gap one place to the right trades one shifted pair for one straight pair). */
                const SP<uint8_t> lg = na >= nb ? pa : pb, sh = na >= nb ? pb : pa; /* the longer, the shorter */
                const int dl = mx - mn;
                int sc = -5 * dl; /* p = 0: every pair shifted */
                for (int i = 0; i < mn; i++) sc += (lg[i + dl] == sh[i]) ? 10 : -5;
                int best = sc;
                for (int p = 0; p < mn; p++) {
                    sc += ((lg[p] == sh[p]) ? 10 : -5) - ((lg[p + dl] == sh[p]) ? 10 : -5);
                    best = sc > best ? sc : best;
                }
                s0 = best > s0 ? best : s0;
            }
            const int num = s0 + 5 * mx;
            const int mlb = num <= 0 ? 0 : (num + 14) / 15;
            int m = mlb;
            const bool need_dp = identity_below_90(mlb, na, nb);
            if (need_dp) m = nw_matches(W, pa, na, pb, nb);
#ifdef MTG_XCHECK /* TEST-ONLY: the bound against the alignment itself */
            {
                const int mx_ = nw_matches(W, pa, na, pb, nb);
                if (mx_ < mlb || (!need_dp && identity_below_90(mx_, na, nb))) { W.status = 0xBAD6; return -1; }
                m = mx_;
            }
#endif
            if (identity_below_90(m, na, nb)) return -1;
        }
    /* most abundant consensus: the sums come with the consensuses (all_consensuses_between) */
    const SP<int32_t> cons_sum = s_cons_sum(W.cfg, W.S);
    unsigned long best = 0;
    int chosen = -1;
    for (int c = 0; c < ncons; c++) {
        int len = cons_len[c];
        if (len == 0) continue;
        unsigned long sum = (unsigned long)cons_sum[c];
#ifdef MTG_XCHECK /* TEST-ONLY: the sum node by node, as the reference computes it */
        {
            unsigned long sum2 = 0;
            Kmer x = start;
            const SP<uint8_t> p = cons + (size_t)c * CONS_LEN;
            for (int i = 0; i < len; i++) {
                sum2 += abundance(W.ix, x, W.lines);
                x = kmer_next(x, p[i], k, W.mk);
            }
            if (sum2 != sum) { W.status = 0xBAD5; return -1; }
        }
#endif
        sum /= (unsigned long)len;
        if (sum > best) { best = sum; chosen = c; }
    }
    if (chosen < 0) return -1;
    if ((int)cons_len[chosen] > W.cfg.mono_max_depth) return -1;
    return chosen;
}

/* ---- the SNP bubble, recognised and answered without the general machinery -------------------------------------------------------
 * Pattern: the node has exactly two out-edges whose targets have in-degree 1; from there both branches are simple paths (every node
 * one in-, one out-edge, known from bucket reads and lookahead runs) that step onto the same node e after the same number L <= 40 of
 * nodes (their nucleotides may differ in more places than the first: two substitutions closer than k); no node of it is marked, and all its k-mers (with the node itself and the
 * previous node) are pairwise distinct as canonical k-mers.  On such a subgraph the reference's explore_branching is determined:
 *   find_end_of_branching  advances both branches level by level (every frontline check passes at once on in-degree 1) and stops at
 *                          depth L + 1 with the single node e;
 *   all_consensuses_between finds the two paths, in A, C, T, G order of their first nucleotide;
 *   validate_consensuses   equal lengths: identity >= 90 % is tested with the same alignment and float code; the
 *                          consensus with the larger mean abundance over [node, branch nodes] wins, the first one on a tie;
 *   marking                of the involved nodes only e is branching.
 * Anything else -- including a fingerprint collision that turns out to be a real duplicate -- returns 0 and the caller runs the
 * general code, so the fast path never decides a case it does not fully understand.  The distinctness test keeps 16-bit fingerprints
 * of the canonical k-mers in a small table (LDS); a fingerprint seen before is checked exactly by walking the branches again. */
#ifdef MTG_XCHECK /* TEST-ONLY: the explicit duplicate search that cross-checks the closed-form distinctness test on every bubble */
MTG_DEV uint32_t fp_hash(uint64_t c) { return (uint32_t)((c * 0x9E3779B97F4A7C15ULL) >> 32); }
/* layout: the slots of a lane in groups of 8 bytes (group g of lane l at g * 512 + l * 8), so that the table is cleared 8 slots at a
 * time and the lanes of a wave never fight for a bank when they do */
MTG_DEV MTG_LDS uint8_t* fp_at(const GapScratch& S, uint32_t s) { return S.fp + (s >> 3) * 512u + (s & 7u); }
MTG_DEV void fp_clear(const GapScratch& S) { for (int g = 0; g < FP_SLOTS / 8; g++) *reinterpret_cast<MTG_LDS uint64_t*>(S.fp + g * 512) = 0; }
/* 0: new; 1: a k-mer with this fingerprint was added before (8-bit fingerprints: a false alarm every hundred additions or so, settled
 * exactly by snp_seen_exactly) */
MTG_DEV int fp_add(const GapScratch& S, uint64_t c)
{
    const uint32_t h = fp_hash(c);
    const uint8_t fp = (uint8_t)((h >> 24) | 1u);
    uint32_t s = h & (FP_SLOTS - 1);
    for (;;) {
        const uint8_t v = *fp_at(S, s);
        if (v == 0) { *fp_at(S, s) = fp; return 0; }
        if (v == fp) return 1;
        s = (s + 1) & (FP_SLOTS - 1);
    }
}
#endif
/* the nucleotides of one branch, 2 bits each, in registers (up to 64) */
struct SnpSeq {
    uint64_t lo, hi;
    MTG_DEV uint32_t get(int i) const { return (uint32_t)((i < 32 ? lo >> (2 * i) : hi >> (2 * (i - 32))) & 3u); }
    MTG_DEV void set(int i, uint32_t nt) { if (i < 32) lo |= (uint64_t)nt << (2 * i); else hi |= (uint64_t)nt << (2 * (i - 32)); }
    /* nucleotides a .. a + c - 1 (c <= 16), the first one in the lowest bits */
    MTG_DEV uint32_t bits(int a, int c) const
    {
        const uint64_t w = a < 32 ? ((lo >> (2 * a)) | (a ? hi << (64 - 2 * a) : 0ull)) : (hi >> (2 * (a - 32)));
        return (uint32_t)w & (c >= 16 ? 0xFFFFFFFFu : ((1u << (2 * c)) - 1u));
    }
};
#ifdef MTG_XCHECK
/* is canonical k-mer c among: the node, prev_c, and the first `steps` nodes of each branch (the walk is replayed from the nucleotides
 * known so far), the node of branch `skip_branch` at position `skip_pos` excepted */
MTG_DEV bool snp_seen_exactly(const Worker& W, const Kmer& cur, uint64_t prev_c, const SnpSeq* seq, int steps, uint64_t c, int skip_branch, int skip_pos)
{
    if (canon(cur) == c || prev_c == c) return true;
MTG_UNROLL
    for (int br = 0; br < 2; br++) {
        Kmer x = cur;
        for (int i = 0; i < steps; i++) {
            x = kmer_next(x, seq[br].get(i), W.k, W.mk);
            if (br == skip_branch && i + 1 == skip_pos) continue;
            if (canon(x) == c) return true;
        }
    }
    return false;
}
#endif
enum { SNP_MAX_L = 62 }; /* two substitutions closer than k = 31 make branches of up to 2k - 1 nodes; the nucleotides of a branch fit two registers */

/* one of two values by a run-time flag, field by field: `arr[flag]` on a local array is a load through a computed ADDRESS, and an object whose
 * address is computed lives in private memory (scratch) with everything in it; selected values stay in registers */
MTG_DEV uint32_t pick(bool second, uint32_t a, uint32_t b) { return second ? b : a; }
MTG_DEV uint64_t pick(bool second, uint64_t a, uint64_t b) { return second ? b : a; }
MTG_DEV bool pick(bool second, bool a, bool b) { return second ? b : a; }
MTG_DEV int pick(bool second, int a, int b) { return second ? b : a; }
/* (structs: every field is loaded by value first -- `c ? b.f : a.f` on two lvalues is an lvalue, i.e. a selected address again) */
MTG_DEV Kmer pick(bool second, const Kmer& a, const Kmer& b) { Kmer r; r.f = pick(second, a.f, b.f); r.r = pick(second, a.r, b.r); return r; }
MTG_DEV SnpSeq pick(bool second, const SnpSeq& a, const SnpSeq& b) { SnpSeq r; r.lo = pick(second, a.lo, b.lo); r.hi = pick(second, a.hi, b.hi); return r; }
MTG_DEV Adj pick(bool second, const Adj& a, const Adj& b) { Adj r; r.out = pick(second, a.out, b.out); r.in = pick(second, a.in, b.in); r.la = pick(second, a.la, b.la); r.up = pick(second, a.up, b.up); return r; }
/* ---- a node with TWO out-edges, read once for all the fast forms (round 6).
 * tip_fast, indel_bulk and snp_bulk each read what they needed when they needed it: the unitig headers twice over, the branches' last nodes and
 * their junctions one after the other, the node's own abundance at the very end through two table look-ups and a read of the store -- a dozen
 * to twenty DEPENDENT rounds of reads for an insertion / deletion, eight for a SNP, and a wave pays them for every lane that stands on such a node
 * (profiles/r06_walk_stamps_before.txt: indel_bulk 2.2 calls x 101 k ticks, snp_bulk 1.7 x 69 k per lane of the indel set, where a round of
 * reads is worth 15-20 k).  Everything the three forms ask of the index and the store is a function of the two first nodes, so it is read here in
 * THREE rounds, each with all its loads in flight together:
 *   round A  the right neighbourhoods of the two first nodes (their home buckets), and the bucket of the node's own LEFT junction (the node has
 *            two out-edges: its right junction is no unitig's interior; if it lies in a stored unitig at all it is that unitig's last node and its
 *            left junction carries the pointer);
 *   round B  per branch that starts a run: the unitig's header word (its length), the 64 nucleotides behind the first node, nine words of
 *            abundance bytes; the node's own k-mer in the store and its abundance byte;
 *   round C  the right neighbourhoods of the branches' LAST nodes, for branches of at most SNP_MAX_L nodes (the nodes come out of the
 *            nucleotides by shifts, no read).
 * The forms then decide on registers; only the insertion / deletion goes on to read the meeting node's neighbourhood (round D) and the start of its
 * unitig (round E).  Nothing here decides anything: the forms test what they tested before, on the same values. */
struct Fork2 {
    Adj r[2];            /* right neighbourhoods of the branches' first nodes */
    bool run[2];         /* the branch starts a run: ra, lo, hi, ar are valid */
    RunAt ra[2];
    uint64_t lo[2], hi[2]; /* the 64 nucleotides behind the first node in walking order (past the unitig's end: whatever the store holds there) */
    uint64_t ab_at[2];   /* where the abundance bytes of the branch's ahead + 1 k-mers start: us_ab_sum(us.ab, ab_at, ahead + 1), read by the form that accepts the bubble
                            (they were read with everything else at first: thirty-six registers held across two rounds of reads and three forms for the half of the forks that use them) */
    bool has_last[2];    /* last / rl are valid: a branch of one k-mer outside a run (last = the first node, rl = r) or a run of at most SNP_MAX_L nodes */
    Kmer last[2];
    Adj rl[2];
    uint32_t ab_cur;     /* abundance of the node itself */
};
/* the node t <= 62 nodes ahead of x along the nucleotides lo | hi (first one in the lowest bits of lo) */
MTG_DEV Kmer kmer_ahead(const Kmer& x, uint64_t lo, uint64_t hi, uint32_t t, int k, uint64_t mk)
{
    Kmer y = x;
    uint32_t done = 0;
    while (done < t) {
        const uint32_t cmax = k < 15 ? (uint32_t)k : 15u, c = t - done < cmax ? t - done : cmax;
        const uint64_t w = done < 32u ? ((lo >> (2u * done)) | (done ? hi << (64u - 2u * done) : 0ull)) : (hi >> (2u * (done - 32u)));
        y = kmer_advance(y, (uint32_t)w & ((1u << (2u * c)) - 1u), c, k, mk);
        done += c;
    }
    return y;
}
MTG_DEV_NOINLINE void fork2_read(Worker& W, const Kmer& cur, const Kmer x[2], Fork2& F)
{
    const UStore& us = W.ix.us;
    const int k = W.k;
    const uint64_t mk = W.mk, mk1 = W.mk1, cmpl = 0xAAAAAAAAAAAAAAAAULL & mk;
    /* ---- round A */
    const Table& t = W.ix.adj;
    const uint64_t p = cur.f >> 2, rp = cur.r & mk1;
    const bool pf = p <= rp;
    const uint64_t Hc = mix(pf ? p : rp, t.key_bits);
    const uint64_t want_c = (Hc & ((1ULL << t.tag_bits) - 1)) << MTG_DISP_BITS;
    U64x2 qc[MTG_ADJ_SLOTS];
    {
        const U64x2* pc = reinterpret_cast<const U64x2*>(t.slots + bucket_of(Hc, t.nbuckets, t.key_bits) * (2 * MTG_ADJ_SLOTS));
MTG_UNROLL
        for (int i = 0; i < MTG_ADJ_SLOTS; i++) qc[i] = ld_table(pc + i);
    }
    adj_right2_raw(W.ix, x[0], x[1], mk1, W.lines, F.r[0], F.r[1]);
    /* the node's own place: through the pointer of its left junction, if the home bucket holds it */
    uint64_t aux_c = 0;
    bool hit_c = false;
MTG_UNROLL
    for (int i = 0; i < MTG_ADJ_SLOTS; i++) {
        const bool h = (qc[i].x >> 8) == want_c && qc[i].x != 0;
        hit_c = hit_c || h;
        aux_c |= h ? qc[i].y : 0ull;
    }
    const bool cur_ptr = hit_c && up_is(aux_c) && us.nwords != 0;
    uint64_t cur_at = 0;
    bool cur_bwd = false;
    if (cur_ptr) { /* as abundance(): the k-mer behind (forward) or before (backward) the junction */
        const uint64_t w = up_resolve(aux_c, pf);
        const uint32_t off = up_off(w);
        cur_bwd = up_bwd(w);
        cur_at = (up_hdr(w) + 1) * 32 + (cur_bwd ? off - 1u : off);
    }
    /* ---- round B */
    uint64_t cur_le = 0;
    uint32_t cur_byte = 0;
    const bool cur_ok = cur_ptr && !(cur_bwd && up_off(up_resolve(aux_c, pf)) < 1u);
    if (cur_ok) { cur_le = us_kmer_le(us.words, cur_at, k); cur_byte = us.ab[cur_at]; }
    uint32_t len_w[2] = {0, 0};
MTG_UNROLL
    for (int br = 0; br < 2; br++) {
        F.run[br] = us.nwords != 0 && F.r[br].up && popc4(F.r[br].out) == 1 && popc4(F.r[br].in) == 1;
        F.has_last[br] = false;
        F.ra[br].ahead = 0; F.ra[br].kpos = 0; F.ra[br].hdr = 0; F.ra[br].bwd = false;
        F.lo[br] = F.hi[br] = 0; F.ab_at[br] = 0;
        if (!F.run[br]) continue;
        const uint64_t hdr = up_hdr(F.r[br].up);
        const uint32_t off = up_off(F.r[br].up);
        const bool bw = up_bwd(F.r[br].up);
        const uint32_t idx = bw ? off : off - 1u;
        F.ra[br].hdr = (uint32_t)hdr;
        F.ra[br].bwd = bw;
        F.ra[br].kpos = (hdr + 1) * 32 + idx;
        len_w[br] = bw ? 0u : (uint32_t)us.words[hdr];
        const uint64_t pos = bw ? F.ra[br].kpos - 1u : F.ra[br].kpos + (uint32_t)k;
        F.lo[br] = us_peek64(us.words, pos, 32u, bw);
        F.hi[br] = (bw && off <= 32u) ? 0ull : us_peek64(us.words, bw ? pos - 32 : pos + 32, 32u, bw); /* backward: no more than off nucleotides lie before the junction */
        F.ab_at[br] = bw ? (hdr + 1) * 32 : F.ra[br].kpos;
        F.ra[br].ahead = bw ? idx : 0u;
    }
    W.lines += 2;
MTG_UNROLL
    for (int br = 0; br < 2; br++) {
        /* a first node outside a run that has one successor: the branch may be that one k-mer (indel_bulk's one-k-mer branch); a run of at most
         * SNP_MAX_L nodes: its last node out of the nucleotides.  (One assignment of each field on every path: stores to the same field from
         * several places were merged into one store through a selected address, which put the whole structure into private memory.) */
        const bool single = !F.run[br] && popc4(F.r[br].out) == 1;
        uint32_t ahead = F.ra[br].ahead;
        if (F.run[br] && !F.ra[br].bwd) ahead = len_w[br] - (uint32_t)k + 1u - 1u - (uint32_t)(F.ra[br].kpos - ((uint64_t)F.ra[br].hdr + 1) * 32);
        const bool short_run = F.run[br] && ahead + 1u <= (uint32_t)SNP_MAX_L;
        const Kmer z = kmer_ahead(x[br], F.lo[br], F.hi[br], short_run ? ahead : 0u, k, mk); /* (0 nodes ahead: the first node itself) */
        F.ra[br].ahead = ahead;
        F.has_last[br] = single || short_run;
        F.last[br].f = z.f; F.last[br].r = z.r;
    }
    /* ---- round C */
    {
        const bool n0 = F.run[0] && F.has_last[0], n1 = F.run[1] && F.has_last[1];
        Adj q0 = F.r[0], q1 = F.r[1]; /* (a branch of one k-mer: its last node is its first) */
        if (n0 && n1) adj_right2_raw(W.ix, F.last[0], F.last[1], mk1, W.lines, q0, q1);
        else if (n0) q0 = adj_right_t(t, F.last[0], mk1, W.lines);
        else if (n1) q1 = adj_right_t(t, F.last[1], mk1, W.lines);
        F.rl[0].out = q0.out; F.rl[0].in = q0.in; F.rl[0].la = q0.la; F.rl[0].up = q0.up;
        F.rl[1].out = q1.out; F.rl[1].in = q1.in; F.rl[1].la = q1.la; F.rl[1].up = q1.up;
    }
    /* the node's abundance: the byte next to its k-mer when the left junction's pointer led to it, else the ordinary look-up */
    if (cur_ok && cur_le == (cur_bwd ? (cur.f ^ cmpl) : (cur.r ^ cmpl))) { F.ab_cur = cur_byte; W.lines += 2; }
    else F.ab_cur = abundance(W.ix, cur, W.lines);
#ifdef MTG_XCHECK /* TEST-ONLY: every value against the functions the forms called before */
    {
        uint32_t l_ = 0;
        if (F.ab_cur != abundance(W.ix, cur, l_)) W.status = 0xBAE4;
        for (int br = 0; br < 2; br++) {
            RunAt q;
            const bool rn = us.nwords != 0 && run_at(us, F.r[br], k, q, l_);
            if (rn != F.run[br]) { W.status = 0xBAE4; continue; }
            if (!rn) continue;
            if (q.kpos != F.ra[br].kpos || q.ahead != F.ra[br].ahead || q.bwd != F.ra[br].bwd || q.hdr != F.ra[br].hdr) W.status = 0xBAE4;
            if (F.has_last[br]) {
                const Kmer z = run_node(us, q.kpos, q.bwd, q.ahead, k);
                const Adj a2 = adj_right_t(t, z, mk1, l_);
                if (z.f != F.last[br].f || z.r != F.last[br].r || a2.out != F.rl[br].out || a2.in != F.rl[br].in || a2.up != F.rl[br].up) W.status = 0xBAE4;
            }
        }
    }
#endif
}

/* The same pattern read off the unitig store: between the node and e each branch of a SNP bubble is exactly one stored unitig (its first
 * node follows the node's two-way junction, its last node precedes e's), so everything the step-by-step loop of snp_bubble_fast learns
 * from one read per node -- the nucleotides, that the nodes are simple, their abundances -- comes with one round of reads: the two
 * headers, then the two sequences and abundance runs together, then the two end junctions.  Strictly the pattern "both branches are
 * single stored unitigs of equal length <= SNP_MAX_L whose last nodes lead to the same node"; anything else returns false and the
 * step-by-step loop decides as before.  Nodes inside a stored unitig have one in- and one out-edge, are never marked (only branching
 * nodes are, ever), are pairwise distinct and no palindromic junction or self-complementary k-mer lies inside a unitig (mtg_dev.h:
 * us_eligible); the remaining tests of the loop (a branch node equal to the previous node, a node followed by its reverse complement at
 * the ends) are made here on the k-mers.  x[] = the first nodes of the branches, r[] = their right neighbourhoods. */
MTG_DEV_NOINLINE bool snp_bulk(Worker& W, uint64_t prev_c, const Kmer x[2], const uint32_t nt0[2], const Fork2& F, int& L, int& h, SnpSeq seq[2], unsigned long sum[2], Kmer& e, bool& hopeless)
{
    const int k = W.k;
    const Adj* const r = F.r;
MTG_UNROLL
    for (int br = 0; br < 2; br++)
        if (!(r[br].up && popc4(r[br].out) == 1 && popc4(r[br].in) == 1) || !F.run[br]) return false;
    /* the two headers (lengths), both sequences at their greatest possible length and both abundance runs came with fork2_read; lengths are applied here */
    uint64_t lo[2] = {F.lo[0], F.lo[1]}, hi[2] = {F.hi[0], F.hi[1]};
    const uint32_t left[2] = {F.ra[0].ahead, F.ra[1].ahead};
    /* a branch that stays inside one unitig for more than SNP_MAX_L nodes cannot meet the other one in time (the meeting node has two
     * in-edges: it is inside no unitig): the step-by-step loop would walk all SNP_MAX_L steps to find that out */
    if (left[0] + 1u > (uint32_t)SNP_MAX_L || left[1] + 1u > (uint32_t)SNP_MAX_L) { hopeless = true; return false; }
    if (left[0] != left[1]) {
        /* Branches of different lengths (an insertion or deletion).  The step-by-step loop can only succeed if both branches step onto the
         * same node at the same step; it gives up where the shorter branch leaves its unitig unless the junction there is simple (one in-,
         * one out-edge: a unitig that ends at a palindromic junction) -- the longer branch is still inside its own unitig then, so the node
         * stepped onto cannot be common.  That junction came with fork2_read. */
        const bool sb1 = !(left[0] < left[1]);
        const uint32_t re_out = pick(sb1, F.rl[0].out, F.rl[1].out), re_in = pick(sb1, F.rl[0].in, F.rl[1].in);
        if (popc4(re_out) != 1 || popc4(re_in) != 1) hopeless = true;
        return false;
    }
    const uint32_t m = left[0]; /* nodes of a branch behind its first one */
    uint32_t s[2];
MTG_UNROLL
    for (int br = 0; br < 2; br++) {
        lo[br] &= m >= 32u ? ~0ull : ((1ull << (2u * m)) - 1ull);
        hi[br] = m > 32u ? (hi[br] & ((1ull << (2u * (m - 32u))) - 1ull)) : 0ull;
        s[br] = 0;
    }
    Kmer z[2] = {x[0], x[1]};
    {   /* the nodes of both branches: none its own reverse complement, none followed by it, none the previous node.  Both branches in one iteration
         * and no early exit (measured late in round 5: the two branches one after the other, each a chain of up to 64 dependent iterations, were
         * 40 % of this function's time on the SNP set) */
        const uint64_t mk = W.mk;
        Kmer z0 = x[0], z1 = x[1];
        bool bad = z0.f == z0.r || z1.f == z1.r || canon(z0) == prev_c || canon(z1) == prev_c;
        uint64_t l0 = lo[0], l1 = lo[1];
        for (uint32_t i = 0; i < m; i++) {
            if (i == 32u) { l0 = hi[0]; l1 = hi[1]; }
            const Kmer y0 = kmer_next(z0, (uint32_t)(l0 & 3ull), k, mk), y1 = kmer_next(z1, (uint32_t)(l1 & 3ull), k, mk);
            l0 >>= 2; l1 >>= 2;
            bad |= (y0.f == z0.r) | (y1.f == z1.r) | (y0.f == y0.r) | (y1.f == y1.r) | (canon(y0) == prev_c) | (canon(y1) == prev_c);
            z0 = y0; z1 = y1;
        }
        if (bad) return false;
        z[0] = z0; z[1] = z1;
    }
#ifdef MTG_XCHECK
    if (z[0].f != F.last[0].f || z[1].f != F.last[1].f || !F.has_last[0] || !F.has_last[1]) { W.status = 0xBAE4; return false; }
#endif
    const Adj re[2] = {F.rl[0], F.rl[1]}; /* the branches' end junctions (fork2_read, round C) */
    /* a branch whose unitig ends in a dead end or a fork: the step-by-step loop would walk both unitigs to find just that */
    if (popc4(re[0].out) != 1 || popc4(re[1].out) != 1) { hopeless = true; return false; }
    const uint32_t ne[2] = {(uint32_t)ctz4(re[0].out), (uint32_t)ctz4(re[1].out)};
    const Kmer y0 = kmer_next(z[0], ne[0], k, W.mk), y1 = kmer_next(z[1], ne[1], k, W.mk);
    if (y0.f == z[0].r || y1.f == z[1].r) return false;
    if (y0.f != y1.f) return false; /* longer branches (several unitigs): the loop follows them */
    s[0] = us_ab_sum(W.ix.us.ab, F.ab_at[0], m + 1u); /* the pattern holds: the abundances of both branches (the reads of the two sums are in flight together) */
    s[1] = us_ab_sum(W.ix.us.ab, F.ab_at[1], m + 1u);
MTG_UNROLL
    for (int br = 0; br < 2; br++) {
        seq[br].lo = (uint64_t)nt0[br] | (lo[br] << 2);
        seq[br].hi = (lo[br] >> 62) | (hi[br] << 2);
        seq[br].set((int)m + 1, ne[br]);
        sum[br] = s[br];
    }
    const uint64_t dl = lo[0] ^ lo[1], dh = hi[0] ^ hi[1];
#ifdef MTG_EMU
    h = 1 + __builtin_popcountll((dl | (dl >> 1)) & 0x5555555555555555ULL) + __builtin_popcountll((dh | (dh >> 1)) & 0x5555555555555555ULL) + (ne[0] != ne[1] ? 1 : 0);
#else
    h = 1 + __popcll((dl | (dl >> 1)) & 0x5555555555555555ULL) + __popcll((dh | (dh >> 1)) & 0x5555555555555555ULL) + (ne[0] != ne[1] ? 1 : 0);
#endif
    L = (int)m + 1;
    e = y0;
    return true;
}

/* returns the consensus length (cons[chosen] filled) or 0: not the pattern */
MTG_DEV_NOINLINE int explore_branching(Worker& W, const Kmer& cur, uint64_t prev_c, int& chosen);
#ifdef MTG_XCHECK
/* TEST-ONLY: a fast form says "the reference finds no consensus at this node" -- the general code must say the same and mark nothing (else `code`) */
inline unsigned long* refusal_counts() { static unsigned long n[4] = {0, 0, 0, 0}; return n; } /* per code: 0xBAE0 (one successor), 0xBAE1 (a marked successor), 0xBAE2 (SNP pattern onto a marked node) */
inline void xcheck_refusal(Worker& W, const Kmer& cur, uint64_t prev_c, uint32_t code)
{
    refusal_counts()[code & 3u]++;
    const uint32_t nm0 = W.n_marked;
    int ch = -1;
    const int n2 = explore_branching(W, cur, prev_c, ch);
    if (W.status == GAP_OK && !(n2 == 0 && W.n_marked == nm0)) W.status = code;
}
#endif
#ifdef MTG_EMU
inline unsigned long& tip_fast_answers() { static unsigned long n = 0; return n; } /* TEST-ONLY: tips the fast path has answered (the tests want to see some) */
inline unsigned long& indel_bulk_answers() { static unsigned long n = 0; return n; } /* the same for the unequal-length bubbles */
inline unsigned long& merge_fast_answers() { static unsigned long n = 0; return n; } /* and for the nodes whose one successor has another predecessor */
#endif
/* ---- the TIP, recognised and answered like the SNP bubble (round 4).  Sequencing errors near the end of a read leave short dead-end branches
 * (abundance >= the cut-off three times over: rare per site, but 8-12 % of the walks of a reads-built graph meet one).  Pattern: the node has two
 * out-edges whose targets have in-degree 1; one branch T is a simple path of Lt <= k nodes that ends in a dead end -- one k-mer of no unitig, or
 * one whole stored unitig whose last node has no successor --, the other branch M is a stored unitig with at least Lt + 1 nodes behind its first
 * (so that m(Lt+1) is an interior node).  On such a subgraph the reference's explore_branching is determined:
 *   find_end_of_branching   advances both branches level by level (in-degree 1 everywhere: every frontline check passes); at depth Lt the tip's
 *                           last node is reached -- branching, so it must not be node-marked --, at depth Lt + 1 it contributes nothing and the
 *                           frontline is the single node m(Lt+1): end, depth Lt + 1;
 *   all_consensuses_between finds one path (the tip dies out), of Lt + 1 nucleotides;
 *   validate_consensuses    one consensus: accepted iff its length <= k + 1 (a "short dead-end alternative");
 *   marking                 of the involved nodes only the tip's last node is branching.
 * Distinctness needs no set, as for the SNP pattern: T and M are whole chains from their first nodes (their left junction is the node's
 * two-way junction), of different lengths, so neither is the other's reverse complement and their canonical k-mers are pairwise distinct; the
 * node (two out-edges) and its reverse complement (two in-edges) are no interior node and not the dead end (in-degree 1); the previous node leads
 * to the node, which no node of T or M does, and its reverse complement would make the node its own reverse complement (tested).  Anything else
 * returns 0 and the general code decides.  The TEST-ONLY emulation runs the general code next to every answer (0xBADE). */
MTG_DEV_NOINLINE int tip_fast(Worker& W, const Kmer& cur, uint64_t prev_c, const Kmer x[2], const uint32_t nt0[2], const Fork2& F, SnpSeq& out_seq)
{
    const Adj* const r = F.r;
    const int k = W.k;
    const UStore& us = W.ix.us;
    if (cur.f == cur.r || !us.nwords) return 0;
    bool tb1; /* the tip is branch 1 (else branch 0); the other branch M goes on */
    uint32_t Lt = 0;
    Kmer last;
    last.f = last.r = 0;
    const RunAt* const ra = F.ra;
    const bool* const has_run = F.run;
    if (r[0].out == 0 || r[1].out == 0) {
        if (r[0].out == 0 && r[1].out == 0) return 0; /* both branches die: the frontline empties, the general code says so */
        tb1 = r[0].out != 0;
        Lt = 1;
        last = pick(tb1, x[0], x[1]);
    } else {
        if (!has_run[0] || !has_run[1] || ra[0].ahead == ra[1].ahead) return 0;
        tb1 = !(ra[0].ahead < ra[1].ahead);
        const uint32_t t_ahead = pick(tb1, ra[0].ahead, ra[1].ahead);
        if (t_ahead + 1u > (uint32_t)k) return 0; /* a dead-end alternative longer than k nodes: the reference rejects it (and the contig ends) */
        if (!pick(tb1, F.has_last[0], F.has_last[1])) return 0; /* (k <= 31 < SNP_MAX_L: the last node of so short a branch is always known) */
        last = pick(tb1, F.last[0], F.last[1]);
        if (pick(tb1, F.rl[0].out, F.rl[1].out) != 0) return 0; /* the shorter branch goes on: not a tip */
        Lt = t_ahead + 1u;
    }
    const bool mb1 = !tb1;
    if (!pick(mb1, has_run[0], has_run[1]) || pick(mb1, ra[0].ahead, ra[1].ahead) < Lt + 1u) return 0; /* m(Lt+1) must be an interior node of M's unitig */
    const uint64_t c_last = canon(last);
    if (last.f == last.r || c_last == prev_c || c_last == canon(cur)) return 0;
    if (W.is_marked(c_last)) return 0; /* the bubble touches an assembled region */
    const int n = (int)Lt + 1;
    if (n > W.cfg.mono_max_depth) return 0;
    /* the consensus: M's first nucleotide and the Lt behind it, off the store */
    SnpSeq seq;
    seq.lo = pick(mb1, nt0[0], nt0[1]); seq.hi = 0;
    {   /* Lt <= k <= 31 nucleotides: they lie in the first word read behind M's first node */
        seq.lo |= (pick(mb1, F.lo[0], F.lo[1]) & ((1ull << (2u * Lt)) - 1ull)) << 2; /* positions 1 .. Lt: 2 + 2 Lt <= 64 bits */
    }
#ifdef MTG_XCHECK /* TEST-ONLY: the general code on the same node: same length, same consensus, the one mark */
    {
        const uint32_t nm0 = W.n_marked;
        int ch2 = -1;
        const int n2 = explore_branching(W, cur, prev_c, ch2);
        if (W.status == GAP_OK) {
            bool same = n2 == n && W.n_marked - nm0 == 1u && W.is_marked(c_last);
            const SP<uint8_t> p2 = s_cons(W.cfg, W.S) + (size_t)(ch2 < 0 ? 0 : ch2) * CONS_LEN;
            for (int i = 0; i < n && same; i++) same = p2[i] == (uint8_t)seq.get(i);
            if (!same) W.status = 0xBADE;
        }
        out_seq = seq;
        if (W.status == GAP_OK) tip_fast_answers()++;
        return W.status == GAP_OK ? n : 0; /* the general code has made the mark */
    }
#else
    W.mark_canon(c_last);
    out_seq = seq;
#ifdef MTG_EMU
    tip_fast_answers()++;
#endif
    return n;
#endif
}

/* ---- the INDEL bubble: two whole-unitig branches of DIFFERENT lengths that lead to the same node (round 4).  A heterozygous insertion or
 * deletion: the node has two out-edges whose targets have in-degree 1, branch S is a stored unitig (or one k-mer) of Ls nodes, branch G one of
 * Lg = Ls + delta nodes, the last nodes of both have the single successor e (two in-edges), and e starts a stored unitig with at least delta + 1
 * nodes behind it.  On such a subgraph the reference's explore_branching is determined:
 *   find_end_of_branching   both branches advance level by level; S steps onto e at depth Ls + 1 and goes on along e's unitig; when G steps
 *                           onto e (depth Lg + 1) e is already seen, the frontline is the single node c(delta) -- delta nodes past e --: end,
 *                           depth d = Lg + 1.  (The in-branching check of e walks back along G and ends at the node, which is marked: no large
 *                           in-branching; see delta = 2 below for the node that is not.)
 *   all_consensuses_between two paths to that end: through S, d nucleotides; through G, d + delta.  The depth allowance is d + 1 and a frame
 *                           deeper than d + 2 fails the whole enumeration: delta >= 3 -> explore_branching returns 0; delta = 1, 2 -> two consensuses.
 *   validate_consensuses    lengths d and d + delta (the standard-deviation rule passes), identity by the alignment's lower bounds exactly as the
 *                           general code computes them (if they do not settle it the general code is left to decide, with the alignment);
 *                           the larger integer mean abundance over [node, path nodes without the end] wins, the first (smaller first
 *                           nucleotide) on a tie;
 *   marking                 of the involved nodes only e is branching.
 * Returns the length of the chosen consensus (its nucleotides in out_seq), -1 when the reference's answer is "no consensus" (delta >= 3: the
 * contig ends here, nothing is marked), 0 when this is not the pattern.  The TEST-ONLY emulation runs the general code next to every answer (0xBADF). */
MTG_DEV_NOINLINE int indel_bulk(Worker& W, const Kmer& cur, uint64_t prev_c, const Kmer x[2], const uint32_t nt0[2], const Fork2& F, SnpSeq& out_seq)
{
    const Adj* const r = F.r;
    const int k = W.k;
    const UStore& us = W.ix.us;
    if (cur.f == cur.r || !us.nwords) return 0;
    /* the two branches: whole chains from their first nodes; a branch of one k-mer has its end junction right behind its first node */
    uint32_t Lb[2];
    Kmer lastn[2];
    const RunAt* const ra = F.ra;
    bool run[2];
    Adj rl[2];
    /* the lengths first: equal ones are the SNP forms' (the common case on a heterozygous set) */
MTG_UNROLL
    for (int br = 0; br < 2; br++) {
        run[br] = false;
        if (popc4(r[br].out) != 1) return 0;
        if (popc4(r[br].in) == 1) {
            if (!F.run[br]) return 0;
            run[br] = true;
            Lb[br] = ra[br].ahead + 1u;
            if (Lb[br] > (uint32_t)SNP_MAX_L) return 0;
        } else Lb[br] = 1u;
    }
    if (Lb[0] == Lb[1]) return 0;
MTG_UNROLL
    for (int br = 0; br < 2; br++) {
        if (!F.has_last[br]) return 0; /* (a run of at most SNP_MAX_L nodes, or one k-mer with one successor: both were read) */
        lastn[br] = F.last[br]; rl[br] = F.rl[br];
        if (popc4(rl[br].out) != 1 || popc4(rl[br].in) != 2) return 0;
        if (lastn[br].f == lastn[br].r) return 0;
    }
    const Kmer e = kmer_next(lastn[0], (uint32_t)ctz4(rl[0].out), k, W.mk);
    {
        const Kmer e1 = kmer_next(lastn[1], (uint32_t)ctz4(rl[1].out), k, W.mk);
        if (e.f != e1.f) return 0;
    }
    const uint64_t ce = canon(e);
    if (e.f == e.r || ce == canon(cur) || ce == prev_c || e.f == lastn[0].r || e.f == lastn[1].r) return 0;
    const bool sb1 = !(Lb[0] < Lb[1]); /* the short branch is branch 1 */
    const uint32_t Ls_ = pick(sb1, Lb[0], Lb[1]), Lg_ = pick(!sb1, Lb[0], Lb[1]);
    const uint32_t delta = Lg_ - Ls_, d = Lg_ + 1u;
    if ((int)(d + delta) > SNP_MAX_L || (int)d > W.cfg.mono_max_depth) return 0;
    /* e's continuation: a stored unitig that starts with e, delta + 1 nodes at least behind it (round D: e's neighbourhood; round E, all in flight
     * together: the unitig's length, the nucleotide behind c(1), the abundance bytes of e and c(1) -- e is solid (an edge of the exact table leads to
     * it) and the k-mer before a simple junction of a stored unitig is unique, so e's byte is the one at its place) */
    const Adj re = adj_right_t(W.ix.adj, e, W.mk1, W.lines);
    RunAt rae;
    if (!(re.up && popc4(re.out) == 1 && popc4(re.in) == 1)) return 0;
    uint32_t ab_e = 0, ab_c1 = 0, nt_c2 = 0;
    {
        const uint64_t hdr = up_hdr(re.up);
        const uint32_t off = up_off(re.up);
        rae.hdr = (uint32_t)hdr; rae.bwd = up_bwd(re.up);
        const uint32_t idx = rae.bwd ? off : off - 1u;
        rae.kpos = (hdr + 1) * 32 + idx;
        const uint32_t lw = rae.bwd ? 0u : (uint32_t)us.words[hdr];
        /* the reads of what delta <= 2 needs, issued before the length is known (one or two nodes past e: inside the padded store whatever the length) */
        ab_e = us.ab[rae.kpos];
        ab_c1 = us.ab[rae.bwd ? rae.kpos - 1u : rae.kpos + 1u];
        nt_c2 = us_peek(us.words, rae.bwd ? rae.kpos - 2u : rae.kpos + (uint32_t)k + 1u, 1u, rae.bwd);
        rae.ahead = rae.bwd ? idx : lw - (uint32_t)k + 1u - 1u - idx;
        W.lines++;
    }
    if (rae.ahead < delta + 1u) return 0;
    /* e marked: the bubble touches an assembled region -- the frontline gives up when S steps onto e, whatever the rest looks like */
    const bool e_marked = W.is_marked(ce);
    /* delta = 2: when e is checked for in-branching its second predecessor (G's last node) has not been reached yet, and a frontline walks back
     * from it along G; it stops at the node BECAUSE the node is marked (every branching node the walk has stepped onto is) -- the first node of
     * a contig is not, the frontline goes on behind it and what it meets there joins the involved nodes: the general code's business */
    if (!e_marked && delta == 2u && !W.is_marked(canon(cur))) return 0;
    int answer;
    SnpSeq seq[2];
    unsigned long sum[2] = {0, 0};
    int len[2] = {0, 0};
    if (delta >= 3u || e_marked) answer = -1;
    else {
        /* the nucleotides: first one, the branch's own (off the store), the step onto e, delta of e's unitig */
        const uint32_t ne_first = (uint32_t)ctz4(re.out);
        uint32_t tail = ne_first; /* c(1) .. c(delta): at most two nucleotides */
        if (delta == 2u) tail |= nt_c2 << 2; /* the nucleotide behind c(1) */
        /* abundances of the nodes both paths share: the node, e, c(1) .. c(delta - 1) */
        unsigned long shared = (unsigned long)F.ab_cur + ab_e;
        if (delta == 2u) shared += ab_c1;
#ifdef MTG_XCHECK /* TEST-ONLY: the bytes against the look-ups they replace */
        {
            uint32_t l_ = 0;
            if (ab_e != abundance(W.ix, e, l_) || (delta == 2u && (ab_c1 != abundance(W.ix, run_node(us, rae.kpos, rae.bwd, 1u, k), l_) ||
                                                                    nt_c2 != us_peek(us.words, rae.bwd ? rae.kpos - 2u : rae.kpos + (uint32_t)k + 1u, 1u, rae.bwd)))) { W.status = 0xBAE4; return 0; }
        }
#endif
MTG_UNROLL
        for (int br = 0; br < 2; br++) {
            seq[br].lo = nt0[br]; seq[br].hi = 0;
            int n = 1;
            unsigned long sb_ = 0;
            if (run[br]) {
                const uint32_t m = ra[br].ahead; /* nodes behind the first: m + 1 <= SNP_MAX_L */
                const uint64_t l = m >= 32u ? F.lo[br] : (F.lo[br] & ((1ull << (2u * m)) - 1ull));
                const uint64_t h = m > 32u ? (F.hi[br] & ((1ull << (2u * (m - 32u))) - 1ull)) : 0ull;
                seq[br].lo |= l << 2;
                seq[br].hi = (l >> 62) | (h << 2);
                n += (int)m;
                /* abundance bytes of the branch's m + 1 k-mers: they start at kpos (forward) or kpos - m (backward) */
                sb_ = us_ab_sum(us.ab, F.ab_at[br], m + 1u);
#ifdef MTG_XCHECK
                { const uint64_t a0 = ra[br].bwd ? ra[br].kpos - m : ra[br].kpos; if (a0 != F.ab_at[br]) { W.status = 0xBAE4; return 0; } }
#endif
            } else sb_ = abundance(W.ix, x[br], W.lines);
            seq[br].set(n++, (uint32_t)ctz4(rl[br].out));
            for (uint32_t j = 0; j < delta; j++) /* both paths end at c(delta) */ seq[br].set(n++, (tail >> (2u * j)) & 3u);
            len[br] = n;
            sum[br] = shared + sb_;
        }
        W.lines += 4;
        if (pick(sb1, len[0], len[1]) != (int)d || pick(!sb1, len[0], len[1]) != (int)(d + delta)) return 0;
        /* validate_consensuses on the two strings (the same integers) */
        const int na = len[0], nb = len[1], mn = na < nb ? na : nb, mx = na < nb ? nb : na;
        int mean = (na + nb) / 2;
        if (mean > W.cfg.mono_max_depth) answer = -1;
        else {
            long long ss = 0;
            { const long long d0 = (long long)na - mean, d1 = (long long)nb - mean; ss = d0 * d0 + d1 * d1; }
            const long long t5 = mean / 5;
            if (ss > t5 * t5 * 2) answer = -1;
            else {
                const SnpSeq lg = pick(!(na >= nb), seq[0], seq[1]);
                const SnpSeq sh = pick(na >= nb, seq[0], seq[1]);
                int s0 = -5 * (mx - mn);
                for (int i = 0; i < mn; i++) s0 += (seq[0].get(i) == seq[1].get(i)) ? 10 : -5;
                if (identity_below_90((s0 + 5 * mx <= 0) ? 0 : (s0 + 5 * mx + 14) / 15, na, nb)) {
                    const int dl = mx - mn;
                    int sc = -5 * dl;
                    for (int i = 0; i < mn; i++) sc += (lg.get(i + dl) == sh.get(i)) ? 10 : -5;
                    int best = sc;
                    for (int p2 = 0; p2 < mn; p2++) {
                        sc += ((lg.get(p2) == sh.get(p2)) ? 10 : -5) - ((lg.get(p2 + dl) == sh.get(p2)) ? 10 : -5);
                        best = sc > best ? sc : best;
                    }
                    s0 = best > s0 ? best : s0;
                }
                const int num = s0 + 5 * mx;
                const int mlb = num <= 0 ? 0 : (num + 14) / 15;
                if (identity_below_90(mlb, na, nb)) return 0; /* the bound does not settle it: the general code runs the alignment */
                unsigned long best = 0;
                int ch = -1;
                for (int c = 0; c < 2; c++) { const unsigned long m2 = sum[c] / (unsigned long)len[c]; if (m2 > best) { best = m2; ch = c; } }
                const int len_ch = pick(ch == 1, len[0], len[1]);
                if (ch < 0 || len_ch > W.cfg.mono_max_depth) answer = -1;
                else { answer = len_ch; out_seq = pick(ch == 1, seq[0], seq[1]); }
            }
        }
    }
#ifdef MTG_XCHECK /* TEST-ONLY: the general code on the same node */
    {
        const uint32_t nm0 = W.n_marked;
        int ch2 = -1;
        const int n2 = explore_branching(W, cur, prev_c, ch2);
        if (W.status == GAP_OK) {
            bool same;
            if (answer < 0) same = n2 == 0 && W.n_marked == nm0;
            else {
                same = n2 == answer && W.n_marked - nm0 == 1u && W.is_marked(ce);
                const SP<uint8_t> p2 = s_cons(W.cfg, W.S) + (size_t)(ch2 < 0 ? 0 : ch2) * CONS_LEN;
                for (int i = 0; i < answer && same; i++) same = p2[i] == (uint8_t)out_seq.get(i);
            }
            if (!same) W.status = 0xBADF;
        }
        if (W.status != GAP_OK) return 0;
        indel_bulk_answers()++;
        return answer; /* the general code has made the mark */
    }
#else
    if (answer > 0) W.mark_canon(ce);
#ifdef MTG_EMU
    indel_bulk_answers()++;
#endif
    return answer;
#endif
}

/* ---- A node with ONE successor that has another predecessor as well (a side branch running into a junction: every contig that starts on an
 * allele of a refused bubble ends up here, every tip walked from its far end, every read-error branch).  The simple-path rule stops at such a
 * node (in-degree of the successor > 1) and the reference calls explore_branching, which on this subgraph is determined (default
 * end rule; the caller has excluded the other):
 *   find_end_of_branching   the frontline {node} moves to {e} at depth 1 -- unless e is the node's own reverse complement or the previous node
 *                           (already seen: the frontline empties) or a marked node (the bubble touches an assembled region: the second contig to
 *                           arrive at a junction), which all mean "no consensus" -- and a frontline of one node is the end: depth 1, no
 *                           in-branching check is ever run (checks start at depth 1, on the way to depth 2);
 *   all_consensuses_between the one nucleotide;
 *   validate_consensuses    one consensus of length 1: passes; its integer mean abundance is the node's own abundance, which must beat 0;
 *   marking                 the involved nodes are {e}: branching (two in-edges), marked.
 * Returns 1 (the nucleotide in out_seq), -1 ("no consensus": the contig ends here, nothing is marked, nothing parked) or 0 (not this pattern).  The TEST-ONLY
 * emulation runs the general code next to every answer (0xBAE0). */
MTG_DEV_NOINLINE int merge_fast(Worker& W, const Kmer& cur, uint64_t prev_c, const Adj& a, SnpSeq& out_seq)
{
    if (popc4(a.out) != 1) return 0;
    const uint32_t nt = (uint32_t)ctz4(a.out);
    const Kmer e = kmer_next(cur, nt, W.k, W.mk);
    const uint64_t ce = canon(e);
    if (ce == canon(cur) || ce == prev_c || W.is_marked(ce)) { /* seen already, or marked: the frontline has nowhere to go or gives up -- "no consensus", nothing marked */
#ifdef MTG_XCHECK
        xcheck_refusal(W, cur, prev_c, 0xBAE0);
        if (W.status) return 0;
#endif
#ifdef MTG_EMU
        merge_fast_answers()++;
#endif
        return -1;
    }
    if (abundance(W.ix, cur, W.lines) == 0) return 0;
    out_seq.lo = nt; out_seq.hi = 0;
#ifdef MTG_XCHECK
    {
        const uint32_t nm0 = W.n_marked;
        int ch2 = -1;
        const int n2 = explore_branching(W, cur, prev_c, ch2);
        if (W.status == GAP_OK && !(n2 == 1 && W.n_marked - nm0 == 1u && W.is_marked(ce) && s_cons(W.cfg, W.S)[(size_t)(ch2 < 0 ? 0 : ch2) * CONS_LEN] == (uint8_t)nt)) W.status = 0xBAE0;
        if (W.status != GAP_OK) return 0;
        merge_fast_answers()++;
        return 1; /* the general code has made the mark */
    }
#else
    W.mark_canon(ce);
#ifdef MTG_EMU
    merge_fast_answers()++;
#endif
    return 1;
#endif
}

MTG_DEV_NOINLINE int snp_bubble_fast(Worker& W, const Kmer& cur, uint64_t prev_c, const Adj& a, int& chosen, SnpSeq& chosen_seq)
{
    if (!W.S.snp_fast || W.cfg.end_rule_nonbranching) return 0;
    if (popc4(a.out) == 1) {
        MTG_F0(f_m);
        const int mn = merge_fast(W, cur, prev_c, a, chosen_seq);
        MTG_F1(W, f_m, 0);
        if (mn > 0) { chosen = 0; MTG_COUNT(W, 14); }
        return mn;
    }
    {   /* a successor that is marked (and not seen already: the node's reverse complement, the previous node): the reference's frontline gives up on
         * its first step, whatever else leaves the node -- "no consensus", nothing marked */
        const uint64_t cc = canon(cur);
        MTG_F0(f_ms);
        for (uint32_t em = a.out & 15u; em; em &= em - 1u) {
            const uint64_t cs = canon(kmer_next(cur, low_nt(em), W.k, W.mk));
            if (cs == cc || cs == prev_c || !W.is_marked(cs)) continue;
#ifdef MTG_XCHECK
            xcheck_refusal(W, cur, prev_c, 0xBAE1);
            if (W.status) return 0;
#endif
            return -1;
        }
        MTG_F1(W, f_ms, 2);
    }
    if (!(popc4(a.out) == 2 && popc4(a.in) == 1)) return 0;
    /* Pairwise distinctness of the canonical k-mers without a set.  Branch nodes have one in- and one out-edge, so do their reverse
     * complements; the node has two out-edges and e two in-edges, so rc(node) has two in-edges and rc(e) two out-edges: neither can be
     * a branch node, nor can the node itself.  Two branch nodes u, v = rc(u) of ONE branch force, by following the unique edges
     * inwards, a node that is its own reverse complement or whose successor is; of DIFFERENT branches they force, the same way,
     * rc(node) or rc(e) onto a branch (impossible) or e = rc(node).  Equal forward k-mers at different places force the node onto a
     * branch (impossible).  What remains is tested per node below: self-rc, successor = rc, the previous node, and e against the
     * node.  (The TEST-ONLY emulation keeps the fingerprint table and checks this reasoning against an explicit search on every bubble.) */
    const int k = W.k;
    const uint32_t nt0[2] = {(uint32_t)ctz4(a.out), (uint32_t)ctz4(a.out & (a.out - 1))};
    Kmer x[2];
    uint32_t aux[2] = {AUX_IN1, AUX_IN1};
    SnpSeq seq[2];
MTG_UNROLL
    for (int br = 0; br < 2; br++) { x[br] = kmer_next(cur, nt0[br], k, W.mk); seq[br].lo = nt0[br]; seq[br].hi = 0; }
#ifdef MTG_XCHECK
    bool dup_exact = false; /* cross-check of the closed-form test */
    fp_clear(W.S);
    fp_add(W.S, canon(cur));
    fp_add(W.S, prev_c);
#endif
    int L = 0, h = 1; /* h: positions at which the two consensuses differ */
    /* abundances for the choice between the consensuses (mean over the node and the L nodes of a branch, validate_consensuses): the bucket
     * of a node is requested when the node is reached and read one step later, behind the arithmetic of the step */
    unsigned long sum[2] = {0, 0};
    AbPending pend[2], pend_cur;
    bool have_pend = false;
    if (!W.ix.us.nwords) ab_issue(W.ix, canon(cur), pend_cur); /* with a unitig store the node's abundance comes with fork2_read */
    /* Marked nodes.  Only branching nodes are ever marked (every mark_canon is behind an in-/out-degree test) and the nodes of the two branches
     * are tested simple below, so the one node of the pattern that can be marked is the meeting node e: ONE lookup at the end.  (Rounds 2-3 kept
     * up to four "perhaps marked" branch nodes per bubble from the register signature and gave the bubble to the general code at the fifth: a
     * thousandth of the heterozygous set's walks parked for that alone.  The TEST-ONLY emulation still looks every branch node up: 0xBAE3.) */
#ifdef MTG_XCHECK
    bool branch_node_marked = false;
#endif
    /* the bulk form first (both branches read off the unitig store); the device takes its answer, the TEST-ONLY emulation runs the loop
     * as well and compares (status 0xBAD2: different answers, 0xBAD3: the loop rejected what the bulk form accepted) */
    Adj r1[2];
    bool have_r1 = false, bulk_ok = false, hopeless = false;
    int bL = 0, bh = 0;
    SnpSeq bseq[2];
    unsigned long bsum[2] = {0, 0};
    Kmer be;
    be.f = be.r = 0;
    bseq[0].lo = bseq[0].hi = bseq[1].lo = bseq[1].hi = 0;
    Fork2 F;
    F.ab_cur = 0;
    if (W.ix.us.nwords) {
        MTG_F0(f_r1);
        fork2_read(W, cur, x, F); /* everything the three forms below ask of the index and the store, in three rounds of reads */
        r1[0] = F.r[0]; r1[1] = F.r[1];
        MTG_F1(W, f_r1, 4);
        if (W.status) return 0;
        have_r1 = true;
        {   /* a tip first (one branch dies within k nodes, the other goes on): answered on the spot */
            MTG_F0(f_t);
            const int tn = tip_fast(W, cur, prev_c, x, nt0, F, chosen_seq);
            MTG_F1(W, f_t, 6);
            if (W.status) return 0;
            if (tn > 0) { chosen = 0; MTG_COUNT(W, 14); return tn; }
        }
        {   /* two unitig branches of different lengths onto one node: an insertion / deletion, answered or refused on the spot (-1: no consensus) */
            MTG_F0(f_i);
            const int in_ = indel_bulk(W, cur, prev_c, x, nt0, F, chosen_seq);
            MTG_F1(W, f_i, 8);
            if (W.status) return 0;
            if (in_ != 0) { chosen = 0; return in_; }
        }
        MTG_F0(f_s);
        bulk_ok = snp_bulk(W, prev_c, x, nt0, F, bL, bh, bseq, bsum, be, hopeless);
        MTG_F1(W, f_s, 10);
#ifndef MTG_XCHECK
        if (hopeless) return 0;
#endif
#ifdef MTG_TRACE_BULK
        fprintf(stderr, "SNP bulk %d L=%d\n", (int)bulk_ok, bL);
#endif
    }
#ifdef MTG_XCHECK
    const bool run_loop = true;
#define MTG_SNP_FAIL(code) do { if (bulk_ok && (code) != 1) W.status = 0xBAD3; return 0; } while (0)
#else
    const bool run_loop = !bulk_ok;
#if defined(MTG_STAMPS) && !defined(MTG_EMU)
#define MTG_SNP_FAIL(code) do { W.form_acc[12] += __builtin_amdgcn_s_memtime() - f_loop0; return 0; } while (0)
#else
#define MTG_SNP_FAIL(code) return 0
#endif
#endif
    if (!run_loop) { L = bL; h = bh; seq[0] = bseq[0]; seq[1] = bseq[1]; sum[0] = bsum[0]; sum[1] = bsum[1]; x[0] = be; }
    else if (have_r1) { adj_resolve_la(W.ix, r1[0], W.lines); adj_resolve_la(W.ix, r1[1], W.lines); }
#if defined(MTG_STAMPS) && !defined(MTG_EMU)
    const unsigned long long f_loop0 = __builtin_amdgcn_s_memtime();
    if (run_loop) W.form_acc[13] += 1ull;
#endif
    for (int step = 1; run_loop && step <= SNP_MAX_L; step++) {
#if defined(MTG_STAMPS) && !defined(MTG_EMU)
        W.form_acc[14] += 1ull;
#endif
        /* the nodes at position `step` of both branches: unmarked, new; their single out-edges (both bucket reads in flight together) */
        uint32_t nt[2];
        Adj r[2];
        const bool need0 = !(aux[0] & 15u), need1 = !(aux[1] & 15u);
        if (step == 1 && have_r1) { r[0] = r1[0]; r[1] = r1[1]; }
        else if (need0 && need1) adj_right2(W.ix, x[0], x[1], W.mk1, W.lines, r[0], r[1]);
        else if (need0) r[0] = adj_right(W.ix, x[0], W.mk1, W.lines);
        else if (need1) r[1] = adj_right(W.ix, x[1], W.mk1, W.lines);
        if (have_pend) { sum[0] += ab_finish(W.ix, pend[0], W.lines); sum[1] += ab_finish(W.ix, pend[1], W.lines); }
MTG_UNROLL
        for (int br = 0; br < 2; br++) { /* unrolled: the per-branch state must stay in registers */
            const uint64_t c = canon(x[br]);
            ab_issue(W.ix, c, pend[br]);
#ifdef MTG_XCHECK
            if (W.is_marked(c)) branch_node_marked = true;
#endif
#ifdef MTG_XCHECK
            if (fp_add(W.S, c) && snp_seen_exactly(W, cur, prev_c, seq, step, c, br, step)) dup_exact = true;
#endif
            if (x[br].f == x[br].r || c == prev_c) MTG_SNP_FAIL(2);
            if (aux[br] & 15u) { nt[br] = (aux[br] >> 4) & 3u; aux[br] = aux_step(aux[br]); }
            else {
                if (popc4(r[br].out) != 1) MTG_SNP_FAIL(2); /* dead end or a branching inside the bubble */
                nt[br] = (uint32_t)ctz4(r[br].out);
                aux[br] = aux_of_children(r[br]);
            }
        }
        have_pend = true;
        if (W.status) return 0;
        h += nt[0] != nt[1];
        seq[0].set(step, nt[0]);
        seq[1].set(step, nt[1]);
        const Kmer y0 = kmer_next(x[0], nt[0], k, W.mk), y1 = kmer_next(x[1], nt[1], k, W.mk);
        if (y0.f == x[0].r || y1.f == x[1].r) MTG_SNP_FAIL(2); /* a node followed by its own reverse complement */
        if (y0.f == y1.f) { L = step; x[0] = y0; break; } /* the branches meet: x[0] = e */
        if (!(aux[0] & AUX_IN1) || !(aux[1] & AUX_IN1)) MTG_SNP_FAIL(2); /* a node with another way in: the frontline check would have work to do */
        x[0] = y0;
        x[1] = y1;
    }
    if (L == 0) MTG_SNP_FAIL(2);
#if defined(MTG_STAMPS) && !defined(MTG_EMU)
    if (run_loop) W.form_acc[12] += __builtin_amdgcn_s_memtime() - f_loop0;
#endif
#ifdef MTG_XCHECK
    if (hopeless) { W.status = 0xBAD8; return 0; } /* the bulk form gave up on a bubble the loop answers */
#endif
#undef MTG_SNP_FAIL
    const Kmer e = x[0];
    const uint64_t ce = canon(e);
#ifdef MTG_XCHECK
    if (branch_node_marked) { W.status = 0xBAE3; return 0; } /* a simple node in the marked set: the reasoning above is wrong */
#endif
    if (W.is_marked(ce)) {
        /* The bubble touches an assembled region: the reference's frontline, which has come as far as this loop has (same nodes, same order,
         * no in-branching to check), gives up when it steps onto e: "no consensus", nothing marked -- the contig ends here, without a park. */
#ifdef MTG_XCHECK
        xcheck_refusal(W, cur, prev_c, 0xBAE2);
        if (W.status) return 0;
#endif
        return -1;
    }
#ifdef MTG_XCHECK
    if (fp_add(W.S, ce) && snp_seen_exactly(W, cur, prev_c, seq, L, ce, -1, 0)) dup_exact = true;
    if (dup_exact && !(ce == canon(cur) || ce == prev_c)) { W.status = 0xBAD0; return 0; } /* the closed-form test missed a duplicate */
#endif
    if (ce == canon(cur) || ce == prev_c) return 0; /* e is the node, its reverse complement or the previous node */
    const int n = L + 1;
    if (n > W.cfg.mono_max_depth) return 0;
    /* one substitution: the diagonal is the unique optimal alignment (see validate_consensuses); more: the exact banded alignment, which
     * wants the consensus strings in the work area (the caller takes the chosen one from registers) */
    int matches = n - h;
    if (bulk_ok) MTG_COUNT(W, 13);
    if (h >= 2) {
#if defined(MTG_STAMPS) && !defined(MTG_EMU)
        unsigned long long* stamp_acc = W.stamp_acc;
#endif
        /* every optimal alignment of two sequences of length n that differ in h places has at least n - h matches (validate_consensuses):
         * when that many pass the 90 % test, so does the traceback's count, whatever it is */
        bool need_dp = identity_below_90(n - h, n, n);
        if (need_dp && W.no_dp) return 0; /* the general code answers this bubble the same way, with the alignment */
#ifdef MTG_XCHECK
        const bool bound_says_pass = !need_dp;
        need_dp = true;
#endif
        if (need_dp) {
            MTG_COUNT(W, 11);
            MTG_T0(t_nw);
            const SP<uint8_t> cons = s_cons(W.cfg, W.S);
MTG_UNROLL
            for (int br = 0; br < 2; br++)
                for (int i = 0; i < n; i++) cons[(size_t)br * CONS_LEN + i] = (uint8_t)seq[br].get(i);
            matches = nw_matches(W, cons, n, cons + (size_t)CONS_LEN, n);
            MTG_T1(t_nw, 12);
#ifdef MTG_XCHECK
            if (matches < n - h || (bound_says_pass && identity_below_90(matches, n, n))) { W.status = 0xBAD6; return 0; }
#endif
        }
    }
    if (identity_below_90(matches, n, n)) return 0;
    /* most abundant consensus: the last step's buckets and the node's own */
    if (have_pend) { sum[0] += ab_finish(W.ix, pend[0], W.lines); sum[1] += ab_finish(W.ix, pend[1], W.lines); }
#ifdef MTG_XCHECK
    if (bulk_ok && (L != bL || h != bh || e.f != be.f || seq[0].lo != bseq[0].lo || seq[0].hi != bseq[0].hi || seq[1].lo != bseq[1].lo || seq[1].hi != bseq[1].hi ||
                    sum[0] != bsum[0] || sum[1] != bsum[1])) { W.status = 0xBAD2; return 0; }
#endif
    {
        const uint32_t a0 = W.ix.us.nwords ? F.ab_cur : ab_finish(W.ix, pend_cur, W.lines);
        sum[0] += a0;
        sum[1] += a0;
    }
    sum[0] /= (unsigned long)n;
    sum[1] /= (unsigned long)n;
    unsigned long best = 0;
    chosen = -1;
    for (int c = 0; c < 2; c++) if (sum[c] > best) { best = sum[c]; chosen = c; }
    if (chosen < 0) return 0;
    chosen_seq = pick(chosen == 1, seq[0], seq[1]);
    W.mark_canon(ce); /* e has two in-edges: the one branching node among the involved ones */
    MTG_COUNT(W, 14);
    return n;
}

/* [MEM] MonumentTraversal::explore_branching.  On success the consensus sits in
 * S.cons[chosen] and its length is returned; 0 on failure. */
MTG_DEV_NOINLINE int explore_branching(Worker& W, const Kmer& cur, uint64_t prev_c, int& chosen)
{
    W.n_inv = 0;
    uint64_t end_f = 0, end_rp = 0;
#if defined(MTG_STAMPS) && !defined(MTG_EMU)
    unsigned long long* stamp_acc = W.stamp_acc;
#endif
    MTG_COUNT(W, 10);
    MTG_T0(t_fe);
    int d = find_end_of_branching(W, cur, prev_c, end_f, end_rp);
    W.seen_clear();
    MTG_T1(t_fe, 2);
#ifdef MTG_TRACE
    fprintf(stderr, "EB cur=%llx d=%d end=%llx ninv=%u\n", (unsigned long long)cur.f, d, (unsigned long long)end_f, W.n_inv);
#endif
    if (!d || W.status) return 0;
    Kmer e = make_kmer(end_f, W.k);
    int ncons = 0;
    MTG_T0(t_dfs);
    const bool okc = all_consensuses_between(W, cur, canon(e), end_rp, d + 1, ncons);
    MTG_T1(t_dfs, 3);
#ifdef MTG_XCHECK /* TEST-ONLY: the enumeration node by node must give the same consensuses in the same order */
    if (!W.status) {
        auto digest = [&](int n) -> uint64_t {
            uint64_t h = 0x9E3779B97F4A7C15ULL * (uint64_t)(n + 1);
            const SP<uint8_t> cons = s_cons(W.cfg, W.S);
            const SP<uint16_t> cons_len = s_conslen(W.cfg, W.S);
            for (int c = 0; c < n; c++) {
                h = (h ^ cons_len[c]) * 0x100000001B3ULL;
                for (int i = 0; i < (int)cons_len[c]; i++) h = (h ^ cons[(size_t)c * CONS_LEN + i]) * 0x100000001B3ULL;
            }
            return h;
        };
        const uint64_t h1 = okc ? digest(ncons) : 0;
        int ncons2 = 0;
        const bool okc2 = all_consensuses_between_nodes(W, cur, canon(e), d + 1, ncons2);
        if (W.status) return 0;
        if (okc != okc2 || (okc && (ncons != ncons2 || h1 != digest(ncons2)))) { W.status = 0xBAD4; return 0; }
    }
#endif
    if (!okc) return 0;
    MTG_T0(t_val);
    chosen = validate_consensuses(W, cur, ncons);
    MTG_T1(t_val, 4);
#ifdef MTG_TRACE
    fprintf(stderr, "   ncons=%d chosen=%d\n", ncons, chosen);
#endif
    if (chosen < 0) return 0;
    /* mark all involved extensions (only the node bit of branching k-mers is ever read back) */
    const SP<uint64_t> inv = s_inv(W.cfg, W.S);
    MTG_T0(t_mi);
    for (uint32_t i = 0; i < W.n_inv; i++) {
        const uint64_t e = inv[i];
        if (e & INV_SIMPLE) continue; /* one in-edge, one out-edge: not branching */
        Kmer x;
        x.f = e & INV_KMER;
        x.r = revcomp(x.f, W.k);
        if (W.is_branching(x)) W.mark_canon(x.f);
    }
    MTG_T1(t_mi, 5);
    return s_conslen(W.cfg, W.S)[chosen];
}

/* result of one gap */
struct GapOut {
    uint32_t n_contigs;
    uint32_t status;
    uint32_t lines;
    uint32_t total_nt;
    uint32_t n_words; /* words of the contig arena in use */
    uint32_t store_reads; /* runs taken from the unitig store (one header read each) */
    uint32_t run_nt;      /* nucleotides of the contigs that came out of the store */
    uint32_t n_cmds;      /* deferred copies left in s_cmd */
    uint32_t copy_words;  /* words of the contig arena they will fill */
};

/* the swf pattern R (gapFillFromSource's targetSequence, src/Filler.cpp:884): 2-bit packed, 32 nt per
 * word, rlen nts.  bkpt mode: the target k-mer; contig mode: the concatenation of all targets
 * (src/Filler.cpp:530,537).  r0 = its first k-mer (valid when rlen >= k). */
struct SwfPattern {
    const uint64_t* words;
    uint32_t rlen;
    uint64_t r0;
};
MTG_DEV uint32_t packed_nt(const uint64_t* w, uint32_t i) { return (uint32_t)(w[i >> 5] >> (2 * (i & 31))) & 3u; }

/* does the contig contain R literally (strstr at IterativeExtensions [MEM]) */
MTG_DEV_NOINLINE bool contig_contains(const uint64_t* wd, uint32_t clen, const SwfPattern& R)
{
    if (R.rlen == 0) return true;
    if (R.rlen > clen) return false;
    for (uint32_t p = 0; p + R.rlen <= clen; p++) {
        bool ok = true;
        for (uint32_t j = 0; j < R.rlen && ok; j++) ok = packed_nt(wd, p + j) == packed_nt(R.words, j);
        if (ok) return true;
    }
    return false;
}

} // namespace mtg
#include "mtg_bubble.h"
namespace mtg {

/* [MEM] IterativeExtensions::construct_linear_seqs + Traversal::traverse (A.5).
 *
 * Three forms of the same walk:
 *   WALK_CLASSIC  everything by the one lane, the general bubble code from HBM scratch (the reference form; tests, fallback);
 *   WALK_PARK     the walk kernel: simple paths and the strict SNP pattern; at any other branching node the walk's state goes to the gap's
 *                 WalkSave and the gap comes back GAP_PARKED;
 *   WALK_FINISH   the finishing kernel: G lanes resume ONE parked gap.  The walk itself is run by all G lanes with the same values (a load
 *                 is one request per group, the stores write the same bytes), so that the lanes arrive together at every branching node,
 *                 which the group then resolves from its LDS area (mtg_bubble.h); what does not fit there runs the general code;
 *   WALK_SIMPLE   the light walk kernel (round 5): simple paths ONLY -- it parks at every branching node, none of the bubble code is in its
 *                 call graph (a tenth of WALK_PARK's 260 KB of instructions, a fraction of its registers).  For launches on data where
 *                 hardly any walk meets a branching node (a haploid donor); the parked few are resumed by the finishing kernel. */
enum { WALK_CLASSIC = 0, WALK_PARK = 1, WALK_FINISH = 2, WALK_SIMPLE = 3 };

#ifdef MTG_EMU /* TEST-ONLY: how the group form answered (MTG_EMU_COOP_STATS=1 prints the tally when the process ends) */
} // namespace mtg
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
namespace mtg {
/* TEST-ONLY (MTG_EMU_PARK_STATS=1): what the nodes look like at which the walk kernel parks a gap -- a tally by shape, printed when the process
 * ends (scripts/r6_park_reasons.py: which pattern the fast forms would have to learn next) */
struct ParkTally {
    std::map<std::string, unsigned long> n;
    std::mutex m;
    ~ParkTally() { if (getenv("MTG_EMU_PARK_STATS")) for (auto& kv : n) fprintf(stderr, "[emu] parked %8lu  %s\n", kv.second, kv.first.c_str()); }
};
inline ParkTally& park_tally_state() { static ParkTally t; return t; }
inline void emu_park_note(Worker& W, const Kmer& cur, uint64_t prev_c, const Adj& a)
{
    if (!getenv("MTG_EMU_PARK_STATS")) return;
    char buf[256];
    const int o = popc4(a.out), in = popc4(a.in);
    std::string key;
    snprintf(buf, sizeof buf, "out %d in %d", o, in);
    key = buf;
    if (o == 2 && in == 1 && W.ix.us.nwords) {
        const uint32_t nt0[2] = {(uint32_t)ctz4(a.out), (uint32_t)ctz4(a.out & (a.out - 1))};
        Kmer x[2] = {kmer_next(cur, nt0[0], W.k, W.mk), kmer_next(cur, nt0[1], W.k, W.mk)};
        Fork2 F;
        F.ab_cur = 0;
        const uint32_t st = W.status, ln = W.lines;
        fork2_read(W, cur, x, F);
        W.status = st; W.lines = ln;
        for (int br = 0; br < 2; br++) {
            snprintf(buf, sizeof buf, " | br%d %s ahead %s first(out %d in %d)", br, F.run[br] ? "run" : "norun", !F.run[br] ? "-" : F.ra[br].ahead + 1u <= (uint32_t)SNP_MAX_L ? (F.ra[br].ahead < (uint32_t)W.k ? "<k" : "<=62") : ">62",
                     popc4(F.r[br].out), popc4(F.r[br].in));
            key += buf;
            if (F.has_last[br]) { snprintf(buf, sizeof buf, " last(out %d in %d)", popc4(F.rl[br].out), popc4(F.rl[br].in)); key += buf; }
        }
        if (F.run[0] && F.run[1]) {
            const long d = (long)F.ra[0].ahead - (long)F.ra[1].ahead;
            snprintf(buf, sizeof buf, " | dlen %ld", d < 0 ? -d : d);
            key += buf;
            if (F.has_last[0] && F.has_last[1] && popc4(F.rl[0].out) == 1 && popc4(F.rl[1].out) == 1) {
                const Kmer e0 = kmer_next(F.last[0], (uint32_t)ctz4(F.rl[0].out), W.k, W.mk), e1 = kmer_next(F.last[1], (uint32_t)ctz4(F.rl[1].out), W.k, W.mk);
                key += e0.f == e1.f ? " same-e" : " diff-e";
                if (e0.f == e1.f) key += W.is_marked(canon(e0)) ? " e-marked" : " e-unmarked";
            }
        }
        key += W.is_marked(canon(cur)) ? " | node marked" : " | node unmarked";
    }
    ParkTally& t = park_tally_state();
    std::lock_guard<std::mutex> lk(t.m);
    t.n[key]++;
}
struct CoopTally {
    unsigned long ok = 0, fail = 0, big[10] = {0};
    ~CoopTally()
    {
        if (getenv("MTG_EMU_COOP_STATS"))
            fprintf(stderr, "[emu] group form of explore_branching: %lu consensus, %lu rejected; too big: sets %lu, in-branching check %lu, depth %lu, consensuses %lu, frames %lu, path set %lu, alignment needed %lu, end rule %lu\n",
                    ok, fail, big[1], big[2], big[3], big[4], big[5], big[6], big[7], big[8]);
    }
};
inline CoopTally& coop_tally_state() { static CoopTally t; return t; }
inline void coop_tally(int n)
{
    CoopTally& t = coop_tally_state();
    if (n > 0) t.ok++; else if (n == COOP_FAIL) t.fail++; else t.big[-n < 10 ? -n : 9]++;
}
#endif
/* The walk's loop-carried state -- contig writer, queue indices, the run it is in, where start node and target sit, counters: some forty words a
 * lane -- is of no use to the forms that answer a branching node, and the forms need a hundred registers of their own (fork2_read).  Kept in
 * registers across them, the two together exceed the kernel's 256 and the compiler spills to private memory: 615 dwords in round 6's first build,
 * 1 350 spill stores per wave and launch, two thirds of the kernel's write requests (PMC: 168 per gap), each a trip to the L2 or further.  The
 * walking lane therefore puts its own state into the wave's LDS right before the forms and takes it back right after (volatile: the compiler may
 * not keep the values in registers "as well"): their live ranges end at the forms' door.  Forty ds_write + forty ds_read per branching node. */
enum { WALK_PARK_WORDS = 40 };
template <int MODE, int G>
MTG_DEV void stage_a_walk(const Index& ix, const FillCfg& cfg, const GapScratch& S, uint64_t src_f, const SwfPattern& R, GapOut& out, BubbleLdsBig* L, bool resume = (MODE == WALK_FINISH),
                          uint32_t* met_branching = nullptr /* set to 1 when the walk stood on a branching node with successors at least once */,
                          uint32_t* walk_park = nullptr /* WALK_PARK_WORDS x 64 words of LDS of the wave (word i of lane l at [i * 64 + l]): where the walk's own state waits while the fork forms run */)
{
    Worker W(ix, cfg, S);
    W.no_dp = MODE == WALK_PARK;
#if defined(MTG_STAMPS) && !defined(MTG_EMU)
    const unsigned long long t_life0 = __builtin_amdgcn_s_memtime();
#endif
    const int k = W.k;
    const uint64_t mk = W.mk, mk1 = W.mk1;
    const uint32_t MAXLEN = 10u * 1000 * 1000;
    const Table adj = ix.adj; /* local copy: the hot loop must not reload the table shape through the Worker */
    const bool r_is_kmer = (R.rlen == (uint32_t)k);
    const SP<uint64_t> q_f = s_qf(cfg, S);
    const SP<uint64_t> q_c = s_qc(cfg, S);
    const SP<int32_t> q_d = s_qd(cfg, S);

    /* contig writer and hot counters live in registers: the Worker's address escapes to the noinline bubble code, so its fields are
     * memory-resident and must stay out of the per-nucleotide path */
    uint64_t* const words = s_words(cfg, S);
    uint64_t acc = 0;
    uint32_t nacc = 0, wpos = 0, lines = 0;
    bool ovf = false;
    auto push_nt = [&](uint32_t nt) {
        acc |= (uint64_t)nt << (2 * nacc);
        if (++nacc == 32) {
            if (wpos >= cfg.cap_words) ovf = true; else words[wpos] = acc;
            wpos++; acc = 0; nacc = 0;
        }
    };
    auto flush = [&]() {
        if (nacc == 0) return;
        if (wpos >= cfg.cap_words) ovf = true; else words[wpos] = acc;
        wpos++; acc = 0; nacc = 0;
    };

    int head = 0, tail = 0;
    if (!resume) {
        q_f[0] = src_f;
        q_c[0] = canon(make_kmer(src_f, k));
        q_d[0] = 0;
        tail = 1;
    }
    uint32_t nb = 0, total_nt = 0;

    /* One flat loop per lane instead of nested BFS / traverse loops: every iteration a lane (W) walks its simple path up to the next
     * event, (B) resolves a branching node, (E) closes a contig and pops the next start node.  Lanes of a wave therefore reach the
     * expensive bubble code of phase B together and run it concurrently, instead of stalling each other one bubble at a time. */
    bool in_contig = false;
    Kmer cur;
    cur.f = cur.r = 0;
    uint64_t start_c = 0, prev_c = 0;
    uint32_t len = 0, c_first = 0;
    int node_depth = 0;
    bool found_R = false;
    Adj a;
    a.out = a.in = a.la = 0;
    bool a_is_cur = false; /* a is the right neighbourhood of cur (false from the moment cur moves on until the next read) */
    /* bulk steps of phase W: low halves of the k-mers to watch for, and the deferred "previous node" */
    const bool bulk_ok = k >= 16; /* a step never covers more than MTG_LA_MAX + 1 <= k nucleotides, and 32 bits are a suffix of the k-mer */
    const uint32_t r0_lo = (uint32_t)R.r0;
    uint32_t start_lo = 0, start_rc_lo = 0;
    bool watch_r = false, lazy_prev = false;
    Kmer pv;
    pv.f = pv.r = 0;
    uint32_t pv_seq = 0, pv_cnt = 0;
    /* A run: the rest of a unitig, announced by the pointer in the entry of one of its junctions.  While it lasts the walk gets its
     * neighbourhoods -- single out-edge, in-degree 1, lookahead of up to MTG_LA_MAX nucleotides -- from sequential reads of the unitig
     * store instead of one dependent random read per MTG_LA_MAX + 1 nodes; everything that consumes them is the lookahead code. */
    const UStore us = ix.us;
    uint64_t run_pos = 0;              /* store position of the next nucleotide of the run */
    uint64_t run_base = 0;             /* store position of the first nucleotide of the run's unitig */
    uint32_t run_left = 0, run_take = 0; /* nucleotides left / handed out with the last neighbourhood */
    uint32_t run_c0 = 0;               /* where, in the contig arena (nucleotide index), the stretch of the store the run belongs to begins: the first nucleotide of the node it was entered by */
    bool run_bwd = false;
    uint32_t store_reads = 0, run_nt = 0;
    /* long runs are not copied by the lane but left as commands for copy_gap -- unless the pattern is searched in the contigs right here
     * (contig mode: contig_contains reads them) or the launch has no copy pass */
    const SP<CopyCmd> cmds = s_cmd(cfg, S);
    const bool defer = r_is_kmer && cfg.cmd_cap != 0;
    uint32_t ncmd = 0, copy_words = 0;
    /* index, in its unitig, of the k-mer the walk stands on while a run lasts (the run's next nucleotide extends that k-mer) */
    auto run_idx = [&]() -> uint32_t { return run_bwd ? (uint32_t)(run_pos - run_base) + 1u : (uint32_t)(run_pos - run_base) - (uint32_t)k; };
    /* the next nucleotides of the run as a neighbourhood with lookahead */
    auto run_chunk = [&](const Kmer& node) -> Adj {
        Adj r;
        run_take = run_left < (uint32_t)MTG_LA_MAX + 1 ? run_left : (uint32_t)MTG_LA_MAX + 1;
        const uint32_t seq = us_peek(us.words, run_pos, run_take, run_bwd);
        run_nt += run_take;
        r.out = 1u << (seq & 3u);
        r.in = 1u << ((uint32_t)(node.f >> (2 * (k - 1))) & 3u);
        r.la = (run_take - 1) | ((seq >> 2) << 4);
        r.up = 0;
        return r;
    };
    /* neighbourhood of the node the walk stands on */
    auto next_adj = [&](const Kmer& node) -> Adj {
        if (run_left) { /* the last neighbourhood came from the run and all its nucleotides have been taken */
            run_left -= run_take;
            run_pos = run_bwd ? run_pos - run_take : run_pos + run_take;
        }
        if (run_left == 0) {
#if defined(MTG_STAMPS) && !defined(MTG_EMU)
            unsigned long long* stamp_acc = W.stamp_acc;
#endif
            MTG_T0(t_rd);
            const Adj r = adj_right_t(adj, node, mk1, lines);
            if (!(r.up && popc4(r.out) == 1 && popc4(r.in) == 1)) { run_take = 0; MTG_T1(t_rd, 9); return r; }
            /* the unitig's length (header word; a walk against the stored orientation does not need it) and the first nucleotides of the
             * run are read together: sixteen nucleotides whatever the run holds, cut to its length afterwards */
            const uint64_t hdr = up_hdr(r.up);
            const uint32_t off = up_off(r.up);
            run_bwd = up_bwd(r.up);
            run_base = (hdr + 1) * 32;
            run_pos = run_bwd ? run_base + off - 1u : run_base + off + (uint32_t)k - 1u;
            /* the node's k nucleotides are the last ones written, and they precede the run in the store -- if the node is the k-mer the junction
             * follows there (a source k-mer that is not in the graph may share its last k - 1 nucleotides with one that is) */
            run_c0 = 32u * wpos + nacc - ((r.in == (1u << ((uint32_t)(node.f >> (2 * (k - 1))) & 3u))) ? (uint32_t)k : 0u);
            const uint32_t seq16 = us_peek(us.words, run_pos, 16u, run_bwd);
            run_left = run_bwd ? off : (uint32_t)us.words[hdr] - (off + (uint32_t)k - 1u);
            store_reads++;
            MTG_T1(t_rd, 9);
            Adj c;
            run_take = run_left < (uint32_t)MTG_LA_MAX + 1 ? run_left : (uint32_t)MTG_LA_MAX + 1;
            const uint32_t seq = seq16 & (run_take >= 16u ? 0xFFFFFFFFu : ((1u << (2u * run_take)) - 1u));
            run_nt += run_take;
            c.out = 1u << (seq & 3u);
            c.in = 1u << ((uint32_t)(node.f >> (2 * (k - 1))) & 3u);
            c.la = (run_take - 1) | ((seq >> 2) << 4);
            c.up = 0;
            return c;
        }
        return run_chunk(node);
    };
    /* Where, in the unitig store, the two k-mers sit that a walk along a simple path has to notice: the contig's start node (looping
     * contig) and, below the first BFS level, the first k-mer of the target.  A k-mer followed by a junction inside a unitig is found
     * through that junction's pointer (base = ~0: it is not).  The canonical k-mers of the stored unitigs are pairwise distinct, so
     * a long run can only meet such a k-mer in its own unitig, at a position known beforehand; a k-mer that is the last of its unitig
     * in its walking direction, or in no unitig, is never INSIDE a run (it can only be a run's last node, which the long step leaves to
     * the step-by-step code). */
    uint64_t start_base = ~0ull, r_base = ~0ull;
    uint32_t start_idx = 0, r_idx = 0;
    bool r_fwd = false, r_known = false;
    auto locate = [&](const Adj& e, uint64_t& base, uint32_t& idx, bool& fwd) {
        base = ~0ull;
        if (!(e.up && popc4(e.out) == 1 && popc4(e.in) == 1)) return;
        base = (up_hdr(e.up) + 1) * 32;
        fwd = !up_bwd(e.up);
        idx = fwd ? up_off(e.up) - 1u : up_off(e.up);
    };
    /* A long step: nb >= 32 nucleotides of the run straight into the contig (no node among them is the start node or, where it
     * matters, the target's first k-mer; at least one nucleotide of the run stays behind for the step-by-step code). */
    auto run_long_step = [&](uint32_t nbulk) {
        const uint64_t* sw = us.words;
        auto append = [&](uint64_t piece, uint32_t n) { /* n <= 32 nucleotides, zero above them */
            acc |= piece << (2 * nacc);
            const uint32_t tot = nacc + n;
            if (tot >= 32) {
                if (wpos >= cfg.cap_words) ovf = true; else words[wpos] = acc;
                wpos++;
                acc = nacc ? piece >> (2 * (32 - nacc)) : 0ull;
                nacc = tot - 32;
            } else nacc = tot;
        };
#ifdef MTG_XCHECK /* TEST-ONLY cross-check of the reasoning above: every node of the long step is looked at */
        {
            Kmer x = cur;
            uint64_t p = run_pos;
            for (uint32_t t = 0; t < nbulk; t++) {
                x = kmer_next(x, us_peek(sw, p, 1, run_bwd), k, mk);
                p = run_bwd ? p - 1 : p + 1;
                if (canon(x) == start_c || (watch_r && x.f == R.r0)) W.status = 0xBAD1;
            }
        }
#endif
        run_nt -= run_take; /* the neighbourhood handed out last is taken back: its nucleotides are part of this step */
        uint64_t p = run_pos;
        uint32_t left = nbulk;
        if (defer && ncmd < cfg.cmd_cap) {
            /* Deferred: the lane completes the word it is filling, leaves the whole words in between to copy_gap (one command), and takes
             * the nucleotides past the last whole word into its accumulator: three short reads, all independent of each other, however
             * long the run. */
            if (nacc) { /* nbulk >= 32: the head always completes the word */
                const uint32_t n0 = 32u - nacc;
                append(us_peek64(sw, p, n0, run_bwd), n0);
                p = run_bwd ? p - n0 : p + n0; left -= n0;
            }
            const uint32_t nw = left >> 5;
            if (nw) {
                if (wpos + nw > cfg.cap_words) ovf = true;
                else {
                    CopyCmd c;
                    c.src = (p << 1) | (run_bwd ? 1ull : 0ull);
                    c.dst = wpos;
                    c.nwords = nw;
                    c.lead = 32u * wpos - run_c0;
                    c.pad_ = 0;
                    cmds[ncmd++] = c;
                    copy_words += nw;
                }
                wpos += nw;
                p = run_bwd ? p - 32ull * nw : p + 32ull * nw; left -= 32u * nw;
            }
            if (left) { append(us_peek64(sw, p, left, run_bwd), left); p = run_bwd ? p - left : p + left; }
            store_reads += 6;
        } else if (!run_bwd) {
            store_reads += (nbulk >> 5) + 2;
            const uint32_t inw = (uint32_t)(p & 31u);
            if (inw) { /* up to the end of the source word */
                const uint32_t n0 = 32u - inw < left ? 32u - inw : left;
                append(us_peek64(sw, p, n0, false), n0);
                p += n0; left -= n0;
            }
            const uint64_t* q = sw + (p >> 5);
            const uint32_t nw = left >> 5;
            for (uint32_t i = 0; i < nw; i += 8) { /* eight source words in flight */
                uint64_t w[8];
MTG_UNROLL
                for (uint32_t u = 0; u < 8; u++) w[u] = (i + u < nw) ? q[i + u] : 0ull;
MTG_UNROLL
                for (uint32_t u = 0; u < 8; u++) if (i + u < nw) append(w[u], 32u);
            }
            p += 32ull * nw; left -= 32u * nw;
            if (left) { append(us_peek64(sw, p, left, false), left); p += left; }
        } else {
            store_reads += (nbulk >> 5) + 2;
            const uint32_t inw = (uint32_t)(p & 31u) + 1u; /* nucleotides of the source word at or below p */
            if (inw < 32u) {
                const uint32_t n0 = inw < left ? inw : left;
                append(us_peek64(sw, p, n0, true), n0);
                p -= n0; left -= n0;
            }
            const uint32_t nw = left >> 5;
            if (nw) {
                const uint64_t* q = sw + (p >> 5); /* p is the last nucleotide of this word */
                for (uint32_t i = 0; i < nw; i += 8) {
                    uint64_t w[8];
MTG_UNROLL
                    for (uint32_t u = 0; u < 8; u++) w[u] = (i + u < nw) ? *(q - (i + u)) : 0ull;
MTG_UNROLL
                    for (uint32_t u = 0; u < 8; u++) if (i + u < nw) append(rev_fields64(w[u]) ^ 0xAAAAAAAAAAAAAAAAULL, 32u);
                }
                p -= 32ull * nw; left -= 32u * nw;
            }
            if (left) { append(us_peek64(sw, p, left, true), left); p -= left; }
        }
        /* the node the walk now stands on: the last k nucleotides taken */
        const uint64_t tail = us_peek64(sw, run_bwd ? p + (uint32_t)k : p - (uint32_t)k, (uint32_t)k, run_bwd);
        cur.r = tail ^ (0xAAAAAAAAAAAAAAAAULL & mk); /* little-endian image = reversed order: complementing it gives the reverse complement */
        cur.f = revcomp(cur.r, k);
        run_pos = p;
        run_left -= nbulk;
        run_nt += nbulk;
        len += nbulk;
    };
    /* ---- a parked walk goes on where it stopped: at the branching node, the SNP attempt behind it */
    bool resuming = false, parked = false, answered = false;
    int saved_n = 0, saved_chosen = -1;
    bool answered_snp = false; /* the bubble kernel found the strict SNP pattern at the node the walk stands on: its answer is consumed like the walk's own */
    SnpSeq saved_snp;
    saved_snp.lo = saved_snp.hi = 0;
    /* S.snp_fast == 2 (the walk kernel, when the launch serves bubbles in rounds): the walk parks at EVERY branching node, also where the SNP
     * fast path would answer -- the lanes of a wave meet their SNPs at different steps, and a wave in which one lane at a time runs the bubble
     * code while 63 wait spends four times as long there as on the walking; the bubble kernel runs it for all parked gaps at once */
    const bool park_all = MODE == WALK_SIMPLE || (MODE == WALK_PARK && S.snp_fast == 2);
    if (resume) {
        const WalkSave sv = *s_save(cfg, S);
        resuming = true;
        answered = sv.answered == 1;
        answered_snp = sv.answered == 2;
        saved_n = sv.bn; saved_chosen = sv.bchosen;
        saved_snp.lo = sv.snp_lo; saved_snp.hi = sv.snp_hi;
        cur = make_kmer(sv.cur_f, k);
        prev_c = sv.prev_c;
        const Kmer st = make_kmer(sv.start_f, k);
        start_c = canon(st); start_lo = (uint32_t)st.f; start_rc_lo = (uint32_t)st.r;
        acc = sv.acc; nacc = sv.nacc; wpos = sv.wpos;
        start_base = sv.start_base; start_idx = sv.start_idx;
        r_base = sv.r_base; r_idx = sv.r_idx;
        found_R = (sv.flags & 1u) != 0; r_fwd = (sv.flags & 2u) != 0; r_known = (sv.flags & 4u) != 0;
        if (sv.flags & 0x80000000u) W.status = GAP_OVF_SEEN; /* the bubble kernel ran out of a work area: the gap is run again in a larger scratch tier */
        if (sv.flags & 0x40000000u) W.status = 0xBADC;       /* TEST-ONLY emulation: the group form and the one-lane form disagreed */
        len = sv.len; c_first = sv.c_first; node_depth = sv.node_depth;
        head = (int)sv.head; tail = (int)sv.tail; nb = sv.nb; total_nt = sv.total_nt;
        ncmd = sv.ncmd; copy_words = sv.copy_words; store_reads = sv.store_reads; run_nt = sv.run_nt; lines = sv.lines;
        W.n_marked = sv.n_marked; W.msig0 = sv.msig[0]; W.msig1 = sv.msig[1]; W.msig2 = sv.msig[2]; W.msig3 = sv.msig[3];
        watch_r = r_is_kmer && node_depth > k;
        in_contig = true;
        uint32_t l_ = 0; /* the neighbourhood the walk stood on (read once already: not counted again) */
        a = adj_right_t(adj, cur, mk1, l_);
        a_is_cur = true;
    }
    /* the walk's own state to the wave's LDS (out_) and back (see WALK_PARK_WORDS) */
    auto walk_state = [&](bool out_) {
#ifdef MTG_EMU
        volatile uint32_t* const pk = walk_park + (S.lane & 63u);
#else   /* an LDS pointer by type: through the generic one the accesses were flat_load / flat_store (round 6: twice the kernel's vector memory reads) */
        volatile __attribute__((address_space(3))) uint32_t* const pk = (volatile __attribute__((address_space(3))) uint32_t*)walk_park + (threadIdx.x & 63u); /* the lane of the WAVE (S.lane is the slot's: the resumed gaps of a work list share values) */
#endif
        uint32_t i_ = 0;
#define MTG_PK32(v) do { if (out_) pk[i_ * 64u] = (uint32_t)(v); else (v) = (decltype(v))pk[i_ * 64u]; i_++; } while (0)
#define MTG_PK64(v) do { if (out_) { pk[i_ * 64u] = (uint32_t)(v); pk[(i_ + 1u) * 64u] = (uint32_t)((uint64_t)(v) >> 32); } \
                         else { const uint64_t lo_ = pk[i_ * 64u], hi_ = pk[(i_ + 1u) * 64u]; (v) = lo_ | (hi_ << 32); } i_ += 2u; } while (0)
        uint32_t fl = 0;
        if (out_) fl = (ovf ? 1u : 0u) | (in_contig ? 2u : 0u) | (found_R ? 4u : 0u) | (a_is_cur ? 8u : 0u) | (watch_r ? 16u : 0u) | (run_bwd ? 32u : 0u) | (r_fwd ? 64u : 0u) |
                       (r_known ? 128u : 0u) | (answered ? 256u : 0u);
        MTG_PK32(fl);
        if (!out_) { ovf = fl & 1u; in_contig = fl & 2u; found_R = fl & 4u; a_is_cur = fl & 8u; watch_r = fl & 16u; run_bwd = fl & 32u; r_fwd = fl & 64u; r_known = fl & 128u; answered = fl & 256u; }
        MTG_PK64(acc); MTG_PK64(start_c); MTG_PK64(run_pos); MTG_PK64(run_base); MTG_PK64(start_base); MTG_PK64(r_base);
        MTG_PK32(nacc); MTG_PK32(wpos); MTG_PK32(lines); MTG_PK32(head); MTG_PK32(tail); MTG_PK32(nb); MTG_PK32(total_nt); MTG_PK32(len); MTG_PK32(c_first);
        MTG_PK32(node_depth); MTG_PK32(start_lo); MTG_PK32(start_rc_lo); MTG_PK32(run_left); MTG_PK32(run_take); MTG_PK32(run_c0); MTG_PK32(store_reads);
        MTG_PK32(run_nt); MTG_PK32(ncmd); MTG_PK32(copy_words); MTG_PK32(start_idx); MTG_PK32(r_idx); MTG_PK32(saved_n); MTG_PK32(saved_chosen);
#undef MTG_PK32
#undef MTG_PK64
        if (out_) { pv.f = pv.r = 0; pv_seq = pv_cnt = 0; } /* (only read under lazy_prev, which is false here: dead, and now the compiler knows) */
    };
    MTG_GUARD_DECL(g_flat);
#if defined(MTG_STAMPS) && !defined(MTG_EMU)
    unsigned long long t_last_iter = t_life0;
#endif
    for (;;) {
        MTG_GUARD(g_flat, 200000u, 1, { W.status = 0xD1E; break; });
#if defined(MTG_STAMPS) && !defined(MTG_EMU)
        W.phase_acc[4] += 1ull;
        t_last_iter = __builtin_amdgcn_s_memtime();
#endif
        bool end_contig = false;
        if (!resuming) {
        if (!in_contig) {
            if (!(head < tail) || W.status != GAP_OK) break;
#if defined(MTG_STAMPS) && !defined(MTG_EMU)
            const unsigned long long t_cs0 = __builtin_amdgcn_s_memtime();
#endif
            const uint64_t node_f = q_f[head];
            node_depth = q_d[head];
            head++;
            /* ---- traverse(node) ---- */
            c_first = wpos;
            for (int i = k - 1; i >= 0; i--) push_nt((uint32_t)(node_f >> (2 * i)) & 3u);
            cur = make_kmer(node_f, k);
            start_c = canon(cur);
            prev_c = 0; /* gatb: default-constructed previousNode has k-mer value 0 */
            len = 0;
            found_R = (r_is_kmer && cur.f == R.r0);
            start_lo = (uint32_t)cur.f;
            start_rc_lo = (uint32_t)cur.r;
            watch_r = r_is_kmer && node_depth > k; /* found_R only matters there (see phase E) */
            run_left = 0;
            a = next_adj(cur);
            a_is_cur = true;
            start_base = ~0ull;
            if (run_left) { start_base = run_base; start_idx = run_idx(); }
            if (watch_r && !r_known && us.nwords) { /* first contig below the first BFS level: where the target's first k-mer sits */
                r_known = true;
                const Adj e = adj_right_t(adj, make_kmer(R.r0, k), mk1, lines);
                locate(e, r_base, r_idx, r_fwd);
            }
            in_contig = true;
#if defined(MTG_STAMPS) && !defined(MTG_EMU)
            W.phase_acc[0] += __builtin_amdgcn_s_memtime() - t_cs0; W.phase_acc[1] += 1ull;
#endif
        }
#if defined(MTG_STAMPS) && !defined(MTG_EMU)
        unsigned long long* stamp_acc = W.stamp_acc;
#endif
        MTG_T0(t_w);
        /* ---- phase W: simple path.  The entry just read also lists up to MTG_LA_MAX further nucleotides along which every node has
         * exactly one in- and one out-edge (lookahead): those nodes are non-branching (nothing to mark) and need no read. */
        MTG_GUARD_DECL(g_w);
        while (popc4(a.out) == 1 && popc4(a.in) <= 1) {
            MTG_GUARD(g_w, 2000000u, 2, { W.status = 0xD1E; end_contig = true; break; });
            if (run_left >= 64u) {
                /* a long run ahead: take it in one step, up to the nucleotide before the first node that needs a look */
                uint32_t nbulk = run_left - 1u;
                const uint32_t here = run_idx();
                if (start_base == run_base) {
                    const uint32_t t = run_bwd ? here - start_idx : start_idx - here; /* nucleotides until the walk stands on the start node */
                    if (t - 1u < nbulk) nbulk = t - 1u; /* t == 0 (or behind the walk: wraps to a large value) leaves nbulk alone */
#ifdef MTG_TRACE_BULK
                    if (nbulk == t - 1u) fprintf(stderr, "BULK start-clamp t=%u\n", t);
#endif
                }
                if (watch_r && r_base == run_base && r_fwd != run_bwd) {
                    const uint32_t t = run_bwd ? here - r_idx : r_idx - here;
                    if (t - 1u < nbulk) nbulk = t - 1u;
#ifdef MTG_TRACE_BULK
                    if (nbulk == t - 1u) fprintf(stderr, "BULK r-clamp t=%u\n", t);
#endif
                }
                if (len + nbulk + 32u > MAXLEN) nbulk = MAXLEN > len + 32u ? MAXLEN - len - 32u : 0u;
                if (nbulk >= 32u) {
                    MTG_T0(t_ls);
                    a_is_cur = false;
                    run_long_step(nbulk);
                    MTG_T1(t_ls, 8);
                    lazy_prev = false;
                    if (ovf || W.status) { end_contig = true; break; }
                    a = run_chunk(cur);
                    a_is_cur = true;
                    continue;
                }
            }
            uint32_t nt = (uint32_t)ctz4(a.out);
            uint32_t indeg = (uint32_t)popc4(a.in); /* in-degree of the node we step onto */
            uint32_t la = (indeg == 1) ? a.la : 0u;
            uint32_t known = la & 15u;              /* nodes ahead known to be simple */
            la >>= 4;
            /* The step covers j = known + 1 nucleotides.  Per nucleotide the reference only (a) compares the node with the start node
             * (looping contig) and (b), below the first BFS level, with the first k-mer of R.  Both are evaluated for all j nodes on the
             * low 32 bits of the forward k-mers; unless one of them may hit, or a limit is near, the step is then taken in one go. */
            a_is_cur = false; /* the walk moves on */
            const uint32_t j = known + 1;
            const uint32_t seq = nt | (la << 2);
            bool bulk = bulk_ok && len + j <= MAXLEN;
            if (bulk) {
                uint32_t x = (uint32_t)cur.f, s = seq, hit = 0;
                MTG_UNROLL
                for (uint32_t i = 1; i <= MTG_LA_MAX + 1; i++) {
                    x = (x << 2) | (s & 3u);
                    s >>= 2;
                    const bool at_start = (x == start_lo) | (x == start_rc_lo);
                    const bool at_r = (x == r0_lo);
                    hit |= ((at_start & (i < j)) | (at_r & (i <= j) & watch_r)) ? 1u : 0u;
                }
                bulk = hit == 0;
            }
            if (bulk) {
                lazy_prev = true;
                pv = cur; pv_seq = seq; pv_cnt = known;
                cur = kmer_advance(cur, seq, j, k, mk);
                acc |= (uint64_t)seq << (2 * nacc);
                nacc += j;
                if (nacc >= 32) {
                    if (wpos >= cfg.cap_words) ovf = true; else words[wpos] = acc;
                    wpos++;
                    nacc -= 32;
                    acc = (uint64_t)(seq >> (2 * (j - nacc))); /* the nacc nucleotides that did not fit */
                }
                len += j;
                if (known) indeg = 1;
            } else {
                lazy_prev = false;
                for (;;) {
                    prev_c = canon(cur);
                    cur = kmer_next(cur, nt, k, mk);
                    push_nt(nt);
                    len++;
                    if (r_is_kmer && cur.f == R.r0) found_R = true;
                    if (known == 0) break;              /* this node's neighbourhood has to be read */
                    /* known simple node: terminator.mark() is a no-op on it */
                    if (canon(cur) == start_c || len > MAXLEN || ovf) { end_contig = true; break; }
                    nt = la & 3u;
                    la >>= 2;
                    known--;
                    indeg = 1;
                }
                if (end_contig) break;
            }
            const Adj a2 = next_adj(cur);
            if (!(popc4(a2.out) == 1 && indeg == 1)) W.mark_canon(canon(cur)); /* terminator.mark(cur) */
            a = a2;
            a_is_cur = true;
            if (canon(cur) == start_c || len > MAXLEN || ovf || W.status) { end_contig = true; break; } /* looping / limits */
        }
        if (lazy_prev) { prev_c = canon(kmer_advance(pv, pv_seq & ((1u << (2 * pv_cnt)) - 1u), pv_cnt, k, mk)); lazy_prev = false; }
        MTG_T1(t_w, 0);
        /* ---- phase B: branching node ---- */
        if (!end_contig && a.out == 0) {
            /* dead end.  The reference still calls explore_branching here: its frontline has no successor to move to and gives up at once,
             * nothing is marked and its visited set is dropped -- the contig ends, which is all that is left of the call. */
            end_contig = true;
#ifdef MTG_XCHECK /* TEST-ONLY: run it anyway and see that nothing comes of it */
            { int ch_ = -1; const uint32_t nm_ = W.n_marked; if (explore_branching(W, cur, prev_c, ch_) != 0 || W.n_marked != nm_ || W.n_seen != 0) W.status = 0xBAD9; }
#endif
        }
        } /* !resuming */
#if defined(MTG_STAMPS) && !defined(MTG_EMU)
        unsigned long long* stamp_acc = W.stamp_acc;
#endif
        MTG_T0(t_b);
        if (!end_contig) {
            if (met_branching) *met_branching = 1u;
            int chosen = -1;
            MTG_T0(t_snp);
            SnpSeq fast_seq;
            fast_seq.lo = fast_seq.hi = 0;
            int n = 0;
            if (resuming && answered_snp) { n = saved_n; chosen = saved_chosen; fast_seq = saved_snp; answered_snp = false; }
            else if (MODE != WALK_SIMPLE && !resuming && !park_all) {
                if (walk_park) walk_state(true);
                n = snp_bubble_fast(W, cur, prev_c, a, chosen, fast_seq);
                if (walk_park) walk_state(false);
            }
            resuming = false;
            MTG_T1(t_snp, 6);
            const bool fast = n > 0; /* its nodes are simple and the last one is already marked: nothing to ask the index on the way */
            const bool refused = n < 0 || W.status != GAP_OK; /* the fast forms know the reference's answer is "no consensus" (indel_bulk): the contig ends here, nothing is parked.
                                                               * (A status can only come from the TEST-ONLY cross-checks inside the fast forms: it must end the gap here -- a parked walk
                                                               * is resumed with a fresh status and would bury it.) */
            bool coop = false;        /* the consensus sits in the group's LDS area */
            if (refused) n = 0;
            else if (!fast && answered) {
                /* a bubble kernel has answered this branching node while the gap was parked (the consensus where the one-lane code leaves it) */
                n = saved_n; chosen = saved_chosen;
                answered = false;
            } else if (!fast) {
                if (MODE == WALK_PARK || MODE == WALK_SIMPLE) {
                    /* not the strict SNP pattern: the gap is parked here and a group of lanes takes it over (k_bubble / k_finish) */
                    WalkSave sv;
                    sv.answered = 0; sv.bn = 0; sv.bchosen = -1;
                    sv.cur_f = cur.f; sv.prev_c = prev_c;
                    sv.start_f = ((uint32_t)start_c == start_lo) ? start_c : revcomp(start_c, k); /* the oriented start node, from its canonical form and the low half of its forward form (only the canonical form and the two low halves are used) */
                    sv.acc = acc; sv.nacc = nacc; sv.wpos = wpos;
                    sv.start_base = start_base; sv.start_idx = start_idx; sv.r_base = r_base; sv.r_idx = r_idx;
                    sv.flags = (found_R ? 1u : 0u) | (r_fwd ? 2u : 0u) | (r_known ? 4u : 0u);
                    sv.len = len; sv.c_first = c_first; sv.node_depth = node_depth;
                    sv.head = (uint32_t)head; sv.tail = (uint32_t)tail; sv.nb = nb; sv.total_nt = total_nt;
                    sv.ncmd = ncmd; sv.copy_words = copy_words; sv.store_reads = store_reads; sv.run_nt = run_nt; sv.lines = W.lines + lines;
                    sv.n_marked = W.n_marked; sv.msig[0] = W.msig0; sv.msig[1] = W.msig1; sv.msig[2] = W.msig2; sv.msig[3] = W.msig3;
                    sv.pad_ = 0;
                    *s_save(cfg, S) = sv;
                    parked = true;
#ifdef MTG_EMU
                    emu_park_note(W, cur, prev_c, a);
#endif
                    break;
                }
                if (MODE == WALK_FINISH) {
#ifdef MTG_COOP_OFF /* diagnostics: the finishing kernel without the group form (every bubble by the one-lane code) */
                    n = COOP_TOOBIG;
#else
#ifdef MTG_EMU_LANES /* TEST-ONLY: the group form by N lanes in lock step (mtg_bubble.h: EmuLanes); each lane has its own Worker, lane 0's comes back */
                    if (L) {
                        n = emu_group_run([&](uint32_t lane) {
                            Worker Wl = W;
                            int ch = -1;
                            const int r = coop_explore<G>(Wl, *L, cur, prev_c, ch);
                            if (lane == 0) { chosen = ch; W.lines = Wl.lines; W.status = Wl.status; }
                            return r > 0 ? (r << 4) | (ch & 15) : r; /* the lanes must agree on the length AND on the consensus chosen */
                        });
                        if (n == -0x7BADD) { W.status = 0xBADD; n = COOP_FAIL; }
                        else if (n > 0) n >>= 4;
                    } else n = COOP_TOOBIG;
#else
                    n = L ? coop_explore<G>(W, *L, cur, prev_c, chosen) : COOP_TOOBIG; /* no LDS areas: the one-lane finishing kernel (a handful of parked gaps) */
#endif
#endif
#ifdef MTG_EMU
                    coop_tally(n);
#endif
#ifdef MTG_XCHECK /* TEST-ONLY: the general code next to every answer of the group form -- same verdict, same consensus, same marks (0xBADC) */
                    if (n >= 0 && W.status == GAP_OK) {
                        uint32_t planned_new = 0;
                        for (uint32_t i = 0; i < (n > 0 ? L->n_marks : 0u); i++) {
                            bool dup = W.is_marked(L->inv[i]);
                            for (uint32_t j = 0; j < i && !dup; j++) dup = L->inv[j] == L->inv[i];
                            planned_new += dup ? 0u : 1u;
                        }
                        const uint32_t nm0 = W.n_marked;
                        int ch2 = -1;
                        const int n2 = explore_branching(W, cur, prev_c, ch2);
                        if (W.status == GAP_OK) {
                            bool same = (n2 > 0) == (n > 0);
                            if (same && n > 0) {
                                same = n2 == n && W.n_marked - nm0 == planned_new;
                                const SP<uint8_t> p2 = s_cons(cfg, S) + (size_t)ch2 * CONS_LEN;
                                for (int i = 0; i < n && same; i++) same = p2[i] == L->b.cons[chosen][i];
                                for (uint32_t i = 0; i < L->n_marks && same; i++) same = W.is_marked(L->inv[i]);
                            }
                            if (same && n <= 0) same = W.n_marked == nm0;
                            if (!same) W.status = 0xBADC;
                        }
                    }
#endif
                    if (n < 0) n = explore_branching(W, cur, prev_c, chosen); /* too big for the LDS areas: the one-lane form from HBM scratch */
                    else { coop = true; if (n > 0) coop_apply_marks<G>(W, *L); }
                } else if (MODE == WALK_CLASSIC) n = explore_branching(W, cur, prev_c, chosen);
            }
            if (MODE == WALK_SIMPLE || n <= 0) { /* (the light kernel never has a consensus to consume: it parked above, or the contig ends on a status) */
                end_contig = true;
            } else {
                const SP<uint8_t> p = s_cons(cfg, S) + (size_t)chosen * CONS_LEN;
                bool looping = false;
                MTG_T0(t_cons);
                /* terminator.mark() along the consensus of the general code: only branching nodes are marked, and a node inside a stored unitig
                 * (not its end node) has one in- and one out-edge -- c_ra = nodes known to lie ahead of the current one in its unitig */
                uint32_t c_ra = 0;
                a_is_cur = false;
                /* The consensus of the SNP fast path in one go: its nodes are simple (nothing to mark); what the loop below tests per node -- the
                 * start node, the target's first k-mer -- is tested on the low 32 bits of all n forward k-mers first, and only a possible hit
                 * sends the walk through the loop. */
                bool appended = false;
                if (fast && bulk_ok) {
                    uint32_t xlo = (uint32_t)cur.f, hit = 0;
                    for (int i = 0; i < n; i++) {
                        xlo = (xlo << 2) | fast_seq.get(i);
                        hit |= ((xlo == start_lo) | (xlo == start_rc_lo) | ((xlo == r0_lo) & r_is_kmer)) ? 1u : 0u;
                    }
                    if (!hit) {
                        Kmer y = cur;
                        for (int done = 0; done < n - 1;) {
                            const int c = n - 1 - done < 15 ? n - 1 - done : 15;
                            y = kmer_advance(y, fast_seq.bits(done, c), (uint32_t)c, k, mk);
                            done += c;
                        }
                        const Kmer z = kmer_next(y, fast_seq.get(n - 1), k, mk);
#ifdef MTG_XCHECK /* TEST-ONLY: node by node */
                        {
                            Kmer q = cur, qp = cur;
                            for (int i = 0; i < n; i++) { qp = q; q = kmer_next(q, fast_seq.get(i), k, mk); if (canon(q) == start_c || (r_is_kmer && q.f == R.r0)) W.status = 0xBADB; }
                            if (q.f != z.f || q.r != z.r || qp.f != y.f) W.status = 0xBADB;
                        }
#endif
                        prev_c = canon(y);
                        cur = z;
                        const uint32_t n_lo = n < 32 ? (uint32_t)n : 32u;
                        for (uint32_t part = 0; part < 2; part++) {
                            const uint32_t cnt = part == 0 ? n_lo : (uint32_t)n - n_lo;
                            if (!cnt) break;
                            const uint64_t piece = part == 0 ? (n_lo < 32u ? fast_seq.lo & ((1ull << (2 * n_lo)) - 1ull) : fast_seq.lo) : (fast_seq.hi & ((1ull << (2 * cnt)) - 1ull));
                            acc |= piece << (2 * nacc);
                            const uint32_t tot = nacc + cnt;
                            if (tot >= 32) {
                                if (wpos >= cfg.cap_words) ovf = true; else words[wpos] = acc;
                                wpos++;
                                acc = nacc ? piece >> (2 * (32 - nacc)) : 0ull;
                                nacc = tot - 32;
                            } else nacc = tot;
                        }
                        len += (uint32_t)n;
                        appended = true;
                    }
                }
                for (int i = 0; i < (appended ? 0 : n); i++) {
                    const uint32_t nti = fast ? fast_seq.get(i) : coop ? (uint32_t)L->b.cons[chosen][i] : (uint32_t)p[i];
                    prev_c = canon(cur);
                    cur = kmer_next(cur, nti, k, mk);
                    push_nt(nti);
                    len++;
                    if (!fast) {
#ifdef MTG_XCHECK
                        const bool br_ref = W.is_branching(cur);
                        bool br_here = false;
#endif
                        if (c_ra >= 2u) c_ra--;
                        else {
                            const bool in1 = c_ra == 1u; /* the end node of the unitig the path was in: its in-degree is known */
                            c_ra = 0;
                            const Adj r = adj_right_t(adj, cur, mk1, lines);
                            const bool branching = popc4(r.out) != 1 || (!in1 && popc4(adj_left(W.ix, cur, mk1, lines).in) != 1);
                            if (branching) W.mark_canon(canon(cur));
                            RunAt ru;
                            if (us.nwords && run_at(us, r, k, ru, lines)) c_ra = ru.ahead;
#ifdef MTG_XCHECK
                            br_here = branching;
#endif
                        }
#ifdef MTG_XCHECK
                        if (br_ref != br_here) W.status = 0xBAD7;
#endif
                    }
                    if (r_is_kmer && cur.f == R.r0) found_R = true;
                    if (canon(cur) == start_c) looping = true;
                }
                MTG_T1(t_cons, 7);
                if (looping || len > MAXLEN || ovf || W.status) end_contig = true;
                else { a = next_adj(cur); a_is_cur = true; }
            }
        }
        MTG_T1(t_b, 1);
        if (!end_contig) continue;
        /* ---- phase E: the contig is complete ---- */
#if defined(MTG_STAMPS) && !defined(MTG_EMU)
        const unsigned long long t_e0 = __builtin_amdgcn_s_memtime();
        struct PhaseE { unsigned long long t0; unsigned long long* acc; __device__ ~PhaseE() { acc[2] += __builtin_amdgcn_s_memtime() - t0; acc[3] += 1ull; } } phase_e_{t_e0, W.phase_acc};
#endif
        in_contig = false;
        flush();
        if (ovf) W.status = GAP_OVF_CONTIG;
        if (W.status) break;
        const uint32_t clen = (uint32_t)k + len;
        s_cstart(cfg, S)[nb] = c_first;
        s_clen(cfg, S)[nb] = clen;
        nb++;
        total_nt += clen;
        /* swf: stop when R occurs in the contig and depth > k */
        if (!r_is_kmer && node_depth > k) found_R = contig_contains(s_words(cfg, S) + c_first, clen, R);
        if (found_R && node_depth > k) break;
        if ((int)nb > cfg.max_nodes) break;
        if (node_depth + (int)clen > cfg.max_depth) continue;
        /* push the successors that were never extended from */
        const Adj ea = a_is_cur ? a : adj_right_t(adj, cur, mk1, lines); /* the walk usually stopped on a node whose neighbourhood it has just read */
#ifdef MTG_XCHECK
        { uint32_t l_ = 0; if (adj_right_t(adj, cur, mk1, l_).out != ea.out) W.status = 0xBADA; }
#endif
        for (uint32_t em = ea.out & 15u; em; em &= em - 1u) {
            const uint32_t nt = low_nt(em);
            const Kmer s = kmer_next(cur, nt, k, mk);
            const uint64_t cs = canon(s);
            bool seen = false;
            for (int i = 1; i < tail; i++) if (q_c[i] == cs) { seen = true; break; }
            if (seen) continue;
            if ((uint32_t)tail >= cfg.qcap) { W.status = GAP_OVF_QUEUE; break; }
            q_f[tail] = s.f;
            q_c[tail] = cs;
            q_d[tail] = node_depth + (int)len + 1;
            tail++;
        }
    }
#if defined(MTG_STAMPS) && !defined(MTG_EMU)
    for (int i = 0; i < 15; i++) atomicAdd(&g_stamps[i], W.stamp_acc[i]);
    atomicAdd(&g_stamps[15], 1ull);
    for (int i = 0; i < 15; i++) if (W.form_acc[i]) atomicAdd(&g_forms[i], W.form_acc[i]);
    atomicAdd(&g_forms[15], 1ull);
    {
        const unsigned long long t_end = __builtin_amdgcn_s_memtime(), life = t_end - t_life0;
        atomicAdd(&g_life[63 - __clzll(life | 1ull)], 1ull);
        atomicMax(&g_life[32], ~t_life0);
        atomicMax(&g_life[33], t_end);
        W.phase_acc[5] = t_end - t_last_iter;
        for (int i = 0; i < 6; i++) atomicAdd(&g_phase[i], W.phase_acc[i]);
    }
#endif
    /* leave the zero-initialised region as it was found, whatever the exit path (a parked walk keeps its marked set: it goes on) */
    W.seen_clear();
    W.iseen_clear();
    if (!parked) W.marked_clear();
    out.n_contigs = nb;
    out.status = parked ? (uint32_t)GAP_PARKED : W.status;
    out.lines = W.lines + lines;
    out.store_reads = store_reads;
    out.run_nt = run_nt;
    out.n_cmds = ncmd;
    out.copy_words = copy_words;
    out.total_nt = total_nt;
    out.n_words = wpos;
}
MTG_DEV void stage_a_gap(const Index& ix, const FillCfg& cfg, const GapScratch& S, uint64_t src_f, const SwfPattern& R, GapOut& out)
{
    stage_a_walk<WALK_CLASSIC, 1>(ix, cfg, S, src_f, R, out, nullptr);
}

/* ---- the branching node of a parked gap, answered on its own (the rounds between the walk kernel's launches): the walk's state says which
 * node, the answer goes back into it (WalkSave::answered) and the walk kernel takes it from there.
 * bubble_coop: by a group of G lanes from LDS; false = too big for the LDS areas (the gap goes to bubble_classic).
 * bubble_classic: by one lane from HBM scratch. */
MTG_DEV void bubble_store(const FillCfg& cfg, const GapScratch& S, Worker& W, WalkSave& sv, int n, int chosen, uint32_t kind = 1)
{
    sv.answered = kind; sv.bn = n; sv.bchosen = chosen;
    sv.n_marked = W.n_marked; sv.msig[0] = W.msig0; sv.msig[1] = W.msig1; sv.msig[2] = W.msig2; sv.msig[3] = W.msig3;
    sv.lines += W.lines;
    *s_save(cfg, S) = sv;
}
MTG_DEV void bubble_load(Worker& W, const WalkSave& sv)
{
    W.n_marked = sv.n_marked; W.msig0 = sv.msig[0]; W.msig1 = sv.msig[1]; W.msig2 = sv.msig[2]; W.msig3 = sv.msig[3];
}
/* the strict SNP pattern at the parked node (the walk kernel parks there too when the launch serves bubbles in rounds): the fast path's answer,
 * exactly as the walking lane would have computed it (the node's neighbourhood is read the way a resumed walk reads it); true = answered */
MTG_DEV bool bubble_snp(const Index& ix, const FillCfg& cfg, const GapScratch& S, Worker& W, WalkSave& sv, const Kmer& cur, bool store)
{
    if (!S.snp_fast) return false;
    const Adj a = adj_right_t(ix.adj, cur, W.mk1, W.lines);
    SnpSeq fs;
    fs.lo = fs.hi = 0;
    int chosen = -1;
    const int n = snp_bubble_fast(W, cur, sv.prev_c, a, chosen, fs);
    if (n <= 0 || W.status) return false;
    sv.snp_lo = fs.lo; sv.snp_hi = fs.hi;
    if (store) bubble_store(cfg, S, W, sv, n, chosen, 2);
    else { sv.bn = n; sv.bchosen = chosen; }
    return true;
}
template <int G> MTG_DEV bool bubble_coop(const Index& ix, const FillCfg& cfg, const GapScratch& S, BubbleLds& L)
{
    typedef Grp<G> GP;
    WalkSave sv = *s_save(cfg, S);
    Worker W(ix, cfg, S);
    bubble_load(W, sv);
    const Kmer cur = make_kmer(sv.cur_f, ix.k);
    {   /* every lane of the group runs the fast path (as the group's lanes run the walk in k_finish); one stores */
        Worker Ws(ix, cfg, S);
        bubble_load(Ws, sv);
        WalkSave svs = sv;
        if (bubble_snp(ix, cfg, S, Ws, svs, cur, false)) { if (GP::gl() == 0) bubble_store(cfg, S, Ws, svs, svs.bn, svs.bchosen, 2); GP::sync(); return true; }
        if (Ws.status) { /* TEST-ONLY emulation: a cross-check of the fast path failed */
            if (GP::gl() == 0) { svs.flags |= 0x40000000u; bubble_store(cfg, S, Ws, svs, 0, -1); }
            GP::sync();
            return true;
        }
    }
    int chosen = -1;
#ifdef MTG_EMU_LANES /* TEST-ONLY: by N lanes in lock step (see stage_a_walk) */
    int n = emu_group_run([&](uint32_t lane) {
        Worker Wl = W;
        int ch = -1;
        const int r = coop_explore<G>(Wl, L, cur, sv.prev_c, ch);
        if (lane == 0) { chosen = ch; W.lines = Wl.lines; W.status = Wl.status; }
        return r > 0 ? (r << 4) | (ch & 15) : r;
    });
    if (n == -0x7BADD) { sv.flags |= 0x40000000u; n = COOP_FAIL; }
    else if (n > 0) n >>= 4;
#else
    const int n = coop_explore<G>(W, L, cur, sv.prev_c, chosen);
#endif
#ifdef MTG_EMU
    coop_tally(n);
#endif
    if (n < 0) return false;
#ifdef MTG_XCHECK /* TEST-ONLY: the general code next to the group form's answer: same verdict, consensus and marks */
    {
        uint32_t planned_new = 0;
        for (uint32_t i = 0; i < (n > 0 ? L.n_marks : 0u); i++) {
            bool dup = W.is_marked(L.inv[i]);
            for (uint32_t j = 0; j < i && !dup; j++) dup = L.inv[j] == L.inv[i];
            planned_new += dup ? 0u : 1u;
        }
        Worker W2(ix, cfg, S);
        bubble_load(W2, sv);
        const uint32_t nm0 = W2.n_marked;
        int ch2 = -1;
        const int n2 = explore_branching(W2, cur, sv.prev_c, ch2);
        bool same = W2.status != GAP_OK || (n2 > 0) == (n > 0);
        if (W2.status == GAP_OK && same && n > 0) {
            same = n2 == n && W2.n_marked - nm0 == planned_new;
            const SP<uint8_t> p2 = s_cons(cfg, S) + (size_t)ch2 * CONS_LEN;
            for (int i = 0; i < n && same; i++) same = p2[i] == L.b.cons[chosen][i];
            for (uint32_t i = 0; i < L.n_marks && same; i++) same = W2.is_marked(L.inv[i]);
        }
        if (W2.status == GAP_OK && same && n <= 0) same = W2.n_marked == nm0;
        if (!same) sv.flags |= 0x40000000u;
        /* the general code has made the marks: undo them, the group form makes its own below (the same ones) */
        for (uint32_t i = nm0; i < W2.n_marked; i++) s_marked(cfg, S)[s_marklog(cfg, S)[i]] = 0;
    }
#endif
    if (n > 0) {
        coop_apply_marks<G>(W, L);
        /* the consensus where the walk expects it: consensus 0 of the gap's area in HBM (the lanes of the group write it together) */
        const SP<uint8_t> cons = s_cons(cfg, S);
#ifdef MTG_EMU_LANES
        for (int i = 0; i < n; i++) cons[(size_t)i] = L.b.cons[chosen][i]; /* outside the group run: one lane */
#else
        for (int i = (int)GP::gl(); i < n; i += GP::N) cons[(size_t)i] = L.b.cons[chosen][i];
#endif
    }
    if (GP::gl() == 0) bubble_store(cfg, S, W, sv, n, 0);
    return true;
}
MTG_DEV void bubble_classic(const Index& ix, const FillCfg& cfg, const GapScratch& S)
{
    WalkSave sv = *s_save(cfg, S);
    Worker W(ix, cfg, S);
    bubble_load(W, sv);
    const Kmer cur = make_kmer(sv.cur_f, ix.k);
    if (bubble_snp(ix, cfg, S, W, sv, cur, true)) return;
    if (W.status) { sv.flags |= 0x40000000u; bubble_store(cfg, S, W, sv, 0, -1); return; } /* TEST-ONLY emulation: a cross-check of the fast path failed */
    int chosen = -1;
    int n = explore_branching(W, cur, sv.prev_c, chosen);
    if (W.status) { n = 0; sv.flags |= 0x80000000u; } /* a work area of this scratch tier overflowed: the walk ends the gap with that status */
    bubble_store(cfg, S, W, sv, n, chosen);
#if defined(MTG_STAMPS) && !defined(MTG_EMU)
    for (int i = 0; i < 15; i++) if (W.stamp_acc[i]) atomicAdd(&g_stamps[i], W.stamp_acc[i]);
#endif
}

} // namespace mtg
#endif
