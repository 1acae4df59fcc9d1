/*
 * mtg_emit.h -- the results of a batch, produced on the device in their final form:
 *   - where every gap's output goes (exclusive prefix sums over the slots of a part: no atomic reservation, and the sequences come out
 *     in gap order, which is the serialised form of a batch)
 *   - the filled sequence of the common case (contig0[k:pos], src/GraphAnalysis.cpp:386-423; reverse-complemented for a reverse attempt,
 *     src/Filler.cpp:998-1001, src/Utils.hpp:78-83) and the extension sequence (get_first_contig, src/Filler.cpp:1381-1407) as ASCII
 *   - the C-ABI records mtg_gap_result / mtg_filled (filled_insertion_t, src/Utils.hpp:46-104; infostring fields src/Filler.cpp:905,
 *     1012-1016; compute_qual src/Utils.hpp:85-103), pointers already those of the host arrays the device buffers are copied into
 * so that the host touches a gap only when it takes the multi-contig path or has to be re-run in a larger scratch tier.
 * Compiled for gfx950 and, TEST-ONLY, for tests/emu (one lane).
 */
#ifndef MTG_EMIT_H
#define MTG_EMIT_H
#include "../../include/mtg_fill.h"
#include "mtg_copy.h"

namespace mtg {

enum : uint8_t { GAPF_REPEATED = 1, GAPF_REVERSE = 2 };

/* what a gap contributes to the arrays of its batch */
MTG_DEV void emit_plan(const GapOut& o, const PostOut& p, bool want_all, int k, uint32_t& nw, uint32_t& nc, uint32_t& asc, uint32_t& ext)
{
    nw = nc = asc = ext = 0;
    if (o.status != GAP_OK) return;
    if (want_all || (p.fast == 0 && p.nb_terminal > 0)) { nw = o.n_words; nc = o.n_contigs; } /* every contig: stage-A entry / multi-contig path on the host */
    if (want_all) return;
    if (p.fast == 1) asc = p.pos - (uint32_t)k + 1u;                                            /* the fill and its NUL */
    else if (p.nb_terminal == 0 && o.n_contigs > 0 && p.clen0 > (uint32_t)k) ext = p.clen0 - (uint32_t)k + 1u; /* no terminal node: contig 0 past the seed */
}

/* totals of one part of a batch (device -> host) */
struct PartTot {
    uint64_t begin[4], end[4]; /* cursors before / after the part: dense words, dense metadata entries, sequence bytes, extension bytes */
    uint64_t lines, store_runs, run_nt, contig_nt, contig_words, post_lines, cov_kmers, copy_words, copy_cmds, cov_direct, n_lean;
    uint64_t copy_words_exec, copy_cmds_exec, scan_words; /* what k_copy executed and k_post scanned: nothing of a lean gap */
    uint32_t n_retry, n_general, n_filled, n_ext;
    uint32_t n_parked, n_branching; /* gaps the walk kernel's first launch parked / whose walk stood on a branching node at least once (the hints for the workspace's next launch) */
};

/* one slot's share of the sums (the scan kernel adds them up in slot order) */
struct SlotSums {
    uint64_t v[4];
};
MTG_DEV SlotSums slot_sums(const SlotRec& r)
{
    SlotSums s;
    s.v[0] = r.nw; s.v[1] = r.nc; s.v[2] = r.asc; s.v[3] = r.ext;
    return s;
}

/* ASCII of 16 consecutive nucleotides of a 2-bit packed sequence, first one in the lowest byte: codes = 32 bits, nucleotide i at bits 2i.
 * comp: the complement letters. */
MTG_DEV void ascii16(uint32_t codes, bool comp, uint32_t out[4])
{
    const uint32_t lut = comp ? 0x43414754u /* T G A C */ : 0x47544341u /* A C T G */;
MTG_UNROLL
    for (int q = 0; q < 4; q++) {
        uint32_t w = 0;
MTG_UNROLL
        for (int b = 0; b < 4; b++) w |= ((lut >> (8u * ((codes >> (2 * (4 * q + b))) & 3u))) & 0xFFu) << (8 * b);
        out[q] = w;
    }
}
/* 16 nucleotides starting at nucleotide i of the packed words w (reads one word past the one holding nucleotide i + 15 at most) */
MTG_DEV uint32_t codes16(const uint64_t* w, uint32_t i)
{
    const uint32_t s = 2u * (i & 31u);
    uint64_t v = w[i >> 5] >> s;
    if (s > 32u) v |= w[(i >> 5) + 1] << (64u - s);
    return (uint32_t)v;
}

/* dst[0, L) = ASCII of src nucleotides [from, from + L), reverse-complemented when rc; dst[L] = 0.  The lanes of a wave write aligned
 * 16-byte pieces of the destination (byte stores at the ragged ends), so the arena needs no padding between sequences. */
template <uint32_t GW> MTG_DEV void emit_ascii_g(const uint64_t* src, uint32_t from, uint32_t L, bool rc, char* dst, uint32_t lane)
{
    const uint64_t d0 = (uint64_t)(uintptr_t)dst;
    const uint32_t head = (uint32_t)((16u - (d0 & 15u)) & 15u); /* bytes before the first aligned piece */
    const uint32_t npieces = 1u + (L > head ? (L - head + 15u) / 16u : 0u); /* piece 0 = the head (possibly empty) */
    for (uint32_t pc = lane; pc < npieces; pc += GW) {
        const uint32_t o0 = pc == 0 ? 0u : head + 16u * (pc - 1u);
        uint32_t n = pc == 0 ? (head < L ? head : L) : (L - o0 < 16u ? L - o0 : 16u);
        if (n == 0) continue;
        /* output characters o0 .. o0 + n - 1: forward = nucleotides from + o0 ..; reverse = complements of from + L - 1 - o0 downwards */
        uint32_t codes;
        if (!rc) codes = codes16(src, from + o0);
        else {
            const uint32_t hi = from + L - 1u - o0;      /* the nucleotide of the first output character */
            const uint32_t lo = hi >= 15u ? hi - 15u : 0u; /* 16 nucleotides ending at hi (fewer at the very start of the sequence) */
            const uint32_t got = hi - lo + 1u;
            codes = rev_fields32(codes16(src, lo)) >> (2u * (16u - got)); /* nucleotide hi first */
        }
        uint32_t out[4];
        ascii16(codes, rc, out);
        char* d = dst + o0;
        if (n == 16u) {
            U64x2 v;
            v.x = (uint64_t)out[0] | ((uint64_t)out[1] << 32);
            v.y = (uint64_t)out[2] | ((uint64_t)out[3] << 32);
            *reinterpret_cast<U64x2*>(d) = v; /* aligned: pc >= 1 */
        } else {
            for (uint32_t b = 0; b < n; b++) d[b] = (char)((out[b >> 2] >> (8u * (b & 3u))) & 0xFFu);
        }
    }
    if (lane == 0) dst[L] = 0;
}
/* by the lanes of a wave (one lane in the emulation) */
MTG_DEV void emit_ascii(const uint64_t* src, uint32_t from, uint32_t L, bool rc, char* dst) { emit_ascii_g<MTG_NLANES>(src, from, L, rc, dst, MTG_LANE()); }

/* host addresses of the arrays the device buffers of a batch are copied into */
struct EmitHost {
    char* seq;          /* sequence arena */
    char* ext;          /* extension arena; ext[0] is a NUL every record without extension points to (the arena's cursor starts at 1) */
    mtg_filled* fil;    /* one slot per gap (the common case has at most one solution) */
};
struct EmitDev {
    char* seq;
    char* ext;
    uint64_t seq_cap, ext_cap;
    mtg_gap_result* res; /* per slot of the launch */
    mtg_filled* fil;     /* per slot of the launch */
    uint64_t* dense_words;
    uint32_t* dense_meta;
    uint64_t dense_cap_words, dense_cap_contigs; /* room in the two dense arrays (words; contigs, five metadata entries each) */
    /* the batch in relocatable form (include/mtg_fill.h: mtg_wire_*), produced next to the records: nullptr = not wanted.  Only a batch
     * that is one launch in gap order can leave this way (the totals tell; the host falls back to mtg_results_to_wire otherwise). */
    uint8_t* wire;
    uint64_t wire_cap, wire_tag;
    const PartTot* tot;  /* the launch's totals (device memory, complete before k_emit starts) */
    uint32_t wire_gaps;  /* gaps of the batch */
};
/* where the sections of a relocatable batch begin */
struct WireLayout {
    uint64_t o_gaps, o_filled, o_seq, o_ext, total;
};
MTG_HD WireLayout wire_layout(uint64_t n_gaps, uint64_t n_filled, uint64_t seq_bytes, uint64_t ext_bytes)
{
    WireLayout w;
    w.o_gaps = sizeof(mtg_wire_header);
    w.o_filled = w.o_gaps + ((n_gaps * sizeof(mtg_wire_gap) + 7) & ~7ull);
    w.o_seq = w.o_filled + ((n_filled * sizeof(mtg_wire_filled) + 7) & ~7ull);
    w.o_ext = w.o_seq + ((seq_bytes + 7) & ~7ull);
    w.total = w.o_ext + ((ext_bytes + 7) & ~7ull);
    return w;
}
/* checksum of a payload's body (mtg_wire_header::checksum): word i contributes (w ^ i * C1) * C2, summed modulo 2^64 */
MTG_HD uint64_t wire_word_sum(uint64_t w, uint64_t i) { return (w ^ (i * 0x9E3779B97F4A7C15ull)) * 0xBF58476D1CE4E5B9ull; }

MTG_DEV int qual_of(uint32_t errors, bool repeated) /* compute_qual, src/Utils.hpp:85-103, for a single solution */
{
    int q = repeated ? 25 : 50;
    if (errors == 1) q = 10;
    if (errors == 2) q = 5;
    return q;
}

/* everything a gap leaves behind: one wave per gap (one lane in the emulation).  gap = its index in the batch. */
/* the record of the one solution the common path found (fast == 1): src/Filler.cpp:959-1003 */
MTG_DEV mtg_filled filled_of(const PostOut& p, uint64_t abase, uint32_t flags, const EmitHost& H)
{
    mtg_filled f;
    f.seq = H.seq + abase;
    f.nb_errors_in_anchor = (int)p.errors;
    f.target_index = (int)p.target;
#ifdef MTG_EMU
    f.avg_coverage = (float)p.ab_sum / (float)p.ab_n;
#else
    f.avg_coverage = __fdiv_rn((float)p.ab_sum, (float)p.ab_n);
#endif
    f.median_coverage = (p.ab_n & 1u) ? (float)p.med_hi : 0.5f * (float)(p.med_hi + p.med_lo); /* exact: integers and halves */
    f.qual = qual_of(p.errors, (flags & GAPF_REPEATED) != 0);
    f.solution_count = 1;
    f.solution_rank = 1;
    return f;
}

/* ---- The lean gap with its one solution, by GW lanes (k_emit_lean: eight lanes per gap, eight gaps per wave; one in the TEST-ONLY emulation,
 * which runs it next to emit_gap and compares): the fill is a stretch of the unitig store, the two records are what emit_gap writes for
 * fast == 1.  A wave per gap left two thirds of its lanes without a piece to write (300 characters are 20 pieces of 16) and spent the rest of
 * its instructions on lane 0's record.  `r` holds the slot's offsets relative to its scan block, `abase` is the absolute one; nothing is
 * written back to the slot record (the host looks at the records of the gaps it has to re-run or finish, never at a lean one's).
 * false: not such a gap (or the batch leaves in relocatable form): the general kernel's. */
MTG_DEV bool emit_is_lean(const SlotRec& r, const EmitDev& D) { return r.o.status == GAP_OK && r.p.lean != 0 && r.p.fast == 1 && r.nc == 0 && r.ext == 0 && D.wire == nullptr; }
template <uint32_t GW> MTG_DEV void emit_lean(const UStore& us, const FillCfg& cfg, const GapScratch& S, const SlotRec& r, uint64_t abase, uint32_t flags, uint32_t slot, uint64_t gap, int k,
                                              const EmitDev& D, const EmitHost& H, uint32_t gl)
{
    const bool reverse = (flags & GAPF_REVERSE) != 0;
    if (r.asc && D.seq && abase + r.asc <= D.seq_cap) { /* an arena that is too small: the host grows it and asks again */
        const CopyCmd cm = s_cmd(cfg, S)[r.p.lean - 1u];
        const uint32_t L = r.asc - 1u;
        const int64_t x0 = (int64_t)(32ull * s_cstart(cfg, S)[0]) + k;
        const bool bwd = (cm.src & 1ull) != 0;
        const uint64_t from = bwd ? cmd_store_nt(cm, x0 + (int64_t)L - 1) : cmd_store_nt(cm, x0);
        emit_ascii_g<GW>(us.words + (from >> 5), (uint32_t)(from & 31ull), L, bwd ? !reverse : reverse, D.seq + abase, gl);
    }
    if (gl != 0) return;
    mtg_gap_result g;
    g.nb_nodes = (int)r.o.n_contigs;
    g.total_nt = (int)r.o.total_nt;
    g.nb_terminal = (int)r.p.nb_terminal;
    g.has_solution_counts = 1;
    g.nb_total_filled = g.nb_reported = g.n_filled = 1;
    g.filled = H.fil + gap;
    g.extension = H.ext; /* "" */
    D.fil[slot] = filled_of(r.p, abase, flags, H);
    D.res[slot] = g;
}

MTG_DEV void emit_gap(const UStore& us, const FillCfg& cfg, const GapScratch& S, const SlotRec& r, uint32_t flags, uint32_t slot, uint64_t gap, int k, const EmitDev& D, const EmitHost& H)
{
    const uint32_t lane = MTG_LANE();
    const uint64_t* w = s_words(cfg, S);
    const bool reverse = (flags & GAPF_REVERSE) != 0;
    /* relocatable form: the sequence section of the payload IS the batch's sequence arena; the sections' places follow from the totals */
    WireLayout wl;
    wl.o_gaps = wl.o_filled = wl.o_seq = wl.o_ext = wl.total = 0;
    bool wire_ok = false;
    if (D.wire) {
        wl = wire_layout(D.wire_gaps, D.tot->n_filled, D.tot->end[2], D.tot->end[3]);
        wire_ok = wl.total <= D.wire_cap && D.tot->n_retry == 0 && D.tot->n_general == 0;
    }
    char* const seq_arena = wire_ok ? (char*)D.wire + wl.o_seq : D.seq;
    const uint64_t seq_cap = wire_ok ? D.tot->end[2] : D.seq_cap;
    const bool seq_ok = r.asc && seq_arena && r.abase + r.asc <= seq_cap, ext_ok = r.ext && r.ebase + r.ext <= D.ext_cap; /* an arena that is too small: the host grows it and asks again */
    if (seq_ok && r.p.lean) {
        /* the lean form: the contig was never materialised, its fill is a stretch of the unitig store (forward, or read backwards and complemented) */
        const CopyCmd cm = s_cmd(cfg, S)[r.p.lean - 1u];
        const uint32_t L = r.asc - 1u;
        const int64_t x0 = (int64_t)(32ull * s_cstart(cfg, S)[0]) + k;
        const bool bwd = (cm.src & 1ull) != 0;
        const uint64_t from = bwd ? cmd_store_nt(cm, x0 + (int64_t)L - 1) : cmd_store_nt(cm, x0);
        emit_ascii(us.words + (from >> 5), (uint32_t)(from & 31ull), L, bwd ? !reverse : reverse, seq_arena + r.abase);
    } else if (seq_ok) emit_ascii(w, (uint32_t)k, r.asc - 1u, reverse, seq_arena + r.abase);
    if (ext_ok) emit_ascii(w, (uint32_t)k, r.ext - 1u, false, D.ext + r.ebase);
    if (wire_ok && r.ext) emit_ascii(w, (uint32_t)k, r.ext - 1u, false, (char*)D.wire + wl.o_ext + r.ebase);
    /* the dense arrays are sized by the last need and k_emit runs before the launch's totals are known: a gap that does not fit writes
     * nothing, the host sees the totals, grows the arrays and emits the launch again (like the two arenas above) */
    const bool dense_ok = r.wbase + r.nw <= D.dense_cap_words && r.cbase + r.nc <= D.dense_cap_contigs;
    uint64_t* dw = D.dense_words + r.wbase;
    for (uint32_t i = lane; i < (dense_ok ? r.nw : 0u); i += MTG_NLANES) dw[i] = w[i];
    uint32_t* dm = D.dense_meta + 5 * r.cbase;
    for (uint32_t i = lane; i < (dense_ok ? r.nc : 0u); i += MTG_NLANES) {
        dm[i] = s_clen(cfg, S)[i];
        dm[r.nc + i] = s_cstart(cfg, S)[i];
        dm[2 * r.nc + i] = s_tpos(cfg, S)[i];
        dm[3 * r.nc + i] = s_terr(cfg, S)[i];
        dm[4 * r.nc + i] = s_ttgt(cfg, S)[i];
    }
    if (lane != 0) return;
    mtg_gap_result g;
    g.nb_nodes = (int)r.o.n_contigs;
    g.total_nt = (int)r.o.total_nt;
    g.nb_terminal = (int)r.p.nb_terminal;
    g.has_solution_counts = 0;
    g.nb_total_filled = g.nb_reported = g.n_filled = 0;
    g.filled = H.fil + gap;
    g.extension = H.ext; /* "" */
    if (r.o.status == GAP_OK) {
        if (r.p.fast == 1) {
            D.fil[slot] = filled_of(r.p, r.abase, flags, H);
            g.has_solution_counts = 1;
            g.nb_total_filled = g.nb_reported = g.n_filled = 1;
        } else if (r.p.fast == 2) {
            g.has_solution_counts = reverse ? 1 : 0; /* src/Filler.cpp:1012: counts are appended when something was found or on the reverse attempt */
        } else if (r.p.nb_terminal == 0) {
            if (r.ext) g.extension = H.ext + r.ebase;
        }
        /* fast == 0 with terminal nodes: the host fills the record in (multi-contig path) */
    }
    D.res[slot] = g;
    if (wire_ok) {
        mtg_wire_gap wg;
        wg.nb_nodes = g.nb_nodes; wg.total_nt = g.total_nt; wg.nb_terminal = g.nb_terminal; wg.has_solution_counts = g.has_solution_counts;
        wg.nb_total_filled = g.nb_total_filled; wg.nb_reported = g.nb_reported; wg.n_filled = g.n_filled;
        wg.first_filled = r.fpos;
        wg.ext_off = (r.o.status == GAP_OK && r.p.fast == 0 && r.p.nb_terminal == 0 && r.ext) ? r.ebase : 0ull;
        reinterpret_cast<mtg_wire_gap*>(D.wire + wl.o_gaps)[gap] = wg;
        if (g.n_filled == 1) {
            const mtg_filled& f = D.fil[slot];
            mtg_wire_filled wf;
            wf.seq_off = r.abase; wf.seq_len = r.asc - 1u;
            wf.nb_errors_in_anchor = f.nb_errors_in_anchor; wf.target_index = f.target_index; wf.qual = f.qual;
            wf.solution_count = 1; wf.solution_rank = 1;
            wf.avg_coverage = f.avg_coverage; wf.median_coverage = f.median_coverage;
            reinterpret_cast<mtg_wire_filled*>(D.wire + wl.o_filled)[r.fpos] = wf;
        }
        if (gap == 0) { /* the header (its checksum comes from wire_finish), the empty extension and the sections' padding */
            mtg_wire_header h;
            h.magic = 0x3145524957474D54ull; h.tag = D.wire_tag; h.n_gaps = D.wire_gaps; h.n_filled = D.tot->n_filled;
            h.seq_bytes = D.tot->end[2]; h.ext_bytes = D.tot->end[3]; h.total_bytes = wl.total; h.checksum = 0;
            *reinterpret_cast<mtg_wire_header*>(D.wire) = h;
            D.wire[wl.o_ext] = 0;
            for (uint64_t x = wl.o_seq + h.seq_bytes; x < wl.o_ext; x++) D.wire[x] = 0;
            for (uint64_t x = wl.o_ext + h.ext_bytes; x < wl.total; x++) D.wire[x] = 0;
        }
    }
}

} // namespace mtg
#endif
