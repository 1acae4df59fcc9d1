/*
 * mtg_format.h -- the text of the tool's output files, produced where the results are: on the device.
 *
 * The reference's writers (writeFilledBreakpoint /root/reference/src/Filler.cpp:1029-1093, writeVcf :1095-1214, the info line :1082-1090)
 * run on the host behind fprintf; in this library they held `MindTheGap fill` at a sixteenth of the library's rate (10-15 CPU-ms per batch
 * of 100 000 sites for 88 MB of text).  For the site every breakpoint file is full of -- ONE solution, found by the forward attempt -- the
 * three pieces of text are a function of the site's name, its source k-mer, its record and its sequence, all of which the device holds
 * when k_emit is done: format_site below writes them, one wave per site, the small parts by lane 0 and the sequence copies by all lanes
 * (64 consecutive bytes per instruction).  It runs twice: with a counting sink (sizes -> exclusive prefix sums in site order -> the
 * arenas ARE the files' next bytes), then with the writing one.  Every other site (no solution: the reverse attempt follows; several
 * solutions; a record the host wrote) is left to the host's writers, which are the specification: the emulation build formats every
 * simple site both ways and compares.
 *
 * "%.2f" without printf: a float promoted to double times 100.0 is exact (24 + 7 significant bits), so floor and the comparison of the
 * fraction with one half are exact, and glibc's round-half-to-even on the exact binary value is reproduced bit for bit.
 */
#ifndef MTG_FORMAT_H
#define MTG_FORMAT_H
#include "mtg_post.h"
#include <math.h>

namespace mtg {

enum { FMT_FASTA = 0, FMT_INFO = 1, FMT_VCF = 2, FMT_STREAMS = 3 };

/* one site as the formatter sees it (pointers into device memory; host memory in the emulation) */
struct FmtSite {
    const char* name;     /* breakpointName: the left record's header up to its first space */
    uint32_t name_len;
    const char* source;   /* sourceSequence, as the file has it */
    uint32_t source_len;
    const char* seq;      /* the solution, seq_len characters */
    uint32_t seq_len;
    int32_t nb_nodes, total_nt, nb_terminal, has_counts, nb_total_filled, nb_reported;
    int32_t qual, solution_count;
    float avg, median;
};

/* sinks: all lanes of the wave call them with the same arguments */
struct FmtCount {
    uint32_t n[FMT_STREAMS];
    MTG_DEV void ch(int s, char) { n[s]++; }
    MTG_DEV void bytes(int s, const char*, uint32_t len) { n[s] += len; }
};
struct FmtWrite {
    char* p[FMT_STREAMS];
    MTG_DEV void ch(int s, char c) { if (MTG_LANE() == 0) *p[s] = c; p[s]++; }
    MTG_DEV void bytes(int s, const char* src, uint32_t len)
    {
        for (uint32_t i = MTG_LANE(); i < len; i += MTG_NLANES) p[s][i] = src[i];
        p[s] += len;
    }
};

template <typename Sink> MTG_DEV void fmt_lit(Sink& o, int s, const char* lit) { for (; *lit; lit++) o.ch(s, *lit); }
template <typename Sink> MTG_DEV void fmt_int(Sink& o, int s, long long v)
{
    char b[24];
    int n = 0;
    unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
    do { b[n++] = (char)('0' + (int)(u % 10)); u /= 10; } while (u);
    if (v < 0) b[n++] = '-';
    while (n) o.ch(s, b[--n]);
}
/* "%.2f" of a float (see the header); false: a value the shortcut does not cover (negative, huge, not finite): the site goes to the host */
MTG_DEV bool fmt_fixed2_ok(float x) { return x >= 0.0f && x < 1.0e9f; }
template <typename Sink> MTG_DEV void fmt_fixed2(Sink& o, int s, float x)
{
    const double t = (double)x * 100.0, fl = ::floor(t), fr = t - fl;
    const unsigned long long f0 = (unsigned long long)fl;
    const unsigned long long r = f0 + ((fr > 0.5 || (fr == 0.5 && (f0 & 1ull))) ? 1ull : 0ull);
    fmt_int(o, s, (long long)(r / 100));
    o.ch(s, '.');
    o.ch(s, (char)('0' + (int)((r / 10) % 10)));
    o.ch(s, (char)('0' + (int)(r % 10)));
}
/* atoi of a token that is not NUL-terminated */
MTG_DEV long long fmt_atoi(const char* p, uint32_t n)
{
    uint32_t i = 0;
    while (i < n && (p[i] == ' ' || (p[i] >= '\t' && p[i] <= '\r'))) i++;
    bool neg = false;
    if (i < n && (p[i] == '+' || p[i] == '-')) { neg = p[i] == '-'; i++; }
    long long v = 0;
    while (i < n && p[i] >= '0' && p[i] <= '9') { v = v * 10 + (p[i] - '0'); i++; }
    return (long long)(int)(neg ? -v : v); /* atoi returns int */
}
MTG_DEV bool fmt_tok_is(const char* p, uint32_t n, const char* lit)
{
    uint32_t i = 0;
    for (; lit[i]; i++) if (i >= n || p[i] != lit[i]) return false;
    return i == n;
}

/* may the device write this site?  (one solution, a sequence and a source that are there, numbers the shortcut covers) */
MTG_DEV bool fmt_site_simple(const FmtSite& t)
{
    return t.solution_count <= 1 && t.seq_len > 0 && t.source_len >= 1 && fmt_fixed2_ok(t.avg) && fmt_fixed2_ok(t.median);
}

/* the three pieces of text of a simple site */
template <typename Sink> MTG_DEV void format_site(Sink& o, const FmtSite& t)
{
    /* ---- FASTA, src/Filler.cpp:1052-1054 as it prints on x86-64: >NAME_len_L_qual_Q_avg_cov_A_median_cov_M   SOLU */
    o.ch(FMT_FASTA, '>');
    o.bytes(FMT_FASTA, t.name, t.name_len);
    fmt_lit(o, FMT_FASTA, "_len_"); fmt_int(o, FMT_FASTA, (long long)t.seq_len);
    fmt_lit(o, FMT_FASTA, "_qual_"); fmt_int(o, FMT_FASTA, t.qual);
    fmt_lit(o, FMT_FASTA, "_avg_cov_"); fmt_fixed2(o, FMT_FASTA, t.avg);
    fmt_lit(o, FMT_FASTA, "_median_cov_"); fmt_fixed2(o, FMT_FASTA, t.median);
    fmt_lit(o, FMT_FASTA, "   \n"); /* one solution: the solution string is empty */
    o.bytes(FMT_FASTA, t.seq, t.seq_len);
    o.ch(FMT_FASTA, '\n');
    /* ---- info line, :1082-1090 with the infostring of :905,1012-1016 */
    o.bytes(FMT_INFO, t.name, t.name_len);
    o.ch(FMT_INFO, '\t');
    o.ch(FMT_INFO, '\t'); fmt_int(o, FMT_INFO, t.nb_nodes);
    o.ch(FMT_INFO, '\t'); fmt_int(o, FMT_INFO, t.total_nt);
    o.ch(FMT_INFO, '\t'); fmt_int(o, FMT_INFO, t.nb_terminal);
    if (t.nb_terminal > 0 && t.has_counts) { o.ch(FMT_INFO, '\t'); fmt_int(o, FMT_INFO, t.nb_total_filled); o.ch(FMT_INFO, '\t'); fmt_int(o, FMT_INFO, t.nb_reported); }
    o.ch(FMT_INFO, '\n');
    /* ---- VCF, :1095-1214 */
    const uint32_t srcn = t.source_len, slen = t.seq_len;
    uint32_t repeat = 0; /* longest common suffix of source and sequence, the sequence read circularly (:1107-1126) */
    {
        int i = (int)srcn - 1, j = (int)slen - 1;
        while (i > 0 && j >= 0) {
            if (t.source[i] != t.seq[j]) break;
            repeat++; i--; j--;
            if (j == -1) j = (int)slen - 1;
        }
    }
    const uint32_t tail0 = srcn - (repeat + 1), tail_n = repeat + 1, ins_n = slen + 1;
    /* the name cut at its underscores (:1165-1182): token i = name[tb[i], te[i]) */
    uint32_t tb[9], te[9], ntok = 0;
    {
        uint32_t b = 0;
        for (;;) {
            uint32_t e = b;
            while (e < t.name_len && t.name[e] != '_') e++;
            if (ntok < 9) { tb[ntok] = b; te[ntok] = e; }
            ntok++;
            if (e >= t.name_len) break;
            b = e + 1;
        }
        if (t.name_len > 0 && t.name[t.name_len - 1] == '_') ntok--; /* getline yields no empty last token */
    }
    const bool named = ntok == 7 || ntok == 8;
    const uint32_t gi = ntok == 7 ? 6u : 7u; /* the genotype token */
    if (named) o.bytes(FMT_VCF, t.name + tb[1], te[1] - tb[1]); else o.ch(FMT_VCF, '.');
    o.ch(FMT_VCF, '\t');
    if (named) { const uint32_t pi = ntok == 7 ? 3u : 4u; fmt_int(o, FMT_VCF, fmt_atoi(t.name + tb[pi], te[pi] - tb[pi]) - (long long)repeat); } else o.ch(FMT_VCF, '.');
    o.ch(FMT_VCF, '\t');
    if (!named) o.bytes(FMT_VCF, t.name, t.name_len);
    else { o.bytes(FMT_VCF, t.name + tb[0], te[0] - tb[0]); if (ntok == 8) o.bytes(FMT_VCF, t.name + tb[2], te[2] - tb[2]); }
    o.ch(FMT_VCF, '\t'); o.ch(FMT_VCF, t.source[tail0]); o.ch(FMT_VCF, '\t');
    {
        const uint32_t from_tail = tail_n < ins_n ? tail_n : ins_n;
        o.bytes(FMT_VCF, t.source + tail0, from_tail);
        if (ins_n > from_tail) o.bytes(FMT_VCF, t.seq, ins_n - from_tail);
    }
    fmt_lit(o, FMT_VCF, "\t.\tPASS"); /* one solution: never LOW_QUAL */
    fmt_lit(o, FMT_VCF, "\tTYPE=INS;LEN="); fmt_int(o, FMT_VCF, (long long)ins_n - 1);
    fmt_lit(o, FMT_VCF, ";QUAL="); fmt_int(o, FMT_VCF, t.qual);
    fmt_lit(o, FMT_VCF, ";NSOL="); fmt_int(o, FMT_VCF, t.solution_count);
    fmt_lit(o, FMT_VCF, ";NPOS="); fmt_int(o, FMT_VCF, (long long)repeat + 1);
    fmt_lit(o, FMT_VCF, ";AVK="); fmt_fixed2(o, FMT_VCF, t.avg);
    fmt_lit(o, FMT_VCF, ";MDK="); fmt_fixed2(o, FMT_VCF, t.median);
    fmt_lit(o, FMT_VCF, "\tGT\t");
    fmt_lit(o, FMT_VCF, !named ? "./." : fmt_tok_is(t.name + tb[gi], te[gi] - tb[gi], "HOM") ? "1/1" : "0/1");
    o.ch(FMT_VCF, '\n');
}

/* length of a NUL-terminated string, by the whole wave */
MTG_DEV uint32_t fmt_strlen(const char* s)
{
    for (uint32_t base = 0;; base += MTG_NLANES) {
        const int f = wave_first(s[base + MTG_LANE()] == 0);
        if (f >= 0) return base + (uint32_t)f;
    }
}
/* the name of a site: its header up to the first space */
MTG_DEV uint32_t fmt_name_len(const char* hdr, uint32_t hdr_len)
{
    uint32_t n = 0;
    while (n < hdr_len && hdr[n] != ' ') n++;
    return n;
}

/* what the size pass leaves per site and the scan turns into offsets */
struct FmtRec {
    uint32_t size[FMT_STREAMS]; /* bytes of FASTA / info / VCF text (0, 0, 0: the site is the host's) */
    uint32_t simple;
    uint64_t off[FMT_STREAMS];  /* exclusive prefix sums in site order (k_fmt_scan) */
};

} // namespace mtg
#endif
