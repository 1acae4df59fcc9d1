/* Marshalling of gapFillFromSource calls from a block of text, per gap and per dictionary entry: what FillInput::set_common / set_target
 * (mtg_host.cpp) do with the caller's strings on the host, as functions the device runs (k_marshal_text, k_marshal_targets in mtg_gpu_fill.hip;
 * the emulation build calls them on the host).  Same results bit for bit: the two paths are compared by the tests. */
#pragma once
#include "mtg_dev.h"
#include "mtg_post.h"

namespace mtg {

MTG_DEV uint32_t text_nt_code(uint32_t ch) { return (ch >> 1) & 3u; } /* A 0, C 1, T 2, G 3: bits 1-2 of the ASCII code, either case */

/* Gap: source = text[soff, soff + slen) (slen >= k), pattern = text[poff, poff + plen).
 *   src      the first k characters of the source as an oriented k-mer (first character in the highest field)
 *   rw       (plen + 31) / 32 + 1 words: the pattern's codes, character i at bits 2(i % 32) of word i / 32, the rest zero
 *   r0       the pattern's first k-mer (0 when it has fewer than k characters)
 *   rlen     plen, or 0xFFFFFFFF (and r0 = 0) for a pattern with a character other than A, C, G, T: the early stop is a literal search in
 *            upper-case contigs (IterativeExtensions [MEM]), such a pattern never matches
 *   fast_ok  the source is exactly k characters, none of them with bit 3 of its code set (nt_bad) */
MTG_DEV void marshal_text_gap(const uint8_t* text, uint64_t soff, uint32_t slen, uint64_t poff, uint32_t plen, int k, uint64_t* rw, uint64_t& src, uint64_t& r0, uint32_t& rlen,
                                    uint8_t& fast_ok)
{
    uint64_t s = 0;
    uint32_t bad = 0;
    for (int i = 0; i < k; i++) {
        const uint32_t c = text[soff + (uint64_t)i];
        s = (s << 2) | text_nt_code(c);
        bad |= c & 8u;
    }
    src = s;
    fast_ok = (slen == (uint32_t)k && !bad) ? 1 : 0;
    const uint32_t nw = (plen + 31u) / 32u + 1u;
    bool upper = true;
    uint64_t first = 0;
    for (uint32_t w = 0; w < nw; w++) {
        uint64_t v = 0;
        const uint32_t lo = w * 32u, hi = lo + 32u < plen ? lo + 32u : plen;
        for (uint32_t i = lo; i < hi; i++) {
            const uint32_t c = text[poff + i];
            v |= (uint64_t)text_nt_code(c) << (2u * (i - lo));
            upper = upper && (c == 'A' || c == 'C' || c == 'G' || c == 'T');
        }
        rw[w] = v;
        if (w == 0) first = v;
    }
    uint64_t r = 0;
    if (plen >= (uint32_t)k) { /* the k lowest fields of the first word, reversed: the first character in the highest field */
        for (int i = 0; i < k; i++) r = (r << 2) | ((first >> (2 * i)) & 3ull);
    }
    r0 = upper ? r : 0;
    rlen = upper ? plen : 0xFFFFFFFFu;
}

/* Dictionary entry: key = text[off, off + len).  Only its first k characters matter, and whether it has them (mtg_post.h: encode_target). */
MTG_DEV void marshal_text_target(const uint8_t* text, uint64_t off, uint32_t len, int k, uint64_t& le, uint64_t& bad)
{
    uint8_t slot[TARGET_SLOT];
    for (int i = 0; i < TARGET_SLOT; i++) slot[i] = 0;
    const bool usable = len >= (uint32_t)k;
    if (usable) for (int i = 0; i < k; i++) slot[i] = text[off + (uint64_t)i];
    slot[TARGET_SLOT - 1] = usable ? 1 : 0;
    encode_target(slot, k, le, bad);
}

} // namespace mtg
