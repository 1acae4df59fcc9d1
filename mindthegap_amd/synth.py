"""Synthetic benchmark sets of SURVEY.md 8(d) / BASELINE.md: an i.i.d. donor genome laid out as many short sequences
(prevents unbounded unitig walks), one insertion per chosen sequence, breakpoints = the 31-mers flanking it.

The index is built from the donor's k-mers directly (error-free, abundance = lo + hash % span, i.e. the
"donor k-mers with synthetic abundance" variant the survey allows instead of simulating 30x reads)."""
import numpy as np

NT = np.frombuffer(b"ACTG", dtype=np.uint8)  # code -> letter (A=0 C=1 T=2 G=3)


class SynthSet:
    """het_snps > 0: diploid variant -- sequences nseq/2 .. nseq-1 are second haplotypes of sequences 0 .. nseq/2-1 that differ by het_snps
    substitutions (>= 150 nt away from the insertion anchors and the ends), so every walk crosses SNP bubbles (multi-path bubbles
    are GATB-parity-unpinned logic; used as a secondary, divergence-heavy workload, never for the headline number)."""

    def __init__(self, nseq, n_sites, seq_len=5000, seed=1, ins_min=50, ins_max=1000, k=31, het_snps=0, het_indels=0, tips=0.0):
        nloci = nseq // 2 if het_snps else nseq
        assert n_sites <= nloci
        rng = np.random.default_rng(seed)
        self.k, self.nseq, self.n_sites, self.seq_len, self.het_snps = k, nseq, n_sites, seq_len, het_snps
        self.het_indels = het_indels
        self.ins_len = np.exp(rng.uniform(np.log(ins_min), np.log(ins_max), n_sites)).astype(np.int64)  # log-uniform
        self.pos = rng.integers(1000, seq_len - 1000 + 1, n_sites)
        self.lens = np.full(nseq, seq_len, dtype=np.uint32)
        self.lens[:n_sites] += self.ins_len.astype(np.uint32)
        self.words_per_seq = int((seq_len + ins_max + 31) // 32 + 1)
        # random 2-bit nucleotides, 32 per word
        self.words = rng.integers(0, 2**64, size=(nseq, self.words_per_seq), dtype=np.uint64)
        self.word_off = (np.arange(nseq, dtype=np.uint64) * np.uint64(self.words_per_seq))
        if het_snps:
            self.lens[nloci:2 * nloci] = self.lens[:nloci]
            self.words[nloci:2 * nloci] = self.words[:nloci]
            lens = self.lens[:nloci].astype(np.int64)
            pos = np.full(nloci, seq_len // 2, dtype=np.int64)
            pos[:n_sites] = self.pos
            ins = np.zeros(nloci, dtype=np.int64)
            ins[:n_sites] = self.ins_len
            for s_i in range(het_snps):
                # SNP s_i of every locus: uniform in a window left or right of the insertion, away from anchors and ends
                left = (np.arange(nloci) + s_i) % 2 == 0
                lo = np.where(left, 150, pos + ins + 150)
                hi = np.where(left, pos - 150, lens - 150)
                p = (lo + (rng.random(nloci) * np.maximum(hi - lo, 1)).astype(np.int64)).astype(np.int64)
                delta = rng.integers(1, 4, nloci).astype(np.uint64)  # add 1..3 to the 2-bit code
                w, b = p // 32, (2 * (p % 32)).astype(np.uint64)
                rows = np.arange(nloci) + nloci
                old = (self.words[rows, w] >> b) & np.uint64(3)
                new = (old + delta) & np.uint64(3)
                self.words[rows, w] = (self.words[rows, w] & ~(np.uint64(3) << b)) | (new << b)

        if het_snps and het_indels:
            # het_indels deletions of 1..3 nt per locus in the second haplotype (alternating left / right of the insertion like the SNPs, 100 nt
            # away from anchors, ends and -- mostly -- SNPs): bubbles whose branches differ in length (the general bubble code, not the SNP path)
            sh = (np.arange(32, dtype=np.uint64) * np.uint64(2))
            for j in range(nloci):
                row = nloci + j
                c = ((self.words[row][:, None] >> sh[None, :]) & np.uint64(3)).astype(np.uint8).reshape(-1)[: int(self.lens[row])]
                p0, L = int(pos[j]), int(lens[j])
                cut = []
                for d_i in range(het_indels):
                    lo, hi = (100, p0 - 100) if (j + d_i) % 2 == 0 else (p0 + int(ins[j]) + 100, L - 100)
                    if hi - lo < 10:
                        continue
                    q = int(lo + rng.integers(0, hi - lo))
                    cut.append((q, int(rng.integers(1, 4))))
                keep = np.ones(len(c), dtype=bool)
                for q, n in cut:
                    keep[q:q + n] = False
                c = c[keep]
                self.lens[row] = len(c)
                pad = np.zeros(self.words_per_seq * 32, dtype=np.uint64)
                pad[: len(c)] = c
                self.words[row] = (pad.reshape(-1, 32) << sh[None, :]).sum(axis=1, dtype=np.uint64)

        # tips > 0: erroneous fragments, what sequencing errors that survive the solidity cut-off leave in a real graph: `tips` fragments per
        # donor sequence, each a copy of k+1 .. k+45 donor nucleotides with one substitution -- near an end of the fragment a dead-end branch
        # (a tip, up to k nodes), in its middle a bubble of k nodes.  They are extra sequences of the index (extra_*), never sites.
        self.extra_words = np.zeros((0, 3), dtype=np.uint64)
        self.extra_lens = np.zeros(0, dtype=np.uint32)
        self.extra_rows = np.zeros(0, dtype=np.int64)
        if tips > 0:
            nfrag = int(tips * nseq)
            frng = np.random.default_rng(seed + 1000003)
            rows = frng.integers(0, nseq, nfrag)
            flen = frng.integers(k + 1, k + 46, nfrag)
            p = (frng.random(nfrag) * (self.lens[rows].astype(np.int64) - flen)).astype(np.int64)
            q = (frng.random(nfrag) * flen).astype(np.int64)
            w0, sh_ = p // 32, (2 * (p % 32)).astype(np.uint64)
            W = np.stack([self.words[rows, np.minimum(w0 + i, self.words_per_seq - 1)] for i in range(4)], axis=1)
            hi_sh = (np.uint64(64) - sh_) % np.uint64(64)
            out = np.empty((nfrag, 3), dtype=np.uint64)
            for i in range(3):
                lo_part = W[:, i] >> sh_
                hi_part = np.where(sh_ > 0, W[:, i + 1] << hi_sh, np.uint64(0))
                out[:, i] = lo_part | hi_part
            # the substitution: add 1..3 to the code at q
            delta = frng.integers(1, 4, nfrag).astype(np.uint64)
            qw, qb = q // 32, (2 * (q % 32)).astype(np.uint64)
            idx = np.arange(nfrag)
            old_c = (out[idx, qw] >> qb) & np.uint64(3)
            out[idx, qw] = (out[idx, qw] & ~(np.uint64(3) << qb)) | (((old_c + delta) & np.uint64(3)) << qb)
            self.extra_words, self.extra_lens = out, flen.astype(np.uint32)
            self.extra_rows = rows  # the donor sequence every fragment is a corrupted copy of

    def packed(self):
        """(words, word offsets, lengths, number of sequences) of everything that goes into the index: the donor and the erroneous fragments"""
        if len(self.extra_lens) == 0:
            return self.words.reshape(-1), self.word_off, self.lens, self.nseq
        base = np.uint64(self.words.size)
        words = np.concatenate([self.words.reshape(-1), self.extra_words.reshape(-1), np.zeros(2, dtype=np.uint64)])
        off = np.concatenate([self.word_off, base + np.arange(len(self.extra_lens), dtype=np.uint64) * np.uint64(3)])
        return words, off, np.concatenate([self.lens, self.extra_lens]), self.nseq + len(self.extra_lens)

    def extra_ascii(self, j):
        w = self.extra_words[j]
        sh = (np.arange(32, dtype=np.uint64) * np.uint64(2))
        c = ((w[:, None] >> sh[None, :]) & np.uint64(3)).astype(np.uint8).reshape(-1)
        return NT[c[: int(self.extra_lens[j])]].tobytes().decode()

    @property
    def total_kmers_upper_bound(self):
        return int(self.lens.astype(np.int64).sum() - (self.k - 1) * self.nseq) + int((self.extra_lens.astype(np.int64) - (self.k - 1)).sum())

    def codes(self, j):
        """2-bit codes of donor sequence j"""
        w = self.words[j]
        sh = (np.arange(32, dtype=np.uint64) * np.uint64(2))
        c = ((w[:, None] >> sh[None, :]) & np.uint64(3)).astype(np.uint8).reshape(-1)
        return c[: int(self.lens[j])]

    def ascii(self, j):
        return NT[self.codes(j)].tobytes().decode()

    def site(self, i):
        """(left k-mer, right k-mer, expected inserted sequence) of site i (donor sequence i)."""
        s = self.ascii(i)
        p, L, k = int(self.pos[i]), int(self.ins_len[i]), self.k
        return s[p - k:p], s[p + L:p + L + k], s[p:p + L]

    def site_name(self, i):
        # 7 '_' tokens so the VCF header parser path of src/Filler.cpp:1165-1172 is exercised
        return "bkpt%d_s%d_pos_%d_fuzzy_0_HOM" % (i, i, int(self.pos[i]))

    def write_breakpoints(self, path, sites=None):
        sites = range(self.n_sites) if sites is None else sites
        with open(path, "w") as f:
            for i in sites:
                l, r, _ = self.site(i)
                f.write(">%s left_kmer\n%s\n>%s right_kmer\n%s\n" % (self.site_name(i), l, self.site_name(i), r))


def simulate_reads(S, path, coverage=30, read_len=150, error_rate=0.0, seed=5):
    """SURVEY 8(d) cfg-2 reads: uniform start, both strands, substitution errors at error_rate (variant E0 = 0, E1 = 0.001).
    Writes a FASTA file; returns the number of reads."""
    rng = np.random.default_rng(seed)
    comp = np.array([2, 3, 0, 1], dtype=np.uint8)  # A<->T (0<->2), C<->G (1<->3)
    n = 0
    with open(path, "w") as f:
        for j in range(S.nseq):
            c = S.codes(j)
            L = len(c)
            nr = int(round(coverage * L / read_len))
            starts = rng.integers(0, L - read_len + 1, nr)
            strands = rng.integers(0, 2, nr)
            for st, sd in zip(starts, strands):
                r = c[st:st + read_len].copy()
                if error_rate > 0:
                    e = np.nonzero(rng.random(read_len) < error_rate)[0]
                    if len(e):
                        r[e] = (r[e] + rng.integers(1, 4, len(e)).astype(np.uint8)) & 3
                if sd:
                    r = comp[r[::-1]]
                f.write(">r%d\n%s\n" % (n, NT[r].tobytes().decode()))
                n += 1
    return n
