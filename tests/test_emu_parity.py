"""CPU-side logic tests: the device code (mtg_dev.h / mtg_traverse.h) executed by the TEST-ONLY host emulation
harness, plus the product's host code and CLI linked against it, checked against the oracle and the goldens.
These do not replace the -m gpu parity tests; they validate kernel and host logic before GPU time is spent."""
import contextlib
import ctypes as C
import os
import random
import struct

import numpy as np
import pytest

from tests import emu_lib, oracle_lib


@contextlib.contextmanager
def _env(name, value):
    """an environment variable (a test hook of the emulation build) set for the calls inside, None = unset"""
    old = os.environ.get(name)
    if value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = value
    try:
        yield
    finally:
        if old is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = old


def _read(p):
    with open(p) as f:
        return f.read()


def _vcf_body(p):
    return [l for l in _read(p).splitlines() if not l.startswith("##")]


def _rc(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def _rand_seq(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def _make_case(seed, k):
    """random genome with repeats, SNP / indel variants (bubbles) and erroneous fragments (tips)"""
    rng = random.Random(seed)
    g = _rand_seq(rng, rng.randrange(800, 4000))
    for _ in range(rng.randrange(0, 3)):
        a = rng.randrange(0, len(g) - 200)
        rl = rng.randrange(k - 5, 150)
        b = rng.randrange(0, len(g))
        g = g[:b] + g[a:a + rl] + g[b:]
    seqs = [g]
    for _ in range(rng.randrange(0, 4)):
        s = list(g)
        for _ in range(rng.randrange(0, 6)):
            p = rng.randrange(len(s))
            s[p] = rng.choice([c for c in "ACGT" if c != s[p]])
        s = "".join(s)
        for _ in range(rng.randrange(0, 3)):
            p = rng.randrange(50, len(s) - 50)
            s = s[:p] + _rand_seq(rng, rng.randrange(1, 40)) + s[p:] if rng.random() < 0.5 else s[:p] + s[p + rng.randrange(1, 40):]
        seqs.append(s)
    for _ in range(rng.randrange(0, 6)):
        p = rng.randrange(0, len(g) - k - 20)
        f = list(g[p:p + rng.randrange(k + 1, k + 45)])
        q = rng.randrange(len(f))
        f[q] = rng.choice([c for c in "ACGT" if c != f[q]])
        seqs.append("".join(f))
    return rng, g, seqs


@pytest.mark.parametrize("k", [31, 21, 16, 13])
def test_looping_contigs_and_target_inside_lookahead_runs(k):
    """circular simple paths: the walk comes back to its start node in the middle of a multi-nucleotide step; and, below the first BFS
    level, the first k-mer of the target is met in the middle of one (both are what the bulk step of phase W has to notice)"""
    rng = random.Random(1234 + k)
    for case in range(12):
        L = rng.randrange(k + 5, 400)
        c = _rand_seq(rng, L)
        seqs = [c + c[:k - 1]]  # every cyclic k-mer, nothing else: one simple cycle
        if case % 3 == 2:       # a branch off the cycle, so that contigs start below the first BFS level too
            p = rng.randrange(0, L - k)
            seqs.append(c[p:p + k - 1] + _rand_seq(rng, 1) + _rand_seq(rng, 200) + c[:k] + c[k:k + 60])
        idx = oracle_lib.Index.from_sequences(seqs, k, 1, 40)
        km, ct = idx.export()
        emu = emu_lib.EmuIndex(km, ct, k, load_factor=0.5)
        cc = c + c
        for _ in range(10):
            p = rng.randrange(0, L)
            s = cc[p:p + k] if rng.random() < 0.7 else _rc(cc[p:p + k])
            tp = rng.randrange(0, L)
            t = cc[tp:tp + k]
            for er in (0, 1):
                oc, _ = idx.stage_a(s, t, oracle_lib.default_params(end_rule_nonbranching=er))
                ec, st, _, _ = emu.stage_a(s, t, 100, 10000, er)
                assert st == 0 and ec == oc, (case, k, s, t, er)
        idx.close()


@pytest.mark.parametrize("k", [31, 21, 13])
def test_stage_a_fuzz_against_oracle(k):
    """bubbles, tips, repeats, loops, cut-offs (max_nodes / max_depth), both end rules, several table load factors"""
    for seed in range(25):
        rng, g, seqs = _make_case(seed, k)
        idx = oracle_lib.Index.from_sequences(seqs, k, 1, 40)
        km, ct = idx.export()
        emu = emu_lib.EmuIndex(km, ct, k, load_factor=rng.choice([0.3, 0.6, 0.9]))
        q = np.concatenate([km, np.array([rng.getrandbits(2 * k) for _ in range(200)], dtype=np.uint64)])
        ab, su, pr = emu.query(q)
        assert (ab == idx.abundance(q)).all()
        for _ in range(8):
            p = rng.randrange(0, len(g) - k)
            s = g[p:p + k] if rng.random() < 0.5 else _rc(g[p:p + k])
            tp = rng.randrange(0, len(g) - k)
            t = g[tp:tp + k] + (g[tp + k:tp + k + rng.randrange(1, 30)] if rng.random() < 0.2 else "")
            mn, md, er = rng.choice([100, 100, 5, 20]), rng.choice([10000, 10000, 300, 1500]), rng.choice([0, 0, 1])
            oc, _ = idx.stage_a(s, t, oracle_lib.default_params(max_nodes=mn, max_depth=md, end_rule_nonbranching=er))
            ec, st, _, _ = emu.stage_a(s, t, mn, md, er)
            assert st == 0 and ec == oc, (seed, k, s, t, mn, md, er)
        idx.close()
        emu.close()


@pytest.mark.parametrize("k", [31, 21, 20])
def test_junctions_the_scan_must_look_at_in_full(k):
    """round 6: for odd k the scan of the junction table judges an entry with one edge bit on each side by its bits alone, unless the entry carries the
    flag that the insertion gives the junctions for which that is not enough (jt_special: a palindromic (k-1)-mer -- one view instead of two --, a run of
    one nucleotide -- the self loop).  Sequences that walk straight through such junctions (one k-mer in, one out): statistics, queries and contigs
    against the oracle; the emulation also compares every unflagged entry with its key.  k = 20: even, no flags."""
    rng = random.Random(1000 + k)
    h = (k - 1) // 2
    seqs = []
    for c in "ACGT":
        seqs.append(_rand_seq(rng, 150) + c * (k + 20) + _rand_seq(rng, 150))     # a homopolymer run longer than k: the junction c^(k-1) entered and left by c
        seqs.append(_rand_seq(rng, 90) + c * (k - 1) + _rand_seq(rng, 90))         # the run as a junction between two ordinary k-mers
    for _ in range(6):
        half = _rand_seq(rng, h)
        pal = half + _rc(half) if (k - 1) % 2 == 0 else half + "A" + _rc(half)     # a (k-1)-mer that is its own reverse complement (k odd)
        seqs.append(_rand_seq(rng, 120) + pal + _rand_seq(rng, 120))
    seqs.append("AC" * 60 + _rand_seq(rng, 80))
    idx = oracle_lib.Index.from_sequences(seqs, k, 1, 40)
    km, ct = idx.export()
    for lf in (0.3, 0.9):
        emu = emu_lib.EmuIndex(km, ct, k, load_factor=lf)
        q = np.concatenate([km, np.array([rng.getrandbits(2 * k) for _ in range(100)], dtype=np.uint64)])
        ab, su, pr = emu.query(q)
        assert (ab == idx.abundance(q)).all()
        for g in seqs:
            for p0 in (0, 40, len(g) // 2 - k, len(g) - k - 60):
                for s_ in (g[p0:p0 + k], _rc(g[p0 + 30:p0 + 30 + k])):
                    t = g[-k:]
                    oc, _ = idx.stage_a(s_, t, oracle_lib.default_params())
                    ec, st, _, _ = emu.stage_a(s_, t, 100, 10000, 0)
                    assert st == 0 and ec == oc, (k, lf, s_)
        emu.close()
    idx.close()


@pytest.mark.parametrize("k", [31, 24, 17, 13])
def test_long_runs_through_the_unitig_store(k):
    """long unitigs (hundreds to thousands of k-mers) joined by forks the bubble code cannot merge, so that the BFS builds several
    contigs: walks take whole runs out of the unitig store in one step, in both orientations, and must still notice the start node
    (a cycle that leads back into the unitig the contig began in) and, below the first BFS level, the target's first k-mer in the
    middle of a run.  Checked against the oracle; the emulation build also looks at every node of every long step (status 0xBAD1)."""
    rng = random.Random(777 + k)
    for case in range(14):
        segs = [_rand_seq(rng, rng.randrange(150, 2200)) for _ in range(rng.randrange(3, 7))]
        g = "".join(segs)
        seqs = [g]
        cuts = np.cumsum([len(x) for x in segs])[:-1].tolist()
        for c in cuts:  # an alternative allele at every joint: unrelated sequence of another length (no merge within the bubble limits sometimes, a bubble otherwise)
            alt = _rand_seq(rng, rng.choice([1, 3, 40, 700]))
            seqs.append(g[max(0, c - 2 * k):c] + alt + g[c + rng.choice([0, 1, 30]):c + 3 * k + 30])
        if case % 3 == 0:  # close the genome into a cycle: a walk comes back to where it started, inside a long run
            seqs.append(g[-(k - 1):] + g[:k + 5])
        if case % 4 == 1:  # a repeat: the same long stretch twice (the second passage meets marked nodes / looping contigs)
            a = rng.randrange(0, len(g) - 400)
            seqs.append(g[a:a + 300] + _rand_seq(rng, 50) + g[a:a + 300])
        idx = oracle_lib.Index.from_sequences(seqs, k, 1, 40)
        km, ct = idx.export()
        emu = emu_lib.EmuIndex(km, ct, k, load_factor=0.5)
        for _ in range(10):
            p = rng.randrange(0, len(g) - k)
            s = g[p:p + k] if rng.random() < 0.6 else _rc(g[p:p + k])
            tp = rng.randrange(0, len(g) - k)
            t = g[tp:tp + k] if rng.random() < 0.6 else _rc(g[tp:tp + k])
            mn, md = rng.choice([100, 100, 8]), rng.choice([10000, 10000, 2500])
            oc, _ = idx.stage_a(s, t, oracle_lib.default_params(max_nodes=mn, max_depth=md))
            for own_copy in (False, True):  # the long runs left to the copy pass (k_copy) / copied by the walk itself
                with _env("MTG_NO_DEFER", "1" if own_copy else None):
                    ec, st, lines, _ = emu.stage_a(s, t, mn, md, 0)
                assert st == 0 and ec == oc, (case, k, s, t, mn, md, own_copy)
        idx.close()
        emu.close()


@pytest.mark.parametrize("k", [31, 22, 13])
def test_looping_contig_through_long_runs(k):
    """a cycle of a few thousand k-mers with short tips hanging off it: the tips are popped (explore_branching), so a walk goes all the
    way round and meets its start node in the middle of a long run of the unitig it began in, walking either way"""
    rng = random.Random(91 + k)
    for case in range(10):
        L = rng.randrange(400, 4000)
        c = _rand_seq(rng, L)
        seqs = [c + c[:k - 1]]
        cc = c + c
        for _ in range(rng.randrange(1, 4)):  # tips: k-1 nucleotides of the cycle, one different, a few more
            q = rng.randrange(k, L)
            seqs.append(cc[q:q + k - 1] + rng.choice([x for x in "ACGT" if x != cc[q + k - 1]]) + _rand_seq(rng, rng.randrange(0, k // 2)))
        idx = oracle_lib.Index.from_sequences(seqs, k, 1, 40)
        km, ct = idx.export()
        emu = emu_lib.EmuIndex(km, ct, k, load_factor=0.5)
        for _ in range(8):
            p = rng.randrange(0, L)
            s = cc[p:p + k] if rng.random() < 0.5 else _rc(cc[p:p + k])
            tp = rng.randrange(0, L)
            t = cc[tp:tp + k] if rng.random() < 0.5 else _rc(cc[tp:tp + k])
            oc, _ = idx.stage_a(s, t, oracle_lib.default_params(max_depth=100000))
            for own_copy in (False, True):
                with _env("MTG_NO_DEFER", "1" if own_copy else None):
                    ec, st, _, _ = emu.stage_a(s, t, 100, 100000, 0)
                assert st == 0 and ec == oc, (case, k, s, t, own_copy)
        idx.close()
        emu.close()


@pytest.mark.parametrize("k", [31, 21, 16])
def test_variant_pairs_at_distance_k(k):
    """two haplotypes that differ by substitutions 1, k-1, k, k+1, 2k-1 and 2k apart, and by three substitutions k apart: at distance
    exactly k the last nodes of the first bubble's branches both lead to both alleles of the next one (four, eight paths of 2k+1, 3k+1
    nodes: the general bubble code, whose path enumeration and marking advance whole unitigs; the emulation build runs the node-by-node
    forms next to them: statuses 0xBAD4..0xBAD8), below k the branches are longer single unitigs (the SNP fast path with an alignment
    bound instead of the alignment)"""
    rng = random.Random(1234 + k)
    other = {"A": "C", "C": "G", "G": "T", "T": "A"}
    for case in range(6):
        g = _rand_seq(rng, 4200)
        h = list(g)
        pos = 300
        for dist in (1, k - 1, k, k + 1, 2 * k - 1, 2 * k):
            for q in (pos, pos + dist):
                h[q] = other[h[q]]
            pos += dist + 6 * k + rng.randrange(0, 40)
        for j in range(3):  # three in a row, k apart
            h[pos + j * k] = other[h[pos + j * k]]
        assert pos + 3 * k < len(g) - 300
        seqs = [g, "".join(h)]
        idx = oracle_lib.Index.from_sequences(seqs, k, 1, 40)
        km, ct = idx.export()
        emu = emu_lib.EmuIndex(km, ct, k, load_factor=0.5)
        for s, t in ((g[100:100 + k], g[-150:-150 + k]), (_rc(g[-120:-120 + k]), _rc(g[150:150 + k])), ("".join(h)[100:100 + k], "".join(h)[2000:2000 + k])):
            oc, _ = idx.stage_a(s, t, oracle_lib.default_params(max_depth=100000))
            ec, st, _, _ = emu.stage_a(s, t, 100, 100000, 0)
            assert st == 0 and ec == oc, (case, k, s, t, hex(st))
        idx.close()
        emu.close()


@pytest.mark.parametrize("k", [31, 17])
def test_more_long_runs_than_copy_commands(k):
    """a contig that crosses more long unitigs (60, separated by tips the bubble code pops) than a gap has copy commands (32): the
    first runs are left to the copy pass, the others are copied by the walk itself, in both orientations"""
    rng = random.Random(5 + k)
    for case in range(4):
        g = _rand_seq(rng, 60 * 110 + 200)
        seqs = [g]
        for j in range(1, 60):
            q = j * 110 + rng.randrange(0, 20)
            seqs.append(g[q:q + k - 1] + rng.choice([x for x in "ACGT" if x != g[q + k - 1]]) + _rand_seq(rng, rng.randrange(0, k // 2)))
            seqs.append(_rc(_rc(g)[len(g) - q - 5:len(g) - q - 5 + k - 1] + "A"))  # some tips on the other strand (harmless when the k-mer exists)
        idx = oracle_lib.Index.from_sequences(seqs, k, 1, 40)
        km, ct = idx.export()
        emu = emu_lib.EmuIndex(km, ct, k, load_factor=0.5)
        for s, t in ((g[:k], g[-k:]), (_rc(g[-k:]), _rc(g[:k])), (g[300:300 + k], g[3000:3000 + k])):
            oc, _ = idx.stage_a(s, t, oracle_lib.default_params(max_depth=100000))
            ec, st, _, _ = emu.stage_a(s, t, 100, 100000, 0)
            assert st == 0 and ec == oc, (case, k, s, t)
            assert max(len(c) for c in oc) > 33 * 110, [len(c) for c in oc]  # one contig does cross more than 32 of them
        idx.close()
        emu.close()


def test_scratch_tier_escalation():
    """a gap that overflows the tier-0 contig arena is re-run in a larger tier with identical contigs"""
    rng = random.Random(5)
    g = _rand_seq(rng, 150000)
    idx = oracle_lib.Index.from_sequences([g], 31, 1, 40)
    km, ct = idx.export()
    emu = emu_lib.EmuIndex(km, ct, 31)
    oc, _ = idx.stage_a(g[:31], g[500:531])
    ec, st, _, tier = emu.stage_a(g[:31], g[500:531])
    assert st == 0 and tier >= 1 and ec == oc and len(oc[0]) == 150000
    _, st0, _, _ = emu.stage_a(g[:31], g[500:531], tier=0)
    assert st0 == 1  # GAP_OVF_CONTIG reported, never silently truncated


@pytest.fixture(scope="module")
def emu_product():
    from mindthegap_amd import lib as L
    saved = L._lib
    yield emu_lib.product_on_emulator()
    L._lib = saved


def test_cli_on_emulator_reproduces_goldens(emu_product, golden_dir, tmp_path):
    d = os.path.join(golden_dir, "data")
    F = emu_product.Filler()
    assert F.run(["-in", os.path.join(d, "reads_r1.fastq") + "," + os.path.join(d, "reads_r2.fastq"), "-bkpt", os.path.join(golden_dir, "full_test", "gold.breakpoints"),
                  "-out", str(tmp_path / "full")]) == 0
    assert _read(tmp_path / "full.insertions.fasta") == _read(os.path.join(golden_dir, "full_test", "gold.insertions.fasta"))
    assert _vcf_body(tmp_path / "full.insertions.vcf") == _vcf_body(os.path.join(golden_dir, "full_test", "gold.insertions.vcf"))
    assert F.run(["-in", os.path.join(d, "contig-reads.fasta.gz"), "-contig", os.path.join(d, "contigs.fasta"), "-abundance-min", "3", "-out", str(tmp_path / "ctg")]) == 0
    g = os.path.join(golden_dir, "contig_test")
    assert _read(tmp_path / "ctg.gfa") == _read(os.path.join(g, "gold.gfa"))
    assert _read(tmp_path / "ctg.insertions.fasta") == _read(os.path.join(g, "gold.insertions.fasta"))
    assert _read(tmp_path / "ctg_seed_dictionary.fasta") == _read(os.path.join(g, "gold_seed_dictionary.fasta"))
    assert F.run(["-bkpt", "x"]) == 1 and F.run(["-in", "a"]) == 1 and F.run(["-in", "a", "-graph", "b", "-bkpt", "x"]) == 1


def _write_idx(path, km, ct, k=31, amin=3):
    with open(path, "wb") as f:
        f.write(b"MTGIDX1\0" + struct.pack("<4i", k, amin, -1, 0) + struct.pack("<Q", len(km)) + km.tobytes() + ct.astype(np.uint32).tobytes())


def test_cli_on_emulator_reverse_attempts_and_extensions(emu_product, tmp_path):
    from mindthegap_amd.synth import SynthSet
    S = SynthSet(nseq=40, n_sites=30, seed=11)
    o = oracle_lib.Index.from_sequences([S.ascii(j) for j in range(S.nseq)], 31, 3, 40)
    km, ct = o.export()
    bk = str(tmp_path / "rev.breakpoints")
    with open(bk, "w") as f:
        for i in range(S.n_sites):
            l, r, _ = S.site(i)
            if i % 3 == 0:
                l = ("A" if l[0] != "A" else "C") + l[1:]
            if i % 3 == 1:
                l = l[:15] + ("A" if l[15] != "A" else "C") + l[16:]
            if i % 7 == 3:
                r = r[:10] + ("A" if r[10] != "A" else "C") + r[11:]
            f.write(">%s left_kmer\n%s\n>%s right_kmer\n%s\n" % (S.site_name(i), l, S.site_name(i), r))
    _write_idx(str(tmp_path / "rev.mtgidx"), km, ct)
    o.fill_files("bkpt", bk, str(tmp_path / "cpu"), params=oracle_lib.default_params(extend=1))
    assert emu_product.Filler().run(["-graph", str(tmp_path / "rev.mtgidx"), "-bkpt", bk, "-out", str(tmp_path / "hip"), "-extend"]) == 0
    for ext in (".insertions.fasta", ".info.txt", ".extensions.fasta"):
        assert _read(str(tmp_path / "hip") + ext) == _read(str(tmp_path / "cpu") + ext), ext
    assert _vcf_body(str(tmp_path / "hip.insertions.vcf")) == _vcf_body(str(tmp_path / "cpu.insertions.vcf"))
    assert _read(str(tmp_path / "hip.insertions.fasta")).count(">") >= 20
    o.close()


def test_cli_on_emulator_short_fills_every_alignment(emu_product, tmp_path):
    """round 4: the lean forms of the post-processing and of the result kernel (mtg_post.h: post_lean_*, mtg_emit.h: emit_lean) read a fill's
    abundance bytes as aligned 8-byte words and write its ASCII as aligned 16-byte pieces: fills of 1 .. 70 nucleotides, a third of them
    found by the reverse attempt (stretches read backwards), against the oracle's files (sum and median are in the FASTA headers).  The
    emulation runs both lean forms next to the general ones for every lean gap and compares records and bytes."""
    from mindthegap_amd.synth import SynthSet
    S = SynthSet(nseq=160, n_sites=140, seed=23, ins_min=1, ins_max=70)
    o = oracle_lib.Index.from_sequences([S.ascii(j) for j in range(S.nseq)], 31, 3, 40)
    km, ct = o.export()
    bk = str(tmp_path / "short.breakpoints")
    with open(bk, "w") as f:
        for i in range(S.n_sites):
            l, r, _ = S.site(i)
            if i % 3 == 1:
                l = l[:15] + ("A" if l[15] != "A" else "C") + l[16:]
            f.write(">%s left_kmer\n%s\n>%s right_kmer\n%s\n" % (S.site_name(i), l, S.site_name(i), r))
    _write_idx(str(tmp_path / "short.mtgidx"), km, ct)
    o.fill_files("bkpt", bk, str(tmp_path / "cpu"))
    assert emu_product.Filler().run(["-graph", str(tmp_path / "short.mtgidx"), "-bkpt", bk, "-out", str(tmp_path / "hip")]) == 0
    for ext in (".insertions.fasta", ".info.txt"):
        assert _read(str(tmp_path / "hip") + ext) == _read(str(tmp_path / "cpu") + ext), ext
    assert _vcf_body(str(tmp_path / "hip.insertions.vcf")) == _vcf_body(str(tmp_path / "cpu.insertions.vcf"))
    assert _read(str(tmp_path / "hip.insertions.fasta")).count(">") >= 125
    o.close()


def test_cli_on_emulator_bubbly_contig_mode(emu_product, tmp_path):
    """contig mode on a diploid-like graph (SNP bubbles + tips): multi-solution paths, NW dedupe, GFA order"""
    rng = random.Random(99)
    g = _rand_seq(rng, 6000)
    h = list(g)
    for p in range(300, 5800, 400):
        h[p] = rng.choice([c for c in "ACGT" if c != h[p]])
    h = "".join(h)
    h = h[:2500] + _rand_seq(rng, 120) + h[2500:]
    o = oracle_lib.Index.from_sequences([g, h, g[1000:1060] + "A"], 31, 3, 40)
    km, ct = o.export()
    contigs = str(tmp_path / "c.fa")
    with open(contigs, "w") as f:
        for i, (a, b) in enumerate([(0, 700), (1100, 1900), (2300, 2480), (3100, 3900), (4300, 5200)]):
            f.write(">ctg%d\n%s\n" % (i + 1, g[a:b]))
    _write_idx(str(tmp_path / "b.mtgidx"), km, ct)
    o.fill_files("contig", contigs, str(tmp_path / "cpu"))
    assert emu_product.Filler().run(["-graph", str(tmp_path / "b.mtgidx"), "-contig", contigs, "-out", str(tmp_path / "hip")]) == 0
    for ext in (".insertions.fasta", ".info.txt", ".gfa", "_seed_dictionary.fasta"):
        assert _read(str(tmp_path / "hip") + ext) == _read(str(tmp_path / "cpu") + ext), ext
    assert "solution" in _read(str(tmp_path / "hip.gfa")) or _read(str(tmp_path / "hip.gfa")).count("\nS\t") > 5
    o.close()


@pytest.mark.parametrize("err", [0.0, 0.003])
def test_cli_on_emulator_simulated_reads(emu_product, tmp_path, err):
    """cfg-2 shaped set at reduced size, index built from simulated 30x reads (-in path, abundance-min 3): variant E0 (error-free)
    and an error-laden variant whose surviving erroneous k-mers create tips and bubbles.  Both must equal the oracle byte for byte."""
    from mindthegap_amd.synth import SynthSet, simulate_reads
    S = SynthSet(nseq=12, n_sites=10, seed=21)
    reads = str(tmp_path / "reads.fa")
    simulate_reads(S, reads, coverage=30, error_rate=err, seed=5)
    bk = str(tmp_path / "s.breakpoints")
    S.write_breakpoints(bk)
    o = oracle_lib.Index.from_files([reads], 31, 3)
    o.fill_files("bkpt", bk, str(tmp_path / "cpu"))
    assert emu_product.Filler().run(["-in", reads, "-bkpt", bk, "-abundance-min", "3", "-out", str(tmp_path / "hip")]) == 0
    for ext in (".insertions.fasta", ".info.txt"):
        assert _read(str(tmp_path / "hip") + ext) == _read(str(tmp_path / "cpu") + ext), ext
    assert _vcf_body(str(tmp_path / "hip.insertions.vcf")) == _vcf_body(str(tmp_path / "cpu.insertions.vcf"))
    filled = _read(str(tmp_path / "hip.insertions.fasta")).count(">")
    assert filled >= 8
    if err == 0.0:  # error-free reads: every fill is exactly the inserted sequence
        seqs = [l for l in _read(str(tmp_path / "hip.insertions.fasta")).splitlines() if not l.startswith(">")]
        assert seqs == [S.site(i)[2] for i in range(S.n_sites)]
    o.close()


def _scan_case(mtg_mod):
    """shared by the emulator (CPU) and the GPU test: sequence scans against the oracle's exact membership"""
    from mindthegap_amd.synth import SynthSet
    S = SynthSet(nseq=30, n_sites=10, seed=17)
    seqs = [S.ascii(j) for j in range(S.nseq)]
    o = oracle_lib.Index.from_sequences(seqs[:20], 31, 3, 40)  # sequences 20..29 are absent from the graph
    km, ct = o.export()
    g = mtg_mod.Index.from_kmers(km, ct, 31)
    rng = random.Random(4)
    q = list(seqs[15:25])
    mutated = list(q[0])
    for p in range(100, len(mutated), 97):
        mutated[p] = "ACGT"[("ACGT".index(mutated[p]) + 1) % 4]
    q.append("".join(mutated))
    q.append(q[1][:700] + "N" + q[1][701:1500] + "nn" + q[1][1502:2000])
    q.append(q[2][:40])
    q.append("ACGT")
    exact, st = g.scan_sequences(q, exact=True)
    maybe, st0 = g.scan_sequences(q, exact=False)
    M = (1 << 62) - 1
    for s, e, m in zip(q, exact, maybe):
        n = max(len(s) - 30, 0)
        assert len(e) == n
        if n == 0:
            continue
        codes = np.array([(ord(c) >> 1) & 3 for c in s], dtype=np.uint64)
        bad = np.array([(ord(c) >> 3) & 1 for c in s], dtype=np.int64)
        kmers = np.zeros(n, dtype=np.uint64)
        f = 0
        for i, c in enumerate(codes.tolist()):
            f = ((f << 2) | c) & M
            if i >= 30:
                kmers[i - 30] = f
        truth = o.contains(kmers)
        cb = np.concatenate([[0], np.cumsum(bad)])
        truth = np.where((cb[31:31 + n] - cb[:n]) > 0, 0, truth).astype(np.uint8)
        assert (e == truth).all()              # exact mode == Graph::contains per position
        assert (m >= e).all()                  # the Bloom answer has no false negatives
    assert st["confirmed"] <= st["bloom_positive"] <= st["n_kmers"]
    neg = st0["n_kmers"] - st["confirmed"]
    assert (st0["bloom_positive"] - st["confirmed"]) <= 0.05 * neg + 5  # false-positive rate of the pre-filter
    assert st["blocks_staged"] * 3 < st["n_kmers"]  # minimizer coherence: far fewer blocks than k-mers
    g.close()
    o.close()


def test_sequence_scan_on_emulator(emu_product):
    _scan_case(emu_product)


def _contig_gap_case(mtg_mod, tmp_path, nseq, mutate=False):
    """contig mode at scale: every donor sequence is cut into two contigs around a hole; each seed sees the targets of all other
    contigs (2 * (#contigs) - 1 anchors), the fill must bridge the hole.  Files must equal the oracle's byte for byte.
    mutate: the right contigs' target k-mers (contig[31:62]) differ from the graph -- one, two or three substitutions, an N, lowercase -- so that the
    terminal search has to find them within nb_mis_allowed = 2 among the hundreds of targets of the dictionary (or not at all), src/Filler.cpp:1341-1351."""
    from mindthegap_amd.synth import SynthSet
    S = SynthSet(nseq=nseq, n_sites=nseq, seed=31)
    seqs = [S.ascii(j) for j in range(S.nseq)]
    o = oracle_lib.Index.from_sequences(seqs, 31, 3, 40)
    km, ct = o.export()
    contigs = str(tmp_path / "contigs.fa")

    def sub(t, i):
        return t[:i] + ("A" if t[i] != "A" else "C") + t[i + 1:]
    with open(contigs, "w") as f:
        for j, s in enumerate(seqs):
            p, L = int(S.pos[j]), int(S.ins_len[j])
            right = s[p + L:]
            if mutate:
                m = j % 8
                if m == 1: right = sub(right, 31 + 5)
                elif m == 2: right = sub(sub(right, 31), 31 + 30)          # the first and the last nucleotide of the target: the middle piece is intact
                elif m == 3: right = sub(sub(sub(right, 33), 44), 57)      # three: not found
                elif m == 4: right = right[:40] + "N" + right[41:]         # one forced mismatch
                elif m == 5: right = right[:31] + right[31:62].lower() + right[62:]
                elif m == 6: right = sub(sub(right, 31 + 9), 31 + 10)      # both in one piece
            f.write(">c%dL\n%s\n>c%dR\n%s\n" % (j, s[:p], j, right))
    idxf = str(tmp_path / "c.mtgidx")
    _write_idx(idxf, km, ct)
    o.fill_files("contig", contigs, str(tmp_path / "cpu"), params=oracle_lib.default_params(nb_cores=4))
    assert mtg_mod.Filler().run(["-graph", idxf, "-contig", contigs, "-out", str(tmp_path / "hip")]) == 0
    assert _read(str(tmp_path / "hip_seed_dictionary.fasta")) == _read(str(tmp_path / "cpu_seed_dictionary.fasta"))
    # the oracle ran on 4 threads: its records come in completion order, compare as multisets of lines / records
    assert sorted(_read(str(tmp_path / "hip.info.txt")).splitlines()) == sorted(_read(str(tmp_path / "cpu.info.txt")).splitlines())
    assert sorted(_read(str(tmp_path / "hip.gfa")).splitlines()) == sorted(_read(str(tmp_path / "cpu.gfa")).splitlines())
    assert sorted(_read(str(tmp_path / "hip.insertions.fasta")).splitlines()) == sorted(_read(str(tmp_path / "cpu.insertions.fasta")).splitlines())
    nfill = sum(1 for l in _read(str(tmp_path / "hip.gfa")).splitlines() if l.startswith("S") and ";" in l)
    assert nfill >= (nseq if mutate else 2 * nseq)  # both directions of every hole (mutated targets: the forward direction of five in eight at least)
    o.close()


def test_contig_mode_many_targets_on_emulator(emu_product, tmp_path):
    _contig_gap_case(emu_product, tmp_path, 8)


def test_contig_mode_inexact_targets_among_many_on_emulator(emu_product, tmp_path):
    """round 6: the piece index of the terminal search (mtg_post.h) on targets that match within two differences only; the emulation compares every
    contig's arg-max with the pass over all targets"""
    _contig_gap_case(emu_product, tmp_path, 16, mutate=True)


def _edge_case_files(tmp_path):
    """bkpt-mode inputs with the odd cases of src/Filler.cpp:623-699: REPEATED anchors (no mismatch allowed, qual 25), long / lowercase
    source records, 'N' inside an anchor, anchors with 1-2 mismatches, unfillable sites; on a diploid-like graph with SNP bubbles."""
    rng = random.Random(77)
    g = _rand_seq(rng, 9000)
    h = list(g)
    for p in range(500, 8500, 700):
        h[p] = rng.choice([c for c in "ACGT" if c != h[p]])
    h = "".join(h)
    ins = _rand_seq(rng, 260)
    donor = g[:4000] + ins + g[4000:]
    o = oracle_lib.Index.from_sequences([donor, h[:3000], h[5000:], g[2000:2040] + "T" + "ACGT" * 3], 31, 3, 40)
    km, ct = o.export()
    _write_idx(str(tmp_path / "e.mtgidx"), km, ct)
    L, R = donor[4000 - 31:4000], donor[4260:4260 + 31]

    def mut(s, i):
        return s[:i] + ("A" if s[i] != "A" else "C") + s[i + 1:]
    sites = [
        ("bkpt1_chr1_pos_10_fuzzy_0_HOM", L, R, ""),
        ("bkpt2_chr1_pos_20_fuzzy_0_HET", L, mut(R, 5), ""),                      # 1 mismatch in the anchor -> qual 10
        ("bkpt3_chr1_pos_30_fuzzy_0_HOM", L, mut(mut(R, 5), 20), ""),              # 2 mismatches -> qual 5
        ("bkpt4_chr1_pos_40_fuzzy_0_HOM", L, mut(mut(mut(R, 5), 20), 25), ""),     # 3 mismatches -> not found, reverse attempt
        ("bkpt5_chr1_pos_50_fuzzy_0_HOM", L, mut(R, 5), " REPEATED"),              # repeated: no mismatch allowed
        ("bkpt6_chr1_pos_60_fuzzy_0_HOM", L, R, " REPEATED"),                      # repeated exact -> qual 25
        ("bkpt7_chr1_pos_70_fuzzy_0_HOM", L.lower(), R, ""),                       # lowercase source
        ("bkpt8_chr1_pos_80_fuzzy_0_HOM", L + "ACGTACGTAC", R, ""),                # source record longer than k
        ("bkpt9_chr1_pos_90_fuzzy_0_HOM", L, R[:12] + "N" + R[13:], ""),           # N in the anchor: one forced mismatch
        ("bkpt10_chr1_pos_100_fuzzy_0_HOM", _rand_seq(rng, 31), _rand_seq(rng, 31), ""),  # nothing to find
        ("bkpt11_weird_name", donor[1000 - 31:1000], donor[1200:1231], ""),        # 200-nt stretch of the donor itself, odd header
        ("bkpt12_chr1_pos_120_fuzzy_0_HOM", g[2000 - 31 + 9:2000 + 9], g[6000:6031], ""),  # long walk through SNP bubbles
        ("bkpt13_chr1_pos_130_fuzzy_0_HOM", g[2000 - 31 + 9:2000 + 9], g[6000:6031].lower(), ""),  # lowercase anchor: matched case-insensitively, never by strstr
        ("bkpt14_chr1_pos_140_fuzzy_0_HOM", g[2000 - 31 + 9:2000 + 9], g[6000:6040], ""),  # anchor record longer than k
        ("bkpt15_chr1_pos_150_fuzzy_0_HOM", donor[1000 - 31:1000], donor[1000:1031], ""),  # adjacent anchors (target at pos == k): empty fill, no record
        ("bkpt16_chr1_pos_160_fuzzy_0_HOM", donor[1000 - 31:1000], donor[1000 - 10:1021], ""),  # overlapping anchors (pos < k)
        ("bkpt17_chr1_pos_170_fuzzy_0_HOM", donor[1000 - 31:1000], donor[1001:1032], ""),  # one nucleotide between the anchors
    ]
    bk = str(tmp_path / "edge.breakpoints")
    with open(bk, "w") as f:
        for name, l, r, tag in sites:
            f.write(">%s%s left_kmer\n%s\n>%s%s right_kmer\n%s\n" % (name, tag, l, name, tag, r))
    return o, str(tmp_path / "e.mtgidx"), bk


def _edge_case_run(mtg_mod, tmp_path):
    o, idxf, bk = _edge_case_files(tmp_path)
    for tag, extra, okw in (("a", [], {}), ("b", ["-fwd-only", "-filter", "-extend"], dict(fwd_only=1, filter=1, extend=1)),
                            ("c", ["-max-nodes", "3", "-max-length", "400"], dict(max_nodes=3, max_depth=400))):
        o.fill_files("bkpt", bk, str(tmp_path / ("cpu" + tag)), params=oracle_lib.default_params(**okw))
        assert mtg_mod.Filler().run(["-graph", idxf, "-bkpt", bk, "-out", str(tmp_path / ("hip" + tag))] + extra) == 0
        exts = [".insertions.fasta", ".info.txt"] + ([".extensions.fasta"] if "-extend" in extra else [])
        for ext in exts:
            assert _read(str(tmp_path / ("hip" + tag)) + ext) == _read(str(tmp_path / ("cpu" + tag)) + ext), (tag, ext)
        assert _vcf_body(str(tmp_path / ("hip%s.insertions.vcf" % tag))) == _vcf_body(str(tmp_path / ("cpu%s.insertions.vcf" % tag))), tag
    fa = _read(str(tmp_path / "hipa.insertions.fasta"))
    assert "_qual_50_" in fa and "_qual_10_" in fa and "_qual_5_" in fa and "_qual_25_" in fa
    o.close()


def test_cli_edge_cases_on_emulator(emu_product, tmp_path):
    _edge_case_run(emu_product, tmp_path)


def _diploid_case(mtg_mod, tmp_path, nloci):
    """diploid donor: every walk crosses heterozygous-SNP bubbles (consensus enumeration, NW identity, most-abundant choice)"""
    from mindthegap_amd.synth import SynthSet
    S = SynthSet(nseq=2 * nloci, n_sites=nloci, seed=13, het_snps=4)
    o = oracle_lib.Index.from_sequences([S.ascii(j) for j in range(S.nseq)], 31, 3, 40)
    km, ct = o.export()
    idxf = str(tmp_path / "d.mtgidx")
    _write_idx(idxf, km, ct)
    bk = str(tmp_path / "d.breakpoints")
    S.write_breakpoints(bk)
    o.fill_files("bkpt", bk, str(tmp_path / "cpu"))
    assert mtg_mod.Filler().run(["-graph", idxf, "-bkpt", bk, "-out", str(tmp_path / "hip")]) == 0
    for ext in (".insertions.fasta", ".info.txt"):
        assert _read(str(tmp_path / "hip") + ext) == _read(str(tmp_path / "cpu") + ext), ext
    assert _vcf_body(str(tmp_path / "hip.insertions.vcf")) == _vcf_body(str(tmp_path / "cpu.insertions.vcf"))
    assert _read(str(tmp_path / "hip.insertions.fasta")).count(">") >= nloci
    o.close()


def _allelic_inserts_case(mtg_mod, tmp_path, nloci):
    """two alleles per insertion site that the bubble code cannot merge (a 30-nt indel, 15 % substitutions, unrelated inserts): the BFS
    returns several contigs, the contig graph several paths per target, and remove_almost_identical_solutions aligns sequences of several
    hundred nucleotides (the device's k_nw in the product, its stand-in on the emulator)"""
    rng = random.Random(77)
    seqs, sites = [], []
    for i in range(nloci):
        L, R = _rand_seq(rng, 400), _rand_seq(rng, 400)
        a = _rand_seq(rng, rng.randrange(150, 900))
        kind = i % 4
        if kind == 0:    # long indel inside the insert
            p = rng.randrange(40, len(a) - 80)
            b = a[:p] + a[p + 30:]
        elif kind == 1:  # diverged copy
            b = "".join(c if rng.random() > 0.15 else rng.choice([x for x in "ACGT" if x != c]) for c in a)
        elif kind == 2:  # unrelated insert of another length
            b = _rand_seq(rng, rng.randrange(150, 900))
        else:            # indel + a few substitutions far apart
            p = rng.randrange(40, len(a) - 80)
            b = list(a[:p] + _rand_seq(rng, 35) + a[p:])
            for q in range(10, len(b) - 10, 97):
                b[q] = rng.choice([x for x in "ACGT" if x != b[q]])
            b = "".join(b)
        seqs += [L + a + R, L + b + R]
        sites.append((L[-31:], R[:31]))
    o = oracle_lib.Index.from_sequences(seqs, 31, 3, 40)
    km, ct = o.export()
    idxf = str(tmp_path / "al.mtgidx")
    _write_idx(idxf, km, ct)
    bk = str(tmp_path / "al.breakpoints")
    with open(bk, "w") as f:
        for i, (l, r) in enumerate(sites):
            f.write(">bkpt%d_s%d_pos_400_fuzzy_0_HOM left_kmer\n%s\n>bkpt%d_s%d_pos_400_fuzzy_0_HOM right_kmer\n%s\n" % (i, i, l, i, i, r))
    o.fill_files("bkpt", bk, str(tmp_path / "cpu"))
    assert mtg_mod.Filler().run(["-graph", idxf, "-bkpt", bk, "-out", str(tmp_path / "hip")]) == 0
    for ext in (".insertions.fasta", ".info.txt"):
        assert _read(str(tmp_path / "hip") + ext) == _read(str(tmp_path / "cpu") + ext), ext
    assert _vcf_body(str(tmp_path / "hip.insertions.vcf")) == _vcf_body(str(tmp_path / "cpu.insertions.vcf"))
    fa = _read(str(tmp_path / "hip.insertions.fasta"))
    assert "solution 2/2" in fa and fa.count(">") > nloci  # both alleles reported somewhere, one allele kept elsewhere
    o.close()


def test_cli_on_emulator_allelic_inserts(emu_product, tmp_path):
    _allelic_inserts_case(emu_product, tmp_path, 16)


def test_multi_contig_gaps_by_the_device_function(emu_product, tmp_path, monkeypatch):
    """mtg_general.h (k_general on the device): the multi-contig gaps of the allelic-insert set -- paths as a set, paths_to_sequences,
    remove_almost_identical_solutions with the wave's alignment, coverage, quality, ASCII -- are finished by the device function, and the host's
    path, run next to every one of them by the emulation build, agrees (HostChunk::gen_check: a disagreement is an error of the fill).  Then the
    same with work arenas too small for most gaps (they fall back to the host, as on the device when an arena overflows), with the host's path
    alone (HOST_GENERAL), on reverse attempts, and against the oracle."""
    rng = random.Random(78)
    seqs, gaps = [], []
    for i in range(24):
        L, R = _rand_seq(rng, 300), _rand_seq(rng, 300)
        a = _rand_seq(rng, rng.randrange(120, 700))
        if i % 3 == 0:
            p = rng.randrange(40, len(a) - 60)
            b = a[:p] + a[p + 30:]                      # the alleles align at > 90 %: one solution survives
        elif i % 3 == 1:
            b = _rand_seq(rng, rng.randrange(120, 700))  # unrelated alleles: two solutions
        else:
            b = a[:60] + _rand_seq(rng, 40) + a[60:]     # an insertion inside the insert
        seqs += [L + a + R, L + b + R]
        gaps.append((L[-31:], R[:31]))
    o = oracle_lib.Index.from_sequences(seqs, 31, 3, 40)
    km, ct = o.export()
    idx = emu_product.Index.from_kmers(km, ct, 31)

    def run(reverse):
        gl = []
        for (l, r) in gaps:
            if reverse:
                gl.append(emu_product.Gap(_rc(r), _rc(l), [(_rc(l), "x", False)], reverse=True))
            else:
                gl.append(emu_product.Gap(l, r, [(r, "x", False)]))
        res = idx.fill_batch(gl)
        return res, emu_product.last_batch_stats()

    res, st = run(False)
    assert st["n_general_device"] >= 12 and st["n_general_host"] == 0, st
    assert sum(1 for r in res if len(r["filled"]) == 2) >= 4 and sum(1 for r in res if len(r["filled"]) == 1) >= 4
    monkeypatch.setenv("MTG_HOST_GENERAL", "1")
    res_h, st_h = run(False)
    assert st_h["n_general_device"] == 0 and st_h["n_general_host"] >= 12
    assert res_h == res
    monkeypatch.delenv("MTG_HOST_GENERAL")
    monkeypatch.setenv("MTG_EMU_GEN_TINY", "1")
    res_t, st_t = run(False)
    assert st_t["n_general_host"] >= 6 and st_t["n_general_device"] >= 1, st_t
    assert res_t == res
    monkeypatch.delenv("MTG_EMU_GEN_TINY")
    res_r, st_r = run(True)
    assert st_r["n_general_device"] >= 12
    for a, b in zip(res, res_r):  # the reverse attempt reports the reverse complements of what it assembled: as many solutions, and the same inserts where both alleles survive
        assert len(a["filled"]) == len(b["filled"])
        if len(a["filled"]) == 2:
            assert sorted(f["seq"] for f in a["filled"]) == sorted(f["seq"] for f in b["filled"])
    idx.close()
    o.close()


def test_cli_on_emulator_diploid(emu_product, tmp_path):
    _diploid_case(emu_product, tmp_path, 10)


def test_banded_nw_equals_full_matrix(oracle):
    """the device NW (exact band, packed cells) against src/Utils.cpp:87-189 as restated by the oracle"""
    lib = emu_lib.load()
    rng = random.Random(123)
    for it in range(600):
        n = rng.randrange(1, 160)
        a = _rand_seq(rng, n)
        kind = it % 6
        if kind == 0:
            b = _rand_seq(rng, rng.randrange(1, 160))
        else:
            b = list(a)
            for _ in range(rng.randrange(0, 1 + kind * 2)):
                if not b:
                    break
                p = rng.randrange(len(b))
                r = rng.random()
                if r < 0.5:
                    b[p] = rng.choice("ACGT")
                elif r < 0.75:
                    del b[p:p + rng.randrange(1, 6)]
                else:
                    b[p:p] = list(_rand_seq(rng, rng.randrange(1, 6)))
            b = "".join(b) or "A"
        ident = oracle.mtgo_needleman_wunsch(a.encode(), b.encode())
        expect = round(ident * max(len(a), len(b)))
        assert lib.emu_nw_matches(a.encode(), b.encode()) == expect, (a, b)


def _abi_edge_cases(mtg_mod):
    """empty batches (plain and serialised), a source shorter than k, a sequence buffer that is too small"""
    s = _rand_seq(random.Random(11), 300)
    o = oracle_lib.Index.from_sequences([s], 31, 1, 40)
    km, ct = o.export()
    idx = mtg_mod.Index.from_kmers(km, ct, 31)
    assert idx.fill_batch([]) == []
    out = np.empty(1024, dtype=np.uint8)
    h, nf, nb = idx.fill_prepared_serial(mtg_mod.Index.prepare_gaps([]), out)
    idx.free_results(h)
    assert len(nf) == 0 and nb == 0
    with pytest.raises(mtg_mod.MtgError):
        idx.fill_batch([mtg_mod.Gap("ACGT", "ACGT", [("ACGT", "x", False)])])
    g = mtg_mod.Gap(s[:31], s[100:131], [(s[100:131], "t", False)])
    plain = idx.fill_batch([g])
    assert plain[0]["filled"] and plain[0]["filled"][0]["seq"] == s[31:100]
    with pytest.raises(mtg_mod.MtgError):
        idx.fill_prepared_serial(mtg_mod.Index.prepare_gaps([g]), np.empty(8, dtype=np.uint8))
    # several blocks of gaps through the input marshalling: the first batch of a shape makes the staging blocks grow (two passes), the
    # following ones are written in one pass; a malformed gap in a late block is refused whatever the blocks before it wrote, a larger
    # batch grows the blocks again, and the results never change
    rng = random.Random(12)
    many = []
    for _ in range(1500):
        a = rng.randrange(0, 150)
        b = rng.randrange(a + 40, 260)
        many.append(mtg_mod.Gap(s[a:a + 31], s[b:b + 31], [(s[b:b + 31], "t%d" % b, False), (_rand_seq(rng, 31), "decoy", False)]))
    want = idx.fill_batch(many)
    assert sum(1 for r in want if r["filled"]) > 1000
    for gq, r in zip(many, want):  # the walk from s[a:a+31] meets the target at b: the fill is what lies between
        a0, b0 = s.index(gq.source), s.index(gq.target)
        assert [f["seq"] for f in r["filled"]] == [s[a0 + 31:b0]], (a0, b0)
    assert idx.fill_batch(many) == want and idx.fill_batch(many) == want
    bad = list(many)
    bad[1300] = mtg_mod.Gap("ACGT", many[1300].target, many[1300].targets)
    with pytest.raises(mtg_mod.MtgError):
        idx.fill_batch(bad)
    assert idx.fill_batch(many) == want
    assert idx.fill_batch(many + many[:700]) == want + want[:700]
    assert idx.fill_batch(many[:600]) == want[:600]
    idx.close()
    o.close()


def test_abi_edge_cases_on_emulator(emu_product):
    _abi_edge_cases(emu_product)


def _snp_case(rng, k, kind):
    """a locus with two (or three) alleles around one site, built to sit on or just outside the pattern of the SNP fast path"""
    L, R = _rand_seq(rng, rng.randrange(k + 5, 120)), _rand_seq(rng, rng.randrange(k + 5, 160))
    sub = lambda c: rng.choice([x for x in "ACGT" if x != c])
    a = L + R
    p = len(L)
    alleles = [a]
    if kind == 0:      # one substitution
        alleles.append(a[:p] + sub(a[p]) + a[p + 1:])
    elif kind == 1:    # two substitutions closer than k
        q = p + rng.randrange(1, k)
        b = list(a); b[p] = sub(b[p]); b[min(q, len(b) - 2)] = sub(b[min(q, len(b) - 2)])
        alleles.append("".join(b))
    elif kind == 2:    # three alleles
        x = sub(a[p]); y = rng.choice([c for c in "ACGT" if c not in (a[p], x)])
        alleles += [a[:p] + x + a[p + 1:], a[:p] + y + a[p + 1:]]
    elif kind == 3:    # substitution next to a short indel
        alleles.append(a[:p] + sub(a[p]) + a[p + 1:p + 5] + a[p + 5 + rng.randrange(1, 4):])
    elif kind == 4:    # substitution inside a reverse-complement palindrome (canonical duplicates along the branches)
        h = _rand_seq(rng, rng.randrange(k // 2 + 2, k + 6))
        pal = h + _rc(h)
        a = L + pal + R
        p = len(L) + rng.randrange(0, len(pal))
        alleles = [a, a[:p] + sub(a[p]) + a[p + 1:]]
    elif kind == 5:    # substitution inside a tandem repeat (loops, nodes met twice)
        u = _rand_seq(rng, rng.randrange(3, k + 8))
        rep = u * rng.randrange(2, 5)
        a = L + rep + R
        p = len(L) + rng.randrange(0, len(rep))
        alleles = [a, a[:p] + sub(a[p]) + a[p + 1:]]
    elif kind == 6:    # the bubble twice: a repeat that contains the site (second passage meets marked nodes)
        core = a[p - k - 3:p + k + 3]
        a2 = a + _rand_seq(rng, 60) + core + _rand_seq(rng, 80)
        alleles = [a2, a2[:p] + sub(a2[p]) + a2[p + 1:]]
    elif kind == 7:    # a tip hanging off a branch node, and one off the node before the bubble
        b = a[:p] + sub(a[p]) + a[p + 1:]
        q = p + rng.randrange(1, k - 1)
        tip = b[q - k + 1:q + 1][:k - 1] + sub(b[q]) if q + 1 < len(b) else ""
        alleles += [b, b[q - k + 1:q] + sub(b[q]) + _rand_seq(rng, rng.randrange(0, 12))]
        alleles.append(a[p - k:p - 1] + sub(a[p - 1]) + _rand_seq(rng, rng.randrange(0, 10)))
    elif kind == 8:    # substitution close to the end of the sequence (a branch that dead-ends before the others meet)
        a = L + R[:rng.randrange(2, k + 2)]
        alleles = [a, a[:p] + sub(a[p]) + a[p + 1:]]
    elif kind == 10:   # both alleles are reverse-complement palindromes: the branches run into their own reverse complements and meet at rc(node)
        comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
        U = _rand_seq(rng, rng.randrange(k + 2, 70))
        W = _rand_seq(rng, rng.randrange(0, max(1, (k - 3) // 2)))
        x = rng.choice("ACGT"); y = sub(x)
        a = U + x + W + _rc(W) + comp[x] + _rc(U)
        alleles = [a, U + y + W + _rc(W) + comp[y] + _rc(U)]
    else:              # two independent substitutions further apart than k (two clean bubbles in a row)
        q = p + k + rng.randrange(1, 30)
        b = list(a)
        b[p] = sub(b[p])
        if q < len(b) - 1:
            b[q] = sub(b[q])
        alleles += ["".join(b), a[:p] + sub(a[p]) + a[p + 1:]]
    return a, [x for x in alleles if len(x) >= k]


@pytest.mark.parametrize("k", [31, 21, 16, 13])
def test_snp_fast_path_adversarial(k):
    """the SNP fast path of the traversal against the oracle's general bubble code: loci on and just outside its pattern, both strands,
    both end rules, abundances that differ between the alleles"""
    rng = random.Random(4242 + k)
    for case in range(286):
        kind = case % 11
        a, alleles = _snp_case(rng, k, kind)
        seqs = []
        for i, x in enumerate(alleles):  # unequal multiplicities give the alleles different abundances
            seqs += [x] * (1 + (case + i) % 3)
        idx = oracle_lib.Index.from_sequences(seqs, k, 1, 40)
        km, ct = idx.export()
        emu = emu_lib.EmuIndex(km, ct, k, load_factor=0.5)
        srcs = [a[:k], _rc(a[-k:])]
        if len(a) > 3 * k:
            srcs.append(a[k // 2:k // 2 + k])
        for s in srcs:
            for er in (0, 1):
                t = a[-k:]
                oc, _ = idx.stage_a(s, t, oracle_lib.default_params(end_rule_nonbranching=er))
                ec, st, _, _ = emu.stage_a(s, t, 100, 10000, er)
                assert st == 0 and ec == oc, (case, kind, k, s, er, seqs)
        idx.close()


def _container_v3_kmers(raw, k):
    """(canonical k-mers, abundances) of a version 3 index container: the stored unitigs expanded + the k-mers of no unitig"""
    assert raw[:8] == b"MTGIDX3\0"
    kk, amin, aauto, _ = struct.unpack("<4i", raw[8:24])
    nb_solid, nb_br, nb_sat, n_words, n_unitigs, n_left = struct.unpack("<6Q", raw[24:72])
    assert kk == k
    nw = n_words + 8 if n_words else 0
    words = np.frombuffer(raw[72:72 + 8 * nw], dtype="<u8")
    ab = np.frombuffer(raw[72 + 8 * nw:72 + 40 * nw], dtype=np.uint8)
    rec = np.frombuffer(raw[72 + 40 * nw:], dtype=np.dtype([("k", "<u8"), ("a", "<u4")]))
    assert len(rec) == n_left
    km, ct = [int(x) for x in rec["k"]], [int(x) for x in rec["a"]]
    h, nu = 0, 0
    while h < n_words:
        ln = int(words[h])
        codes = []
        for i in range(ln):
            p = (h + 1) * 32 + i
            codes.append((int(words[p >> 5]) >> (2 * (p & 31))) & 3)
        for i in range(ln - k + 1):
            f = 0
            for c in codes[i:i + k]:
                f = (f << 2) | c
            r = 0
            for c in reversed(codes[i:i + k]):
                r = (r << 2) | (c ^ 2)
            km.append(min(f, r))
            ct.append(int(ab[(h + 1) * 32 + i]))
        h += 1 + (ln + 31) // 32
        nu += 1
    assert nu == n_unitigs and len(km) == nb_solid
    return np.array(km, dtype=np.uint64), np.array(ct, dtype=np.uint32), n_unitigs


def test_index_container_written_from_the_tables(emu_product, tmp_path, monkeypatch):
    """mtg_index_save writes the index as it is (version 3 container): the unitig store -- 2-bit sequences and one abundance byte per k-mer --
    and the k-mers of no stored unitig (read back from the ABND table: lossless bucket + tag, inverted hash).  Together they are exactly the
    k-mers and abundances the index was built from; the file loads again (tables derived from the store), and files of versions 1 and 2
    (k-mer lists) still load"""
    rng = random.Random(8)
    for k in (31, 21, 12):
        seqs = [_rand_seq(rng, rng.randrange(k, 900)) for _ in range(6)]
        o = oracle_lib.Index.from_sequences(seqs, k, 1, 40)
        km, ct = o.export()
        ct = ct.copy()
        ct[::7] = 250 + (np.arange(len(ct[::7])) % 400).astype(ct.dtype)  # abundances up to and beyond the 8-bit ceiling
        g = emu_product.Index.from_kmers(km, ct, k)
        assert g.info()["nb_saturated"] == int((ct > 255).sum()) > 0  # counts above the ceiling are reported, not silently clamped
        p = str(tmp_path / ("i%d.mtgidx" % k))
        g.save(p)
        fk, fa, nu = _container_v3_kmers(open(p, "rb").read(), k)
        assert nu == g.info()["nb_unitigs"] > 0
        order = np.argsort(fk)
        assert (fk[order] == np.sort(km)).all()
        assert (fa[order] == np.minimum(ct[np.argsort(km)], 255)).all()
        h = emu_product.Index.load(p)
        q = np.concatenate([km, np.array([rng.getrandbits(2 * k) for _ in range(300)], dtype=np.uint64)])
        assert (h.abundance(q) == g.abundance(q)).all()
        hs, hp = h.neighbors(q)
        gs, gp = g.neighbors(q)
        assert (hs == gs).all() and (hp == gp).all() and h.info()["nb_unitigs"] == nu
        _write_idx(str(tmp_path / "v1.mtgidx"), km, ct, k=k)
        v1 = emu_product.Index.load(str(tmp_path / "v1.mtgidx"))
        assert (v1.abundance(q) == g.abundance(q)).all()
        with open(str(tmp_path / "v2.mtgidx"), "wb") as f:  # version 2: (k-mer, abundance) records
            f.write(b"MTGIDX2\0" + struct.pack("<4i", k, 3, -1, 0) + struct.pack("<Q", len(km)))
            f.write(np.rec.fromarrays([km, ct.astype(np.uint32)], dtype=np.dtype([("k", "<u8"), ("a", "<u4")])).tobytes())
        monkeypatch.setenv("MTG_LOAD_PIECE", "100")  # the reader hands the records over in pieces
        v2 = emu_product.Index.load(str(tmp_path / "v2.mtgidx"))
        assert (v2.abundance(q) == g.abundance(q)).all()
        # a damaged container is refused, not half-loaded
        raw = open(p, "rb").read()
        open(str(tmp_path / "cut.mtgidx"), "wb").write(raw[:-5])
        with pytest.raises(Exception):
            emu_product.Index.load(str(tmp_path / "cut.mtgidx"))
        for x in (g, h, v1, v2):
            x.close()
        o.close()


def test_serialised_batch_left_on_the_device(emu_product):
    """mtg_fill_prepared_serial_device on the emulation build (where device memory is host memory): the caller's buffer receives the same
    bytes as the host variant, in place for single-contig gaps and re-laid -- through the host and back -- when a gap takes the
    multi-contig path"""
    rng = random.Random(3)
    seqs, sites = [], []
    for i in range(24):
        L, R = _rand_seq(rng, 300), _rand_seq(rng, 300)
        a, b = _rand_seq(rng, 200 + i), _rand_seq(rng, 300 + i)
        seqs += [L + a + R, L + b + R] if i % 3 == 0 else [L + a + R]
        sites.append((L[-31:], R[:31]))
    o = oracle_lib.Index.from_sequences(seqs, 31, 3, 40)
    km, ct = o.export()
    idx = emu_product.Index.from_kmers(km, ct, 31)
    for sel in (range(1, 24, 3), range(24)):  # single-contig gaps only; all of them
        gaps = [emu_product.Gap(sites[i][0], sites[i][1], [(sites[i][1], "t%d" % i, False)]) for i in sel]
        out = np.empty(1 << 20, dtype=np.uint8)
        h, nf, nb = idx.fill_prepared_serial(emu_product.Index.prepare_gaps(gaps), out)
        idx.free_results(h)
        batch = idx.prepare_batch(gaps)
        dev = np.full(1 << 20, 0xEE, dtype=np.uint8)
        h, nf2, nb2 = idx.fill_prepared_serial_device(batch, dev.ctypes.data, dev.size)
        idx.free_results(h)
        assert nb2 == nb and (nf2 == nf).all() and dev[:nb2].tobytes() == out[:nb].tobytes() and nb > 0
        dev[:] = 0xEE  # a host copy next to the device buffer: both hold the serialised batch
        hcopy = np.full(1 << 20, 0xDD, dtype=np.uint8)
        h, nf3, nb3 = idx.fill_prepared_serial_device(batch, dev.ctypes.data, dev.size, host_out=hcopy)
        idx.free_results(h)
        assert nb3 == nb and dev[:nb3].tobytes() == out[:nb].tobytes() and hcopy[:nb3].tobytes() == out[:nb].tobytes()
        with pytest.raises(emu_product.MtgError):
            idx.fill_prepared_serial_device(batch, dev.ctypes.data, 64)
        batch.close()
    assert int(nf.max()) == 2  # the last selection held multi-contig gaps
    idx.close()
    o.close()


def test_stage_a_entry_of_the_product(emu_product):
    """mtg_stage_a_batch (contigs only, no records) through the product's host code on the emulated device"""
    rng = random.Random(4)
    g = _rand_seq(rng, 3000)
    seqs = [g, g[:1500] + ("A" if g[1500] != "A" else "C") + g[1501:], g[700:760] + "T"]
    o = oracle_lib.Index.from_sequences(seqs, 31, 1, 40)
    km, ct = o.export()
    idx = emu_product.Index.from_kmers(km, ct, 31)
    src = [g[0:31], g[100:131], _rc(g[2000:2031]), g[650:681]]
    tgt = [g[500:531], g[900:931], _rc(g[1000:1031]), g[1200:1231]]
    got = idx.stage_a(src, tgt)
    for s, t, c in zip(src, tgt, got):
        assert c == o.stage_a(s, t)[0], (s, t)
    idx.close()
    o.close()


def test_cli_on_two_emulated_devices(emu_product, tmp_path, monkeypatch):
    """the tool's multi-device driver (one host thread per device, an index replica each, sites dealt in batches, records written in
    input order) on the emulation build with two pretended devices and batches of a few sites: the files equal the oracle's (written
    with one thread, i.e. in input order), in both modes"""
    monkeypatch.setenv("MTG_EMU_DEVICES", "2")
    monkeypatch.setenv("MTG_CLI_BATCH", "5")
    (tmp_path / "e").mkdir(); (tmp_path / "c").mkdir()
    _edge_case_run(emu_product, tmp_path / "e")
    _contig_gap_case(emu_product, tmp_path / "c", 6)
    # the replicas are real copies: an index and its replica answer alike
    rng = random.Random(2)
    s = _rand_seq(rng, 400)
    o = oracle_lib.Index.from_sequences([s], 31, 1, 40)
    km, ct = o.export()
    a = emu_product.Index.from_kmers(km, ct, 31)
    import ctypes as C
    h = C.c_void_p()
    assert a.lib.mtg_index_replicate(a.h, 1, C.byref(h)) == 0
    b = emu_product.Index(h)
    assert (a.abundance(km) == b.abundance(km)).all() and a.info()["nb_unitigs"] == b.info()["nb_unitigs"]
    a.close(); b.close(); o.close()


def test_cli_on_eight_emulated_devices(emu_product, tmp_path, monkeypatch, capfd):
    """a node of eight: the index reaches the other seven devices as a doubling tree (0->1; 0->2, 1->3; 0->4, 1->5, 2->6, 3->7: three rounds, the
    pairs of a round at the same time), the files are the oracle's; a batch failing on device 5 stops the tool with exit code 1 and a prefix of
    the files; a REPLICATION failing (to device 6) stops it before any site is filled, with that replication's message"""
    monkeypatch.setenv("MTG_EMU_DEVICES", "8")
    monkeypatch.setenv("MTG_CLI_BATCH", "2")
    monkeypatch.setenv("MTG_CLI_IN_FLIGHT", "2")
    log = str(tmp_path / "replicas.log")
    monkeypatch.setenv("MTG_EMU_REPLICATE_LOG", log)
    from mindthegap_amd.synth import SynthSet
    S = SynthSet(nseq=48, n_sites=48, seed=23)
    o = oracle_lib.Index.from_sequences([S.ascii(j) for j in range(S.nseq)], 31, 3, 40)
    km, ct = o.export()
    idxf = str(tmp_path / "t.mtgidx")
    _write_idx(idxf, km, ct)
    bk = str(tmp_path / "t.breakpoints")
    S.write_breakpoints(bk)
    o.fill_files("bkpt", bk, str(tmp_path / "cpu"))
    assert emu_product.Filler().run(["-graph", idxf, "-bkpt", bk, "-out", str(tmp_path / "hip")]) == 0
    for ext in (".insertions.fasta", ".info.txt"):
        assert _read(str(tmp_path / "hip") + ext) == _read(str(tmp_path / "cpu") + ext), ext
    assert _vcf_body(str(tmp_path / "hip.insertions.vcf")) == _vcf_body(str(tmp_path / "cpu.insertions.vcf"))
    pairs = sorted(tuple(int(x) for x in l.split()) for l in open(log))
    assert pairs == sorted([(0, 1), (0, 2), (1, 3), (0, 4), (1, 5), (2, 6), (3, 7)])
    # fewer devices than the node has (-nb-gpus 5): 0->1; 0->2, 1->3; 0->4
    os.remove(log)
    assert emu_product.Filler().run(["-graph", idxf, "-bkpt", bk, "-out", str(tmp_path / "five"), "-nb-gpus", "5"]) == 0
    assert _read(str(tmp_path / "five.insertions.fasta")) == _read(str(tmp_path / "cpu.insertions.fasta"))
    assert sorted(tuple(int(x) for x in l.split()) for l in open(log)) == sorted([(0, 1), (0, 2), (1, 3), (0, 4)])
    # every batch of device 5 fails
    monkeypatch.setenv("MTG_CLI_BATCH", "1")
    monkeypatch.setenv("MTG_EMU_FAIL_DEVICE", "5")
    assert emu_product.Filler().run(["-graph", idxf, "-bkpt", bk, "-out", str(tmp_path / "bad")]) == 1
    full, part = _read(str(tmp_path / "cpu.insertions.fasta")), _read(str(tmp_path / "bad.insertions.fasta"))
    assert full.startswith(part) and len(part) < len(full)
    monkeypatch.delenv("MTG_EMU_FAIL_DEVICE")
    # the replication to device 6 fails: nothing is filled, the message is the replication's
    monkeypatch.setenv("MTG_EMU_FAIL_REPLICATE", "6")
    capfd.readouterr()
    assert emu_product.Filler().run(["-graph", idxf, "-bkpt", bk, "-out", str(tmp_path / "norep")]) == 1
    assert "replicating to device 6" in capfd.readouterr().err
    assert _read(str(tmp_path / "norep.insertions.fasta")) == ""
    o.close()


def test_cli_on_three_emulated_devices_with_a_failing_one(emu_product, tmp_path, monkeypatch):
    """the tool's driver (MTG_CLI_IN_FLIGHT host threads per device, batches handed out as a stream, text written in input order) on three
    pretended devices; then the same with every batch of device 1 failing: the tool stops with the reference's exit code 1 and the first
    error's message, and what it wrote before is a prefix of the complete files (nothing out of order, nothing after the failure)"""
    monkeypatch.setenv("MTG_EMU_DEVICES", "3")
    monkeypatch.setenv("MTG_CLI_BATCH", "3")
    monkeypatch.setenv("MTG_CLI_IN_FLIGHT", "2")
    from mindthegap_amd.synth import SynthSet
    S = SynthSet(nseq=40, n_sites=40, seed=21)
    o = oracle_lib.Index.from_sequences([S.ascii(j) for j in range(S.nseq)], 31, 3, 40)
    km, ct = o.export()
    idxf = str(tmp_path / "t.mtgidx")
    _write_idx(idxf, km, ct)
    bk = str(tmp_path / "t.breakpoints")
    S.write_breakpoints(bk)
    o.fill_files("bkpt", bk, str(tmp_path / "cpu"))
    assert emu_product.Filler().run(["-graph", idxf, "-bkpt", bk, "-out", str(tmp_path / "hip")]) == 0
    for ext in (".insertions.fasta", ".info.txt"):
        assert _read(str(tmp_path / "hip") + ext) == _read(str(tmp_path / "cpu") + ext), ext
    assert _vcf_body(str(tmp_path / "hip.insertions.vcf")) == _vcf_body(str(tmp_path / "cpu.insertions.vcf"))
    # the same input gzip-compressed, with the sequences of some records wrapped over several lines and CR LF line ends
    import gzip
    lines = _read(bk).splitlines()
    with gzip.open(str(tmp_path / "w.breakpoints.gz"), "wt", newline="") as f:
        for i, l in enumerate(lines):
            if not l.startswith(">") and i % 6 == 1:
                f.write(l[:10] + "\r\n" + l[10:20] + "\n" + l[20:] + "\n")
            else:
                f.write(l + ("\r\n" if i % 4 == 0 else "\n"))
    assert emu_product.Filler().run(["-graph", idxf, "-bkpt", str(tmp_path / "w.breakpoints.gz"), "-out", str(tmp_path / "gz")]) == 0
    for ext in (".insertions.fasta", ".info.txt"):
        assert _read(str(tmp_path / "gz") + ext) == _read(str(tmp_path / "cpu") + ext), ext
    # device 1 fails (one site per batch: forty batches, so that its two host threads certainly get some)
    monkeypatch.setenv("MTG_CLI_BATCH", "1")
    monkeypatch.setenv("MTG_EMU_FAIL_DEVICE", "1")
    assert emu_product.Filler().run(["-graph", idxf, "-bkpt", bk, "-out", str(tmp_path / "bad")]) == 1
    full = _read(str(tmp_path / "cpu.insertions.fasta"))
    part = _read(str(tmp_path / "bad.insertions.fasta"))
    assert full.startswith(part) and len(part) < len(full)
    assert _read(str(tmp_path / "cpu.info.txt")).startswith(_read(str(tmp_path / "bad.info.txt")))
    o.close()


def test_results_over_the_wire(emu_product):
    """the relocatable form of a result set (what the ranks of a multi-GPU job send to the rank that writes the files): serialised on the
    host (mtg_results_to_wire) and by the result kernel (mtg_fill_prepared_wire_device: the same bytes), rebuilt (mtg_results_from_wire:
    validated), and formatted by the tool's writers (mtg_format_bkpt) to the same text as the original records"""
    from mindthegap_amd import lib as L
    from mindthegap_amd.synth import SynthSet
    mtg = emu_product
    for het in (0, 4):  # haploid: every gap on the common path (device emission); diploid with allelic inserts: some through the general path
        S = SynthSet(nseq=60 if not het else 80, n_sites=30, seed=5, het_snps=het)
        o = oracle_lib.Index.from_sequences([S.ascii(j) for j in range(S.nseq)], 31, 3, 40)
        km, ct = o.export()
        idx = mtg.Index.from_kmers(km, ct, 31)
        sites, gaps = [], []
        for i in range(S.n_sites):
            l, r, _ = S.site(i)
            if i % 7 == 3:
                r = _rc(l)[:31]  # nothing to find: extension instead of a fill
            sites.append((S.site_name(i), S.site_name(i), l, r))
            gaps.append(mtg.Gap(l, r, [(r, S.site_name(i), False)]))
        params = mtg.FillParams()
        strings = mtg.Index.prepare_gaps(gaps)
        batch = idx.prepare_batch(strings, params)
        h, nf, _ = idx.fill_prepared(batch, params, want_seqs=False)
        direct = L.format_bkpt(sites, h, extend=True)
        assert direct["fasta"].count(b">") == int((nf > 0).sum()) and direct["info"].count(b"\n") == len(sites)
        payload = L.results_to_wire(h, 1234)
        hd = L.wire_header(payload)
        assert hd["tag"] == 1234 and hd["n_gaps"] == len(sites) and hd["total_bytes"] == payload.size
        w = L.WireResults(payload)
        assert w.tag == 1234
        assert L.format_bkpt(sites, w, extend=True) == direct
        w.close()
        # the result kernel's payload: byte for byte the host's
        buf = np.zeros(payload.size + 4096, dtype=np.uint8)
        h2, nf2, nb = idx.fill_prepared_wire_device(batch, 1234, buf.ctypes.data, buf.size, params)
        assert nb == payload.size and (nf2 == nf).all()
        assert buf[:nb].tobytes() == payload.tobytes()
        idx.free_results(h2)
        # a buffer that is too small for the payload is an error, not a truncated payload
        small = np.zeros(payload.size // 2, dtype=np.uint8)
        with pytest.raises(L.MtgError):
            idx.fill_prepared_wire_device(batch, 1, small.ctypes.data, small.size, params)
        # a damaged payload is refused
        bad = payload.copy()
        bad[bad.size // 2] ^= 1
        with pytest.raises(L.MtgError):
            L.WireResults(bad)
        bad = payload[: payload.size - 8].copy()
        with pytest.raises(L.MtgError):
            L.WireResults(bad)
        idx.free_results(h)
        batch.close(); idx.close(); o.close()


def test_text_batches_marshalled_by_the_device_code(emu_product):
    """mtg_fill_text on the emulator: tests/text_cases.py"""
    from tests import text_cases
    text_cases.run(emu_product, oracle_lib)


def test_cli_reads_its_breakpoint_file_mapped_and_with_read(emu_product, tmp_path, monkeypatch):
    """the tool's reader: a plain breakpoint file is mapped and every worker copies its batch (the default); MTG_CLI_NO_MMAP=1 reads it with
    read(2) in 8 MB pieces -- batches of three sites, so that records straddle the boundaries; both give the oracle's files (a .gz goes
    through zlib: test_cli_on_three_emulated_devices_with_a_failing_one)"""
    monkeypatch.setenv("MTG_CLI_BATCH", "3")
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    _edge_case_run(emu_product, tmp_path / "a")
    monkeypatch.setenv("MTG_CLI_NO_MMAP", "1")
    _edge_case_run(emu_product, tmp_path / "b")


def _lean_cases(rng, k):
    """graphs that meet every branch of the lean build: random genomes with bubbles and tips, circular simple paths (closed chains: the late
    pass), palindromic junctions, homopolymer loops, self-complementary k-mers (even k), single k-mers between two forks"""
    cases = []
    _, g, seqs = _make_case(rng.randrange(1 << 30), k)
    cases.append(("random", seqs))
    c = _rand_seq(rng, rng.randrange(3 * k, 400))
    cases.append(("circle", [c + c[:k - 1]]))                       # every junction simple: one closed chain, no chain start
    cases.append(("circle+genome", seqs + [c + c[:k - 1]]))
    half = _rand_seq(rng, (k - 1) // 2 + ((k - 1) & 1))
    pal = half[:(k - 1) // 2] + _rc(half[:(k - 1) // 2]) if (k - 1) % 2 == 0 else None
    if pal:
        cases.append(("palindromic junction", [_rand_seq(rng, 60) + pal + _rand_seq(rng, 60), _rand_seq(rng, 40) + pal + _rand_seq(rng, 40)]))
    cases.append(("homopolymer", [_rand_seq(rng, 50) + "A" * (k + 7) + _rand_seq(rng, 50), "T" * (2 * k)]))
    if k % 2 == 0:
        h = _rand_seq(rng, k // 2)
        cases.append(("self-complementary k-mer", [_rand_seq(rng, 70) + h + _rc(h) + _rand_seq(rng, 70)]))
    a, b = _rand_seq(rng, 200), _rand_seq(rng, 200)
    x = _rand_seq(rng, k)                                            # one k-mer with two ways in and two ways out
    cases.append(("single k-mer between forks", [a[:100] + x + a[100:], b[:100] + x + b[100:]]))
    cases.append(("short sequences", [_rand_seq(rng, k), _rand_seq(rng, k + 1), _rand_seq(rng, k - 1)]))
    return cases


@pytest.mark.parametrize("k", [31, 22, 16, 13])
def test_lean_build_equals_the_legacy_build(emu_product, tmp_path, k):
    """round 4: the index built from the junction table alone (no dense ADJ / ABND tables, chain starts and the k-mers of no unitig out of one
    streaming pass, abundances asked of their source) is the index the construction of rounds 1-3 builds: same statistics, same stored
    k-mers and abundances, same answers to every query -- on graphs with closed chains (the late pass), palindromic junctions,
    self-complementary k-mers, homopolymer loops and isolated k-mers"""
    rng = random.Random(400 + k)
    seen_late = False
    for rep in range(15):
        for name, seqs in _lean_cases(rng, k):
            o = oracle_lib.Index.from_sequences(seqs, k, 1, 40)
            km, ct = o.export()
            o.close()
            if len(km) == 0:
                continue
            ct = ct.copy()
            ct[::5] = 200 + (np.arange(len(ct[::5])) % 300).astype(ct.dtype)
            lean = emu_product.Index.from_kmers(km, ct, k)
            with _env("MTG_LEGACY_BUILD", "1"):
                old = emu_product.Index.from_kmers(km, ct, k)
            li, oi = lean.info(), old.info()
            for f in ("nb_solid_kmers", "nb_branching", "nb_unitigs", "nb_saturated"):  # (the legacy emulation does not count the k-mers outside unitigs)
                assert li[f] == oi[f], (name, k, f, li[f], oi[f], seqs)
            assert li["nb_solid_kmers"] == len(km)
            seen_late = seen_late or lean.build_profile()["phases"][0]["name"] == "emulated_lean_late"
            lean.save(str(tmp_path / "l.idx"))
            old.save(str(tmp_path / "o.idx"))
            lk, la, lu = _container_v3_kmers(open(str(tmp_path / "l.idx"), "rb").read(), k)
            ok, oa, ou = _container_v3_kmers(open(str(tmp_path / "o.idx"), "rb").read(), k)
            assert lu == ou and (np.sort(lk) == np.sort(ok)).all() and (la[np.argsort(lk)] == oa[np.argsort(ok)]).all(), (name, k)
            assert (la[np.argsort(lk)] == np.minimum(ct[np.argsort(km)], 255)).all()
            nb = []
            for x in km[:400]:
                x = int(x)
                for b in range(4):
                    nb.append(((x << 2) | b) & ((1 << (2 * k)) - 1))
                    nb.append((x >> 2) | (b << (2 * (k - 1))))
            q = np.concatenate([km, np.array(nb + [rng.getrandbits(2 * k) for _ in range(300)], dtype=np.uint64)])
            assert (lean.abundance(q) == old.abundance(q)).all(), (name, k)
            ls, lp = lean.neighbors(q)
            os_, op = old.neighbors(q)
            assert (ls == os_).all() and (lp == op).all(), (name, k)
            lean.close()
            old.close()
    assert seen_late  # a closed chain was met


@pytest.mark.parametrize("serial", [False, True])
def test_walkers_of_a_chain_meet_in_the_middle(emu_product, tmp_path, monkeypatch, serial):
    """round 5: the two walkers of a chain stop where they meet (marks every 32 junctions, mtg_build.h: JtWalker) and the unitig is the
    owner's sequence joined with the partner's reverse complement.  Long chains (hundreds to thousands of k-mers, their lengths around the
    multiples of 32 and of a chunk's 992 nucleotides), walkers stepped in turns -- they meet in the middle, near an end, on the very junction
    both mark -- or one after the other (the second walker meets the first one's mark after a few steps); the emulation compares every stored
    unitig with ONE walk of its chain, and the index answers every query like the dense reference construction."""
    if serial:
        monkeypatch.setenv("MTG_EMU_WALK_SERIAL", "1")
    rng = random.Random(900 + int(serial))
    k = 31
    before = emu_lib.walk_counts(full=True)
    for rep in range(6):
        seqs = []
        for L in [33, 61, 62, 63, 64, 65, 66, 95, 96, 97, 127, 128, 129, 500, 991, 992, 993, 1022, 1023, 1024, 1025, 2000, 2017, 3100, 4999]:
            seqs.append(_rand_seq(rng, L + k - 1 + rng.randrange(0, 3)))
        a = _rand_seq(rng, 3000)
        seqs += [a[:1500], a[1400:]]                      # a chain through the overlap of two sequences
        seqs.append(_rand_seq(rng, 700) + a[200:260] + _rand_seq(rng, 900))  # a repeat: the chains around it end at its forks
        o = oracle_lib.Index.from_sequences(seqs, k, 1, 40)
        km, ct = o.export()
        o.close()
        lean = emu_product.Index.from_kmers(km, ct, k)
        with _env("MTG_LEGACY_BUILD", "1"):
            old = emu_product.Index.from_kmers(km, ct, k)
        li, oi = lean.info(), old.info()
        for f in ("nb_solid_kmers", "nb_branching", "nb_unitigs"):
            assert li[f] == oi[f], (f, li[f], oi[f])
        q = np.concatenate([km, np.array([rng.getrandbits(2 * k) for _ in range(500)], dtype=np.uint64)])
        assert (lean.abundance(q) == old.abundance(q)).all()
        ls, lp = lean.neighbors(q)
        os_, op = old.neighbors(q)
        assert (ls == os_).all() and (lp == op).all()
        lean.close()
        old.close()
    met, whole = (x - y for x, y in zip(emu_lib.walk_counts(full=True), before))
    assert met >= 60 and whole >= 20, (met, whole)   # long chains are joined from two walks; chains under 32 junctions are walked whole


def test_wire_payload_with_wrapping_sizes_is_refused(emu_product):
    """mtg_results_from_wire bounds every section by the payload before it adds sizes up: a header whose seq_bytes is close to 2^64 makes
    the sum of the sections wrap back onto total_bytes (and carries a valid checksum -- it is not cryptographic); such a payload is
    MTG_ERR_FORMAT, not a read far outside the buffer"""
    from mindthegap_amd import lib as L
    M = (1 << 64) - 1
    body = np.zeros(5, dtype=np.uint64)  # one mtg_wire_filled (40 bytes): seq_off = 2^40, seq_len = 0
    body[0] = 1 << 40
    c1, c2 = 0x9E3779B97F4A7C15, 0xBF58476D1CE4E5B9
    cs = 0
    for i, w in enumerate(body.tolist()):
        cs = (cs + ((w ^ ((i * c1) & M)) * c2)) & M
    hdr = np.array([L.WIRE_MAGIC if hasattr(L, "WIRE_MAGIC") else 0x3145524957474D54, 7, 0, 1, (1 << 64) - 8, 8, 104, cs], dtype=np.uint64)
    payload = np.concatenate([hdr, body]).view(np.uint8)
    assert payload.size == 104
    with pytest.raises(Exception) as e:
        L.WireResults(payload)
    assert "FORMAT" in str(e.value) or "payload" in str(e.value)
    # and a well-formed payload still loads (test_results_over_the_wire covers the contents)


def test_device_formatted_text_equals_the_host_writers(emu_product, tmp_path):
    """round 4: `MindTheGap fill -bkpt` writes the text of every site with one solution on the device (mtg_format.h: FASTA header + sequence, info
    line, VCF line with its name tokens, repeat size, %.2f numbers); the host's writers -- the restatement of src/Filler.cpp:1029-1214 the
    goldens pin -- take the others.  The same run with MTG_HOST_FORMAT=1 (every site by the host) must give the same files, byte for
    byte: names with 7, 8 and other numbers of tokens, a trailing underscore, positions atoi reads in its own way, sites without solution
    (reverse attempt), -extend and -filter, batches of 4 sites so that host and device text interleave"""
    from mindthegap_amd.synth import SynthSet
    S = SynthSet(nseq=40, n_sites=36, seed=9)
    o = oracle_lib.Index.from_sequences([S.ascii(j) for j in range(S.nseq)], 31, 3, 40)
    km, ct = o.export()
    o.close()
    idxf = str(tmp_path / "s.mtgidx")
    _write_idx(idxf, km, ct)
    names = ["bkpt%d_chr1_pos_%d_fuzzy_0_HOM", "bkpt%d_chr2_x_pos_%d_fuzzy_0_HET", "plain%d_%d", "bkpt%d_chr1_pos_+%d_fuzzy_0_HET", "bkpt%d_chr1_pos_x%d_fuzzy_0_HOM",
             "bkpt%d_chr1_pos_%d_fuzzy_0_HOM_", "a%d_b_c_d_e_f_g_h_i_j_%d", "bkpt%d_chrX_pos_00%d_fuzzy_3_HET"]
    bk = str(tmp_path / "s.breakpoints")
    with open(bk, "w") as f:
        for i in range(S.n_sites):
            l, r, _ = S.site(i)
            if i % 6 == 4:
                r = _rc(l)  # nothing in between: no solution forward, reverse attempt
            if i % 9 == 7:
                l = l[:30] + ("A" if l[30] != "A" else "C")  # a source that is not in the graph: no solution either way
            nm = names[i % len(names)] % (i, 1000 + 37 * i)
            f.write(">%s left_kmer\n%s\n>%s right_kmer\n%s\n" % (nm, l, nm, r))
    outs = {}
    for mode, env in (("device", None), ("host", "1")):
        with _env("MTG_HOST_FORMAT", env), _env("MTG_CLI_BATCH", "4"):
            assert emu_product.Filler().run(["-graph", idxf, "-bkpt", bk, "-out", str(tmp_path / mode), "-extend", "-filter"]) == 0
        outs[mode] = {e: open(str(tmp_path / mode) + e, "rb").read() for e in (".insertions.fasta", ".info.txt", ".extensions.fasta")}
        outs[mode][".vcf"] = b"\n".join(l for l in open(str(tmp_path / mode) + ".insertions.vcf", "rb").read().split(b"\n") if not l.startswith(b"##"))
    for e in outs["host"]:
        assert outs["device"][e] == outs["host"][e], e
    assert outs["host"][".insertions.fasta"].count(b">") >= 20 and len(outs["host"][".vcf"]) > 2000


@pytest.mark.parametrize("lanes,rounds", [(4, "0"), (4, "2"), (16, "0")])
def test_group_form_of_the_bubble_code_on_lanes_in_lock_step(lanes, rounds):
    """round 4: the cooperative code where it is cooperative.  The emulation build with -DMTG_EMU_LANES=N runs the group form of
    explore_branching (mtg_bubble.h: frontline expansion with ballot compaction, chunk-wise de-duplication in lane order, the four
    successors of a node by four lanes, LDS sets filled by compare-and-swap, the enumeration's frames, the marks' plan) on N lanes, one
    host thread each, meeting at every collective (ballot / shuffle / minimum / sync), instead of one lane that plays all of them.  Every
    answer is compared with the one-lane general code (0xBADC), the lanes must agree on every uniform result (0xBADD), and the contigs
    must be the oracle's -- on the random graphs of the fuzz (bubbles, tips, repeats, loops) and the variant pairs whose bubbles the group
    form resolves.  rounds = 0: every parked gap goes to the finishing kernel's group form (large LDS areas); 2: bubble kernel (small
    areas) + resumed walks first.  (The library with the lanes is a build of its own; tests/emu_lib.load(lanes, tsan=True) builds it under
    ThreadSanitizer.)"""
    with _env("MTG_ROUNDS", rounds):
        before = emu_lib.coop_counts(lanes)
        k = 21
        for seed in range(6 if lanes > 4 else 14):
            rng, g, seqs = _make_case(100 + seed, k)
            idx = oracle_lib.Index.from_sequences(seqs, k, 1, 40)
            km, ct = idx.export()
            emu = emu_lib.EmuIndex(km, ct, k, lanes=lanes)
            for _ in range(6):
                p = rng.randrange(0, len(g) - k)
                s_ = g[p:p + k] if rng.random() < 0.5 else _rc(g[p:p + k])
                tp = rng.randrange(0, len(g) - k)
                t = g[tp:tp + k]
                er = rng.choice([0, 0, 1])
                oc, _ = idx.stage_a(s_, t, oracle_lib.default_params(end_rule_nonbranching=er))
                ec, st, _, _ = emu.stage_a(s_, t, 100, 10000, er)
                assert st == 0 and ec == oc, (lanes, rounds, seed, s_, t, er, hex(st))
            idx.close()
            emu.close()
        after = emu_lib.coop_counts(lanes)
        assert after[0] - before[0] >= 5 and after[1] - before[1] >= 1, (before, after)  # the lanes did answer bubbles: consensuses and rejections


@pytest.mark.parametrize("k", [31, 21, 13])
def test_tip_fast_path_against_the_general_code(k):
    """round 4: a walk that meets a short dead-end branch (what a sequencing error near the end of a read leaves) answers it on the spot
    (mtg_traverse.h: tip_fast) instead of parking the gap.  The emulation runs the general explore_branching next to every answer (0xBADE:
    length, consensus, the one mark) and the contigs must be the oracle's.  Tips of 1 .. k+2 nodes on either branch (the k+1-node tip is the
    longest the reference pops, the next one ends the contig), on both strands, a tip whose end is reached twice, two tips at one node."""
    rng = random.Random(77 + k)
    before = emu_lib.coop_counts()[3]
    for rep in range(12):
        g = _rand_seq(rng, rng.randrange(600, 1500))
        seqs = [g]
        spots = sorted(rng.sample(range(60, len(g) - 80), 5))
        for j, p in enumerate(spots):
            L = [1, 2, k - 1, k, k + 1, k + 2, 3, 7][(rep + j) % 8]
            first = rng.choice([c for c in "ACGT" if c != g[p]])  # leaves the genome after g[p-k:p], never comes back
            tip = g[p - k:p] + first + _rand_seq(rng, L - 1)
            seqs.append(tip if rng.random() < 0.5 else _rc(tip))
            if rep % 4 == 3 and j == 0:  # a second tip at the same node
                other = rng.choice([c for c in "ACGT" if c not in (g[p], first)])
                seqs.append(g[p - k:p] + other + _rand_seq(rng, 2))
        idx = oracle_lib.Index.from_sequences(seqs, k, 1, 40)
        km, ct = idx.export()
        emu = emu_lib.EmuIndex(km, ct, k)
        for s_, t in ((g[:k], g[-k:]), (_rc(g[-k:]), _rc(g[:k]))):
            for er in (0, 1):
                oc, _ = idx.stage_a(s_, t, oracle_lib.default_params(end_rule_nonbranching=er))
                ec, st, _, _ = emu.stage_a(s_, t, 100, 10000, er)
                assert st == 0 and ec == oc, (k, rep, hex(st), seqs)
        idx.close()
        emu.close()
    assert emu_lib.coop_counts()[3] - before >= 20  # the fast path did answer tips


@pytest.mark.parametrize("k", [31, 21, 13])
def test_indel_bubbles_answered_by_the_walking_lane(k):
    """round 4: two unitig branches of different lengths onto one node (a heterozygous insertion / deletion) are answered where the walk meets
    them (mtg_traverse.h: indel_bulk; merge_fast for the nodes that run into a junction): a length difference of 1 or 2 gives two consensuses and the more abundant one is taken, 3 and more is
    the reference's "no consensus" (the depth allowance of its path enumeration) and the contig ends without a park.  The emulation runs the
    general explore_branching next to every answer (0xBADF: length, consensus, marks) and the contigs must be the oracle's.  Deletions of
    1 .. 5 nucleotides in one allele, on both strands, next to each other and next to SNPs, with abundances that favour either allele."""
    rng = random.Random(1234 + k)
    before, merges_before, refused_before = emu_lib.coop_counts()[4], emu_lib.coop_counts()[5], emu_lib.coop_counts()[6]
    for rep in range(14):
        g = _rand_seq(rng, rng.randrange(700, 1600))
        h2 = list(g)
        cuts = sorted(rng.sample(range(80, len(g) - 120), 5), reverse=True)
        for j, p in enumerate(cuts):
            dl = [1, 2, 3, 4, 5, 1, 2, 2][(rep + j) % 8]
            del h2[p:p + dl]
            if (rep + j) % 5 == 0:  # a SNP close by: the branches are no single unitigs any more, the general code decides
                q = p - rng.randrange(2, k + 3)
                h2[q] = rng.choice([c for c in "ACGT" if c != h2[q]])
        h2 = "".join(h2)
        seqs = [g, h2] + ([g] if rep % 3 == 0 else []) + ([h2, h2] if rep % 3 == 1 else [])  # abundances: equal, first allele heavier, second heavier
        idx = oracle_lib.Index.from_sequences(seqs, k, 1, 0) if rep % 2 else oracle_lib.Index.from_sequences(seqs, k, 1, 40)
        km, ct = idx.export()
        emu = emu_lib.EmuIndex(km, ct, k)
        for s_, t in ((g[:k], g[-k:]), (_rc(g[-k:]), _rc(g[:k])), (h2[:k], h2[-k:])):
            oc, _ = idx.stage_a(s_, t, oracle_lib.default_params())
            ec, st, _, _ = emu.stage_a(s_, t, 100, 10000, 0)
            assert st == 0 and ec == oc, (k, rep, hex(st), seqs)
        idx.close()
        emu.close()
    assert emu_lib.coop_counts()[4] - before >= 30  # the bulk form did answer such bubbles
    # the contigs that start on the alleles of a refused bubble run into the node behind it, which has two predecessors: the general code's one-nucleotide
    # answer, given by the walking lane as well (merge_fast, cross-checked the same way: 0xBAE0)
    assert emu_lib.coop_counts()[5] - merges_before >= 10
    # ... and the second contig to arrive at that node finds it marked: "no consensus" said on the spot (the general code run next to it agrees, 0xBAE0)
    assert emu_lib.coop_counts()[6] - refused_before >= 5


@pytest.mark.parametrize("nb,key_bits", [(1_000_000_007, 60), (1 << 30, 60), ((1 << 30) + 1, 60), (3_000_000_019, 60), (1 << 11, 60),
                                          (40_000_003, 40), (1 << 36, 60), (977, 24)])
def test_bucket_arithmetic_of_the_scan_at_real_table_sizes(nb, key_bits):
    """bucket_first_h (mtg_build.h) estimates in double and corrects without a 64-bit division; the emulator's copy aborts when it differs from
    the 128-bit quotient.  The emulated builds use small tables, this sweeps the bucket counts of a 3.1 G-k-mer table."""
    lib = emu_lib.load()
    lib.emu_bucket_first_sweep.restype = C.c_uint64
    lib.emu_bucket_first_sweep.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64]
    lib.emu_bucket_first_sweep(nb, key_bits, 2_000_000, 12345)


def _contig_several_targets_case(mtg_mod, tmp_path, monkeypatch, nloci, counts=None):
    """contig mode with seeds whose local graph reaches SEVERAL other contigs (two or three haplotypes that continue into different contigs): the
    reference groups the paths by target name in an unordered_map and reports the groups in libstdc++'s iteration order (src/Filler.cpp:924-936).
    The device function answers every group (mtg_general.h: header records), the host orders them with the same map (mtg_host.cpp: run_general).
    Names of many shapes, so that the map's order differs from the insertion order somewhere.  Files against the oracle's, with the device
    function and with the host's path alone (HOST_GENERAL)."""
    rng = random.Random(4242)
    seqs, contigs = [], []
    names = []
    for i in range(nloci):
        X, Y, Z = _rand_seq(rng, 500), _rand_seq(rng, 500), _rand_seq(rng, 500)
        a, b = _rand_seq(rng, rng.randrange(60, 300)), _rand_seq(rng, rng.randrange(60, 300))
        if i % 4 == 3:  # a third continuation: three groups
            V, c = _rand_seq(rng, 500), _rand_seq(rng, rng.randrange(60, 300))
            seqs.append(X + c + V)
            contigs.append(V[80:])
        seqs += [X + a + Y, X + b + Z]
        contigs += [X[:420], Y[80:], Z[80:]]
    for i in range(len(contigs)):
        names.append(rng.choice(["ctg%d", "c%d", "scaffold_%d_len", "NODE_%d_length_100_cov_3.5", "x%dy"]) % rng.randrange(1, 10 ** rng.randrange(1, 7)) + "_%d" % i)
    o = oracle_lib.Index.from_sequences(seqs, 31, 3, 40)
    km, ct = o.export()
    cf = str(tmp_path / "c.fa")
    with open(cf, "w") as f:
        for n, c in zip(names, contigs):
            f.write(">%s\n%s\n" % (n, c))
    _write_idx(str(tmp_path / "m.mtgidx"), km, ct)
    o.fill_files("contig", cf, str(tmp_path / "cpu"))
    before = counts() if counts else None
    assert mtg_mod.Filler().run(["-graph", str(tmp_path / "m.mtgidx"), "-contig", cf, "-out", str(tmp_path / "hip")]) == 0
    after = counts() if counts else None
    for ext in (".insertions.fasta", ".info.txt", ".gfa", "_seed_dictionary.fasta"):
        assert _read(str(tmp_path / "hip") + ext) == _read(str(tmp_path / "cpu") + ext), ext
    assert _read(str(tmp_path / "hip.insertions.fasta")).count(">") >= 4 * nloci  # both continuations from the X contig's end, and each back from the other side
    monkeypatch.setenv("MTG_HOST_GENERAL", "1")
    assert mtg_mod.Filler().run(["-graph", str(tmp_path / "m.mtgidx"), "-contig", cf, "-out", str(tmp_path / "host")]) == 0
    monkeypatch.delenv("MTG_HOST_GENERAL")
    for ext in (".insertions.fasta", ".info.txt", ".gfa"):
        assert _read(str(tmp_path / "host") + ext) == _read(str(tmp_path / "cpu") + ext), ext
    o.close()
    return before, after


def test_contig_mode_gaps_that_reach_several_targets_by_the_device_function(emu_product, tmp_path, monkeypatch):
    """the emulation build runs the host's whole path next to every gap the device function finishes (HostChunk::gen_check) and counts the gaps with
    several groups it finished / the multi-target gaps it left to the host"""
    before, after = _contig_several_targets_case(emu_product, tmp_path, monkeypatch, 14, emu_lib.gen_counts)
    assert after[0] - before[0] >= 10, (before, after)   # the seeds at the ends of the X contigs, at least
    assert after[1] - before[1] == 0, (before, after)    # none of this set's multi-target gaps needs the host


def _several_targets_batch_case(mtg, monkeypatch):
    if os.environ.get("MTG_HOST_GENERAL") or os.environ.get("MTG_HOST_PATHS") or any(w in os.environ.get("MTG_TUNING", "") for w in ("HOST_GENERAL", "HOST_PATHS")):
        pytest.skip("the sweep forces the host's path: nothing for the device function to finish")
    """the same shape through the batch entry, where the statistics of the call say who finished the multi-contig gaps: the device (k_general with
    groups), none left to the host; the records equal those of the host's path alone (which the file tests pin against the oracle)"""
    rng = random.Random(515)
    seqs, gaps = [], []
    for i in range(40):
        X, Y, Z = _rand_seq(rng, 300), _rand_seq(rng, 300), _rand_seq(rng, 300)
        a, b = _rand_seq(rng, rng.randrange(60, 300)), _rand_seq(rng, rng.randrange(60, 300))
        seqs += [X + a + Y, X + b + Z]
        tg = [(Y[40:71], "n%d_%d" % (rng.randrange(10 ** 6), i), False), (Z[40:71], "q%d" % rng.randrange(10 ** 4), True), (_rand_seq(rng, 31), "unreached", False)]
        rng.shuffle(tg)
        gaps.append(mtg.Gap(X[200:231], "".join(t[0] for t in tg), tg))
    o = oracle_lib.Index.from_sequences(seqs, 31, 3, 40)
    km, ct = o.export()
    o.close()
    idx = mtg.Index.from_kmers(km, ct, 31)
    res = idx.fill_batch(gaps)
    st = mtg.last_batch_stats()
    assert st["n_general_device"] >= 40 and st["n_general_host"] == 0, st
    assert all(len(r["filled"]) == 2 for r in res)
    monkeypatch.setenv("MTG_HOST_GENERAL", "1")
    res_h = idx.fill_batch(gaps)
    st_h = mtg.last_batch_stats()
    assert st_h["n_general_device"] == 0 and st_h["n_general_host"] >= 40, st_h
    assert res_h == res
    idx.close()


def _wide_dictionary_case(mtg):
    """gaps whose dictionary has more than a thousand targets (the marshaller deals such a dictionary to the pool target by target instead of gap by
    gap, round 6) next to ordinary ones: the reached target is found wherever it stands in the dictionary -- first, last, in the middle, within two
    differences, lowercase -- and the records equal those of the same gaps with the unreached targets left out (indices mapped)"""
    rng = random.Random(99)
    seqs, gaps, small = [], [], []
    for i in range(12):
        X, Y = _rand_seq(rng, 300), _rand_seq(rng, 300)
        a = _rand_seq(rng, rng.randrange(60, 300))
        seqs.append(X + a + Y)
        true = Y[40:71]
        if i % 4 == 1:
            true = true[:7] + ("A" if true[7] != "A" else "C") + true[8:]
        if i % 4 == 2:
            true = true.lower()
        n = 0 if i % 3 == 2 else 1500 + 37 * i
        decoys = [(_rand_seq(rng, 31), "d%d_%d" % (i, j), bool(j & 1)) for j in range(n)]
        at = {0: 0, 1: n, 2: n // 2}[i % 3] if n else 0
        tg = decoys[:at] + [(true, "hit%d" % i, False)] + decoys[at:]
        gaps.append(mtg.Gap(X[200:231], "".join(t[0] for t in tg), tg))
        small.append((mtg.Gap(X[200:231], "".join(t[0] for t in tg), [(true, "hit%d" % i, False)]), at))
    o = oracle_lib.Index.from_sequences(seqs, 31, 3, 40)
    km, ct = o.export()
    o.close()
    idx = mtg.Index.from_kmers(km, ct, 31)
    res = idx.fill_batch(gaps)
    ref = idx.fill_batch([g for g, _ in small])
    assert all(len(r["filled"]) == 1 for r in res)
    for r, q, (_, at) in zip(res, ref, small):
        assert [dict(f, target_index=0) for f in r["filled"]] == [dict(f, target_index=0) for f in q["filled"]]
        assert r["filled"][0]["target_index"] == at
    idx.close()


def test_dictionaries_of_more_than_a_thousand_targets_on_emulator(emu_product):
    _wide_dictionary_case(emu_product)


def test_gaps_with_several_reached_targets_stay_on_the_device_function(emu_product, monkeypatch):
    _several_targets_batch_case(emu_product, monkeypatch)


def test_light_walk_form_on_data_with_bubbles(emu_product, tmp_path, monkeypatch):
    """round 5: the light walk kernel (WALK_SIMPLE: simple paths only, the gap parks at its first branching node whatever its shape) is what a launch
    starts with when the previous one hardly met a branching node -- and a batch of another kind may follow: every bubble case must come out the
    same when the first walk is the light one (LIGHT_WALK=1 forces it; the rounds and the finishing form take the parked gaps from there)."""
    monkeypatch.setenv("MTG_LIGHT_WALK", "1")
    _diploid_case(emu_product, tmp_path, 10)
    (tmp_path / "al").mkdir()
    _allelic_inserts_case(emu_product, tmp_path / "al", 8)
    (tmp_path / "ct").mkdir()
    _contig_several_targets_case(emu_product, tmp_path / "ct", monkeypatch, 6)


def _duplicate_target_names_case(mtg):
    """two REACHED targets of a gap under one name (and orientation): the reference keeps their paths in one group of its map (src/Filler.cpp:924-936), the
    device function answers two -- the host cannot take such a gap from the device's answer and the batch is run again with the host's path for its
    multi-contig gaps (advisor r5: a launch whose gaps the device all finished has no contigs on the host).  Records == those of the host's path alone."""
    rng = random.Random(616)
    seqs, gaps = [], []
    for i in range(24):
        X, Y, Z = _rand_seq(rng, 300), _rand_seq(rng, 300), _rand_seq(rng, 300)
        a, b = _rand_seq(rng, rng.randrange(60, 300)), _rand_seq(rng, rng.randrange(60, 300))
        seqs += [X + a + Y, X + b + Z]
        dup = i % 3 != 2
        tg = [(Y[40:71], "twin_%d" % i, False), (Z[40:71], "twin_%d" % i if dup else "other_%d" % i, False), (_rand_seq(rng, 31), "unreached", True)]
        gaps.append(mtg.Gap(X[200:231], "".join(t[0] for t in tg), tg))
    o = oracle_lib.Index.from_sequences(seqs, 31, 3, 40)
    km, ct = o.export()
    o.close()
    idx = mtg.Index.from_kmers(km, ct, 31)
    res = idx.fill_batch(gaps)
    mtg.tuning_set("HOST_GENERAL", "1")
    try:
        want = idx.fill_batch(gaps)
    finally:
        mtg.tuning_set("HOST_GENERAL", None)
    idx.close()
    assert res == want
    assert sum(1 for r in res if len(r["filled"]) >= 2) >= 20


def test_two_reached_targets_under_one_name(emu_product):
    _duplicate_target_names_case(emu_product)
