"""Parity tests proper: the HIP path (through the C ABI of include/mtg_fill.h) against the CPU oracle, the committed
golden files and size-independent properties.  Bit-exact everywhere (integer / byte / text work)."""
import os
import random
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read(p):
    with open(p) as f:
        return f.read()


def _vcf_body(p):
    return [l for l in _read(p).splitlines() if not l.startswith("##")]


def _rc(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def _fasta(path):
    recs = []
    for l in open(path):
        l = l.rstrip("\n")
        if l.startswith(">"):
            recs.append([l[1:], ""])
        elif recs:
            recs[-1][1] += l
    return recs


@pytest.fixture(scope="module")
def mtg():
    import torch
    torch.cuda.init()  # torch bundles its own HIP runtime: initialise it before libmtgfill.so touches the device
    import mindthegap_amd
    mindthegap_amd.load_library()
    assert mindthegap_amd.device_count() >= 1, "these tests need a HIP device"
    return mindthegap_amd


@pytest.fixture(scope="module")
def full_idx(mtg, golden_dir):
    from tests import oracle_lib
    d = os.path.join(golden_dir, "data")
    o = oracle_lib.Index.from_files([os.path.join(d, "reads_r1.fastq"), os.path.join(d, "reads_r2.fastq")], 31, 7)
    km, ct = o.export()
    return o, mtg.Index.from_kmers(km, ct, 31), km, ct


@pytest.fixture(scope="module")
def ctg_idx(mtg, golden_dir):
    from tests import oracle_lib
    o = oracle_lib.Index.from_files([os.path.join(golden_dir, "data", "contig-reads.fasta.gz")], 31, 3)
    km, ct = o.export()
    return o, mtg.Index.from_kmers(km, ct, 31), km, ct


def test_index_queries_match_oracle(full_idx):
    o, g, km, ct = full_idx
    info = g.info()
    assert (info["nb_solid_kmers"], info["nb_branching"]) == (7419, 36)  # test/full_test/gold_fill.output:13-14
    rng = np.random.default_rng(3)
    M = np.uint64((1 << 62) - 1)
    succ = ((km[:, None] << np.uint64(2)) & M) | np.arange(4, dtype=np.uint64)[None, :]
    pred = (km[:, None] >> np.uint64(2)) | (np.arange(4, dtype=np.uint64)[None, :] << np.uint64(60))
    q = np.concatenate([km, succ.reshape(-1), pred.reshape(-1), rng.integers(0, 1 << 62, 5000, dtype=np.uint64)])
    assert (g.contains(q) == o.contains(q)).all()
    assert (g.abundance(q) == o.abundance(q)).all()
    s, p = g.neighbors(km)
    es = (o.contains(succ.reshape(-1)).reshape(-1, 4) * (1 << np.arange(4))).sum(1)
    ep = (o.contains(pred.reshape(-1)).reshape(-1, 4) * (1 << np.arange(4))).sum(1)
    assert (s == es).all() and (p == ep).all()


def test_stage_a_matches_oracle_on_goldens(full_idx, ctg_idx, golden_dir):
    o, g, _, _ = full_idx
    recs = _fasta(os.path.join(golden_dir, "full_test", "gold.breakpoints"))
    src, tgt = [], []
    for i in range(0, len(recs), 2):
        src += [recs[i][1], _rc(recs[i + 1][1])]
        tgt += [recs[i + 1][1], _rc(recs[i][1])]
    got = g.stage_a(src, tgt)
    for s, t, c in zip(src, tgt, got):
        assert c == o.stage_a(s, t)[0]
    o2, g2, _, _ = ctg_idx
    seeds = _fasta(os.path.join(golden_dir, "contig_test", "gold_seed_dictionary.fasta"))
    src = [s for _, s in seeds]
    tgt = ["ACGT" * 20] * len(src)
    got = g2.stage_a(src, tgt)
    assert max(len(c) for c in got) == 22  # multi-contig gaps with tips are exercised
    for s, t, c in zip(src, tgt, got):
        assert c == o2.stage_a(s, t)[0]


def test_cli_bkpt_mode_equals_golden_and_oracle(mtg, full_idx, golden_dir, tmp_path):
    d = os.path.join(golden_dir, "data")
    bk = os.path.join(golden_dir, "full_test", "gold.breakpoints")
    rc = mtg.Filler().run(["-in", os.path.join(d, "reads_r1.fastq") + "," + os.path.join(d, "reads_r2.fastq"), "-bkpt", bk, "-out", str(tmp_path / "hip")])
    assert rc == 0
    assert _read(tmp_path / "hip.insertions.fasta") == _read(os.path.join(golden_dir, "full_test", "gold.insertions.fasta"))
    assert _vcf_body(tmp_path / "hip.insertions.vcf") == _vcf_body(os.path.join(golden_dir, "full_test", "gold.insertions.vcf"))
    o = full_idx[0]
    o.fill_files("bkpt", bk, str(tmp_path / "cpu"))
    assert _read(tmp_path / "hip.info.txt") == _read(tmp_path / "cpu.info.txt")
    # -graph round trip through the index container written by -in
    rc = mtg.Filler().run(["-graph", str(tmp_path / "hip.mtgidx"), "-bkpt", bk, "-out", str(tmp_path / "hip2"), "-extend"])
    assert rc == 0
    assert _read(tmp_path / "hip2.insertions.fasta") == _read(tmp_path / "hip.insertions.fasta")


def test_cli_contig_mode_equals_golden_and_oracle(mtg, ctg_idx, golden_dir, tmp_path):
    d = os.path.join(golden_dir, "data")
    g = os.path.join(golden_dir, "contig_test")
    rc = mtg.Filler().run(["-in", os.path.join(d, "contig-reads.fasta.gz"), "-contig", os.path.join(d, "contigs.fasta"), "-abundance-min", "3", "-out", str(tmp_path / "hip")])
    assert rc == 0
    assert _read(tmp_path / "hip.gfa") == _read(os.path.join(g, "gold.gfa"))
    assert _read(tmp_path / "hip.insertions.fasta") == _read(os.path.join(g, "gold.insertions.fasta"))
    assert _read(tmp_path / "hip_seed_dictionary.fasta") == _read(os.path.join(g, "gold_seed_dictionary.fasta"))
    ctg_idx[0].fill_files("contig", os.path.join(d, "contigs.fasta"), str(tmp_path / "cpu"))
    assert _read(tmp_path / "hip.info.txt") == _read(tmp_path / "cpu.info.txt")


def test_cli_errors(mtg, tmp_path):
    assert mtg.Filler().run(["-bkpt", "x"]) == 1  # -graph xor -in, src/Filler.cpp:140-145
    assert mtg.Filler().run(["-in", "a", "-graph", "b", "-bkpt", "x"]) == 1
    assert mtg.Filler().run(["-in", "a"]) == 1  # -bkpt xor -contig, src/Filler.cpp:147-150
    assert mtg.Filler().run(["-in", "/nonexistent.fq", "-bkpt", "x", "-out", str(tmp_path / "e")]) == 1


def test_synthetic_sites_packed_device_index(mtg, tmp_path):
    """config-2 shaped set at reduced size: device-built index from packed sequences, every site must be filled with exactly
    its inserted sequence, and files must equal the oracle's byte for byte."""
    import torch
    from mindthegap_amd.synth import SynthSet
    from tests import oracle_lib
    S = SynthSet(nseq=300, n_sites=200, seed=7)
    w = torch.from_numpy(S.words.view(np.int64)).cuda()
    wo = torch.from_numpy(S.word_off.view(np.int64)).cuda()
    ln = torch.from_numpy(S.lens.view(np.int32)).cuda()
    g = mtg.Index.from_packed_device(w.data_ptr(), wo.data_ptr(), ln.data_ptr(), S.nseq, S.total_kmers_upper_bound, 31, 3, 40)
    seqs = [S.ascii(j) for j in range(S.nseq)]
    o = oracle_lib.Index.from_sequences(seqs, 31, 3, 40)
    assert g.info()["nb_solid_kmers"] == len(o)
    gaps = []
    for i in range(S.n_sites):
        l, r, _ = S.site(i)
        gaps.append(mtg.Gap(l, r, [(r, S.site_name(i), False)]))
    res = g.fill_batch(gaps)
    for i, r in enumerate(res):
        assert len(r["filled"]) == 1 and r["filled"][0]["seq"] == S.site(i)[2] and r["filled"][0]["qual"] == 50
    bk = str(tmp_path / "syn.breakpoints")
    S.write_breakpoints(bk)
    idxf = str(tmp_path / "syn.mtgidx")
    km, ct = o.export()
    g2 = mtg.Index.from_kmers(km, ct, 31)
    # same graph through the k-mer list path and the CLI
    import ctypes as C
    g2.lib.mtg_index_save.argtypes  # noqa
    o.fill_files("bkpt", bk, str(tmp_path / "cpu"))
    # write an index container for the CLI
    import struct
    with open(idxf, "wb") as f:
        f.write(b"MTGIDX1\0" + struct.pack("<4i", 31, 3, -1, 0) + struct.pack("<Q", len(km)) + km.tobytes() + np.minimum(ct, 255).astype(np.uint32).tobytes())
    assert mtg.Filler().run(["-graph", idxf, "-bkpt", bk, "-out", str(tmp_path / "hip")]) == 0
    for ext in (".insertions.fasta", ".info.txt"):
        assert _read(str(tmp_path / "hip") + ext) == _read(str(tmp_path / "cpu") + ext)
    assert _vcf_body(str(tmp_path / "hip.insertions.vcf")) == _vcf_body(str(tmp_path / "cpu.insertions.vcf"))
    g.close(); g2.close(); o.close()


def test_reverse_attempt_and_unfillable(mtg, tmp_path):
    """sites whose forward source k-mer is absent from the graph are rescued by the reverse attempt (src/Filler.cpp:669-680);
    sites with neither anchor produce no sequence; both must match the oracle byte for byte."""
    from mindthegap_amd.synth import SynthSet
    from tests import oracle_lib
    S = SynthSet(nseq=40, n_sites=30, seed=11)
    seqs = [S.ascii(j) for j in range(S.nseq)]
    o = oracle_lib.Index.from_sequences(seqs, 31, 3, 40)
    km, ct = o.export()
    g = mtg.Index.from_kmers(km, ct, 31)
    bk = str(tmp_path / "rev.breakpoints")
    with open(bk, "w") as f:
        for i in range(S.n_sites):
            l, r, _ = S.site(i)
            if i % 3 == 0:  # break the left anchor's first base: forward start is not solid but may still extend
                l = ("A" if l[0] != "A" else "C") + l[1:]
            if i % 3 == 1:  # break the left anchor in the middle: forward fails, reverse succeeds
                l = l[:15] + ("A" if l[15] != "A" else "C") + l[16:]
            if i % 7 == 3:  # mismatch inside the right anchor (within nb_mis_allowed)
                r = r[:10] + ("A" if r[10] != "A" else "C") + r[11:]
            f.write(">%s left_kmer\n%s\n>%s right_kmer\n%s\n" % (S.site_name(i), l, S.site_name(i), r))
    import struct
    idxf = str(tmp_path / "rev.mtgidx")
    with open(idxf, "wb") as fh:
        fh.write(b"MTGIDX1\0" + struct.pack("<4i", 31, 3, -1, 0) + struct.pack("<Q", len(km)) + km.tobytes() + ct.astype(np.uint32).tobytes())
    o.fill_files("bkpt", bk, str(tmp_path / "cpu"), params=oracle_lib.default_params(extend=1))
    assert mtg.Filler().run(["-graph", idxf, "-bkpt", bk, "-out", str(tmp_path / "hip"), "-extend"]) == 0
    for ext in (".insertions.fasta", ".info.txt", ".extensions.fasta"):
        assert _read(str(tmp_path / "hip") + ext) == _read(str(tmp_path / "cpu") + ext), ext
    assert _vcf_body(str(tmp_path / "hip.insertions.vcf")) == _vcf_body(str(tmp_path / "cpu.insertions.vcf"))
    assert ">" in _read(str(tmp_path / "hip.insertions.fasta"))
    g.close(); o.close()


@pytest.mark.parametrize("host_format", [False, True])
def test_short_fills_every_alignment_both_attempts(mtg, tmp_path, monkeypatch, host_format):
    """round 4: the lean forms of the post-processing and of the result kernel (k_post_lean, k_emit_lean: eight lanes per gap) read the
    abundance bytes of a fill as aligned 8-byte words and write its ASCII as aligned 16-byte pieces -- fills of 1 .. 70 nucleotides put
    the byte range and the pieces at every alignment, with ragged heads and tails shorter than a word; a third of the sites are filled
    by the reverse attempt (stretches read backwards and complemented).  Sum and median of the abundances are in the FASTA headers:
    the files must be the oracle's, through the device's formatter and through the host's writers (records and sequences on the host)."""
    from mindthegap_amd.synth import SynthSet
    from tests import oracle_lib
    if host_format:
        monkeypatch.setenv("MTG_HOST_FORMAT", "1")
    S = SynthSet(nseq=700, n_sites=600, seed=23, ins_min=1, ins_max=70)
    seqs = [S.ascii(j) for j in range(S.nseq)]
    o = oracle_lib.Index.from_sequences(seqs, 31, 3, 40)
    km, ct = o.export()
    bk = str(tmp_path / "short.breakpoints")
    with open(bk, "w") as f:
        for i in range(S.n_sites):
            l, r, _ = S.site(i)
            if i % 3 == 1:  # break the left anchor in the middle: forward fails, the reverse attempt fills
                l = l[:15] + ("A" if l[15] != "A" else "C") + l[16:]
            f.write(">%s left_kmer\n%s\n>%s right_kmer\n%s\n" % (S.site_name(i), l, S.site_name(i), r))
    import struct
    idxf = str(tmp_path / "short.mtgidx")
    with open(idxf, "wb") as fh:
        fh.write(b"MTGIDX1\0" + struct.pack("<4i", 31, 3, -1, 0) + struct.pack("<Q", len(km)) + km.tobytes() + ct.astype(np.uint32).tobytes())
    o.fill_files("bkpt", bk, str(tmp_path / "cpu"))
    assert mtg.Filler().run(["-graph", idxf, "-bkpt", bk, "-out", str(tmp_path / "hip")]) == 0
    for ext in (".insertions.fasta", ".info.txt"):
        assert _read(str(tmp_path / "hip") + ext) == _read(str(tmp_path / "cpu") + ext), ext
    assert _vcf_body(str(tmp_path / "hip.insertions.vcf")) == _vcf_body(str(tmp_path / "cpu.insertions.vcf"))
    assert _read(str(tmp_path / "hip.insertions.fasta")).count(">") >= 550
    o.close()


@pytest.mark.parametrize("err", [0.0, 0.002])
def test_simulated_reads_cfg2_shape(mtg, tmp_path, err, monkeypatch):
    """BASELINE config 2 at reduced size: 30x simulated reads (E0 error-free / E1-like with substitutions), index built by -in with
    -abundance-min 3, CLI outputs byte-identical to the oracle."""
    from mindthegap_amd.synth import SynthSet, simulate_reads
    from tests import oracle_lib
    S = SynthSet(nseq=60, n_sites=50, seed=21)
    reads = str(tmp_path / "reads.fa")
    simulate_reads(S, reads, coverage=30, error_rate=err, seed=5)
    bk = str(tmp_path / "s.breakpoints")
    S.write_breakpoints(bk)
    o = oracle_lib.Index.from_files([reads], 31, 3)
    o.fill_files("bkpt", bk, str(tmp_path / "cpu"))
    if err > 0:
        monkeypatch.setenv("MTG_COUNT_PASSES", "4")  # exercise the multi-pass k-mer counting on the erroneous set
    assert mtg.Filler().run(["-in", reads, "-bkpt", bk, "-abundance-min", "3", "-out", str(tmp_path / "hip")]) == 0
    for ext in (".insertions.fasta", ".info.txt"):
        assert _read(str(tmp_path / "hip") + ext) == _read(str(tmp_path / "cpu") + ext), ext
    assert _vcf_body(str(tmp_path / "hip.insertions.vcf")) == _vcf_body(str(tmp_path / "cpu.insertions.vcf"))
    if err == 0.0:
        seqs = [l for l in _read(str(tmp_path / "hip.insertions.fasta")).splitlines() if not l.startswith(">")]
        assert seqs == [S.site(i)[2] for i in range(S.n_sites)]
    o.close()


def test_sequence_scan_bloom_kernel(mtg):
    """k_scan (rolling k-mer + LDS-staged minimizer-blocked Bloom + exact confirmation) against the oracle's Graph::contains"""
    from tests.test_emu_parity import _scan_case
    _scan_case(mtg)


def test_contig_mode_many_targets(mtg, tmp_path):
    """400 contigs / 800 seeds / ~800 anchors per seed: exercises the multi-target terminal search of k_post"""
    from tests.test_emu_parity import _contig_gap_case
    _contig_gap_case(mtg, tmp_path, 200)


def test_contig_mode_inexact_targets_among_many(mtg, tmp_path, monkeypatch):
    """round 6: the terminal search through the piece index of the batch's dictionaries (k_post_index; a contig position's nb_mis + 1 pieces looked up
    instead of a count against each of ~480 targets): targets that differ from the graph by one, two (in different pieces, in one piece), three
    substitutions, an N, lowercase -- files equal to the oracle's, with the index and (NO_POST_INDEX) with the pass over every target"""
    from tests.test_emu_parity import _contig_gap_case
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    _contig_gap_case(mtg, tmp_path / "a", 120, mutate=True)
    monkeypatch.setenv("MTG_NO_POST_INDEX", "1")
    _contig_gap_case(mtg, tmp_path / "b", 120, mutate=True)


def test_dictionaries_of_more_than_a_thousand_targets(mtg):
    from tests.test_emu_parity import _wide_dictionary_case
    _wide_dictionary_case(mtg)


def test_cli_edge_cases(mtg, tmp_path):
    """REPEATED anchors, mismatching / N / lowercase / long anchors, unfillable sites, -fwd-only -filter -extend, -max-nodes / -max-length"""
    from tests.test_emu_parity import _edge_case_run
    _edge_case_run(mtg, tmp_path)


def test_diploid_bubbles(mtg, tmp_path):
    from tests.test_emu_parity import _diploid_case
    _diploid_case(mtg, tmp_path, 200)


@pytest.mark.parametrize("chunk", [37, 100])
def test_device_formatted_text_of_a_batch_of_several_launches(mtg, tmp_path, monkeypatch, chunk):
    """round 4: with the text formatted on the device the ASCII arena of a batch exists only in the workspace; a batch that needs several
    launches (more gaps than the scratch holds; here MAX_CHUNK of the tuning table) grows that arena between its launches -- and lost what
    the earlier launches had written there (records of length 0 in the FASTA file).  The diploid case through the tool, 37 and 100 gaps
    per launch, against the oracle's files."""
    from tests.test_emu_parity import _diploid_case
    monkeypatch.setenv("MTG_MAX_CHUNK", str(chunk))
    _diploid_case(mtg, tmp_path, 200)


def test_allelic_inserts_general_path(mtg, tmp_path):
    """multi-contig gaps: contig graph, reverse DFS, path sequences and k_nw-based de-duplication against the oracle"""
    from tests.test_emu_parity import _allelic_inserts_case
    _allelic_inserts_case(mtg, tmp_path, 120)


def test_contig_mode_gaps_that_reach_several_targets(mtg, tmp_path, monkeypatch):
    """k_general with several groups per gap (contig mode: a seed whose graph reaches two or three other contigs), the host ordering the groups like
    the reference's unordered_map; against the oracle's files, and the host's path alone (HOST_GENERAL) against them too"""
    from tests.test_emu_parity import _contig_several_targets_case
    _contig_several_targets_case(mtg, tmp_path, monkeypatch, 60)


def test_gaps_with_several_reached_targets_stay_on_the_device(mtg, monkeypatch):
    from tests.test_emu_parity import _several_targets_batch_case
    _several_targets_batch_case(mtg, monkeypatch)


def test_scratch_tier_retry_inside_a_batch(mtg):
    """one gap of the batch walks a 150 kb unitig and overflows the tier-0 contig arena: it is re-run in a larger tier on the device while
    its neighbours keep their tier-0 results; contigs identical to the oracle's"""
    import random
    from tests import oracle_lib
    rng = random.Random(5)
    long_seq = "".join(rng.choice("ACGT") for _ in range(150000))
    shorts = ["".join(rng.choice("ACGT") for _ in range(3000)) for _ in range(70)]
    o = oracle_lib.Index.from_sequences([long_seq] + shorts, 31, 3, 40)
    km, ct = o.export()
    g = mtg.Index.from_kmers(km, ct, 31)
    src = [s[:31] for s in shorts[:35]] + [long_seq[:31]] + [s[:31] for s in shorts[35:]]
    tgt = [s[500:531] for s in shorts[:35]] + [long_seq[500:531]] + [s[500:531] for s in shorts[35:]]
    got = g.stage_a(src, tgt)
    assert mtg.last_batch_stats()["n_retried_gaps"] >= 1
    for s, t, c in zip(src, tgt, got):
        assert c == o.stage_a(s, t)[0]
    assert len(got[35]) == 1 and len(got[35][0]) == 150000
    res = g.fill_batch([mtg.Gap(s, t, [(t, "x", False)]) for s, t in zip(src, tgt)])
    assert all(len(r["filled"]) == 1 and len(r["filled"][0]["seq"]) == 469 for r in res)
    g.close(); o.close()


def test_needleman_wunsch_kernel_matches_oracle(mtg):
    """k_nw against the oracle's restatement of src/Utils.cpp:87-189: lengths around the 64-column strips, low-complexity sequences (ties in
    the traceback), unrelated pairs, indels, a 6 kb pair"""
    import random
    from tests import oracle_lib
    olib = oracle_lib.load()
    rng = random.Random(5)
    pairs = []
    for _ in range(300):
        na = rng.choice([1, 2, 3, 31, 63, 64, 65, 127, 128, 129, 200, 500, rng.randrange(1, 700)])
        alpha = rng.choice(["ACGT", "AC", "A", "ACGT", "AAAC"])
        a = "".join(rng.choice(alpha) for _ in range(na))
        kind = rng.randrange(4)
        if kind == 0:
            b = "".join(rng.choice(alpha) for _ in range(rng.randrange(1, 700)))
        else:
            b = list(a)
            for _ in range(rng.randrange(0, 1 + len(b) // 8)):
                p = rng.randrange(len(b) + 1)
                op = rng.randrange(3)
                if op == 0 and p < len(b):
                    b[p] = rng.choice("ACGT")
                elif op == 1:
                    b[p:p] = [rng.choice("ACGT") for _ in range(rng.randrange(1, 20))]
                elif p < len(b):
                    del b[p:p + rng.randrange(1, 20)]
            b = "".join(b) or "A"
        pairs.append((a, b))
    big = "".join(rng.choice("ACGT") for _ in range(6000))
    big2 = big[:1000] + big[1100:3000] + "ACGTACGT" * 20 + big[3000:]
    pairs += [(big, big2), (big2, big), (big, big)]
    got = mtg.nw_matches(pairs)
    for (a, b), m in zip(pairs, got):
        want = olib.mtgo_needleman_wunsch(a.encode(), b.encode())
        assert np.float32(np.float32(m) / np.float32(max(len(a), len(b)))) == np.float32(want), (len(a), len(b), int(m), want)


def test_pipelined_gather_on_rccl_single_rank(mtg):
    """the N > 1 result path of bench.py (page-locked staging, side stream, asynchronous RCCL gather, double buffering) in a world of one
    rank: the only RCCL configuration a one-GPU box offers"""
    import torch
    import torch.distributed as dist
    from mindthegap_amd.shard import PipelinedGather
    import socket
    with socket.socket() as sk:  # a free port for the rendezvous
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        dev = torch.device("cuda", 0)
        pg = PipelinedGather(1 << 20, dst=0, device=dev)
        rng = np.random.default_rng(1)
        last = None
        for step in range(5):
            n = int(rng.integers(1, 1 << 20))
            payload = rng.integers(0, 256, n, dtype=np.uint8)
            pg.buffer()[:n] = payload
            pg.submit(n)
            last = payload
        pg.drain()
        got = pg.last()
        assert len(got) == 1 and got[0].tobytes() == last.tobytes()
        # bench.py's default: steps in flight on caller threads sharing one gather (acquire / submit(j)); every payload ends with its
        # own checksum, so whichever step was submitted last must come back intact
        import threading
        import zlib
        pg2 = PipelinedGather(1 << 20, dst=0, device=dev, depth=4)
        todo, lock, errors = iter(range(24)), threading.Lock(), []

        def worker(seed):
            try:
                torch.cuda.set_device(0)
                r2 = np.random.default_rng(seed)
                while True:
                    with lock:
                        if next(todo, None) is None:
                            return
                    j, buf = pg2.acquire()
                    n2 = int(r2.integers(8, 1 << 20))
                    body = r2.integers(0, 256, n2 - 4, dtype=np.uint8)
                    buf[: n2 - 4] = body
                    buf[n2 - 4: n2] = np.frombuffer(np.uint32(zlib.crc32(body.tobytes())).tobytes(), dtype=np.uint8)
                    pg2.submit(n2, j)
            except BaseException as e:  # noqa: BLE001
                errors.append(repr(e))

        ts = [threading.Thread(target=worker, args=(s,)) for s in (5, 6, 7)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errors, errors
        pg2.drain()
        got = pg2.last()[0]
        assert len(got) >= 8 and zlib.crc32(got[:-4].tobytes()) == int(np.frombuffer(got[-4:].tobytes(), dtype=np.uint32)[0])
        # payloads produced ON the device (bench.py with N > 1: the fill writes its sequences into the gather's device buffer,
        # mtg_fill_prepared_serial_device): only the length goes through the staging buffer
        pg3 = PipelinedGather(1 << 20, dst=0, device=dev, depth=3)
        last = None
        for step in range(6):
            j, _ = pg3.acquire()
            ptr, cap = pg3.device_area(j)
            assert cap == 1 << 20 and ptr % 16 == 0
            n = int(rng.integers(1, 1 << 20))
            payload = torch.from_numpy(rng.integers(0, 256, n, dtype=np.uint8)).to(dev)
            pg3.dbuf[j][pg3.HEADER: pg3.HEADER + n].copy_(payload)  # stands for the result kernel
            torch.cuda.synchronize()
            pg3.submit(n, j, on_device=True, host_copy=True)
            last, last_j = payload.cpu().numpy(), j
        pg3.drain()
        got = pg3.last()
        assert len(got) == 1 and got[0].tobytes() == last.tobytes()
        assert pg3.stage[last_j].numpy()[pg3.HEADER: pg3.HEADER + len(last)].tobytes() == last.tobytes()  # the rank's own host copy
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k", [31, 21, 16, 13])
def test_stage_a_fuzz_on_device(mtg, k):
    """the random graphs of the CPU-side fuzz (repeats, SNP / indel bubbles, tips, loops) and circular genomes through k_stage_a, every
    k-mer size class: 31, 21 (bulk steps), 16 (smallest bulk size), 13 (per-nucleotide steps only); cut-offs and both end rules"""
    import random
    from tests import oracle_lib
    from tests.test_emu_parity import _make_case, _rand_seq
    for seed in range(12):
        rng, g, seqs = _make_case(seed, k)
        if seed % 3 == 2:  # a circular simple path next to the linear genome
            c = _rand_seq(rng, rng.randrange(k + 5, 300))
            seqs = seqs + [c + c[:k - 1]]
            g = g + "N" + c + c[:k - 1]
        o = oracle_lib.Index.from_sequences(seqs, k, 1, 40)
        km, ct = o.export()
        idx = mtg.Index.from_kmers(km, ct, k)
        for mn, md, er in ((100, 10000, 0), (5, 300, 1), (20, 1500, 0)):
            src, tgt = [], []
            while len(src) < 24:
                p = rng.randrange(0, len(g) - k)
                s = g[p:p + k]
                tp = rng.randrange(0, len(g) - k)
                t = g[tp:tp + k]
                if "N" in s or "N" in t:
                    continue
                src.append(s if rng.random() < 0.5 else _rc(s))
                tgt.append(t)
            got = idx.stage_a(src, tgt, mtg.FillParams(max_nodes=mn, max_depth=md, end_rule_nonbranching=er))
            for s, t, c in zip(src, tgt, got):
                assert c == o.stage_a(s, t, oracle_lib.default_params(max_nodes=mn, max_depth=md, end_rule_nonbranching=er))[0], (seed, k, s, t, mn, md, er)
        idx.close()
        o.close()


@pytest.mark.parametrize("chunk,parts,host_paths,own_copy", [(37, 3, False, False), (1000, 8, False, False), (0, 1, True, False), (0, 1, False, True)])
def test_batches_split_into_launches_and_parts(mtg, tmp_path, monkeypatch, chunk, parts, host_paths, own_copy):
    """the batch machinery under test hooks: several traversal launches per batch (MTG_MAX_CHUNK), post-processing parts per launch
    (MTG_POST_PARTS), path enumeration left to the host (MTG_HOST_PATHS), walks that copy their long runs themselves instead of leaving
    them to k_copy (MTG_NO_DEFER); results must not depend on any of them"""
    from tests.test_emu_parity import _allelic_inserts_case, _diploid_case, _edge_case_run
    if own_copy:
        monkeypatch.setenv("MTG_NO_DEFER", "1")
    if chunk:
        monkeypatch.setenv("MTG_MAX_CHUNK", str(chunk))
    monkeypatch.setenv("MTG_POST_PARTS", str(parts))
    if host_paths:
        monkeypatch.setenv("MTG_HOST_PATHS", "1")
    (tmp_path / "a").mkdir(); (tmp_path / "d").mkdir(); (tmp_path / "e").mkdir()
    _allelic_inserts_case(mtg, tmp_path / "a", 60)
    _diploid_case(mtg, tmp_path / "d", 100)
    _edge_case_run(mtg, tmp_path / "e")


def test_serialised_batch_in_place_and_fallback(mtg, tmp_path):
    """mtg_fill_batch_serial: the sequences of a batch decoded straight into the caller's buffer (common path, several parts) and laid out
    again when multi-contig gaps are present; both equal the concatenation of the per-gap results"""
    from mindthegap_amd.synth import SynthSet
    from tests.test_emu_parity import _rand_seq
    import random
    S = SynthSet(nseq=40000, n_sites=40000, seed=9)
    import torch
    dev = torch.device("cuda", 0)
    w = torch.from_numpy(S.words.view(np.int64)).to(dev)
    wo = torch.from_numpy(S.word_off.view(np.int64)).to(dev)
    ln = torch.from_numpy(S.lens.view(np.int32)).to(dev)
    idx = mtg.Index.from_packed_device(w.data_ptr(), wo.data_ptr(), ln.data_ptr(), S.nseq, S.total_kmers_upper_bound, 31, 3, 40)
    gaps = []
    for i in range(S.n_sites):
        l, r, _ = S.site(i)
        gaps.append(mtg.Gap(l, r, [(r, S.site_name(i), False)]))
    prep = mtg.Index.prepare_gaps(gaps)
    out = np.empty(64 << 20, dtype=np.uint8)
    h, nf, nb = idx.fill_prepared_serial(prep, out)
    got = out[:nb].tobytes()
    idx.free_results(h)
    assert (nf == 1).all()
    assert got == b"".join(S.site(i)[2].encode() + b"\0" for i in range(S.n_sites))
    # too small a buffer is refused
    with pytest.raises(mtg.MtgError):
        idx.fill_prepared_serial(prep, np.empty(1 << 20, dtype=np.uint8))
    # the same with the sequences left in device memory (mtg_fill_prepared_serial_device): the caller's device buffer holds the same bytes
    batch = idx.prepare_batch(prep)
    dbuf = torch.full((64 << 20,), 0xEE, dtype=torch.uint8, device=dev)
    h, nf, nb2 = idx.fill_prepared_serial_device(batch, dbuf.data_ptr(), dbuf.numel())
    idx.free_results(h)
    assert nb2 == nb and (nf == 1).all() and dbuf[:nb2].cpu().numpy().tobytes() == got
    dbuf.fill_(0xEE)  # and with a host copy next to it (the multi-GPU gather: device buffer for RCCL, host buffer for the rank's own results)
    hcopy = torch.empty(dbuf.numel(), dtype=torch.uint8).pin_memory()
    h, nf, nb3 = idx.fill_prepared_serial_device(batch, dbuf.data_ptr(), dbuf.numel(), host_out=hcopy.numpy())
    idx.free_results(h)
    assert nb3 == nb and dbuf[:nb3].cpu().numpy().tobytes() == got and hcopy.numpy()[:nb3].tobytes() == got
    with pytest.raises(mtg.MtgError):
        idx.fill_prepared_serial_device(batch, dbuf.data_ptr(), 1 << 20)
    batch.close()
    idx.close()
    # multi-contig gaps: the fallback lays the sequences out again; compare with the plain results
    rng = random.Random(3)
    seqs, sites = [], []
    for i in range(40):
        L, R = _rand_seq(rng, 300), _rand_seq(rng, 300)
        a, b = _rand_seq(rng, 200 + i), _rand_seq(rng, 300 + i)
        seqs += [L + a + R, L + b + R] if i % 2 else [L + a + R]
        sites.append((L[-31:], R[:31]))
    from tests import oracle_lib
    o = oracle_lib.Index.from_sequences(seqs, 31, 3, 40)
    km, ct = o.export()
    idx2 = mtg.Index.from_kmers(km, ct, 31)
    gaps2 = [mtg.Gap(l, r, [(r, "t%d" % i, False)]) for i, (l, r) in enumerate(sites)]
    plain = idx2.fill_batch(gaps2)
    h, nf, nb = idx2.fill_prepared_serial(mtg.Index.prepare_gaps(gaps2), out)
    got = out[:nb].tobytes()
    idx2.free_results(h)
    want = b"".join(f["seq"].encode() + b"\0" for g in plain for f in g["filled"])
    assert got == want and max(len(g["filled"]) for g in plain) == 2
    batch2 = idx2.prepare_batch(gaps2)  # device buffer: the re-laid form goes through the host and back
    h, nf, nb2 = idx2.fill_prepared_serial_device(batch2, dbuf.data_ptr(), dbuf.numel())
    idx2.free_results(h)
    assert dbuf[:nb2].cpu().numpy().tobytes() == want
    dbuf.fill_(0xEE)
    h, nf, nb3 = idx2.fill_prepared_serial_device(batch2, dbuf.data_ptr(), dbuf.numel(), host_out=hcopy.numpy())
    idx2.free_results(h)
    assert dbuf[:nb3].cpu().numpy().tobytes() == want and hcopy.numpy()[:nb3].tobytes() == want
    batch2.close()
    idx2.close()
    o.close()


def test_concurrent_batches_on_two_indexes(mtg, full_idx, ctg_idx, golden_dir):
    """four host threads, two per index, filling batches at the same time: two batches of an index run side by side on two of its
    workspaces and streams, the traversal kernel's constants are per module (one set per workspace number, locked while a traversal reads it), the worker pool and the result
    cache are shared: every result equals the single-threaded one"""
    import threading
    _, g1, _, _ = full_idx
    _, g2, _, _ = ctg_idx
    recs = _fasta(os.path.join(golden_dir, "full_test", "gold.breakpoints"))
    gaps1 = [mtg.Gap(recs[i][1], recs[i + 1][1], [(recs[i + 1][1], recs[i][0].split()[0], False)]) for i in range(0, len(recs), 2)] * 40
    ctgs = _fasta(os.path.join(golden_dir, "data", "contigs.fasta"))
    gaps2 = []
    for n, c in ctgs:  # contig mode in miniature: from the end of a contig to the starts of the others
        tg = [(d[:31], m, False) for m, d in ctgs if m != n]
        gaps2.append(mtg.Gap(c[-31:], "".join(t[0] for t in tg), tg))
    gaps2 = gaps2 * 10
    want1, want2 = g1.fill_batch(gaps1), g2.fill_batch(gaps2)
    assert any(r["filled"] for r in want1) and any(r["filled"] for r in want2)
    errors = []

    def work(g, gaps, want):
        try:
            for _ in range(15):
                if g.fill_batch(gaps) != want:
                    errors.append("mismatch")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=work, args=a) for a in ((g1, gaps1, want1), (g2, gaps2, want2), (g1, gaps1, want1), (g2, gaps2, want2))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:3]


def test_abi_edge_cases(mtg):
    from tests.test_emu_parity import _abi_edge_cases
    _abi_edge_cases(mtg)


@pytest.mark.parametrize("k", [31, 21, 16])
def test_snp_fast_path_adversarial_on_device(mtg, k):
    """the loci of the CPU-side adversarial test (substitutions alone, in pairs, in palindromes and tandem repeats, next to indels and
    tips, three alleles, repeated loci) through k_stage_a, whose SNP fast path has no cross-check on the device: contigs == oracle"""
    import random
    from tests import oracle_lib
    from tests.test_emu_parity import _snp_case
    rng = random.Random(31337 + k)
    for case in range(132):
        kind = case % 11
        a, alleles = _snp_case(rng, k, kind)
        seqs = []
        for i, x in enumerate(alleles):
            seqs += [x] * (1 + (case + i) % 3)
        o = oracle_lib.Index.from_sequences(seqs, k, 1, 40)
        km, ct = o.export()
        idx = mtg.Index.from_kmers(km, ct, k)
        src = [a[:k], _rc(a[-k:])] + ([a[k // 2:k // 2 + k]] if len(a) > 3 * k else [])
        tgt = [a[-k:]] * len(src)
        for er in (0, 1):
            got = idx.stage_a(src, tgt, mtg.FillParams(end_rule_nonbranching=er))
            for s, c in zip(src, got):
                assert c == o.stage_a(s, a[-k:], oracle_lib.default_params(end_rule_nonbranching=er))[0], (case, kind, k, s, er, seqs)
        idx.close()
        o.close()


@pytest.mark.parametrize("variant,err", [("E0", 0.0), ("E1", 0.001)])
def test_config2_full_size(mtg, tmp_path, monkeypatch, variant, err):
    """BASELINE config 2 at FULL size: 5 Mbp donor as 1000 x 5 kb sequences, 1000 insertion sites, 30x 150-bp simulated reads
    (1.06 M reads; E0 error-free, E1 0.1 % substitutions), index built from the reads by `-in` with -abundance-min 3 (k-mer counting on
    the device), `MindTheGap fill` outputs byte-identical to the CPU oracle: FASTA with headers, info, VCF body.
    Mirrors /root/reference/test/simple_full_test.sh:128-163 (index from reads, fill, compare with the expected files)."""
    from mindthegap_amd.synth import SynthSet, simulate_reads
    from tests import oracle_lib
    S = SynthSet(nseq=1000, n_sites=1000, seed=1)
    bk = str(tmp_path / "s.breakpoints")
    S.write_breakpoints(bk)
    reads = str(tmp_path / (variant + ".fa"))
    n = simulate_reads(S, reads, 30, 150, err, seed=5)
    assert n > 1000000
    assert mtg.fill_main(["-in", reads, "-bkpt", bk, "-abundance-min", "3", "-out", str(tmp_path / "hip")]) == 0
    o = oracle_lib.Index.from_files([reads], 31, 3)
    st = o.fill_files("bkpt", bk, str(tmp_path / "cpu"), params=oracle_lib.default_params(nb_cores=1))
    assert st["records"] == 1000
    for ext in (".insertions.fasta", ".info.txt"):
        assert _read(str(tmp_path / "hip") + ext) == _read(str(tmp_path / "cpu") + ext), (variant, ext)
    assert _vcf_body(str(tmp_path / "hip.insertions.vcf")) == _vcf_body(str(tmp_path / "cpu.insertions.vcf")), variant
    seqs = [l for l in _read(str(tmp_path / "hip.insertions.fasta")).splitlines() if not l.startswith(">")]
    if err == 0.0:  # error-free reads: the graph is the donor's, every fill is exactly the inserted sequence
        assert seqs == [S.site(i)[2] for i in range(S.n_sites)]
    else:
        assert len(seqs) >= 990
    # the index the CLI left behind (written from the device tables) loads again -- its records go to the device piece by piece, straight
    # from the file (here in pieces of 700 000 k-mers) -- and holds the oracle's solid k-mers; the tool run on it writes the same files
    monkeypatch.setenv("MTG_LOAD_PIECE", "700000")
    g = mtg.Index.load(str(tmp_path / "hip.mtgidx"))
    assert g.info()["nb_solid_kmers"] == len(o)
    km, ct = o.export()
    sel = np.random.default_rng(1).choice(len(km), 20000, replace=False)
    assert (g.abundance(km[sel]) == np.minimum(ct[sel], 255)).all()
    g.close()
    assert mtg.fill_main(["-graph", str(tmp_path / "hip.mtgidx"), "-bkpt", bk, "-out", str(tmp_path / "again")]) == 0
    for ext in (".insertions.fasta", ".info.txt"):
        assert _read(str(tmp_path / "again") + ext) == _read(str(tmp_path / "hip") + ext), (variant, ext)
    o.close()


def test_index_replica_and_tool_on_replicas(mtg, tmp_path, monkeypatch):
    """mtg_index_replicate (device-to-device copy of tables and unitig store; here onto the same device, the only one of the box) gives an
    index that answers and fills like its source; and the tool run with small batches (MTG_CLI_BATCH) writes the same files as with one"""
    import ctypes as C
    from mindthegap_amd.synth import SynthSet
    from tests import oracle_lib
    from tests.test_emu_parity import _edge_case_run
    S = SynthSet(nseq=300, n_sites=300, seed=5)
    o = oracle_lib.Index.from_sequences([S.ascii(j) for j in range(S.nseq)], 31, 3, 40)
    km, ct = o.export()
    a = mtg.Index.from_kmers(km, ct, 31)
    h = C.c_void_p()
    assert a.lib.mtg_index_replicate(a.h, 0, C.byref(h)) == 0
    b = mtg.Index(h)
    assert b.info()["nb_unitigs"] == a.info()["nb_unitigs"] > 0
    q = np.concatenate([km, np.random.default_rng(3).integers(0, 1 << 62, 5000, dtype=np.uint64)])
    assert (a.abundance(q) == b.abundance(q)).all()
    sa, pa = a.neighbors(q)
    sb, pb = b.neighbors(q)
    assert (sa == sb).all() and (pa == pb).all()
    gaps = []
    for i in range(S.n_sites):
        l, r, _ = S.site(i)
        gaps.append(mtg.Gap(l, r, [(r, S.site_name(i), False)]))
    ra, rb = a.fill_batch(gaps), b.fill_batch(gaps)
    assert ra == rb and [r["filled"][0]["seq"] for r in ra] == [S.site(i)[2] for i in range(S.n_sites)]
    a.close()
    assert b.fill_batch(gaps[:50]) == rb[:50]  # the replica does not depend on its source
    b.close(); o.close()
    monkeypatch.setenv("MTG_CLI_BATCH", "4")
    _edge_case_run(mtg, tmp_path)


def test_text_batches_marshalled_on_the_device(mtg):
    """mtg_fill_text: strings as offsets into one block of text, encoded by k_marshal_text / k_marshal_targets, against mtg_fill_batch on the
    same strings (lower case, N, keys and patterns shorter or longer than k, empty and multiple dictionaries, repeated anchors, reverse
    attempts); malformed batches are refused"""
    from tests import oracle_lib, text_cases
    text_cases.run(mtg, oracle_lib)


def test_first_stage_a_batch_larger_than_the_initial_dense_arrays(mtg):
    """the dense contig arrays of a fresh workspace start small (1 MB of words) and k_emit is queued before the totals of the launch are known:
    a FIRST stage-A batch with 6 MB of contigs (every gap hands all its words over) must come back complete -- k_emit skips what does not
    fit, the host grows the arrays and emits again -- and nothing outside them may have been written (the fills afterwards are intact)"""
    from mindthegap_amd.synth import SynthSet
    from tests import oracle_lib
    S = SynthSet(nseq=2400, n_sites=2400, seed=21)
    o = oracle_lib.Index.from_sequences([S.ascii(j) for j in range(S.nseq)], 31, 3, 0)
    km, ct = o.export()
    g = mtg.Index.from_kmers(km, ct, 31)  # a fresh index: fresh workspaces
    src, tgt = [], []
    for i in range(S.n_sites):
        l, r, _ = S.site(i)
        src.append(l)
        tgt.append("ACGT" * 20)  # never found: the walk runs to the end of its donor sequence (a contig of 2-5 kb per gap)
    got = g.stage_a(src, tgt)
    total_nt = sum(len(c) for cs in got for c in cs)
    assert total_nt > 5_000_000
    for i in list(range(0, 60)) + list(range(S.n_sites - 60, S.n_sites)) + list(range(700, 2400, 97)):
        assert got[i] == o.stage_a(src[i], tgt[i])[0], i
    # the same workspaces afterwards: ordinary fills, every one identical to its inserted sequence
    gaps, want = [], []
    for i in range(600):
        l, r, ins = S.site(i)
        gaps.append(mtg.Gap(l, r, [(r, S.site_name(i), False)]))
        want.append(ins)
    res = g.fill_batch(gaps)
    assert [r["filled"][0]["seq"] if r["filled"] else None for r in res] == want
    g.close(); o.close()


LAUNCH_SEQUENCES = [
    {"MTG_ROUNDS": "6"},                                   # six rounds of bubble kernel (one lane per bubble) + resumed walks, then the finishing kernel
    {"MTG_ROUNDS": "3", "MTG_FINISH_G": "16"},             # three rounds, the tail by k_finish<16>
    {"MTG_ROUNDS": "0", "MTG_FINISH_G": "64"},             # every parked gap straight to k_finish<64>
    {"MTG_ROUNDS": "0", "MTG_FINISH_G": "8"},              # k_finish<8>: the grid's whole workgroups and k_finish_lane's first entry must meet (round-4 advisor)
    {"MTG_FINISH_G": "1"},                                 # one lane per parked gap (k_finish_lane)
    {"MTG_LIGHT_WALK": "1"},                               # the first walk by the light kernel (every branching node parks), then whatever the park share asks for
    {"MTG_LIGHT_WALK": "1", "MTG_ROUNDS": "2"},            # light walk, two rounds of bubble kernel + resumed full walks, the finishing kernel
    {"MTG_LIGHT_WALK": "0"},                               # never the light kernel
]


@pytest.mark.gpu
@pytest.mark.parametrize("seq", LAUNCH_SEQUENCES, ids=lambda e: ",".join("%s=%s" % kv for kv in sorted(e.items())))
def test_bubble_tests_under_every_launch_sequence(seq):
    """the walk / bubble / finishing kernels are chosen per launch from what the previous launch parked; every sequence the library can
    queue -- rounds of bubble kernel and resumed walks, the finishing kernel with 8, 16 or 64 lanes per gap or one -- must give the oracle's contigs and files on the tests whose walks cross bubbles.  The switches
    are read once per process: each sequence runs the selected tests in a process of its own."""
    env = dict(os.environ, **seq)
    sel = "fuzz_on_device or adversarial or diploid or allelic or tier_retry or golden or synthetic_sites"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_micro_cases.py"), "-m", "gpu", "-x", "-q",
                        "-p", "no:cacheprovider", "-k", "(%s) and not launch_sequence" % sel], env=env, capture_output=True, text=True, timeout=1500)
    import re
    summary = re.findall(r"(\d+) passed", r.stdout)  # (the tool's own report on stdout may follow pytest's line)
    assert r.returncode == 0 and summary and int(summary[-1]) >= 10 and not re.search(r"\d+ (failed|error)", r.stdout), (seq, r.stdout[-1500:], r.stderr[-800:])


@pytest.mark.gpu
def test_partitioned_junction_table_equals_the_scattered_insertion(mtg):
    """Graph::create from packed sequences (src/Filler.cpp:172-213): the junction table built partition by partition in LDS (round 6: BUILD_PARTITIONED=1, the
    default from 2^27 junction positions on) and by scattered insertion (=0) must describe the same graph -- statistics, every neighbourhood / abundance
    query on a sample of solid and absent k-mers, the same fills -- on a donor with repeats (sequences that share stretches, a palindrome, a tandem repeat)."""
    import torch
    from mindthegap_amd.synth import SynthSet
    S = SynthSet(nseq=3000, n_sites=400, seed=23, k=31, het_snps=2)  # diploid: every shared unitig occurs in two sequences
    rng = np.random.default_rng(5)
    dev = torch.device("cuda", 0)
    pw, po, pl, pn = S.packed()
    # ... and one 64-mer 6 000 times (between stretches of 40 random nucleotides): every occurrence of each of its junctions lands in ONE segment of the
    # partitioned table, whose list holds ~4 600 records -- the rest goes through the list of leftovers and the ordinary insertion
    rep = rng.integers(0, 4, 64).astype(np.uint8)
    beads = [np.concatenate([np.concatenate([rep, rng.integers(0, 4, 40).astype(np.uint8)]) for _ in range(60)]) for _ in range(100)]
    ew, eo, el = _pack_codes(beads)
    pw = np.concatenate([pw, ew, np.zeros(2, dtype=np.uint64)])
    po = np.concatenate([po, eo + np.uint64(pw.size - ew.size - 2)])
    pl = np.concatenate([pl, el])
    pn = pn + len(beads)
    ub = int(np.maximum(pl.astype(np.int64) - 30, 0).sum())
    built = {}
    for part in ("0", "1"):
        mtg.tuning_set("BUILD_PARTITIONED", part)
        try:
            w = torch.from_numpy(pw.view(np.int64)).to(dev)
            wo = torch.from_numpy(po.view(np.int64)).to(dev)
            ln = torch.from_numpy(pl.view(np.int32)).to(dev)
            idx = mtg.Index.from_packed_device(w.data_ptr(), wo.data_ptr(), ln.data_ptr(), pn, ub, 31, 3, 0)
        finally:
            mtg.tuning_set("BUILD_PARTITIONED", None)
        names = [p["name"] for p in idx.build_profile()["phases"]]
        assert ("jt_build_segments" in names) == (part == "1"), names
        info = idx.info()
        # k-mers of the donor (solid) and random ones (absent, mostly)
        codes = S.codes(7)
        km = []
        for p0 in range(0, 2000, 3):
            v = 0
            for c in codes[p0:p0 + 31]:
                v = (v << 2) | int(c)
            rc = 0
            for c in codes[p0:p0 + 31][::-1]:
                rc = (rc << 2) | (int(c) ^ 2)
            km.append(min(v, rc))
        km = np.array(km + [int(x) for x in rng.integers(0, 2**62, 500)], dtype=np.uint64)
        ab = idx.abundance(km)
        succ, pred = idx.neighbors(km)
        gaps = []
        for i in range(S.n_sites):
            l, r, _ = S.site(i)
            gaps.append(mtg.Gap(l, r, [(r, S.site_name(i), False)]))
        res = idx.fill_batch(gaps)
        built[part] = ({k2: info[k2] for k2 in ("nb_solid_kmers", "nb_branching", "nb_unitigs", "nb_kmers_outside_unitigs")}, ab.tolist(), succ.tolist(), pred.tolist(),
                       [[(f["seq"], f["nb_errors_in_anchor"], f["qual"]) for f in r["filled"]] for r in res])
        idx.close()
        del w, wo, ln
    assert built["0"] == built["1"]
    assert built["1"][0]["nb_solid_kmers"] > 5 * 10**6 and sum(1 for r in built["1"][4] if r) > 300


def _pack_codes(seqs):
    """2-bit code arrays -> (words, word offsets, lengths): 32 nucleotides a word, lowest bits first, every sequence from a word boundary (+ one spare word)"""
    words, off, lens = [], [], []
    at = 0
    sh = np.arange(32, dtype=np.uint64) * np.uint64(2)
    for c in seqs:
        n = len(c)
        nw = (n + 31) // 32 + 1
        buf = np.zeros(nw * 32, dtype=np.uint64)
        buf[:n] = c
        words.append((buf.reshape(nw, 32) << sh[None, :]).sum(axis=1, dtype=np.uint64))
        off.append(at)
        lens.append(n)
        at += nw
    return np.concatenate(words), np.array(off, dtype=np.uint64), np.array(lens, dtype=np.uint32)


@pytest.mark.gpu
def test_chains_found_by_position_equal_the_walked_ones(mtg):
    """Graph::create from packed sequences (src/Filler.cpp:172-213), the unitigs of the index: found by position in the sequences (round 6: BUILD_POSITIONAL=1, the
    default) and walked on the junction table (=0) must give the same index -- statistics, contains / abundance / neighbourhood of EVERY k-mer of every sequence
    (each goes through its unitig in the store) and of absent ones, the same fills -- on a diploid donor plus the shapes that decide who stores a chain: a
    sequence twice, a sequence and its reverse complement, two overlapping windows of one stretch (no sequence holds the chain whole: it is walked), a
    substitution (branching), a tandem repeat, a homopolymer, a hairpin (a stretch followed by its reverse complement), a cycle written twice round,
    sequences of k - 1, k and k + 1 nucleotides, one chain of 70 000 nucleotides (tiles of k_pos_plan without a stop), a string of 400 short
    stretches between copies of one k-mer (hundreds of stops a tile)."""
    import torch
    from mindthegap_amd.synth import SynthSet
    k = 31
    S = SynthSet(nseq=2000, n_sites=300, seed=29, k=k, het_snps=2)
    rng = np.random.default_rng(11)
    rc = lambda c: (c[::-1] ^ 2).astype(np.uint8)
    g = rng.integers(0, 4, 6000).astype(np.uint8)
    h = rng.integers(0, 4, 900).astype(np.uint8)
    snp = h.copy(); snp[450] ^= 1
    cyc = rng.integers(0, 4, 211).astype(np.uint8)
    long_one = rng.integers(0, 4, 70000).astype(np.uint8)  # seventeen tiles of k_pos_plan with no stop between the first and the last
    linker = rng.integers(0, 4, k).astype(np.uint8)
    beads = np.concatenate([np.concatenate([linker, rng.integers(0, 4, 9 + i % 23).astype(np.uint8)]) for i in range(400)])  # a stop every few dozen positions: hundreds a tile
    extra = [long_one, beads, g[:1500], g[:1500], rc(g[1500:2600]), g[1500:2600], g[2600:4200], g[3600:5200], h, snp,
             np.tile(np.array([0, 1], dtype=np.uint8), 100), np.zeros(120, dtype=np.uint8), np.concatenate([g[5200:5500], rc(g[5200:5500])]),
             np.concatenate([cyc, cyc, cyc[:k]]), g[5600:5600 + k - 1], g[5700:5700 + k], g[5800:5800 + k + 1], rc(g[5800:5800 + k + 1])]
    ew, eo, el = _pack_codes(extra)
    pw, po, pl, pn = S.packed()
    words = np.concatenate([pw, ew, np.zeros(2, dtype=np.uint64)])
    off = np.concatenate([po, eo + np.uint64(pw.size)])
    lens = np.concatenate([pl, el])
    nseq = pn + len(extra)
    ub = int(np.maximum(lens.astype(np.int64) - (k - 1), 0).sum())
    # every k-mer of the extra sequences and of a sample of the donor's, canonical; and absent ones
    def kmers_of(c):
        out = []
        for p0 in range(0, len(c) - k + 1):
            v = 0
            for x in c[p0:p0 + k]:
                v = (v << 2) | int(x)
            r = 0
            for x in c[p0:p0 + k][::-1]:
                r = (r << 2) | (int(x) ^ 2)
            out.append(min(v, r))
        return out
    km = []
    for c in extra:
        km += kmers_of(c)
    for j in (0, 1, 2, 3, 1001):
        km += kmers_of(S.codes(j)[:1200])
    km = np.array(km + [int(x) for x in rng.integers(0, 2**62, 500)], dtype=np.uint64)
    dev = torch.device("cuda", 0)
    built = {}
    for pos in ("0", "1"):
        mtg.tuning_set("BUILD_POSITIONAL", pos)
        try:
            w = torch.from_numpy(words.view(np.int64)).to(dev)
            wo = torch.from_numpy(off.view(np.int64)).to(dev)
            ln = torch.from_numpy(lens.view(np.int32)).to(dev)
            idx = mtg.Index.from_packed_device(w.data_ptr(), wo.data_ptr(), ln.data_ptr(), nseq, ub, k, 3, 40)
        finally:
            mtg.tuning_set("BUILD_POSITIONAL", None)
        names = [p["name"] for p in idx.build_profile()["phases"]]
        assert ("pos_plan" in names) == (pos == "1"), names
        info = idx.info()
        has = idx.contains(km)
        ab = idx.abundance(km)
        succ, pred = idx.neighbors(km)
        gaps = []
        for i in range(S.n_sites):
            l, r, _ = S.site(i)
            gaps.append(mtg.Gap(l, r, [(r, S.site_name(i), False)]))
        res = idx.fill_batch(gaps)
        built[pos] = ({k2: info[k2] for k2 in ("nb_solid_kmers", "nb_branching", "nb_unitigs", "nb_kmers_outside_unitigs", "unitig_bytes")}, has.tolist(), ab.tolist(), succ.tolist(), pred.tolist(),
                      [[(f["seq"], f["nb_errors_in_anchor"], f["qual"]) for f in r["filled"]] for r in res])
        idx.close()
        del w, wo, ln
    for a, b, what in zip(built["0"], built["1"], ("info", "contains", "abundance", "successors", "predecessors", "fills")):
        assert a == b, what
    assert all(built["1"][1][:len(km) - 500]) and sum(1 for r in built["1"][5] if r) > 200


@pytest.mark.gpu
def test_long_sequences_taken_in_pieces_give_the_same_index(mtg):
    """Graph::create from packed sequences (src/Filler.cpp:172-213): a sequence of more than 131 072 nucleotides is taken in pieces of 65 536 that share
    k - 1 nucleotides (round 6: the streaming kernels give a sequence to a workgroup, and a donor of a few chromosomes would keep a few busy) -- the same
    k-mers once, the junction at a cut from its two pieces -- and must give the index of the unsplit input (NO_SPLIT_LONG=1): statistics, contains /
    abundance / neighbourhood of the k-mers around every cut and of samples elsewhere, fills of sites that lie in the long sequences."""
    import torch
    from mindthegap_amd.synth import SynthSet
    k = 31
    S = SynthSet(nseq=300, n_sites=100, seed=41, k=k)
    rng = np.random.default_rng(17)
    rep = rng.integers(0, 4, 3000).astype(np.uint8)
    a = rng.integers(0, 4, 400000).astype(np.uint8)
    a[65536 - 1500:65536 + 1500] = rep          # a repeat across the first cut ...
    b = rng.integers(0, 4, 140000).astype(np.uint8)
    b[20000:23000] = rep                        # ... and inside another sequence: the chains around the cut branch there
    extra = [a, b, rng.integers(0, 4, 131072).astype(np.uint8), rng.integers(0, 4, 131073).astype(np.uint8), rng.integers(0, 4, 2 * 65536 + 65536 + 20).astype(np.uint8)]
    ew, eo, el = _pack_codes(extra)
    pw, po, pl, pn = S.packed()
    words = np.concatenate([pw, ew, np.zeros(2, dtype=np.uint64)])
    off = np.concatenate([po, eo + np.uint64(pw.size)])
    lens = np.concatenate([pl, el])
    nseq = pn + len(extra)
    ub = int(np.maximum(lens.astype(np.int64) - (k - 1), 0).sum())

    def kmers_of(c):
        out = []
        for p0 in range(0, len(c) - k + 1):
            v = 0
            for x in c[p0:p0 + k]:
                v = (v << 2) | int(x)
            r = 0
            for x in c[p0:p0 + k][::-1]:
                r = (r << 2) | (int(x) ^ 2)
            out.append(min(v, r))
        return out
    km = []
    for c in extra:
        for cut in range(65536, len(c), 65536):
            km += kmers_of(c[max(0, cut - 80):cut + 80])
        km += kmers_of(c[:200]) + kmers_of(c[-200:]) + kmers_of(c[len(c) // 3:len(c) // 3 + 150])
    # the unitig behind the repeat of the first sequence has 333 000 k-mers: the link and the abundances take it in pieces of 32 768 k-mers, from either end
    for j in range(1, 11):
        for at in (65536 + 1500 + j * 32768, 400000 - (k - 1) - j * 32768):
            km += kmers_of(extra[0][at - 70:at + 70])
    km = np.array(km + [int(x) for x in rng.integers(0, 2**62, 300)], dtype=np.uint64)
    dev = torch.device("cuda", 0)
    built = {}
    for whole in ("1", None):
        mtg.tuning_set("NO_SPLIT_LONG", whole)
        try:
            w = torch.from_numpy(words.view(np.int64)).to(dev)
            wo = torch.from_numpy(off.view(np.int64)).to(dev)
            ln = torch.from_numpy(lens.view(np.int32)).to(dev)
            idx = mtg.Index.from_packed_device(w.data_ptr(), wo.data_ptr(), ln.data_ptr(), nseq, ub, k, 3, 40)
        finally:
            mtg.tuning_set("NO_SPLIT_LONG", None)
        info = idx.info()
        has, ab = idx.contains(km), idx.abundance(km)
        succ, pred = idx.neighbors(km)
        # sites in the long sequences: from 31 nucleotides before a cut to 31 after it, and across the repeat
        gaps = []
        for c in (extra[0], extra[4]):
            for cut in range(65536, len(c) - 2000, 65536):
                for l0, r0 in ((cut - 400, cut + 300), (cut - 31, cut + 31), (cut + 5, cut + 900)):
                    l = "".join("ACTG"[x] for x in c[l0 - k:l0]); r = "".join("ACTG"[x] for x in c[r0:r0 + k])
                    gaps.append(mtg.Gap(l, r, [(r, "s%d_%d" % (cut, l0), False)]))
        for i in range(S.n_sites):
            l, r, _ = S.site(i)
            gaps.append(mtg.Gap(l, r, [(r, S.site_name(i), False)]))
        res = idx.fill_batch(gaps)
        built[whole] = ({k2: info[k2] for k2 in ("nb_solid_kmers", "nb_branching", "nb_unitigs", "nb_kmers_outside_unitigs", "unitig_bytes")}, has.tolist(), ab.tolist(), succ.tolist(), pred.tolist(),
                        [[(f["seq"], f["nb_errors_in_anchor"], f["qual"]) for f in r["filled"]] for r in res])
        idx.close()
        del w, wo, ln
    for x, y, what in zip(built["1"], built[None], ("info", "contains", "abundance", "successors", "predecessors", "fills")):
        assert x == y, what
    assert all(built[None][1][:len(km) - 300]) and sum(1 for r in built[None][5] if r) > 100


@pytest.mark.gpu
def test_two_reached_targets_under_one_name_on_device(mtg):
    from tests.test_emu_parity import _duplicate_target_names_case
    _duplicate_target_names_case(mtg)
